# Test items for DiffPointRasterisationAMDGPUExt, to sit next to the reference's test/cuda.jl
# (/root/reference/test/cuda.jl:2-74 has the same three argument sets for CuArray).  They use the
# reference's own fixtures (`include("data.jl")`, module D, /root/reference/test/data.jl).
# UNEXECUTED SOURCE: no Julia runtime here; see tests/test_parity_gpu.py for what runs.

@testitem "AMDGPU forward" begin
    using Adapt, AMDGPU
    AMDGPU.allowscalar(false)
    include("data.jl")
    include("util_amdgpu.jl")
    have_gpu = AMDGPU.functional()

    cases = (
        # 3-D -> 3-D, every optional argument given
        (D.grid_size_3d, D.more_points, D.rotations_static, D.translations_3d_static,
         D.backgrounds, D.weights, D.more_point_weights),
        # defaults (FillArrays Zeros / Ones travel as NULL pointers)
        (D.grid_size_3d, D.more_points, D.rotations_static, D.translations_3d_static),
        # 3-D -> 2-D projection
        (D.grid_size_2d, D.more_points, D.projections_static, D.translations_2d_static,
         D.backgrounds, D.weights, D.more_point_weights),
    )
    for args in cases
        @test roc_cpu_agree(raster, args...) skip = !have_gpu
    end
end

@testitem "AMDGPU backward" begin
    using Adapt, AMDGPU
    AMDGPU.allowscalar(false)
    include("data.jl")
    include("util_amdgpu.jl")
    have_gpu = AMDGPU.functional()

    g3 = randn(D.grid_size_3d..., D.batch_size)
    g2 = randn(D.grid_size_2d..., D.batch_size)
    cases = (
        (g3, D.more_points, D.rotations_static, D.translations_3d_static, D.backgrounds,
         D.weights, D.more_point_weights),
        (g3, D.more_points, D.rotations_static, D.translations_3d_static),
        (g2, D.more_points, D.projections_static, D.translations_2d_static, D.backgrounds,
         D.weights, D.more_point_weights),
    )
    for args in cases
        @test roc_cpu_agree(raster_pullback!, args...) skip = !have_gpu
    end
end

@testitem "AMDGPU single pose pullback" begin
    # the CUDA extension has no single-image pullback (ext/DiffPointRasterisationCUDAExt.jl:213-228);
    # here it is a batch of one
    using Adapt, AMDGPU
    include("data.jl")
    include("util_amdgpu.jl")
    have_gpu = AMDGPU.functional()
    g = randn(D.grid_size_3d...)
    @test roc_cpu_agree(raster_pullback!, g, D.more_points, D.rotation, D.translation_3d) skip = !have_gpu
end

@testitem "AMDGPU mixed element types" begin
    # Float32 grid with Float64 points: promoted like ext/DiffPointRasterisationCUDAExt.jl:246
    using Adapt, AMDGPU
    include("data.jl")
    include("util_amdgpu.jl")
    have_gpu = AMDGPU.functional()
    g = randn(Float32, D.grid_size_3d..., D.batch_size)
    @test roc_cpu_agree(raster_pullback!, g, D.more_points, D.rotations_static,
                        D.translations_3d_static) skip = !have_gpu
end

@testitem "AMDGPU rrule shares the binning" begin
    using Adapt, AMDGPU, ChainRulesCore
    include("data.jl")
    include("util_amdgpu.jl")
    if AMDGPU.functional()
        pts = adapt(ROCArray, D.more_points)
        out, pb = ChainRulesCore.rrule(raster, D.grid_size_3d, pts, D.rotation, D.translation_3d)
        @test Array(out) ≈ raster(D.grid_size_3d, D.more_points, D.rotation, D.translation_3d)
        g = randn(D.grid_size_3d...)
        ref = raster_pullback!(g, D.more_points, D.rotation, D.translation_3d)
        tangents = pb(adapt(ROCArray, g))
        @test Array(reinterpret(reshape, Float64, tangents[3])) ≈ ref.points
        @test Array(tangents[4]) ≈ ref.rotation
        @test Array(tangents[5]) ≈ ref.translation
        # a second call through the same closure re-bins instead of reusing a consumed binning
        again = pb(adapt(ROCArray, g))
        @test Array(again[4]) ≈ ref.rotation
    end
end

@testitem "AMDGPU sorted cloud with the coherence flag" begin
    # sort_points + raster_coherent! / raster_pullback_coherent! (not in the reference: dpr.h
    # dpr_sort_points_*, DPR_FLAG_COHERENT_POINTS) against the CPU path on the ORIGINAL order
    using Adapt, AMDGPU, StaticArrays, FillArrays
    include("data.jl")
    include("util_amdgpu.jl")
    if AMDGPU.functional()
        ext = Base.get_extension(DiffPointRasterisation, :DiffPointRasterisationAMDGPUExt)
        pts = adapt(ROCArray, D.more_points)
        sorted, _, perm = ext.sort_points(pts)
        p = Array(perm) .+ 1
        @test Array(sorted) == D.more_points[p]
        out = AMDGPU.zeros(Float64, D.grid_size_3d..., D.batch_size)
        B = D.batch_size
        ext.raster_coherent!(out, sorted, D.rotations_static, D.translations_3d_static,
                             Zeros(Float64, B), Ones(Float64, B), Ones(Float64, length(p)))
        @test Array(out) ≈ raster(D.grid_size_3d, D.more_points, D.rotations_static, D.translations_3d_static)
        g = randn(D.grid_size_3d..., B)
        ref = raster_pullback!(g, D.more_points, D.rotations_static, D.translations_3d_static)
        got = ext.raster_pullback_coherent!(adapt(ROCArray, g), sorted, D.rotations_static,
                                            D.translations_3d_static, Ones(Float64, B), Ones(Float64, length(p)))
        back = similar(ref.points)
        back[:, p] = Array(got.points)   # ds_dpoints[perm[i]] = ds_dpoints_sorted[i]
        @test back ≈ ref.points
        @test Array(got.rotation) ≈ ref.rotation
        @test Array(got.translation) ≈ ref.translation
    end
end
