# DiffPointRasterisationAMDGPUExt -- binding of libdpr.so (include/dpr.h) for AMDGPU.jl arrays.
#
# UNEXECUTED SOURCE: neither this container nor the GPU box has a Julia runtime, so this
# file has never been run.  It mirrors the structure of the reference's CUDA extension
# (ext/DiffPointRasterisationCUDAExt.jl:231-333): array-type-specialised methods of the two
# canonical signatures plus the two allocator hooks; everything above them
# (src/interface.jl, ext/DiffPointRasterisationChainRulesCoreExt.jl) is reused unchanged.
#
# Project.toml additions (next to CUDA, Project.toml:16-22):
#   [weakdeps]    AMDGPU = "21141c5a-9bdb-4563-92ae-f87d6854732e"
#   [extensions]  DiffPointRasterisationAMDGPUExt = "AMDGPU"
module DiffPointRasterisationAMDGPUExt

using DiffPointRasterisation, AMDGPU
using ArgCheck, FillArrays, StaticArrays

const libdpr = get(ENV, "LIBDPR", "libdpr.so")

const ROCOrFillVector{T} = Union{ROCVector{T},FillArrays.AbstractFill{T,1}}

suffix(::Type{Float32}) = :f32
suffix(::Type{Float64}) = :f64

# FillArrays defaults (src/interface.jl:368-394) travel as NULL pointers
devptr(::FillArrays.AbstractFill, ::Type{T}) where {T} = Ptr{T}(C_NULL)
devptr(a::ROCArray, ::Type{T}) where {T} = Ptr{T}(UInt(pointer(a)))

# Device buffer whose memory is the elements of `a` converted to T, in `a`'s own order
# (Vector{SMatrix} is already "B x column-major N_out x N_in").  Pose vectors may arrive as
# host SVector{1} wrappers (single-pose path, src/interface.jl:113-116): those are uploaded.
devbuf(a::ROCArray{<:Number}, ::Type{T}) where {T} = eltype(a) === T ? a : T.(a)
devbuf(a::ROCArray{<:StaticArray}, ::Type{T}) where {T} =
    eltype(eltype(a)) === T ? a : map(x -> T.(x), a)
devbuf(a::FillArrays.AbstractFill, ::Type) = a
devbuf(a::AbstractVector{<:StaticArray}, ::Type{T}) where {T} = ROCArray(map(x -> T.(x), collect(a)))
devbuf(a::AbstractVector{<:Number}, ::Type{T}) where {T} = ROCArray(T.(collect(a)))

function check(status::Cint)
    status == 0 && return nothing
    msg = unsafe_string(ccall((:dpr_last_error, libdpr), Cstring, ()))
    # the reference throws DimensionMismatch / ArgumentError from @argcheck (src/raster.jl:14-23)
    status == -2 ? throw(ArgumentError(msg)) : error("libdpr status $status: $msg")
end

function workspace(op::Integer, ::Type{T}, n_in, n_out, grid, P, B) where {T}
    g = collect(Int64, grid)
    nbytes = T === Float32 ?
        ccall((:dpr_workspace_bytes_f32, libdpr), Csize_t, (Cint, Cint, Cint, Cint, Ptr{Int64}, Int64, Int64), op, 0, n_in, n_out, g, P, B) :
        ccall((:dpr_workspace_bytes_f64, libdpr), Csize_t, (Cint, Cint, Cint, Cint, Ptr{Int64}, Int64, Int64), op, 0, n_in, n_out, g, P, B)
    return ROCVector{UInt8}(undef, max(nbytes, 16))
end

# ---- forward: canonical method of src/raster.jl:5-13 for ROCArray outputs ------------------
function DiffPointRasterisation.raster!(
    out::ROCArray{T,N_out_p1},
    points::ROCVector{<:StaticVector{N_in,T}},
    rotation::AbstractVector{<:StaticMatrix{N_out,N_in,<:Number}},
    translation::AbstractVector{<:StaticVector{N_out,<:Number}},
    background::AbstractVector{<:Number},
    out_weight::AbstractVector{<:Number},
    point_weight::AbstractVector{<:Number},
) where {T<:Union{Float32,Float64},N_in,N_out,N_out_p1}
    @argcheck N_out == N_out_p1 - 1 DimensionMismatch
    B = size(out, N_out_p1)
    @argcheck B == length(rotation) == length(translation) == length(background) == length(out_weight) DimensionMismatch
    P = length(points)
    @argcheck length(point_weight) == P
    rot, tr = devbuf(rotation, T), devbuf(translation, T)
    bg, ow, pw = devbuf(background, T), devbuf(out_weight, T), devbuf(point_weight, T)
    grid = collect(Int64, size(out)[1:N_out])
    ws = workspace(0, T, N_in, N_out, grid, P, B)
    stream = AMDGPU.stream()
    GC.@preserve out points rot tr bg ow pw ws begin
        st = if T === Float32
            ccall((:dpr_raster_f32, libdpr), Cint,
                (Ptr{Cvoid}, Cint, Cint, Ptr{Int64}, Int64, Int64, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{Cvoid}, Csize_t),
                stream.stream, N_in, N_out, grid, P, B, devptr(out, T), devptr(points, T),
                devptr(rot, T), devptr(tr, T), devptr(bg, T), devptr(ow, T), devptr(pw, T),
                Ptr{Cvoid}(UInt(pointer(ws))), length(ws))
        else
            ccall((:dpr_raster_f64, libdpr), Cint,
                (Ptr{Cvoid}, Cint, Cint, Ptr{Int64}, Int64, Int64, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{Cvoid}, Csize_t),
                stream.stream, N_in, N_out, grid, P, B, devptr(out, T), devptr(points, T),
                devptr(rot, T), devptr(tr, T), devptr(bg, T), devptr(ow, T), devptr(pw, T),
                Ptr{Cvoid}(UInt(pointer(ws))), length(ws))
        end
        check(st)
    end
    return out   # same array, asynchronous on the task's stream like the reference
end

# ---- pullback: mirror of ext/DiffPointRasterisationCUDAExt.jl:231-321 ------------------------
function DiffPointRasterisation.raster_pullback!(
    ds_dout::ROCArray{T,N_out_p1},
    points::ROCVector{<:StaticVector{N_in,T}},
    rotation::AbstractVector{<:StaticMatrix{N_out,N_in,<:Number}},
    translation::AbstractVector{<:StaticVector{N_out,<:Number}},
    background::ROCOrFillVector{<:Number},
    out_weight::ROCOrFillVector{<:Number},
    point_weight::ROCOrFillVector{<:Number},
    ds_dpoints::ROCMatrix{T},
    ds_drotation::ROCArray{T,3},
    ds_dtranslation::ROCMatrix{T},
    ds_dbackground::ROCVector{T},
    ds_dout_weight::ROCVector{T},
    ds_dpoint_weight::ROCVector{T},
) where {T<:Union{Float32,Float64},N_in,N_out,N_out_p1}
    batch_axis = axes(ds_dout, N_out_p1)
    @argcheck N_out == N_out_p1 - 1
    @argcheck batch_axis == axes(rotation, 1) == axes(translation, 1) == axes(background, 1) == axes(out_weight, 1)
    @argcheck batch_axis == axes(ds_drotation, 3) == axes(ds_dtranslation, 2) == axes(ds_dbackground, 1) == axes(ds_dout_weight, 1)
    P = length(points)
    @argcheck length(ds_dpoint_weight) == P
    B = length(batch_axis)
    rot, tr = devbuf(rotation, T), devbuf(translation, T)
    ow, pw = devbuf(out_weight, T), devbuf(point_weight, T)
    grid = collect(Int64, size(ds_dout)[1:N_out])
    ws = workspace(1, T, N_in, N_out, grid, P, B)
    GC.@preserve ds_dout points rot tr ow pw ws begin
        args = (AMDGPU.stream().stream, N_in, N_out, grid, P, B, devptr(ds_dout, T),
            devptr(points, T), devptr(rot, T), devptr(tr, T),
            devptr(ow, T), devptr(pw, T), devptr(ds_dpoints, T), devptr(ds_drotation, T),
            devptr(ds_dtranslation, T), devptr(ds_dbackground, T), devptr(ds_dout_weight, T),
            devptr(ds_dpoint_weight, T), Ptr{Cvoid}(UInt(pointer(ws))), length(ws))
        st = T === Float32 ?
            ccall((:dpr_raster_pullback_f32, libdpr), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Int64}, Int64, Int64, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{Cvoid}, Csize_t), args...) :
            ccall((:dpr_raster_pullback_f64, libdpr), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Int64}, Int64, Int64, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{Cvoid}, Csize_t), args...)
        check(st)
    end
    # same arrays, fixed field order (src/raster_pullback.jl:140-147; the rrule slices it
    # positionally, ext/DiffPointRasterisationChainRulesCoreExt.jl:23,70)
    return (;
        points=ds_dpoints,
        rotation=ds_drotation,
        translation=ds_dtranslation,
        background=ds_dbackground,
        out_weight=ds_dout_weight,
        point_weight=ds_dpoint_weight,
    )
end

# ---- allocator hooks: flat (un-slabbed) buffers, as ext/DiffPointRasterisationCUDAExt.jl:323-333
function DiffPointRasterisation.default_ds_dpoints_batched(
    points::ROCVector{<:AbstractVector{TP}}, N_in, batch_size
) where {TP<:Number}
    return similar(points, TP, (N_in, length(points)))
end

function DiffPointRasterisation.default_ds_dpoint_weight_batched(
    points::ROCVector{<:AbstractVector{<:Number}}, T, batch_size
)
    return similar(points, T)
end

# ---- optional: squared-error loss without materialising ds_dout -------------------------------
# `raster_residual_pullback!(out, target, scale, points, ...; loss)`: the README's explicit
# recipe (README.md:151-165: ds_dout = 2 .* (target .- raster(...)); raster_pullback!(...)) in
# one call to dpr_raster_residual_pullback_*; `out` is the result of raster! for the same
# arguments, `loss[b] = sum(abs2, out[.., b] - target[.., b])`.  Not part of the reference API.
function raster_residual_pullback!(
    out::ROCArray{T,N_out_p1},
    target::ROCArray{T,N_out_p1},
    scale::Real,
    points::ROCVector{<:StaticVector{N_in,T}},
    rotation::AbstractVector{<:StaticMatrix{N_out,N_in,<:Number}},
    translation::AbstractVector{<:StaticVector{N_out,<:Number}},
    out_weight::ROCOrFillVector{<:Number},
    point_weight::ROCOrFillVector{<:Number},
    loss::ROCVector{T},
    ds_dpoints::ROCMatrix{T},
    ds_drotation::ROCArray{T,3},
    ds_dtranslation::ROCMatrix{T},
    ds_dbackground::ROCVector{T},
    ds_dout_weight::ROCVector{T},
    ds_dpoint_weight::ROCVector{T},
) where {T<:Union{Float32,Float64},N_in,N_out,N_out_p1}
    @argcheck N_out == N_out_p1 - 1
    @argcheck size(out) == size(target)
    P, B = length(points), size(out, N_out_p1)
    @argcheck length(loss) == B
    rot, tr = devbuf(rotation, T), devbuf(translation, T)
    ow, pw = devbuf(out_weight, T), devbuf(point_weight, T)
    grid = collect(Int64, size(out)[1:N_out])
    ws = workspace(1, T, N_in, N_out, grid, P, B)
    GC.@preserve out target points rot tr ow pw ws begin
        args = (AMDGPU.stream().stream, N_in, N_out, grid, P, B, devptr(out, T), devptr(target, T),
            Float64(scale), devptr(points, T), devptr(rot, T), devptr(tr, T), devptr(ow, T),
            devptr(pw, T), devptr(loss, T), devptr(ds_dpoints, T), devptr(ds_drotation, T),
            devptr(ds_dtranslation, T), devptr(ds_dbackground, T), devptr(ds_dout_weight, T),
            devptr(ds_dpoint_weight, T), Ptr{Cvoid}(UInt(pointer(ws))), length(ws))
        st = T === Float32 ?
            ccall((:dpr_raster_residual_pullback_f32, libdpr), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Int64}, Int64, Int64, Ptr{T}, Ptr{T}, Cdouble, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{Cvoid}, Csize_t), args...) :
            ccall((:dpr_raster_residual_pullback_f64, libdpr), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Int64}, Int64, Int64, Ptr{T}, Ptr{T}, Cdouble, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{Cvoid}, Csize_t), args...)
        check(st)
    end
    return (;
        points=ds_dpoints,
        rotation=ds_drotation,
        translation=ds_dtranslation,
        background=ds_dbackground,
        out_weight=ds_dout_weight,
        point_weight=ds_dpoint_weight,
        loss=loss,
    )
end

end  # module
