# Two-trigger package extension: loaded when BOTH AMDGPU and ChainRulesCore are loaded.
# Project.toml of DiffPointRasterisation (next to the reference's entries, Project.toml:16-22):
#
#   [weakdeps]
#   AMDGPU = "21141c5a-9bdb-4563-92ae-f87d6854732e"
#   ChainRulesCore = "d360d2e6-b24c-11e9-a2a3-2a2ae2dbcce4"
#
#   [extensions]
#   DiffPointRasterisationAMDGPUExt = "AMDGPU"
#   DiffPointRasterisationAMDGPUChainRulesCoreExt = ["AMDGPU", "ChainRulesCore"]
#
# It specialises the reference's single-image rrule
# (ext/DiffPointRasterisationChainRulesCoreExt.jl:6-27) for ROCArray points: the forward call
# keeps its binning in a workspace the pullback closure owns (DPR_FLAG_KEEP_BINNING), the first
# call of the closure reuses it (DPR_FLAG_REUSE_BINNING, validated on the device), later calls
# re-bin through the generic method.  The tangent order is the reference's (:20-23).
# UNEXECUTED here: neither this image nor the GPU box has a Julia runtime (SURVEY.md section 0).
module DiffPointRasterisationAMDGPUChainRulesCoreExt

using DiffPointRasterisation, AMDGPU, ChainRulesCore, StaticArrays
using FillArrays: Zeros, Ones

# helpers of the AMDGPU extension (loaded before this one: it is one of the two triggers)
const AMDExt = Base.get_extension(DiffPointRasterisation, :DiffPointRasterisationAMDGPUExt)
const workspace_pair = AMDExt.workspace_pair
const raster_keep! = AMDExt.raster_keep!
const raster_pullback_reuse! = AMDExt.raster_pullback_reuse!
const devbuf = AMDExt.devbuf
const raster_keep_batch! = AMDExt.raster_keep_batch!
const raster_pullback_reuse_batch! = AMDExt.raster_pullback_reuse_batch!

function ChainRulesCore.rrule(
    ::typeof(DiffPointRasterisation.raster),
    grid_size,
    points::ROCVector{<:StaticVector{N_in,T}},
    rotation::AbstractMatrix{<:Number},
    translation::AbstractVector{<:Number},
    optional_args...,
) where {N_in,T<:Union{Float32,Float64}}
    N_out = length(grid_size)
    P = length(points)
    bg = length(optional_args) >= 1 ? [T(optional_args[1])] : Zeros{T}(1)
    ow = length(optional_args) >= 2 ? [T(optional_args[2])] : Ones{T}(1)
    pw = length(optional_args) >= 3 ? optional_args[3] : Ones{T}(P)
    out = similar(points, T, Tuple(grid_size))
    ws = workspace_pair(T, N_in, N_out, collect(Int64, grid_size), P, 1)
    rot_d, tr_d, ow_d, pw_d = raster_keep!(out, points, rotation, translation, bg, ow, pw, ws)
    consumed = Ref(false)
    function raster_pullback(ds_dout)
        g = devbuf(ChainRulesCore.unthunk(ds_dout), T)
        pb = if consumed[]   # a second call through the same closure re-bins (generic path)
            DiffPointRasterisation.raster_pullback!(g, points, rotation, translation, optional_args...)
        else
            consumed[] = true
            # (point_weight defaulted: its tangent is dropped below -- not computed either)
            raster_pullback_reuse!(g, points, rot_d, tr_d, ow_d, pw_d, ws; want_pw=length(optional_args) >= 3)
        end
        ds_dpoints = reinterpret(reshape, SVector{N_in,T}, pb.points)
        single = (dropdims(pb.rotation; dims=3), vec(pb.translation),
                  sum(pb.background), sum(pb.out_weight), pb.point_weight)
        return ChainRulesCore.NoTangent(), ChainRulesCore.NoTangent(), ds_dpoints,
               single[1:(2 + length(optional_args))]...
    end
    return out, raster_pullback
end

# batch of images: the reference's rrule (ext/DiffPointRasterisationChainRulesCoreExt.jl:47-74)
# specialised for ROCArray points; DPR_ALGO_AUTO decides for the raster + pullback PAIR and shares
# where that pays (every pose keeps its binning on the tiled path; the sorted copy of the cloud on
# the chunk-owner path), else ignores the two flags -- the closure does not need to know which.
function ChainRulesCore.rrule(
    ::typeof(DiffPointRasterisation.raster),
    grid_size,
    points::ROCVector{<:StaticVector{N_in,T}},
    rotation::AbstractVector{<:StaticMatrix{N_out,N_in,TR}},
    translation::AbstractVector{<:StaticVector{N_out,TT}},
    optional_args...,
) where {N_in,N_out,T<:Union{Float32,Float64},TR<:Number,TT<:Number}
    P, B = length(points), length(rotation)
    bg = length(optional_args) >= 1 ? optional_args[1] : Zeros{T}(B)
    ow = length(optional_args) >= 2 ? optional_args[2] : Ones{T}(B)
    pw = length(optional_args) >= 3 ? optional_args[3] : Ones{T}(P)
    out = similar(points, T, (grid_size..., B))
    ws = workspace_pair(T, N_in, N_out, collect(Int64, grid_size), P, B)
    rot_d, tr_d, ow_d, pw_d = raster_keep_batch!(out, points, rotation, translation, bg, ow, pw, ws)
    consumed = Ref(false)
    function raster_pullback(ds_dout)
        g = devbuf(ChainRulesCore.unthunk(ds_dout), T)
        pb = if consumed[]   # a second call through the same closure re-bins (generic path)
            DiffPointRasterisation.raster_pullback!(g, points, rotation, translation, optional_args...)
        else
            consumed[] = true
            raster_pullback_reuse_batch!(g, points, rot_d, tr_d, ow_d, pw_d, ws; want_pw=length(optional_args) >= 3)
        end
        ds_dpoints = reinterpret(reshape, SVector{N_in,T}, pb.points)
        L = N_out * N_in
        ds_drotation = reinterpret(reshape, SMatrix{N_out,N_in,T,L}, reshape(pb.rotation, L, :))
        ds_dtranslation = reinterpret(reshape, SVector{N_out,T}, pb.translation)
        return ChainRulesCore.NoTangent(), ChainRulesCore.NoTangent(), ds_dpoints, ds_drotation,
               ds_dtranslation, values(pb)[4:(3 + length(optional_args))]...
    end
    return out, raster_pullback
end

end  # module
