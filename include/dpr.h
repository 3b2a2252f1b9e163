/*
 * libdpr -- MI355X (gfx950) implementation of DiffPointRasterisation.jl's
 * `raster!` / `raster_pullback!` hot path behind a C ABI.
 *
 * This header is the drop-in boundary: the entry points are what a Julia
 * `DiffPointRasterisationAMDGPUExt` would `ccall` from array-type-specialised
 * methods of the reference's two canonical signatures (INTEGRATION.md shows the
 * binding):
 *
 *   dpr_raster_<T>           replaces  raster!  7-arg canonical method
 *                            /root/reference/src/raster.jl:5-34  (+ kernel :36-66)
 *   dpr_raster_pullback_<T>  replaces  raster_pullback!  13-arg batched method
 *                            /root/reference/src/raster_pullback.jl:85-148 and its
 *                            CUDA specialisation
 *                            /root/reference/ext/DiffPointRasterisationCUDAExt.jl:231-321
 *   (flat, un-slabbed output buffers as chosen by the allocator hooks at
 *    ext/DiffPointRasterisationCUDAExt.jl:323-333)
 *
 * Memory layouts are the reference's own (Julia column-major / AoS), so device
 * buffers can be passed without copies:
 *
 *   points        P x n_in AoS                      Vector{SVector{N_in,T}}
 *   rotation      B x (n_out x n_in, column-major)  Vector{SMatrix{N_out,N_in,T}}
 *   translation   B x n_out                         Vector{SVector{N_out,T}}
 *   background    B   or NULL => 0   (FillArrays.Zeros default, src/interface.jl:368-380)
 *   out_weight    B   or NULL => 1   (FillArrays.Ones  default, src/interface.jl:382-390)
 *   point_weight  P   or NULL => 1   (src/interface.jl:392-394)
 *   out, ds_dout  (n_1, .., n_N, B) column-major: axis 1 fastest, pose slowest
 *   ds_dpoints    n_in x P column-major (= AoS like points)
 *   ds_drotation  n_out x n_in x B column-major
 *   ds_dtranslation n_out x B ; ds_dbackground, ds_dout_weight: B ; ds_dpoint_weight: P
 *
 * All data pointers are DEVICE pointers owned by the caller (including the
 * workspace).  Calls only enqueue work on `stream` (a hipStream_t; NULL = the
 * default stream); they never synchronise the device and keep no device memory
 * or global state between calls.  Supported (n_in, n_out): 1 <= n_in, n_out <= 4 in any
 * combination -- the reference is generic in both (src/raster.jl:5-13).
 * (2,2), (3,3), (3,2) -- the shapes the reference tests (src/raster.jl:112,
 * test/data.jl:13-19) -- have every algorithm; all other pairs (incl. n_out > n_in and 4-D) run on
 * DPR_ALGO_ATOMIC (what DPR_ALGO_AUTO resolves to for them; the other algorithms
 * and the KEEP / REUSE flags return DPR_ERR_UNSUPPORTED_ALGO).
 *
 * Error behaviour: the reference throws DimensionMismatch / ArgumentError before
 * any launch (src/raster.jl:14-23, ext/...CUDAExt.jl:246-262).  Here every entry
 * point returns a status (0 = ok) and records a message retrievable with
 * dpr_last_error() (thread-local); nothing is launched on error.
 */
#ifndef DPR_H
#define DPR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPR_VERSION 105 /* 0.1.5: + DPR_OP_RESIDUAL_PULLBACK; fp32 fixed-point sums fall back to f64 on wide weight ranges */

/* status codes */
#define DPR_OK 0
#define DPR_ERR_UNSUPPORTED_DIMS (-1) /* not 1 <= n_in, n_out <= 4 */
#define DPR_ERR_INVALID_ARG (-2)      /* NULL required pointer, negative size, bad grid */
#define DPR_ERR_WORKSPACE (-3)        /* workspace NULL/too small for the chosen algorithm */
#define DPR_ERR_HIP (-4)              /* a HIP runtime call failed */
#define DPR_ERR_UNSUPPORTED_ALGO (-5) /* algorithm not available for this shape */

/* operations, for dpr_workspace_bytes_* and dpr_resolve_* */
#define DPR_OP_RASTER 0
#define DPR_OP_PULLBACK 1
/* the pullback entered through dpr_raster_residual_pullback_*: with DPR_ALGO_AUTO it never resolves to the
 * 3-D DPR_ALGO_CHUNKED pullback (which has no residual variant), so its workspace can differ from
 * DPR_OP_PULLBACK's for the same shape -- query with this op before a residual call */
#define DPR_OP_RESIDUAL_PULLBACK 2

/* algorithms (the *_ex entry points; the plain ones use DPR_ALGO_AUTO) */
#define DPR_ALGO_AUTO 0
#define DPR_ALGO_ATOMIC 1 /* thread per point, direct global float atomics / gathers */
#define DPR_ALGO_TILED 2  /* per-pose binning of points into voxel tiles, LDS-resident
                             tile accumulation (fp32 data: exact 64-bit fixed-point sums, scale
                             from |out_weight| * max|point_weight|; NaN / Inf weights and fp64
                             data: f64 sums), plain-store flush (no global atomics).  Grids of
                             more than 32768 tiles (e.g. 1024^3) are walked in slabs of tile
                             layers along the last axis -- no KEEP / REUSE there; a single tile
                             layer of more than 16384 tiles is DPR_ERR_UNSUPPORTED_ALGO.
                             AUTO picks it from ~2.5e5 points on where the cloud is dense enough
                             on the grid (the path costs ~25 ns per tile, with or without points
                             in it): >= 60 points per tile for the forward (48 on 2-D grids),
                             >= 320 for the pullback (150-430 on 2-D grids). */
#define DPR_ALGO_CHUNKED 3 /* chunks of consecutive points of a spatially coherent cloud.
                              2-D grids (projections; what AUTO picks for several poses of a
                              cloud, by a cost model -- dpr_resolve_algo_ex): a
                              block owns a chunk of 4096 points and an LDS tile under its
                              small projected footprint, and loops over the poses -- forward:
                              LDS accumulation + row-shaped global atomic flush, pullback:
                              LDS-staged ds_dout, gradients in registers across poses, no
                              atomics.  Unless DPR_FLAG_COHERENT_POINTS says the cloud is coherent
                              it is sorted into the workspace first (a counting sort into 4096
                              Hilbert-numbered cells for up to 96 poses, a radix sort on 15-bit
                              Hilbert keys beyond: compact 4096-point chunks are what the kernels
                              need, not sorted neighbours).
                              3-D grids: three kernel families behind one name.  FORWARD:
                              voxel tiles of 32 x 32 x 14 cells that own their cells outright
                              (no halo exchange, no global atomics, exact 64-bit fixed-point sums
                              for fp32 data) and find their points through a hierarchy of
                              grid-frame boxes over groups of 16 / 1024 / 65536 consecutive
                              points, built per call by one pass over the cloud -- no per-point
                              record is written; a cloud that is SPARSE on the grid (P * 10 <= G)
                              over >= 4 poses runs small 32 x 16 x 8 tiles fed by per-tile lists
                              of 64-point chunks instead.  PULLBACK: a thread per point in cloud
                              order gathers its eight ds_dout cells straight from memory (no
                              workspace beyond the partial sums: 0.9 MB per pose, 64 poses at
                              most); fp32 batches run the pose loop inside the kernel, fp64
                              batches one launch per pose.
                              What AUTO picks with DPR_FLAG_COHERENT_POINTS: the owner-tile
                              forward for >= 2 poses of a DENSE cloud (0.4 <= P / G <= 2) on a
                              grid of >= 1024 such tiles (256^3: 1216); the chunk-list forward
                              for >= 4 poses of a sparse cloud (P >= 30000, P * 10 <= G from 16
                              poses on, P * 25 <= G for 4..15 poses); the pullback for one pose
                              from 1e4 points on, for 2..31 poses from P >= G / 28, for any
                              number of poses from 3e6 points (1e6 on grids of <= 1024 tiles of
                              DPR_ALGO_TILED).  KEEP / REUSE flags are dropped where the pullback
                              is this one (it reads nothing a forward could leave).
                              Without the flag, from 8 poses and 2e5 points on, the 3-D calls
                              Hilbert-sort the cloud into the workspace themselves (the workspace
                              query says how much more that takes) and AUTO picks them for large
                              batches: the pullback from 16 poses of 3e6 points on (8 of 1e7, 64 of
                              1e6), the forward from 32 poses of a dense cloud of 3e6 points and
                              from 16 poses of a sparse one (P * 10 <= G) of 2e5 points.
                              Correct for any point order. */

/* SUMMATION ORDER.  The reference promises none for its float atomics (src/raster.jl:64) and sums
 * serially per pose on the CPU (src/raster_pullback.jl:39-72).  Here, per algorithm and output:
 *
 *   algorithm            forward `out`                           ds_dpoints / ds_dpoint_weight         per-pose sums (ds_drotation, ds_dtranslation,
 *                                                                                                       ds_dout_weight, ds_dbackground)
 *   DPR_ALGO_ATOMIC      global float atomics: depends on the    one thread per point, poses in        wave -> block -> one float atomic per block:
 *                        execution order (varies run to run,     index order: bit-reproducible         varies run to run at rounding level
 *                        rounding level)
 *   DPR_ALGO_TILED       fp32: 64-bit fixed-point sums per       one gradient record per (point,       per-thread sums in T over the tile's records
 *                        tile, EXACT -- independent of the       pose), poses added in index order:    (their order in the tile's list varies), then
 *                        order of the points and of execution;   bit-reproducible                      f64 across threads and tiles in a fixed order:
 *                        tiles split into parts (> max(4096,                                           rounding level of T within a tile
 *                        P/256) records) add their parts in
 *                        fp32 in a fixed order, the records
 *                        of a part vary: rounding level.
 *                        fp64: f64 LDS atomics, order-dependent
 *                        at 1e-16 relative
 *   DPR_ALGO_CHUNKED     2-D: exact fixed-point sums per chunk   registers across the poses, fixed      2-D: the 4096 terms of a (chunk, pose) as a
 *                        (fp32), then float atomics into the     order: bit-reproducible for a given    fixed tree in T (fp32 data; fp64: f64 across
 *                        image across chunks: run to run,        point order (2-D); 3-D: one thread     threads), the partials per (chunk, pose) in f64
 *                        rounding level.  3-D owner tiles:       per point, poses added in index        in a fixed order.  3-D: per-thread sums in T over
 *                        fp32 EXACT (64-bit fixed point, split   order through memory: bit-             a slice of the cloud (its size follows the CU
 *                        tiles summed as integers) -- the same   reproducible                           count of the device), then f64: one pose per launch
 *                        bits for any point order; fp64: f64                                            in a fixed order; the fp32 batch kernel (pose loop
 *                        LDS atomics, rounding level.  3-D                                              inside) adds its waves' sums with f64 LDS atomics in
 *                        chunk lists (sparse clouds): f64 LDS                                           arrival order: rounding level of f64
 *                        atomics + diverted global atomics
 *
 * "Rounding level" = the differences any two summation orders of the same terms show in the
 * accumulation type; no output depends on the order beyond that.  The contributions themselves
 * (cell choice, weights) are computed with the reference's operation order in T and do not depend
 * on any order.
 *
 * PRECISION OF THE FIXED-POINT SUMS (fp32 data; the reference's float atomics, src/raster.jl:62-64, keep
 * the relative precision of every cell whatever its magnitude).  A contribution is rounded to a multiple of
 * 2^-sexp, with sexp chosen per work item so that n * maxw * 2^sexp <= 2^62 (n = records of the item,
 * maxw = |out_weight| * max |point_weight| of the SCOPE: the call on DPR_ALGO_TILED, the candidate chunks of
 * a tile on the 3-D DPR_ALGO_CHUNKED forward, the 4096-point chunk on the 2-D one): a contribution keeps at
 * least 38 bits below maxw (items hold < 2^24 records; typically 49).  Where the NON-ZERO |point_weight| of
 * a scope span more than 2^10 -- or a weight is NaN / Inf -- the scope accumulates with IEEE f64 atomics
 * instead (since 0.1.5; before, cells reached only by points lighter than ~2^-38 of the call's heaviest came
 * out as 0).  So every contribution keeps >= 28 bits (typically 39) below the SMALLEST weight next to it,
 * and a cell's absolute error is <= k * 2^-39 * (smallest weight in scope) for its k contributions: below
 * fp32 rounding (2^-24 relative) for every cell except far corners of their only contributors (value below
 * ~2^-12 of the local weight), which keep at least 16 good bits.  Default weights (NULL) never trip the guard.
 * tests/test_parity_gpu.py::test_fp32_cells_reached_only_by_small_weights_keep_their_relative_precision.
 *
 * ENVIRONMENT.  The library reads one variable, DPR_MAX_TILES (16..32768, default 32768): the number of tiles
 * DPR_ALGO_TILED handles per launch sequence before it cuts the grid into slabs along the last axis -- a test
 * hook that lets a small grid walk the slab code; it moves slab boundaries, never results.  The A/B switches
 * named in profiles/r0N_experiments.md (DPR_FIXED_POINT, DPR_SORT_BITS, DPR_CO_SORT, DPR_OWN_*, ...) exist
 * only in builds made with `make EXPERIMENTS=1` (-DDPR_EXPERIMENTS); the shipped library ignores them. */

/* flags (the *_ex entry points).  DPR_ALGO_TILED: any B -- with B > 1 every pose keeps its own
 * binning (the per-pose part of the workspace is laid out B times; pose groups are off);
 * DPR_ALGO_CHUNKED on 2-D grids: any B (what is kept there is the sorted copy of the cloud and its
 * permutation); 3-D DPR_ALGO_CHUNKED: accepted and ignored (its pullback reads nothing a forward
 * could leave); DPR_ALGO_ATOMIC has nothing to keep (error):
 * KEEP_BINNING  (raster)   leave the per-tile binning of the points (incl. original
 *                          indices) in the workspace for the pullback of the same call pair
 * REUSE_BINNING (pullback) the workspace still holds the binning written by the preceding
 *                          raster call with the SAME points, pose, grid and point_weight
 *                          pointer-ness; skips the count/scan/scatter stages.
 * This is the cache an rrule keeps between `raster` and its pullback closure
 * (ext/DiffPointRasterisationChainRulesCoreExt.jl:6-27); the reference itself recomputes
 * (src/raster_pullback.jl:20-22). */
#define DPR_FLAG_KEEP_BINNING 1u
#define DPR_FLAG_REUSE_BINNING 2u
/* A REUSE_BINNING pullback validates ON THE DEVICE that the workspace holds the binning of a
 * KEEP_BINNING forward with the same P, grid, element type, points / point_weight buffers and
 * pose values, and that no pullback has consumed it yet (the gradient records overwrite the
 * point records), and that it was written in the workspace LAYOUT of this call (same
 * DPR_FLAG_COHERENT_POINTS / DPR_FLAG_MAX_POSE_GROUP on both calls of the pair).  If not, nothing
 * is read through the stale lists and the outputs come back as NaN; the status is still 0
 * because the host cannot see the mismatch without synchronising.  Which outputs: DPR_ALGO_TILED
 * -- all six and `loss`; DPR_ALGO_CHUNKED on 2-D grids (what is reused there is the sorted copy of
 * the cloud) -- ds_dpoints, ds_dpoint_weight, ds_drotation, ds_dtranslation and ds_dout_weight,
 * while ds_dbackground and `loss` do not depend on the points and stay valid.
 *
 * DPR_FLAG_MAX_POSE_GROUP(n), n in 1..16 (0 = library default 16): DPR_ALGO_TILED bins up to n
 * poses of a batch together when the grid has few tiles (n * tiles <= 4096).  Larger groups are
 * faster (the points are read once per group) but the record / slot-map part of the workspace
 * grows n-fold: (20 | 36) bytes * P * n for fp32 | fp64 -- e.g. 10 M points, 512^2 grid:
 * 0.28 GB at n = 1, 1.66 GB at n = 8.  Pass the same flags to dpr_workspace_bytes_ex_*. */
#define DPR_FLAG_MAX_POSE_GROUP(n) (((unsigned)(n) & 0xffu) << 8)
/* DPR_FLAG_COHERENT_POINTS: the caller states that neighbouring points in memory are
 * neighbours in space (e.g. the output of dpr_sort_points_*).  DPR_ALGO_CHUNKED on 2-D grids
 * then skips its own Hilbert sort (and the workspace shrinks to the per-pose partial sums);
 * DPR_ALGO_TILED bins such a cloud locally (sub-chunks of 4096 / 2048 consecutive points ordered
 * by tile in LDS, run descriptors instead of a count pass; grids of up to 16384 tiles) and, for a
 * batch, all poses of up to 8 in one launch.  Without the flag batched calls on grids of more than
 * 4096 tiles order the cloud themselves first (a counting sort into 4096 cells of the model frame,
 * once per call).  A wrong claim costs speed, never correctness. */
#define DPR_FLAG_COHERENT_POINTS 4u
/* DPR_FLAG_NO_POINT_WEIGHT_GRAD (pullback entry points): the caller does not need
 * ds_dpoint_weight -- the reference's rrule drops that tangent whenever `point_weight` was
 * defaulted (ext/DiffPointRasterisationChainRulesCoreExt.jl:23,70).  The pointer may be NULL and
 * nothing is written through it (a P-element store per call less; every algorithm honours it).
 * The five other outputs are unchanged. */
#define DPR_FLAG_NO_POINT_WEIGHT_GRAD 8u

int dpr_version(void);

/* Thread-local message of the last failing call on this host thread ("" if none). */
const char *dpr_last_error(void);

/* Algorithm DPR_ALGO_AUTO resolves to for this problem (DPR_ALGO_ATOMIC, DPR_ALGO_TILED or, on
 * 2-D grids, DPR_ALGO_CHUNKED), or a negative status.  The choice depends on the flags:
 *  - DPR_FLAG_COHERENT_POINTS makes the paths that exploit it cheaper;
 *  - with DPR_FLAG_KEEP_BINNING or DPR_FLAG_REUSE_BINNING the choice is made for the raster +
 *    pullback PAIR (both calls must run the same algorithm), from arguments both calls share.
 *    When the pair's algorithm has nothing to share (DPR_ALGO_ATOMIC; DPR_ALGO_TILED with
 *    B > 1 on a grid small enough for pose groups, or with more than 2^29 (point, pose) pairs
 *    to keep), AUTO ignores the two flags -- each call then works on its own -- where an
 *    explicitly named algorithm returns an error (DPR_ALGO_ATOMIC) or shares at any size
 *    (DPR_ALGO_TILED: every pose of the batch keeps its own binning, B-fold workspace).  So a caller may always pass AUTO + KEEP to
 *    raster and AUTO + REUSE to the pullback of the same arguments.
 * dpr_resolve_algo is dpr_resolve_algo_ex with flags = 0. */
int dpr_resolve_algo(int op, int n_in, int n_out, const int64_t *grid, int64_t P, int64_t B);
int dpr_resolve_algo_ex(int op, unsigned flags, int n_in, int n_out, const int64_t *grid,
                        int64_t P, int64_t B);
/* The flags DPR_ALGO_AUTO will act on for this problem (>= 0), or a negative status: `flags`
 * with DPR_FLAG_KEEP_BINNING / DPR_FLAG_REUSE_BINNING cleared where the pair's algorithm has
 * nothing to share.  A host that wants to know whether the pullback will really skip its binning
 * (e.g. to report it) asks here. */
int dpr_resolve_flags_ex(int op, unsigned flags, int n_in, int n_out, const int64_t *grid,
                         int64_t P, int64_t B);

/* Optional per-stage device timing (used by bench.py for the roofline numbers): arm an
 * array of `capacity` hipEvent_t created by the caller; until dpr_stage_timing_end() every
 * raster / pullback call on THIS host thread records events[0] when it starts enqueuing
 * and the next event after each stage of its pipeline, on the call's stream.
 * dpr_stage_timing_end() disarms and returns the number of events recorded.
 * Stage order -- DPR_ALGO_ATOMIC raster: fill, splat; pullback: zero+grid_sum, gather.
 * DPR_ALGO_TILED, per pose -- raster: count, scan, scatter, tile_splat, halo;
 * pullback: count, scan, scatter, tile_gather, unpermute, pose_reduce.
 * DPR_ALGO_CHUNKED, 3-D grids -- raster: boxes, plan, own_splat, combine;
 * pullback: direct_gather, pose_reduce.
 * DPR_ALGO_CHUNKED, 2-D grids -- raster: sort, fill, chunk_splat;
 * pullback: sort, grid_sum, chunk_gather, reduce+unsort. */
int dpr_stage_timing_begin(void **events, int capacity);
int dpr_stage_timing_end(void);

/* Bytes of caller-provided device workspace needed by `op` with `algo`
 * (may be 0).  Returns (size_t)-1 on invalid arguments.  The workspace pointer must be
 * 256-byte aligned (DPR_ERR_WORKSPACE otherwise; hipMalloc / AMDGPU.jl / torch allocations
 * are); every non-NULL data pointer must be aligned to its element type
 * (DPR_ERR_INVALID_ARG otherwise, before any launch).  DPR_ALGO_TILED with a shape it would
 * refuse (a tile layer of more than 16384 tiles, P >= 2^32, KEEP / REUSE flags on a grid of more
 * than 32768 tiles) returns (size_t)-1 here too. */
size_t dpr_workspace_bytes_f32(int op, int algo, int n_in, int n_out, const int64_t *grid,
                               int64_t P, int64_t B);
size_t dpr_workspace_bytes_f64(int op, int algo, int n_in, int n_out, const int64_t *grid,
                               int64_t P, int64_t B);
/* The same for a call that will pass `flags`: query with EXACTLY the flags of the call.
 * DPR_FLAG_MAX_POSE_GROUP and DPR_FLAG_COHERENT_POINTS change the layout (and the size) of the
 * tiled and the chunk-owner workspaces, and with DPR_ALGO_AUTO the KEEP / REUSE flags take part
 * in choosing the algorithm (see dpr_resolve_algo_ex).  A workspace that serves a KEEP raster
 * and its REUSE pullback is the larger of the two queries. */
size_t dpr_workspace_bytes_ex_f32(int op, int algo, unsigned flags, int n_in, int n_out,
                                  const int64_t *grid, int64_t P, int64_t B);
size_t dpr_workspace_bytes_ex_f64(int op, int algo, unsigned flags, int n_in, int n_out,
                                  const int64_t *grid, int64_t P, int64_t B);

/* Forward: out[.., b] = background[b] + sum_p splat(R[b] p + t[b]) * out_weight[b] * point_weight[p]
 * `out` is fully overwritten (src/raster.jl:27). */
int dpr_raster_f32(void *stream, int n_in, int n_out, const int64_t *grid, int64_t P, int64_t B,
                   float *out, const float *points, const float *rotation,
                   const float *translation, const float *background, const float *out_weight,
                   const float *point_weight, void *workspace, size_t workspace_bytes);
int dpr_raster_f64(void *stream, int n_in, int n_out, const int64_t *grid, int64_t P, int64_t B,
                   double *out, const double *points, const double *rotation,
                   const double *translation, const double *background, const double *out_weight,
                   const double *point_weight, void *workspace, size_t workspace_bytes);
int dpr_raster_ex_f32(void *stream, int algo, unsigned flags, int n_in, int n_out, const int64_t *grid, int64_t P,
                      int64_t B, float *out, const float *points, const float *rotation,
                      const float *translation, const float *background, const float *out_weight,
                      const float *point_weight, void *workspace, size_t workspace_bytes);
int dpr_raster_ex_f64(void *stream, int algo, unsigned flags, int n_in, int n_out, const int64_t *grid, int64_t P,
                      int64_t B, double *out, const double *points, const double *rotation,
                      const double *translation, const double *background,
                      const double *out_weight, const double *point_weight, void *workspace,
                      size_t workspace_bytes);

/* Pullback.  All six outputs are OVERWRITTEN (ext/...CUDAExt.jl:272-276).
 * `background` is not an input of the arithmetic (src/raster_pullback.jl:7,78). */
int dpr_raster_pullback_f32(void *stream, int n_in, int n_out, const int64_t *grid, int64_t P,
                            int64_t B, const float *ds_dout, const float *points,
                            const float *rotation, const float *translation,
                            const float *out_weight, const float *point_weight, float *ds_dpoints,
                            float *ds_drotation, float *ds_dtranslation, float *ds_dbackground,
                            float *ds_dout_weight, float *ds_dpoint_weight, void *workspace,
                            size_t workspace_bytes);
int dpr_raster_pullback_f64(void *stream, int n_in, int n_out, const int64_t *grid, int64_t P,
                            int64_t B, const double *ds_dout, const double *points,
                            const double *rotation, const double *translation,
                            const double *out_weight, const double *point_weight,
                            double *ds_dpoints, double *ds_drotation, double *ds_dtranslation,
                            double *ds_dbackground, double *ds_dout_weight,
                            double *ds_dpoint_weight, void *workspace, size_t workspace_bytes);
int dpr_raster_pullback_ex_f32(void *stream, int algo, unsigned flags, int n_in, int n_out, const int64_t *grid,
                               int64_t P, int64_t B, const float *ds_dout, const float *points,
                               const float *rotation, const float *translation,
                               const float *out_weight, const float *point_weight,
                               float *ds_dpoints, float *ds_drotation, float *ds_dtranslation,
                               float *ds_dbackground, float *ds_dout_weight,
                               float *ds_dpoint_weight, void *workspace, size_t workspace_bytes);
int dpr_raster_pullback_ex_f64(void *stream, int algo, unsigned flags, int n_in, int n_out, const int64_t *grid,
                               int64_t P, int64_t B, const double *ds_dout, const double *points,
                               const double *rotation, const double *translation,
                               const double *out_weight, const double *point_weight,
                               double *ds_dpoints, double *ds_drotation, double *ds_dtranslation,
                               double *ds_dbackground, double *ds_dout_weight,
                               double *ds_dpoint_weight, void *workspace, size_t workspace_bytes);

/* Residual pullback: the caller one step out of raster_pullback!.
 * `out` is the result of dpr_raster_* for the same points / poses / weights, `target` a grid
 * of the same layout.  Computes, without materialising the sensitivity,
 *     ds_dout = residual_scale * (out - target)        README.md:151 (scale -2 there; +2 is
 *                                                      the gradient of examples/logo.jl:40-44)
 *     <all six outputs of dpr_raster_pullback_*>(ds_dout, ...)
 *     loss[b] = sum over pose b's grid of (out - target)^2      (loss may be NULL)
 * The sensitivity is formed in the kernels that consume it: the grid is read twice (out,
 * target) instead of read twice, written once and read again.  Flags as for
 * dpr_raster_pullback_ex_*; workspace: dpr_workspace_bytes_*(DPR_OP_RESIDUAL_PULLBACK, ...).  An explicit
 * DPR_ALGO_CHUNKED on a 3-D grid is refused (DPR_ERR_UNSUPPORTED_ALGO: the direct gather kernels have no
 * residual variant); DPR_ALGO_AUTO picks among the algorithms that have one. */
int dpr_raster_residual_pullback_f32(void *stream, int n_in, int n_out, const int64_t *grid,
                                     int64_t P, int64_t B, const float *out, const float *target,
                                     double residual_scale, const float *points,
                                     const float *rotation, const float *translation,
                                     const float *out_weight, const float *point_weight,
                                     float *loss, float *ds_dpoints, float *ds_drotation,
                                     float *ds_dtranslation, float *ds_dbackground,
                                     float *ds_dout_weight, float *ds_dpoint_weight,
                                     void *workspace, size_t workspace_bytes);
int dpr_raster_residual_pullback_f64(void *stream, int n_in, int n_out, const int64_t *grid,
                                     int64_t P, int64_t B, const double *out,
                                     const double *target, double residual_scale,
                                     const double *points, const double *rotation,
                                     const double *translation, const double *out_weight,
                                     const double *point_weight, double *loss, double *ds_dpoints,
                                     double *ds_drotation, double *ds_dtranslation,
                                     double *ds_dbackground, double *ds_dout_weight,
                                     double *ds_dpoint_weight, void *workspace,
                                     size_t workspace_bytes);
int dpr_raster_residual_pullback_ex_f32(void *stream, int algo, unsigned flags, int n_in,
                                        int n_out, const int64_t *grid, int64_t P, int64_t B,
                                        const float *out, const float *target,
                                        double residual_scale, const float *points,
                                        const float *rotation, const float *translation,
                                        const float *out_weight, const float *point_weight,
                                        float *loss, float *ds_dpoints, float *ds_drotation,
                                        float *ds_dtranslation, float *ds_dbackground,
                                        float *ds_dout_weight, float *ds_dpoint_weight,
                                        void *workspace, size_t workspace_bytes);
int dpr_raster_residual_pullback_ex_f64(void *stream, int algo, unsigned flags, int n_in,
                                        int n_out, const int64_t *grid, int64_t P, int64_t B,
                                        const double *out, const double *target,
                                        double residual_scale, const double *points,
                                        const double *rotation, const double *translation,
                                        const double *out_weight, const double *point_weight,
                                        double *loss, double *ds_dpoints, double *ds_drotation,
                                        double *ds_dtranslation, double *ds_dbackground,
                                        double *ds_dout_weight, double *ds_dpoint_weight,
                                        void *workspace, size_t workspace_bytes);

/* Pose-independent spatial pre-sort of the model-frame points along a Hilbert curve (any run
 * of consecutive sorted points is a compact blob; 3-D: 30-bit keys, 1024^3 cells over [-1, 1)^3) --
 * not in the reference; every algorithm here is faster on coherent input and the sort only depends
 * on the points.  points_sorted[i] = points[perm[i]] (and point weights likewise; pass NULL for
 * both weight pointers when unused).  Gradients of the sorted cloud go back with
 * ds_dpoints[perm[i]] = ds_dpoints_sorted[i].  n_in = 2 or 3; P < 2^32. */
size_t dpr_sort_points_workspace_bytes(int64_t P);
int dpr_sort_points_f32(void *stream, int n_in, int64_t P, const float *points,
                        float *points_sorted, uint32_t *perm, const float *point_weight,
                        float *point_weight_sorted, void *workspace, size_t workspace_bytes);
int dpr_sort_points_f64(void *stream, int n_in, int64_t P, const double *points,
                        double *points_sorted, uint32_t *perm, const double *point_weight,
                        double *point_weight_sorted, void *workspace, size_t workspace_bytes);

/* ---- multi-GPU: pose sharding over RCCL (one process or task per GPU) ------------------------
 * The batched pullback shards over poses: rank r owns the contiguous pose block
 * dpr_shard_range(B, r, world), the points are replicated, every per-pose output is disjoint and
 * the point gradients sum over ranks with ONE all-reduce -- the multi-process form of the
 * reference's per-thread slabs + sum (src/raster_pullback.jl:112-147).  The forward needs no
 * exchange: call dpr_raster_* on the local pose block.
 *
 *   rank 0:  dpr_comm_unique_id(id, DPR_COMM_ID_BYTES); ship the 128 bytes to the other ranks by
 *            the host's own means (MPI, a file, Julia Distributed); every rank, with its GPU
 *            current: dpr_comm_init(&comm, world, rank, id); ... ; dpr_comm_destroy(comm).
 * RCCL is loaded at run time (librccl.so.1); DPR_ERR_HIP if it is not available. */
#define DPR_COMM_ID_BYTES 128
typedef struct dpr_comm dpr_comm_t;
int dpr_comm_unique_id(void *id_out, size_t id_bytes);
int dpr_comm_init(dpr_comm_t **comm_out, int world, int rank, const void *id);
int dpr_comm_destroy(dpr_comm_t *comm);
int dpr_comm_world(const dpr_comm_t *comm);
int dpr_comm_rank(const dpr_comm_t *comm);
void dpr_shard_range(int64_t batch, int rank, int world, int64_t *lo, int64_t *hi);
/* dpr_raster_pullback_<T> on this rank's B_local poses (all `_local` arguments are the rank's
 * slices), then all-reduce(sum) of ds_dpoints / ds_dpoint_weight over the communicator on
 * `stream`: after the call they hold the global sums on every rank.  One all-reduce when the two
 * buffers are adjacent (ds_dpoint_weight == ds_dpoints + n_in * P), a grouped pair otherwise.
 *
 * When the LOCAL pullback of a rank fails (non-zero status, e.g. a workspace sized for another
 * rank's B_local) the rank still JOINS the all-reduce, with its two gradient buffers filled with
 * NaN, and returns its own error afterwards: no peer is left blocked in the collective, and every
 * rank sees NaN point gradients instead of sums that silently miss one rank's poses.  Only a rank
 * whose gradient buffers are NULL cannot take part; its error text says so, and the communicator
 * must then be destroyed on every rank (the peers are blocked).  dpr_shard_range with an invalid
 * (rank, world) yields the empty range [0, 0). */
int dpr_raster_pullback_sharded_f32(dpr_comm_t *comm, void *stream, int n_in, int n_out,
                                    const int64_t *grid, int64_t P, int64_t B_local,
                                    const float *ds_dout_local, const float *points,
                                    const float *rotation_local, const float *translation_local,
                                    const float *out_weight_local, const float *point_weight,
                                    float *ds_dpoints, float *ds_drotation_local,
                                    float *ds_dtranslation_local, float *ds_dbackground_local,
                                    float *ds_dout_weight_local, float *ds_dpoint_weight,
                                    void *workspace, size_t workspace_bytes);
int dpr_raster_pullback_sharded_f64(dpr_comm_t *comm, void *stream, int n_in, int n_out,
                                    const int64_t *grid, int64_t P, int64_t B_local,
                                    const double *ds_dout_local, const double *points,
                                    const double *rotation_local, const double *translation_local,
                                    const double *out_weight_local, const double *point_weight,
                                    double *ds_dpoints, double *ds_drotation_local,
                                    double *ds_dtranslation_local, double *ds_dbackground_local,
                                    double *ds_dout_weight_local, double *ds_dpoint_weight,
                                    void *workspace, size_t workspace_bytes);

#ifdef __cplusplus
}
#endif
#endif /* DPR_H */
