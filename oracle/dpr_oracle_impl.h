/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Type-generic body of the CPU restatement; included twice by dpr_oracle.c
 * with REAL = float / double and SUF = f32 / f64.  Every function cites the
 * reference (DiffPointRasterisation.jl @ 2024_10_08) lines it restates.
 *
 * Conventions shared with include/dpr.h:
 *   points      P x n_in   AoS (Vector{SVector{N_in,T}})
 *   rotation    B x (n_out x n_in column-major)  (Vector{SMatrix{N_out,N_in,T}})
 *   translation B x n_out
 *   out/ds_dout column-major (n_1..n_N, B): axis 1 fastest
 *   NULL background => 0, NULL out_weight => 1, NULL point_weight => 1
 *   (FillArrays Zeros/Ones defaults, src/interface.jl:368-394)
 *
 * Compiled with -ffp-contract=off: Julia/StaticArrays do not fuse mul+add.
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

#define DPR_MAX_DIM 4

/* src/raster.jl:85-101 reference_coordinate_and_deltas.
 * Returns 0 if the point has no in-range neighbour on some axis (then no
 * voxel_idx passes `in CartesianIndices(out)`, src/raster.jl:62); this test is
 * done in floating point BEFORE float->int conversion (Julia would throw
 * InexactError on NaN/huge; the build defines such points as skipped).
 * ref0 is the 0-based index of the LOWER neighbour (may be -1);
 * dlo = deltas[:,1] (distance to lower voxel centre, in (0,1]). */
static inline int FN(ref_and_deltas)(const REAL *p, const REAL *R, const REAL *t,
                                     const REAL *scale, int n_in, int n_out,
                                     const int64_t *grid, int64_t *ref0, REAL *dlo)
{
    for (int d = 0; d < n_out; ++d) {
        /* projected_point = rotation * point  (src/raster.jl:88); StaticArrays
         * emits a left-to-right sum of products per row. */
        REAL proj = R[d] * p[0];
        for (int j = 1; j < n_in; ++j) proj = proj + R[d + j * n_out] * p[j];
        /* origin = -ones - translation (src/raster.jl:53) */
        REAL origin = (REAL)(-1) - t[d];
        /* coord = (projected_point - origin) .* scale (src/raster.jl:92) */
        REAL coord = (proj - origin) * scale[d];
        /* round.(Int, coord .- T(0.5), RoundUp) (src/raster.jl:94) */
        REAL c = coord - (REAL)0.5;
        if (!(c > (REAL)(-1) && c <= (REAL)grid[d])) return 0;
        REAL r = CEIL(c); /* 1-based index of lower neighbour, in [0, n] */
        ref0[d] = (int64_t)r - 1;
        /* deltas_lower = coord - (ref - T(0.5)) (src/raster.jl:97) */
        dlo[d] = coord - (r - (REAL)0.5);
    }
    return 1;
}

/* src/raster.jl:103-108 voxel_weight: prod_d deltas[d, mod1(shift_d,2)] * w.
 * shift 0 -> column 2 = 1 - dlo ; shift 1 -> column 1 = dlo. */
static inline REAL FN(voxel_weight)(const REAL *dlo, int s, int n_out, REAL w)
{
    REAL v = ((s >> 0) & 1) ? dlo[0] : ((REAL)1 - dlo[0]);
    for (int d = 1; d < n_out; ++d) v = v * (((s >> d) & 1) ? dlo[d] : ((REAL)1 - dlo[d]));
    return v * w;
}

/* src/raster_pullback.jl:150-160 interpolation_weight */
static inline REAL FN(interp_weight)(int n, const REAL *dlo, int s, int n_out)
{
    REAL v = ((s >> n) & 1) ? (REAL)1 : (REAL)(-1);
    for (int m = 0; m < n_out; ++m) {
        if (m == n) continue;
        v *= ((s >> m) & 1) ? dlo[m] : ((REAL)1 - dlo[m]);
    }
    return v;
}

/* linear (column-major) offset of neighbour s of ref0, or -1 if out of range
 * (src/raster.jl:58-62; neighbour order src/util.jl:7-8,26-27: dim 1 = LSB) */
static inline int64_t FN(nbr_offset)(const int64_t *ref0, int s, int n_out, const int64_t *grid)
{
    int64_t off = 0, stride = 1;
    for (int d = 0; d < n_out; ++d) {
        int64_t i = ref0[d] + ((s >> d) & 1);
        if (i < 0 || i >= grid[d]) return -1;
        off += i * stride;
        stride *= grid[d];
    }
    return off;
}

/* ------------------------------------------------------------------------
 * Forward.  src/raster.jl:5-34 (driver: scale, background fill) + :36-66
 * (kernel body).  Serial, deterministic order (pose, point, neighbour).
 * ------------------------------------------------------------------------ */
int FN(oracle_raster)(int n_in, int n_out, const int64_t *grid, int64_t P, int64_t B,
                      REAL *out, const REAL *points, const REAL *rotation,
                      const REAL *translation, const REAL *background,
                      const REAL *out_weight, const REAL *point_weight)
{
    if (n_in < 1 || n_in > DPR_MAX_DIM || n_out < 1 || n_out > DPR_MAX_DIM) return -1;
    int64_t G = 1;
    REAL scale[DPR_MAX_DIM];
    for (int d = 0; d < n_out; ++d) {
        G *= grid[d];
        scale[d] = (REAL)grid[d] / (REAL)2; /* src/raster.jl:25 */
    }
    const int ns = 1 << n_out;
    for (int64_t b = 0; b < B; ++b) {
        REAL *o = out + b * G;
        const REAL bg = background ? background[b] : (REAL)0;
        for (int64_t i = 0; i < G; ++i) o[i] = bg; /* src/raster.jl:27 */
        const REAL *R = rotation + b * n_out * n_in;
        const REAL *t = translation + b * n_out;
        const REAL ow = out_weight ? out_weight[b] : (REAL)1;
        for (int64_t p = 0; p < P; ++p) {
            int64_t ref0[DPR_MAX_DIM];
            REAL dlo[DPR_MAX_DIM];
            const REAL w = ow * (point_weight ? point_weight[p] : (REAL)1); /* :52 */
            if (!FN(ref_and_deltas)(points + p * n_in, R, t, scale, n_in, n_out, grid, ref0, dlo))
                continue;
            for (int s = 0; s < ns; ++s) {
                int64_t off = FN(nbr_offset)(ref0, s, n_out, grid);
                if (off < 0) continue;
                o[off] += FN(voxel_weight)(dlo, s, n_out, w); /* :63-64 */
            }
        }
    }
    return 0;
}

/* Julia's sum(::Array) is pairwise with 1024-element leaves
 * (Base.mapreduce_impl); restated for ds_dbackground = sum(ds_dout)
 * (src/raster_pullback.jl:78). */
static REAL FN(pairwise_sum)(const REAL *a, int64_t lo, int64_t hi)
{
    if (hi - lo <= 1024) {
        REAL s = 0;
        for (int64_t i = lo; i < hi; ++i) s += a[i];
        return s;
    }
    int64_t mid = lo + ((hi - lo) >> 1);
    return FN(pairwise_sum)(a, lo, mid) + FN(pairwise_sum)(a, mid, hi);
}

/* One pose of the pullback, accumulating into ds_dpoints / ds_dpoint_weight
 * (src/raster_pullback.jl:2-82 with accumulate_ds_dpoints=true). */
static void FN(pullback_one_pose)(int n_in, int n_out, const int64_t *grid, int64_t P,
                                  const REAL *g, const REAL *points, const REAL *R,
                                  const REAL *t, REAL ow, const REAL *point_weight,
                                  REAL *ds_dpoints, REAL *ds_dR, REAL *ds_dt,
                                  REAL *ds_dbg, REAL *ds_dow, REAL *ds_dpw)
{
    int64_t G = 1;
    REAL scale[DPR_MAX_DIM];
    for (int d = 0; d < n_out; ++d) {
        G *= grid[d];
        scale[d] = (REAL)grid[d] / (REAL)2; /* :29 */
    }
    const int ns = 1 << n_out;
    REAL acc_t[DPR_MAX_DIM] = {0};                  /* :34 */
    REAL acc_R[DPR_MAX_DIM * DPR_MAX_DIM] = {0};    /* :35 */
    REAL acc_ow = 0;                                 /* :36 */
    for (int64_t p = 0; p < P; ++p) {               /* :39 */
        const REAL *pt = points + p * n_in;
        const REAL pw = point_weight ? point_weight[p] : (REAL)1;
        int64_t ref0[DPR_MAX_DIM];
        REAL dlo[DPR_MAX_DIM];
        if (!FN(ref_and_deltas)(pt, R, t, scale, n_in, n_out, grid, ref0, dlo)) continue;
        REAL dcoord[DPR_MAX_DIM] = {0};             /* :46 */
        REAL dpw_i = 0;                              /* :47 */
        for (int s = 0; s < ns; ++s) {              /* :49 */
            int64_t off = FN(nbr_offset)(ref0, s, n_out, grid);
            if (off < 0) continue;                   /* :51 */
            const REAL gi = g[off];                  /* :53 */
            const REAL dweight = FN(voxel_weight)(dlo, s, n_out, gi); /* :55 */
            acc_ow += dweight * pw;                  /* :57 */
            dpw_i += dweight * ow;                   /* :58 */
            const REAL factor = gi * ow * pw;        /* :60 */
            for (int n = 0; n < n_out; ++n)          /* :62-65 */
                dcoord[n] += factor * FN(interp_weight)(n, dlo, s, n_out);
        }
        REAL scaled[DPR_MAX_DIM];
        for (int n = 0; n < n_out; ++n) {
            scaled[n] = dcoord[n] * scale[n];        /* :67 */
            acc_t[n] += scaled[n];                   /* :68 */
            for (int j = 0; j < n_in; ++j)           /* :69 scaled * point' */
                acc_R[n + j * n_out] += scaled[n] * pt[j];
        }
        for (int j = 0; j < n_in; ++j) {             /* :70-71 rotation' * scaled */
            REAL v = R[0 + j * n_out] * scaled[0];
            for (int n = 1; n < n_out; ++n) v = v + R[n + j * n_out] * scaled[n];
            ds_dpoints[p * n_in + j] += v;
        }
        ds_dpw[p] += dpw_i;                          /* :72 */
    }
    for (int n = 0; n < n_out; ++n) ds_dt[n] = acc_t[n];
    for (int k = 0; k < n_out * n_in; ++k) ds_dR[k] = acc_R[k];
    *ds_dbg = FN(pairwise_sum)(g, 0, G);             /* :78 */
    *ds_dow = acc_ow;
}

/* ------------------------------------------------------------------------
 * Batched pullback with the GPU-style flat outputs
 * (src/raster_pullback.jl:85-148 with a single pose chunk;
 *  output shapes ext/DiffPointRasterisationCUDAExt.jl:313-333).
 * All six outputs are OVERWRITTEN (src/raster_pullback.jl:112-113,134-137).
 * ------------------------------------------------------------------------ */
int FN(oracle_raster_pullback)(int n_in, int n_out, const int64_t *grid, int64_t P, int64_t B,
                               const REAL *ds_dout, const REAL *points, const REAL *rotation,
                               const REAL *translation, const REAL *out_weight,
                               const REAL *point_weight, REAL *ds_dpoints, REAL *ds_drotation,
                               REAL *ds_dtranslation, REAL *ds_dbackground,
                               REAL *ds_dout_weight, REAL *ds_dpoint_weight)
{
    if (n_in < 1 || n_in > DPR_MAX_DIM || n_out < 1 || n_out > DPR_MAX_DIM) return -1;
    int64_t G = 1;
    for (int d = 0; d < n_out; ++d) G *= grid[d];
    memset(ds_dpoints, 0, sizeof(REAL) * (size_t)(P * n_in));      /* :112 */
    memset(ds_dpoint_weight, 0, sizeof(REAL) * (size_t)P);         /* :113 */
    for (int64_t b = 0; b < B; ++b) {
        FN(pullback_one_pose)(n_in, n_out, grid, P, ds_dout + b * G, points,
                              rotation + b * n_out * n_in, translation + b * n_out,
                              out_weight ? out_weight[b] : (REAL)1, point_weight, ds_dpoints,
                              ds_drotation + b * n_out * n_in, ds_dtranslation + b * n_out,
                              ds_dbackground + b, ds_dout_weight + b, ds_dpoint_weight);
    }
    return 0;
}

/* ------------------------------------------------------------------------
 * Threaded variants: the reference's CPU parallel decomposition, used ONLY as
 * the timed cpu_baseline in bench.py (kind "port").
 *   forward : every (point,pose) scatters with atomic adds
 *             (KernelAbstractions CPU backend + Atomix, src/raster.jl:29-32,64);
 *             the transform is computed once per point rather than once per
 *             neighbour work-item (a favour to the baseline).
 *   backward: poses split into min(B, nthreads) chunks with private
 *             ds_dpoints / ds_dpoint_weight slabs, then summed
 *             (src/raster_pullback.jl:112-147; src/interface.jl:402-412).
 *             With B = 1 this is serial, as in the reference.
 * ------------------------------------------------------------------------ */
int FN(oracle_raster_threaded)(int n_in, int n_out, const int64_t *grid, int64_t P, int64_t B,
                               REAL *out, const REAL *points, const REAL *rotation,
                               const REAL *translation, const REAL *background,
                               const REAL *out_weight, const REAL *point_weight)
{
    if (n_in < 1 || n_in > DPR_MAX_DIM || n_out < 1 || n_out > DPR_MAX_DIM) return -1;
    int64_t G = 1;
    REAL scale[DPR_MAX_DIM];
    for (int d = 0; d < n_out; ++d) {
        G *= grid[d];
        scale[d] = (REAL)grid[d] / (REAL)2;
    }
    const int ns = 1 << n_out;
    for (int64_t b = 0; b < B; ++b) {
        REAL *o = out + b * G;
        const REAL bg = background ? background[b] : (REAL)0;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < G; ++i) o[i] = bg;
    }
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        for (int64_t p = 0; p < P; ++p) {
            REAL *o = out + b * G;
            const REAL *R = rotation + b * n_out * n_in;
            const REAL *t = translation + b * n_out;
            const REAL ow = out_weight ? out_weight[b] : (REAL)1;
            int64_t ref0[DPR_MAX_DIM];
            REAL dlo[DPR_MAX_DIM];
            const REAL w = ow * (point_weight ? point_weight[p] : (REAL)1);
            if (!FN(ref_and_deltas)(points + p * n_in, R, t, scale, n_in, n_out, grid, ref0, dlo))
                continue;
            for (int s = 0; s < ns; ++s) {
                int64_t off = FN(nbr_offset)(ref0, s, n_out, grid);
                if (off < 0) continue;
                const REAL v = FN(voxel_weight)(dlo, s, n_out, w);
#pragma omp atomic
                o[off] += v;
            }
        }
    }
    return 0;
}

int FN(oracle_raster_pullback_threaded)(int n_in, int n_out, const int64_t *grid, int64_t P,
                                        int64_t B, const REAL *ds_dout, const REAL *points,
                                        const REAL *rotation, const REAL *translation,
                                        const REAL *out_weight, const REAL *point_weight,
                                        REAL *ds_dpoints, REAL *ds_drotation,
                                        REAL *ds_dtranslation, REAL *ds_dbackground,
                                        REAL *ds_dout_weight, REAL *ds_dpoint_weight,
                                        int n_threads)
{
    if (n_in < 1 || n_in > DPR_MAX_DIM || n_out < 1 || n_out > DPR_MAX_DIM) return -1;
    int64_t G = 1;
    for (int d = 0; d < n_out; ++d) G *= grid[d];
    if (n_threads < 1) n_threads = 1;
    const int n_chunks = (int)(B < n_threads ? B : n_threads); /* src/interface.jl:405 */
    if (n_chunks < 1) {
        memset(ds_dpoints, 0, sizeof(REAL) * (size_t)(P * n_in));
        memset(ds_dpoint_weight, 0, sizeof(REAL) * (size_t)P);
        return 0;
    }
    REAL *slab_pts = (REAL *)calloc((size_t)n_chunks * (size_t)(P * n_in), sizeof(REAL));
    REAL *slab_pw = (REAL *)calloc((size_t)n_chunks * (size_t)P, sizeof(REAL));
    if (!slab_pts || !slab_pw) {
        free(slab_pts);
        free(slab_pw);
        return -2;
    }
#pragma omp parallel for schedule(static) num_threads(n_chunks)
    for (int c = 0; c < n_chunks; ++c) {
        /* ChunkSplitters.chunks(batch_axis, n): contiguous, sizes differ by <= 1 */
        const int64_t base = B / n_chunks, rem = B % n_chunks;
        const int64_t lo = c * base + (c < rem ? c : rem);
        const int64_t hi = lo + base + (c < rem ? 1 : 0);
        for (int64_t b = lo; b < hi; ++b) {
            FN(pullback_one_pose)(n_in, n_out, grid, P, ds_dout + b * G, points,
                                  rotation + b * n_out * n_in, translation + b * n_out,
                                  out_weight ? out_weight[b] : (REAL)1, point_weight,
                                  slab_pts + (size_t)c * (size_t)(P * n_in),
                                  ds_drotation + b * n_out * n_in, ds_dtranslation + b * n_out,
                                  ds_dbackground + b, ds_dout_weight + b,
                                  slab_pw + (size_t)c * (size_t)P);
        }
    }
    /* sum(ds_dpoints; dims=3), sum(ds_dpoint_weight; dims=2)  (:141,:146) */
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < P * n_in; ++i) {
        REAL s = 0;
        for (int c = 0; c < n_chunks; ++c) s += slab_pts[(size_t)c * (size_t)(P * n_in) + i];
        ds_dpoints[i] = s;
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < P; ++i) {
        REAL s = 0;
        for (int c = 0; c < n_chunks; ++c) s += slab_pw[(size_t)c * (size_t)P + i];
        ds_dpoint_weight[i] = s;
    }
    free(slab_pts);
    free(slab_pw);
    return 0;
}

#undef FN
#undef CAT
#undef CAT_
