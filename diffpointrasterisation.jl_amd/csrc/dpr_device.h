// Device-side arithmetic shared by every kernel of libdpr (gfx950 only).
//
// The operation ORDER below is the reference's (see the cited lines): the whole
// library is compiled with -ffp-contract=off so that the fp32 cell choice
// `ceil(coord - 1/2)` and every per-neighbour weight are bit-identical to the
// fp32 CPU oracle; only the order in which contributions are summed differs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dpr {

constexpr int kWave = 64;  // CDNA wavefront

template <typename T> __device__ __forceinline__ T ceil_t(T x);
template <> __device__ __forceinline__ float ceil_t<float>(float x) { return ceilf(x); }
template <> __device__ __forceinline__ double ceil_t<double>(double x) { return ceil(x); }

// Per-pose parameters; read from global memory with wave-uniform addresses
// (the compiler turns them into scalar loads).
template <typename T, int NI, int NO> struct Pose {
    T R[NO * NI];  // column-major N_out x N_in (SMatrix memory order)
    T t[NO];
    T ow;
};

template <typename T, int NI, int NO>
__device__ __forceinline__ Pose<T, NI, NO> load_pose(const T* __restrict__ rot,
                                                    const T* __restrict__ trans,
                                                    const T* __restrict__ ow, int64_t b) {
    Pose<T, NI, NO> ps;
#pragma unroll
    for (int k = 0; k < NO * NI; ++k) ps.R[k] = rot[b * (NO * NI) + k];
#pragma unroll
    for (int d = 0; d < NO; ++d) ps.t[d] = trans[b * NO + d];
    ps.ow = ow ? ow[b] : T(1);
    return ps;
}

template <int NO> struct GridDesc {
    int n[NO];        // voxels per axis
    int64_t G;        // voxels per pose
};

// /root/reference/src/raster.jl:85-101 reference_coordinate_and_deltas, with
// origin = -1 - t (src/raster.jl:53) and scale = n/2 (src/raster.jl:25).
// ref0 = 0-based index of the LOWER neighbour (may be -1); dlo = deltas[:,1].
// Returns false when some axis has no in-range neighbour (range test in floating
// point before the float->int conversion; also rejects NaN/Inf).
template <typename T, int NI, int NO>
__device__ __forceinline__ bool ref_and_deltas(const T (&p)[NI], const Pose<T, NI, NO>& ps,
                                               const GridDesc<NO>& gd, int (&ref0)[NO],
                                               T (&dlo)[NO]) {
    bool ok = true;
#pragma unroll
    for (int d = 0; d < NO; ++d) {
        T proj = ps.R[d] * p[0];
#pragma unroll
        for (int j = 1; j < NI; ++j) proj = proj + ps.R[d + j * NO] * p[j];
        const T origin = T(-1) - ps.t[d];
        const T scale = T(gd.n[d]) / T(2);
        const T coord = (proj - origin) * scale;
        const T c = coord - T(0.5);
        ok = ok && (c > T(-1)) && (c <= T(gd.n[d]));
        const T r = ceil_t<T>(c);
        ref0[d] = ok ? (int)r - 1 : 0;
        dlo[d] = coord - (r - T(0.5));
    }
    return ok;
}

// src/raster.jl:103-108 voxel_weight: prod_d deltas[d, mod1(shift_d, 2)] * w;
// neighbour s has shift_d = bit d of s (src/util.jl:7-8,26-27).
template <typename T, int NO>
__device__ __forceinline__ T voxel_weight(const T (&dlo)[NO], int s, T w) {
    T v = (s & 1) ? dlo[0] : (T(1) - dlo[0]);
#pragma unroll
    for (int d = 1; d < NO; ++d) v = v * (((s >> d) & 1) ? dlo[d] : (T(1) - dlo[d]));
    return v * w;
}

// src/raster_pullback.jl:150-160 interpolation_weight
template <typename T, int NO>
__device__ __forceinline__ T interp_weight(int n, const T (&dlo)[NO], int s) {
    T v = ((s >> n) & 1) ? T(1) : T(-1);
#pragma unroll
    for (int m = 0; m < NO; ++m) {
        if (m == n) continue;
        v *= ((s >> m) & 1) ? dlo[m] : (T(1) - dlo[m]);
    }
    return v;
}

// column-major offset of neighbour s inside one pose, or -1 if out of range
// (individual drop, src/raster.jl:62 / src/raster_pullback.jl:51)
template <int NO>
__device__ __forceinline__ int nbr_offset(const int (&ref0)[NO], int s, const GridDesc<NO>& gd) {
    int off = 0, stride = 1;
    bool ok = true;
#pragma unroll
    for (int d = 0; d < NO; ++d) {
        const int i = ref0[d] + ((s >> d) & 1);
        ok = ok && (i >= 0) && (i < gd.n[d]);
        off += i * stride;
        stride *= gd.n[d];
    }
    return ok ? off : -1;
}

template <typename T, int NI>
__device__ __forceinline__ void load_point(const T* __restrict__ points, int64_t p, T (&v)[NI]) {
#pragma unroll
    for (int j = 0; j < NI; ++j) v[j] = points[p * NI + j];
}

// Per-point backward quantities for one pose (src/raster_pullback.jl:46-67):
// scaled = ds_dcoord .* scale (with out_weight*point_weight inside),
// W = sum_s g_s * prod_d omega_d  (so d out_weight += W*pw, d point_weight += W*ow).
// `fetch(off)` returns ds_dout at column-major offset `off` of this pose.
template <typename T, int NI, int NO, typename Fetch>
__device__ __forceinline__ void point_backward(const int (&ref0)[NO], const T (&dlo)[NO],
                                               const GridDesc<NO>& gd, T ow, T pw, Fetch fetch,
                                               T (&scaled)[NO], T& dow_part, T& dpw_part) {
    T dcoord[NO];
#pragma unroll
    for (int n = 0; n < NO; ++n) dcoord[n] = T(0);
    dow_part = T(0);
    dpw_part = T(0);
    // All 2^N gathers are requested before the first is used: with the fetch inside the
    // "neighbour is in the grid" branch every gather waited for the one before it (2^N dependent
    // round trips per point and pose).  A dropped neighbour fetches cell 0 and adds nothing
    // (selects, not multiplications by zero: weights may be non-finite).
    int off[1 << NO];
    T gq[1 << NO];
#pragma unroll
    for (int s = 0; s < (1 << NO); ++s) {
        off[s] = nbr_offset<NO>(ref0, s, gd);
        gq[s] = fetch(off[s] < 0 ? 0 : off[s]);
    }
#pragma unroll
    for (int s = 0; s < (1 << NO); ++s) {
        const bool in = off[s] >= 0;
        const T gi = gq[s];
        const T dweight = voxel_weight<T, NO>(dlo, s, gi);
        dow_part += in ? dweight * pw : T(0);
        dpw_part += in ? dweight * ow : T(0);
        const T factor = gi * ow * pw;
#pragma unroll
        for (int n = 0; n < NO; ++n)
            dcoord[n] += in ? factor * interp_weight<T, NO>(n, dlo, s) : T(0);
    }
#pragma unroll
    for (int n = 0; n < NO; ++n) scaled[n] = dcoord[n] * (T(gd.n[n]) / T(2));
}

// Fused residual sensitivity (dpr_raster_residual_pullback_*): when `target` is set the
// grid argument of a pullback kernel is the forward result `out` and the sensitivity is formed
// on the fly, ds_dout = scale * (out - target)  (README.md:151 computes exactly this grid on
// the host with scale = -2; examples/logo.jl:40-44 is the scale = +2 case), so ds_dout is
// never written to or re-read from HBM.  loss[b] = sum((out - target)^2) over pose b.
template <typename T> struct Residual {
    const T* target;  // nullptr: the grid argument already is ds_dout
    T scale;
    T* loss;          // B values or nullptr
};

// ---- 64-bit fixed-point LDS accumulators for fp32 data ---------------------------------------
// ds_add_u64 retires a wave-instruction in 13.6 cycles, ds_add_f64 in 26.3 (ds_add_f32: 193;
// profiles/r02_microbench_lds_conflicts.txt), and integer sums are EXACT: the forward result no
// longer depends on the order in which a tile's records are accumulated.  A contribution v (an
// fp32 value, exactly representable in f64) becomes round(v * 2^sexp) through the magic-number
// trick: fma(v, 2^sexp, 1.5 * 2^52) has the integer in its low mantissa bits as long as
// |v * 2^sexp| < 2^51, so the conversion costs one v_fma_f64 and one 32-bit subtract.
// The scale is chosen per work item from
//   maxw >= |every contribution|  (|out_weight| * max|point_weight|; the products of the deltas
//                                  are <= 1 in fp32 as well) and
//   n    >= contributions to one cell (a record adds to a cell at most once: n = records of the item)
// such that n * maxw * 2^sexp <= 2^62 (the sum cannot overflow) and maxw * 2^sexp <= 2^50.
// An item holds at most max(4096, P / 256) < 2^24 records, so a contribution keeps at least 38
// bits below the largest weight (fp32 carries 24); typical tiles (thousands of records) keep 49.
// Non-finite weights (NaN / Inf out_weight or point_weight) switch the item to f64 atomics, which
// propagate them the IEEE way.
struct FixScale {
    double mul;  // 2^sexp, 0 = fixed point off (f64 atomics)
    double inv;  // 2^-sexp
};
constexpr double kFixMagic = 6755399441055744.0;  // 1.5 * 2^52
// The scale comes from the LARGEST weight of the scope it is chosen for (the call on the tiled path, the
// candidate chunks of a tile on the owner path, the 4096-point chunk on the 2-D chunk-owner path).  Where the
// non-zero |point_weight| of that scope span more than 2^10 the scope uses f64 atomics instead: otherwise
// cells that only small-weight points reach would lose relative precision against the reference's float
// atomics (/root/reference/src/raster.jl:62-64), down to exactly 0 below 2^-38 of the largest weight.
// With the guard every contribution keeps at least 28 bits (typically 39) below the SMALLEST weight.
constexpr float kFixMaxWeightRange = 1024.f;
// maxw, or +Inf (= no fixed point) when maxw > minw * range or either is NaN; minw = +Inf: no non-zero weight
__device__ __forceinline__ float fix_guard_range(float maxw, float minw) {
    return (maxw <= minw * kFixMaxWeightRange) ? maxw : __builtin_inff();
}
// Running (max, min non-zero) of |point_weight| in ONE register: a = upper 16 bits of |w| (sign 0, 8
// exponent, 7 mantissa bits; NaN / Inf sort on top) -- high half: max a; low half: max of (0x10000 - a)
// over a >= 1, i.e. the minimum turned into a maximum (0 = nothing but zeros seen).  Both halves merge
// with one v_pk_max_u16.  The bounds that come back out are conservative by construction: wrange_max
// >= every |w| (within 2^-7), wrange_min <= every non-zero |w| -- which is all the scale (an upper
// bound of the contributions) and the range guard need.  (Two full-precision words cost the binning
// kernels two live registers and pushed three of them into scratch.)
typedef unsigned short dpr_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t wrange_key(float w) {
    const uint32_t a = __float_as_uint(fabsf(w)) >> 16;
    return (a << 16) | (((a - 1u) ^ 0xffffu) & 0xffffu);
}
__device__ __forceinline__ uint32_t wrange_merge(uint32_t x, uint32_t y) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(dpr_u16x2, x),
                                                                  __builtin_bit_cast(dpr_u16x2, y)));
}
__device__ __forceinline__ uint32_t wrange_wave(uint32_t key) {  // all 64 lanes must call
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) key = wrange_merge(key, (uint32_t)__shfl_xor((int)key, o, kWave));
    return key;
}
// from the two halves (as published / reduced separately: hi = key >> 16, lo = key & 0xffff)
__device__ __forceinline__ float wrange_max(uint32_t hi) { return __uint_as_float((hi + 1u) << 16); }
__device__ __forceinline__ float wrange_min(uint32_t lo) {
    return lo ? __uint_as_float((0x10000u - lo) << 16) : __builtin_inff();
}
// largest |point_weight| to scale by, or +Inf (f64 atomics) when the range guard trips
__device__ __forceinline__ float wrange_guarded_max(uint32_t hi, uint32_t lo) {
    return fix_guard_range(wrange_max(hi), wrange_min(lo));
}
constexpr int kFixNone = 1 << 30;  // "no fixed point" exponent
// exponent sexp of the scale 2^sexp, or kFixNone
__device__ __forceinline__ int fix_exponent(float maxw, uint32_t n, int enabled) {
    if (!enabled || !(maxw < __builtin_inff())) return kFixNone;
    int e = 0;
    if (maxw > 0.f) (void)frexpf(maxw, &e);  // maxw < 2^e
    const int bits_n = 32 - __clz((int)(n | 1u));  // n < 2^bits_n
    const int sexp = 62 - bits_n;
    return (sexp > 50 ? 50 : sexp) - e;
}
// (the exponent is wave-uniform by contract: forcing it into an SGPR keeps the scale -- built
// from integer bits, |sexp| < 256 -- out of the vector registers)
__device__ __forceinline__ FixScale fix_scale_from_exponent(int sexp) {
    sexp = __builtin_amdgcn_readfirstlane(sexp);
    FixScale fs;
    fs.mul = sexp == kFixNone ? 0.0 : __longlong_as_double((long long)(1023 + sexp) << 52);
    fs.inv = sexp == kFixNone ? 0.0 : __longlong_as_double((long long)(1023 - sexp) << 52);
    return fs;
}
__device__ __forceinline__ FixScale fix_scale(float maxw, uint32_t n, int enabled) {
    return fix_scale_from_exponent(fix_exponent(maxw, n, enabled));
}
__device__ __forceinline__ unsigned long long fix_bits(float v, const FixScale& fs) {
    const double x = fma((double)v, fs.mul, kFixMagic);
    return (unsigned long long)(__double_as_longlong(x) - __double_as_longlong(kFixMagic));
}
__device__ __forceinline__ double fix_value(double cell, const FixScale& fs) {
    return fs.mul != 0.0 ? (double)__double_as_longlong(cell) * fs.inv : cell;
}
// one contribution into an f64-sized LDS cell
template <bool FIX, typename T>
__device__ __forceinline__ void cell_add(double* cell, T v, const FixScale& fs) {
    if constexpr (FIX && sizeof(T) == 4) atomicAdd((unsigned long long*)cell, fix_bits((float)v, fs));
    else atomicAdd(cell, (double)v);
}
// explicit fused multiply-add (the library is compiled with -ffp-contract=off; see the call sites
// for why a given sum may fuse)
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

// wave-level sum (all 64 lanes must call)
template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// f32 wave sum through the DPP crossbar (row shifts, then the two row broadcasts): the total
// arrives in lane 63; no LDS traffic, ~12 full-rate instructions (ds_bpermute-based __shfl_xor
// is a chain of six LDS round trips).  All 64 lanes must call.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_shift_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float wave_sum_lane63(float v) {
    v = dpp_shift_add<0x111, 0xf>(v);  // row_shr:1
    v = dpp_shift_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_shift_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_shift_add<0x118, 0xf>(v);  // row_shr:8   (lane 15 of each row: the row's sum)
    v = dpp_shift_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1, 3
    v = dpp_shift_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2, 3
    return v;
}

template <typename T> __device__ __forceinline__ void atomic_add(T* addr, T v) {
    unsafeAtomicAdd(addr, v);  // native global_atomic_add_f32 / _f64 on gfx950
}

}  // namespace dpr
