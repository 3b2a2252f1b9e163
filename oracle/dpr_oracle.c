/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of DiffPointRasterisation.jl's `raster!` /
 * `raster_pullback!` arithmetic (reference @ 2024_10_08; pure Julia, cannot be
 * executed in this pipeline -- no Julia runtime in the image).
 *
 * Pinned by the reference's own known-answer tests (tests/golden/ *.json,
 * transcribed from src/raster.jl:143-309 and README.md:41-68, 84-183) --
 * see tests/test_oracle_golden.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (libdpr.so) never links or calls it.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define REAL float
#define SUF f32
#define CEIL ceilf
#include "dpr_oracle_impl.h"
#undef REAL
#undef SUF
#undef CEIL

#define REAL double
#define SUF f64
#define CEIL ceil
#include "dpr_oracle_impl.h"
#undef REAL
#undef SUF
#undef CEIL

/* src/util.jl:7-8 digitstuple / :26-27 voxel_shifts: neighbour k (0-based) has
 * component d = bit d of k (dim 1 = LSB).  out is (2^n) x n, row-major. */
void oracle_voxel_shifts(int n, int64_t *out)
{
    for (int k = 0; k < (1 << n); ++k)
        for (int d = 0; d < n; ++d) out[k * n + d] = (k >> d) & 1;
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
