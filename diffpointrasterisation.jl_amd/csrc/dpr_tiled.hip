// DPR_ALGO_TILED -- placeholder translation unit (filled in below in later commits)
#include "../../include/dpr.h"
#include "dpr_tiled.h"

namespace dpr {

bool tiled_preferred(int, int, int64_t, int64_t, int64_t) { return false; }

size_t tiled_workspace_bytes(size_t, int, int, int, const int64_t*, int64_t, int64_t) { return 0; }

template <typename T, int NI, int NO>
int raster_tiled(hipStream_t, const int64_t*, int64_t, int64_t, int64_t, T*, const T*, const T*,
                 const T*, const T*, const T*, const T*, void*, size_t) {
    return fail(DPR_ERR_UNSUPPORTED_ALGO, "DPR_ALGO_TILED not built");
}

template <typename T, int NI, int NO>
int pullback_tiled(hipStream_t, const int64_t*, int64_t, int64_t, int64_t, const T*, const T*,
                   const T*, const T*, const T*, const T*, T*, T*, T*, T*, T*, T*, void*, size_t) {
    return fail(DPR_ERR_UNSUPPORTED_ALGO, "DPR_ALGO_TILED not built");
}

#define DPR_INST(T, NI, NO)                                                                      \
    template int raster_tiled<T, NI, NO>(hipStream_t, const int64_t*, int64_t, int64_t, int64_t, \
                                         T*, const T*, const T*, const T*, const T*, const T*,   \
                                         const T*, void*, size_t);                               \
    template int pullback_tiled<T, NI, NO>(hipStream_t, const int64_t*, int64_t, int64_t,        \
                                           int64_t, const T*, const T*, const T*, const T*,      \
                                           const T*, const T*, T*, T*, T*, T*, T*, T*, void*,    \
                                           size_t);
DPR_INST(float, 2, 2)
DPR_INST(float, 3, 3)
DPR_INST(float, 3, 2)
DPR_INST(double, 2, 2)
DPR_INST(double, 3, 3)
DPR_INST(double, 3, 2)
}  // namespace dpr
