// placeholder until the k-means kernels land
#include "hvs_internal.h"
extern "C" int hvs_kmeans_assign(hvs_index *, const float *, int, int32_t *, float *, int64_t *, int) {
    hvs_set_error("hvs_kmeans_assign: not built");
    return 3;
}
