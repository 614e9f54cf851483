// placeholder until the MFMA path lands
#include "hvs_internal.h"
bool hvs_mfma_supported(hvs_index *, int, int, int) { return false; }
int hvs_search_mfma(hvs_index *, const float *, const float *, int, int, int, int32_t, const uint32_t *, float, int64_t,
                    int64_t, float *, int64_t *, int32_t *) {
    hvs_set_error("MFMA path not built");
    return 3;
}
