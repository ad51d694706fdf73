// mcr_aggregate.hip — device-side aggregation (quantile bands, histogram) — placeholder:
// the real kernels land in the next commit; until then the entry points fail loudly.
#include "../../include/mcr.h"
#include "mcr_host.h"

using namespace mcr;

extern "C" {
int64_t mcr_row_quantiles_scratch_bytes(int32_t, int32_t) { return 0; }
int mcr_row_quantiles(const double*, int64_t, int32_t, int64_t, const double*, int32_t, double*, uint64_t*,
                      void*, int, void*) {
    set_error("mcr_row_quantiles: not built yet");
    return MCR_ERR_UNSUPPORTED;
}
int mcr_minmax_success(const double*, const uint8_t*, int64_t, double*, int, void*) {
    set_error("mcr_minmax_success: not built yet");
    return MCR_ERR_UNSUPPORTED;
}
int mcr_histogram_success(const double*, const uint8_t*, int64_t, const double*, int32_t, uint64_t*, int, void*) {
    set_error("mcr_histogram_success: not built yet");
    return MCR_ERR_UNSUPPORTED;
}
}
