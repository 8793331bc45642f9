// Placeholder entry points for the ExactOctreeSdf path until exact_build.hip / exact_query.hip land.
#include "sdfhip_internal.h"
using namespace sdfhip;
extern "C" {
int sdfhip_exact_build(sdfhip_ctx*, sdfhip_mesh*, const float*, const float*, uint32_t, uint32_t, uint32_t, sdfhip_exact**) { setError("ExactOctreeSdf build is not provided yet"); return SDFHIP_E_UNSUPPORTED; }
int sdfhip_exact_destroy(sdfhip_exact*) { return SDFHIP_OK; }
int sdfhip_exact_get_info(sdfhip_exact*, sdfhip_exact_info*) { setError("ExactOctreeSdf is not provided yet"); return SDFHIP_E_UNSUPPORTED; }
int sdfhip_exact_download(sdfhip_exact*, uint32_t*, uint8_t*, uint32_t*, uint8_t*) { setError("ExactOctreeSdf is not provided yet"); return SDFHIP_E_UNSUPPORTED; }
int sdfhip_exact_query(sdfhip_exact*, const float*, uint64_t, float*, float*, uint32_t*, int) { setError("ExactOctreeSdf is not provided yet"); return SDFHIP_E_UNSUPPORTED; }
int sdfhip_is_near_minimize(sdfhip_ctx*, const float*, const float*, const float*, const float*, uint64_t, uint8_t*) { setError("not provided yet"); return SDFHIP_E_UNSUPPORTED; }
}
