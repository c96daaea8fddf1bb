// orb_kernels.hip — placeholder until the ORB / matcher kernels land (next commits).
#include "cvhip_internal.hpp"
using namespace cvhip;
extern "C" int cvhip_orb_extract(cvhip_device *, const uint8_t *, uint32_t, uint32_t, uint32_t, uint32_t *,
                                 uint32_t *, uint32_t *)
{
    return fail(CVHIP_ERR_UNSUPPORTED, "cvhip_orb_extract: not implemented yet");
}
extern "C" int cvhip_match_points(cvhip_device *, const uint32_t *, const uint32_t *, uint32_t, const uint32_t *,
                                  const uint32_t *, uint32_t, uint32_t, uint32_t *, uint32_t *, uint32_t *)
{
    return fail(CVHIP_ERR_UNSUPPORTED, "cvhip_match_points: not implemented yet");
}
