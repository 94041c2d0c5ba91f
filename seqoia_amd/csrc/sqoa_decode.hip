// sqoa_decode.hip -- SQOA / QOI batch decoder for gfx950 (placeholder until the kernel lands).
#include "sqoa_common.h"

extern "C" hipError_t sqoa_launch_decode(const sqoa_hip_job*, const int*, int n, sqoa_hip_result*, hipStream_t) {
    return n > 0 ? hipErrorNotSupported : hipSuccess;
}
