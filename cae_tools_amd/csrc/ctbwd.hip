// ctbwd.hip - the LDS-staged backward kernel of the channel-rich decoder layers (kernels_ctbwd.h) in a code object of its own.
// It is an opt-in (cae_set_kernel_mode bit 1 / CAE_CTBWD), and compiled into engine.hip's code object its mere presence moved
// the default path's kernels and cost 1.2 us per step (178.9 against 177.7: measured with and without it, twice each).
// The shared device helpers come from the same headers, wrapped in a namespace of this file so that their kernels' host
// stubs do not collide with engine.hip's.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <type_traits>

namespace ctbtu {
#include "kernels_generic.h"
#include "kernels_gemm.h"
#define CAE_CTBWD_KERNEL 1
#include "kernels_ctbwd.h"
}  // namespace ctbtu

namespace cae_internal {

// args: engine.hip's cae::CtBwd (the same header, hence the same layout)
int ctbwd_launch(const void* args, size_t bytes, unsigned gx, unsigned gy, unsigned gz, size_t lds, hipStream_t s) {
    ctbtu::cae::CtBwd c;
    if (bytes != sizeof c) return -1;
    memcpy(&c, args, sizeof c);
    static size_t granted = 64 * 1024, granted_band = 64 * 1024;
    if (c.bands > 1) {   // one band of one image per workgroup
        if (lds > granted_band) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ctbtu::cae::k_ct_bwd_band), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            granted_band = lds;
        }
        hipLaunchKernelGGL(ctbtu::cae::k_ct_bwd_band, dim3(gx, gy, gz), dim3(ctbtu::cae::kCtbThreads), lds, s, c);
        return 0;
    }
    if (lds > granted) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ctbtu::cae::k_ct_bwd_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        granted = lds;
    }
    hipLaunchKernelGGL(ctbtu::cae::k_ct_bwd_lds, dim3(gx, gy, gz), dim3(ctbtu::cae::kCtbThreads), lds, s, c);
    return 0;
}

}  // namespace cae_internal
