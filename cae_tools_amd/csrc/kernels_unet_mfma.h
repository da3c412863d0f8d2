// kernels_unet_mfma.h — LDS-tiled MFMA convolutions for the 4x4 / stride 2 / padding 1 layers of the UNET path.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_unet.h"

namespace unet {

inline bool mfma_down_eligible(const Geom&) { return false; }
inline bool mfma_up_eligible(const Geom&) { return false; }
inline bool mfma_wgrad_eligible(const Geom&) { return false; }
inline void mfma_down_launch(const Geom&, const float*, const float*, const float*, float*, hipStream_t) {}
inline void mfma_up_launch(const Geom&, const float*, const float*, const float*, float*, hipStream_t) {}
inline void mfma_wgrad_launch(const Geom&, const float*, const float*, double*, hipStream_t) {}

}  // namespace unet
