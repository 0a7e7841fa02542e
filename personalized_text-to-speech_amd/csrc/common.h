// Shared host/device helpers for libvitsmi.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vitsmi.h"

namespace vits {

constexpr int kWave = 64;
constexpr int kLdsBytesMax = 160 * 1024;   // MI355X: 160 KiB LDS per CU

// Records the text of a HIP error for vits_last_error() and maps it to VITS_E_LAUNCH.
int note_hip_error(hipError_t e, const char* where);

inline int check_launch(const char* where) {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VITS_OK : note_hip_error(e, where);
}

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// csrc/conv1d_flat.hip: flat-row (gathering) variant of the channels-last convolution
int conv1d_flat_dispatch(const vits_conv_desc& d, int t_out, hipStream_t s);

}  // namespace vits
