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
// csrc/conv1d_ring.hip: LDS-DMA ring variant (bf16, c_in % 64 == 0, k >= 2); VITS_E_UNSUPPORTED = take another kernel
int conv1d_ring_dispatch(const vits_conv_desc& d, int t_out, hipStream_t s);
// ... `count` problems of one kernel instance side by side in ONE launch (all or nothing)
int conv1d_ring_multi_dispatch(const vits_conv_desc* d, const int* t_out, int count, hipStream_t s);

// csrc/conv1d_wgrad_ring.hip: large-tile, deep-prefetch weight-gradient kernel (bf16); plan.TC == 0: not eligible
struct WgradRingPlan { int TC, TK, KT, XR, S; };
WgradRingPlan wgrad_ring_plan(const vits_wgrad_desc& d, int t_out, int s_max);
int wgrad_ring_launch(const vits_wgrad_desc& d, int t_out, const WgradRingPlan& p, float* partial, float* partial_db, size_t slab, hipStream_t s);

// Raise a kernel's dynamic-LDS limit to the hardware maximum (a per-device function attribute): once per device and
// kernel, to a FIXED value — a per-launch value would be whatever the LAST call set by the time a captured graph replays.
inline hipError_t ensure_max_dynamic_lds(const void* kern, int reserve_static = 0) {
  return hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesMax - reserve_static);
}

}  // namespace vits
