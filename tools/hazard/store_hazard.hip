// Isolated reproducer for the hazard behind the spline backward's irreproducible `gh` elements:
//   global_store_dwordx4 vDATA ; s_nop 1 ; v_* vDATA+1 (overwrite of the store's data registers)
// which is what hipcc (ROCm 7.2, gfx950) emitted in spline_kernel<bf16, BWD, FLOW>.  LLVM's hazard recognizer asks for 2 wait
// states between a >64-bit VMEM store and a VALU write of its data VGPRs on gfx940+.  This program issues that exact sequence
// with NOPS wait states while a second stream streams through HBM, and counts store words that arrive with the overwritten
// value.  Build: hipcc --offload-arch=gfx950 -O2 store_hazard.hip -o store_hazard ; run: ./store_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

constexpr unsigned JUNK = 0xdeadbeefu;

// one thread = one 64-byte row per iteration; the x4 store goes to bytes [16, 32) of the row like the second store of the spline
template <int NOPS>
__global__ void victim(unsigned* out, int rows_per_iter, int iters) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    unsigned* p = out + ((size_t)it * rows_per_iter + e) * 16 + 4;
    const unsigned a = 0x10000000u + (unsigned)e * 4u + (unsigned)it * 0x10000u;
    if (NOPS == 1)
      asm volatile("v_add_u32 v20, %1, 0\n v_add_u32 v21, %1, 1\n v_add_u32 v22, %1, 2\n v_add_u32 v23, %1, 3\n s_nop 4\n"
                   "global_store_dwordx4 %0, v[20:23], off\n s_nop 1\n"
                   "v_mov_b32 v21, %2\n v_mov_b32 v20, %2\n v_mov_b32 v23, %2\n v_mov_b32 v22, %2\n"
                   :: "v"(p), "v"(a), "v"(JUNK) : "v20", "v21", "v22", "v23", "memory");
    else if (NOPS == 0)
      asm volatile("v_add_u32 v20, %1, 0\n v_add_u32 v21, %1, 1\n v_add_u32 v22, %1, 2\n v_add_u32 v23, %1, 3\n s_nop 4\n"
                   "global_store_dwordx4 %0, v[20:23], off\n"
                   "v_mov_b32 v21, %2\n v_mov_b32 v20, %2\n v_mov_b32 v23, %2\n v_mov_b32 v22, %2\n"
                   :: "v"(p), "v"(a), "v"(JUNK) : "v20", "v21", "v22", "v23", "memory");
    else
      asm volatile("v_add_u32 v20, %1, 0\n v_add_u32 v21, %1, 1\n v_add_u32 v22, %1, 2\n v_add_u32 v23, %1, 3\n s_nop 4\n"
                   "global_store_dwordx4 %0, v[20:23], off\n s_nop 7\n"
                   "v_mov_b32 v21, %2\n v_mov_b32 v20, %2\n v_mov_b32 v23, %2\n v_mov_b32 v22, %2\n"
                   :: "v"(p), "v"(a), "v"(JUNK) : "v20", "v21", "v22", "v23", "memory");
  }
}

__global__ void aggressor(const float4* __restrict__ src, float4* __restrict__ dst, size_t n, int reps) {
  for (int r = 0; r < reps; ++r)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
      float4 v = src[i]; v.x += 1.f; dst[i] = v;
    }
}

template <int NOPS>
static void run(const char* name, bool with_aggressor) {
  const int blocks = 52, threads = 128, iters = 512, rounds = 40;
  const int rows = blocks * threads;
  const size_t words = (size_t)rows * iters * 16;
  unsigned* out; CK(hipMalloc(&out, words * 4));
  const size_t n4 = (size_t)(512u << 20) / 16;
  float4 *src, *dst; CK(hipMalloc(&src, n4 * 16)); CK(hipMalloc(&dst, n4 * 16)); CK(hipMemset(src, 0, n4 * 16));
  hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
  std::vector<unsigned> h(words);
  long bad[4] = {0, 0, 0, 0}, other = 0, total = 0;
  for (int r = 0; r < rounds; ++r) {
    CK(hipMemsetAsync(out, 0, words * 4, s0)); CK(hipStreamSynchronize(s0));
    if (with_aggressor) hipLaunchKernelGGL(aggressor, dim3(1024), dim3(256), 0, s1, src, dst, n4, 2);
    hipLaunchKernelGGL(victim<NOPS>, dim3(blocks), dim3(threads), 0, s0, out, rows, iters);
    CK(hipGetLastError()); CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), out, words * 4, hipMemcpyDeviceToHost));
    for (int it = 0; it < iters; ++it)
      for (int e = 0; e < rows; ++e) {
        const unsigned a = 0x10000000u + (unsigned)e * 4u + (unsigned)it * 0x10000u;
        const unsigned* p = &h[((size_t)it * rows + e) * 16 + 4];
        for (int j = 0; j < 4; ++j) { ++total; if (p[j] != a + j) { if (p[j] == JUNK) ++bad[j]; else ++other; } }
      }
  }
  printf("%-28s aggressor=%d  stores=%ld words: overwritten-value seen in dword0..3 = %ld %ld %ld %ld, other mismatches = %ld\n",
         name, (int)with_aggressor, total, bad[0], bad[1], bad[2], bad[3], other);
  CK(hipFree(out)); CK(hipFree(src)); CK(hipFree(dst));
}

int main() {
  run<1>("store; s_nop 1; overwrite", false);
  run<1>("store; s_nop 1; overwrite", true);
  run<0>("store; overwrite (no nop)", true);
  run<7>("store; s_nop 7; overwrite", true);
  return 0;
}
