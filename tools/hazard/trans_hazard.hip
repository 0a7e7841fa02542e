// Isolated reproducer for the irreproducible lanes 48..63 of the spline backward: a transcendental VALU result
// consumed by the next VALU instruction behind hipcc's `s_nop 0`:
//     v_rcp_f32 vR, vB ; s_nop N ; v_fma_f32 vE, -vB, vR, 1.0          (E = 1 - B * rcp(B) must be ~0)
// vR holds a STALE reciprocal (of another number) before the v_rcp, exactly as in the kernel's unrolled divisions, which all
// reuse one register.  If the consumer reads vR before the transcendental pipe has written its last lanes, |E| is large there.
// A second stream runs a transcendental-heavy kernel (the WaveNet gates' exp / rcp) on every CU at the same time.
// Build: hipcc --offload-arch=gfx950 -O2 trans_hazard.hip -o trans_hazard ; run: ./trans_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

template <int NOPS>
__global__ void victim(float* worst, int iters) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  float b = 1.0f + 0.001f * (float)(threadIdx.x & 63), w = 0.f;
  for (int it = 0; it < iters; ++it) {
    float stale = 37.0f + (float)it, r, err;
    if (NOPS == 0)
      asm volatile("v_mov_b32 %0, %3\n s_nop 4\n v_rcp_f32 %0, %2\n s_nop 0\n v_fma_f32 %1, -%2, %0, 1.0\n s_nop 4" : "=&v"(r), "=&v"(err) : "v"(b), "v"(stale));
    else if (NOPS == 1)
      asm volatile("v_mov_b32 %0, %3\n s_nop 4\n v_rcp_f32 %0, %2\n s_nop 1\n v_fma_f32 %1, -%2, %0, 1.0\n s_nop 4" : "=&v"(r), "=&v"(err) : "v"(b), "v"(stale));
    else if (NOPS == 3)
      asm volatile("v_mov_b32 %0, %3\n s_nop 4\n v_rcp_f32 %0, %2\n s_nop 3\n v_fma_f32 %1, -%2, %0, 1.0\n s_nop 4" : "=&v"(r), "=&v"(err) : "v"(b), "v"(stale));
    else if (NOPS == -1)   /* no nop at all: is the hardware interlocked? */
      asm volatile("v_mov_b32 %0, %3\n s_nop 4\n v_rcp_f32 %0, %2\n v_fma_f32 %1, -%2, %0, 1.0\n s_nop 4" : "=&v"(r), "=&v"(err) : "v"(b), "v"(stale));
    else
      asm volatile("v_mov_b32 %0, %3\n s_nop 4\n v_rcp_f32 %0, %2\n s_nop 7\n v_fma_f32 %1, -%2, %0, 1.0\n s_nop 4" : "=&v"(r), "=&v"(err) : "v"(b), "v"(stale));
    w = fmaxf(w, fabsf(err));
    b += 0.0001f;
  }
  worst[e] = w;
}

// the pattern hipcc emitted in spline_kernel (two divisions vectorised together):
//     v_rcp_f32 vR0, vB0 ; v_rcp_f32 vR1, vB1 ; s_nop N ; v_fma_f32 vE, -vB1, vR1, 1.0      (consumer of the SECOND result)
template <int NOPS>
__global__ void victim2(float* worst, int iters) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  float b = 1.0f + 0.001f * (float)(threadIdx.x & 63), w = 0.f;
  for (int it = 0; it < iters; ++it) {
    float stale = 37.0f + (float)it, r0, r1, err;
    const float b0 = b * 1.7f;
    if (NOPS == 0)
      asm volatile("v_mov_b32 %1, %5\n s_nop 4\n v_rcp_f32 %0, %4\n v_rcp_f32 %1, %3\n s_nop 0\n v_fma_f32 %2, -%3, %1, 1.0\n s_nop 4" : "=&v"(r0), "=&v"(r1), "=&v"(err) : "v"(b), "v"(b0), "v"(stale));
    else if (NOPS == 1)
      asm volatile("v_mov_b32 %1, %5\n s_nop 4\n v_rcp_f32 %0, %4\n v_rcp_f32 %1, %3\n s_nop 1\n v_fma_f32 %2, -%3, %1, 1.0\n s_nop 4" : "=&v"(r0), "=&v"(r1), "=&v"(err) : "v"(b), "v"(b0), "v"(stale));
    else if (NOPS == 2)
      asm volatile("v_mov_b32 %1, %5\n s_nop 4\n v_rcp_f32 %0, %4\n v_rcp_f32 %1, %3\n s_nop 2\n v_fma_f32 %2, -%3, %1, 1.0\n s_nop 4" : "=&v"(r0), "=&v"(r1), "=&v"(err) : "v"(b), "v"(b0), "v"(stale));
    else
      asm volatile("v_mov_b32 %1, %5\n s_nop 4\n v_rcp_f32 %0, %4\n v_rcp_f32 %1, %3\n s_nop 4\n v_fma_f32 %2, -%3, %1, 1.0\n s_nop 4" : "=&v"(r0), "=&v"(r1), "=&v"(err) : "v"(b), "v"(b0), "v"(stale));
    w = fmaxf(w, fabsf(err) + 0.f * r0);
    b += 0.0001f;
  }
  worst[e] = w;
}

// the exact consumer of the kernel: a PACKED fma over the register pair the two reciprocals wrote
//     v_rcp_f32 v14, v12 ; v_rcp_f32 v15, v13 ; s_nop N ; v_pk_fma_f32 v[4:5], v[12:13], v[14:15], 1.0 (-a*b+1)
template <int NOPS>
__global__ void victim3(float* worst, int iters) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  float b = 1.0f + 0.001f * (float)(threadIdx.x & 63), w = 0.f;
  for (int it = 0; it < iters; ++it) {
    float stale = 37.0f + (float)it, e0, e1;
    const float b0 = b * 1.7f;
#define SEQ(NOP) "v_mov_b32 v12, %2\n v_mov_b32 v13, %3\n v_mov_b32 v14, %4\n v_mov_b32 v15, %4\n s_nop 4\n" \
                 "v_rcp_f32 v14, v12\n v_rcp_f32 v15, v13\n " NOP "v_pk_fma_f32 v[4:5], v[12:13], v[14:15], 1.0 op_sel_hi:[1,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n" \
                 "s_nop 4\n v_mov_b32 %0, v4\n v_mov_b32 %1, v5\n"
    if (NOPS == 0) asm volatile(SEQ("s_nop 0\n ") : "=v"(e0), "=v"(e1) : "v"(b0), "v"(b), "v"(stale) : "v4", "v5", "v12", "v13", "v14", "v15");
    else if (NOPS == 1) asm volatile(SEQ("s_nop 1\n ") : "=v"(e0), "=v"(e1) : "v"(b0), "v"(b), "v"(stale) : "v4", "v5", "v12", "v13", "v14", "v15");
    else if (NOPS == 2) asm volatile(SEQ("s_nop 2\n ") : "=v"(e0), "=v"(e1) : "v"(b0), "v"(b), "v"(stale) : "v4", "v5", "v12", "v13", "v14", "v15");
    else asm volatile(SEQ("s_nop 5\n ") : "=v"(e0), "=v"(e1) : "v"(b0), "v"(b), "v"(stale) : "v4", "v5", "v12", "v13", "v14", "v15");
    w = fmaxf(w, fmaxf(fabsf(e0), fabsf(e1)));
    b += 0.0001f;
  }
  worst[e] = w;
}

// WAR: the SOURCE register of a transcendental is overwritten by the next VALU instruction while the (quarter-rate, possibly
// queued behind other waves' transcendentals) op may still have lanes to read:
//     v_rcp_f32 vR, vB ; [s_nop N] ; v_mov_b32 vB, junk ; ... ; check vR * b == 1
template <int NOPS>
__global__ void victim4(float* worst, int iters) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  float b = 1.0f + 0.001f * (float)(threadIdx.x & 63), w = 0.f;
  for (int it = 0; it < iters; ++it) {
    float r, src;
    const float junk = 1.0e10f;
#define WAR(NOP) "v_mov_b32 %1, %2\n s_nop 4\n v_rcp_f32 %0, %1\n " NOP "v_mov_b32 %1, %3\n s_nop 7\n"
    if (NOPS < 0) asm volatile(WAR("") : "=&v"(r), "=&v"(src) : "v"(b), "v"(junk));
    else if (NOPS == 0) asm volatile(WAR("s_nop 0\n ") : "=&v"(r), "=&v"(src) : "v"(b), "v"(junk));
    else if (NOPS == 1) asm volatile(WAR("s_nop 1\n ") : "=&v"(r), "=&v"(src) : "v"(b), "v"(junk));
    else if (NOPS == 4) asm volatile(WAR("s_nop 4\n ") : "=&v"(r), "=&v"(src) : "v"(b), "v"(junk));
    else asm volatile(WAR("s_nop 7\n s_nop 7\n ") : "=&v"(r), "=&v"(src) : "v"(b), "v"(junk));
    w = fmaxf(w, fabsf(fmaf(-b, r, 1.0f)) + 0.f * src);
    b += 0.0001f;
  }
  worst[e] = w;
}

// Packed fp32 with a scalar operand, the instruction that fed the wrong stores (v_pk_mul_f32 v[..], s[0:1], v[..] op_sel_hi:[0,1]):
//   KIND 0: s_mov s10, h ; s_mov s11, junk ; v_pk_mul_f32 (SALU write right in front of the read)
//   KIND 1: v_pk_mul_f32 ; s_mov s10, junk ; s_mov s11, junk (SALU overwrites the pair right behind the read)
//   KIND 2: both
template <int KINDV>
__global__ void victim5(float* worst, int iters) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  float a = 1.0f + 0.001f * (float)(threadIdx.x & 63), w = 0.f;
  const float hs = 0.0721687836f, junk = 12345.0f;
  for (int it = 0; it < iters; ++it) {
    float r0, r1;
    const float a1 = a * 1.5f;
    if (KINDV == 0)
      asm volatile("v_mov_b32 v6, %2\n v_mov_b32 v7, %3\n s_nop 4\n s_mov_b32 s10, %4\n s_mov_b32 s11, %5\n"
                   "v_pk_mul_f32 v[4:5], s[10:11], v[6:7] op_sel_hi:[0,1]\n s_nop 7\n v_mov_b32 %0, v4\n v_mov_b32 %1, v5\n"
                   : "=v"(r0), "=v"(r1) : "v"(a), "v"(a1), "s"(hs), "s"(junk) : "v4", "v5", "v6", "v7", "s10", "s11");
    else if (KINDV == 1)
      asm volatile("v_mov_b32 v6, %2\n v_mov_b32 v7, %3\n s_mov_b32 s10, %4\n s_mov_b32 s11, %5\n s_nop 4\n"
                   "v_pk_mul_f32 v[4:5], s[10:11], v[6:7] op_sel_hi:[0,1]\n s_mov_b32 s10, %5\n s_mov_b32 s11, %5\n s_nop 7\n v_mov_b32 %0, v4\n v_mov_b32 %1, v5\n"
                   : "=v"(r0), "=v"(r1) : "v"(a), "v"(a1), "s"(hs), "s"(junk) : "v4", "v5", "v6", "v7", "s10", "s11");
    else
      asm volatile("v_mov_b32 v6, %2\n v_mov_b32 v7, %3\n s_nop 4\n s_mov_b32 s10, %4\n s_mov_b32 s11, %5\n"
                   "v_pk_mul_f32 v[4:5], s[10:11], v[6:7] op_sel_hi:[0,1]\n s_mov_b32 s10, %5\n s_mov_b32 s11, %5\n s_nop 7\n v_mov_b32 %0, v4\n v_mov_b32 %1, v5\n"
                   : "=v"(r0), "=v"(r1) : "v"(a), "v"(a1), "s"(hs), "s"(junk) : "v4", "v5", "v6", "v7", "s10", "s11");
    w = fmaxf(w, fmaxf(fabsf(r0 - hs * a), fabsf(r1 - hs * a1)));
    a += 0.0001f;
  }
  worst[e] = w;
}

__global__ void aggressor(float* out, int iters) {
  float v = 0.001f * (float)threadIdx.x, acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    acc += __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    v += 0.37f;
    acc += __builtin_amdgcn_rcpf(1.0f + __expf(-v * 0.5f));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int NOPS, int KIND = 0>
static void run(const char* name, int mode) {          // mode 0: 1 wave per SIMD at most (52 x 2 waves), alone; 1: + aggressor stream; 2: + aggressor, 1024-thread blocks;
                                                       // 3: the victim itself with 8 waves per SIMD (8192 waves): its waves contend with each other for the transcendental pipe
  const int blocks = mode == 3 ? 2048 : 52, threads = mode == 3 ? 256 : 128, iters = mode == 3 ? 4000 : 20000, rounds = mode == 3 ? 5 : 20;
  const int n = blocks * threads;
  float* worst; CK(hipMalloc(&worst, n * 4));
  float* junk; CK(hipMalloc(&junk, 2048 * 1024 * 4));
  hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
  std::vector<float> h(n);
  long bad_by_quarter[4] = {0, 0, 0, 0};
  float maxerr = 0.f;
  for (int r = 0; r < rounds; ++r) {
    if (mode == 1 || mode == 2) hipLaunchKernelGGL(aggressor, dim3(2048), dim3(mode == 1 ? 256 : 1024), 0, s1, junk, 4000);
    if (KIND == 4) hipLaunchKernelGGL(victim5<NOPS>, dim3(blocks), dim3(threads), 0, s0, worst, iters);
    else if (KIND == 3) hipLaunchKernelGGL(victim4<NOPS>, dim3(blocks), dim3(threads), 0, s0, worst, iters);
    else if (KIND == 2) hipLaunchKernelGGL(victim3<NOPS>, dim3(blocks), dim3(threads), 0, s0, worst, iters);
    else if (KIND == 1) hipLaunchKernelGGL(victim2<NOPS>, dim3(blocks), dim3(threads), 0, s0, worst, iters);
    else hipLaunchKernelGGL(victim<NOPS>, dim3(blocks), dim3(threads), 0, s0, worst, iters);
    CK(hipGetLastError()); CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), worst, n * 4, hipMemcpyDeviceToHost));
    for (int e = 0; e < n; ++e) {
      if (h[e] > 1e-5f) ++bad_by_quarter[(e & 63) >> 4];
      maxerr = fmaxf(maxerr, h[e]);
    }
  }
  printf("%-34s mode=%d  threads with |1 - b*rcp(b)| > 1e-5 by lane quarter (0-15, 16-31, 32-47, 48-63): %ld %ld %ld %ld   max %.3e\n",
         name, mode, bad_by_quarter[0], bad_by_quarter[1], bad_by_quarter[2], bad_by_quarter[3], maxerr);
  CK(hipFree(worst)); CK(hipFree(junk));
}

int main() {
  // mode 0: two waves per workgroup, 52 workgroups, alone; 1: + a transcendental-heavy kernel on a second stream; 3: the victim itself
  // with 8 waves per SIMD (its waves contend with each other for the issue ports and the transcendental pipe)
  run<-1>("RAW: rcp; fma (no wait state)", 0);                 // wrong in every lane: the hardware does not interlock this
  run<0>("RAW: rcp; s_nop 0; fma", 0);
  run<0>("RAW: rcp; s_nop 0; fma", 1);
  run<0>("RAW: rcp; s_nop 0; fma", 3);
  run<0, 1>("RAW: rcp; rcp; s_nop 0; fma on 2nd", 1);
  run<0, 1>("RAW: rcp; rcp; s_nop 0; fma on 2nd", 3);
  run<0, 2>("RAW: rcp; rcp; s_nop 0; v_pk_fma", 1);
  run<0, 2>("RAW: rcp; rcp; s_nop 0; v_pk_fma", 3);
  run<-1, 3>("WAR: rcp; overwrite src (no nop)", 1);
  run<-1, 3>("WAR: rcp; overwrite src (no nop)", 3);
  run<0, 4>("pk_mul: SALU write in front", 3);
  run<1, 4>("pk_mul: SALU overwrite behind", 3);
  run<2, 4>("pk_mul: both", 3);
  return 0;
}
