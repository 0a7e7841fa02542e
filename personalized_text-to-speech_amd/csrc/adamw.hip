// AdamW over a flat parameter buffer + the global L2 norm of the gradients, as ONE multi-tensor pass (gfx950).
//
// Replaces: torch.optim.AdamW.step (finetune_speaker_v2.py:113-120,213-214,230-231) and commons.clip_grad_value_(params, None)
//           (commons.py:149-164: with clip_value None it only RETURNS the total L2 norm) — the reference walks ~1900 parameter
//           tensors several times each (and syncs the host once per tensor for the norm).
//
// Layout: parameters, first and second moments live in three flat fp32 buffers with the same offsets (optim.FlatAdamW re-points
// every nn.Parameter at a view of the first).  Gradients stay where autograd / the weight arena left them: an ENTRY names one
// contiguous run of gradient memory and the flat offset it updates — the arena's whole gradient buffer is one entry.  Entries are
// passed BY VALUE in the kernel arguments (no device table to keep alive or to upload: capturable, and eager steps may hand in
// different gradient addresses every time), kMaxEntries per launch.
//
// One workgroup = one 4096-element chunk of one entry (found by binary search over the per-entry first-block prefix).  Pure
// streaming: 16 B read of g, p, m, v and 12 B written per element => HBM-bound, 28 B / element.  Every workgroup also leaves
// the sum of g^2 of its chunk; vits_gradnorm_final adds those in a fixed order (bitwise reproducible, no atomics).
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kChunk = 4096;           // elements per workgroup
constexpr int kMaxEntries = 160;

struct Table {
  const float* g[kMaxEntries];
  unsigned long long off[kMaxEntries];
  unsigned int n[kMaxEntries];
  unsigned int block0[kMaxEntries + 1];  // first workgroup of the entry inside this launch
  int count;
};

struct Hyper {
  float beta1, beta2, eps, weight_decay;
  float one_minus_beta1, one_minus_beta2;   // rounded from double on the host, as torch rounds its Python-side `1 - beta`
};

__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];       // (every thread; fixed order)
}

// state[0] = learning rate, state[1] = number of completed steps (float; incremented by vits_gradnorm_final)
template <bool UPDATE>
__global__ __launch_bounds__(kThreads) void adamw_kernel(Table tab, float* __restrict__ P, float* __restrict__ M, float* __restrict__ V,
                                                         const float* __restrict__ state, Hyper h, float* __restrict__ partials) {
  __shared__ float sh[4];
  int lo = 0, hi = tab.count;                  // entry e with block0[e] <= blockIdx.x < block0[e+1]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tab.block0[mid] <= blockIdx.x) lo = mid; else hi = mid;
  }
  const unsigned int n = tab.n[lo];
  const unsigned int c0 = (blockIdx.x - tab.block0[lo]) * (unsigned)kChunk;
  const unsigned int c1 = (n - c0 < (unsigned)kChunk) ? n : c0 + kChunk;
  const float* __restrict__ g = tab.g[lo];
  const unsigned long long off = tab.off[lo];
  float lr = 0.f, step_size = 0.f, inv_bc2_sqrt = 0.f, decay = 0.f;
  if (UPDATE) {
    lr = state[0];
    const float t = state[1] + 1.0f;
    const float bc1 = 1.0f - powf(h.beta1, t), bc2 = 1.0f - powf(h.beta2, t);
    step_size = lr / bc1;
    inv_bc2_sqrt = 1.0f / sqrtf(bc2);
    decay = 1.0f - lr * h.weight_decay;
  }
  // No contraction into fused multiply-adds here: left to the compiler, the float4 path and the scalar path of this kernel
  // were contracted differently, and the same gradient gave parameters one ulp apart depending on whether its buffer was
  // 16-byte aligned (found when data-parallel replays kept gradients in the arena while the eager step used bucket views).
  auto one = [&](float gi, float& p, float& m, float& v) {
#pragma clang fp contract(off)
    p *= decay;                                         // decoupled weight decay
    m = m + (gi - m) * h.one_minus_beta1;               // exp_avg.lerp_(grad, 1 - beta1)
    v = v * h.beta2 + h.one_minus_beta2 * gi * gi;
    const float denom = sqrtf(v) * inv_bc2_sqrt + h.eps;
    p -= step_size * (m / denom);
  };
  float ss = 0.f;
  const bool vec = ((reinterpret_cast<uintptr_t>(g) | (uintptr_t)(off * 4)) & 15) == 0;
  if (vec) {
    const unsigned int v1 = c0 + ((c1 - c0) & ~3u);
    for (unsigned int i = c0 + threadIdx.x * 4; i < v1; i += kThreads * 4) {
      const float4 gv = *reinterpret_cast<const float4*>(g + i);
      ss += gv.x * gv.x + gv.y * gv.y + gv.z * gv.z + gv.w * gv.w;
      if (UPDATE) {
        float4 p = *reinterpret_cast<const float4*>(P + off + i), m = *reinterpret_cast<const float4*>(M + off + i),
               v = *reinterpret_cast<const float4*>(V + off + i);
        one(gv.x, p.x, m.x, v.x); one(gv.y, p.y, m.y, v.y); one(gv.z, p.z, m.z, v.z); one(gv.w, p.w, m.w, v.w);
        *reinterpret_cast<float4*>(P + off + i) = p;
        *reinterpret_cast<float4*>(M + off + i) = m;
        *reinterpret_cast<float4*>(V + off + i) = v;
      }
    }
    for (unsigned int i = v1 + threadIdx.x; i < c1; i += kThreads) {
      const float gi = g[i];
      ss += gi * gi;
      if (UPDATE) one(gi, P[off + i], M[off + i], V[off + i]);
    }
  } else {
    for (unsigned int i = c0 + threadIdx.x; i < c1; i += kThreads) {
      const float gi = g[i];
      ss += gi * gi;
      if (UPDATE) one(gi, P[off + i], M[off + i], V[off + i]);
    }
  }
  const float tot = block_sum(ss, sh);
  if (partials && threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// One workgroup: sums `n` partials in a fixed order (thread-strided, then a fixed tree), writes sqrt(sum) to norm_out and adds
// 1 to the step counter (state[1]) when asked.
__global__ __launch_bounds__(1024) void gradnorm_final_kernel(const float* __restrict__ partials, int n, float* __restrict__ norm_out,
                                                              float* __restrict__ state, int bump_step) {
  __shared__ double sh[1024];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) s += (double)partials[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (norm_out) *norm_out = (float)sqrt(sh[0]);
    if (bump_step && state) state[1] += 1.0f;
  }
}

}  // namespace

extern "C" size_t vits_adamw_blocks(const vits_adamw_entry* host_entries, int n_entries) {
  size_t blocks = 0;
  for (int i = 0; i < n_entries; ++i) blocks += (host_entries[i].n + (size_t)kChunk - 1) / kChunk;
  return blocks;
}

extern "C" int vits_adamw(float* p, float* m, float* v, const vits_adamw_entry* host_entries, int n_entries, const float* state,
                          double beta1, double beta2, double eps, double weight_decay, float* partials, size_t partials_len,
                          void* stream) {
  if (!host_entries || n_entries <= 0) return VITS_E_BADARG;
  const bool update = p != nullptr;
  if (update && (!m || !v || !state)) return VITS_E_BADARG;
  if (partials && partials_len < vits_adamw_blocks(host_entries, n_entries)) return VITS_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const Hyper h{(float)beta1, (float)beta2, (float)eps, (float)weight_decay, (float)(1.0 - beta1), (float)(1.0 - beta2)};
  size_t done_blocks = 0;
  for (int e0 = 0; e0 < n_entries; e0 += kMaxEntries) {
    Table tab;
    const int cnt = n_entries - e0 < kMaxEntries ? n_entries - e0 : kMaxEntries;
    unsigned int b = 0;
    for (int i = 0; i < cnt; ++i) {
      const vits_adamw_entry& e = host_entries[e0 + i];
      if (!e.g || e.n == 0) return VITS_E_BADARG;
      tab.g[i] = e.g; tab.off[i] = e.offset; tab.n[i] = e.n; tab.block0[i] = b;
      b += (e.n + kChunk - 1) / kChunk;
    }
    tab.block0[cnt] = b;
    tab.count = cnt;
    float* part = partials ? partials + done_blocks : nullptr;
    if (update) hipLaunchKernelGGL(adamw_kernel<true>, dim3(b), dim3(kThreads), 0, s, tab, p, m, v, state, h, part);
    else hipLaunchKernelGGL(adamw_kernel<false>, dim3(b), dim3(kThreads), 0, s, tab, p, m, v, state, h, part);
    const int rc = vits::check_launch("vits_adamw");
    if (rc != VITS_OK) return rc;
    done_blocks += b;
  }
  return VITS_OK;
}

extern "C" int vits_gradnorm_final(const float* partials, size_t n, float* norm_out, float* state, int bump_step, void* stream) {
  if (!partials || n == 0 || n > (size_t)INT32_MAX) return VITS_E_BADARG;
  hipLaunchKernelGGL(gradnorm_final_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), partials, (int)n, norm_out, state,
                     bump_step);
  return vits::check_launch("vits_gradnorm_final");
}
