// Monotonic alignment search on gfx950: maximum-path DP + backtrack, one workgroup per item.
//
// Replaces reference monotonic_align/core.pyx:5-42 (maximum_path_each / maximum_path_c) and the
// host round trip of monotonic_align/__init__.py:6-19.
//
// Two launches: a grid-wide zero fill of the [b, t_t, t_s] output (write-only, as large as the input: every CU takes part),
// then the DP kernel.  Work split inside a DP workgroup (512 threads = 8 waves), pipelined over blocks of R = 16 rows (frames)
// with one workgroup barrier per block:
//   waves 1..7  loaders: the rows of a block are one contiguous span of neg_cent; every loader thread copies a strided set of
//               its dwords HBM -> registers -> LDS ring.  The loads are issued TWO blocks before their registers are written to
//               the ring (two register sets, unconditional range-checked buffer loads so that the compiler can count them), and
//               the ring holds up to 8 blocks: the barrier never waits on an HBM round trip (it did: 7/8 of the old kernel's time).
//   wave 0      the DP on block k, out of LDS.  Lane l owns the E consecutive text columns x = l*E .. l*E+E-1 and keeps the
//               previous DP row in registers; the left neighbour of a lane's first column arrives by one DPP wave-shift per row.
//               Only the 1-bit back-pointer (value[y-1][x] < value[y-1][x-1]) of each cell is kept: 32 rows per lane register,
//               flushed to an LDS bit matrix dir[y/32][x].
//   wave 0      backtrack, 32 rows per step: the lanes fetch the bit words of the 64 columns below the current index in ONE
//               parallel LDS read, then the walk over those rows is scalar (v_readlane of the word of the current column):
//               no dependent LDS round trip per row.  All waves then scatter the t_y ones.
//
// Why the result is bit-identical to the reference: every in-band cell is neg_cent[y][x] plus the
// larger of two previously computed cells — one fp32 add per cell, no reassociation.  In-band
// cells only ever read in-band cells (or the two boundary constants), so whatever is computed
// outside the band [max(0,t_x+y-t_y), min(t_x,y+1)) never reaches a value the backtrack reads.
#include "common.h"

namespace {

constexpr float kNeg = -1e9f;            // max_neg_val of core.pyx:7 (exactly representable)
constexpr int kThreads = 512;
constexpr int kLoaders = kThreads - 64;
constexpr int R = 16;                    // rows per pipeline block

__device__ __forceinline__ float wave_shr1(float src, float fill) {
  // lane l <- lane l-1; lane 0 <- fill   (DPP wave_shr:1, bound_ctrl off => keeps `old`)
  int r = __builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(src), 0x138, 0xf, 0xf, false);
  return __int_as_float(r);
}

// E consecutive floats from LDS; the address is 4*min(E,4)-byte aligned by construction.
template <int E>
__device__ __forceinline__ void read_chunk(float (&dst)[E], const float* p) {
  if constexpr (E == 1) {
    dst[0] = p[0];
  } else if constexpr (E == 2) {
    const float2 v = *reinterpret_cast<const float2*>(p);
    dst[0] = v.x; dst[1] = v.y;
  } else {
#pragma unroll
    for (int q = 0; q < E / 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(p + 4 * q);
      dst[4 * q + 0] = v.x; dst[4 * q + 1] = v.y; dst[4 * q + 2] = v.z; dst[4 * q + 3] = v.w;
    }
  }
}

__host__ __device__ inline int ring_stride(int T_s, int E) {
  // floats per staged row: 16-byte aligned rows, and room for the last lane's full E-chunk
  int w = ((T_s + E - 1) / E) * E;
  return (w + 3) & ~3;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// zero fill of the whole output (16-byte stores, grid-stride)
__global__ __launch_bounds__(256) void mas_zero_kernel(u32x4* __restrict__ out, size_t n16, uint32_t* __restrict__ tail, int n_tail) {
  const u32x4 z = {0u, 0u, 0u, 0u};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = z;
  if (blockIdx.x == 0 && (int)threadIdx.x < n_tail) tail[threadIdx.x] = 0u;
}

// E columns per lane (t_s <= 64*E).
template <int E>
__global__ __launch_bounds__(kThreads) void mas_kernel(const float* __restrict__ neg_cent,
                                                       uint32_t* __restrict__ path,
                                                       const int* __restrict__ t_ys,
                                                       const int* __restrict__ t_xs,
                                                       int T_t, int T_s, int n_slots,
                                                       uint32_t one_bits, int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  constexpr int LPT = (R * 64 * E + kLoaders - 1) / kLoaders;      // dwords of one block per loader thread (upper bound)
  constexpr int NSET = E <= 4 ? 2 : 1;                            // register sets of loads in flight (wide items: one, to stay in registers)
  constexpr bool KEEP_DST = E <= 4;                               // LDS offsets kept in registers, else recomputed per block

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t_y = t_ys[b];
  const int t_x = t_xs[b];
  const bool valid = (t_x >= 1) && (t_x <= t_y) && (t_y <= T_t) && (t_x <= T_s);
  const size_t item_elems = (size_t)T_t * T_s;
  const size_t item_off = (size_t)b * item_elems;
  if (tid == 0 && status != nullptr) status[b] = valid ? 0 : 1;
  if (!valid) return;                                             // (uniform: the output is already zero)

  const int nblk32 = (T_t + 31) >> 5;
  const int rs = ring_stride(T_s, E);
  uint32_t* dir = smem;                                          // [nblk32][rs] back-pointer bits
  int* idxs = reinterpret_cast<int*>(dir + (size_t)nblk32 * rs);  // [T_t] path column per row
  float* ring = reinterpret_cast<float*>(idxs + ((T_t + 3) & ~3)); // [n_slots][R][rs]

  const int n_blk = (t_y + R - 1) / R;                           // pipeline blocks of R rows
  const int D = n_slots - 1;                                     // blocks staged ahead of the DP

  // ---- loader state: element j of a thread is dword (lt + j * kLoaders) of a block's contiguous span of R * T_s dwords
  const int lt = tid - 64;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(neg_cent + item_off), 0,
                                                                         (int)((size_t)t_y * T_s * 4), 0x00020000);   // rows >= t_y read as 0
  const int span = R * T_s;
  const float inv_ts = 1.0f / (float)T_s;
  auto dst_of = [&](int j) -> int {                               // LDS offset inside a slot (row * rs + column), -1 = no element
    const int e = lt + j * kLoaders;
    if (e >= span) return -1;
    int row = (int)((float)e * inv_ts);                           // e / T_s (e < 2^24: exact after the fix-up)
    if (row * T_s > e) --row;
    if ((row + 1) * T_s <= e) ++row;
    return row * rs + (e - row * T_s);
  };
  int ldst[KEEP_DST ? LPT : 1];
  if (KEEP_DST && wave != 0) {
#pragma unroll
    for (int j = 0; j < LPT; ++j) ldst[j] = dst_of(j);
  }
  float regs[NSET][LPT];
  auto load_block = [&](int set, int k) {                        // unconditional: blocks past the end are out of range -> 0
    const unsigned base = (unsigned)k * (unsigned)span * 4u;
#pragma unroll
    for (int j = 0; j < LPT; ++j)
      regs[set][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(base + (unsigned)(lt + j * kLoaders) * 4u), 0, 0));
  };
  auto store_block = [&](int set, int k) {
    float* dst = ring + (size_t)(k % n_slots) * R * rs;
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
      const int o = KEEP_DST ? ldst[j] : dst_of(j);
      if (o >= 0) dst[o] = regs[set][j];
    }
  };

  if (wave != 0) {
    for (int k = 0; k < D; ++k) { load_block(0, k); store_block(0, k); }
    load_block(0, D);
    if constexpr (NSET == 2) load_block(1, D + 1);
  }

  // DP state of wave 0 (kept in registers across blocks)
  const int x0 = lane * E;
  const int xr = x0 < rs ? x0 : rs - E;          // lanes past the staged width re-read the last chunk (unused)
  float prev[E];
  uint32_t acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { prev[e] = kNeg; acc[e] = 0u; }
  // Cell (0,0) is neg_cent[0][0] + max(v_prev = 0, v_cur = -1e9) (core.pyx:17-24).  Feeding it
  // v_cur = 0 / v_prev = -1e9 gives the same sum and lets the shifted-in boundary value be the
  // constant -1e9 on every row (the back-pointer bit of column 0 is never read: `index != 0`).
  if (lane == 0) prev[0] = 0.0f;

  __syncthreads();
#pragma unroll 1
  for (int k0 = 0; k0 < n_blk; k0 += 2) {
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int k = k0 + d;
      if (k < n_blk) {                            // (uniform over the workgroup)
        if (wave == 0) {
          const float* rows = ring + (size_t)(k % n_slots) * R * rs + xr;
          const unsigned sh0 = (unsigned)(k & 1) * R;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int y = k * R + r;
            float in[E];
            read_chunk<E>(in, rows + r * rs);
            const float left = wave_shr1(prev[E - 1], kNeg);         // value[y-1][x0-1]
            float cur[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
              const float v_cur = prev[e];
              const float v_prev = (e == 0) ? left : prev[e - 1];
              const float m = (v_cur > v_prev) ? v_cur : v_prev;      // core.pyx:25 max(v_prev, v_cur)
              const uint32_t bit = (v_cur < v_prev) ? 1u : 0u;        // core.pyx:32 backtrack test
              cur[e] = in[e] + m;
              acc[e] |= bit << (sh0 + r);
            }
#pragma unroll
            for (int e = 0; e < E; ++e) prev[e] = cur[e];
            // cell (y, x = y+1) is what row y+1 reads as v_cur on its diagonal: the reference
            // substitutes max_neg_val there (core.pyx:17-18).  (r+1)%E is a compile-time index.
            prev[(r + 1) % E] = ((unsigned)lane == (unsigned)(y + 1) / E) ? kNeg : prev[(r + 1) % E];
          }
          if ((k & 1) || k == n_blk - 1) {
            uint32_t* drow = dir + (size_t)(k >> 1) * rs + x0;
            if (x0 < rs) {
#pragma unroll
              for (int e = 0; e < E; ++e) drow[e] = acc[e];
            }
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] = 0u;
          }
        } else {
          // block k + D: registers (loaded NSET iterations ago) -> ring slot of block k - 1, which the DP has left;
          // then the same registers take block k + D + NSET
          store_block(d % NSET, k + D);
          load_block(d % NSET, k + D + NSET);
        }
        __syncthreads();
      }
    }
  }

  // ---------------- backtrack (core.pyx:29-33), wave 0, 32 rows per step ----------------
  if (wave == 0) {
    int index = t_x - 1;                                            // (uniform)
    for (int j = (t_y - 1) >> 5; j >= 0; --j) {
      const int base = index;
      const int col = base - lane;                                  // lane l holds the bit word of column base - l
      const uint32_t w = col >= 0 ? dir[(size_t)j * rs + col] : 0u;
      int vidx = 0;
      const int y_hi = (t_y - 1 < 32 * j + 31) ? (t_y - 1) : (32 * j + 31);
      for (int y = y_hi; y >= 32 * j; --y) {
        const int r = y & 31;
        vidx = (lane == r) ? index : vidx;
        if (index != 0) {
          const uint32_t ws = (uint32_t)__builtin_amdgcn_readlane((int)w, base - index);
          if (index == y || ((ws >> r) & 1u)) index = __builtin_amdgcn_readfirstlane(index - 1);
        }
      }
      if (lane < 32 && 32 * j + lane <= y_hi) idxs[32 * j + lane] = vidx;
    }
  }
  __syncthreads();
  uint32_t* out = path + item_off;
  for (int y = tid; y < t_y; y += kThreads) out[(size_t)y * T_s + idxs[y]] = one_bits;
}

template <int E>
int launch(const float* neg_cent, void* path, uint32_t one_bits, const int32_t* t_ys, const int32_t* t_xs,
           int b, int t_t, int t_s, int32_t* status, hipStream_t stream) {
  const size_t rs = ring_stride(t_s, E);
  const size_t fixed = (size_t)((t_t + 31) >> 5) * rs + (size_t)((t_t + 3) & ~3);
  int n_slots = 8;                                   // LDS ring depth: as many blocks as fit, at least 2
  while (n_slots > 2 && (fixed + (size_t)n_slots * R * rs) * 4 > (size_t)vits::kLdsBytesMax) --n_slots;
  const size_t lds = (fixed + (size_t)n_slots * R * rs) * 4;
  if (lds > (size_t)vits::kLdsBytesMax) return VITS_E_UNSUPPORTED;
  auto kern = mas_kernel<E>;
  hipError_t e = vits::ensure_max_dynamic_lds(reinterpret_cast<const void*>(kern));
  if (e != hipSuccess) return vits::note_hip_error(e, "vits_mas_f32/attr");
  const size_t n = (size_t)b * t_t * t_s;            // dwords of the output
  const size_t n16 = n / 4;
  unsigned zb = (unsigned)((n16 + 255) / 256);
  if (zb > 2048) zb = 2048;
  if (zb < 1) zb = 1;
  hipLaunchKernelGGL(mas_zero_kernel, dim3(zb), dim3(256), 0, stream, static_cast<u32x4*>(path), n16,
                     static_cast<uint32_t*>(path) + n16 * 4, (int)(n - n16 * 4));
  hipLaunchKernelGGL(kern, dim3(b), dim3(kThreads), lds, stream, neg_cent, static_cast<uint32_t*>(path),
                     t_ys, t_xs, t_t, t_s, n_slots, one_bits, status);
  return vits::check_launch("vits_mas_f32");
}

}  // namespace

extern "C" int vits_mas_f32(const float* neg_cent, void* path, int path_dtype, const int32_t* t_ys,
                            const int32_t* t_xs, int b, int t_t, int t_s, int32_t* status, void* stream) {
  if (!neg_cent || !path || !t_ys || !t_xs || b <= 0 || t_t <= 0 || t_s <= 0) return VITS_E_BADARG;
  uint32_t one_bits;
  if (path_dtype == VITS_DT_F32) one_bits = 0x3F800000u;
  else if (path_dtype == VITS_DT_I32) one_bits = 1u;
  else return VITS_E_UNSUPPORTED;
  if ((size_t)t_t * (size_t)t_s * 4 > 0xFFFFFFFFull) return VITS_E_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (t_s <= 64) return launch<1>(neg_cent, path, one_bits, t_ys, t_xs, b, t_t, t_s, status, s);
  if (t_s <= 128) return launch<2>(neg_cent, path, one_bits, t_ys, t_xs, b, t_t, t_s, status, s);
  if (t_s <= 256) return launch<4>(neg_cent, path, one_bits, t_ys, t_xs, b, t_t, t_s, status, s);
  if (t_s <= 512) return launch<8>(neg_cent, path, one_bits, t_ys, t_xs, b, t_t, t_s, status, s);
  if (t_s <= 1024) return launch<16>(neg_cent, path, one_bits, t_ys, t_xs, b, t_t, t_s, status, s);
  return VITS_E_UNSUPPORTED;
}
