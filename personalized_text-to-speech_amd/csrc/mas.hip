// Monotonic alignment search on gfx950: maximum-path DP + backtrack, one workgroup per item.
//
// Replaces reference monotonic_align/core.pyx:5-42 (maximum_path_each / maximum_path_c) and the
// host round trip of monotonic_align/__init__.py:6-19.
//
// Two launches: a grid-wide zero fill of the [b, t_t, t_s] output (write-only, as large as the input: every CU takes part),
// then the DP kernel.  Work split inside a DP workgroup (512 threads = 8 waves), pipelined over blocks of R = 16 rows (frames)
// with one workgroup barrier per step:
//   loaders     (8 - W waves) the rows of a block are one contiguous span of neg_cent; every loader thread copies a strided set of
//               its dwords HBM -> registers -> LDS ring.  The loads are issued NSET steps before their registers are written to
//               the ring (NSET register sets, unconditional range-checked buffer loads so that the compiler can count them), and
//               the ring holds up to 8 blocks: the barrier never waits on an HBM round trip (it did: 7/8 of the first kernel's time).
//   DP waves    (W = 1, 2 or 4) the DP out of LDS.  Global lane l = 64 w + lane owns the E consecutive text columns
//               x = l*E .. l*E+E-1 and keeps the previous DP row in registers; the left neighbour of a lane's first column
//               arrives by one DPP wave-shift per row, and across a wave boundary through LDS, one block later (see mas_kernel).
//               Only the 1-bit back-pointer (value[y-1][x] < value[y-1][x-1]) of each cell is kept: the 16 rows of a block in a
//               lane register, flushed to an LDS bit matrix dir[y/16][x] (16-bit entries).
//   wave 0      backtrack, 16 rows per step: the lanes fetch the bit words of the 17 columns at and below the current index in ONE
//               parallel LDS read, then the walk over those rows is scalar (v_readlane of the word of the current column):
//               no dependent LDS round trip per row.  All waves then scatter the t_y ones.
//
// Measured inside the kernel (tools/mas_phases.sh, 100 MHz ticks; C2 item of 500 x 201 / C3 item of 800 x 321, round 3): the
// single-DP-wave kernel spent 41 / 74 us in the DP loop and 40 / 64 us in a branchy scalar backtrack; with the columns over 4 DP
// waves, the block's rows preloaded from LDS, one max + add + compare + add-with-carry per cell, exactly counted load waits
// and a branch-free backtrack it is 25 / 44 us + 12 / 19 us (+ 3-4 us until the first block is staged).  The DP loop is bound
// by the DP waves' own dependent instruction stream (loaders switched off: the same time), ~45 ns per row.
//
// Why the result is bit-identical to the reference: every in-band cell is neg_cent[y][x] plus the
// larger of two previously computed cells — one fp32 add per cell, no reassociation.  In-band
// cells only ever read in-band cells (or the two boundary constants), so whatever is computed
// outside the band [max(0,t_x+y-t_y), min(t_x,y+1)) never reaches a value the backtrack reads.
#include "common.h"
#include <type_traits>

namespace {

constexpr float kNeg = -1e9f;            // max_neg_val of core.pyx:7 (exactly representable)
constexpr int kThreads = 512;
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

// W DP waves x 64 lanes x E columns per lane (t_s <= 64*W*E); the other 8 - W waves load.
//
// DP waves are skewed by one block: in step s wave w works on block s - w.  The value a wave's first column needs from the
// column to its left — the last column of wave w - 1, one row up — was produced one step (one barrier) earlier and goes through
// a small LDS array (`seam`, one float per row and wave boundary); inside a wave it is the DPP shift as before.  The serial
// chain per row is then E <= 4 cells for any t_s <= 1024 (it was up to 16), at the price of W - 1 extra steps.
template <int E, int W>
__global__ __launch_bounds__(kThreads) void mas_kernel(const float* __restrict__ neg_cent,
                                                       uint32_t* __restrict__ path,
                                                       const int* __restrict__ t_ys,
                                                       const int* __restrict__ t_xs,
                                                       int T_t, int T_s, int n_slots,
                                                       uint32_t one_bits, int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  constexpr int NL = kThreads - 64 * W;                            // loader threads
  constexpr int LPT = (R * 64 * W * E + NL - 1) / NL;             // dwords of one block per loader thread (upper bound)
  constexpr int NSET = E > 4 ? 1 : (LPT <= 8 ? 6 : (LPT <= 16 ? 4 : (LPT <= 32 ? 3 : 1)));   // register sets of loads in flight (blocks ahead of the ring)
  constexpr bool KEEP_DST = E <= 4 && LPT <= 32;                           // LDS offsets kept in registers, else recomputed per block

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef VITS_MAS_TIMING
  unsigned long long tap[5];
  tap[0] = __builtin_amdgcn_s_memrealtime();
#endif
  const int t_y = t_ys[b];
  const int t_x = t_xs[b];
  const bool valid = (t_x >= 1) && (t_x <= t_y) && (t_y <= T_t) && (t_x <= T_s);
  const size_t item_elems = (size_t)T_t * T_s;
  const size_t item_off = (size_t)b * item_elems;
  if (tid == 0 && status != nullptr) status[b] = valid ? 0 : 1;
  if (!valid) return;                                             // (uniform: the output is already zero)

  const int nblk_max = (T_t + R - 1) / R;
  const int rs = ring_stride(T_s, E);
  const int seam_len = nblk_max * R + 8;
  uint16_t* dir = reinterpret_cast<uint16_t*>(smem);              // [nblk_max][rs] back-pointer bits, 16 rows per entry
  int* idxs = reinterpret_cast<int*>(smem + (((((size_t)nblk_max * rs + 1) >> 1) + 3) & ~(size_t)3));   // [T_t] path column per row
  float* seam = reinterpret_cast<float*>(idxs + ((T_t + 3) & ~3));                // [W - 1][seam_len]: seam[w][y + 4] = value[y][last column of wave w]
  float* ring = seam + (size_t)(W - 1) * seam_len;                                // [n_slots][R][rs] (+ a dummy row each)

  const int n_blk = (t_y + R - 1) / R;                           // pipeline blocks of R rows
  const int D = n_slots - W;                                     // blocks staged in the ring ahead of DP wave 0 (>= 1)
  const int n_steps = ((n_blk + W - 1 + NSET - 1) / NSET) * NSET;   // a multiple of NSET: the loaders' loop body is NSET whole steps

  // ---- loader state: element j of a thread is dword (lt + j * NL) of a block's contiguous span of R * T_s dwords
  const int lt = tid - 64 * W;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(neg_cent + item_off), 0,
                                                                         (int)((size_t)t_y * T_s * 4), 0x00020000);   // rows >= t_y read as 0
  const int span = R * T_s;
  const float inv_ts = 1.0f / (float)T_s;
  // Elements past the span (the last j of some threads; every j of threads beyond a tiny span) are loaded all the same (from
  // the next block's span, or out of range -> 0) and stored into a dummy row of 64 dwords behind the slot's R rows (one dword
  // per lane: no bank conflict), so that a block is LPT unconditional loads and LPT unconditional stores: straight-line code
  // whose waits the compiler counts exactly (with `if (valid) store` its vmcnt waits degraded to "all but the newest set",
  // i.e. one HBM round trip per step).
  const int slot_stride = R * rs + 64;
  auto elem_of = [&](int j) -> int { return lt + j * NL; };
  auto dst_of = [&](int j) -> int {                               // LDS offset inside a slot (row * rs + column)
    const int e = elem_of(j);
    if (e >= span) return R * rs + lane;
    int row = (int)((float)e * inv_ts);                           // e / T_s (e < 2^24: exact after the fix-up)
    if (row * T_s > e) --row;
    if ((row + 1) * T_s <= e) ++row;
    return row * rs + (e - row * T_s);
  };
  int ldst[KEEP_DST ? LPT : 1];
  if (KEEP_DST && wave >= W) {
#pragma unroll
    for (int j = 0; j < LPT; ++j) ldst[j] = dst_of(j);
  }
  float regs[NSET][LPT];
  auto load_block = [&](int set, int k) {                        // unconditional: blocks past the end are out of range -> 0
    const unsigned base = (unsigned)k * (unsigned)span * 4u;
#pragma unroll
    for (int j = 0; j < LPT; ++j)
      regs[set][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(base + (unsigned)elem_of(j) * 4u), 0, 0));
  };
  auto store_block = [&](int set, int k) {
    float* dst = ring + (size_t)(k % n_slots) * slot_stride;
#pragma unroll
    for (int j = 0; j < LPT; ++j) dst[KEEP_DST ? ldst[j] : dst_of(j)] = regs[set][j];
  };
  if (wave >= W) {
    // the first D blocks go straight into the ring, NSET at a time (their loads in flight together), then the sets take D .. D+NSET-1
    for (int k = 0; k < D; k += NSET) {
#pragma unroll
      for (int d = 0; d < NSET; ++d) if (k + d < D) load_block(d, k + d);
#pragma unroll
      for (int d = 0; d < NSET; ++d) if (k + d < D) store_block(d, k + d);
    }
#pragma unroll
    for (int d = 0; d < NSET; ++d) load_block(d, D + d);
  }

  // DP state of the DP waves (kept in registers across blocks)
  const int gl = wave * 64 + lane;               // (DP waves) global lane: owns columns gl*E .. gl*E + E-1
  const int x0 = gl * E;
  const int xr = x0 < rs ? x0 : rs - E;          // lanes past the staged width re-read the last chunk (unused)
  float prev[E];
  uint32_t acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { prev[e] = kNeg; acc[e] = 0u; }
  // Cell (0,0) is neg_cent[0][0] + max(v_prev = 0, v_cur = -1e9) (core.pyx:17-24).  Feeding it
  // v_cur = 0 / v_prev = -1e9 gives the same sum and lets the shifted-in boundary value be the
  // constant -1e9 on every row (the back-pointer bit of column 0 is never read: `index != 0`).
  if (gl == 0) prev[0] = 0.0f;
  if (W > 1 && tid < W - 1) seam[(size_t)tid * seam_len + 3] = kNeg;   // "row -1" of every wave boundary

  __syncthreads();
#ifdef VITS_MAS_TIMING
  tap[1] = __builtin_amdgcn_s_memrealtime();
#endif
  // Two loops with the same number of barriers: the DP waves', and the loaders' (whose body is straight-line code).
  if (wave < W) {
#pragma unroll 1
    for (int s = 0; s < n_steps; ++s) {
      const int k = s - wave;
      if (k >= 0 && k < n_blk) {              // (uniform over the wave)
        const float* rows = ring + (size_t)(k % n_slots) * slot_stride + xr;
        float fill[R];
#pragma unroll
        for (int r = 0; r < R; ++r) fill[r] = kNeg;
        if (W > 1 && wave > 0) {
          const float* sm = seam + (size_t)(wave - 1) * seam_len + k * R + 3;   // sm[r] = value[y - 1][x0 - 1] of row y = k*R + r
#pragma unroll
          for (int r = 0; r < R; ++r) fill[r] = sm[r];
        }
        // the block's input rows: all reads issued up front (E <= 4: R*E registers), so that the row loop below never
        // waits on an LDS round trip
        constexpr bool PRELOAD = E <= 4;
        float in_all[PRELOAD ? R : 1][E];
        if constexpr (PRELOAD) {
#pragma unroll
          for (int r = 0; r < R; ++r) read_chunk<E>(in_all[r], rows + r * rs);
        }
        // One row = per cell a max, an add, a compare and an add-with-carry (the back-pointer bit shifted into `acc`: row r of
        // the block ends up at bit 15 - r), plus one DPP shift; the diagonal fix (2 more) only in the blocks that still touch
        // the diagonal, the seam store (lane 63 only) only in the waves that have a right neighbour.
        float sv[R];                                                  // this wave's last column after each row (lane 63's copy goes to the seam)
        auto rows16 = [&](auto diag_tag) {
          constexpr bool DIAG = decltype(diag_tag)::value;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int y = k * R + r;
            float in[E];
            if constexpr (PRELOAD) {
#pragma unroll
              for (int e = 0; e < E; ++e) in[e] = in_all[r][e];
            } else {
              read_chunk<E>(in, rows + r * rs);
            }
            const float left = wave_shr1(prev[E - 1], fill[r]);      // value[y-1][x0-1]
            float cur[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
              const float v_cur = prev[e];
              const float v_prev = (e == 0) ? left : prev[e - 1];
              float m;                                                                    // core.pyx:25 max(v_prev, v_cur) as ONE v_max_f32
              asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(v_cur), "v"(v_prev));             // (fmaxf / fmed3 come with two canonicalising maxes in front)
              cur[e] = in[e] + m;
              acc[e] = acc[e] + acc[e] + ((v_cur < v_prev) ? 1u : 0u);                    // core.pyx:32 backtrack test
            }
#pragma unroll
            for (int e = 0; e < E; ++e) prev[e] = cur[e];
            // cell (y, x = y+1) is what row y+1 reads as v_cur on its diagonal: the reference
            // substitutes max_neg_val there (core.pyx:17-18).  (r+1)%E is a compile-time index (R is a multiple of E).
            if constexpr (DIAG) prev[(r + 1) % E] = ((unsigned)gl == (unsigned)(y + 1) / E) ? kNeg : prev[(r + 1) % E];
            sv[r] = prev[E - 1];
          }
        };
#ifndef VITS_MAS_NO_DP                        // (timing builds: the loaders alone)
#ifdef VITS_MAS_NO_DIAG                       // (timing builds)
        rows16(std::false_type{});
#else
        if (k * R < 64 * W * E) rows16(std::true_type{}); else rows16(std::false_type{});      // (column y + 1 exists only while y + 1 < 64 W E)
#endif
#endif
        if (W > 1 && wave < W - 1 && lane == 63) {                    // one masked burst per block (a store per row cost ~30 cycles each)
          float* sw = seam + (size_t)wave * seam_len + k * R + 4;       // seam[w][y + 4] = value[y][last column of wave w]; 16-byte aligned
#pragma unroll
          for (int r = 0; r < R; r += 4) *reinterpret_cast<float4*>(sw + r) = make_float4(sv[r], sv[r + 1], sv[r + 2], sv[r + 3]);
        }
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = __builtin_bitreverse32(acc[e]) >> 16;             // row r of the block -> bit r
        if (x0 < rs) {
          uint16_t* drow = dir + (size_t)k * rs + x0;
#pragma unroll
          for (int e = 0; e < E; ++e) drow[e] = (uint16_t)acc[e];
        }
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = 0u;
      }
      __syncthreads();
    }
  } else {
#pragma unroll 1
    for (int s0 = 0; s0 < n_steps; s0 += NSET) {
#pragma unroll
      for (int d = 0; d < NSET; ++d) {
        // block s + D: registers (loaded NSET steps ago) -> the ring slot of block s - W, which the last DP wave left in
        // the previous step; then the same registers take block s + D + NSET
#ifndef VITS_MAS_NO_LOAD                      // (timing builds: the DP waves alone, on whatever the ring holds)
        store_block(d, s0 + d + D);
        load_block(d, s0 + d + D + NSET);
#endif
        __syncthreads();
      }
    }
  }

#ifdef VITS_MAS_TIMING
  tap[2] = __builtin_amdgcn_s_memrealtime();
#endif
  // ---------------- backtrack (core.pyx:29-33), wave 0, 16 rows per step ----------------
  if (wave == 0) {
    int index = t_x - 1;                                            // (uniform)
    for (int j = (t_y - 1) >> 4; j >= 0; --j) {
      // Lane o holds column base - o.  D = the 16 "step left" decisions of that column for the rows of this group (bit r: row
      // 16 j + r): the back-pointer bit, or the diagonal (index == y), never at column 0, never for rows >= t_y.  The walk is
      // then branch-free scalar code: one readlane of D at the current offset, one bit test, one add per row.
      const int base = index, y0 = R * j;
      const int col = base - lane;
      uint32_t D = (col > 0 && lane <= R) ? (uint32_t)dir[(size_t)j * rs + col] : 0u;
      const int rd = col - y0;
      if (col > 0 && rd >= 0 && rd < R) D |= 1u << rd;
      const int nrow = (t_y - y0 < R) ? (t_y - y0) : R;
      D &= (1u << nrow) - 1u;
      int o = 0, vidx = 0;
#pragma unroll
      for (int r = R - 1; r >= 0; --r) {
        vidx = (lane == r) ? base - o : vidx;                         // path column of row y0 + r (rows >= nrow: not stored)
        const uint32_t dw = (uint32_t)__builtin_amdgcn_readlane((int)D, o);
        o += (int)((dw >> r) & 1u);
      }
      if (lane < nrow) idxs[y0 + lane] = vidx;
      index = base - o;
    }
  }
#ifdef VITS_MAS_TIMING
  tap[3] = __builtin_amdgcn_s_memrealtime();
#endif
  __syncthreads();
  uint32_t* out = path + item_off;
  for (int y = tid; y < t_y; y += kThreads) out[(size_t)y * T_s + idxs[y]] = one_bits;
#ifdef VITS_MAS_TIMING
  // debug build: the phase boundaries (100 MHz ticks since the start) of this item into the LAST row of its output (corrupts it)
  __syncthreads();
  tap[4] = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) for (int i = 1; i < 5; ++i) out[(size_t)(T_t - 1) * T_s + i] = (uint32_t)(tap[i] - tap[0]);
#endif
}

template <int E, int W>
size_t lds_dwords(int t_t, int t_s, int n_slots) {
  const size_t rs = ring_stride(t_s, E), nblk = (size_t)(t_t + R - 1) / R;
  return ((((nblk * rs + 1) >> 1) + 3) & ~(size_t)3) + (size_t)((t_t + 3) & ~3) + (size_t)(W - 1) * (nblk * R + 8) + (size_t)n_slots * (R * rs + 64);
}

// -> VITS_E_UNSUPPORTED when the item does not fit the LDS with at least one block staged ahead (the caller then tries fewer DP waves)
template <int E, int W>
int launch(const float* neg_cent, void* path, uint32_t one_bits, const int32_t* t_ys, const int32_t* t_xs,
           int b, int t_t, int t_s, int32_t* status, hipStream_t stream) {
  int n_slots = 8;                                   // LDS ring depth: as many blocks as fit, at least W + 1
  while (n_slots > W + 1 && lds_dwords<E, W>(t_t, t_s, n_slots) * 4 > (size_t)vits::kLdsBytesMax) --n_slots;
  const size_t lds = lds_dwords<E, W>(t_t, t_s, n_slots) * 4;
  if (lds > (size_t)vits::kLdsBytesMax) return VITS_E_UNSUPPORTED;
  auto kern = mas_kernel<E, W>;
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
  // widest split of the columns over DP waves that fits the LDS (long items of wide texts fall back to fewer DP waves)
#define VITS_MAS_TRY(E_, W_)                                                                        \
  {                                                                                                 \
    const int rc = launch<E_, W_>(neg_cent, path, one_bits, t_ys, t_xs, b, t_t, t_s, status, s);      \
    if (rc != VITS_E_UNSUPPORTED) return rc;                                                        \
  }
  if (t_s <= 64) { VITS_MAS_TRY(1, 1) }
  else if (t_s <= 128) { VITS_MAS_TRY(1, 2) VITS_MAS_TRY(2, 1) }
  else if (t_s <= 256) { VITS_MAS_TRY(1, 4) VITS_MAS_TRY(2, 2) VITS_MAS_TRY(4, 1) }
  else if (t_s <= 512) { VITS_MAS_TRY(2, 4) VITS_MAS_TRY(4, 2) VITS_MAS_TRY(8, 1) }
  else if (t_s <= 1024) { VITS_MAS_TRY(4, 4) VITS_MAS_TRY(8, 2) VITS_MAS_TRY(16, 1) }
#undef VITS_MAS_TRY
  return VITS_E_UNSUPPORTED;
}
