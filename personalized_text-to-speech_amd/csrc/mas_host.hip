// vits_mas_f32_cpu — the HOST twin of vits_mas_f32 (same contract, host pointers): for callers of the reference's FFI that
// hold host buffers (monotonic_align/core.pyx:36-42 is a host routine) and for checking the device kernel without a GPU.
// Never called by this package's own path (monotonic_align.py raises on host tensors): it is an entry point, not a fallback.
//
// Same algorithm as the device kernel, not the reference's in-place table: two rolling rows of values and ONE bit per cell
// (the back-pointer test value[y-1][x] < value[y-1][x-1], core.pyx:32) — neg_cent is never written; the path is fully
// overwritten; an item outside 1 <= t_x <= t_y <= t_t, t_x <= t_s gets an all-zero path and status 1.
#include <cstring>
#include <vector>

#include "common.h"

namespace {
constexpr float kNeg = -1e9f;            // max_neg_val of core.pyx:7

template <typename P>
void one_item(const float* nc, P* path, P one, int t_y, int t_x, int t_t, int t_s) {
  std::memset(path, 0, sizeof(P) * (size_t)t_t * t_s);
  const int words = (t_x + 31) >> 5;
  std::vector<uint32_t> dir((size_t)t_y * words, 0u);      // bit x of row y: step left when leaving (y, x)
  std::vector<float> prev(t_x, kNeg), cur(t_x, kNeg);
  for (int y = 0; y < t_y; ++y) {
    const int x_lo = (t_x + y - t_y > 0) ? t_x + y - t_y : 0;
    const int x_hi = (y + 1 < t_x) ? y + 1 : t_x;
    uint32_t* drow = dir.data() + (size_t)y * words;
    for (int x = x_lo; x < x_hi; ++x) {
      const float v_cur = (x == y) ? kNeg : prev[x];                                  // core.pyx:17-20
      const float v_prev = (x == 0) ? (y == 0 ? 0.0f : kNeg) : prev[x - 1];           // core.pyx:21-24
      cur[x] = nc[(size_t)y * t_s + x] + ((v_cur > v_prev) ? v_cur : v_prev);        // core.pyx:25
      if (x > 0 && (x == y || prev[x] < prev[x - 1])) drow[x >> 5] |= 1u << (x & 31); // core.pyx:32, decided while row y-1 is at hand
    }
    // columns outside the band keep -1e9 for the next row's reads (the reference never reads them either)
    for (int x = 0; x < x_lo; ++x) cur[x] = kNeg;
    for (int x = x_hi; x < t_x; ++x) cur[x] = kNeg;
    prev.swap(cur);
  }
  int index = t_x - 1;
  for (int y = t_y - 1; y >= 0; --y) {                                               // core.pyx:29-33
    path[(size_t)y * t_s + index] = one;
    if (index != 0 && ((dir[(size_t)y * words + (index >> 5)] >> (index & 31)) & 1u)) --index;
  }
}
}  // namespace

extern "C" int vits_mas_f32_cpu(const float* neg_cent, void* path, int path_dtype, const int32_t* t_ys, const int32_t* t_xs,
                                int b, int t_t, int t_s, int32_t* status) {
  if (!neg_cent || !path || !t_ys || !t_xs || b <= 0 || t_t <= 0 || t_s <= 0) return VITS_E_BADARG;
  if (path_dtype != VITS_DT_F32 && path_dtype != VITS_DT_I32) return VITS_E_UNSUPPORTED;
  const size_t item = (size_t)t_t * t_s;
  for (int i = 0; i < b; ++i) {
    const int t_y = t_ys[i], t_x = t_xs[i];
    const bool valid = t_x >= 1 && t_x <= t_y && t_y <= t_t && t_x <= t_s;
    if (status) status[i] = valid ? 0 : 1;
    if (path_dtype == VITS_DT_F32) {
      float* p = static_cast<float*>(path) + i * item;
      if (valid) one_item<float>(neg_cent + i * item, p, 1.0f, t_y, t_x, t_t, t_s);
      else std::memset(p, 0, sizeof(float) * item);
    } else {
      int32_t* p = static_cast<int32_t*>(path) + i * item;
      if (valid) one_item<int32_t>(neg_cent + i * item, p, 1, t_y, t_x, t_t, t_s);
      else std::memset(p, 0, sizeof(int32_t) * item);
    }
  }
  return VITS_OK;
}
