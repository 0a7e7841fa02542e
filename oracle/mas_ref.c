/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Plain-C restatement of the reference's monotonic alignment search
 * (reference: monotonic_align/core.pyx:5-33 `maximum_path_each`,
 *  core.pyx:36-42 `maximum_path_c`; generated C monotonic_align/core.c:2042-2480).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this.  It is pinned against tests/golden/mas_*.npz, which were produced
 * by the reference's own Cython routine (see tools/gen_golden_mas.py).
 *
 * Semantics restated (cell (y, x); y = spectrogram frame, x = text token):
 *   forward, y in [0,t_y), x in [max(0, t_x+y-t_y), min(t_x, y+1)):
 *       v_cur  = (x == y) ? -1e9 : value[y-1][x]
 *       v_prev = (x == 0) ? ((y == 0) ? 0 : -1e9) : value[y-1][x-1]
 *       value[y][x] += (v_cur > v_prev) ? v_cur : v_prev        (tie/NaN -> v_prev)
 *   backtrack, index = t_x-1, y from t_y-1 down to 0:
 *       path[y][index] = 1
 *       if index != 0 and (index == y or value[y-1][index] < value[y-1][index-1]) index--
 * `values` is scratch and is overwritten, exactly like the reference.
 * Domain: 1 <= t_x <= t_y.  Outside that domain the reference reads out of
 * bounds (core.pyx:32 with wraparound(False)); this restatement returns -1 for
 * such an item and leaves its path untouched instead.
 */
#include <stdint.h>
#include <stddef.h>

#define MAS_NEG (-1e9f)

static int mas_each(int32_t *path, float *value, int t_y, int t_x, int ld)
{
    if (t_x < 1 || t_x > t_y) return -1;
    for (int y = 0; y < t_y; ++y) {
        int x_lo = t_x + y - t_y; if (x_lo < 0) x_lo = 0;
        int x_hi = (y + 1 < t_x) ? y + 1 : t_x;
        for (int x = x_lo; x < x_hi; ++x) {
            float v_cur = (x == y) ? MAS_NEG : value[(size_t)(y - 1) * ld + x];
            float v_prev;
            if (x == 0) v_prev = (y == 0) ? 0.0f : MAS_NEG;
            else        v_prev = value[(size_t)(y - 1) * ld + x - 1];
            value[(size_t)y * ld + x] += (v_cur > v_prev) ? v_cur : v_prev;
        }
    }
    int index = t_x - 1;
    for (int y = t_y - 1; y >= 0; --y) {
        path[(size_t)y * ld + index] = 1;
        if (index != 0 &&
            (index == y || value[(size_t)(y - 1) * ld + index] < value[(size_t)(y - 1) * ld + index - 1]))
            index--;
    }
    return 0;
}

/* paths[b][t_t][t_s] must be pre-zeroed by the caller (monotonic_align/__init__.py:14). */
int oracle_mas_f32(int32_t *paths, float *values, const int32_t *t_ys, const int32_t *t_xs,
                   int b, int t_t, int t_s)
{
    int bad = 0;
    for (int i = 0; i < b; ++i) {
        size_t off = (size_t)i * t_t * t_s;
        if (t_ys[i] > t_t || t_xs[i] > t_s) { bad++; continue; }
        if (mas_each(paths + off, values + off, t_ys[i], t_xs[i], t_s) != 0) bad++;
    }
    return -bad;
}
