/*
 * vitsmi.h — C ABI of libvitsmi.so: the MI355X (gfx950) VITS hot-path kernels.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer unless the name says `host_`;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every call is
 *     asynchronous on that stream and performs no allocation, no host sync, no global state,
 *     so it may be captured into a hipGraph;
 *   - return value: 0 on success, a negative VITS_E_* code on a rejected argument set or
 *     a launch failure (nothing has been written in that case);
 *   - tensors are dense, row-major, in the reference's layouts ([b, channels, time] for
 *     activations, [b, t_t, t_s] for alignment matrices).
 *
 * Each declaration cites the reference interface it replaces (paths relative to the
 * reference repository root).
 */
#ifndef VITSMI_H
#define VITSMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VITS_OK              0
#define VITS_E_BADARG      (-1)   /* null pointer / non-positive size */
#define VITS_E_UNSUPPORTED (-2)   /* shape outside what the kernel was built for */
#define VITS_E_LAUNCH      (-3)   /* hipLaunchKernel reported an error */

#define VITS_DT_F32  0
#define VITS_DT_I32  1
#define VITS_DT_BF16 2

/* ABI version: bumped whenever a signature changes. */
int vits_abi_version(void);
/* Text of the last HIP error seen by this thread (empty string if none). */
const char* vits_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Monotonic alignment search (maximum-path DP + backtrack).
 *
 * Replaces: monotonic_align/core.pyx:36-42  `maximum_path_c(int[:,:,::1] paths,
 *           float[:,:,::1] values, int[::1] t_ys, int[::1] t_xs)`  (called from
 *           monotonic_align/__init__.py:18, itself from models.py:480), including the
 *           host round trip of monotonic_align/__init__.py:12-19.
 *
 *   neg_cent [b, t_t, t_s] float32, read-only (the reference overwrites its `values`
 *            scratch copy in place; this entry point never writes to it);
 *   path     [b, t_t, t_s] fully overwritten with 0/1 in `path_dtype`
 *            (VITS_DT_F32 = what monotonic_align/__init__.py:19 returns for fp32 models,
 *             VITS_DT_I32 = the Cython routine's native int32);
 *   t_ys[b]  number of valid rows (frames)  = mask.sum(1)[:,0];
 *   t_xs[b]  number of valid columns (text) = mask.sum(2)[:,0];
 *   status   optional int32[b] (may be NULL): 0 per item, 1 for an item outside the domain
 *            1 <= t_x <= t_y <= t_t, t_x <= t_s.  Such an item gets an all-zero path (the
 *            reference reads out of bounds there: core.pyx:32 with wraparound(False)).
 *
 * Limits: t_s <= 1024; LDS = 4*(ceil(t_t/32)*w + t_t + 2*16*w) bytes <= 160 KiB, w = roundup(t_s, 16)
 *         (t_t = 1000 fits up to t_s ~ 640; the reference caps t_s at 381).  Else VITS_E_UNSUPPORTED.
 * Result is bit-identical to the reference for NaN-free inputs (one fp32 add per cell).
 * ------------------------------------------------------------------------------------------ */
int vits_mas_f32(const float* neg_cent, void* path, int path_dtype,
                 const int32_t* t_ys, const int32_t* t_xs,
                 int b, int t_t, int t_s, int32_t* status, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VITSMI_H */
