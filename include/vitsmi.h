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

/* The host twin of vits_mas_f32: the same contract on HOST pointers, no stream (the reference's FFI,
 * monotonic_align/core.pyx:36-42, is a host routine).  An entry point for callers that hold host buffers and for checking the
 * device kernel without a GPU — this package's own path never calls it (its Python operators raise on host tensors). */
int vits_mas_f32_cpu(const float* neg_cent, void* path, int path_dtype, const int32_t* t_ys, const int32_t* t_xs,
                     int b, int t_t, int t_s, int32_t* status);

/* ------------------------------------------------------------------------------------------
 * Channels-last 1-D convolution, stride 1, on the matrix cores (forward and data gradient).
 *
 * Replaces: every stride-1 torch.nn.Conv1d call of the generator — modules.py:211-223 (ResBlock1),
 *           :240-250 (ResBlock2), :157-172 (WN in_layers / res_skip_layers), attentions.py:286-293
 *           (FFN), models.py:271-273,286 (conv_pre, cond, conv_post), the 1x1 pre/post/proj layers
 *           (models.py:231-238, modules.py:321-328) — together with the element-wise ops the
 *           reference runs around them (F.leaky_relu before the conv, `+ x` residual, `* x_mask`,
 *           `xs / num_kernels`, tanh), which are fused here as prologue / epilogue.
 *
 *   x      [b][t][c_in]      activations, channels last (the reference holds [b, c, t])
 *   w      [k][c_out][c_in]  weights, tap-major (reference Conv1d.weight is [c_out, c_in, k])
 *   y      [b][t_out][c_out], t_out = t + 2*pad - dil*(k-1)
 *   Y[b,t,co] = f( scale * ( sum_{tap,ci} w[tap,co,ci] * lrelu_{in_slope}(x[b, t+tap*dil-pad, ci])
 *                            + bias[co] + bias_b[b,co] + res[b,t,co] ) )
 *   bias   float32[c_out] or NULL;   bias_b float32[b][c_out] or NULL (per-item conditioning);
 *   res    [b][t_out][c_out] or NULL (same dtype as x);
 *   mg_src [b][t_out][c_out] or NULL: multiply the result by lrelu'_{mg_slope}(mg_src) — the chain
 *          rule of a fused input activation when this call computes a data gradient;
 *   lengths int32[b] (valid rows per item) — needed by VITS_CONV_MASK_IN / _MASK_OUT;
 *   The arguments travel in a vits_conv_desc (below); x and y may be channel slices of wider
 *   tensors (row pitches ldx / ldy) and the convolution may be strided in time (discriminators).
 *   in_slope = 1 disables the input activation.  dtype: VITS_DT_BF16 (c_in % 8 == 0, fp32
 *   accumulate) or VITS_DT_F32 (c_in % 4 == 0; exact fp32 fmaf chain on the matrix core).
 * The data gradient of a convolution is the same call on dY with w' [k][c_in][c_out],
 * w'[tap][ci][co] = w[k-1-tap][co][ci], and pad' = dil*(k-1) - pad.
 * ------------------------------------------------------------------------------------------ */
#define VITS_CONV_MASK_IN   1   /* rows t >= lengths[b] of x read as zero  (x * x_mask before the conv) */
#define VITS_CONV_MASK_OUT  2   /* rows t >= lengths[b] of y written as zero ((...) * x_mask after)     */
#define VITS_CONV_TANH      4   /* y = tanh(.)                                                          */
#define VITS_CONV_ACCUM     8   /* y += result                                                          */
#define VITS_CONV_RES_AFTER 16  /* add `res` after scale and the mg_src multiplier instead of before:
                                   y = scale*(conv + bias)*lrelu'(mg_src) + res  (skip path of a data gradient) */
#define VITS_CONV_GATE      32  /* WaveNet gate (commons.py:103-110 fused_add_tanh_sigmoid_multiply): c_out = 2*gate_h,
                                   y[b,t,c] = tanh(v[c]) * sigmoid(v[c + gate_h]) for c < gate_h, v = conv + bias + bias_b;
                                   y has gate_h columns; y2 (optional, 2*gate_h columns) receives v for the backward */
#define VITS_CONV_GATE_BWD  64  /* chain rule of that gate: c_out = gate_h, v = conv(+res)*scale is d(acts); with
                                   (a, b) = mg_src[.., c], mg_src[.., c + gate_h] (the saved v of the forward):
                                   y[.., c] = v * sigmoid(b) * (1 - tanh(a)^2),  y[.., c + gate_h] = v * tanh(a) * sigmoid(b) * (1 - sigmoid(b));
                                   y and mg_src have 2*gate_h columns */

#define VITS_CONV_FLAT      256 /* force the flat-row kernel (rows = (item, time) pairs; chosen automatically for strided,
                                   divided and short-sequence launches) */
#define VITS_CONV_BIG_TILES 512 /* vits_conv1d_cl_wgrad only: take the 128 x 128-tile kernel wherever it can run (by default only
                                   where it is the faster one: stride 1, >= 512 channels on both sides) */
#define VITS_CONV_OUT_LRELU 128 /* y = leaky_relu(., out_slope) applied after residual/scale (discriminator feature maps) */
#define VITS_CONV_RES_SKIP 1024 /* the res_skip layer of a WaveNet layer in one launch (modules.py:166-174): c_out = 2*gate_h; columns
                                   [0, gate_h): y = (conv + bias + res) * mask (y, res: gate_h columns, pitch ldy); columns [gate_h, 2*gate_h):
                                   y2 (+)= (conv + bias) * mask (gate_h columns, pitch ldy2; VITS_CONV_ACCUM applies to y2 only).
                                   Needs lengths (the mask) and y2; tiled kernel only. */

/* All sizes in elements.  Zero in ldx / ldy / ldy2 / stride means "dense" / 1. */
typedef struct vits_conv_desc {
  int32_t dtype;            /* VITS_DT_BF16 | VITS_DT_F32 (x, w, y, y2, res, mg_src share it)           */
  int32_t b, t, c_in, c_out, k, dil, pad;
  int32_t stride;           /* time stride: t_out = (t + 2*pad - dil*(k-1) - 1) / stride + 1             */
  int32_t flags;            /* VITS_CONV_*                                                                */
  int32_t ldx;              /* row pitch of x   (>= c_in: x may be a channel slice of a wider tensor)    */
  int32_t ldy;              /* row pitch of y, res and mg_src                                             */
  int32_t ldy2;             /* row pitch of y2                                                            */
  int32_t gate_h;           /* H of the gate flags                                                        */
  int32_t ldw;              /* row pitch of w (>= c_in; 0 = dense): w rows may be channel slices            */
  int32_t in_div;           /* > 1: data gradient of a stride-`in_div` convolution (stride must be 1): the input time of
                               tap j for output t is (t + j*dil - pad) / in_div when divisible, else the tap is zero      */
  int32_t t_out_override;   /* output length when in_div > 1 (= the forward convolution's input length)                */
  int32_t groups;           /* > 1: grouped convolution given as DENSE block-diagonal operands w [k][c_out][c_in] (what
                               vits_weight_prep layout 3 writes): the kernel only walks the input channels a tile of output
                               channels can see.  Flat-row kernel only.                                                  */
  int64_t w_batch_stride;   /* elements between the operands of consecutive batch items; 0 = one shared w:
                               with k = 1 this makes the call a batched product Y[b] = X[b] . W[b]^T (attention) */
  float in_slope, mg_slope, out_scale, out_slope;
  const void* x;  const void* w;  const float* bias;  const float* bias_b;
  const void* res;  const void* mg_src;  void* y;  void* y2;  const int32_t* lengths;
} vits_conv_desc;

int vits_conv1d_cl(const vits_conv_desc* desc, void* stream);

/* `count` (2..8) independent convolutions in ONE launch — the same layer of the five period discriminators
 * (models.py:299-335: same channels, taps and stride; other row counts, other weights), forward or data gradient: alone each is
 * one partly filled round of workgroups, side by side they fill each other's tails and share one launch boundary.  All or
 * nothing: VITS_E_UNSUPPORTED (nothing launched) unless every descriptor is a launch vits_conv1d_cl would give to the same
 * instance of its deep-prefetch kernel (bf16, c_in >= 128, c_out >= 96, k >= 2, plain epilogues) — the caller then issues
 * `count` vits_conv1d_cl calls.  Results are bitwise those of the separate calls. */
int vits_conv1d_cl_multi(const vits_conv_desc* descs, int count, void* stream);

/* Weight gradient of vits_conv1d_cl (same x, lengths, in_slope, MASK flags as the forward call):
 *   dw[tap][co][ci] (+)= sum_{b,t} dy[b][t][co] * lrelu_{in_slope}(x[b][t + tap*dil - pad][ci])
 * Replaces the weight half of autograd's conv1d backward for the layers listed above.
 *   dy [b][t_out][c_out] (dtype of x);  dw float32 [k][c_out][c_in];  any k;
 *   workspace: device scratch of at least vits_conv1d_cl_wgrad_workspace(...) bytes (per-split fp32
 *   slabs, summed in a fixed order: results are bitwise reproducible);
 *   flags: VITS_CONV_MASK_IN (x rows >= lengths[b] are zero), VITS_CONV_MASK_OUT (dy rows >=
 *   lengths[b] are zero), VITS_CONV_ACCUM (add to dw instead of overwriting). */
typedef struct vits_wgrad_desc {
  int32_t dtype;            /* dtype of x and dy                                                          */
  int32_t b, t, c_in, c_out, k, dil, pad, stride, flags;
  int32_t ldx, lddy;        /* row pitches of x and dy (0 = dense)                                        */
  float in_slope;
  int32_t groups;           /* > 1: grouped convolution (see vits_conv_desc.groups): only the block-diagonal part is computed and
                               dw is the COMPACT float32 [k][c_out][c_in/groups]                                        */
  const void* x;  const void* dy;  float* dw;  void* workspace;  size_t workspace_bytes;
  const int32_t* lengths;
  float* dbias;             /* optional float32[c_out]: (+)= sum_{b,t} dy[b][t][co] (rows masked like dy), same launch */
  int32_t* counters;        /* optional: >= (c_out/64)*(c_in/64)*taps ints, ZERO before the first use and owned by one stream: the last
                               split to finish a tile then sums the slabs itself (same fixed order, same bits) and re-arms its
                               counter, so the call is ONE launch; without it a second launch sums the slabs              */
  size_t counters_len;
} vits_wgrad_desc;

size_t vits_conv1d_cl_wgrad_workspace(int b, int t_out, int c_in, int c_out, int k);
int vits_conv1d_cl_wgrad(const vits_wgrad_desc* desc, void* stream);

/* The same call with its second stage (the fixed-order sum of the per-split slabs) DEFERRED: `pending` (host memory) receives
 * what vits_wgrad_reduce_pending needs; the workspace of a deferred call must stay untouched until that runs.  A fused layer
 * node issues all its weight gradients deferred and sums them in one launch at the end of its backward (bitwise the same
 * result as the immediate form).  pending->splits == 0: the call wrote dw itself, nothing is pending. */
typedef struct vits_wgrad_pending {
  const float* partial;  float* dw;  float* dbias;
  size_t n, nb, slab;       /* elements of dw, of dbias (0 if none), floats per split slab                                  */
  int32_t splits, accumulate;
} vits_wgrad_pending;
int vits_conv1d_cl_wgrad_deferred(const vits_wgrad_desc* desc, void* stream, vits_wgrad_pending* pending);
int vits_wgrad_reduce_pending(const vits_wgrad_pending* list, int count, void* stream);

/* The weight (+ bias) gradients of a GROUP of stride-1 "same" convolutions (t_out == t, groups == 1; in_slope as in
 * vits_conv1d_cl_wgrad) in one launch per taps-per-group class (csrc/conv1d_wgrad_batch.hip): the layers of a stack are each
 * other's parallelism, so with enough 64 x 64 tiles in the group every workgroup walks its whole (b, t) reduction and nothing but
 * dw is written (no per-split slabs).  descs[i] are ordinary vits_wgrad_desc (same dtype).  Entries with a long reduction over
 * few tiles are still split: vits_conv1d_cl_wgrad_batch_plan(descs, count, splits) tells how many slabs entry i will use —
 * give every entry with splits[i] > 1 a `workspace` of at least splits[i] * (k*c_out*c_in + c_out) floats and pass a `pending`
 * array of `count` entries (pending[i].splits == 0 afterwards: entry i is final), then run vits_wgrad_reduce_pending on it;
 * an entry without workspace / pending runs unsplit.
 * Both return VITS_E_UNSUPPORTED if any entry is not eligible (strided, grouped, dilated beyond the staged halo ...): the caller
 * then issues the per-layer calls (the plan call is also the host-side eligibility test: nothing is launched by it). */
int vits_conv1d_cl_wgrad_batch_plan(const vits_wgrad_desc* descs, int count, int* splits_out);
int vits_conv1d_cl_wgrad_batch(const vits_wgrad_desc* descs, int count, void* stream, vits_wgrad_pending* pending);

/* ------------------------------------------------------------------------------------------
 * One WaveNet layer per launch (csrc/wn_layer.hip).
 *
 * Replaces: one iteration of modules.WN.forward's loop (modules.py:157-176) incl. the TorchScript gate
 *           commons.fused_add_tanh_sigmoid_multiply (commons.py:103-110) and, for the backward call, autograd's chain through it:
 *     x_in = in_layers[i](x);  acts = tanh((x_in + g_l)[:, :H]) * sigmoid((x_in + g_l)[:, H:]);
 *     res_skip = res_skip_layers[i](acts);  x = (x + res_skip[:, :H]) * x_mask;  output += res_skip[:, H:]
 *   (last layer: res_skip has H rows, all of them skip).  Channels-last [b][t][c]; rows t >= lengths[b] of x must be zero.
 *   forward : x [b][t][H];  w_in [k][2H][H], w_rs [1][2H | H][H] (dtype of x; the weight arena's forward operands, PACKED), b_in / b_rs
 *             float32, cond float32 [b][2H] (this layer's slice of cond_layer(g)) or NULL;
 *             pre [b][t][2H] (pre-activations incl. bias and cond; optional) and acts [b][t][H] (optional) are the backward's
 *             saved tensors;  h_out [b][t][H] = (x + res) * mask (NULL for the last layer);  skip [b][t][H] (+)= skip * mask.
 *   backward: dcat [b][t][2H] = [d_h | d_o] (d_o = the masked gradient of the stack's output, the same for every layer; the last
 *             layer passes d_o alone, lddcat = its row pitch);  w_rs_t [1][H][2H | H] and w_in_t [k][H][2H] (tap-reversed) are the
 *             arena's data-gradient operands, PACKED;  d_pre [b][t][2H] = gate'(pre) * (dcat . W_rs), masked — written for the weight
 *             gradients;  d_h_out [b][t][H] = d_h + conv^T(d_pre; W_in), masked (last layer: no d_h term).
 *   dtype VITS_DT_BF16 or VITS_DT_F32 (exact fp32 products); H % 16 == 0, H <= 192, k odd.
 *   Returns VITS_E_UNSUPPORTED for other shapes (the caller then composes the layer from vits_conv1d_cl launches).
 * ------------------------------------------------------------------------------------------ */
/* Operand packing for the two calls below: w_in / w_rs (forward) and w_rs_t / w_in_t (backward) are NOT the arena's row-major
 * operands but their re-ordering by vits_wn_pack into MFMA-fragment order ([column tile][reduction step][lane][16 bytes]), so that
 * every weight load of a wave reads 1 KiB of consecutive memory.  One vits_wn_pack launch packs all operands of a stack:
 *   mode 0: w_in  [taps][2H][H]      -> 32-column tiles, gate-interleaved (16 tanh rows | their 16 sigmoid rows), 32-byte steps
 *   mode 1: w_rs  [1][2H | H][H]     -> 32-column tiles in natural order, 32-byte steps
 *   mode 2: w_rs_t [1][H][2H | H], w_in_t [taps][H][2H] (the data-gradient operands) -> 16-row tiles, 64-byte steps (tails zero)
 * src: taps x rows x rowbytes bytes, row-major; dst: vits_wn_pack_bytes(...) bytes; spt = steps per tap (rowbytes / 32 for modes
 * 0 and 1, ceil(rowbytes / 64) for mode 2). */
typedef struct vits_wn_pack_seg {
  const void* src;  void* dst;
  int32_t mode, h, rows, rowbytes, taps, spt;
} vits_wn_pack_seg;
size_t vits_wn_pack_bytes(int mode, int dtype, int h, int rows, int k_elems, int taps);
int vits_wn_pack(const vits_wn_pack_seg* segs, int count, void* stream);

typedef struct vits_wn_layer_desc {
  int32_t dtype, b, t, h, k, dil;
  int32_t last;             /* the stack's last layer: res_skip has H rows (skip only)                         */
  int32_t accumulate;       /* skip += (0: skip =)                                                             */
  int32_t ldx, ldh, ldskip, ldacts, ldpre;      /* row pitches in elements; 0 = dense                         */
  const void* x;  const void* w_in;  const float* b_in;  const float* cond;
  const void* w_rs;  const float* b_rs;
  void* pre;  void* acts;  void* h_out;  void* skip;
  const int32_t* lengths;
} vits_wn_layer_desc;
int vits_wn_layer_fwd(const vits_wn_layer_desc* desc, void* stream);

typedef struct vits_wn_layer_bwd_desc {
  int32_t dtype, b, t, h, k, dil;
  int32_t last;             /* the stack's last layer: no d_h input, W_rs^T has H columns                       */
  int32_t ld_dh, ld_do, ldpre, lddpre, ldout;   /* row pitches in elements; 0 = dense                          */
  const void* d_h;  const void* d_o;  const void* pre;
  const void* w_rs_t;  const void* w_in_t;
  void* d_pre;  void* d_h_out;
  const int32_t* lengths;
} vits_wn_layer_bwd_desc;
int vits_wn_layer_bwd(const vits_wn_layer_bwd_desc* desc, void* stream);

/* ------------------------------------------------------------------------------------------
 * Channels-last ConvTranspose1d = 1x1 matrix-core product + overlap-add.
 *
 * Replaces: models.py:277 `self.ups[i](x)` (weight-normed torch.nn.ConvTranspose1d(c_in, c_out, k,
 *           u, padding=(k-u)//2), models.py:254-258) and its backward.
 *   forward : p = vits_conv1d_cl(x, w1 [1][k*c_out][c_in])  with  w1[0][j*c_out+co][ci] = W[ci][co][j],
 *             then vits_convt_fold_cl:  y[b][to][co] = bias[co] + sum_{j: (to+pad-j)%u==0} p[b][(to+pad-j)/u][j*c_out+co]
 *             with t_out = (t_in-1)*u - 2*pad + k;
 *   backward: vits_convt_unfold_cl: dp[b][ti][j*c_out+co] = dy[b][ti*u + j - pad][co] (0 outside),
 *             then the 1x1 data / weight gradients through vits_conv1d_cl / vits_conv1d_cl_wgrad.
 *   p [b][t_in][k*c_out], y/dy [b][t_out][c_out]; bias float32[c_out] or NULL;
 *   dtype VITS_DT_BF16 (c_out % 8 == 0) or VITS_DT_F32 (c_out % 4 == 0).
 * ------------------------------------------------------------------------------------------ */
int vits_convt_fold_cl(int dtype, const void* p, const float* bias, void* y, int b, int t_in, int c_out, int k,
                       int u, int pad, void* stream);
int vits_convt_unfold_cl(int dtype, const void* dy, void* dp, int b, int t_in, int c_out, int k, int u, int pad,
                         void* stream);

/* ------------------------------------------------------------------------------------------
 * Multi-tensor weight preparation (weight-norm + kernel layouts + compute dtype) and its backward.
 *
 * Replaces: the per-forward weight_norm recomputation of every wrapped convolution
 *           (torch.nn.utils.weight_norm hooks: models.py:254,304-312,339-348; modules.py:128,135,145,
 *           191-206,236-239) and autograd's backward through it — one launch per GROUP of layers
 *           instead of ~8 small kernels per layer.
 *   entries  device array, one per (slice of a) convolution weight, sorted by row0:
 *     v        fp32 master weight in torch layout: Conv1d [c_out_total][c_in][k] (layout 0) or
 *              ConvTranspose1d [c_in_total][c_out][k] (layout 1); layout 2 = Conv1d whose operand keeps the torch
 *              layout (weight-norm + dtype only; consumed by a library convolution); layout 3 = grouped Conv1d
 *              [c_out][c_in/groups][k] emitted as DENSE block-diagonal operands (only the diagonal blocks are written: the
 *              arenas are allocated zeroed), its dw is the compact [k][c_out][c_in/groups];  g  weight_g [rows_total] or NULL;
 *     row_lo   first weight-norm row of the parameter covered by this entry, n_rows rows are covered
 *              (layout 0: rows are output channels, c_out = n_rows; layout 1: rows are input channels);
 *     c_out_p / c_in_p   padded channel counts of the emitted operands (zero outside; pad areas are
 *              never written, allocate the arenas zeroed);
 *     off      element offset of this entry in w_fwd / w_bwd / dw;  off_dv / off_dg  element offsets of
 *              the parameter's gradients in dparam;  row0  global index of the entry's first row.
 *   w_fwd   [k][c_out_p][c_in_p] (layout 1: [1][k*c_out][c_in_p]),  w_bwd [k][c_in_p][c_out_p] taps
 *           reversed (layout 1: [1][c_in][k*c_out]), both in `dtype`;  dw fp32 in the w_fwd layout.
 * ------------------------------------------------------------------------------------------ */
typedef struct vits_prep_entry {
  const float* v;  const float* g;
  int64_t off, off_dv, off_dg;
  int32_t layout, c_out, c_in, k, c_out_p, c_in_p, row_lo, n_rows, row0, groups;
} vits_prep_entry;

/* Layout 4 = layout 0 whose tap-reversed transposed operand w_bwd is NOT written by vits_weight_prep (a strided 2-byte scatter)
 * but by vits_weight_prep_transpose afterwards: an LDS transpose of the 64 x 64 tiles of w_fwd listed in `tiles`
 * (entry index, tap, first output channel, first input channel), coalesced on both sides. */
typedef struct vits_prep_tile { int32_t entry, tap, co0, ci0; } vits_prep_tile;
int vits_weight_prep_transpose(const vits_prep_tile* tiles, int n_tiles, const vits_prep_entry* entries, int dtype,
                               const void* w_fwd, void* w_bwd, void* stream);

int vits_weight_prep(const vits_prep_entry* entries, int n_entries, int total_rows, int dtype, void* w_fwd,
                     void* w_bwd, void* stream);
int vits_weight_prep_bwd(const vits_prep_entry* entries, int n_entries, int total_rows, const float* dw,
                         float* dparam, void* stream);

/* ------------------------------------------------------------------------------------------
 * Row-wise channels-last kernels ([rows][c], c <= 1024; dtype VITS_DT_F32 or VITS_DT_BF16 for the
 * activations, parameters always float32).
 *
 * vits_ln_act_cl:  y = [res +] act( LayerNorm_c(x) * gamma + beta ),  act 0 = identity, 1 = GELU (erf).
 *   Replaces modules.LayerNorm.forward (modules.py:29-32: transpose, F.layer_norm over channels,
 *   transpose) fused with the F.gelu and `x = x + y` around it in DDSConv (modules.py:100-107) and with
 *   the post-norm residual of attentions.Encoder (attentions.py:41-46: res + LayerNorm is written
 *   there as LayerNorm(x + y); use res = NULL and pass x + y).
 * vits_ln_act_cl_bwd: dx, and dgamma / dbeta (+)= per-channel sums (reproducible two-stage sums).
 *   The gradient of `res` is dy itself.
 * vits_dwconv_cl:  y[b,t,c] = bias[c] + sum_j w[c][j] * xm[b, t + (j - (k-1)/2) * dil, c],  xm = x with rows
 *   t >= lengths[b] read as zero (lengths may be NULL), "same" padding, odd k <= 7.
 *   Replaces the groups=channels Conv1d of DDSConv (modules.py:84-86,98: convs_sep[i](x * x_mask)).
 * vits_dwconv_cl_bwd: dx (masked like x), dw [c][k] and dbias [c] (+)=.
 * workspace: at least vits_rowops_workspace(rows, c, k) bytes (k = 1 for the LayerNorm backward).
 * ------------------------------------------------------------------------------------------ */
size_t vits_rowops_workspace(int rows, int c, int k);
int vits_ln_act_cl(int dtype, const void* x, const float* gamma, const float* beta, const void* res, void* y,
                   int rows, int c, float eps, int act, void* stream);
int vits_ln_act_cl_bwd(int dtype, const void* x, const float* gamma, const float* beta, const void* dy, void* dx,
                       float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, int rows, int c,
                       float eps, int act, int accumulate, void* stream);
int vits_dwconv_cl(int dtype, const void* x, const float* w, const float* bias, const int32_t* lengths, void* y,
                   int b, int t, int c, int k, int dil, void* stream);
int vits_dwconv_cl_bwd(int dtype, const void* x, const float* w, const int32_t* lengths, const void* dy, void* dx,
                       float* dw, float* dbias, void* workspace, size_t workspace_bytes, int b, int t, int c, int k,
                       int dil, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * Piecewise rational-quadratic spline, linear tails, 10 bins (forward, inverse, backward).
 *
 * Replaces: transforms.piecewise_rational_quadratic_transform(inputs, unnormalized_widths,
 *           unnormalized_heights, unnormalized_derivatives, inverse, tails='linear', tail_bound)
 *           (transforms.py:12-43, :55-95, :97-193) as called from modules.py:375-384, and its autograd.
 *   x [n] float32;  h [n][ldh] (float32 or bf16): columns 0..9 widths, 10..19 heights, 20..28 interior
 *   derivatives, exactly the ConvFlow projection's channel order (modules.py:371-377); widths and
 *   heights are multiplied by hscale (= 1/sqrt(filter_channels), modules.py:373-374) inside;
 *   y, logabsdet [n] float32.  Outside [-tail_bound, tail_bound] (inclusive): y = x, logabsdet = 0.
 *   Unlike the reference, a call whose inputs ALL lie outside the interval does not raise.
 *   backward: gx [n] and gh [n][ldh] (gradient wrt the raw h, columns >= 29 zeroed) from gy, glogabsdet.
 * ------------------------------------------------------------------------------------------ */
int vits_rq_spline(int h_dtype, const float* x, const void* h, int ldh, float hscale, int inverse, float tail_bound,
                   float* y, float* logabsdet, int n, void* stream);
int vits_rq_spline_bwd(int h_dtype, const float* x, const void* h, int ldh, float hscale, int inverse, float tail_bound,
                       const float* gy, const float* glogabsdet, float* gx, void* gh, int n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Windowed relative-position attention: the row kernels between the matrix products.
 *
 * Replaces: attentions.MultiHeadAttention.attention (attentions.py:150-182) — the scale, the relative-key
 *           logits' skew (:214-229), masked_fill(-1e4), softmax, attention dropout, and the abs->rel skew of
 *           the probabilities (:231-243) — and their autograd.  The products QK^T, Q E_k^T, P V, P_band E_v and
 *           all their gradients are vits_conv1d_cl calls with k = 1 and per-item operands (w_batch_stride).
 *   s  [b][t][ld]  raw scores Q K^T of ONE head (ld >= t, pad columns ignored / written as 0);
 *   r  [b][t][16]  raw relative-key logits Q E_k^T (columns 0..2w used) or NULL;
 *   keep [b][t][ld] dropout keep-scale (0 or 1/(1-p)) or NULL;  lengths int32[b] or NULL;
 *   p, pd [b][t][ld]  softmax and dropped softmax (pd may be NULL);  pband [b][t][16] = pd[i][i+m-w] or NULL.
 *   backward: ds = d(raw scores), dsband [b][t][16] = d(raw relative-key logits), from dpd, dpband.
 *   t <= 1024, 2*window+1 <= 16.
 * ------------------------------------------------------------------------------------------ */
int vits_relsoftmax(int dtype, const void* s, const void* r, const void* keep, const int32_t* lengths, void* p, void* pd,
                    void* pband, int b, int t, int ld, int window, float scale, void* stream);
int vits_relsoftmax_bwd(int dtype, const void* p, const void* dpd, const void* dpband, const void* keep,
                        const int32_t* lengths, void* ds, void* dsband, int b, int t, int ld, int window, float scale,
                        void* stream);

/* ------------------------------------------------------------------------------------------
 * Deterministic two-stage reductions for the loss terms (csrc/reduce.hip).  No zero-initialised memory, no atomics:
 * stage 1 writes one partial per workgroup into `workspace` (>= vits_reduce_workspace(n_seg) bytes), stage 2 sums them in
 * a fixed order.
 *   vits_absdiff_sum   out[0] (+)= scale * sum_i |a[i] - b[i]|   — one term of feature_loss (reference losses.py:7-15:
 *                      mean |rl - gl| over a feature map; a = real half, b = generated half of one channels-last tensor)
 *   vits_absdiff_bwd   db[i] = -sign(a[i]-b[i]) * scale * g[0];  da (optional) = 0  (the reference detaches the real half)
 *   vits_segsum_f32    out[s] = sum_i x[s*seg_len + i]            — torch.sum(x, [1,2]) of the duration predictor
 *                      (models.py:71-102) and whole-tensor sums (n_seg = 1: kl_loss, losses.py:46-61)
 * ------------------------------------------------------------------------------------------ */
size_t vits_reduce_workspace(int n_seg);
int vits_absdiff_sum(int dtype, const void* a, const void* b, size_t n, float scale, float* out, int accumulate,
                     void* workspace, size_t workspace_bytes, void* stream);
int vits_absdiff_bwd(int dtype, const void* a, const void* b, size_t n, const float* g, float scale, void* da, void* db,
                     void* stream);
int vits_segsum_f32(const float* x, int n_seg, size_t seg_len, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* out[seg][c] = sum_r x[seg][r][c]  (x dense [n_seg][rows][c] of `dtype`, out float32): per-item / whole-batch column sums of a
 * channels-last tensor — conditioning and bias gradients (torch `dy.sum(1)` / `dy.sum((0,1))`).  Fixed-order, no atomics;
 * workspace of vits_colsum_workspace(...) bytes (may be 0 when that returns 0... pass the shared scratch). */
size_t vits_colsum_workspace(int n_seg, int rows, int c);
int vits_colsum(int dtype, const void* x, int n_seg, int rows, int c, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* out = dy * (y > 0 ? 1 : slope), rows t >= lengths[b] zeroed; y and lengths optional; dy, y, out dense [b][t][c] of `dtype`.
 * The chain rule of vits_conv1d_cl's fused output leaky-relu (VITS_CONV_OUT_LRELU: sign(y) = sign of the pre-activation)
 * and output mask ahead of the weight / data gradient launches (reference F.leaky_relu, models.py:327,352). */
int vits_lrelu_mask_bwd(int dtype, const void* dy, const void* y, float slope, const int32_t* lengths, int b, int t, int c,
                        void* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * STFT magnitude -> mel filter bank -> log-clamp in one kernel per direction (csrc/stft_mel.hip), behind the DFT product of
 * vits_conv1d_cl.  Replaces: mel_processing.py:63-69 `sqrt(re^2 + im^2 + 1e-6)`, :73-82 / :105-111 `log(clamp(mel_basis @ spec, 1e-5))`
 * and their autograd.
 *   ri [rows = b * frames][ld] fp32: real parts in columns [0, F), imaginary parts in [Fp, Fp + F); basis [M][F] fp32;
 *   mel, lin [b][M][frames] fp32 (lin = basis @ magnitude, optional in fwd, required by bwd); clip = 1e-5 in the reference.
 *   bwd: d ri [rows][ld] (columns outside the two ranges are written as zero).  F <= 2048, M <= 256.
 * ------------------------------------------------------------------------------------------ */
int vits_stft_mel_fwd(const float* ri, const float* basis, float* mel, float* lin, int rows, int frames, int F, int Fp, int ld, int M,
                      float clip, void* stream);
int vits_stft_mel_bwd(const float* ri, const float* basis, const float* lin, const float* dmel, float* dri, int rows, int frames, int F,
                      int Fp, int ld, int M, float clip, void* stream);

/* ------------------------------------------------------------------------------------------
 * Grouped strided convolutions of DiscriminatorS (reference models.py:343-349: 4 input channels per group, 16 or 4 output
 * channels per group, k = 41, stride 4) as direct kernels — csrc/grouped.hip.  bf16 only (VITS_E_UNSUPPORTED otherwise, and
 * for other group shapes: the caller then uses vits_conv1d_cl with `groups`).
 *   x [n][t_in][c_in], y [n][t_out][c_out] channels-last, t_out = (t_in + 2 pad - k) / stride + 1;
 *   w = the DENSE block-diagonal operand [k][c_out][c_in] vits_conv1d_cl takes for the same layer (only the diagonal blocks are
 *   read) — for both directions;
 *   fwd:   y = leaky_relu(conv(x) + bias, out_slope);
 *   dgrad: dx = (conv^T(dy) + res) * (mg_src > 0 ? 1 : mg_slope)   (res, mg_src optional, [n][t_in][c_in]).
 *   wgrad (16 output channels per group only): the compact dw [k][c_out][c_in / groups] (+)= and dbias [c_out] (+)= as
 *          vits_conv1d_cl_wgrad_deferred writes them for `groups` > 1: per-split fp32 slabs in `workspace`
 *          (>= vits_grouped_conv_wgrad_workspace bytes), summed in split order by vits_wgrad_reduce_pending with the entry
 *          returned in `pending` (required).
 * ------------------------------------------------------------------------------------------ */
int vits_grouped_conv_fwd(int dtype, const void* x, const void* w, const float* bias, void* y, int n, int t_in, int c_in, int c_out,
                          int k, int stride, int pad, int groups, float out_slope, void* stream);
int vits_grouped_conv_dgrad(int dtype, const void* dy, const void* w, const void* res, const void* mg_src, void* dx, int n, int t_in,
                            int c_in, int c_out, int k, int stride, int pad, int groups, float mg_slope, void* stream);
size_t vits_grouped_conv_wgrad_workspace(int n, int t_out, int c_out, int k, int groups);
int vits_grouped_conv_wgrad(int dtype, const void* x, const void* dy, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                            int n, int t_in, int c_in, int c_out, int k, int stride, int pad, int groups, int accumulate,
                            vits_wgrad_pending* pending, void* stream);

/* ------------------------------------------------------------------------------------------
 * Discriminator edge layers as bandwidth kernels (csrc/disc_edge.hip).
 *
 * Replaces: the first convolution of every discriminator together with the pad / view it sits behind —
 *           models.py:318-325 (DiscriminatorP: F.pad(reflect) to a multiple of the period, view [b,1,T/p,p],
 *           weight-normed Conv2d(1, 32, (5,1), (3,1), padding (2,0)), leaky_relu 0.1) and models.py:353-356
 *           (DiscriminatorS: Conv1d(1, 16, 15, 1, padding 7), leaky_relu) — and conv_post (models.py:312,330 /
 *           :349,358: 1024 -> 1 channels, k 3), and the autograd of both.
 *   x      float32 [n][T]  raw waveforms (real items first, generated items second);
 *   period p (1 = DiscriminatorS): item j = n_idx*p + w of the folded tensor is column w of the [T/p, p] view;
 *   h1     [(n,w)][R1][c_out] channels-last, R1 = vits_disc_first_rows(T, p, k, s1, pad), leaky_relu(slope) applied;
 *   w      the arena operand [k][c_out][8] (input channel 0 live), dw the arena's fp32 gradient of the same shape
 *          (only [tap][co][0] is written); bias / dbias float32[c_out];
 *   dy     gradient wrt the PRE-activation of h1 (the caller's data-gradient launch applies lrelu'), same layout as h1;
 *   vits_disc_first_dgrad: dx float32 [n - n_lo][T] (+)= gradient wrt the waveforms of items n_lo..n-1 (the generated half
 *          in the generator step), reflect pad folded back.
 *   conv_post: h [(n,w)][R][c_in], w arena operand [k][8][c_in] (output channel 0 live), y8 / dy8 [(n,w)][R][8]
 *          (channel 0 live; y8 channels 1..7 are written as 0);
 *   vits_disc_post_dgrad: dh = (conv^T(dy8) + res) * lrelu'_{slope}(h) for the rows of items j_lo..J-1 (res optional:
 *          the feature-matching gradient of h); vits_disc_post_wgrad: dw [tap][0][:] and dbias[0] (+)=.
 *   workspaces: per-block partial sums, reduced in a fixed order by a second launch (bitwise reproducible).
 * ------------------------------------------------------------------------------------------ */
int vits_disc_first_rows(int T, int p, int k, int s1, int pad);
int vits_disc_first_fwd(int dtype, const float* x, const void* w, const float* bias, void* y, int n, int T, int p, int k, int s1,
                        int pad, int c_out, float slope, void* stream);
size_t vits_disc_first_wgrad_workspace(int n, int T, int p, int k, int s1, int pad, int c_out);
int vits_disc_first_wgrad(int dtype, const float* x, const void* dy, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                          int n, int T, int p, int k, int s1, int pad, int c_out, int accumulate, void* stream);
int vits_disc_first_dgrad(int dtype, const void* dy, const void* w, float* dx, int n, int n_lo, int T, int p, int k, int s1, int pad,
                          int c_out, int accumulate, void* stream);
int vits_disc_post_fwd(int dtype, const void* h, const void* w, const float* bias, void* y8, int J, int R, int c_in, int k, int pad,
                       void* stream);
int vits_disc_post_dgrad(int dtype, const void* dy8, const void* w, const void* res, const void* h, void* dh, int J, int R, int c_in,
                         int k, int pad, int j_lo, float slope, void* stream);
size_t vits_disc_post_wgrad_workspace(int J, int R, int c_in, int k);
int vits_disc_post_wgrad(int dtype, const void* dy8, const void* h, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                         int J, int R, int c_in, int k, int pad, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * AdamW over flat parameter / moment buffers + the global L2 norm of the gradients (csrc/adamw.hip).
 *
 * Replaces: torch.optim.AdamW(...).step() as finetune_speaker_v2.py:113-120 constructs it and :213-214 / :230-231 call it
 *           (through GradScaler.step), and commons.clip_grad_value_(parameters, None) (commons.py:149-164), which with
 *           clip_value None only returns the total L2 norm of the gradients (finetune_speaker_v2.py:212,229).
 *   p, m, v      float32 flat buffers (parameters, exp_avg, exp_avg_sq) sharing one offset space; p == NULL: norm only;
 *   host_entries HOST array: entry i = one contiguous run of `n` gradient floats at device address `g` updating the flat
 *                offsets [offset, offset + n) (passed on to the kernels by value: nothing is kept after the call returns);
 *   state        device float[2]: state[0] = learning rate, state[1] = number of completed steps (the bias corrections use
 *                state[1] + 1); vits_gradnorm_final(bump_step = 1) increments it — run it after the update;
 *   partials     device float[>= vits_adamw_blocks(entries)] (or NULL): per-workgroup sums of g^2;
 *   vits_gradnorm_final: norm_out[0] = sqrt(sum of partials[0..n)) in a fixed order (bitwise reproducible).
 *   Update rule (decoupled weight decay, no amsgrad), t = state[1] + 1:
 *     p *= 1 - lr*wd;  m += (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g^2;
 *     p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* g;              /* device: first gradient element of the run */
  unsigned long long offset;   /* first flat element the run updates */
  unsigned int n;              /* elements */
  unsigned int reserved;
} vits_adamw_entry;
size_t vits_adamw_blocks(const vits_adamw_entry* host_entries, int n_entries);
int vits_adamw(float* p, float* m, float* v, const vits_adamw_entry* host_entries, int n_entries, const float* state,
               double beta1, double beta2, double eps, double weight_decay, float* partials, size_t partials_len, void* stream);
int vits_gradnorm_final(const float* partials, size_t n, float* norm_out, float* state, int bump_step, void* stream);

/* ------------------------------------------------------------------------------------------
 * The two-channel flows of the stochastic duration predictor without their glue (csrc/rq_spline.hip, csrc/flow_edge.hip).
 *
 * Replaces, per modules.ConvFlow layer (modules.py:346-390) and the modules.Flip after it (modules.py:273-279):
 *   - `x0, x1 = torch.split(x, ...)`, `self.pre(x0)` (Conv1d(1, C, 1)) and the `x + g` of DDSConv.forward (modules.py:96):
 *       vits_flow_front:      h[r][c] = x[r][c0] * w[c] + bias[c] (+ g[r][c])       x fp32 [rows][xs], xs = 1 | 2; h, g `dtype`
 *       vits_flow_front_bwd:  dx [rows][xs] (channel c0 = sum_c dh w, the other 0), dw[c] = sum_r dh x, db[c] = sum_r dh
 *   - the spline call with `x = torch.cat([x0, x1], 1) * x_mask` and `logdet = torch.sum(logabsdet * x_mask, [1, 2])` around it:
 *       vits_flow_spline:     y2[r][c1] = spline(x2[r][c1]; h[r]) * mask[r],  y2[r][1-c1] = x2[r][1-c1] * mask[r],
 *                             lad_masked[r] = logabsdet[r] * mask[r]          (sum it per item with vits_segsum_f32)
 *       vits_flow_spline_bwd: dx2, gh from dy2 [rows][2] and dlogdet [rows / t] (per item)
 *   - Flip: alternate c1 between consecutive layers instead of reversing the two channels in memory.
 * ------------------------------------------------------------------------------------------ */
/* Mean-only residual coupling layer (modules.py:330-343) + the Flip after it (modules.py:273-279), element-wise part:
 *   y = flip_channels([x0, stats + x1 * mask])  (flip = 0: no flip);  x, y [b][t][c], stats [b][t][c - half], mask = t < lengths[b];
 *   backward: dx and dstats from dy. */
int vits_coupling_tail(int dtype, const void* x, const void* stats, const int32_t* lengths, void* y, int b, int t, int c, int half, int flip,
                       void* stream);
int vits_coupling_tail_bwd(int dtype, const void* dy, const int32_t* lengths, void* dx, void* dstats, int b, int t, int c, int half, int flip,
                           void* stream);
/* modules.ElementwiseAffine (modules.py:280-295) on a channels-last float32 [b][t][c] state (c <= 8; swap: parameters indexed as if
 * the channels were flipped): y = (m + exp(logs) x) mask, logdet[b] = sum(logs) len[b]; inverse = 1: y = (x - m) exp(-logs) mask
 * (logdet may be NULL).  Backward (forward direction): dx, dm[c], dlogs[c] from dy and dlogdet (either may be NULL = zero). */
int vits_flow_affine(const float* x, const float* m, const float* logs, const int32_t* lengths, float* y, float* logdet, int b, int t, int c,
                     int swap, int inverse, void* stream);
int vits_flow_affine_bwd(const float* x, const float* logs, const int32_t* lengths, const float* dy, const float* dlogdet, float* dx, float* dm,
                         float* dlogs, int b, int t, int c, int swap, void* stream);
/* The variational-dequantisation step between the two flow chains of StochasticDurationPredictor.forward (models.py:71-80) with
 * modules.Log (modules.py:259-267): zq [b][t][2] = [z_u, z1], w [b][t] durations ->
 *   out [b][t][2] = [log(clamp_min((w - sigmoid(z_u) m) m, 1e-5)) m, z1],  s1[b] = sum_t (logsigmoid(z_u) + logsigmoid(-z_u)) m,
 *   s2[b] = sum_t -out[..., 0];  backward: dzq from dout, ds1, ds2 (each may be NULL = zero). */
int vits_flow_dequant_log(const float* zq, const float* w, const int32_t* lengths, float* out, float* s1, float* s2, int b, int t, void* stream);
int vits_flow_dequant_log_bwd(const float* zq, const float* w, const int32_t* lengths, const float* dout, const float* ds1, const float* ds2,
                              float* dzq, int b, int t, void* stream);
int vits_flow_front(int dtype, const float* x, int xs, int c0, const float* w, const float* bias, const void* g, void* h, int rows,
                    int c, void* stream);
size_t vits_flow_front_workspace(int rows, int c);
int vits_flow_front_bwd(int dtype, const float* x, int xs, int c0, const float* w, const void* dh, float* dx, float* dw, float* db,
                        void* workspace, size_t workspace_bytes, int rows, int c, void* stream);
int vits_flow_spline(int h_dtype, const float* x2, const void* h, int ldh, float hscale, int inverse, float tail_bound,
                     const float* mask, int c1, float* y2, float* lad_masked, int n, void* stream);
int vits_flow_spline_bwd(int h_dtype, const float* x2, const void* h, int ldh, float hscale, int inverse, float tail_bound,
                         const float* mask, int c1, const float* dy2, const float* dlogdet, int t, float* dx2, void* gh, int n,
                         void* stream);

/* ------------------------------------------------------------------------------------------
 * Feature-matching loss over all feature maps of all discriminators in one pass (csrc/reduce.hip).
 *
 * Replaces: losses.feature_loss (losses.py:7-15): for every feature map `torch.mean(torch.abs(rl.detach() - gl))`, summed,
 *           times 2 — 37 maps per step, each a subtraction, abs, mean and add, plus their autograd.
 *   item e : h = one feature map, contiguous, 2 n elements: first half the real items, second half the generated ones;
 *            scale = 2 / (elements of one half, not counting zero padding channels);  dh (backward only): same shape,
 *            every element written (the real half gets zeros: the reference detaches it);
 *   out[0] = sum_e scale_e * sum |real - generated|   (fixed summation order);   g: device float, gradient of out[0];
 *   host_items is a HOST array (passed on by value), n_items <= 48.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const void* h;
  void* dh;
  size_t n;
  float scale;
} vits_feat_item;
size_t vits_feature_l1_workspace(int n_items);
int vits_feature_l1(int dtype, const vits_feat_item* host_items, int n_items, float* out, void* workspace, size_t workspace_bytes,
                    void* stream);
int vits_feature_l1_bwd(int dtype, const vits_feat_item* host_items, int n_items, const float* g, void* stream);

/* ------------------------------------------------------------------------------------------
 * Least-squares GAN losses over the logits of all discriminators in one pass (csrc/reduce.hip).
 *
 * Replaces: losses.discriminator_loss (losses.py:18-32) and losses.generator_loss (losses.py:35-43): per discriminator
 *           `torch.mean((1 - dr) ** 2)`, `torch.mean(dg ** 2)` resp. `torch.mean((1 - dg) ** 2)`, their sum, and the autograd
 *           of those ~10 element-wise ops per discriminator.
 *   item d : y8 [J][R][8] logits of discriminator d as vits_disc_post_fwd leaves them (channel 0 live), items j < J/2 real,
 *            the rest generated; dy8 same shape (backward only): every element is written (zeros where no term applies);
 *   mode 0 : out[0] = sum_d ( mean_real (1 - y)^2 + mean_generated y^2 );  mode 1: out[0] = sum_d mean_generated (1 - y)^2;
 *            out[1 + 2d], out[2 + 2d] = the real / generated term of discriminator d (r_losses / g_losses / gen_losses);
 *            total (optional) receives a second copy of out[0];
 *   g      : device float, the gradient of out[0];  host_items is a HOST array (passed on by value), n_items <= 8.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const void* y8;
  void* dy8;
  int J, R;
} vits_lsgan_item;
size_t vits_lsgan_workspace(int n_items);
int vits_lsgan_loss(int dtype, const vits_lsgan_item* host_items, int n_items, int mode, float* out, float* total,
                    void* workspace, size_t workspace_bytes, void* stream);
int vits_lsgan_loss_bwd(int dtype, const vits_lsgan_item* host_items, int n_items, int mode, const float* g, void* stream);

/* ------------------------------------------------------------------------------------------
 * Kernels around the alignment step (csrc/align.hip).
 *
 * vits_neg_cent — replaces models.py:470-477 (s_p_sq_r, neg_cent1..4 and their sum: two matmuls, two reductions and
 *   six element-wise ops over [b, t_t, t_s] tensors) by one fp32 matrix-core product [z^2 | z] x [-0.5 r | m r]^T + bias:
 *     z    [b][t_t][c] channels-last latent frames (row pitch ldz elements), z_dtype VITS_DT_F32 | VITS_DT_BF16;
 *     m, logs [b][t_s][c] prior statistics (row pitch lds elements; they may be the two halves of one [b][t_s][2c]
 *          tensor), s_dtype likewise;
 *     nc   [b][t_t][t_s] float32, fully written — the `neg_cent` operand of vits_mas_f32.
 * vits_slice_segments — replaces commons.slice_segments (commons.py:48-57; used by rand_slice_segments :60-67 and
 *   finetune_speaker_v2.py:190,203) and, with backward = 1, its gradient:
 *     time_inner = 0: x [b][t][d] channels-last -> y [b][seg][d];  time_inner = 1: x [b][d][t] -> y [b][d][seg];
 *     item i starts at ids[i] * ids_mul (int64 ids as torch makes them); rows outside [0, t) read as 0;
 *     backward = 1: x is the gradient of the slice ([b][seg][d] / [b][d][seg]) and y the full-size gradient, every
 *     element written (zeros outside the segment).  elem_bytes 2 | 4 (values are moved, not interpreted).
 * vits_generate_path — replaces commons.generate_path (commons.py:131-146):
 *     duration float32 [b][t_x], mask float32 [b][t_y][t_x] -> path float32 [b][t_y][t_x].
 * ------------------------------------------------------------------------------------------ */
int vits_neg_cent(int z_dtype, const void* z, long ldz, int s_dtype, const void* m, const void* logs, long lds, float* nc,
                  int b, int t_t, int t_s, int c, void* stream);
int vits_slice_segments(int elem_bytes, int time_inner, const void* x, const int64_t* ids, long ids_mul, void* y, int b, int d,
                        int t, int seg, int backward, void* stream);
int vits_generate_path(const float* duration, const float* mask, float* path, int b, int t_y, int t_x, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VITSMI_H */
