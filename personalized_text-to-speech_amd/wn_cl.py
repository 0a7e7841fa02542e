"""Channels-last building blocks on the HIP convolution kernels: a differentiable fused
convolution (`conv_cl`) and the gated WaveNet stack of the reference (modules.WN, modules.py:111-184)
as one autograd node with a hand-written backward.

WN layer i in the reference graph (modules.py:157-176, commons.py:103-110)       here
----------------------------------------------------------------------------  ---------------------------
x_in = in_layers[i](x)  (+ bias)                                               conv k, GATE epilogue:
g_l  = g[:, i*2H:(i+1)*2H]                                                        bias_b = cond[i] (per item),
acts = tanh((x_in+g_l)[:, :H]) * sigmoid((x_in+g_l)[:, H:])                       y = acts, y2 = pre-activations
res_skip = res_skip_layers[i](acts)                                            two 1x1 convs on acts:
x = (x + res_skip[:, :H]) * x_mask                                                res  -> `res` + MASK_OUT epilogue
output = output + res_skip[:, H:]          ... return output * x_mask             skip -> MASK_OUT + ACCUM into `output`
"""
import torch

from . import _lib
from . import kernels as K
from . import weight_arena as WA


def flip_t(w):
    """[k][co][ci] -> data-gradient weights [k][ci][co] with the taps reversed."""
    return w.flip(0).transpose(1, 2).contiguous()


def lengths_of(mask):
    """int32 [b] valid lengths of a prefix mask [b, 1, t] (commons.sequence_mask output)."""
    return mask[:, 0, :].sum(-1).to(torch.int32)


def compute_dtype():
    return torch.bfloat16 if torch.is_autocast_enabled() else torch.float32


def prep_conv(weight):
    """torch Conv1d weight [c_out, c_in, k] -> kernel layout [k][c_out][c_in] (fp32, autograd-connected)."""
    return weight.permute(2, 0, 1).contiguous()


class ConvCLFn(torch.autograd.Function):
    """y = mask_out( conv(lrelu_slope(mask_in(x)), w) + bias + res ), x [b,t,c_in] channels-last (may be a
    channel slice), w: arena handle or fp32 kernel-layout weight [k][c_out][c_in].  Data and weight gradients
    by the same HIP kernels (in_slope = 0 is a fused ReLU on the input)."""

    @staticmethod
    def forward(ctx, dtype, x, w, bias, lengths, dil, pad, in_slope, mask_in, mask_out, res, stride, out_slope, groups=1):
        xd = x.detach()
        if xd.dtype != dtype:
            xd = xd.to(dtype)
        R = WA.resolve(w, dtype)
        flags = (K.CONV_MASK_IN if mask_in else 0) | (K.CONV_MASK_OUT if mask_out else 0)
        rd = None if res is None else res.detach().to(dtype).contiguous()
        y = K.conv1d_cl_raw(xd, R.fwd, None if bias is None else bias.detach().float(), res=rd, lengths=lengths, dil=dil, pad=pad,
                            stride=stride, in_slope=in_slope, flags=flags, out_slope=out_slope, groups=groups)
        ctx.groups = groups
        ctx.bias_slot = WA.bias_slot(bias)
        ctx.save_for_backward(xd, y if out_slope is not None else xd)
        ctx.R, ctx.out_slope = R, out_slope
        ctx.lengths, ctx.cfg, ctx.has_bias, ctx.x_dtype = lengths, (dil, pad, in_slope, mask_in, mask_out, stride), bias is not None, x.dtype
        ctx.res_dtype = None if res is None else res.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        xd, ysave = ctx.saved_tensors
        R = ctx.R
        dil, pad, in_slope, mask_in, mask_out, stride = ctx.cfg
        k = R.fwd.size(0)
        dy = dy.contiguous()
        if ctx.out_slope is not None or mask_out:
            # chain rule of the fused output leaky-relu (sign of y = sign of the pre-activation) and of the output mask
            dy = K.lrelu_mask_bwd(dy, ysave if ctx.out_slope is not None else None, ctx.out_slope if ctx.out_slope is not None else 1.0,
                                  ctx.lengths if mask_out else None)
        dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[3]
        if ctx.needs_input_grad[2]:
            dw_out = R.claim_dw(ctx)
            deferred = dw_out is not None and R.defer is not None
            if want_db and dy.size(2) % 8 == 0:
                # riding in the weight-gradient launch.  When that launch's second stage is deferred (arena), the result is only
                # final after the arena's flush: then it must live in the arena's own bias-gradient buffer, which reaches the
                # parameter through PrepFn.backward — a fresh tensor handed to autograd now could be read before it is written
                if ctx.bias_slot is not None:
                    db = ctx.bias_slot[0].db_views[ctx.bias_slot[1]]
                elif not deferred:
                    db = torch.empty(dy.size(2), dtype=torch.float32, device=dy.device)
            # inside an arena the whole launch is deferred to the arena's flush and batched with the other single convolutions
            # of its stream (stride-1 "same" convolutions; the rest launch here, only their second stage deferred)
            if deferred and stride == 1 and ctx.groups == 1 and R.defer.add_wgrad(
                    dict(x=xd, dy=dy, k=k, out=dw_out, dbias=db, lengths=ctx.lengths, dil=dil, pad=pad, in_slope=in_slope,
                         flags=K.CONV_MASK_IN if mask_in else 0)):
                dw = dw_out
            else:
                dw = K.conv1d_cl_wgrad_raw(xd, dy, k, lengths=ctx.lengths, dil=dil, pad=pad, stride=stride, in_slope=in_slope,
                                           flags=K.CONV_MASK_IN if mask_in else 0, out=dw_out, dbias=db, groups=ctx.groups,
                                           defer=R.defer if deferred else None)
        if want_db and db is None:
            db = K.colsum(dy)
        dx = None
        if ctx.needs_input_grad[1]:
            xs = xd if xd.is_contiguous() else xd.contiguous()
            # stride > 1: the kernel walks dY on the input's time grid phase by phase (in_div), no zero-insertion
            dx = K.conv1d_cl_raw(dy, WA.bwd_operand(R), None, mg_src=xs if in_slope != 1.0 else None, lengths=ctx.lengths, dil=dil,
                                 pad=dil * (k - 1) - pad, mg_slope=in_slope, flags=K.CONV_MASK_OUT if mask_in else 0,
                                 in_div=stride, t_out=xd.size(1) if stride != 1 else None, groups=ctx.groups)
            if dx.dtype != ctx.x_dtype:
                dx = dx.to(ctx.x_dtype)
        dres = None
        if ctx.res_dtype is not None and ctx.needs_input_grad[10]:
            dres = dy if dy.dtype == ctx.res_dtype else dy.to(ctx.res_dtype)
        return None, dx, dw, db, None, None, None, None, None, None, dres, None, None, None


def conv_cl(x, w, bias=None, lengths=None, dil=1, pad=0, in_slope=1.0, mask_in=False, mask_out=False, dtype=None, res=None, stride=1,
            out_slope=None, groups=1):
    return ConvCLFn.apply(dtype or compute_dtype(), x, w, bias, lengths, dil, pad, in_slope, mask_in, mask_out, res, stride, out_slope, groups)


def weight_of(module, part=None, pad_in=0, pad_out=0):
    """Kernel-layout weight of a conv module: the arena handle inside a weight_arena.scope, else the
    torch-prepared fp32 tensor (optionally with zero-padded input / output channels)."""
    h = WA.handle_for(module, part)
    if h is not None:
        return h
    w = module.weight
    if pad_in or pad_out:
        w = torch.nn.functional.pad(w, (0, 0, 0, pad_in, 0, pad_out))
    return prep_conv(w)


def bias_of(module, pad_out=0):
    """The bias operand of a conv module: the arena's padded fp32 bias handle inside a weight_arena.scope whose Spec manages it,
    else the parameter itself (zero-padded by `pad_out` output channels)."""
    h = WA.bias_handle_for(module)
    if h is not None:
        assert h.numel() == module.bias.numel() + pad_out
        return h
    b = module.bias
    return torch.nn.functional.pad(b, (0, pad_out)) if (pad_out and b is not None) else b


class WNPlan:
    def __init__(self, wn):
        self.H, self.k, self.L = wn.hidden_channels, wn.kernel_size[0], wn.n_layers
        self.dils = [wn.dilation_rate ** i for i in range(self.L)]
        self.has_cond = wn.gin_channels != 0
        self.packed = {}             # (dtype, device) -> kernels.WnPacked: the stack's operands in fragment order


def wn_prepared_weights(wn):
    """Per layer: w_in, b_in, w_rs (res rows first, then skip rows; skip only for the last layer), b_rs."""
    out = []
    for i in range(wn.n_layers):
        rs = wn.res_skip_layers[i]
        w_in = WA.handle_for(wn.in_layers[i])
        if w_in is not None:
            w_rs = WA.handle_for(rs)
        else:
            w_in, w_rs = prep_conv(wn.in_layers[i].weight), prep_conv(rs.weight)
        out += [w_in, wn.in_layers[i].bias, w_rs, rs.bias]
    return out


def wn_cond(wn, g, n_items):
    """cond_layer(g) (modules.py:152-153) as [L][b][2H] float32, or None."""
    if g is None:
        return None
    if g.is_cuda and WA.handle_for(wn.cond_layer) is not None:
        # arena-managed: weight-norm by the arena's preparation launch, the product and both gradients on the convolution kernels
        dt = compute_dtype()
        c = conv_cl(g[:, :, 0].unsqueeze(1).to(dt), weight_of(wn.cond_layer), bias_of(wn.cond_layer), dtype=dt).float()      # [b, 1, 2HL]
    else:
        c = torch.nn.functional.linear(g[:, :, 0].float(), wn.cond_layer.weight[:, :, 0].float(), wn.cond_layer.bias.float())
    return c.view(n_items, wn.n_layers, 2 * wn.hidden_channels).transpose(0, 1).contiguous()


# True: one launch per WaveNet layer and direction (csrc/wn_layer.hip: gate convolution + gate + res/skip product with the gate
# tile in LDS; its mirror image for the data gradients).  Measured on the step's shape (b 16, t 500, H 192; tools/ubench_wn.py,
# profiles/r03_pmc_wn.txt): 38 us forward / 45 us backward per layer against 2 x ~10.8 us / 2 x ~10.8 us for the composition below
# inside the captured step — one wave per SIMD (152 KB of LDS, 223 + 96 registers) leaves the layer's ~6 k vector instructions of
# gate / epilogue arithmetic and its waits with nothing to overlap with, while the composition's launches run 1.5-3 workgroups
# per CU.  So the composition stays the default; the fused kernels are kept tested (tests/test_wn_layer_gpu.py) as the starting
# point for a two-waves-per-SIMD version.
FUSED_LAYERS = False


class WNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, dtype, x, lengths, cond, *wb):
        """x [b,t,H] channels-last with rows >= lengths already zero; cond [L][b][2H] fp32 or None;
        wb from wn_prepared_weights()."""
        C = K.conv1d_cl_raw
        H, L, k = plan.H, plan.L, plan.k
        R = [WA.resolve(t, dtype) if (t is not None and t.dim() == 3) else None for t in wb]
        bias = [t.detach().float() if (t is not None and t.dim() == 1) else None for t in wb]
        h = x.detach().to(dtype).contiguous()
        cd = None if cond is None else cond.detach().float().contiguous()
        out = torch.empty_like(h)
        saved = []
        fused = FUSED_LAYERS and lengths is not None and K.wn_fusable(H, k, max(plan.dils), h.element_size())
        packed = None
        if fused:
            # the stack's operands in MFMA-fragment order: one packing launch per forward (and, when a backward will follow, the
            # data-gradient operands in the same launch)
            packed = plan.packed.get((dtype, h.device))
            if packed is None:
                packed = plan.packed[(dtype, h.device)] = K.WnPacked(H, k, L, dtype, h.device)
            need_bwd = any(ctx.needs_input_grad)       # (grad mode is off inside forward; this is what tells whether a backward can follow)
            packed.fill([(R[4 * i].fwd, R[4 * i + 2].fwd) for i in range(L)],
                        [(WA.bwd_operand(R[4 * i + 2]), WA.bwd_operand(R[4 * i])) for i in range(L)] if need_bwd else None)
        for i in range(L):
            r_in, b_in, r_rs, b_rs = R[4 * i], bias[4 * i + 1], R[4 * i + 2], bias[4 * i + 3]
            d = plan.dils[i]
            pre = torch.empty(h.size(0), h.size(1), 2 * H, device=h.device, dtype=dtype)
            if fused:                                       # the whole layer as one launch (csrc/wn_layer.hip)
                acts = torch.empty_like(h)
                h_next = K.wn_layer_fwd(h, packed, i, b_in, None if cd is None else cd[i], b_rs, lengths, d, out,
                                        accumulate=i > 0, last=i == L - 1, pre=pre, acts=acts)
            else:
                acts = C(h, r_in.fwd, b_in, bias_b=None if cd is None else cd[i], dil=d, pad=(k * d - d) // 2,
                         flags=K.CONV_GATE, gate_h=H, out2=pre)
                acc = K.CONV_ACCUM if i > 0 else 0
                if i < L - 1:                               # rows [0, H) of the res_skip operand feed the residual, [H, 2H) the skip sum:
                    h_next = C(acts, r_rs.fwd, b_rs, res=h, lengths=lengths, flags=K.CONV_RES_SKIP | acc, gate_h=H, out2=out)   # one launch
                else:
                    h_next = None
                    C(acts, r_rs.fwd, b_rs, out=out, lengths=lengths, flags=K.CONV_MASK_OUT | acc)
            saved += [h, pre, acts]
            h = h_next
        ctx.plan, ctx.dtype, ctx.lengths, ctx.has_cond, ctx.R = plan, dtype, lengths, cd is not None, R
        ctx.fused, ctx.packed = fused, packed
        ctx.save_for_backward(*saved)
        return out

    @staticmethod
    def backward(ctx, d_out):
        """Per layer TWO data-gradient launches — through the res_skip convolution with the gate's chain rule as epilogue (reduction
        over 2H channels of the layer's own [d_h | d_o] buffer), and through the in-layer convolution with the residual path, written
        into the next layer's buffer — and then ALL weight / bias gradients of the stack as one batched launch per taps-per-group
        class (vits_conv1d_cl_wgrad_batch) plus ONE per-item column sum for the gradient of the conditioning."""
        plan, dtype, lengths, R = ctx.plan, ctx.dtype, ctx.lengths, ctx.R
        C, WG = K.conv1d_cl_raw, K.conv1d_cl_wgrad_raw
        H, L, k = plan.H, plan.L, plan.k
        saved = list(ctx.saved_tensors)
        grads = [None] * len(R)
        b, t = d_out.size(0), d_out.size(1)
        dev = d_out.device
        rowmask = (torch.arange(t, device=dev)[None, :, None] < lengths[:, None, None])
        # layer i reads dcat_all[i] = [d_h_i | d_o] (d_o = d(output * x_mask), the same for every layer: one broadcast copy)
        dcat_all = torch.empty(L, b, t, 2 * H, device=dev, dtype=dtype)
        dcat_all[..., H:] = (d_out * rowmask).to(dtype).unsqueeze(0)
        defer = K.DeferredReductions(dev)               # slab reductions of a small stack: one launch at the end
        if ctx.fused:
            return WNFn._backward_fused(ctx, dcat_all[L - 1], saved, grads, defer)
        d_pre_all = torch.empty(L, b, t, 2 * H, device=dev, dtype=dtype)
        dx_buf = torch.empty(b, t, 2 * H, device=dev, dtype=dtype)
        batch = []
        for i in reversed(range(L)):
            acts, pre, h = saved.pop(), saved.pop(), saved.pop()
            r_in, r_rs = R[4 * i], R[4 * i + 2]
            d = plan.dils[i]
            pad = (k * d - d) // 2
            last = i == L - 1
            dy_rs = dcat_all[i][..., H:] if last else dcat_all[i]
            C(dy_rs, WA.bwd_operand(r_rs), None, mg_src=pre, out=d_pre_all[i], lengths=lengths, flags=K.CONV_GATE_BWD | K.CONV_MASK_OUT, gate_h=H)
            dst = (dcat_all[i - 1] if i > 0 else dx_buf)[..., :H]
            C(d_pre_all[i], WA.bwd_operand(r_in), None, res=None if last else dcat_all[i][..., :H], out=dst, lengths=lengths, dil=d, pad=pad,
              flags=K.CONV_MASK_OUT | (0 if last else K.CONV_RES_AFTER))
            dw_rs, dw_in = r_rs.claim_dw(ctx), r_in.claim_dw(ctx)
            if dw_rs is None:
                dw_rs = torch.empty(1, H if last else 2 * H, H, device=dev, dtype=torch.float32)
            if dw_in is None:
                dw_in = torch.empty(k, 2 * H, H, device=dev, dtype=torch.float32)
            db_rs = torch.empty(H if last else 2 * H, dtype=torch.float32, device=dev)
            db_in = torch.empty(2 * H, dtype=torch.float32, device=dev)
            batch.append(dict(x=acts, dy=dy_rs, k=1, out=dw_rs, dbias=db_rs))
            batch.append(dict(x=h, dy=d_pre_all[i], k=k, dil=d, pad=pad, out=dw_in, dbias=db_in))
            grads[4 * i + 2], grads[4 * i + 3], grads[4 * i], grads[4 * i + 1] = dw_rs, db_rs, dw_in, db_in
        if not K.conv1d_cl_wgrad_batch(batch, defer):
            for e in batch:                                         # (not eligible: one launch per convolution)
                WG(e["x"], e["dy"], e["k"], dil=e.get("dil", 1), pad=e.get("pad", 0), out=e["out"], dbias=e["dbias"], defer=defer)
        defer.flush()
        dc = K.colsum(d_pre_all.view(L * b, t, 2 * H), per_item=True).view(L, b, 2 * H) if ctx.has_cond else None
        return (None, None, dx_buf[..., :H].contiguous(), None, dc, *grads)

    @staticmethod
    def _backward_fused(ctx, dcat, saved, grads, defer):
        """One data-gradient launch per layer (vits_wn_layer_bwd: the 1x1 data gradient, the gate's chain rule and the k-tap data
        gradient with its residual path); every layer writes its d_h into a buffer of its own, so neighbouring time tiles
        never see a half-updated row and the weight gradients can wait: ALL of the stack's weight / bias gradients are then
        one batched launch per taps-per-group class (vits_conv1d_cl_wgrad_batch: no per-split slabs when the stack has enough
        tiles), and the gradient of the conditioning is ONE per-item column sum over all layers' d_pre."""
        plan, lengths, R = ctx.plan, ctx.lengths, ctx.R
        WG = K.conv1d_cl_wgrad_raw
        H, L, k = plan.H, plan.L, plan.k
        b, t = dcat.size(0), dcat.size(1)
        dev, dtype = dcat.device, dcat.dtype
        d_pre_all = torch.empty(L, b, t, 2 * H, device=dev, dtype=dtype)
        # every layer keeps its own d_h (the res half of its res_skip weight gradient reads it after the loop)
        dh_all = torch.empty(max(L - 1, 1), b, t, H, device=dev, dtype=dtype)
        d_o = dcat[..., H:]
        batch, d_h = [], None
        for i in reversed(range(L)):
            acts, pre, h = saved.pop(), saved.pop(), saved.pop()
            r_in, r_rs = R[4 * i], R[4 * i + 2]
            d = plan.dils[i]
            pad = (k * d - d) // 2
            last = i == L - 1
            dst = dh_all[i - 1] if i > 0 else torch.empty(b, t, H, device=dev, dtype=dtype)
            K.wn_layer_bwd(d_h, d_o, pre, ctx.packed, i, lengths, d, last, d_pre_all[i], dst)
            # weight gradients: collected, launched together below (the layers of the stack are each other's parallelism)
            dw_rs, dw_in = r_rs.claim_dw(ctx), r_in.claim_dw(ctx)
            if dw_rs is None:
                dw_rs = torch.empty(1, H if last else 2 * H, H, device=dev, dtype=torch.float32)
            if dw_in is None:
                dw_in = torch.empty(k, 2 * H, H, device=dev, dtype=torch.float32)
            db_rs = torch.empty(H if last else 2 * H, dtype=torch.float32, device=dev)
            db_in = torch.empty(2 * H, dtype=torch.float32, device=dev)
            if last:
                batch.append(dict(x=acts, dy=d_o, k=1, out=dw_rs, dbias=db_rs))
            else:                                                   # rows [0, H) of W_rs see d_h, rows [H, 2H) see d_o
                batch.append(dict(x=acts, dy=d_h, k=1, out=dw_rs[:, :H], dbias=db_rs[:H]))
                batch.append(dict(x=acts, dy=d_o, k=1, out=dw_rs[:, H:], dbias=db_rs[H:]))
            batch.append(dict(x=h, dy=d_pre_all[i], k=k, dil=d, pad=pad, out=dw_in, dbias=db_in))
            grads[4 * i + 2], grads[4 * i + 3], grads[4 * i], grads[4 * i + 1] = dw_rs, db_rs, dw_in, db_in
            d_h = dst
        if not K.conv1d_cl_wgrad_batch(batch, defer):
            for e in batch:                                         # (not eligible: one launch per convolution)
                WG(e["x"], e["dy"], e["k"], dil=e.get("dil", 1), pad=e.get("pad", 0), out=e["out"], dbias=e["dbias"], defer=defer)
        defer.flush()
        dc = K.colsum(d_pre_all.view(L * b, t, 2 * H), per_item=True).view(L, b, 2 * H) if ctx.has_cond else None
        return (None, None, d_h, None, dc, *grads)


def wn_forward_cl(wn, x_cl, lengths, g):
    """modules.WN.forward on channels-last x [b,t,H] (rows >= lengths zero)."""
    if getattr(wn, "_plan", None) is None:
        wn._plan = WNPlan(wn)
    cond = wn_cond(wn, g, x_cl.size(0))
    return WNFn.apply(wn._plan, compute_dtype(), x_cl, lengths, cond, *wn_prepared_weights(wn))
