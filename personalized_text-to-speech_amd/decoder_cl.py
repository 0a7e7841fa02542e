"""HiFi-GAN decoder (reference models.Generator, models.py:244-289; modules.ResBlock1/2 :187-256) as
ONE autograd node over the channels-last HIP kernels.

Forward = 1 + 4*(1+1) + 72 + 1 launches for the reference config (conv_pre, per stage the
upsampler's 1x1 product + fold, 2 convolutions per ResBlock1 unit, conv_post); every element-wise
op of the reference graph is folded into a convolution's prologue/epilogue:

    reference op (models.py / modules.py)            here
    -----------------------------------------------  -------------------------------------------------
    x + self.cond(g)                     :272-273    per-item bias `bias_b` of conv_pre
    F.leaky_relu(x, 0.1) before ups / c1 / c2        `in_slope` prologue of the consuming convolution
    xt + x  (ResBlock skip)              :220        `res` epilogue of c2
    xs += resblock(x);  xs / num_kernels :279-284    c2 of the last unit writes (.)/3 into xs (ACCUM)
    F.leaky_relu(x) ; conv_post ; tanh   :285-287    `in_slope=0.01` prologue + TANH epilogue

Backward is written by hand with the same kernels: data gradients are the forward kernel on dY with
tap-flipped transposed weights and the leaky-relu derivative as `mg_src` epilogue; weight gradients
come from vits_conv1d_cl_wgrad (fp32, reproducible).
"""
import torch

from . import _lib
from . import kernels as K
from . import weight_arena as WA


def convt_fold(p, bias, c_out, k, u, pad):
    b, t_in, _ = p.shape
    t_out = (t_in - 1) * u - 2 * pad + k
    y = torch.empty((b, t_out, c_out), device=p.device, dtype=p.dtype)
    rc = _lib.lib().vits_convt_fold_cl(K._DT[p.dtype], p.data_ptr(), None if bias is None else bias.data_ptr(), y.data_ptr(),
                                       b, t_in, c_out, k, u, pad, _lib.stream_ptr())
    _lib.check(rc, "vits_convt_fold_cl")
    return y


def convt_unfold(dy, t_in, k, u, pad):
    b, t_out, c_out = dy.shape
    dp = torch.empty((b, t_in, k * c_out), device=dy.device, dtype=dy.dtype)
    rc = _lib.lib().vits_convt_unfold_cl(K._DT[dy.dtype], dy.data_ptr(), dp.data_ptr(), b, t_in, c_out, k, u, pad, _lib.stream_ptr())
    _lib.check(rc, "vits_convt_unfold_cl")
    return dp


def flip_t(w):
    """[k][co][ci] -> data-gradient weights [k][ci][co] with the taps reversed."""
    return w.flip(0).transpose(1, 2).contiguous()


class DecoderPlan:
    """Static description of a Generator instance (shapes only; no tensors)."""

    def __init__(self, gen):
        self.resblock1 = gen.resblocks[0].__class__.__name__ == "ResBlock1"
        self.num_kernels = gen.num_kernels
        self.ups = [(m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding) for m in gen.ups]
        self.res = []                      # per resblock: (channels, k, dilations)
        for rb in gen.resblocks:
            convs = rb.convs1 if self.resblock1 else rb.convs
            self.res.append((convs[0].in_channels, convs[0].kernel_size, [c.dilation for c in convs]))
        self.c_last = gen.conv_post.in_channels


def prepared_weights(gen):
    """Weights in DecoderFn's consumption order.  Inside a weight_arena.scope these are the arena's
    handles; otherwise fp32 kernel-layout tensors prepared with torch ops (autograd-connected):
    conv [c_out,c_in,k] -> [k][c_out][c_in];  conv-transpose [c_in,c_out,k] -> [1][k*c_out][c_in];
    conv_post padded to 8 output channels (vector width of the kernels)."""
    def conv(m):
        h = WA.handle_for(m)
        return h if h is not None else m.weight.permute(2, 0, 1).contiguous()

    def convt(m):
        h = WA.handle_for(m)
        if h is not None:
            return h
        return m.weight.permute(2, 1, 0).reshape(1, m.kernel_size * m.out_channels, m.in_channels).contiguous()

    out = [conv(gen.conv_pre), gen.conv_pre.bias]
    for i, up in enumerate(gen.ups):                               # consumption order of DecoderFn.forward
        out += [convt(up), up.bias]
        for rb in gen.resblocks[i * gen.num_kernels:(i + 1) * gen.num_kernels]:
            pairs = zip(rb.convs1, rb.convs2) if hasattr(rb, "convs1") else [(c,) for c in rb.convs]
            for group in pairs:
                for c in group:
                    out += [conv(c), c.bias]
    h = WA.handle_for(gen.conv_post)
    if h is None:
        wp = gen.conv_post.weight                                  # [1, c, 7]
        wp = torch.cat([wp, wp.new_zeros(7, wp.size(1), wp.size(2))], 0)
        h = wp.permute(2, 0, 1).contiguous()
    out.append(h)
    return out


class DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, dtype, z, cond, *wb):
        """z [b, t, c] channels-last (any float dtype), cond float32 [b, c_up0] or None, wb from
        prepared_weights().  Returns y [b, t*prod(u), 8] in `dtype` (channel 0 is the waveform)."""
        C = K.conv1d_cl_raw
        R = [WA.resolve(t, dtype) if t.dim() == 3 else None for t in wb]
        w = [r.fwd if r is not None else t.detach().float() for r, t in zip(R, wb)]        # weights in compute dtype, biases fp32
        it = iter(range(len(w)))
        saved = []
        z = z.detach().to(dtype).contiguous()
        i_pre = (next(it), next(it))
        h = C(z, w[i_pre[0]], w[i_pre[1]], bias_b=None if cond is None else cond.detach().float().contiguous(), pad=3)
        saved.append(z)
        idx = {"pre": i_pre, "ups": [], "res": []}
        ri = 0
        for (c_in, c_out, k, u, pad) in plan.ups:
            iu = (next(it), next(it))
            idx["ups"].append(iu)
            p = C(h, w[iu[0]], None, in_slope=0.1)
            x = convt_fold(p, w[iu[1]], c_out, k, u, pad)
            saved.append(h)
            xs = torch.empty_like(x)
            # weight slots in module order (resblock by resblock, unit by unit) ...
            nk, stage = plan.num_kernels, plan.res[ri:ri + plan.num_kernels]
            slots = []
            for j in range(nk):
                per = []
                for _ in stage[j][2]:
                    i1 = (next(it), next(it))
                    per.append((i1, (next(it), next(it)) if plan.resblock1 else None))
                slots.append(per)
            ri += nk
            # ... but the launches go unit by unit ACROSS the parallel resblocks: the k = 3 / 7 / 11 resblocks of a stage read the
            # same x and are independent until their sum — at the first stage each of their convolutions alone is half a
            # round of workgroups, side by side (K.conv1d_cl_multi) they share the chip and the launch boundary.  The last
            # convolution of each resblock accumulates into the shared sum and stays one launch after the other.
            nd = len(stage[0][2])
            assert all(len(st[2]) == nd for st in stage)
            r = [x] * nk
            rec = [[None] * nd for _ in range(nk)]
            for l in range(nd):
                last = l == nd - 1
                geo = [(stage[j][1], stage[j][2][l]) for j in range(nk)]                      # (kernel size, dilation)
                if plan.resblock1:
                    t1s = K.conv1d_cl_multi([(r[j], w[slots[j][l][0][0]], dict(bias=w[slots[j][l][0][1]], dil=d, pad=(rk * d - d) // 2, in_slope=0.1))
                                             for j, (rk, d) in enumerate(geo)])
                    for j in range(nk):
                        rec[j][l] = (r[j], t1s[j])
                    if last:
                        for j, (rk, d) in enumerate(geo):
                            i2 = slots[j][l][1]
                            C(t1s[j], w[i2[0]], w[i2[1]], res=r[j], out=xs, pad=(rk - 1) // 2, in_slope=0.1,
                              out_scale=1.0 / nk, flags=K.CONV_ACCUM if j > 0 else 0)
                    else:
                        r = K.conv1d_cl_multi([(t1s[j], w[slots[j][l][1][0]], dict(bias=w[slots[j][l][1][1]], res=r[j], pad=(rk - 1) // 2, in_slope=0.1))
                                               for j, (rk, d) in enumerate(geo)])
                else:
                    for j in range(nk):
                        rec[j][l] = (r[j],)
                    if last:
                        for j, (rk, d) in enumerate(geo):
                            i1 = slots[j][l][0]
                            C(r[j], w[i1[0]], w[i1[1]], res=r[j], out=xs, dil=d, pad=(rk * d - d) // 2, in_slope=0.1,
                              out_scale=1.0 / nk, flags=K.CONV_ACCUM if j > 0 else 0)
                    else:
                        r = K.conv1d_cl_multi([(r[j], w[slots[j][l][0][0]], dict(bias=w[slots[j][l][0][1]], res=r[j], dil=d, pad=(rk * d - d) // 2, in_slope=0.1))
                                               for j, (rk, d) in enumerate(geo)])
            for j in range(nk):                                   # (saved in module order: the backward pops it in reverse)
                idx["res"].append([(slots[j][l][0], slots[j][l][1], stage[j][2][l]) for l in range(nd)])
                for l in range(nd):
                    saved += list(rec[j][l])
            h = xs
        i_post = next(it)
        y = C(h, w[i_post], None, pad=3, in_slope=0.01, flags=K.CONV_TANH)
        saved += [h, y]
        ctx.plan, ctx.dtype, ctx.idx, ctx.i_post = plan, dtype, idx, i_post
        ctx.has_cond = cond is not None
        ctx.R = R
        ctx.save_for_backward(*saved)
        return y

    @staticmethod
    def backward(ctx, dy):
        plan, dtype, idx = ctx.plan, ctx.dtype, ctx.idx
        C, WG = K.conv1d_cl_raw, K.conv1d_cl_wgrad_raw
        saved = list(ctx.saved_tensors)
        R = ctx.R
        grads = [None] * len(R)
        defer = K.DeferredReductions(dy.device)      # every slab reduction of this backward in one launch (or a few) at the end

        def flip_t(r):                       # data-gradient operand of weight slot r
            return WA.bwd_operand(R[r])

        batch = []                           # the weight gradients of the whole decoder: launched together at the end

        def WGo(x, dy, k, slot, bias_slot=None, **kw):     # weight gradient into the arena's dw region when there is one;
            db = None                                      # the bias gradient (column sums of dy) rides in the same launch
            if bias_slot is not None:
                db = torch.empty(dy.size(2), dtype=torch.float32, device=dy.device)
                grads[bias_slot] = db
            out = R[slot].claim_dw(ctx)
            if out is None:
                out = torch.empty(k, dy.size(2), x.size(2), dtype=torch.float32, device=dy.device)
            # (x and dy stay referenced by the batch until it is launched: the intermediate gradients are fresh tensors)
            batch.append(dict(x=x, dy=dy if dy.is_contiguous() else dy.contiguous(), k=k, out=out, dbias=db, **kw))
            return out

        def bias_grad(d):
            return K.colsum(d if d.is_contiguous() else d.contiguous())

        y = saved.pop()
        h = saved.pop()
        dpre = (dy.to(torch.float32) * (1.0 - y.float() ** 2)).to(dtype).contiguous()      # tanh'
        grads[ctx.i_post] = WGo(h, dpre, 7, ctx.i_post, pad=3, in_slope=0.01)
        dh = C(dpre, flip_t(ctx.i_post), None, mg_src=h, pad=3, mg_slope=0.01)

        ri = len(plan.res)
        for s in reversed(range(len(plan.ups))):
            c_in, c_out, k, u, pad = plan.ups[s]
            dxs = dh * (1.0 / plan.num_kernels)
            dx = torch.empty_like(dxs)
            nk = plan.num_kernels
            ri -= nk
            stage, units = plan.res[ri:ri + nk], idx["res"][ri:ri + nk]
            nd = len(stage[0][2])
            rec = [[None] * nd for _ in range(nk)]                 # saved in module order: pop resblocks and units in reverse
            for j in reversed(range(nk)):
                for l in reversed(range(nd)):
                    if plan.resblock1:
                        t1 = saved.pop()
                        rec[j][l] = (saved.pop(), t1)
                    else:
                        rec[j][l] = (saved.pop(),)
            dr = [dxs] * nk
            for l in reversed(range(nd)):                          # unit by unit across the parallel resblocks (see forward)
                first = l == 0
                geo = [(stage[j][1], units[j][l][2]) for j in range(nk)]
                if plan.resblock1:
                    for j, (rk, d) in enumerate(geo):
                        i2 = units[j][l][1]
                        grads[i2[0]] = WGo(rec[j][l][1], dr[j], rk, i2[0], i2[1], pad=(rk - 1) // 2, in_slope=0.1)
                    dt1 = K.conv1d_cl_multi([(dr[j], flip_t(units[j][l][1][0]), dict(mg_src=rec[j][l][1], pad=(rk - 1) // 2, mg_slope=0.1))
                                             for j, (rk, d) in enumerate(geo)])
                else:
                    dt1 = dr
                for j, (rk, d) in enumerate(geo):
                    i1 = units[j][l][0]
                    grads[i1[0]] = WGo(rec[j][l][0], dt1[j], rk, i1[0], i1[1], dil=d, pad=(rk * d - d) // 2, in_slope=0.1)
                if first:      # gradient wrt the stage input x: accumulated over the parallel resblocks, one launch after the other
                    for j in reversed(range(nk)):
                        rk, d = geo[j]
                        _dgrad_res(dt1[j], flip_t(units[j][l][0][0]), rec[j][l][0], dr[j], d, (rk * d - d) // 2, out=dx, accum=j < nk - 1)
                else:
                    dr = K.conv1d_cl_multi([(dt1[j], flip_t(units[j][l][0][0]),
                                             dict(res=dr[j], mg_src=rec[j][l][0], dil=d, pad=(rk * d - d) // 2, mg_slope=0.1, flags=K.CONV_RES_AFTER))
                                            for j, (rk, d) in enumerate(geo)])
            # upsampler: x = fold(conv1x1(lrelu(h_prev)))
            h_prev = saved.pop()
            iu = idx["ups"][s]
            grads[iu[1]] = bias_grad(dx)
            dp = convt_unfold(dx, h_prev.size(1), k, u, pad)
            grads[iu[0]] = WGo(h_prev, dp, 1, iu[0], in_slope=0.1)
            dh = C(dp, flip_t(iu[0]), None, mg_src=h_prev, mg_slope=0.1)
        z = saved.pop()
        i_pre = idx["pre"]
        grads[i_pre[0]] = WGo(z, dh, 7, i_pre[0], i_pre[1], pad=3)
        dz = C(dh, flip_t(i_pre[0]), None, pad=3)
        dcond = K.colsum(dh if dh.is_contiguous() else dh.contiguous(), per_item=True) if ctx.has_cond else None
        if not K.conv1d_cl_wgrad_batch(batch, defer):
            for e in batch:                                         # (not eligible: one launch per convolution)
                WG(e["x"], e["dy"], e["k"], dil=e.get("dil", 1), pad=e.get("pad", 0), in_slope=e.get("in_slope", 1.0), out=e["out"],
                   dbias=e["dbias"], defer=defer)
        defer.flush()
        return (None, None, dz, dcond, *grads)


def _dgrad_res(dy, w_t, x_in, skip, dil, pad, out=None, accum=False):
    """d/dx of  conv(lrelu_0.1(x)) + x  given the output gradient:  conv^T(dy) * lrelu'(x) + skip,
    one kernel (RES_AFTER adds the skip term after the activation-derivative multiplier)."""
    return K.conv1d_cl_raw(dy, w_t, None, res=skip, mg_src=x_in, out=out, dil=dil, pad=pad, mg_slope=0.1,
                           flags=K.CONV_RES_AFTER | (K.CONV_ACCUM if accum else 0))
