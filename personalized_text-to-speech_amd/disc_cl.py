"""The discriminators (reference models.py:299-386: DiscriminatorP x5, DiscriminatorS, MultiPeriodDiscriminator) as ONE
autograd node over the channels-last HIP kernels, with a hand-written backward.

Per discriminator (period p; DiscriminatorS is p = 1):

    reference graph                                      here
    ---------------------------------------------------  ------------------------------------------------------------------
    F.pad(reflect) to a multiple of p, view [b,1,T/p,p]  index arithmetic inside the first-layer kernels (vits_disc_first_*),
    first conv (1 -> 32 | 16 channels) + leaky_relu        which read the raw fp32 waveforms [n][T]: no padded 8-channel input
    middle convs (k 5 stride 3 | k 41 stride 4 grouped)  vits_conv1d_cl flat-row launches, leaky_relu as epilogue
    conv_post (1024 -> 1, k 3)                           vits_disc_post_* (a 3 x 1024 dot product per row)
    backward: leaky_relu', `+ d fmap` of feature_loss    epilogue of the NEXT layer's data-gradient launch (res + mg_src)

The items of a batch are [real ; generated].  In the generator step the discriminator is frozen and only the generated
half carries a gradient (reference losses.py:11 detaches the real feature maps): `n_lo` = first item that needs an input
gradient, and every backward launch then runs on the rows of items >= n_lo only.

The kernel wrappers below are module-level functions so that the CPU logic tests can replace them with torch emulations
(tests/cl_emul.py)."""
import torch

from . import _lib
from . import kernels as K
from . import weight_arena as WA

SLOPE = 0.1                 # modules.LRELU_SLOPE
_DT = K._DT


# ------------------------------------------------------------------------------------------------ kernel wrappers
def first_rows(T, p, k, s1, pad):
    return ((T + p - 1) // p + 2 * pad - k) // s1 + 1


def first_fwd(x, w, bias, p, k, s1, pad, c_out, dtype):
    """x float32 [n][T]; w arena operand [k][c_out][8] of `dtype` -> h1 [(n*p)][R1][c_out], leaky_relu applied."""
    _lib.require_cuda(x, w)
    n, T = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and w.dtype == dtype and w.is_contiguous() and tuple(w.shape) == (k, c_out, 8)
    y = torch.empty((n * p, first_rows(T, p, k, s1, pad), c_out), device=x.device, dtype=dtype)
    rc = _lib.lib().vits_disc_first_fwd(_DT[dtype], x.data_ptr(), w.data_ptr(), None if bias is None else bias.data_ptr(), y.data_ptr(),
                                        n, T, p, k, s1, pad, c_out, SLOPE, _lib.stream_ptr())
    _lib.check(rc, "vits_disc_first_fwd")
    return y


def grouped_direct_ok(x, c_in, c_out, k, stride, groups):
    """The group shapes csrc/grouped.hip covers (DiscriminatorS, models.py:343-349), bf16 on the GPU only."""
    return (x.is_cuda and x.dtype == torch.bfloat16 and groups > 1 and groups % 4 == 0 and c_in == 4 * groups and
            c_out // groups in (4, 16) and c_out % groups == 0 and k <= 64 and stride <= 4)


def grouped_fwd(x, w, bias, k, stride, pad, groups):
    """leaky_relu(grouped conv(x) + bias): x [n][t][c_in], w the dense block-diagonal operand [k][c_out][c_in]."""
    _lib.require_cuda(x, w)
    n, t, c_in = x.shape
    c_out = w.size(1)
    assert x.is_contiguous() and w.is_contiguous() and tuple(w.shape) == (k, c_out, c_in) and w.dtype == x.dtype
    y = torch.empty((n, (t + 2 * pad - k) // stride + 1, c_out), device=x.device, dtype=x.dtype)
    e0 = _lib.timer.start("vits_grouped_conv")
    rc = _lib.lib().vits_grouped_conv_fwd(_DT[x.dtype], x.data_ptr(), w.data_ptr(), None if bias is None else bias.data_ptr(), y.data_ptr(),
                                          n, t, c_in, c_out, k, stride, pad, groups, SLOPE, _lib.stream_ptr())
    if e0 is not None:
        _lib.timer.stop("vits_grouped_conv", e0, (2.0 * y.numel() * 4 * k, 2.0 * (x.numel() + y.numel()) + 2.0 * k * c_out * 4))
    _lib.check(rc, "vits_grouped_conv_fwd")
    return y


def grouped_dgrad(dy, w, res, mg_src, t_in, k, stride, pad, groups):
    """(conv^T(dy) + res) * lrelu'(mg_src): dy [n][t_out][c_out], res / mg_src [n][t_in][c_in]."""
    _lib.require_cuda(dy, w)
    n, t_out, c_out = dy.shape
    c_in = w.size(2)
    assert dy.is_contiguous() and w.is_contiguous() and tuple(w.shape) == (k, c_out, c_in) and (t_in + 2 * pad - k) // stride + 1 == t_out
    for t in (res, mg_src):
        assert t is None or (t.is_contiguous() and tuple(t.shape) == (n, t_in, c_in) and t.dtype == dy.dtype)
    dx = torch.empty((n, t_in, c_in), device=dy.device, dtype=dy.dtype)
    e0 = _lib.timer.start("vits_grouped_conv")
    rc = _lib.lib().vits_grouped_conv_dgrad(_DT[dy.dtype], dy.data_ptr(), w.data_ptr(), None if res is None else res.data_ptr(),
                                            None if mg_src is None else mg_src.data_ptr(), dx.data_ptr(), n, t_in, c_in, c_out, k, stride,
                                            pad, groups, SLOPE, _lib.stream_ptr())
    if e0 is not None:
        _lib.timer.stop("vits_grouped_conv", e0, (2.0 * dy.numel() * 4 * k, 2.0 * (dy.numel() + dx.numel() * (1 + (res is not None) + (mg_src is not None))) + 2.0 * k * c_out * 4))
    _lib.check(rc, "vits_grouped_conv_dgrad")
    return dx


def grouped_wgrad(x, dy, k, stride, pad, groups, out, dbias, defer):
    """Compact dw [k][c_out][4] into `out` and the bias gradient into `dbias` (both float32), second stage deferred to `defer`.
    Returns None when the layer is not covered (4 output channels per group) or the collector has no slab space left — the caller
    then takes vits_conv1d_cl_wgrad."""
    import ctypes
    _lib.require_cuda(x, dy)
    n, t_in, c_in = x.shape
    c_out = dy.size(2)
    if c_out // groups != 16 or 4 * k > 176:
        return None
    assert x.is_contiguous() and dy.is_contiguous() and x.dtype == dy.dtype
    assert out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (k, c_out, c_in // groups)
    assert dbias.dtype == torch.float32 and dbias.is_contiguous() and dbias.numel() == c_out
    L = _lib.lib()
    ws = defer.alloc(L.vits_grouped_conv_wgrad_workspace(n, dy.size(1), c_out, k, groups))
    if ws is None:
        return None
    pend = _lib.WgradPending()
    e0 = _lib.timer.start("vits_grouped_conv")
    rc = L.vits_grouped_conv_wgrad(_DT[x.dtype], x.data_ptr(), dy.data_ptr(), out.data_ptr(), dbias.data_ptr(), ws.data_ptr(), ws.numel(),
                                   n, t_in, c_in, c_out, k, stride, pad, groups, 0, ctypes.addressof(pend), _lib.stream_ptr())
    if e0 is not None:
        _lib.timer.stop("vits_grouped_conv", e0, (2.0 * dy.numel() * 4 * k, 2.0 * (x.numel() + dy.numel()) + 4.0 * out.numel()))
    _lib.check(rc, "vits_grouped_conv_wgrad")
    defer.add(pend)
    return out


def first_wgrad(x, dy, dw, p, k, s1, pad, c_out):
    """dw: float32 [k][c_out][8] (column 0 written); returns dbias float32 [c_out]."""
    n, T = x.shape
    L = _lib.lib()
    nbytes = L.vits_disc_first_wgrad_workspace(n, T, p, k, s1, pad, c_out)
    ws = K.workspace(nbytes, x.device)
    db = torch.empty(c_out, device=x.device, dtype=torch.float32)
    assert dy.is_contiguous() and dw.dtype == torch.float32 and dw.is_contiguous()
    rc = L.vits_disc_first_wgrad(_DT[dy.dtype], x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(),
                                 n, T, p, k, s1, pad, c_out, 0, _lib.stream_ptr())
    _lib.check(rc, "vits_disc_first_wgrad")
    return db


def first_dgrad(dy, w, dx, n, n_lo, p, k, s1, pad, c_out, accumulate):
    """dx float32 [n][T]: rows n_lo.. (+)= gradient wrt the waveforms; dy holds the rows of items n_lo*p.. only."""
    T = dx.size(1)
    assert dy.is_contiguous() and dx.is_contiguous() and dx.dtype == torch.float32
    es = dy.element_size()
    # the kernel indexes dy by absolute folded item: hand it the address the full tensor would have
    dy_base = dy.data_ptr() - n_lo * p * dy.size(1) * dy.size(2) * es
    rc = _lib.lib().vits_disc_first_dgrad(_DT[dy.dtype], dy_base, w.data_ptr(), dx.data_ptr() + n_lo * T * 4, n, n_lo, T, p, k, s1, pad, c_out,
                                          1 if accumulate else 0, _lib.stream_ptr())
    _lib.check(rc, "vits_disc_first_dgrad")


def post_fwd(h, w, bias, k, pad):
    """h [J][R][c_in], w arena operand [k][8][c_in] -> y8 [J][R][8] (channel 0 live)."""
    _lib.require_cuda(h, w)
    J, R, c_in = h.shape
    assert h.is_contiguous() and w.is_contiguous() and tuple(w.shape) == (k, 8, c_in) and w.dtype == h.dtype
    y8 = torch.empty((J, R, 8), device=h.device, dtype=h.dtype)
    rc = _lib.lib().vits_disc_post_fwd(_DT[h.dtype], h.data_ptr(), w.data_ptr(), None if bias is None else bias.data_ptr(), y8.data_ptr(),
                                       J, R, c_in, k, pad, _lib.stream_ptr())
    _lib.check(rc, "vits_disc_post_fwd")
    return y8


def post_dgrad(dy8, w, res, h, j_lo, k, pad):
    """(conv^T(dy8) + res) * lrelu'(h) for the rows of items j_lo..; returns the [J - j_lo][R][c_in] tensor."""
    J, R, c_in = h.shape
    assert dy8.is_contiguous() and h.is_contiguous() and (res is None or (res.is_contiguous() and res.shape == h.shape))
    dh = torch.empty((J - j_lo, R, c_in), device=h.device, dtype=h.dtype)
    es = h.element_size()
    rc = _lib.lib().vits_disc_post_dgrad(_DT[h.dtype], dy8.data_ptr(), w.data_ptr(), None if res is None else res.data_ptr(), h.data_ptr(),
                                         dh.data_ptr() - j_lo * R * c_in * es, J, R, c_in, k, pad, j_lo, SLOPE, _lib.stream_ptr())
    _lib.check(rc, "vits_disc_post_dgrad")
    return dh


def post_wgrad(dy8, h, dw, k, pad):
    """dw float32 [k][8][c_in] (row 0 of every tap written); returns dbias float32 [1]."""
    J, R, c_in = h.shape
    L = _lib.lib()
    ws = K.workspace(L.vits_disc_post_wgrad_workspace(J, R, c_in, k), h.device)
    db = torch.empty(1, device=h.device, dtype=torch.float32)
    assert dy8.is_contiguous() and dw.dtype == torch.float32 and dw.is_contiguous()
    rc = L.vits_disc_post_wgrad(_DT[h.dtype], dy8.data_ptr(), h.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(),
                                J, R, c_in, k, pad, 0, _lib.stream_ptr())
    _lib.check(rc, "vits_disc_post_wgrad")
    return db


# ------------------------------------------------------------------------------------------------ plan + weights
class DiscPlan:
    """Static description of one discriminator (shapes only)."""

    def __init__(self, d):
        self.period = getattr(d, "period", 1)
        convs = list(d.convs)
        one = lambda v: v[0] if isinstance(v, (tuple, list)) else v
        f = convs[0]
        self.first = (one(f.kernel_size), one(f.stride), one(f.padding), f.out_channels)        # k, stride, pad, c_out
        self.mid = [(l.in_channels, l.out_channels, one(l.kernel_size), one(l.stride), one(l.padding), getattr(l, "groups", 1)) for l in convs[1:]]
        self.post = (one(d.conv_post.kernel_size), one(d.conv_post.padding), d.conv_post.in_channels)


def _dense_grouped(l):
    """Kernel-layout dense block-diagonal [k][c_out][c_in] operand of a grouped Conv1d (torch ops, autograd-connected)."""
    w = l.weight
    og, ig = w.size(0) // l.groups, w.size(1)
    w = torch.cat([torch.nn.functional.pad(w[g * og:(g + 1) * og], (0, 0, g * ig, (l.groups - 1 - g) * ig)) for g in range(l.groups)], 0)
    return w.permute(2, 0, 1).contiguous()


def prepared_weights(d):
    """[w_1, b_1, ..., w_L, b_L, w_post, b_post] in DiscFn's order: arena handles inside a weight_arena.scope, else fp32
    kernel-layout tensors made with torch ops (autograd-connected).  Also returns the `groups` each middle layer runs with."""
    from . import wn_cl
    out, groups = [], []
    convs = list(d.convs)
    out += [wn_cl.weight_of(convs[0], pad_in=7), convs[0].bias]
    for l in convs[1:]:
        g = getattr(l, "groups", 1)
        h = WA.handle_for(l)
        if h is None:
            h, g = (_dense_grouped(l), 1) if g > 1 else (wn_cl.prep_conv(l.weight), 1)
        out += [h, l.bias]
        groups.append(g)
    out += [wn_cl.weight_of(d.conv_post, pad_out=7), d.conv_post.bias]
    return out, groups


# ------------------------------------------------------------------------------------------------ the node
class DiscFn(torch.autograd.Function):
    """forward(plans, groups, dtype, n_lo, x [n][T] float32, *weights of all discriminators)
       -> for every discriminator: y8 [(n,w)][R][8], h_1 .. h_L  (flat tuple)."""

    @staticmethod
    def forward(ctx, plans, groups, dtype, n_lo, x, *wb):
        xd = x.detach().float().contiguous()
        R = [WA.resolve(t, dtype) if (t is not None and t.dim() == 3) else None for t in wb]
        bases, it = [], 0
        for plan in plans:
            bases.append(it)
            it += 2 * (len(plan.mid) + 2)
        # layer by layer ACROSS the discriminators: the same layer of the five period discriminators is five independent
        # launches of one kernel instance, each a partly filled round of workgroups — K.conv1d_cl_multi puts them side by side
        hs_all = []
        for plan, base in zip(plans, bases):
            k, s1, pad, c1 = plan.first
            hs_all.append([first_fwd(xd, R[base].fwd, _f32(wb[base + 1]), plan.period, k, s1, pad, c1, dtype)])
        for j in range(max(len(plan.mid) for plan in plans)):
            calls, owners = [], []
            for pi, (plan, grp, base) in enumerate(zip(plans, groups, bases)):
                if j >= len(plan.mid):
                    continue
                (ci, co, kk, st, pd, _), g = plan.mid[j], grp[j]
                h, iw = hs_all[pi][-1], base + 2 * (j + 1)
                if grouped_direct_ok(h, ci, co, kk, st, g):
                    hs_all[pi].append(grouped_fwd(h, R[iw].fwd, _f32(wb[iw + 1]), kk, st, pd, g))
                else:
                    calls.append((h, R[iw].fwd, dict(bias=_f32(wb[iw + 1]), pad=pd, stride=st, out_slope=SLOPE, groups=g)))
                    owners.append(pi)
            for pi, y in zip(owners, _multi(calls)):
                hs_all[pi].append(y)
        outs, saved, layout = [], [xd], []
        for plan, base, hs in zip(plans, bases, hs_all):
            ip = base + 2 * len(hs)
            pk, ppad, _ = plan.post
            y8 = post_fwd(hs[-1], R[ip].fwd, _f32(wb[ip + 1]), pk, ppad)
            outs += [y8] + hs
            layout.append((base, len(hs)))
            saved += hs
        ctx.plans, ctx.groups, ctx.dtype, ctx.n_lo, ctx.R, ctx.layout = plans, groups, dtype, n_lo, R, layout
        ctx.save_for_backward(*saved)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gout):
        WG = K.conv1d_cl_wgrad_raw
        saved = list(ctx.saved_tensors)
        xd = saved[0]
        n, T = xd.shape
        n_lo, dtype, R = ctx.n_lo, ctx.dtype, ctx.R
        need_x = ctx.needs_input_grad[4]
        grads = [None] * len(R)
        defer = K.DeferredReductions(xd.device)
        dx = torch.empty((n, T), device=xd.device, dtype=torch.float32) if need_x else None
        cont = lambda t: None if t is None else (t if t.is_contiguous() else t.contiguous())
        # ---- per discriminator: its saved feature maps, incoming gradients, conv_post
        st_all, si, gi = [], 1, 0
        for plan, grp, (base, nh) in zip(ctx.plans, ctx.groups, ctx.layout):
            hs = saved[si:si + nh]
            si += nh
            dy8, dhs = gout[gi], list(gout[gi + 1:gi + 1 + nh])
            gi += 1 + nh
            lo = n_lo * plan.period
            need_w = any(ctx.needs_input_grad[5 + base + 2 * j] for j in range(nh + 1))
            assert not (need_w and n_lo), "weight gradients need the whole batch"
            ip = base + 2 * nh
            pk, ppad, _ = plan.post
            hL = hs[-1]
            dcur = None
            if dy8 is not None:
                dy8 = cont(dy8.to(dtype))
                if need_w:
                    dw = _dw_buffer(R[ip], (pk, 8, hL.size(2)), xd.device, ctx)
                    grads[ip + 1] = post_wgrad(dy8, hL, dw, pk, ppad)
                    grads[ip] = dw
                dcur = post_dgrad(dy8, R[ip].fwd, cont(dhs[-1]), hL, lo, pk, ppad)
            elif dhs[-1] is not None:
                dcur = K.lrelu_mask_bwd(cont(dhs[-1][lo:]), cont(hL[lo:]), SLOPE)
            st_all.append(dict(plan=plan, grp=grp, base=base, nh=nh, hs=hs, dhs=dhs, lo=lo, need_w=need_w, dcur=dcur))
        # ---- middle layers, last to first, layer by layer across the discriminators (see forward)
        for li in range(max(st["nh"] for st in st_all) - 1, 0, -1):
            calls, owners = [], []
            for st in st_all:
                if li >= st["nh"]:
                    continue
                ci, co, kk, sd, pd, _ = st["plan"].mid[li - 1]
                g, lo = st["grp"][li - 1], st["lo"]
                iw = st["base"] + 2 * li
                x_in = st["hs"][li - 1][lo:]
                dprev, dcur = st["dhs"][li - 1], st["dcur"]
                if dcur is None:
                    if dprev is not None:
                        st["dcur"] = K.lrelu_mask_bwd(cont(dprev[lo:]), cont(x_in), SLOPE)
                    continue
                direct = grouped_direct_ok(dcur, ci, co, kk, sd, g)
                if st["need_w"]:
                    db = torch.empty(co, device=xd.device, dtype=torch.float32)
                    dw_out = R[iw].claim_dw(ctx)
                    grads[iw] = grouped_wgrad(cont(x_in), cont(dcur), kk, sd, pd, g, dw_out, db, defer) if direct else None
                    if grads[iw] is None:
                        grads[iw] = WG(x_in, dcur, kk, pad=pd, stride=sd, out=dw_out, dbias=db, groups=g, defer=defer)
                    grads[iw + 1] = db
                if direct:
                    st["dcur"] = grouped_dgrad(cont(dcur), R[iw].fwd, None if dprev is None else cont(dprev[lo:]), cont(x_in), x_in.size(1), kk, sd, pd, g)
                else:
                    calls.append((dcur, WA.bwd_operand(R[iw]), dict(res=None if dprev is None else cont(dprev[lo:]), mg_src=x_in, mg_slope=SLOPE,
                                                                     pad=(kk - 1) - pd, in_div=sd, t_out=x_in.size(1) if sd != 1 else None, groups=g)))
                    owners.append(st)
            for st, y in zip(owners, _multi(calls)):
                st["dcur"] = y
        # ---- first layers
        wrote_dx = False
        for st in st_all:
            dcur, plan, base = st["dcur"], st["plan"], st["base"]
            if dcur is not None:
                k, s1, pad, c1 = plan.first
                if st["need_w"]:
                    dw = _dw_buffer(R[base], (k, c1, 8), xd.device, ctx)
                    grads[base + 1] = first_wgrad(xd, dcur, dw, plan.period, k, s1, pad, c1)
                    grads[base] = dw
                if need_x:
                    first_dgrad(dcur, R[base].fwd, dx, n, n_lo, plan.period, k, s1, pad, c1, accumulate=wrote_dx)
                    wrote_dx = True
        defer.flush()
        if need_x:
            if not wrote_dx:
                dx.zero_()
            elif n_lo:
                dx[:n_lo].zero_()
        return (None, None, None, None, dx, *grads)


def _multi(calls):
    """Launches of one layer level: those that share a shape class (the period discriminators) side by side, the rest alone."""
    if not calls:
        return []
    outs = [None] * len(calls)
    classes = {}
    for i, (x, w, kw) in enumerate(calls):
        classes.setdefault((tuple(w.shape), x.size(2), kw.get("stride", 1), kw.get("in_div", 1), kw.get("groups", 1), x.dtype), []).append(i)
    for idx in classes.values():
        for lo in range(0, len(idx), 8):
            part = idx[lo:lo + 8]
            ys = K.conv1d_cl_multi([calls[i] for i in part]) if len(part) > 1 else [K.conv1d_cl_raw(calls[part[0]][0], calls[part[0]][1], **calls[part[0]][2])]
            for i, y in zip(part, ys):
                outs[i] = y
    return outs


def _f32(b):
    return None if b is None else b.detach().float()


def _dw_buffer(res, shape, device, owner=None):
    """The arena's gradient view of an edge layer (its padding channels stay zero: the edge kernels never write them), or a
    zeroed tensor outside an arena."""
    dw = res.claim_dw(owner)
    if dw is None:
        dw = torch.zeros(shape, device=device, dtype=torch.float32)
    assert tuple(dw.shape) == tuple(shape)
    return dw


def run(discs, x, n_lo=0):
    """discs: list of discriminator modules; x [n][T] float32 -> list of (y8, [h_1..h_L]) per discriminator."""
    plans, wbs, groups = [], [], []
    for d in discs:
        if getattr(d, "_plan", None) is None:
            d._plan = DiscPlan(d)
        plans.append(d._plan)
        wb, grp = prepared_weights(d)
        wbs += wb
        groups.append(tuple(grp))
    from . import wn_cl
    outs = DiscFn.apply(tuple(plans), tuple(groups), wn_cl.compute_dtype(), int(n_lo), x, *wbs)
    res, i = [], 0
    for plan in plans:
        nh = 1 + len(plan.mid)
        res.append((outs[i], list(outs[i + 1:i + 1 + nh])))
        i += 1 + nh
    return res
