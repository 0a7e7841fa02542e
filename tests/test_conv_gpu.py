"""GPU: vits_conv1d_cl (channels-last MFMA convolution) against torch's conv1d in fp32 on the same
inputs.  f32 kernel: tolerance 1e-5 relative (exact-fp32 MFMA, different summation order only);
bf16 kernel: inputs/weights rounded to bf16 first, reference computed in fp32 from the rounded
values, tolerance 1e-2 relative (bf16 output rounding)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def ref_conv(x_cl, w_tap, bias, dil, pad, in_slope):
    x = x_cl.float().transpose(1, 2)                       # [b, c, t]
    if in_slope != 1.0:
        x = F.leaky_relu(x, in_slope)
    w = w_tap.float().permute(1, 2, 0).contiguous()       # [c_out, c_in, k]
    return F.conv1d(x, w, bias, 1, pad, dil).transpose(1, 2)


def rel(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12))


CASES = [  # b, t, c_in, c_out, k, dil
    (2, 96, 32, 32, 3, 1), (2, 300, 64, 64, 7, 3), (1, 257, 128, 128, 11, 5), (3, 200, 192, 384, 5, 1),
    (2, 32, 192, 512, 7, 1), (2, 130, 256, 256, 3, 5), (1, 1000, 32, 32, 11, 1), (2, 77, 96, 192, 1, 1),
    (2, 64, 768, 192, 3, 1), (1, 50, 40, 72, 5, 2),
]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("case", CASES)
def test_plain_conv(pkg, case, dtype, tol):
    b, t, ci, co, k, dil = case
    torch.manual_seed(hash(case) % 1000)
    pad = (k - 1) * dil // 2
    x = torch.randn(b, t, ci, device=DEV).to(dtype)
    w = (torch.randn(k, co, ci, device=DEV) / (ci * k) ** 0.5).to(dtype)
    bias = torch.randn(co, device=DEV)
    y = pkg.kernels.conv1d_cl_raw(x, w, bias, dil=dil, pad=pad)
    assert rel(y, ref_conv(x, w, bias, dil, pad, 1.0)) < tol


@pytest.mark.parametrize("case", [(16, 1, 6144, 256), (16, 1, 1536, 256), (5, 1, 1032, 77), (2, 7, 2048, 130)])
def test_few_rows_long_reduction_kernel(pkg, case):
    """b * t <= 16 rows, k = 1, c_in >= 1024 (the conditioning layers' data gradient): the wave-per-output-channel kernel;
    bias, scale, strided rows, reproducible."""
    b, t, ci, co = case
    torch.manual_seed(ci)
    xw = torch.randn(b, t, ci + 8, device=DEV).bfloat16()
    x = xw[:, :, :ci]                                            # row pitch ci + 8
    w = (torch.randn(1, co, ci, device=DEV) / ci ** 0.5).bfloat16()
    bias = torch.randn(co, device=DEV)
    y = pkg.kernels.conv1d_cl_raw(x, w, bias, out_scale=0.5)
    want = ref_conv(x, w, bias, 1, 0, 1.0) * 0.5
    assert rel(y, want) < 1e-2
    assert torch.equal(y, pkg.kernels.conv1d_cl_raw(x, w, bias, out_scale=0.5))
    y0 = pkg.kernels.conv1d_cl_raw(x, w)
    assert rel(y0, ref_conv(x, w, None, 1, 0, 1.0)) < 1e-2


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1.5e-2)])
def test_fused_prologue_epilogue(pkg, dtype, tol):
    torch.manual_seed(3)
    K = pkg.kernels
    b, t, c, k, dil = 3, 210, 64, 7, 3
    pad = (k - 1) * dil // 2
    x = torch.randn(b, t, c, device=DEV).to(dtype)
    w = (torch.randn(k, c, c, device=DEV) / (c * k) ** 0.5).to(dtype)
    bias, bias_b = torch.randn(c, device=DEV), torch.randn(b, c, device=DEV)
    res = torch.randn(b, t, c, device=DEV).to(dtype)
    lens = torch.tensor([210, 150, 33], device=DEV, dtype=torch.int32)
    mask = (torch.arange(t, device=DEV)[None, :] < lens[:, None]).float().unsqueeze(-1)
    # lrelu prologue + masked input + bias + per-item bias + residual + scale + masked output
    y = K.conv1d_cl_raw(x, w, bias, bias_b, res=res, lengths=lens, dil=dil, pad=pad, in_slope=0.1, out_scale=0.5,
                        flags=K.CONV_MASK_IN | K.CONV_MASK_OUT)
    want = (ref_conv((x.float() * mask).to(dtype), w, bias, dil, pad, 0.1) + bias_b[:, None, :] + res.float()) * 0.5 * mask
    assert rel(y, want) < tol
    # accumulate + tanh
    y0 = torch.randn(b, t, c, device=DEV).to(dtype)
    y1 = y0.clone()
    K.conv1d_cl_raw(x, w, None, out=y1, dil=dil, pad=pad, flags=K.CONV_ACCUM | K.CONV_TANH)
    assert rel(y1, torch.tanh(ref_conv(x, w, None, dil, pad, 1.0)) + y0.float()) < tol
    # chain-rule multiplier of a fused leaky-relu (data-gradient form)
    src = torch.randn(b, t, c, device=DEV).to(dtype)
    y2 = K.conv1d_cl_raw(x, w, None, mg_src=src, dil=dil, pad=pad, mg_slope=0.1)
    assert rel(y2, ref_conv(x, w, None, dil, pad, 1.0) * torch.where(src.float() > 0, 1.0, 0.1)) < tol


def test_data_gradient_identity(pkg):
    """dX of conv(x, w) == the same kernel on dY with tap-flipped, transposed weights and
    pad' = dil*(k-1) - pad (include/vitsmi.h)."""
    torch.manual_seed(5)
    b, t, ci, co, k, dil = 2, 140, 64, 96, 5, 2
    pad = (k - 1) * dil // 2
    x = torch.randn(b, t, ci, device=DEV, requires_grad=True)
    w = torch.randn(k, co, ci, device=DEV) / (ci * k) ** 0.5
    y = ref_conv(x, w, None, dil, pad, 1.0)
    dy = torch.randn_like(y)
    (dx_ref,) = torch.autograd.grad(y, x, dy)
    w_t = w.flip(0).transpose(1, 2).contiguous()               # [k][ci][co]
    dx = pkg.kernels.conv1d_cl_raw(dy.contiguous(), w_t, None, dil=dil, pad=dil * (k - 1) - pad)
    assert rel(dx, dx_ref) < 2e-5


def test_rejects_unsupported(pkg):
    x = torch.zeros(1, 8, 6, device=DEV)
    w = torch.zeros(1, 4, 6, device=DEV)
    with pytest.raises(pkg._lib.VitsKernelError, match="UNSUPPORTED"):
        pkg.kernels.conv1d_cl_raw(x, w)                          # c_in % 4 != 0


WG_CASES = [  # b, t, c_in, c_out, k, dil
    (2, 96, 32, 32, 3, 1), (2, 300, 64, 64, 7, 3), (1, 257, 128, 128, 11, 5), (3, 200, 192, 384, 5, 1),
    (4, 2048, 32, 32, 11, 1), (2, 77, 96, 192, 1, 1), (2, 64, 768, 192, 3, 1), (1, 50, 40, 72, 5, 2),
]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("case", WG_CASES)
def test_weight_gradient(pkg, case, dtype, tol):
    b, t, ci, co, k, dil = case
    torch.manual_seed(hash(case) % 1000)
    pad = (k - 1) * dil // 2
    x = torch.randn(b, t, ci, device=DEV).to(dtype)
    dy = torch.randn(b, t, co, device=DEV).to(dtype)
    lens = torch.randint(t // 2, t + 1, (b,), device=DEV, dtype=torch.int32)
    mask = (torch.arange(t, device=DEV)[None, :] < lens[:, None]).float().unsqueeze(-1)
    w = torch.zeros(k, co, ci, device=DEV, requires_grad=True)
    K = pkg.kernels
    # plain
    y = ref_conv(x, w, None, dil, pad, 0.1)
    (want,) = torch.autograd.grad(y, w, dy.float())
    db = torch.empty(co, device=DEV) if co % 8 == 0 else None
    got = K.conv1d_cl_wgrad_raw(x, dy, k, dil=dil, pad=pad, in_slope=0.1, dbias=db)
    assert rel(got, want) < tol
    if db is not None:                                   # bias gradient from the same launch
        assert rel(db, dy.float().sum((0, 1))) < 1e-5
    # masked input and output rows, accumulated onto an existing gradient
    y = ref_conv((x.float() * mask).to(dtype), w, None, dil, pad, 1.0) * mask
    (want,) = torch.autograd.grad(y, w, dy.float())
    base = torch.randn(k, co, ci, device=DEV)
    got = base.clone()
    K.conv1d_cl_wgrad_raw(x, dy, k, lengths=lens, dil=dil, pad=pad, flags=K.CONV_MASK_IN | K.CONV_MASK_OUT | K.CONV_ACCUM, out=got)
    assert rel(got - base, want) < tol * 2


def test_weight_gradient_is_reproducible(pkg):
    torch.manual_seed(1)
    x = torch.randn(4, 1000, 64, device=DEV).bfloat16()
    dy = torch.randn(4, 1000, 64, device=DEV).bfloat16()
    a = pkg.kernels.conv1d_cl_wgrad_raw(x, dy, 7, dil=1, pad=3).clone()
    b = pkg.kernels.conv1d_cl_wgrad_raw(x, dy, 7, dil=1, pad=3)
    assert torch.equal(a, b)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-5), (torch.bfloat16, 1.5e-2)])
def test_emulation_agrees_on_new_modes(pkg, dtype, tol):
    """Gate / gate-backward epilogues, channel-sliced operands (row pitches) and time stride: the HIP kernel
    against tests/cl_emul.py (the torch statement of include/vitsmi.h used by the CPU logic tests)."""
    import cl_emul
    K = pkg.kernels
    torch.manual_seed(11)
    b, t, H, k = 3, 150, 192, 5
    lens = torch.tensor([150, 99, 17], device=DEV, dtype=torch.int32)
    wide = torch.randn(b, t, 2 * H, device=DEV).to(dtype)
    x = wide[..., H:]                                              # channel slice: ldx = 2H
    w = (torch.randn(k, 2 * H, H, device=DEV) / (H * k) ** 0.5).to(dtype)
    bias, bias_b = torch.randn(2 * H, device=DEV), torch.randn(b, 2 * H, device=DEV)
    kw = dict(bias=bias, bias_b=bias_b, pad=2, flags=K.CONV_GATE, gate_h=H)
    pre_a = torch.empty(b, t, 2 * H, device=DEV, dtype=dtype); pre_b = torch.empty_like(pre_a)
    ya = K.conv1d_cl_raw(x, w, out2=pre_a, **kw)
    yb = cl_emul.conv1d_cl_raw(x, w, out2=pre_b, **kw)
    assert rel(ya, yb) < tol and rel(pre_a, pre_b) < tol
    # gate backward: 1x1 conv producing d(acts) from a 2H-wide gradient, chain rule through the gate
    w1 = (torch.randn(1, H, 2 * H, device=DEV) / (2 * H) ** 0.5).to(dtype)
    d_rs = torch.randn(b, t, 2 * H, device=DEV).to(dtype)
    kw = dict(mg_src=pre_b, lengths=lens, flags=K.CONV_GATE_BWD | K.CONV_MASK_OUT, gate_h=H)
    assert rel(K.conv1d_cl_raw(d_rs, w1, **kw), cl_emul.conv1d_cl_raw(d_rs, w1, **kw)) < tol
    # output into a channel slice of a wider tensor, residual read from the same slice (in place)
    buf_a = torch.randn(b, t, 2 * H, device=DEV).to(dtype); buf_b = buf_a.clone()
    w2 = (torch.randn(3, H, H, device=DEV) / (3 * H) ** 0.5).to(dtype)
    src = torch.randn(b, t, H, device=DEV).to(dtype)
    K.conv1d_cl_raw(src, w2, res=buf_a[..., :H], out=buf_a[..., :H], pad=1, flags=K.CONV_RES_AFTER)
    cl_emul.conv1d_cl_raw(src, w2, res=buf_b[..., :H], out=buf_b[..., :H], pad=1, flags=K.CONV_RES_AFTER)
    assert rel(buf_a, buf_b) < tol
    # time stride (discriminator-style): k=5 stride 3 and k=41 stride 4
    for (kk, st, pd, ci, co) in [(5, 3, 2, 32, 128), (41, 4, 20, 64, 64)]:
        xs = torch.randn(2, 500, ci, device=DEV).to(dtype)
        ws = (torch.randn(kk, co, ci, device=DEV) / (ci * kk) ** 0.5).to(dtype)
        ya, yb = K.conv1d_cl_raw(xs, ws, pad=pd, stride=st, in_slope=0.1), cl_emul.conv1d_cl_raw(xs, ws, pad=pd, stride=st, in_slope=0.1)
        assert ya.shape == yb.shape and rel(ya, yb) < tol
        if kk == 5:
            dy = torch.randn_like(ya)
            ga = K.conv1d_cl_wgrad_raw(xs, dy, kk, pad=pd, stride=st, in_slope=0.1)
            gb = cl_emul.conv1d_cl_wgrad_raw(xs, dy, kk, pad=pd, stride=st, in_slope=0.1)
            assert rel(ga, gb) < tol
    # weight gradient with channel-sliced x and dy
    dyw = torch.randn(b, t, 2 * H, device=DEV).to(dtype)
    ga = K.conv1d_cl_wgrad_raw(wide[..., :H], dyw[..., H:], 1)
    gb = cl_emul.conv1d_cl_wgrad_raw(wide[..., :H], dyw[..., H:], 1)
    assert rel(ga, gb) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-5), (torch.bfloat16, 1.5e-2)])
def test_flat_row_kernel_and_strided_data_gradient(pkg, dtype, tol):
    """csrc/conv1d_flat.hip: rows enumerate (item, time) pairs (period-discriminator shapes: many short items, stride 3)
    and the data gradient of a strided convolution walks dY phase by phase (in_div) instead of zero-insertion."""
    import cl_emul
    K = pkg.kernels
    torch.manual_seed(5)
    for (b, t, ci, co, kk, st, pd) in [(48, 57, 32, 128, 5, 3, 2), (33, 19, 128, 512, 5, 3, 2), (80, 7, 1024, 1024, 5, 1, 2),
                                       (64, 51, 256, 320, 5, 1, 2), (96, 102, 128, 512, 5, 3, 2),      # >= 2048 rows: 64x64-per-wave variant
                                       (16, 40, 64, 32, 3, 1, 1), (5, 301, 32, 96, 41, 4, 20)]:
        x = torch.randn(b, t, ci, device=DEV).to(dtype)
        w = (torch.randn(kk, co, ci, device=DEV) / (ci * kk) ** 0.5).to(dtype)
        bias = torch.randn(co, device=DEV)
        lens = torch.randint(1, t + 1, (b,), device=DEV, dtype=torch.int32)
        kw = dict(bias=bias, pad=pd, stride=st, in_slope=0.1, out_slope=0.2, lengths=lens, flags=K.CONV_MASK_IN | K.CONV_FLAT)
        kwe = dict(kw, flags=K.CONV_MASK_IN)
        ya, yb = K.conv1d_cl_raw(x, w, **kw), cl_emul.conv1d_cl_raw(x, w, **kwe)
        assert ya.shape == yb.shape and rel(ya, yb) < tol, (b, t, ci, co, kk, st)
        # residual + accumulate epilogues through the flat kernel
        r = torch.randn_like(ya)
        oa, ob = ya.clone(), yb.clone()
        K.conv1d_cl_raw(x, w, res=r, out=oa, pad=pd, stride=st, flags=K.CONV_ACCUM | K.CONV_FLAT, out_scale=0.5)
        cl_emul.conv1d_cl_raw(x, w, res=r, out=ob, pad=pd, stride=st, flags=K.CONV_ACCUM, out_scale=0.5)
        assert rel(oa, ob) < tol
        # data gradient: dx = conv(dy, flipped w^T) on the input grid, divided index
        dy = torch.randn_like(ya)
        wb = w.flip(0).transpose(1, 2).contiguous()
        xf = x.float().requires_grad_(True)
        yr = torch.nn.functional.conv1d(xf.transpose(1, 2), w.float().permute(1, 2, 0), None, st, pd).transpose(1, 2)
        (dx_ref,) = torch.autograd.grad(yr, xf, dy.float())
        dxa = K.conv1d_cl_raw(dy, wb, pad=kk - 1 - pd, in_div=st, t_out=t) if st > 1 else K.conv1d_cl_raw(dy, wb, pad=kk - 1 - pd, flags=K.CONV_FLAT)
        assert dxa.shape == dx_ref.shape and rel(dxa, dx_ref) < tol, ("dgrad", b, t, ci, co, kk, st)
        if st > 1:
            dxe = cl_emul.conv1d_cl_raw(dy, wb, pad=kk - 1 - pd, in_div=st, t_out=t)
            assert rel(dxe, dx_ref) < tol
        # weight + bias gradient, flat-row variant (masked rows on both sides)
        if kk <= 5:
            dba, dbb = torch.empty(co, device=DEV), torch.empty(co, device=DEV)
            fl = K.CONV_MASK_IN | (K.CONV_MASK_OUT if st == 1 else 0)
            ga = K.conv1d_cl_wgrad_raw(x, dy, kk, lengths=lens, pad=pd, stride=st, in_slope=0.1, flags=fl | K.CONV_FLAT, dbias=dba)
            gb = cl_emul.conv1d_cl_wgrad_raw(x, dy, kk, lengths=lens, pad=pd, stride=st, in_slope=0.1, flags=fl, dbias=dbb)
            assert rel(ga, gb) < tol and rel(dba, dbb) < tol, ("wgrad", b, t, ci, co, kk, st)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-5), (torch.bfloat16, 1.5e-2)])
def test_grouped_convolution_on_dense_operands(pkg, dtype, tol):
    """DiscriminatorS' grouped layers (reference models.py:343-349): dense block-diagonal operands, the kernel walks only
    the input channels a tile of output channels can see; forward, strided data gradient, compact weight gradient."""
    import cl_emul
    K = pkg.kernels
    torch.manual_seed(9)
    for (b, t, ci, co, groups, kk, st, pd) in [(3, 700, 16, 64, 4, 41, 4, 20), (3, 300, 64, 256, 16, 41, 4, 20),
                                               (2, 200, 256, 1024, 64, 41, 4, 20), (4, 90, 1024, 1024, 256, 41, 4, 20)]:
        ig, og = ci // groups, co // groups
        wg = torch.randn(co, ig, kk, device=DEV) / (ig * kk) ** 0.5                 # torch grouped layout
        dense = torch.zeros(co, ci, kk, device=DEV)
        for g in range(groups):
            dense[g * og:(g + 1) * og, g * ig:(g + 1) * ig] = wg[g * og:(g + 1) * og]
        w = dense.permute(2, 0, 1).contiguous().to(dtype)
        x = torch.randn(b, t, ci, device=DEV).to(dtype)
        bias = torch.randn(co, device=DEV)
        ya = K.conv1d_cl_raw(x, w, bias, pad=pd, stride=st, out_slope=0.1, groups=groups)
        yr = torch.nn.functional.leaky_relu(torch.nn.functional.conv1d(x.float().transpose(1, 2), wg.to(dtype).float(), bias, st, pd, 1, groups), 0.1).transpose(1, 2)
        assert ya.shape == yr.shape and rel(ya, yr) < tol, ("fwd", ci, co, groups)
        dy = torch.randn_like(ya)
        wb = w.flip(0).transpose(1, 2).contiguous()
        xf = x.float().requires_grad_(True)
        wf = wg.to(dtype).float().requires_grad_(True)
        yq = torch.nn.functional.conv1d(xf.transpose(1, 2), wf, None, st, pd, 1, groups).transpose(1, 2)
        dx_ref, dw_ref = torch.autograd.grad(yq, (xf, wf), dy.float())
        dxa = K.conv1d_cl_raw(dy, wb, pad=kk - 1 - pd, in_div=st, t_out=t, groups=groups)
        assert rel(dxa, dx_ref) < tol, ("dgrad", ci, co, groups)
        db = torch.empty(co, device=DEV)
        ga = K.conv1d_cl_wgrad_raw(x, dy, kk, pad=pd, stride=st, dbias=db, groups=groups)          # [k][co][ig]
        assert ga.shape == (kk, co, ig) and rel(ga, dw_ref.permute(2, 0, 1)) < tol, ("wgrad", ci, co, groups)
        assert rel(db, dy.float().sum((0, 1))) < tol
        ge = cl_emul.conv1d_cl_wgrad_raw(x, dy, kk, pad=pd, stride=st, groups=groups)
        assert rel(ge, dw_ref.permute(2, 0, 1)) < tol


def test_wgrad_fused_reduction_is_bitwise_the_two_launch_result(pkg):
    """The last-split-sums-the-slabs path must give exactly the bits of the separate reduce_slabs launch (same fixed order),
    repeatedly (the per-tile counters re-arm themselves), with and without accumulation and bias gradient."""
    K = pkg.kernels
    torch.manual_seed(12)
    for (b, t, ci, co, kk, st, pd, dt) in [(16, 500, 192, 384, 5, 1, 2, torch.bfloat16), (16, 500, 192, 192, 1, 1, 0, torch.bfloat16),
                                           (64, 51, 256, 320, 5, 1, 2, torch.bfloat16), (4, 300, 64, 96, 3, 1, 1, torch.float32),      # (>= 512 channels: another kernel)
                                           (16, 2048, 32, 32, 7, 1, 3, torch.bfloat16)]:
        x = torch.randn(b, t, ci, device=DEV).to(dt)
        dy = torch.randn(b, (t + 2 * pd - kk) // st + 1, co, device=DEV).to(dt)
        res = {}
        for fused in (False, True, True):
            K.FUSED_WGRAD_REDUCE = fused
            try:
                db = torch.empty(co, device=DEV)
                dw = K.conv1d_cl_wgrad_raw(x, dy, kk, pad=pd, stride=st, dbias=db)
                acc = dw.clone(); dba = db.clone()
                K.conv1d_cl_wgrad_raw(x, dy, kk, pad=pd, stride=st, dbias=dba, out=acc, flags=K.CONV_ACCUM)
            finally:
                K.FUSED_WGRAD_REDUCE = False
            if fused in res:
                assert all(torch.equal(u, v) for u, v in zip(res[fused], (dw, db, acc, dba)))
            res[fused] = (dw, db, acc, dba)
        assert all(torch.equal(u, v) for u, v in zip(res[False], res[True])), (b, t, ci, co, kk)


def test_ring_kernel_matches_emulation(pkg):
    """csrc/conv1d_ring.hip (LDS-DMA ring, bf16, c_in % 64 == 0, k >= 2): tiles that span many short items (halo rows come
    from the buffer descriptor's zero fill), strides, dilations, ragged masked inputs, partial last tiles and column tiles,
    and every epilogue; plus the data-gradient form with residual and lrelu' multiplier."""
    import cl_emul
    K = pkg.kernels
    torch.manual_seed(11)
    dtype, tol = torch.bfloat16, 1.5e-2
    cases = [  # b, t, c_in, c_out, k, stride, dil, pad
        (80, 7, 1024, 1024, 5, 1, 1, 2), (352, 10, 1024, 1024, 5, 1, 1, 2), (33, 19, 128, 512, 5, 3, 1, 2), (64, 51, 256, 320, 5, 1, 1, 2),
        (96, 102, 128, 512, 5, 3, 1, 2), (3, 700, 192, 384, 5, 1, 1, 2), (2, 2048, 128, 128, 11, 1, 5, 25), (16, 256, 256, 256, 7, 1, 3, 9),
        (5, 301, 64, 96, 3, 1, 1, 1), (1, 130, 512, 70, 2, 1, 1, 0), (7, 33, 320, 200, 41, 4, 1, 20), (32, 32, 1024, 1024, 5, 1, 1, 2),
    ]
    for (b, t, ci, co, kk, st, dl, pd) in cases:
        x = torch.randn(b, t, ci, device=DEV).to(dtype)
        w = (torch.randn(kk, co, ci, device=DEV) / (ci * kk) ** 0.5).to(dtype)
        bias = torch.randn(co, device=DEV)
        lens = torch.randint(1, t + 1, (b,), device=DEV, dtype=torch.int32)
        kw = dict(bias=bias, pad=pd, stride=st, dil=dl, out_slope=0.2, lengths=lens, flags=K.CONV_MASK_IN)
        ya, yb = K.conv1d_cl_raw(x, w, **kw), cl_emul.conv1d_cl_raw(x, w, **kw)
        assert ya.shape == yb.shape and rel(ya, yb) < tol, ("fwd", b, t, ci, co, kk, st, dl)
        kwi = dict(kw, in_slope=0.1)
        assert rel(K.conv1d_cl_raw(x, w, **kwi), cl_emul.conv1d_cl_raw(x, w, **kwi)) < tol, ("in_slope", b, t, ci, co, kk, st, dl)
        r, mg = torch.randn_like(ya), torch.randn_like(ya)
        bias_b = torch.randn(b, co, device=DEV)
        kw2 = dict(bias_b=bias_b, res=r, mg_src=mg, mg_slope=0.1, pad=pd, stride=st, dil=dl, out_scale=0.5, lengths=lens, flags=K.CONV_MASK_OUT)
        assert rel(K.conv1d_cl_raw(x, w, **kw2), cl_emul.conv1d_cl_raw(x, w, **kw2)) < tol, ("epilogue", b, t, ci, co, kk, st, dl)
        oa, ob = ya.clone(), yb.clone()
        K.conv1d_cl_raw(x, w, res=r, out=oa, pad=pd, stride=st, dil=dl, flags=K.CONV_ACCUM | K.CONV_RES_AFTER | K.CONV_TANH)
        cl_emul.conv1d_cl_raw(x, w, res=r, out=ob, pad=pd, stride=st, dil=dl, flags=K.CONV_ACCUM | K.CONV_RES_AFTER | K.CONV_TANH)
        assert rel(oa, ob) < tol, ("accum", b, t, ci, co, kk, st, dl)
        kw3 = dict(bias=bias, res=r, mg_src=mg, mg_slope=0.1, pad=pd, stride=st, dil=dl, lengths=lens, flags=K.CONV_MASK_OUT | K.CONV_RES_AFTER)
        assert rel(K.conv1d_cl_raw(x, w, **kw3), cl_emul.conv1d_cl_raw(x, w, **kw3)) < tol, ("res_after", b, t, ci, co, kk, st, dl)
        kw4 = dict(res=r, mg_src=mg, mg_slope=0.1, pad=pd, stride=st, dil=dl)                # the discriminators' data-gradient form
        assert rel(K.conv1d_cl_raw(x, w, **kw4), cl_emul.conv1d_cl_raw(x, w, **kw4)) < tol, ("dgrad form", b, t, ci, co, kk, st, dl)
        assert torch.equal(K.conv1d_cl_raw(x, w, **kw), ya)                      # reproducible
    # data gradient of a strided convolution (in_div): dY on the short grid, dX on the long one, phase by phase; odd lengths
    # (the last phase has one row fewer), k < 2 * stride (phases with a single tap), with and without residual / lrelu' multiplier
    for (b, t_dy, ci, co, kk, st, pd, t_x) in [(96, 34, 512, 128, 5, 3, 2, 102), (33, 12, 1024, 512, 5, 3, 2, 34), (44, 13, 1024, 512, 5, 3, 2, 37),
                                                 (6, 90, 256, 256, 41, 4, 20, 358), (64, 23, 128, 128, 5, 3, 2, 69), (40, 16, 192, 96, 4, 2, 1, 33)]:
        dy = torch.randn(b, t_dy, ci, device=DEV).to(dtype)
        w = (torch.randn(kk, co, ci, device=DEV) / (ci * kk) ** 0.5).to(dtype)
        r, mg = torch.randn(b, t_x, co, device=DEV).to(dtype), torch.randn(b, t_x, co, device=DEV).to(dtype)
        for kw in (dict(), dict(res=r, mg_src=mg, mg_slope=0.1), dict(res=r, mg_src=mg, mg_slope=0.1, flags=K.CONV_RES_AFTER, out_scale=0.5)):
            kw = dict(kw, pad=kk - 1 - pd, in_div=st, t_out=t_x)
            ya, yb = K.conv1d_cl_raw(dy, w, **kw), cl_emul.conv1d_cl_raw(dy, w, **kw)
            assert ya.shape == yb.shape and rel(ya, yb) < tol, ("in_div", b, t_dy, ci, co, kk, st, sorted(kw))


def test_large_tile_wgrad_kernel_matches_emulation(pkg):
    """csrc/conv1d_wgrad_ring.hip (bf16, >= 64 channels, k >= 2, >= 1024 flat rows, no masks): the discriminator shapes (many
    short items, stride 3 / 1, 128-1024 channels), the decoder shapes (long items, dilations 1/3/5, fused input leaky-relu),
    partial channel tiles, a last stage that is mostly empty, accumulate mode and the bias gradient."""
    import cl_emul
    K = pkg.kernels
    torch.manual_seed(21)
    dtype, tol = torch.bfloat16, 1.5e-2
    cases = [  # b, t, c_in, c_out, k, stride, dil, pad, in_slope
        (352, 10, 1024, 1024, 5, 1, 1, 2, 1.0), (96, 34, 1024, 1024, 5, 1, 1, 2, 1.0), (160, 61, 512, 1024, 5, 3, 1, 2, 1.0),
        (96, 304, 128, 512, 5, 3, 1, 2, 1.0), (16, 2048, 128, 128, 11, 1, 5, 25, 0.1), (16, 256, 256, 256, 7, 1, 3, 9, 0.1),
        (16, 4096, 64, 64, 11, 1, 1, 5, 0.1), (16, 2048, 128, 128, 3, 1, 1, 1, 0.1), (16, 201, 768, 192, 3, 1, 1, 1, 1.0),
        (7, 333, 200, 136, 5, 1, 2, 4, 0.1), (3, 1000, 64, 320, 2, 1, 1, 0, 1.0), (40, 33, 256, 128, 4, 2, 1, 1, 1.0),
    ]
    for (b, t, ci, co, kk, st, dl, pd, sl) in cases:
        x = torch.randn(b, t, ci, device=DEV).to(dtype)
        t_out = (t + 2 * pd - dl * (kk - 1) - 1) // st + 1
        dy = torch.randn(b, t_out, co, device=DEV).to(dtype)
        dba, dbb = torch.empty(co, device=DEV), torch.empty(co, device=DEV)
        big = K.CONV_BIG_TILES                       # (the default rule takes this kernel for the 1024-channel layers only)
        ga = K.conv1d_cl_wgrad_raw(x, dy, kk, pad=pd, stride=st, dil=dl, in_slope=sl, dbias=dba, flags=big)
        gb = cl_emul.conv1d_cl_wgrad_raw(x, dy, kk, pad=pd, stride=st, dil=dl, in_slope=sl, dbias=dbb)
        assert ga.shape == gb.shape and rel(ga, gb) < tol and rel(dba, dbb) < tol, (b, t, ci, co, kk, st, dl, rel(ga, gb), rel(dba, dbb))
        assert torch.equal(K.conv1d_cl_wgrad_raw(x, dy, kk, pad=pd, stride=st, dil=dl, in_slope=sl, flags=big), ga)      # reproducible
        acc_a, acc_b = ga.clone(), gb.clone()
        K.conv1d_cl_wgrad_raw(x, dy, kk, pad=pd, stride=st, dil=dl, in_slope=sl, out=acc_a, flags=K.CONV_ACCUM | big)
        cl_emul.conv1d_cl_wgrad_raw(x, dy, kk, pad=pd, stride=st, dil=dl, in_slope=sl, out=acc_b, flags=K.CONV_ACCUM)
        assert rel(acc_a, acc_b) < tol, ("accum", b, t, ci, co, kk)


@pytest.mark.parametrize("case", [(3, 203, 4, 16, 41, 4, 20), (2, 64, 8, 4, 41, 4, 20), (2, 1030, 16, 16, 41, 4, 20), (1, 45, 4, 4, 7, 3, 2), (2, 130, 8, 16, 5, 1, 2)])
def test_grouped_direct_kernels_match_torch(pkg, case):
    """csrc/grouped.hip (DiscriminatorS' grouped layers as direct kernels) against torch's grouped conv1d in fp32 on the
    bf16-rounded operands: forward (bias + leaky-relu) and data gradient ((conv^T(dy) + res) * lrelu'(mg))."""
    n, t, groups, og, k, stride, pad = case
    from importlib import import_module
    D = import_module(pkg.__name__ + ".disc_cl")
    torch.manual_seed(t)
    ci, co = 4 * groups, og * groups
    x = torch.randn(n, t, ci, device=DEV).bfloat16()
    wc = (torch.randn(co, 4, k, device=DEV) / (4 * k) ** 0.5).bfloat16()          # torch layout [c_out][c_in / groups][k]
    bias = torch.randn(co, device=DEV)
    dense = torch.zeros(k, co, ci, device=DEV, dtype=torch.bfloat16)
    for g in range(groups):
        dense[:, g * og:(g + 1) * og, g * 4:(g + 1) * 4] = wc[g * og:(g + 1) * og].permute(2, 0, 1)
    assert D.grouped_direct_ok(x, ci, co, k, stride, groups)
    y = D.grouped_fwd(x, dense, bias, k, stride, pad, groups)
    xr = x.float().transpose(1, 2).requires_grad_(True)
    pre = F.conv1d(xr, wc.float(), bias, stride, pad, 1, groups)
    want = F.leaky_relu(pre, D.SLOPE).transpose(1, 2).detach()
    assert tuple(y.shape) == tuple(want.shape) and rel(y, want) < 1e-2
    assert torch.equal(y, D.grouped_fwd(x, dense, bias, k, stride, pad, groups))
    # the same layer through the matrix-core path
    y_mc = pkg.kernels.conv1d_cl_raw(x, dense, bias, pad=pad, stride=stride, out_slope=D.SLOPE, groups=groups)
    assert rel(y, y_mc) < 1.5e-2
    dy = torch.randn_like(y)
    res = torch.randn(n, t, ci, device=DEV).bfloat16()
    mg = torch.randn(n, t, ci, device=DEV).bfloat16()
    (gx,) = torch.autograd.grad(pre, xr, dy.float().transpose(1, 2))
    want_dx = (gx.transpose(1, 2) + res.float()) * torch.where(mg.float() > 0, 1.0, D.SLOPE)
    dx = D.grouped_dgrad(dy, dense, res, mg, t, k, stride, pad, groups)
    assert rel(dx, want_dx) < 1e-2
    dx0 = D.grouped_dgrad(dy, dense, None, None, t, k, stride, pad, groups)
    assert rel(dx0, gx.transpose(1, 2)) < 1e-2
    # weight and bias gradient (compact [k][c_out][4]) through the deferred second stage (16 output channels per group only)
    defer = pkg.kernels.DeferredReductions(x.device)
    out, db = torch.empty(k, co, 4, device=DEV), torch.empty(co, device=DEV)
    got = D.grouped_wgrad(x, dy, k, stride, pad, groups, out, db, defer)
    if og != 16:
        assert got is None
        return
    assert got is out
    defer.flush()
    wf = wc.float().requires_grad_(True)
    pre2 = F.conv1d(x.float().transpose(1, 2), wf, bias, stride, pad, 1, groups)
    (gw,) = torch.autograd.grad(pre2, wf, dy.float().transpose(1, 2))
    assert rel(out, gw.permute(2, 0, 1)) < 2e-3
    assert rel(db, dy.float().sum((0, 1))) < 2e-3
    out2, db2 = torch.empty_like(out), torch.empty_like(db)
    D.grouped_wgrad(x, dy, k, stride, pad, groups, out2, db2, defer)
    defer.flush()
    assert torch.equal(out, out2) and torch.equal(db, db2)


def test_multi_launch_is_bitwise_the_separate_launches(pkg):
    """vits_conv1d_cl_multi (several problems of one kernel instance side by side: the same layer of the period discriminators,
    models.py:299-335) against one vits_conv1d_cl per problem: forward (strided, bias, leaky-relu out) and the strided data
    gradient (residual + lrelu' multiplier), different row counts and weights per problem; and a mixed batch that the library
    must refuse as a whole (kernels.conv1d_cl_multi then launches one by one: same results)."""
    K = pkg.kernels
    torch.manual_seed(5)
    dt = torch.bfloat16
    geo = [(64, 152), (96, 102), (160, 61), (224, 44), (352, 28)]          # (items, rows per item) of the 512 -> 1024 layer's inputs
    calls_f, calls_b = [], []
    for b, t in geo:
        x = torch.randn(b, t, 512, device=DEV).to(dt)
        w = (torch.randn(5, 1024, 512, device=DEV) / 50).to(dt)
        bias = torch.randn(1024, device=DEV)
        calls_f.append((x, w, dict(bias=bias, pad=2, stride=3, out_slope=0.1)))
        t_out = (t + 4 - 5) // 3 + 1
        dy = torch.randn(b, t_out, 1024, device=DEV).to(dt)
        wt = (torch.randn(5, 512, 1024, device=DEV) / 70).to(dt)
        res = torch.randn(b, t, 512, device=DEV).to(dt)
        calls_b.append((dy, wt, dict(res=res, mg_src=x, mg_slope=0.1, pad=2, in_div=3, t_out=t)))
    for calls in (calls_f, calls_b):
        was, K.MULTI_LAUNCH = K.MULTI_LAUNCH, True
        try:
            P = pkg._lib
            n0 = P.timer.enabled
            ys = K.conv1d_cl_multi(calls)
        finally:
            K.MULTI_LAUNCH = was
        refs = [K.conv1d_cl_raw(x, w, **kw) for x, w, kw in calls]
        for y, r in zip(ys, refs):
            assert torch.equal(y, r)
    # the library takes the five problems as ONE launch (not the fall-back)
    import ctypes
    built = [K._conv_desc(x, w, **kw) for x, w, kw in calls_f]
    arr = (pkg._lib.ConvDesc * 5)(*[b[0] for b in built])
    assert pkg._lib.lib().vits_conv1d_cl_multi(ctypes.addressof(arr), 5, pkg._lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    # a batch with a problem of another class is refused as a whole ...
    x = torch.randn(4, 100, 32, device=DEV).to(dt)
    w = (torch.randn(5, 128, 32, device=DEV) / 10).to(dt)
    mixed = calls_f[:2] + [(x, w, dict(pad=2, stride=3, out_slope=0.1))]
    built = [K._conv_desc(a, b, **kw) for a, b, kw in mixed]
    arr = (pkg._lib.ConvDesc * 3)(*[b[0] for b in built])
    assert pkg._lib.lib().vits_conv1d_cl_multi(ctypes.addressof(arr), 3, pkg._lib.stream_ptr()) == pkg._lib.E_UNSUPPORTED
    # ... and the Python entry then launches one by one
    ys = K.conv1d_cl_multi(mixed)
    for y, (a, b, kw) in zip(ys, mixed):
        assert torch.equal(y, K.conv1d_cl_raw(a, b, **kw))
