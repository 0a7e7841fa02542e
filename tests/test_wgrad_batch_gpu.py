"""GPU: vits_conv1d_cl_wgrad_batch (csrc/conv1d_wgrad_batch.hip) — the weight / bias gradients of a group of convolutions in one
launch — against the per-layer kernel it replaces (vits_conv1d_cl_wgrad, itself pinned to torch in tests/test_conv_gpu.py) and
against torch's conv1d weight gradient in fp32."""
import pytest
import torch
import torch.nn.functional as F

from model_util import rel_err

pytestmark = pytest.mark.gpu


def _torch_wgrad(x, dy, k, dil, pad, lengths, flags, in_slope=1.0):
    b, t, c_in = x.shape
    xf, dyf = x.float(), dy.float()
    if in_slope != 1.0:
        xf = F.leaky_relu(xf, in_slope)
    m = None if lengths is None else (torch.arange(t, device=x.device)[None, :, None] < lengths[:, None, None])
    if flags & 1:
        xf = xf * m
    if flags & 2:
        dyf = dyf * m
    w = torch.zeros(dy.size(2), c_in, k, device=x.device, requires_grad=True)
    y = F.conv1d(xf.transpose(1, 2), w, None, 1, pad, dil)
    (g,) = torch.autograd.grad(y, w, dyf.transpose(1, 2))
    return g.permute(2, 0, 1).contiguous(), dyf.sum((0, 1))


def _entries(dtype, shapes, b, t, seed=0):
    torch.manual_seed(seed)
    lengths = torch.tensor([t, max(1, t - 19), max(1, t // 3), 7][:b], dtype=torch.int32, device="cuda")
    out = []
    for (c_in, c_out, k, dil, flags) in shapes:
        x = torch.randn(b, t, c_in, device="cuda").to(dtype)
        dy = torch.randn(b, t, c_out, device="cuda").to(dtype)
        out.append(dict(x=x, dy=dy, k=k, dil=dil, pad=(k - 1) * dil // 2, flags=flags, lengths=lengths if flags else None,
                        out=torch.full((k, c_out, c_in), float("nan"), device="cuda"), dbias=torch.full((c_out,), float("nan"), device="cuda")))
    return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_large_group_runs_unsplit_and_matches_torch(pkg, dtype):
    """A WaveNet-stack-like group (enough tiles: no slabs): k = 5 gate convolutions, 1x1 res/skip halves, masks, odd channel counts."""
    K = pkg.kernels
    shapes = [(192, 384, 5, 1, 0)] * 3 + [(192, 192, 1, 1, 2)] * 4 + [(96, 40, 3, 2, 3), (24, 200, 5, 1, 1), (192, 384, 5, 1, 2)]
    ents = _entries(dtype, shapes, 3, 333)
    assert K.conv1d_cl_wgrad_batch(ents, None)
    torch.cuda.synchronize()
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    for e in ents:
        gw, gb = _torch_wgrad(e["x"], e["dy"], e["k"], e["dil"], e["pad"], e["lengths"], e["flags"])
        assert rel_err(e["out"], gw) < tol and rel_err(e["dbias"], gb) < tol


def test_small_group_splits_into_slabs_and_equals_the_per_layer_kernel(pkg):
    """Few tiles: the launcher splits the (b, t) reductions; the deferred second stage finishes them.  Also the ACCUM flag."""
    K = pkg.kernels
    ents = _entries(torch.bfloat16, [(64, 64, 3, 1, 0), (128, 64, 1, 1, 2)], 4, 700, seed=1)
    defer = K.DeferredReductions(ents[0]["x"].device)
    assert K.conv1d_cl_wgrad_batch(ents, defer)
    assert defer.pending, "a two-entry group must have been split"
    defer.flush()
    for e in ents:
        want = K.conv1d_cl_wgrad_raw(e["x"], e["dy"], e["k"], lengths=e["lengths"], dil=e["dil"], pad=e["pad"], flags=e["flags"])
        assert rel_err(e["out"], want) < 1e-5
        gw, gb = _torch_wgrad(e["x"], e["dy"], e["k"], e["dil"], e["pad"], e["lengths"], e["flags"])
        assert rel_err(e["out"], gw) < 2e-2 and rel_err(e["dbias"], gb) < 2e-2
    # accumulate onto existing values, unsplit
    e = _entries(torch.float32, [(32, 48, 3, 1, 0)] * 40, 2, 130, seed=2)
    for q in e:
        q["out"].fill_(1.5); q["dbias"].fill_(-2.0); q["flags"] = K.CONV_ACCUM
    assert K.conv1d_cl_wgrad_batch(e, None)
    gw, gb = _torch_wgrad(e[7]["x"], e[7]["dy"], 3, 1, 1, None, 0)
    assert rel_err(e[7]["out"], gw + 1.5) < 2e-5 and rel_err(e[7]["dbias"], gb - 2.0) < 2e-5


def test_decoder_like_group_long_reductions_split_and_input_activation(pkg):
    """The decoder's shapes: few tiles x very long reductions (t = 4096, 32 / 64 channels, k = 3 / 7 / 11, dilations, fused
    leaky-relu on the input): the launcher splits such entries into slabs while wide entries of the same call stay unsplit."""
    K = pkg.kernels
    shapes = [(64, 64, 7, 3, 0), (64, 64, 11, 5, 0), (32, 32, 11, 1, 0), (256, 256, 7, 1, 0), (256, 256, 11, 3, 0), (128, 128, 3, 5, 0)]
    ents = _entries(torch.bfloat16, shapes, 2, 4096, seed=3)
    for e in ents:
        e["in_slope"] = 0.1
    defer = K.DeferredReductions(ents[0]["x"].device)
    assert K.conv1d_cl_wgrad_batch(ents, defer)
    assert defer.pending
    defer.flush()
    for e in ents:
        gw, gb = _torch_wgrad(e["x"], e["dy"], e["k"], e["dil"], e["pad"], None, 0, in_slope=0.1)
        assert rel_err(e["out"], gw) < 2e-2 and rel_err(e["dbias"], gb) < 2e-2, (e["k"], e["dil"])


def test_ineligible_entries_are_refused(pkg):
    K = pkg.kernels
    x = torch.randn(2, 64, 32, device="cuda"); dy = torch.randn(2, 62, 32, device="cuda")
    # a "valid" convolution (t_out != t) is not a batch entry
    import ctypes
    L = pkg._lib.lib()
    d = (pkg._lib.WgradDesc * 1)()
    d[0].dtype, d[0].b, d[0].t, d[0].c_in, d[0].c_out, d[0].k, d[0].dil, d[0].pad, d[0].stride = 0, 2, 64, 32, 32, 3, 1, 0, 1
    out = torch.empty(3, 32, 32, device="cuda")
    d[0].x, d[0].dy, d[0].dw = x.data_ptr(), dy.data_ptr(), out.data_ptr()
    assert L.vits_conv1d_cl_wgrad_batch(ctypes.addressof(d), 1, None, None) == pkg._lib.E_UNSUPPORTED
