"""VITS generator and discriminators — mirror of the reference's models.py: same class names,
constructor signatures, parameter names (state_dict keys, incl. weight_g / weight_v) and return
tuples, so checkpoints and callers are interchangeable.  File:line citations point at the reference.
"""
import math

import torch
from torch import nn
from torch.nn import functional as F

from . import attentions
from . import commons
from . import kernels as K
from . import modules
from .commons import get_padding
from .modules import Conv1d, WNConv1d, WNConvTranspose1d
from .rng import noise


class StochasticDurationPredictor(nn.Module):
    # models.py:17-95
    def __init__(self, in_channels, filter_channels, kernel_size, p_dropout, n_flows=4, gin_channels=0):
        super().__init__()
        filter_channels = in_channels        # models.py:20 ("it needs to be removed from future version")
        self.in_channels, self.filter_channels, self.kernel_size = in_channels, filter_channels, kernel_size
        self.p_dropout, self.n_flows, self.gin_channels = p_dropout, n_flows, gin_channels

        self.log_flow = modules.Log()
        self.flows = nn.ModuleList()
        self.flows.append(modules.ElementwiseAffine(2))
        for _ in range(n_flows):
            self.flows.append(modules.ConvFlow(2, filter_channels, kernel_size, n_layers=3))
            self.flows.append(modules.Flip())

        self.post_pre = Conv1d(1, filter_channels, 1)
        self.post_proj = Conv1d(filter_channels, filter_channels, 1)
        self.post_convs = modules.DDSConv(filter_channels, kernel_size, n_layers=3, p_dropout=p_dropout)
        self.post_flows = nn.ModuleList()
        self.post_flows.append(modules.ElementwiseAffine(2))
        for _ in range(4):
            self.post_flows.append(modules.ConvFlow(2, filter_channels, kernel_size, n_layers=3))
            self.post_flows.append(modules.Flip())

        self.pre = Conv1d(in_channels, filter_channels, 1)
        self.proj = Conv1d(filter_channels, filter_channels, 1)
        self.convs = modules.DDSConv(filter_channels, kernel_size, n_layers=3, p_dropout=p_dropout)
        if gin_channels != 0:
            self.cond = Conv1d(gin_channels, filter_channels, 1)

    def forward(self, x, x_mask, w=None, g=None, reverse=False, noise_scale=1.0):
        """Reference call surface ([b, c, t] tensors); everything inside runs channels-last: the 1x1
        convolutions on the matrix cores, DDSConv on the row kernels, the splines in one launch each."""
        from . import wn_cl
        dtype = wn_cl.compute_dtype()
        lengths = wn_cl.lengths_of(x_mask)
        m = x_mask.transpose(1, 2)                                     # [b, t, 1]
        xc = wn_cl.conv_cl(torch.detach(x).transpose(1, 2).contiguous(), wn_cl.weight_of(self.pre), wn_cl.bias_of(self.pre), dtype=dtype)
        if g is not None:
            xc = xc + self.cond(torch.detach(g)).transpose(1, 2).to(dtype)
        xc = self.convs.forward_cl(xc, lengths, m, final_mask=False)             # (proj below is a masked 1x1 convolution)
        xc = wn_cl.conv_cl(xc, wn_cl.weight_of(self.proj), wn_cl.bias_of(self.proj), lengths, mask_out=True, dtype=dtype)

        # modules.Flip (reverse the two channels) is not executed: `swap` records the parity of the flips so far, the ConvFlow
        # layers exchange the roles of the two channels instead, and the element-wise layers index their parameters accordingly
        state = {"swap": False}

        def run(flow, z, cond):
            if isinstance(flow, modules.Flip):
                state["swap"] = not state["swap"]
                return z, None
            if isinstance(flow, modules.ElementwiseAffine):
                from . import rowops
                return rowops.flow_affine(z, flow.m, flow.logs, lengths, state["swap"], reverse)       # (y, logdet | None)
            out = flow.forward_cl(z, lengths, m, cond, reverse, swap=state["swap"])
            return (out[0], out[1]) if not reverse else (out, None)

        if not reverse:
            assert w is not None
            w_cl = w.transpose(1, 2).float()                           # [b, t, 1]
            from . import rowops
            h_w = rowops.flow_front(w_cl, 0, self.post_pre.weight, self.post_pre.bias, None, dtype)       # Conv1d(1, C, 1) on the durations
            h_w = self.post_convs.forward_cl(h_w, lengths, m, final_mask=False)
            h_w = wn_cl.conv_cl(h_w, wn_cl.weight_of(self.post_proj), wn_cl.bias_of(self.post_proj), lengths, mask_out=True, dtype=dtype)
            e_q = noise.randn(w.size(0), 2, w.size(2), device=x.device, dtype=torch.float32).transpose(1, 2) * m
            z_q, logdet_tot_q = e_q, 0
            cond_q = xc + h_w
            for flow in self.post_flows:
                z_q, ld = run(flow, z_q, cond_q)
                if ld is not None:
                    logdet_tot_q = logdet_tot_q + ld
            assert not state["swap"]                                   # an even number of flips: natural channel order
            # u = sigmoid(z_u) m, z0 = (w - u) m, its log-determinant term, modules.Log on z0 and cat([z0, z1]): one kernel each way
            z, ld_u, logdet_tot = rowops.dequant_log(z_q, w_cl, lengths)
            logdet_tot_q = logdet_tot_q + ld_u
            logq = commons.sum12(-0.5 * (commons.LOG_2PI + (e_q ** 2)) * m) - logdet_tot_q
            for flow in self.flows:
                z, ld = run(flow, z, xc)
                if ld is not None:
                    logdet_tot = logdet_tot + ld
            nll = commons.sum12(0.5 * (commons.LOG_2PI + (z ** 2)) * m) - logdet_tot
            return nll + logq                                          # [b]
        flows = list(reversed(self.flows))
        flows = flows[:-2] + [flows[-1]]                              # models.py:88-89 "remove a useless vflow"
        z = noise.randn(x.size(0), 2, x.size(2), device=x.device, dtype=torch.float32).transpose(1, 2) * noise_scale
        for flow in flows:
            z, _ = run(flow, z, xc)
        z0 = z[..., 1:] if state["swap"] else z[..., :1]               # channel 0 of the (virtually flipped) state
        return z0.transpose(1, 2).to(x.dtype)                          # logw [b, 1, t]


class DurationPredictor(nn.Module):
    # models.py:98-132 (used only with use_sdp=False)
    def __init__(self, in_channels, filter_channels, kernel_size, p_dropout, gin_channels=0):
        super().__init__()
        self.in_channels, self.filter_channels, self.kernel_size = in_channels, filter_channels, kernel_size
        self.p_dropout, self.gin_channels = p_dropout, gin_channels
        self.drop = nn.Dropout(p_dropout)
        self.conv_1 = Conv1d(in_channels, filter_channels, kernel_size, padding=kernel_size // 2)
        self.norm_1 = modules.LayerNorm(filter_channels)
        self.conv_2 = Conv1d(filter_channels, filter_channels, kernel_size, padding=kernel_size // 2)
        self.norm_2 = modules.LayerNorm(filter_channels)
        self.proj = Conv1d(filter_channels, 1, 1)
        if gin_channels != 0:
            self.cond = Conv1d(gin_channels, in_channels, 1)

    def forward(self, x, x_mask, g=None):
        """models.py:120-132 on the channels-last kernels: masked k-tap convolutions with ReLU as the next kernel's input
        activation... (ReLU sits BEFORE the LayerNorm here, so it is applied by the row kernel's caller instead)."""
        from . import rowops, wn_cl
        dtype = wn_cl.compute_dtype()
        lengths = wn_cl.lengths_of(x_mask)
        x = torch.detach(x)
        if g is not None:
            x = x + self.cond(torch.detach(g))
        h = x.transpose(1, 2).to(dtype).contiguous()
        pad = self.kernel_size // 2
        for conv, norm in ((self.conv_1, self.norm_1), (self.conv_2, self.norm_2)):
            h = wn_cl.conv_cl(h, wn_cl.weight_of(conv), wn_cl.bias_of(conv), lengths, pad=pad, mask_in=True, dtype=dtype)
            h = self.drop(rowops.ln_act(torch.relu(h), norm.gamma, norm.beta, None, norm.eps, 0))
        cpad = (-1) % 8
        y = wn_cl.conv_cl(h, wn_cl.weight_of(self.proj, pad_out=cpad), wn_cl.bias_of(self.proj, cpad), lengths, mask_in=True, mask_out=True, dtype=dtype)
        return y[..., :1].transpose(1, 2).to(x.dtype)


class _EmbeddingScaledFn(torch.autograd.Function):
    """emb(idx) * scale with the weight gradient as a one-hot product on this library's fp32 weight-gradient kernel:
    torch's embedding backward sorts the 3 216 token ids of a batch (rocprim scan + device memsets — what a replayed graph
    cannot rely on) and is not the bottleneck either way."""

    @staticmethod
    def forward(ctx, idx, weight, scale):
        ctx.save_for_backward(idx)
        ctx.scale, ctx.shape = scale, weight.shape
        return F.embedding(idx, weight.detach()) * scale

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        v, h = ctx.shape
        vp = (v + 7) // 8 * 8
        n = idx.numel()
        onehot = (idx.reshape(n, 1) == torch.arange(vp, device=idx.device)[None, :]).to(torch.float32).view(1, n, vp)
        dyf = (dy.float() * ctx.scale).reshape(1, n, h).contiguous()
        dw = K.conv1d_cl_wgrad_raw(onehot, dyf, 1)                         # [1][h][vp] = sum_n dy[n][:] (x) onehot[n][:]
        return None, dw[0].t()[:v].contiguous(), None


def _embedding_scaled(idx, weight, scale):
    if idx.is_cuda:
        return _EmbeddingScaledFn.apply(idx, weight, scale)
    return F.embedding(idx, weight) * scale


class TextEncoder(nn.Module):
    # models.py:135-176
    def __init__(self, n_vocab, out_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout):
        super().__init__()
        self.n_vocab, self.out_channels, self.hidden_channels, self.filter_channels = n_vocab, out_channels, hidden_channels, filter_channels
        self.n_heads, self.n_layers, self.kernel_size, self.p_dropout = n_heads, n_layers, kernel_size, p_dropout
        self.emb = nn.Embedding(n_vocab, hidden_channels)
        nn.init.normal_(self.emb.weight, 0.0, hidden_channels ** -0.5)
        self.encoder = attentions.Encoder(hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout)
        self.proj = Conv1d(hidden_channels, out_channels * 2, 1)

    def forward(self, x, x_lengths):
        x = _embedding_scaled(x, self.emb.weight, math.sqrt(self.hidden_channels))         # [b, t, h]
        x = torch.transpose(x, 1, -1)                             # [b, h, t]
        x_mask = torch.unsqueeze(commons.sequence_mask(x_lengths, x.size(2)), 1).to(x.dtype)
        x = self.encoder(x * x_mask, x_mask)
        from . import wn_cl                    # 1x1 projection on the channels-last MFMA kernel, `* x_mask` as its epilogue
        stats = wn_cl.conv_cl(x.transpose(1, 2).contiguous(), wn_cl.weight_of(self.proj), wn_cl.bias_of(self.proj),
                              x_lengths.to(torch.int32), mask_out=True).transpose(1, 2)
        m, logs = torch.split(stats, self.out_channels, dim=1)
        return x, m, logs, x_mask


class ResidualCouplingBlock(nn.Module):
    # models.py:179-209
    def __init__(self, channels, hidden_channels, kernel_size, dilation_rate, n_layers, n_flows=4, gin_channels=0):
        super().__init__()
        self.channels, self.hidden_channels, self.kernel_size = channels, hidden_channels, kernel_size
        self.dilation_rate, self.n_layers, self.n_flows, self.gin_channels = dilation_rate, n_layers, n_flows, gin_channels
        self.flows = nn.ModuleList()
        for _ in range(n_flows):
            self.flows.append(modules.ResidualCouplingLayer(channels, hidden_channels, kernel_size, dilation_rate, n_layers,
                                                            gin_channels=gin_channels, mean_only=True))
            self.flows.append(modules.Flip())

    def forward(self, x, x_mask, g=None, reverse=False):
        """x [b, c, t] -> [b, c, t].  The four coupling layers and the channel flips between them run
        channels-last ([b, t, c]) without leaving that layout."""
        mask_cl = x_mask.transpose(1, 2)
        lengths = x_mask[:, 0, :].sum(-1).to(torch.int32)
        h = x.transpose(1, 2).contiguous()
        if not reverse:
            flows = list(self.flows)
            i = 0
            while i < len(flows):
                flow = flows[i]
                if isinstance(flow, modules.Flip):
                    h = torch.flip(h, [2])
                    i += 1
                    continue
                fold = i + 1 < len(flows) and isinstance(flows[i + 1], modules.Flip)       # the Flip after a coupling layer rides in its tail kernel
                h, _ = flow.forward_cl(h, lengths, mask_cl, g=g, reverse=False, flip_after=fold)
                i += 2 if fold else 1
        else:
            for flow in reversed(self.flows):
                if isinstance(flow, modules.Flip):
                    h = torch.flip(h, [2])
                else:
                    h = flow.forward_cl(h, lengths, mask_cl, g=g, reverse=True)
        return h.transpose(1, 2)


class PosteriorEncoder(nn.Module):
    # models.py:212-241
    def __init__(self, in_channels, out_channels, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0):
        super().__init__()
        self.in_channels, self.out_channels, self.hidden_channels = in_channels, out_channels, hidden_channels
        self.kernel_size, self.dilation_rate, self.n_layers, self.gin_channels = kernel_size, dilation_rate, n_layers, gin_channels
        self.pre = Conv1d(in_channels, hidden_channels, 1)
        self.enc = modules.WN(hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=gin_channels)
        self.proj = Conv1d(hidden_channels, out_channels * 2, 1)

    def forward(self, x, x_lengths, g=None):
        """x [b, spec_channels, t] -> z, m, logs [b, c, t] (transposed views of channels-last results), x_mask."""
        from . import wn_cl
        x_mask = torch.unsqueeze(commons.sequence_mask(x_lengths, x.size(2)), 1).to(x.dtype)
        lengths = x_lengths.to(torch.int32)
        dtype = wn_cl.compute_dtype()
        # channels last, input channels padded to the kernels' vector width (zero weights on the pad)
        cin = self.in_channels
        cpad = (-cin) % 8
        x_cl = torch.nn.functional.pad(x.transpose(1, 2), (0, cpad)).to(dtype).contiguous()
        h = wn_cl.conv_cl(x_cl, wn_cl.weight_of(self.pre, pad_in=cpad), wn_cl.bias_of(self.pre), lengths, mask_out=True)
        h = wn_cl.wn_forward_cl(self.enc, h, lengths, g)
        stats = wn_cl.conv_cl(h, wn_cl.weight_of(self.proj), wn_cl.bias_of(self.proj), lengths, mask_out=True).float()
        m, logs = stats[..., :self.out_channels], stats[..., self.out_channels:]
        z = (m + noise.randn_like(m.transpose(1, 2)).transpose(1, 2) * torch.exp(logs)) * x_mask.transpose(1, 2)
        return z.transpose(1, 2), m.transpose(1, 2), logs.transpose(1, 2), x_mask


class Generator(nn.Module):
    # models.py:244-296 (HiFi-GAN decoder)
    def __init__(self, initial_channel, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                 upsample_initial_channel, upsample_kernel_sizes, gin_channels=0):
        super().__init__()
        self.num_kernels = len(resblock_kernel_sizes)
        self.num_upsamples = len(upsample_rates)
        self.conv_pre = Conv1d(initial_channel, upsample_initial_channel, 7, 1, padding=3)
        resblock_cls = modules.ResBlock1 if resblock == "1" else modules.ResBlock2

        self.ups = nn.ModuleList()
        for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
            self.ups.append(WNConvTranspose1d(upsample_initial_channel // (2 ** i), upsample_initial_channel // (2 ** (i + 1)),
                                              k, u, padding=(k - u) // 2))
        self.resblocks = nn.ModuleList()
        for i in range(len(self.ups)):
            ch = upsample_initial_channel // (2 ** (i + 1))
            for k, d in zip(resblock_kernel_sizes, resblock_dilation_sizes):
                self.resblocks.append(resblock_cls(ch, k, d))
        self.conv_post = Conv1d(ch, 1, 7, 1, padding=3, bias=False)
        modules._burn_init_weights(self.ups)                       # models.py:265 `self.ups.apply(init_weights)`
        if gin_channels != 0:
            self.cond = Conv1d(gin_channels, upsample_initial_channel, 1)

    def forward(self, x, g=None):
        """x [b, c, t] (reference layout) -> waveform [b, 1, t * prod(upsample_rates)].  The whole
        stack runs as one autograd node over the channels-last HIP kernels (decoder_cl.DecoderFn);
        bf16 compute under autocast, exact fp32 otherwise."""
        from . import decoder_cl
        if getattr(self, "_plan", None) is None:
            self._plan = decoder_cl.DecoderPlan(self)
        dtype = torch.bfloat16 if torch.is_autocast_enabled() else torch.float32
        cond = self.cond(g).squeeze(-1).float() if g is not None else None           # [b, c_up0], models.py:272-273
        y = decoder_cl.DecoderFn.apply(self._plan, dtype, x.transpose(1, 2), cond, *decoder_cl.prepared_weights(self))
        return y[..., 0].unsqueeze(1).float()

    def remove_weight_norm(self):                              # models.py:291-296
        print('Removing weight norm...')
        for l in self.ups:
            modules.remove_weight_norm(l)
        for l in self.resblocks:
            l.remove_weight_norm()


class _WNConv2dK1(nn.Module):
    """weight_norm(Conv2d(cin, cout, (k,1), (s,1), padding=(p,0))) of DiscriminatorP (models.py:304-312);
    parameters `bias`, `weight_g` [cout,1,1,1], `weight_v` [cout,cin,k,1]."""

    def __init__(self, cin, cout, k, s, p):
        super().__init__()
        proto = nn.Conv2d(cin, cout, (k, 1), (s, 1), padding=(p, 0))
        self.stride, self.padding = (s, 1), (p, 0)
        self.in_channels, self.out_channels, self.kernel_size = cin, cout, k
        self.bias = nn.Parameter(proto.bias.data)
        v = proto.weight.data
        self.weight_g = nn.Parameter(torch.linalg.vector_norm(v, 2, dim=(1, 2, 3), keepdim=True))
        self.weight_v = nn.Parameter(v)

    @property
    def weight(self):
        """weight-normed weight as a Conv1d weight [c_out, c_in, k]"""
        return K.weight_norm(self.weight_v, self.weight_g).squeeze(-1)

    def forward(self, x):
        return F.conv2d(x, K.weight_norm(self.weight_v, self.weight_g).to(x.dtype), self.bias.to(x.dtype), self.stride, self.padding)


class _Fmaps(list):
    """Feature maps in the reference's layout (views) + the contiguous channels-last tensors they view (`cl`, items of the
    real half of the batch first) and the element count of one half of each feature map (`den`): reduce.feature_l1."""

    def __init__(self):
        super().__init__()
        self.cl, self.den = [], []


def _disc_outputs(period, n, y8, hs):
    """(logits [n, t'*p], feature maps in the reference's layout) from one discriminator's channels-last results."""
    fmap = _Fmaps()
    for h in hs:
        # P: [(n, w), t', c] -> view [n, c, t', period];  S: [n, t, c] -> view [n, c, t]
        fmap.append(h.view(n, period, h.size(1), h.size(2)).permute(0, 3, 2, 1) if period > 1 else h.transpose(1, 2))
        fmap.cl.append(h); fmap.den.append(h.numel() // 2)
    y = y8[..., :1]                                                          # channels 1..7 are exactly zero
    y = y.reshape(n, period, y.size(1), 1).permute(0, 3, 2, 1) if period > 1 else y.transpose(1, 2)
    fmap.append(y)
    fmap.cl.append(y8); fmap.den.append(y.numel() // 2)
    return torch.flatten(y, 1, -1), fmap


class DiscriminatorP(nn.Module):
    # models.py:299-335
    def __init__(self, period, kernel_size=5, stride=3, use_spectral_norm=False):
        super().__init__()
        if use_spectral_norm:
            raise NotImplementedError("use_spectral_norm=False in every reference config")
        self.period = period
        self.use_spectral_norm = use_spectral_norm
        p = get_padding(kernel_size, 1)
        self.convs = nn.ModuleList([
            _WNConv2dK1(1, 32, kernel_size, stride, p), _WNConv2dK1(32, 128, kernel_size, stride, p),
            _WNConv2dK1(128, 512, kernel_size, stride, p), _WNConv2dK1(512, 1024, kernel_size, stride, p),
            _WNConv2dK1(1024, 1024, kernel_size, 1, p)])
        self.conv_post = _WNConv2dK1(1024, 1, 3, 1, 1)

    def forward(self, x):
        return self.forward_hip(x)

    def forward_hip(self, x):
        """x [n, 1, t] -> (logits [n, t'*p], fmap list in the reference's [n, c, t', period] layout, as views of the
        channels-last activations).  The (k, 1) convolutions act along t' only: the period axis is folded into the batch
        and the whole discriminator runs as one autograd node (disc_cl.DiscFn)."""
        from . import disc_cl
        (y8, hs), = disc_cl.run([self], x[:, 0, :].float())
        return _disc_outputs(self.period, x.size(0), y8, hs)


class DiscriminatorS(nn.Module):
    # models.py:338-361
    period = 1

    def __init__(self, use_spectral_norm=False):
        super().__init__()
        if use_spectral_norm:
            raise NotImplementedError("use_spectral_norm=False in every reference config")
        self.convs = nn.ModuleList([
            WNConv1d(1, 16, 15, 1, padding=7), WNConv1d(16, 64, 41, 4, groups=4, padding=20),
            WNConv1d(64, 256, 41, 4, groups=16, padding=20), WNConv1d(256, 1024, 41, 4, groups=64, padding=20),
            WNConv1d(1024, 1024, 41, 4, groups=256, padding=20), WNConv1d(1024, 1024, 5, 1, padding=2)])
        self.conv_post = WNConv1d(1024, 1, 3, 1, padding=1)

    def forward(self, x):
        return self.forward_hip(x)

    def forward_hip(self, x):
        """x [n, 1, t] -> (logits [n, t'], fmap list in the reference's [n, c, t'] layout, as views of channels-last
        activations).  One autograd node (disc_cl.DiscFn); the grouped layers (groups 4..256, 4 input channels per group)
        take DENSE block-diagonal operands and the kernel only walks the input channels a tile of output channels can see
        (vits_conv_desc.groups)."""
        from . import disc_cl
        (y8, hs), = disc_cl.run([self], x[:, 0, :].float())
        return _disc_outputs(1, x.size(0), y8, hs)


class MultiPeriodDiscriminator(nn.Module):
    # models.py:364-386
    def __init__(self, use_spectral_norm=False):
        super().__init__()
        periods = [2, 3, 5, 7, 11]
        discs = [DiscriminatorS(use_spectral_norm=use_spectral_norm)]
        discs = discs + [DiscriminatorP(i, use_spectral_norm=use_spectral_norm) for i in periods]
        self.discriminators = nn.ModuleList(discs)

    def forward(self, y, y_hat):
        """Real and generated waveforms go through every discriminator as ONE batch of 2b (the reference runs them as two
        passes, models.py:375-377; the convolutions have no cross-batch coupling, so the results are identical and every
        kernel sees twice the rows), and all six discriminators are ONE autograd node (disc_cl.DiscFn).  When the
        discriminator is frozen (generator step) only the generated half needs a backward: n_lo = b."""
        from . import disc_cl, weight_arena
        from .reduce import FmapLists, LogitLists
        b = y.size(0)
        yy = torch.cat([y, y_hat], 0)
        frozen = not any(p.requires_grad for p in self.parameters())
        with weight_arena.scope(self, MultiPeriodDiscriminator._arena_specs):
            res = disc_cl.run(list(self.discriminators), yy[:, 0, :].float(), n_lo=b if frozen else 0)
        outs = [_disc_outputs(d.period, 2 * b, y8, hs) for d, (y8, hs) in zip(self.discriminators, res)]
        y_d_rs, y_d_gs = LogitLists(), LogitLists()
        y_d_rs.y8 = y_d_gs.y8 = [y8 for y8, _ in res]           # fused least-squares losses read the logits where they lie
        fmap_rs, fmap_gs = FmapLists(), FmapLists()
        cl, den = [], []
        for out, fmap in outs:
            y_d_rs.append(out[:b]); y_d_gs.append(out[b:])
            fmap_rs.append([f[:b] for f in fmap]); fmap_gs.append([f[b:] for f in fmap])
            cl += getattr(fmap, "cl", [None] * len(fmap)); den += getattr(fmap, "den", [0] * len(fmap))
        if all(h is not None for h in cl):            # every discriminator ran channels-last: the fused feature loss applies
            fmap_rs.cl = fmap_gs.cl = (cl, den)
        return y_d_rs, y_d_gs, fmap_rs, fmap_gs

    @staticmethod
    def _arena_specs(net):
        from .weight_arena import Spec
        specs = []
        for d in net.discriminators:
            specs.append(Spec(d.convs[0], c_in_p=8))
            specs += [Spec(l, groups=getattr(l, "groups", 1)) for l in d.convs[1:]]
            specs.append(Spec(d.conv_post, c_out_p=8))
        return specs


class SynthesizerTrn(nn.Module):
    """Synthesizer for training (models.py:390-533)."""

    def __init__(self, n_vocab, spec_channels, segment_size, inter_channels, hidden_channels, filter_channels, n_heads,
                 n_layers, kernel_size, p_dropout, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                 upsample_initial_channel, upsample_kernel_sizes, n_speakers=0, gin_channels=0, use_sdp=True, **kwargs):
        super().__init__()
        self.n_vocab, self.spec_channels, self.inter_channels, self.hidden_channels = n_vocab, spec_channels, inter_channels, hidden_channels
        self.filter_channels, self.n_heads, self.n_layers, self.kernel_size, self.p_dropout = filter_channels, n_heads, n_layers, kernel_size, p_dropout
        self.resblock, self.resblock_kernel_sizes, self.resblock_dilation_sizes = resblock, resblock_kernel_sizes, resblock_dilation_sizes
        self.upsample_rates, self.upsample_initial_channel, self.upsample_kernel_sizes = upsample_rates, upsample_initial_channel, upsample_kernel_sizes
        self.segment_size, self.n_speakers, self.gin_channels, self.use_sdp = segment_size, n_speakers, gin_channels, use_sdp
        self.side_branches = {"enc_p", "dp", "prior"}      # training forward: sub-graphs that run as side-stream branches (kernels.SideBranch)

        self.enc_p = TextEncoder(n_vocab, inter_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout)
        self.dec = Generator(inter_channels, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                             upsample_initial_channel, upsample_kernel_sizes, gin_channels=gin_channels)
        self.enc_q = PosteriorEncoder(spec_channels, inter_channels, hidden_channels, 5, 1, 16, gin_channels=gin_channels)
        self.flow = ResidualCouplingBlock(inter_channels, hidden_channels, 5, 1, 4, gin_channels=gin_channels)
        if use_sdp:
            self.dp = StochasticDurationPredictor(hidden_channels, 192, 3, 0.5, 4, gin_channels=gin_channels)
        else:
            self.dp = DurationPredictor(hidden_channels, 256, 3, 0.5, gin_channels=gin_channels)
        if n_speakers >= 1:
            self.emb_g = nn.Embedding(n_speakers, gin_channels)

    def _speaker(self, sid):
        return self.emb_g(sid).unsqueeze(-1) if self.n_speakers > 0 else None     # [b, h, 1]

    @staticmethod
    def neg_cent(z_p, m_p, logs_p):
        """Negative cross-entropy of every (frame, token) pair (models.py:470-477) as one fp32 matrix-core launch
        (csrc/align.hip) — fp32 also inside an autocast region: this tensor feeds the discrete alignment DP, and bf16
        products (errors ~0.5 on values ~100) change paths."""
        return K.neg_cent(z_p, m_p, logs_p)

    @staticmethod
    def _arena_specs(net):
        """Every convolution of the posterior encoder, the flow and the decoder that runs on the HIP
        kernels, for weight_arena (one preparation launch per forward, one gradient launch per backward)."""
        from .weight_arena import Spec

        def wn_specs(wn):
            H, out = wn.hidden_channels, []
            if wn.gin_channels != 0 and wn.gin_channels % 8 == 0:
                out.append(Spec(wn.cond_layer, bias=True))     # speaker conditioning of all layers: one [b,1,gin] x [gin, 2HL] product
            for i in range(wn.n_layers):
                out.append(Spec(wn.in_layers[i]))
                out.append(Spec(wn.res_skip_layers[i]))        # rows [0, H) residual, [H, 2H) skip: one operand, one gradient launch
            return out

        def dds_specs(dds):
            return [Spec(c, bias=True) for c in dds.convs_1x1]

        # (bias=True: convolutions that run as single wn_cl.ConvCLFn nodes — their bias gradients ride in the deferred
        # weight-gradient launches and must reach the parameters through the arena; the fused WN / decoder nodes flush their own)
        specs = [Spec(net.enc_q.pre, c_in_p=(net.enc_q.in_channels + 7) // 8 * 8, bias=True)] + wn_specs(net.enc_q.enc) + [Spec(net.enc_q.proj, bias=True)]
        # text encoder: q/k/v/o projections, FFN convolutions, output projection
        for att, ffn in zip(net.enc_p.encoder.attn_layers, net.enc_p.encoder.ffn_layers):
            specs += [Spec(m, bias=True) for m in (att.conv_q, att.conv_k, att.conv_v, att.conv_o, ffn.conv_1, ffn.conv_2)]
        specs.append(Spec(net.enc_p.proj, bias=True))
        # stochastic duration predictor: every 192-channel 1x1 convolution; the 29-column spline projections padded to 32
        if isinstance(net.dp, StochasticDurationPredictor):
            dp = net.dp
            specs += [Spec(dp.pre, bias=True), Spec(dp.proj, bias=True), Spec(dp.post_proj, bias=True)]
            specs += dds_specs(dp.convs) + dds_specs(dp.post_convs)
            for fl in list(dp.flows) + list(dp.post_flows):
                if isinstance(fl, modules.ConvFlow):
                    specs += dds_specs(fl.convs) + [Spec(fl.proj, c_out_p=(fl.proj.out_channels + 7) // 8 * 8, bias=True)]
        for fl in net.flow.flows:
            if isinstance(fl, modules.ResidualCouplingLayer):
                specs += [Spec(fl.pre, bias=True)] + wn_specs(fl.enc) + [Spec(fl.post, bias=True)]
        dec = net.dec
        specs.append(Spec(dec.conv_pre))
        for i, up in enumerate(dec.ups):
            specs.append(Spec(up, transpose=True))
            for rb in dec.resblocks[i * dec.num_kernels:(i + 1) * dec.num_kernels]:
                for c in (list(rb.convs1) + list(rb.convs2)) if hasattr(rb, "convs1") else list(rb.convs):
                    specs.append(Spec(c))
        specs.append(Spec(dec.conv_post, c_out_p=8))
        return specs

    def _scope(self):
        from . import weight_arena
        return weight_arena.scope(self, SynthesizerTrn._arena_specs)

    def forward(self, x, x_lengths, y, y_lengths, sid=None):
        with self._scope():
            return self._forward(x, x_lengths, y, y_lengths, sid)

    def infer(self, x, x_lengths, sid=None, noise_scale=1, length_scale=1, noise_scale_w=1.0, max_len=None, durations=None):
        """Reference signature (models.py:499) plus `durations` [b, 1, t_x] (optional, not in the reference): frames per token
        to use instead of ceil(exp(logw)) — benchmark harnesses force a fixed utterance length with it."""
        with self._scope():
            return self._infer(x, x_lengths, sid, noise_scale, length_scale, noise_scale_w, max_len, durations)

    def voice_conversion(self, y, y_lengths, sid_src, sid_tgt):
        with self._scope():
            return self._voice_conversion(y, y_lengths, sid_src, sid_tgt)

    def _forward(self, x, x_lengths, y, y_lengths, sid=None):
        # The text encoder (~600 launches on [b, t_x, 192] tensors, forward + backward) and the posterior encoder + flow do not
        # depend on each other until the alignment: two branches (kernels.SideBranch), forward and — because every backward
        # runs on its forward's stream — backward.  Both branches use ONE side stream (lane 0; they never need each other's time:
        # this one is joined before the alignment, the duration predictor's is opened after it): with a lane each, the HIP
        # runtime may or may not give the two lanes separate hardware queues (it depends on the order streams were created in),
        # and three queues competing cost the main chain more than the extra overlap returns — 24.5 against 21.8 ms/step (C2,
        # round 3; rocprofv3 queue ids, tools/queue_summary.py).
        enc_branch = K.SideBranch(x.device, x, x_lengths, lane=0) if ("enc_p" in self.side_branches and x.is_cuda) else None
        if enc_branch is not None:
            enc_branch.__enter__()
        try:
            x, m_p, logs_p, x_mask = self.enc_p(x, x_lengths)
        finally:
            if enc_branch is not None:
                enc_branch.__exit__(None, None, None)
        g = self._speaker(sid)
        z, m_q, logs_q, y_mask = self.enc_q(y, y_lengths, g=g)
        z_p = self.flow(z, y_mask, g=g)
        if enc_branch is not None and "prior" in self.side_branches and "dp" in self.side_branches:
            return self._forward_prior_on_lane(x, m_p, logs_p, x_mask, z, z_p, m_q, logs_q, y_mask, y_lengths, g)
        if enc_branch is not None:
            x, m_p, logs_p, x_mask = enc_branch.join(x, m_p, logs_p, x_mask)

        with torch.no_grad():
            neg_cent = self.neg_cent(z_p, m_p, logs_p)
            attn_mask = torch.unsqueeze(x_mask, 2) * torch.unsqueeze(y_mask, -1)
            attn = K.maximum_path(neg_cent, attn_mask.squeeze(1)).unsqueeze(1).detach().to(x.dtype)

        w = attn.sum(2)
        # The duration predictor is ~800 launches on [b, t_x, 192] tensors that occupy a few CUs; nothing below needs its result
        # before the loss, and its input is detached (models.py:52-53), so it runs as a side branch next to the decoder — forward
        # here, backward next to the decoder's / discriminators' backward (kernels.SideBranch).
        branch = K.SideBranch(x.device, x, x_mask, w, g) if ("dp" in self.side_branches and x.is_cuda) else None
        if branch is not None:
            branch.__enter__()
        try:
            if self.use_sdp:
                l_length = self.dp(x, x_mask, w, g=g)
            else:
                logw_ = torch.log(w + 1e-6) * x_mask
                logw = self.dp(x, x_mask, g=g)
                l_length = commons.sum12((logw - logw_) ** 2)
        finally:
            if branch is not None:
                branch.__exit__(None, None, None)

        # expand prior
        m_p = torch.matmul(attn.squeeze(1), m_p.transpose(1, 2)).transpose(1, 2)
        logs_p = torch.matmul(attn.squeeze(1), logs_p.transpose(1, 2)).transpose(1, 2)

        z_slice, ids_slice = commons.rand_slice_segments(z, y_lengths, self.segment_size)
        o = self.dec(z_slice, g=g)
        if branch is not None:
            l_length = branch.join(l_length)
        l_length = l_length / torch.sum(x_mask)
        return o, l_length, attn, ids_slice, x_mask, y_mask, (z, z_p, m_p, logs_p, m_q, logs_q)

    def _forward_prior_on_lane(self, x, m_p, logs_p, x_mask, z, z_p, m_q, logs_q, y_mask, y_lengths, g):
        """The rest of _forward with everything between the flow and the losses that is NOT the decoder — alignment scores,
        the alignment search (16 workgroups for ~100 us), the duration predictor, the expansion of the prior — on the side
        stream the text encoder ran on (its results never leave that stream), next to the decoder on the main one."""
        prior = K.SideBranch(z_p.device, z_p, y_mask, g)
        with prior:
            with torch.no_grad():
                neg_cent = self.neg_cent(z_p, m_p, logs_p)
                attn_mask = torch.unsqueeze(x_mask, 2) * torch.unsqueeze(y_mask, -1)
                attn = K.maximum_path(neg_cent, attn_mask.squeeze(1)).unsqueeze(1).detach().to(x.dtype)
            w = attn.sum(2)
            if self.use_sdp:
                l_length = self.dp(x, x_mask, w, g=g)
            else:
                logw_ = torch.log(w + 1e-6) * x_mask
                logw = self.dp(x, x_mask, g=g)
                l_length = commons.sum12((logw - logw_) ** 2)
            l_length = l_length / torch.sum(x_mask)
            m_p = torch.matmul(attn.squeeze(1), m_p.transpose(1, 2)).transpose(1, 2)
            logs_p = torch.matmul(attn.squeeze(1), logs_p.transpose(1, 2)).transpose(1, 2)
        z_slice, ids_slice = commons.rand_slice_segments(z, y_lengths, self.segment_size)
        o = self.dec(z_slice, g=g)
        l_length, attn, m_p, logs_p, x_mask = prior.join(l_length, attn, m_p, logs_p, x_mask)
        return o, l_length, attn, ids_slice, x_mask, y_mask, (z, z_p, m_p, logs_p, m_q, logs_q)

    def _infer(self, x, x_lengths, sid=None, noise_scale=1, length_scale=1, noise_scale_w=1.0, max_len=None, durations=None):
        x, m_p, logs_p, x_mask = self.enc_p(x, x_lengths)
        g = self._speaker(sid)
        if self.use_sdp:
            logw = self.dp(x, x_mask, g=g, reverse=True, noise_scale=noise_scale_w)
        else:
            logw = self.dp(x, x_mask, g=g)
        w = torch.exp(logw) * x_mask * length_scale
        w_ceil = torch.ceil(w) if durations is None else durations.to(w.dtype) * x_mask
        y_lengths = torch.clamp_min(commons.sum12(w_ceil), 1).long()
        y_mask = torch.unsqueeze(commons.sequence_mask(y_lengths, None), 1).to(x_mask.dtype)
        attn_mask = torch.unsqueeze(x_mask, 2) * torch.unsqueeze(y_mask, -1)
        attn = commons.generate_path(w_ceil, attn_mask)

        m_p = torch.matmul(attn.squeeze(1), m_p.transpose(1, 2)).transpose(1, 2)        # [b, d, t']
        logs_p = torch.matmul(attn.squeeze(1), logs_p.transpose(1, 2)).transpose(1, 2)

        z_p = m_p + noise.randn_like(m_p) * torch.exp(logs_p) * noise_scale
        z = self.flow(z_p, y_mask, g=g, reverse=True)
        o = self.dec((z * y_mask)[:, :, :max_len], g=g)
        return o, attn, y_mask, (z, z_p, m_p, logs_p)

    def _voice_conversion(self, y, y_lengths, sid_src, sid_tgt):
        assert self.n_speakers > 0, "n_speakers have to be larger than 0."
        g_src = self.emb_g(sid_src).unsqueeze(-1)
        g_tgt = self.emb_g(sid_tgt).unsqueeze(-1)
        z, m_q, logs_q, y_mask = self.enc_q(y, y_lengths, g=g_src)
        z_p = self.flow(z, y_mask, g=g_src)
        z_hat = self.flow(z_p, y_mask, g=g_tgt, reverse=True)
        o_hat = self.dec(z_hat * y_mask, g=g_tgt)
        return o_hat, y_mask, (z, z_p, z_hat)
