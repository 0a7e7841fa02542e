"""Text-encoder transformer — mirror of the reference's attentions.py (Encoder :13-47,
MultiHeadAttention :101-254, FFN :257-303).  The unused Decoder class is out of scope (never
instantiated by models.py).  Options the VITS path never sets (proximal_bias, block_length,
heads_share=False, causal FFN, gelu FFN) are rejected explicitly."""
import torch
from torch import nn

from . import kernels as K
from .modules import Conv1d, LayerNorm


class Encoder(nn.Module):
    def __init__(self, hidden_channels, filter_channels, n_heads, n_layers, kernel_size=1, p_dropout=0.0, window_size=4, **kwargs):
        super().__init__()
        self.hidden_channels, self.filter_channels, self.n_heads, self.n_layers = hidden_channels, filter_channels, n_heads, n_layers
        self.kernel_size, self.p_dropout, self.window_size = kernel_size, p_dropout, window_size
        self.drop = nn.Dropout(p_dropout)
        self.attn_layers = nn.ModuleList()
        self.norm_layers_1 = nn.ModuleList()
        self.ffn_layers = nn.ModuleList()
        self.norm_layers_2 = nn.ModuleList()
        for _ in range(n_layers):
            self.attn_layers.append(MultiHeadAttention(hidden_channels, hidden_channels, n_heads, p_dropout=p_dropout, window_size=window_size))
            self.norm_layers_1.append(LayerNorm(hidden_channels))
            self.ffn_layers.append(FFN(hidden_channels, hidden_channels, filter_channels, kernel_size, p_dropout=p_dropout))
            self.norm_layers_2.append(LayerNorm(hidden_channels))

    def forward(self, x, x_mask):
        """x [b, c, t] (reference layout).  Runs channels-last: q/k/v/o and FFN convolutions on the MFMA
        kernel (ReLU and masks fused into the second FFN convolution's prologue), post-norm residual
        LayerNorms on the row kernel; the residual add is fused into the producing convolution when
        dropout is off (attentions.py:35-47)."""
        from . import rowops, wn_cl
        dtype = wn_cl.compute_dtype()
        lengths = wn_cl.lengths_of(x_mask)
        m = x_mask.transpose(1, 2)
        attn_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
        h = (x * x_mask).transpose(1, 2).to(dtype).contiguous()
        drop = self.training and self.p_dropout > 0
        for i in range(self.n_layers):
            att, ffn, n1, n2 = self.attn_layers[i], self.ffn_layers[i], self.norm_layers_1[i], self.norm_layers_2[i]
            y = att.forward_cl(h, attn_mask, lengths, None if drop else h)            # conv_o(attention) (+ h)
            h = rowops.ln_act(h + self.drop(y) if drop else y, n1.gamma, n1.beta, None, n1.eps, 0)
            y = ffn.forward_cl(h, lengths, None if drop else h)
            h = rowops.ln_act(h + self.drop(y) if drop else y, n2.gamma, n2.beta, None, n2.eps, 0)
        return (h * m.to(dtype)).transpose(1, 2).to(x.dtype)


class MultiHeadAttention(nn.Module):
    def __init__(self, channels, out_channels, n_heads, p_dropout=0.0, window_size=None, heads_share=True,
                 block_length=None, proximal_bias=False, proximal_init=False):
        super().__init__()
        assert channels % n_heads == 0
        if window_size is None or not heads_share or block_length is not None or proximal_bias or proximal_init:
            raise NotImplementedError("only the configuration attentions.Encoder uses (attentions.py:31) is built")
        self.channels, self.out_channels, self.n_heads, self.p_dropout = channels, out_channels, n_heads, p_dropout
        self.window_size = window_size
        self.attn = None
        self.k_channels = channels // n_heads
        self.conv_q = Conv1d(channels, channels, 1)
        self.conv_k = Conv1d(channels, channels, 1)
        self.conv_v = Conv1d(channels, channels, 1)
        self.conv_o = Conv1d(channels, out_channels, 1)
        self.drop = nn.Dropout(p_dropout)
        rel_stddev = self.k_channels ** -0.5
        self.emb_rel_k = nn.Parameter(torch.randn(1, window_size * 2 + 1, self.k_channels) * rel_stddev)
        self.emb_rel_v = nn.Parameter(torch.randn(1, window_size * 2 + 1, self.k_channels) * rel_stddev)
        nn.init.xavier_uniform_(self.conv_q.weight)
        nn.init.xavier_uniform_(self.conv_k.weight)
        nn.init.xavier_uniform_(self.conv_v.weight)

    def forward(self, x, c, attn_mask=None):
        if x is not c:
            raise NotImplementedError("relative attention is only available for self-attention (attentions.py:157)")
        q, k, v = self.conv_q(x), self.conv_k(c), self.conv_v(c)
        x, self.attn = K.rel_attention(q, k, v, self.emb_rel_k, self.emb_rel_v, attn_mask, self.n_heads,
                                       self.window_size, self.p_dropout, self.training)
        return self.conv_o(x)

    def forward_cl(self, h, attn_mask, lengths, res=None):
        """h [b, t, c] channels-last -> conv_o(attention(h)) (+ res), 1x1 projections on the MFMA kernel."""
        from . import wn_cl
        q = wn_cl.conv_cl(h, wn_cl.weight_of(self.conv_q), wn_cl.bias_of(self.conv_q))
        k = wn_cl.conv_cl(h, wn_cl.weight_of(self.conv_k), wn_cl.bias_of(self.conv_k))
        v = wn_cl.conv_cl(h, wn_cl.weight_of(self.conv_v), wn_cl.bias_of(self.conv_v))
        from . import attention_cl
        o, self.attn = attention_cl.rel_attention_cl(q, k, v, self.emb_rel_k, self.emb_rel_v, lengths, self.n_heads,
                                                     self.window_size, self.p_dropout, self.training, q.dtype)
        return wn_cl.conv_cl(o, wn_cl.weight_of(self.conv_o), wn_cl.bias_of(self.conv_o), res=res)


class FFN(nn.Module):
    def __init__(self, in_channels, out_channels, filter_channels, kernel_size, p_dropout=0.0, activation=None, causal=False):
        super().__init__()
        if activation is not None or causal:
            raise NotImplementedError("only the ReLU, same-padded FFN of attentions.Encoder is built")
        self.in_channels, self.out_channels, self.filter_channels, self.kernel_size = in_channels, out_channels, filter_channels, kernel_size
        self.p_dropout = p_dropout
        # same padding: pad_l = (k-1)//2, pad_r = k//2 (attentions.py:295-303); symmetric for odd k
        assert kernel_size % 2 == 1
        self.conv_1 = Conv1d(in_channels, filter_channels, kernel_size, padding=(kernel_size - 1) // 2)
        self.conv_2 = Conv1d(filter_channels, out_channels, kernel_size, padding=(kernel_size - 1) // 2)
        self.drop = nn.Dropout(p_dropout)

    def forward(self, x, x_mask):
        x = self.conv_1(x * x_mask)
        x = self.drop(torch.relu(x))
        x = self.conv_2(x * x_mask)
        return x * x_mask

    def forward_cl(self, h, lengths, res=None):
        """h [b, t, c] channels-last -> conv_2(relu(conv_1(h * mask)) * mask) * mask (+ res)
        (attentions.py:277-293): mask-in / ReLU are prologues, mask-out / residual epilogues of the two convolutions."""
        from . import wn_cl
        pad = (self.kernel_size - 1) // 2
        y = wn_cl.conv_cl(h, wn_cl.weight_of(self.conv_1), wn_cl.bias_of(self.conv_1), lengths, pad=pad, mask_in=True)
        if self.training and self.p_dropout > 0:
            y = self.drop(torch.relu(y))
            y = wn_cl.conv_cl(y, wn_cl.weight_of(self.conv_2), wn_cl.bias_of(self.conv_2), lengths, pad=pad, mask_in=True, mask_out=True)
            return y if res is None else y + res
        if res is not None:     # (conv * mask) + res: res rows beyond the length are zero in the encoder, so masking the sum is exact
            return wn_cl.conv_cl(y, wn_cl.weight_of(self.conv_2), wn_cl.bias_of(self.conv_2), lengths, pad=pad, in_slope=0.0, mask_in=True, mask_out=True, res=res)
        return wn_cl.conv_cl(y, wn_cl.weight_of(self.conv_2), wn_cl.bias_of(self.conv_2), lengths, pad=pad, in_slope=0.0, mask_in=True, mask_out=True)
