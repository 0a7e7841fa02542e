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
        attn_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
        x = x * x_mask
        for i in range(self.n_layers):
            y = self.drop(self.attn_layers[i](x, x, attn_mask))
            x = self.norm_layers_1[i](x + y)
            y = self.drop(self.ffn_layers[i](x, x_mask))
            x = self.norm_layers_2[i](x + y)
        return x * x_mask


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
