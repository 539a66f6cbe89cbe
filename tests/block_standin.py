"""TEST / MEASUREMENT INFRASTRUCTURE: a stand-in with the attribute surface of the reference's FNOBlocks
(neuralop/layers/fno_block.py:163-240, 377-414) for boxes where /root/reference does not exist (the GPU tier): the
same sub-modules under the same names (convs, fno_skips, channel_mlp[i].fcs, channel_mlp_skips), default
configuration (linear fno skip, soft-gating MLP skip, ChannelMLP with expansion 0.5, GELU, post-activation, no norm),
and a forward that is the reference's op sequence restated.  tests/test_fused_block.py checks the REAL class against
the fused path on the CPU tier; this one lets the GPU tier and scripts/block_time.py compare fused and unfused."""
import torch
import torch.nn.functional as F
from torch import nn


class Flattened1dConv(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, 1, bias=False)

    def forward(self, x):
        s = list(x.shape)
        return self.conv(x.reshape(s[0], s[1], -1)).reshape(s[0], -1, *s[2:])


class SoftGating(nn.Module):
    def __init__(self, c, n_dim):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(1, c, *(1,) * n_dim))
        self.bias = None

    def forward(self, x):
        return self.weight * x


class ChannelMLP(nn.Module):
    def __init__(self, c, hidden):
        super().__init__()
        self.fcs = nn.ModuleList([nn.Conv1d(c, hidden, 1), nn.Conv1d(hidden, c, 1)])
        self.non_linearity = F.gelu
        self.dropout = None

    def forward(self, x):
        s = list(x.shape)
        x = self.fcs[1](F.gelu(self.fcs[0](x.reshape(s[0], s[1], -1))))
        return x.reshape(s[0], -1, *s[2:])


class Blocks(nn.Module):
    def __init__(self, channels, n_modes, n_layers=2, expansion=0.5, preactivation=False):
        super().__init__()
        from neuraloperator_amd import SpectralConv
        nd = len(n_modes)
        self.n_layers, self.non_linearity = n_layers, F.gelu
        self.preactivation, self.norm, self.stabilizer, self.complex_data, self.use_channel_mlp = bool(preactivation), None, None, False, True
        self.convs = nn.ModuleList([SpectralConv(channels, channels, n_modes) for _ in range(n_layers)])
        self.fno_skips = nn.ModuleList([Flattened1dConv(channels, channels) for _ in range(n_layers)])
        self.channel_mlp = nn.ModuleList([ChannelMLP(channels, int(round(channels * expansion))) for _ in range(n_layers)])
        self.channel_mlp_skips = nn.ModuleList([SoftGating(channels, nd) for _ in range(n_layers)])

    def forward(self, x, index=0, output_shape=None):                     # fno_block.py:377-414, defaults
        if self.preactivation:                                            # fno_block.py:416-458, no normalisation layers
            x = F.gelu(x)
            x_skip_fno = self.fno_skips[index](x)
            x_skip_mlp = self.channel_mlp_skips[index](x)
            x = self.convs[index](x) + x_skip_fno
            if index < self.n_layers - 1:
                x = F.gelu(x)
            return self.channel_mlp[index](x) + x_skip_mlp
        x_skip_fno = self.fno_skips[index](x)
        x_skip_mlp = self.channel_mlp_skips[index](x)
        x = self.convs[index](x) + x_skip_fno
        if index < self.n_layers - 1:
            x = F.gelu(x)
        x = self.channel_mlp[index](x) + x_skip_mlp
        if index < self.n_layers - 1:
            x = F.gelu(x)
        return x
