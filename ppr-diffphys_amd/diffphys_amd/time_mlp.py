"""Time-conditioned MLPs that parameterise the control references (out of the hot path: 5 small MLPs on
<= bs*T scalars).  Same constructor surface as the reference's ``TimeMLPWrapper``
(/root/reference/diffphys/torch_utils.py:120-180): Fourier features of normalised time, a skip-connected
MLP trunk of width W and depth D, a linear head and an output scale."""
import math

import torch
from torch import nn


class TimeMLPWrapper(nn.Module):
    def __init__(self, num_frames, frame_info=None, D=5, W=256, num_freq_t=6, out_channels=1, skips=(1, 2, 3, 4),
                 activation=None, time_scale=1.0, output_scale=1.0):
        super().__init__()
        self.num_frames, self.time_scale, self.output_scale, self.skips, self.D = num_frames, time_scale, output_scale, set(skips), D
        self.register_buffer("freqs", 2.0 ** torch.arange(num_freq_t, dtype=torch.float32) * math.pi, persistent=False)
        in_ch = 1 + 2 * num_freq_t
        self.inp = nn.Linear(in_ch, W)
        self.layers = nn.ModuleList([nn.Linear(W + (in_ch if i in self.skips else 0), W) for i in range(D)])
        self.act = activation if activation is not None else nn.ReLU(True)
        self.head = nn.Linear(W, out_channels)
        gen = torch.Generator().manual_seed(8)  # the reference seeds here "to reproduce results"
        with torch.no_grad():
            for mod in self.modules():
                if isinstance(mod, nn.Linear):
                    bound = 1.0 / math.sqrt(mod.weight.shape[1])
                    mod.weight.copy_((torch.rand(mod.weight.shape, generator=gen) * 2 - 1) * bound)
                    mod.bias.copy_((torch.rand(mod.bias.shape, generator=gen) * 2 - 1) * bound)
            self.head.weight.mul_(0.1)
            self.head.bias.zero_()

    def embed(self, frame_id):
        t = (frame_id.float().reshape(-1, 1) / max(1, self.num_frames - 1) * 2 - 1) * self.time_scale
        x = t * self.freqs.to(t.device)
        return torch.cat([t, torch.sin(x), torch.cos(x)], -1)

    def forward(self, frame_id=None):
        if frame_id is None:
            frame_id = torch.arange(self.num_frames, device=self.head.weight.device)
        e = self.embed(frame_id)
        h = self.act(self.inp(e))
        for i, layer in enumerate(self.layers):
            h = self.act(layer(torch.cat([h, e], -1) if i in self.skips else h))
        return self.head(h) * self.output_scale


def interp_wt(x, y, x2, type="linear"):
    """piecewise-linear (or log-linear) schedule between two anchors   (lab4d_utils.py:622-660 of the reference)"""
    assert len(x) == 2 and len(y) == 2
    if x2 <= x[0]:
        return y[0]
    if x2 >= x[1]:
        return y[1]
    a = (x2 - x[0]) / (x[1] - x[0])
    if type == "log":
        return math.exp(math.log(y[0]) * (1 - a) + math.log(y[1]) * a)
    return y[0] * (1 - a) + y[1] * a


def match_param_name(name, param_lr, type):
    """how many keys of param_lr match `name` ("with" = substring, "startwith" = prefix) and the matched value"""
    matched, lr = 0, 0.0
    for k, v in param_lr.items():
        if (type == "with" and k in name) or (type == "startwith" and name.startswith(k)):
            matched, lr = matched + 1, v
    return matched, lr
