"""Time-conditioned MLPs that parameterise the control references (out of the hot path: 5 small MLPs on
<= bs*T scalars).  Same constructor surface as the reference's ``TimeMLPWrapper``
(/root/reference/diffphys/torch_utils.py:120-180): Fourier features of normalised time, a skip-connected
MLP trunk of width W and depth D, a linear head and an output scale."""
import math

import torch
from torch import nn


class _LinearGemmBias(torch.autograd.Function):
    """y = x W^T + b with the bias gradient taken by a GEMM (ones[1, N] @ g) instead of torch's column reduction.  Same forward as
    F.linear.  Why: on this stack (ROCm 7.x / torch 2.10) torch's multi-block reductions -- e.g. ``g.sum(0)`` over [1024, 128] -- come
    out STALE from the second replay of a captured HIP graph on (their semaphore memset is not re-executed; pure-torch reproducer
    scripts/micro/torch_graph_replay2.py), and the bias gradients of these MLPs are exactly that shape.  A GEMM has no such state, so
    an iteration captured by ``phys_model.capture_iteration`` replays bit for bit; the eager path runs the same code."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ weight if ctx.needs_input_grad[0] else None
        gw = _gemm_long_k(g.t(), x) if ctx.needs_input_grad[1] else None
        gb = (_ones_row(g.shape[0], g) @ g).reshape(-1) if ctx.needs_input_grad[2] else None
        return gx, gw, gb


_ONES = {}


def _ones_row(n, like):
    """ones[1, n] on like's device, made once per (n, device): the bias-gradient GEMM's left operand (a fill per layer and backward otherwise)"""
    key = (n, like.device, like.dtype)
    t = _ONES.get(key)
    if t is None:
        if len(_ONES) > 16:
            _ONES.clear()
        t = _ONES[key] = torch.ones(1, n, dtype=like.dtype, device=like.device)
    return t


def _gemm_long_k(a, b):
    """a [m, K] @ b [K, n] with K >> m, n (the weight gradient: K = samples): on MI355X / ROCm 7 rocBLAS has a 24 us kernel for
    256 x 7600 x 256 where hipBLASLt, torch's default, picks a 55 us one (scripts/micro/gemm_shapes.py) -- routed there when K is long.
    The choice depends on shapes only, so the eager and the captured iteration take the same kernel."""
    if a.is_cuda and a.shape[1] >= 2048 and torch.version.hip is not None:
        prev = torch.backends.cuda.preferred_blas_library()
        try:
            torch.backends.cuda.preferred_blas_library("hipblas")
            return a @ b
        finally:
            torch.backends.cuda.preferred_blas_library(prev)
    return a @ b


def _linear(layer, x):
    return _LinearGemmBias.apply(x, layer.weight, layer.bias)


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
        h = self.act(_linear(self.inp, e))
        for i, layer in enumerate(self.layers):
            h = self.act(_linear(layer, torch.cat([h, e], -1) if i in self.skips else h))
        return _linear(self.head, h) * self.output_scale


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
