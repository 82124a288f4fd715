"""Time-conditioned MLPs that parameterise the control references (out of the hot path: 5 small MLPs on <= bs*T scalars): the reference's
``TimeMLPWrapper`` (/root/reference/diffphys/torch_utils.py:116-180) with its lab4d building blocks (diffphys/lab4d_utils.py: PosEmbedding,
TimeEmbedding, InstEmbedding, BaseMLP, ScaleLayer) restated -- same architecture, parameter names, construction order and initialisation, so
that the reference's checkpoints load and its initial weights are reproduced; held to the reference's own module by
tests/test_ref_fixtures.py."""
import math

import torch
from torch import nn


def _weight_bias_grads(g, x, need_w, need_b):
    """(g^T x, column sums of g) for a linear layer's backward.  GPU: the weight gradient as its own GEMM -- a contiguous [out, in] result
    that AccumulateGrad takes over without a copy -- and the bias gradient by the library's column-sum launch (hip_backend.colsum,
    ``pd_colsum``: ~5 us, fixed order).  Round 5 first folded the bias into the GEMM (g^T [x | 1]): a concatenation in front and two strided
    copies behind it per layer, 13 us x 30 layers of an iteration.  CPU (tests, fixtures): the GEMM forms."""
    gw = gb = None
    if g.is_cuda:
        from . import hip_backend

        if need_w and g.dtype == torch.float32 and x.is_contiguous():
            # 256-wide layers (27 of the 30 per iteration): both gradients from the library's matrix-core kernel, ~23 / 31 us against 30 / 46
            both = hip_backend.linear_wgrad(g, x, want_bias=need_b)
            if both is not None:
                return both
        if need_w:
            gw = _gemm_long_k(g.t(), x)
        if need_b:
            gb = hip_backend.colsum(g)
        return gw, gb
    if need_w and need_b:
        gwb = g.t() @ torch.cat([x, _ones_row(x.shape[0], x).t()], 1)
        return gwb[:, :-1], gwb[:, -1]
    if need_w:
        gw = g.t() @ x
    if need_b:
        gb = (_ones_row(g.shape[0], g) @ g).reshape(-1)
    return gw, gb


class _LinearGemmBias(torch.autograd.Function):
    """y = x W^T + b with the bias gradient taken by the library's column-sum launch (a GEMM on the CPU) instead of torch's column reduction.  Same forward as
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
        gw, gb = _weight_bias_grads(g, x, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return gx, gw, gb


class _LinearReluGemmBias(torch.autograd.Function):
    """relu(x W^T + b) with the ReLU in the GEMM's epilogue (torch._addmm_activation: hipBLASLt's RELU_BIAS epilogue on the GPU -- the same
    bits as addmm followed by relu, one launch less per layer: scripts/micro/relu_epilogue_probe.py) and _LinearGemmBias's backward behind
    the ReLU's mask (threshold_backward on the saved output, the kernel nn.ReLU's backward runs)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        out = torch._addmm_activation(bias, x, weight.t())
        ctx.save_for_backward(x, weight, out)
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight, out = ctx.saved_tensors
        g = torch.ops.aten.threshold_backward(g, out, 0).contiguous()   # (dense already on the current path; a permuted upstream gradient must not raise in hip_backend)
        gx = g @ weight if ctx.needs_input_grad[0] else None
        gw, gb = _weight_bias_grads(g, x, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return gx, gw, gb


_ONES = {}


def _ones_row(n, like):
    """ones[1, n] on like's device, made once per (n, device): the bias-gradient GEMM's left operand (a fill per layer and backward otherwise)"""
    key = (n, like.device, like.dtype)
    t = _ONES.get(key)
    if t is None:  # never evicted: a captured iteration reads these BY ADDRESS (one entry per distinct sample count: a handful, n floats each)
        t = _ONES[key] = torch.ones(1, n, dtype=like.dtype, device=like.device)
    return t


def _gemm_long_k(a, b):
    """a [m, K] @ b [K, n] with K >> m, n (the weight gradient of a layer that is NOT 128-aligned: 3 of the 30 per iteration; the aligned ones
    are the library's `pd_linear_wgrad`).  torch's default BLAS.  Round 5 flipped the PROCESS-GLOBAL `preferred_blas_library` to rocBLAS
    around this product (24 us against hipBLASLt's 55 at 256 x 7600 x 256) -- from autograd's backward thread, a data race with any other
    thread issuing GEMMs: dropped; the shapes left here are small-output heads where the two libraries are within
    a few us of each other."""
    return a @ b


def _linear(layer, x):
    return _LinearGemmBias.apply(x, layer.weight, layer.bias)


def _linear_act(seq, x):
    """seq = Sequential(Linear, activation): the ReLU (the default, as in the reference) rides in the GEMM; any other activation follows it"""
    if type(seq[1]) is nn.ReLU and x.dim() == 2:
        return _LinearReluGemmBias.apply(x, seq[0].weight, seq[0].bias)
    return seq[1](_linear(seq[0], x))


class ScaleLayer(nn.Module):
    """x * scale, the scale a persistent buffer (lab4d_utils.py:321-327 of the reference: it is in the checkpoints)"""

    def __init__(self, scale):
        super().__init__()
        self.register_buffer("scale", torch.tensor([float(scale)], dtype=torch.float32))

    def forward(self, x):
        return x * self.scale


class _InstCode(torch.autograd.Function):
    """One instance: every sample reads row 0 of the embedding table.  Forward = an expand; the gradient of the row = the column sums of
    the upstream gradient, as ONE GEMM (nn.Embedding's backward and torch's column reduction do not survive a HIP-graph replay here)."""

    @staticmethod
    def forward(ctx, weight, n):
        ctx.n = n
        return weight[0].unsqueeze(0).expand(n, -1)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        if g.is_cuda:
            from . import hip_backend
            return hip_backend.colsum(g)[None], None
        return _ones_row(g.shape[0], g) @ g, None


class PosEmbedding(nn.Module):
    """x -> (x, sin(2^0 x), cos(2^0 x), sin(2^1 x), cos(2^1 x), ...) per input channel, frequency-major, sin before cos
    (lab4d_utils.py:11-110 of the reference; its annealing window `alpha` stays -1 = off in this code base but is a persistent buffer)."""

    def __init__(self, in_channels, N_freqs):
        super().__init__()
        self.N_freqs, self.in_channels = N_freqs, in_channels
        self.out_channels = in_channels * (2 * N_freqs + 1)
        self.register_buffer("freq_bands", 2 ** torch.linspace(0, N_freqs - 1, N_freqs), persistent=False)
        self.register_buffer("alpha", torch.tensor(-1.0, dtype=torch.float32))

    def forward(self, x):  # (B, in_channels)
        sig = self.freq_bands[None, :, None] * x[:, None, :]                      # (B, N_freqs, in_channels)
        bands = torch.stack([torch.sin(sig), torch.cos(sig)], 2)                  # (B, N_freqs, 2, in_channels)
        return torch.cat([x, bands.reshape(x.shape[0], -1)], -1)


class InstEmbedding(nn.Module):
    """a learnable code per video (lab4d_utils.py:253-318); one video here -> every sample gets row 0"""

    def __init__(self, num_inst, inst_channels):
        super().__init__()
        self.num_inst, self.out_channels = num_inst, inst_channels
        self.mapping = nn.Embedding(num_inst, inst_channels)

    def forward(self, inst_id):
        if self.num_inst == 1:
            return _InstCode.apply(self.mapping.weight, inst_id.shape[0])
        return self.mapping(inst_id)


class TimeEmbedding(nn.Module):
    """frame id -> W features: Fourier features of the frame's normalised time inside its video through `mapping1`, concatenated with
    the video's instance code, through `mapping2` (lab4d_utils.py:137-229).  Time of frame f of a video [s, e): ((f - s) - (e - s) / 2) /
    max_video_length * 2 * time_scale; fractional frame ids keep their fraction (the table lookups use floor(f))."""

    def __init__(self, num_freq_t, frame_info, out_channels=128, time_scale=1.0):
        super().__init__()
        self.fourier_embedding = PosEmbedding(1, num_freq_t)
        off = torch.as_tensor(frame_info["frame_offset_raw"]).long()
        self.frame_offset, self.frame_offset_raw = frame_info["frame_offset"], frame_info["frame_offset_raw"]
        self.num_vids = len(off) - 1
        self.max_ts = float((off[1:] - off[:-1]).max())
        self.time_scale = time_scale
        fid = torch.arange(0, int(off[-1]))
        vid = torch.zeros_like(fid)
        for i in range(self.num_vids):
            vid = torch.where((fid >= off[i]) & (fid < off[i + 1]), torch.full_like(vid, i), vid)
        self.register_buffer("raw_fid_to_vid", vid, persistent=False)
        self.register_buffer("raw_fid_to_vstart", off[vid], persistent=False)
        self.register_buffer("raw_fid_to_vidlen", off[vid + 1] - off[vid], persistent=False)
        self.inst_embedding = InstEmbedding(self.num_vids, inst_channels=out_channels)
        self.mapping1 = nn.Linear(self.fourier_embedding.out_channels, out_channels)
        self.mapping2 = nn.Linear(2 * out_channels, out_channels)

    def frame_to_tid(self, frame_id, shared=None):
        # ((f - start) - length / 2) / max_ts * 2 is the same for every MLP over the same clip: `shared` (a dict that lives for one
        # phys_model.get_net_pred call) lets the five MLPs compute it once -- the same operations in the same order, then x time_scale
        key = ("tid", self.max_ts, tuple(int(x) for x in self.frame_offset_raw))
        base = shared.get(key) if shared is not None else None
        if base is None:
            k = frame_id.long()
            tid_sub = frame_id - self.raw_fid_to_vstart[k]
            base = (tid_sub - self.raw_fid_to_vidlen[k] / 2) / self.max_ts * 2
            if shared is not None:
                shared[key] = base
        return base * self.time_scale

    def forward(self, frame_id, shared=None):
        frame_id = frame_id.reshape(-1)
        fkey = ("fourier", self.max_ts, tuple(int(x) for x in self.frame_offset_raw), self.time_scale, self.fourier_embedding.N_freqs)
        feat = shared.get(fkey) if shared is not None else None
        if feat is None:   # (no gradient flows into the features: MLPs with the same time scale and frequency count share them)
            feat = self.fourier_embedding(self.frame_to_tid(frame_id, shared)[:, None].float())
            if shared is not None:
                shared[fkey] = feat
        coeff = _linear(self.mapping1, feat)
        # one video: every sample reads row 0 of the instance table (_InstCode needs the count only, no lookup)
        inst = self.inst_embedding(frame_id if self.num_vids == 1 else self.raw_fid_to_vid[frame_id.long()])
        return _linear(self.mapping2, torch.cat([coeff, inst], -1))


class TimeMLPWrapper(nn.Module):
    """The reference's ``TimeMLPWrapper`` (torch_utils.py:116-180 over lab4d_utils.TimeMLP / BaseMLP / TimeEmbedding): SAME modules in the SAME
    construction order with the SAME parameter names -- ``time_embedding.{mapping1, mapping2, inst_embedding.mapping}``, ``linear_1 ..
    linear_D`` (each ``Sequential(Linear, act)``: keys ``linear_k.0.*``), ``linear_final.0``, ``head.0`` / ``head.1.scale`` -- so that a
    checkpoint of the reference loads with strict=True, the default initialisation consumes torch's generator exactly as the reference's
    does (identical initial weights, incl. the re-seeding at the end of the constructor), and the outputs agree with the reference's OWN
    module to fp32 round-off (tests/golden/ref_host_timemlp.npz, made by scripts/make_ref_fixtures.py from the reference's code).
    Skip connections concatenate the time embedding in FRONT of the features (BaseMLP.forward); the number of Fourier frequencies follows
    the clip length (num_freq_t + log2(max_video_length / 64), rounded).  What differs is how it runs: every Linear goes through
    _LinearGemmBias (bias gradient by a GEMM), the one-video instance code through _InstCode -- both so that a captured iteration
    replays correctly (phys_model.capture_iteration)."""

    def __init__(self, num_frames, frame_info=None, D=5, W=256, num_freq_t=6, out_channels=1, skips=(1, 2, 3, 4),
                 activation=None, time_scale=1.0, output_scale=1.0):
        super().__init__()
        import numpy as np

        if frame_info is None:
            frame_info = {"frame_offset": np.asarray([0, num_frames]), "frame_mapping": list(range(num_frames)),
                          "frame_offset_raw": np.asarray([0, num_frames])}
        act = activation if activation is not None else nn.ReLU(True)
        self.num_frames, self.D, self.W, self.skips = num_frames, D, W, list(skips)
        if num_freq_t > 0:  # lab4d_utils.py:423-431: scale the frequency count with the clip length (64 frames -> num_freq_t)
            off = np.asarray(frame_info["frame_offset"])
            num_freq_t = int(np.rint(np.log2((off[1:] - off[:-1]).max() / 64) + num_freq_t))
        # construction order = the reference's (BaseMLP layers, then the time embedding, then the head): the initial weights are a
        # function of torch's generator state at entry, as there
        for i in range(D):
            layer = nn.Linear(W, W) if i == 0 else (nn.Linear(W + W, W) if i in self.skips else nn.Linear(W, W))
            setattr(self, "linear_%d" % (i + 1), nn.Sequential(layer, act))
        self.linear_final = nn.Sequential(nn.Linear(W, W), act)
        self.time_embedding = TimeEmbedding(num_freq_t, frame_info, out_channels=W, time_scale=time_scale)
        self.head = nn.Sequential(nn.Linear(W, out_channels), ScaleLayer(output_scale))
        torch.manual_seed(8)  # "to reproduce results" (torch_utils.py:160-161)
        if torch.cuda.is_available():
            torch.cuda.manual_seed(1)

    def forward(self, frame_id=None, shared=None):
        """``shared``: a dict several MLPs evaluated at the SAME frame ids pass to each other (phys_model.get_net_pred): the frame -> time
        mapping and the Fourier features are computed once per distinct (clip, time scale, frequency count)."""
        if frame_id is None:
            frame_id = torch.arange(self.num_frames, device=self.head[0].weight.device)
        x = self.time_embedding(frame_id, shared)
        out = x
        for i in range(self.D):
            if i in self.skips:
                out = torch.cat([x, out], -1)
            out = _linear_act(getattr(self, "linear_%d" % (i + 1)), out)
        out = _linear_act(self.linear_final, out)
        return self.head[1](_linear(self.head[0], out))


def interp_wt(x, y, x2, type="linear"):
    """map x2 from [x0, x1] to [y0, y1]: linear, log (in y) or exp (log in x, x clipped to the range first); the result is clipped to the
    range of y   (lab4d_utils.py:622-672 of the reference; held to its outputs in tests/test_ref_fixtures.py)"""
    import numpy as np

    (x0, x1), (y0, y1) = x, y
    if type == "linear":
        y2 = y0 + (x2 - x0) * (y1 - y0) / (x1 - x0)
    elif type == "log":
        y2 = 10 ** (np.log10(y0) + (x2 - x0) * (np.log10(y1) - np.log10(y0)) / (x1 - x0))
    elif type == "exp":
        assert x0 >= 1 and x1 >= 1
        x2 = np.clip(x2, x0, x1)
        y2 = y0 + (np.log10(x2) - np.log10(x0)) * (y1 - y0) / (np.log10(x1) - np.log10(x0))
    else:
        raise ValueError("interpolation_type must be 'linear' or 'log'")
    return np.clip(y2, np.min(y), np.max(y))


def match_param_name(name, param_lr, type):
    """(matched, lr): does a key of param_lr match `name` ("with" = substring, "startwith" = prefix), and its value; more than one matching
    key is an error, as in the reference   (lab4d_utils.py:587-619)"""
    if type not in ("with", "startwith"):
        raise ValueError("type not found")
    hits = [(k, v) for k, v in param_lr.items() if (k in name if type == "with" else name.startswith(k))]
    if len(hits) > 1:
        raise ValueError("multiple matches found", [k for k, _ in hits])
    return (True, hits[0][1]) if hits else (False, 0.0)
