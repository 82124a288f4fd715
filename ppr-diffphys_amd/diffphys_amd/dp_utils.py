"""Loss / frame utilities around the simulator boundary.
Same names and semantics as /root/reference/diffphys/dp_utils.py (cited per function)."""
import torch

from .dataloader import bullet2gl  # noqa: F401  (re-exported like the reference's dp_utils)


class _PoseOpHip(torch.autograd.Function):
    """One of the three pose compositions below as ONE HIP launch, its vector-Jacobian product as one more (C ABI
    ``pd_pose_op`` / ``pd_pose_op_vjp``; the Jacobian is taken inside the kernel by forward-mode differentiation of the
    code that computed the value)."""

    @staticmethod
    def forward(ctx, op, a, b):
        from . import hip_backend

        a, b = a.detach().contiguous(), b.detach().contiguous()
        ctx.op = op
        ctx.save_for_backward(a, b)
        return hip_backend.pose_op(op, a, b)

    @staticmethod
    def backward(ctx, g):
        from . import hip_backend

        a, b = ctx.saved_tensors
        need_a, need_b = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        g_a, g_b = hip_backend.pose_op_vjp(ctx.op, a, b, g.contiguous(), need_a, need_b)
        return None, g_a, g_b


def _need_gpu(what, *ts):
    """The pose algebra and se3_loss run as HIP kernels and nowhere else: float32 GPU tensors, or an error (no CPU / torch fallback;
    their float64 torch restatement is test infrastructure: oracle/pose_torch.py)."""
    for t in ts:
        if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32):
            raise TypeError("%s needs float32 GPU tensors (the product path has no CPU fallback); got %s" % (
                what, "%s on %s" % (t.dtype, t.device) if torch.is_tensor(t) else type(t).__name__))


def compose_delta(target_q, delta_root):
    """delta (bs,T,6 axis-angle) applied on the left of target (bs,T,7)   dp_utils.py:22-31"""
    _need_gpu("compose_delta", target_q, delta_root)
    if not (target_q.shape[-1] == 7 and delta_root.shape[-1] == 6 and target_q.shape[:-1] == delta_root.shape[:-1]):
        raise ValueError("compose_delta: target (..., 7) and delta (..., 6) with equal leading shapes; got %s and %s" % (tuple(target_q.shape), tuple(delta_root.shape)))
    return _PoseOpHip.apply(0, target_q, delta_root)


def remove_nan(t, bs=None, clip=False):
    """in place NaN -> 0 (and optional +-0.01 clip)   dp_utils.py:43-57"""
    t.masked_fill_(t.isnan(), 0)  # (the reference's t[t.isnan()] = 0 without its host synchronisation)
    if clip:
        t.clamp_(-0.01, 0.01)


def rotate_frame(global_q, target_q):
    """T = T_global @ T_target   dp_utils.py:60-73"""
    _need_gpu("rotate_frame", global_q, target_q)
    if not (global_q.shape == (7,) and target_q.shape[-1] == 7):
        raise ValueError("rotate_frame: one global pose (7,) against targets (..., 7); got %s and %s" % (tuple(global_q.shape), tuple(target_q.shape)))
    return _PoseOpHip.apply(1, global_q, target_q)


def rotate_frame_vel(global_q, target_qd):
    """rotate (linear, angular) halves by the rotation of global_q   dp_utils.py:76-84"""
    _need_gpu("rotate_frame_vel", global_q, target_qd)
    if not (global_q.shape == (7,) and target_qd.shape[-1] == 6):
        raise ValueError("rotate_frame_vel: one global pose (7,) against twists (..., 6); got %s and %s" % (tuple(global_q.shape), tuple(target_qd.shape)))
    return _PoseOpHip.apply(2, global_q, target_qd)


def reduce_loss(loss_seq, clip=False, th=0):
    """(bs,T) -> scalar; with clip, a rollout's loss is zeroed (in place) from the first step that exceeds the threshold on -- the
    threshold being 10x the median of the positive entries of ENV 0: the reference computes it at i == 0 and reuses it for every env
    (dp_utils.py:93-110).  When env 0 has no positive entry the reference's median of an empty selection is NaN, ``th == 0`` is never
    true again and ``loss > NaN`` is false: nothing is clipped in the whole batch -- reproduced here (held to the reference's own outputs,
    tests/golden/ref_host_reduce_loss.npz).  Same values and in-place effect as the reference's per-env loop without its host
    synchronisations (~4 per env and iteration)."""
    if clip and loss_seq.shape[0] > 0:
        row = loss_seq.detach()[0]
        rp = row > 0
        srt = torch.where(rp, row, torch.full_like(row, float("inf"))).sort().values
        med = srt.gather(0, ((rp.sum() - 1).clamp(min=0) // 2).reshape(1))[0]  # torch.median: the lower one
        th0 = torch.where(rp.any(), med * 10, torch.full_like(med, float("nan")))
        if torch.is_tensor(th):
            th = torch.where(th.detach().to(th0) == 0, th0, th.detach().to(th0))
        elif th == 0:
            th = th0
        keep = (loss_seq.detach() > th).cumsum(1) == 0
        loss_seq.masked_fill_(~keep, 0)  # assignment like the reference's loss_seq[i, idx:] = 0: an inf / NaN past the clip is zeroed, not 0 * inf
    pos = loss_seq > 0
    mean_pos = torch.where(pos, loss_seq, torch.zeros_like(loss_seq)).sum() / pos.sum().clamp(min=1)
    return torch.where(loss_seq.sum() > 0, mean_pos, loss_seq.mean())


class _ReduceLossHip(torch.autograd.Function):
    """reduce_loss WITHOUT clipping on a float32 GPU table (bs, F), as the library's one-workgroup launch (``pd_reduce_loss``, the code the
    trajectory-loss entries run; held to the reference's own outputs in tests/test_ref_fixtures.py) and one multiply for the gradient
    (scale = d value / d entry, written by the same launch) -- the torch composition above is ~15 launches forward and ~8 backward."""

    @staticmethod
    def forward(ctx, table):
        from . import hip_backend

        reduced, scale = hip_backend.reduce_loss(table.detach().contiguous(), clip=False, want_scale=table.requires_grad)
        if table.requires_grad:
            ctx.save_for_backward(scale)
        return reduced[0]

    @staticmethod
    def backward(ctx, g):
        (scale,) = ctx.saved_tensors
        return g * scale


def reduce_loss_masked(loss_seq, mask):
    """reduce_loss(loss_seq with the entries under `mask` set to 0) -- the two state losses of phys_model.forward (dp_model.py:783-805 of the
    reference: `loss[outseq_idx] = 0` then reduce_loss).  float32 GPU tables take the HIP launch, anything else the torch composition."""
    x = loss_seq.masked_fill(mask, 0.0)
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 2:
        return _ReduceLossHip.apply(x)
    return reduce_loss(x)


class _Se3LossHip(torch.autograd.Function):
    """se3_loss and both gradients in one HIP launch (C ABI ``pd_se3_loss``)."""

    @staticmethod
    def forward(ctx, pred, gt, rot_ratio):
        from . import hip_backend

        need = pred.requires_grad or gt.requires_grad
        loss, gp, gg = hip_backend.se3_loss(pred.detach().contiguous(), gt.detach().contiguous(), rot_ratio, want_grads=need)
        if need:
            ctx.save_for_backward(gp, gg)
        return loss

    @staticmethod
    def backward(ctx, g):
        gp, gg = ctx.saved_tensors
        g = g.unsqueeze(-1)
        return (g * gp if ctx.needs_input_grad[0] else None), (g * gg if ctx.needs_input_grad[1] else None), None


def se3_loss(pred, gt, rot_ratio=0.1):
    """|dp|^2 + rot_ratio * angle(R_pred R_gt^T); quaternion (real-last) or axis-angle rotations   dp_utils.py:113-138.
    One HIP launch for the value and both gradients (``pd_se3_loss``); float32 GPU tensors only."""
    _need_gpu("se3_loss", pred, gt)
    if pred.shape != gt.shape:
        raise ValueError("se3_loss: pred and gt must have the same shape; got %s and %s" % (tuple(pred.shape), tuple(gt.shape)))
    return _Se3LossHip.apply(pred, gt, rot_ratio)


def compute_com(body_q, part_com, part_mass):
    """mass-weighted COM of one articulation (numpy)   dp_utils.py:86-90"""
    from scipy.spatial.transform import Rotation as R

    c = (R.from_quat(body_q[:, 3:]).as_matrix() @ part_com)[..., 0] + body_q[:, :3]
    return (c * part_mass[:, None]).sum(0) / part_mass.sum()
