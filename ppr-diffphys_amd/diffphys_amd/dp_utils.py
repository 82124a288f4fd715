"""Loss / frame utilities around the simulator boundary.
Same names and semantics as /root/reference/diffphys/dp_utils.py (cited per function)."""
import torch

from .dataloader import bullet2gl  # noqa: F401  (re-exported like the reference's dp_utils)
from .geom_utils import axis_angle_to_matrix, quaternion_invert, quaternion_to_matrix, rot_angle, se3_mat2vec, se3_vec2mat


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


def _hip_pose(*ts):
    """float32 GPU tensors take the HIP kernels (no silent fallback there: a missing library raises); anything else -- the
    CPU host tests, float64 checks -- runs the torch composition, which is also the kernels' test reference."""
    return all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 for t in ts)


def compose_delta(target_q, delta_root):
    """delta (bs,T,6 axis-angle) applied on the left of target (bs,T,7)   dp_utils.py:22-31"""
    if _hip_pose(target_q, delta_root) and target_q.shape[-1] == 7 and delta_root.shape[-1] == 6 and target_q.shape[:-1] == delta_root.shape[:-1]:
        return _PoseOpHip.apply(0, target_q, delta_root)
    return compose_delta_torch(target_q, delta_root)


def compose_delta_torch(target_q, delta_root):
    return se3_mat2vec(se3_vec2mat(delta_root) @ se3_vec2mat(target_q))


def remove_nan(t, bs=None, clip=False):
    """in place NaN -> 0 (and optional +-0.01 clip)   dp_utils.py:43-57"""
    t.masked_fill_(t.isnan(), 0)  # (the reference's t[t.isnan()] = 0 without its host synchronisation)
    if clip:
        t.clamp_(-0.01, 0.01)


def rotate_frame(global_q, target_q):
    """T = T_global @ T_target   dp_utils.py:60-73"""
    if _hip_pose(global_q, target_q) and global_q.shape == (7,) and target_q.shape[-1] == 7:
        return _PoseOpHip.apply(1, global_q, target_q)
    return rotate_frame_torch(global_q, target_q)


def rotate_frame_torch(global_q, target_q):
    gm = se3_vec2mat(global_q)
    if global_q.dim() == 1:
        gm = gm[None, None]
    return se3_mat2vec(gm @ se3_vec2mat(target_q), outdim=target_q.shape[-1])


def rotate_frame_vel(global_q, target_qd):
    """rotate (linear, angular) halves by the rotation of global_q   dp_utils.py:76-84"""
    if _hip_pose(global_q, target_qd) and global_q.shape == (7,) and target_qd.shape[-1] == 6:
        return _PoseOpHip.apply(2, global_q, target_qd)
    return rotate_frame_vel_torch(global_q, target_qd)


def rotate_frame_vel_torch(global_q, target_qd):
    gq = torch.cat([torch.zeros_like(global_q[..., :3]), global_q[..., 3:]], -1)
    rev = torch.cat([target_qd[..., 3:], target_qd[..., :3]], -1)
    return torch.cat([rotate_frame_torch(gq, target_qd)[..., :3], rotate_frame_torch(gq, rev)[..., :3]], -1)


def reduce_loss_loop(loss_seq, clip=False, th=0):
    """The reference's loop, one host synchronisation per env (kept as the test reference of reduce_loss)."""
    if clip:
        for i in range(len(loss_seq)):
            if th == 0:
                sub = loss_seq[i]
                pos = sub[sub > 0]
                th = pos.median() * 10 if pos.numel() > 0 else 0
            if th != 0:
                over = loss_seq[i] > th
                if bool(over.any()):
                    loss_seq[i, int(over.float().argmax()):] = 0
    if loss_seq.sum() > 0:
        return loss_seq[loss_seq > 0].mean()
    return loss_seq.mean()


def reduce_loss(loss_seq, clip=False, th=0):
    """(bs,T) -> scalar; with clip, a rollout's loss is zeroed from the first step that exceeds the threshold on -- the
    threshold being 10x the median of the positive entries of the FIRST env that has any (the reference computes it once
    and reuses it for every env)   dp_utils.py:93-110.  Same values and in-place effect as the reference's per-env loop,
    without its host synchronisations (~4 per env and iteration)."""
    if clip:
        pos = loss_seq > 0
        if not torch.is_tensor(th) and th == 0:
            has = pos.any(1)
            row = loss_seq.index_select(0, has.float().argmax().reshape(1))[0]
            rp = row > 0
            srt = torch.where(rp, row, torch.full_like(row, float("inf"))).sort().values
            med = srt.gather(0, ((rp.sum() - 1).clamp(min=0) // 2).reshape(1))[0]  # torch.median: the lower one
            th = torch.where(has.any(), med * 10, torch.full_like(med, float("inf"))).detach()
        keep = (loss_seq.detach() > th).cumsum(1) == 0
        loss_seq.masked_fill_(~keep, 0)  # assignment like the reference's loss_seq[i, idx:] = 0: an inf / NaN past the clip is zeroed, not 0 * inf
    pos = loss_seq > 0
    mean_pos = (loss_seq * pos).sum() / pos.sum().clamp(min=1)
    return torch.where(loss_seq.sum() > 0, mean_pos, loss_seq.mean())


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
    float32 GPU tensors take the fused HIP kernel (the library must be built: no silent fallback on the GPU); anything
    else (the CPU host tests, float64 checks) runs the torch composition below, which is also the kernel's test reference."""
    if pred.is_cuda and pred.dtype == torch.float32 and gt.dtype == torch.float32 and pred.shape == gt.shape:
        return _Se3LossHip.apply(pred, gt, rot_ratio)
    return se3_loss_torch(pred, gt, rot_ratio)


def se3_loss_torch(pred, gt, rot_ratio=0.1):
    nanid = torch.logical_or(pred.sum(-1).isnan(), gt.sum(-1).isnan())
    trn = (pred[..., :3] - gt[..., :3]).pow(2).sum(-1)
    rp, rg = pred[..., 3:], gt[..., 3:]
    if rp.shape[-1] == 3:
        rp, rgi = axis_angle_to_matrix(rp), axis_angle_to_matrix(rg).transpose(-1, -2)
    else:
        rp = quaternion_to_matrix(torch.cat([rp[..., 3:4], rp[..., 0:3]], -1))
        rgi = quaternion_to_matrix(quaternion_invert(torch.cat([rg[..., 3:4], rg[..., 0:3]], -1)))
    loss = trn + rot_angle(rp @ rgi) * rot_ratio
    return torch.where(nanid, torch.zeros_like(loss), loss)


def compute_com(body_q, part_com, part_mass):
    """mass-weighted COM of one articulation (numpy)   dp_utils.py:86-90"""
    from scipy.spatial.transform import Rotation as R

    c = (R.from_quat(body_q[:, 3:]).as_matrix() @ part_com)[..., 0] + body_q[:, :3]
    return (c * part_mass[:, None]).sum(0) / part_mass.sum()
