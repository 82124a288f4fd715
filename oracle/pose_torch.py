"""ORACLE (test infrastructure, not product code): torch restatements of the reference's loss / frame utilities around the
simulator boundary -- the checkers of the HIP pose / loss / foot-height kernels (csrc/pd_pose.hip, pd_loss.hip; SURVEY.md section 8
rows f2 / f4).  Any dtype, any device; the tests evaluate them in float64.  Follows
    /root/reference/diffphys/dp_utils.py:22-31    compose_delta
    /root/reference/diffphys/dp_utils.py:60-84    rotate_frame, rotate_frame_vel
    /root/reference/diffphys/dp_utils.py:93-110   reduce_loss (the per-env loop, literally)
    /root/reference/diffphys/dp_utils.py:113-138  se3_loss
    /root/reference/diffphys/geom_utils.py:36-45,148-203   rot_angle, se3_vec2mat, se3_mat2vec
    /root/reference/diffphys/dp_model.py:574-579  get_foot_height (on the contact candidates instead of posed visual meshes)
PARITY UNPINNED against the reference's own runs: dqtorch (its quaternion kernels, diffphys/geom_utils.py:5) is absent here; the
conventions are pytorch3d's (real part FIRST inside these helpers, LAST in the 7-vectors at the boundary) and are pinned to
scipy.spatial.transform.Rotation in tests/test_host_plumbing.py.  PINNED since round 5, by outputs of the reference's OWN code
(tests/golden/ref_host_*.npz, scripts/make_ref_fixtures.py, tests/test_ref_fixtures.py): reduce_loss_loop (33 tables), rot_angle,
quaternion_to_axis_angle, quaternion_invert, se3_vec2mat -- the functions of the reference that run here without dqtorch.  Only tests/ and
scripts/ import this module."""
import numpy as np
import torch


def quaternion_to_matrix(q):
    """(..., 4) real-first -> (..., 3, 3)"""
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack(
        (1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
         two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
         two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def _sqrt_pos(x):
    """sqrt(max(0, x)) with a zero subgradient where x <= 0 -- by selection, not by masked assignment (no nonzero(), hence no
    host synchronisation and capturable in a HIP graph)"""
    m = x > 0
    return torch.where(m, torch.sqrt(torch.where(m, x, torch.ones_like(x))), torch.zeros_like(x))


def matrix_to_quaternion(matrix):
    """(..., 3, 3) -> (..., 4) real-first; picks the best-conditioned of the four candidate forms."""
    batch = matrix.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(matrix.reshape(batch + (9,)), dim=-1)
    q_abs = _sqrt_pos(torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], -1))
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], -1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], -1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], -1)], -2)
    cand = cand / (2.0 * q_abs[..., None].clamp(min=0.1))
    best = q_abs.argmax(-1)  # gather instead of boolean-mask indexing: same row, no host synchronisation
    return cand.gather(-2, best[..., None, None].expand(batch + (1, 4))).squeeze(-2)


def axis_angle_to_quaternion(aa):
    ang = torch.norm(aa, p=2, dim=-1, keepdim=True)
    half = 0.5 * ang
    small = ang.abs() < 1e-6
    s = torch.where(small, 0.5 - ang * ang / 48, torch.sin(half) / torch.where(small, torch.ones_like(ang), ang))
    return torch.cat([torch.cos(half), aa * s], -1)


def axis_angle_to_matrix(vec):
    return quaternion_to_matrix(axis_angle_to_quaternion(vec))


def quaternion_to_axis_angle(q):
    norms = torch.norm(q[..., 1:], p=2, dim=-1, keepdim=True)
    half = torch.atan2(norms, q[..., :1])
    ang = 2 * half
    small = ang.abs() < 1e-6
    s = torch.where(small, 0.5 - ang * ang / 48, torch.sin(half) / torch.where(small, torch.ones_like(ang), ang))
    return q[..., 1:] / s


def quaternion_invert(q):
    return torch.cat([q[..., :1], -q[..., 1:]], -1)  # (no host-built constant: capturable in a HIP graph)


def rot_angle(mat):
    """rotation angle of (..., 3, 3), clamped like /root/reference/diffphys/geom_utils.py:36-45"""
    eps = 1e-4
    cos = (mat[..., 0, 0] + mat[..., 1, 1] + mat[..., 2, 2] - 1) / 2
    return torch.acos(cos.clamp(-1 + eps, 1 - eps))


def se3_vec2mat(vec):
    """(..., 7) (p, q real-last) or (..., 6) (p, axis-angle) -> (..., 4, 4)   (geom_utils.py:148-174)"""
    if not torch.is_tensor(vec):
        vec = torch.as_tensor(np.asarray(vec), dtype=torch.float32)
    if vec.shape[-1] == 6:
        rmat = axis_angle_to_matrix(vec[..., 3:6])
    else:
        rmat = quaternion_to_matrix(torch.cat([vec[..., 6:7], vec[..., 3:6]], -1))  # real-first; slices, not a host-built index
    top = torch.cat([rmat, vec[..., :3, None]], -1)  # assembled by concatenation (index assignment of Python scalars uploads them)
    z = torch.zeros_like(top[..., :1, :1])
    return torch.cat([top, torch.cat([z, z, z, z + 1], -1)], -2)


def se3_mat2vec(mat, outdim=7):
    """(..., 4, 4) -> (..., 7) real-last quaternion, or (..., 6) axis-angle   (geom_utils.py:187-203)"""
    quat = matrix_to_quaternion(mat[..., :3, :3])
    if outdim == 7:
        rot = torch.cat([quat[..., 1:4], quat[..., 0:1]], -1)
    elif outdim == 6:
        rot = quaternion_to_axis_angle(quat)
    else:
        raise ValueError("outdim must be 6 or 7")
    return torch.cat([mat[..., :3, 3], rot], -1)



def compose_delta(target_q, delta_root):
    """delta (bs,T,6 axis-angle) applied on the left of target (bs,T,7)   dp_utils.py:22-31"""
    return se3_mat2vec(se3_vec2mat(delta_root) @ se3_vec2mat(target_q))


def rotate_frame(global_q, target_q):
    """T = T_global @ T_target   dp_utils.py:60-73"""
    gm = se3_vec2mat(global_q)
    if global_q.dim() == 1:
        gm = gm[None, None]
    return se3_mat2vec(gm @ se3_vec2mat(target_q), outdim=target_q.shape[-1])


def rotate_frame_vel(global_q, target_qd):
    """rotate (linear, angular) halves by the rotation of global_q   dp_utils.py:76-84"""
    gq = torch.cat([torch.zeros_like(global_q[..., :3]), global_q[..., 3:]], -1)
    rev = torch.cat([target_qd[..., 3:], target_qd[..., :3]], -1)
    return torch.cat([rotate_frame(gq, target_qd)[..., :3], rotate_frame(gq, rev)[..., :3]], -1)


def se3_loss(pred, gt, rot_ratio=0.1):
    """|dp|^2 + rot_ratio * angle(R_pred R_gt^T), 0 where an input is NaN   dp_utils.py:113-138"""
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


def reduce_loss_loop(loss_seq, clip=False, th=0):
    """The reference's per-env loop, control flow as written there (dp_utils.py:93-110), one host synchronisation per env: the test
    reference of the product's synchronisation-free reduce_loss and of the one-workgroup kernel behind pd_rollout_forward_traj_loss.
    Pinned by the reference's own outputs (tests/golden/ref_host_reduce_loss.npz, tests/test_ref_fixtures.py).  NB the threshold is
    taken while ``th == 0`` holds, i.e. at env 0 only: a NaN threshold (env 0 without a positive entry: median of an empty selection)
    is not 0, is never replaced, and compares false with everything -- no env is clipped then."""
    if clip:
        for i in range(len(loss_seq)):
            if th == 0:
                row = loss_seq[i]
                th = row[row > 0].median() * 10
            hit, at = torch.max(loss_seq[i] > th, 0)  # first exceedance (the first maximal index)
            if hit == 1:
                loss_seq[i, at:] = 0
    if loss_seq.sum() > 0:
        return loss_seq[loss_seq > 0].mean()
    return loss_seq.mean()


def foot_height(state_body_q, c_body, c_point, c_dist):
    """lowest ground-contact candidate per pose set: min over candidates of  p_y + (R(q) point)_y - dist  (the quantity behind
    reg_foot, dp_model.py:574-579,762,814).  state_body_q (..., nb, 7); c_body long [nc]; c_point [nc, 3]; c_dist [nc]."""
    X = state_body_q[..., c_body, :]
    q, p = X[..., 3:], X[..., :3]
    qv, w = q[..., :3], q[..., 3:]
    pt = c_point.expand(qv.shape)
    rot = pt * (2 * w * w - 1) + 2 * w * torch.cross(qv, pt, dim=-1) + 2 * qv * (qv * pt).sum(-1, keepdim=True)
    return (p[..., 1] + rot[..., 1] - c_dist).min(-1)[0]
