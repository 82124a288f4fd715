"""SE(3) helpers of the loss plumbing, pure torch.

The reference takes ``quaternion_to_matrix`` / ``matrix_to_quaternion`` /
``axis_angle_to_quaternion`` from dqtorch, a CUDA-only extension
(/root/reference/diffphys/geom_utils.py:5, requirements.txt:16); these are
stand-ins with the same conventions (pytorch3d: real part FIRST).  The 7-vectors
at the simulator boundary are (p, q) with the real part LAST.
"""
import numpy as np
import torch
import torch.nn.functional as F


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


def fid_reindex(fid, num_vids, vid_offset):
    """absolute frame id -> (video id, relative id in [-1, 1])   (geom_utils.py:47-65)"""
    tid = torch.zeros_like(fid).float()
    vid = torch.zeros_like(fid)
    max_ts = (vid_offset[1:] - vid_offset[:-1]).max()
    for i in range(num_vids):
        assign = torch.logical_and(fid >= vid_offset[i], fid < vid_offset[i + 1])
        doffset = vid_offset[i + 1] - vid_offset[i]
        # torch.where instead of the reference's masked assignment: same values, no nonzero() -> no host synchronisation
        vid = torch.where(assign, torch.full_like(vid, i), vid)
        tid = torch.where(assign, (fid.float() - vid_offset[i] - doffset / 2) / max_ts * 2, tid)
    return vid, tid
