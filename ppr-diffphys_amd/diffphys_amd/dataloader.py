"""AMP-format mocap loader and column map.

Same surface as the reference's loader (/root/reference/diffphys/dataloader.py:9-31):
``DataLoader(opts)`` exposes ``amp_info`` [frames, 85], ``frame_interval`` and
``data_info["offset"]``; ``parse_amp`` slices the AMP row; ``bullet2gl`` is the
axis change of /root/reference/diffphys/dp_utils.py:141-156.
"""
import json
import os

import numpy as np


class DataLoader:
    def __init__(self, opts, cap=-1, data_root="./data/motion_sequences"):
        seq = opts["seqname"]
        path = os.path.join(data_root, seq, "amp-%s.txt" % seq)
        if os.path.exists(path):
            with open(path, "r") as f:
                d = json.load(f)
            self.frame_interval = d["FrameDuration"]
            self.amp_info = np.asarray(d["Frames"])
        else:  # compiled fixture (scripts/compile_templates.py)
            from .robots import TEMPLATE_DIR

            with np.load(os.path.join(TEMPLATE_DIR, "mocap_laikago.npz")) as z:
                self.amp_info = z[seq + "/frames"]
                self.frame_interval = float(z[seq + "/frame_duration"])
        self.data_info = {"offset": np.asarray([0, len(self.amp_info)])}


def parse_amp(amp_info):
    return {
        "pos": amp_info[..., 0:3],
        "orn": amp_info[..., 3:7],
        "vel": amp_info[..., 31:34],
        "avel": amp_info[..., 34:37],
        "jang": amp_info[..., 7:19],
        "jvel": amp_info[..., 37:49],
        "kp": amp_info[..., 61:73],
        "kp_vel": amp_info[..., 73:85],
    }


_ISAAC_TO_GL = np.asarray([[0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [1.0, 0.0, 0.0]])


def bullet2gl(msm, in_bullet):
    """In place: (x,y,z) -> (y,z,x) on pos / quaternion vector part / vel / avel."""
    m = _ISAAC_TO_GL
    msm["pos"] = msm["pos"] @ m.T
    if in_bullet:
        from scipy.spatial.transform import Rotation as R

        shape = msm["orn"].shape[:-1]
        orn = R.from_quat(msm["orn"].reshape((-1, 4))).as_matrix()
        msm["orn"] = R.from_matrix(orn @ m).as_quat().reshape(shape + (4,))
    orn = np.array(msm["orn"], copy=True)
    orn[..., :3] = orn[..., :3] @ m.T
    msm["orn"] = orn
    msm["vel"] = msm["vel"] @ m.T
    msm["avel"] = msm["avel"] @ m.T


def mocap_tensors(table, steps_fr):
    """``bullet2gl(parse_amp(interp1d(arange(n), table, 'linear', fill_value='extrapolate')(steps_fr)))`` (not in_bullet) as
    torch ops on the device of ``steps_fr``: ``table`` is the AMP table as a float64 tensor [n, 85] on that device.  Linear
    inter/extrapolation in float64 like scipy (end segments extrapolate), column map of parse_amp, (x, y, z) -> (y, z, x).
    Returns float32 tensors keyed like parse_amp."""
    import torch

    # Column map (parse_amp) and axis change ((x, y, z) -> (y, z, x)) are applied to the TABLE, once per table: a column's interpolation does not
    # care where the column sits, so the row interpolated from the re-ordered table is the re-ordered interpolated row -- the same numbers, with
    # one gather pair, one lerp and one cast per query instead of four concatenations and eight casts.
    tab2 = getattr(table, "_pd_reordered", None)   # (kept ON the table tensor: lives and dies with it)
    if tab2 is None:
        yzx = lambda a: [a + 1, a + 2, a]
        cols = yzx(0) + yzx(3) + [6] + yzx(31) + yzx(34) + list(range(7, 19)) + list(range(37, 49)) + list(range(61, 73)) + list(range(73, 85))
        tab2 = table[:, torch.as_tensor(cols, device=table.device)].contiguous()
        try:
            table._pd_reordered = tab2
        except AttributeError:
            pass
    n = tab2.shape[0]
    x = steps_fr.detach().to(tab2.dtype)
    i0 = x.floor().clamp(0, n - 2).long()
    lo, hi = tab2[i0], tab2[i0 + 1]
    row = ((hi - lo) * (x - i0.to(tab2.dtype)).unsqueeze(-1) + lo).float()
    names = (("pos", 3), ("orn", 4), ("vel", 3), ("avel", 3), ("jang", 12), ("jvel", 12), ("kp", 12), ("kp_vel", 12))
    out, a = {}, 0
    for k, w in names:
        out[k] = row[..., a:a + w]
        a += w
    return out
