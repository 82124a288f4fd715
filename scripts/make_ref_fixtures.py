#!/usr/bin/env python3
"""OUTPUTS OF THE REFERENCE'S OWN CODE for the host functions either side of the hot path.

Runs in the BUILD CONTAINER only (needs /root/reference; nothing from it is copied, and nothing on the GPU box reads it).  The
reference's rollout itself cannot run here (`warp_lang==0.7.2` is not installable), but these functions are plain torch / numpy /
scipy and only their modules' top-level imports of absent packages stop them from loading.  So: EMPTY placeholder modules are
registered for those imports (`dqtorch`, `cv2`, `trimesh`) -- attribute-less `types.ModuleType` objects -- and ONLY functions that
never touch a placeholder are called; a function that did would raise AttributeError, i.e. it cannot silently compute something.

    diffphys.dp_utils     reduce_loss, remove_nan, bullet2gl, compute_com, parse_rtk, project_bodies
    diffphys.geom_utils   rot_angle, fid_reindex, quaternion_to_axis_angle, quaternion_invert, se3_vec2mat (numpy branch), se3_mat2rt
    diffphys.dataloader   DataLoader, parse_amp  (on the five AMP files the reference ships)
    diffphys.torch_utils  TimeMLPWrapper (over diffphys.lab4d_utils: the time-MLPs phys_model is made of) -- initial weights under a fixed
                          seed, outputs and parameter gradients at fractional frame ids, one and two videos

Writes tests/golden/ref_host_*.npz : inputs and what the reference returned / left in place.  torch version recorded (median of an
empty selection is NaN on torch 2.x and the reference's reduce_loss relies on whatever torch does there).

    python scripts/make_ref_fixtures.py
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PPR_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")   # (--out DIR writes elsewhere: tests/test_reference_text.py regenerates and compares)
if "--out" in sys.argv:
    OUT = sys.argv[sys.argv.index("--out") + 1]
SEQS = ("mi-pace", "mi-trot", "mi-spin", "mi-turn", "mi-sidesteps")


def import_reference():
    for name in ("dqtorch", "cv2", "trimesh"):  # absent here; imported at module level, used by none of the functions called below
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                sys.modules[name] = types.ModuleType(name)
    sys.path.insert(0, REF)
    import diffphys.dataloader as rdl
    import diffphys.dp_utils as rdu
    import diffphys.geom_utils as rgu

    for m in (rdl, rdu, rgu):
        assert os.path.realpath(m.__file__).startswith(os.path.realpath(REF)), m.__file__
    return rdu, rgu, rdl


def reduce_loss_cases():
    """name -> (table, clip, th).  float32 like loss_traj at the call site (dp_model.py:777-779); a few float64."""
    g = torch.Generator().manual_seed(20261003)
    f32 = torch.float32

    def rnd(bs, F, lo=0.5, hi=1.5, dtype=f32):
        return (torch.rand(bs, F, generator=g, dtype=torch.float64) * (hi - lo) + lo).to(dtype)

    c = {}
    c["judge_env0_empty"] = (torch.tensor([[0, 0, 0], [1, 1, 1], [1, 50, 1]], dtype=f32), True, 0)
    c["plain_no_clip_needed"] = (rnd(7, 5), True, 0)
    t = rnd(9, 6); t[2, 3] = 40.0; t[5, 0] = 99.0; t[8, 5] = 12.5
    c["three_envs_clipped"] = (t, True, 0)
    t = rnd(6, 8); t[0, 5] = 30.0
    c["env0_clips_itself"] = (t, True, 0)
    t = rnd(5, 4); t[0] = 0
    c["env0_all_zero_others_spike"] = (t.clone().index_put_((torch.tensor([3]), torch.tensor([1])), torch.tensor(1e4)), True, 0)
    t = rnd(5, 4); t[0] = 0; t[1] = 0
    c["first_two_envs_empty"] = (t.clone().index_put_((torch.tensor([4]), torch.tensor([2])), torch.tensor(777.0)), True, 0)
    c["all_zero"] = (torch.zeros(4, 3), True, 0)
    c["all_zero_noclip"] = (torch.zeros(4, 3), False, 0)
    t = rnd(6, 4); t[0, 1] = float("nan"); t[3, 2] = 55.0
    c["env0_has_a_nan"] = (t, True, 0)
    t = rnd(6, 4); t[4, 1] = float("nan"); t[3, 2] = 55.0
    c["nan_in_later_env"] = (t, True, 0)
    t = rnd(6, 4); t[0] = float("nan")
    c["env0_all_nan"] = (t.clone().index_put_((torch.tensor([2]), torch.tensor([2])), torch.tensor(500.0)), True, 0)
    t = rnd(5, 4); t[0, 2] = float("inf")
    c["env0_has_inf"] = (t, True, 0)
    t = rnd(5, 4); t[3, 1] = float("inf")
    c["inf_in_later_env"] = (t, True, 0)
    t = rnd(5, 6); t[1, 2] = -3.0; t[2, 0] = -1.0; t[4, 4] = 60.0
    c["negative_entries"] = (t, True, 0)
    t = -rnd(4, 3)
    c["all_negative"] = (t, True, 0)
    c["single_frame"] = (torch.tensor([[2.0], [19.9], [20.1], [0.0], [3.0]], dtype=f32), True, 0)
    c["even_count_lower_median"] = (torch.tensor([[1, 2, 3, 4], [25, 1, 1, 1], [1, 1, 15, 35]], dtype=f32), True, 0)
    c["odd_count_with_zeros"] = (torch.tensor([[0, 1, 0, 3, 2, 0], [1, 1, 1, 21, 1, 1], [1, 19, 1, 1, 1, 1]], dtype=f32), True, 0)
    c["ties_in_env0"] = (torch.tensor([[2, 2, 2, 2], [20, 20, 20.5, 1], [1, 1, 1, 1]], dtype=f32), True, 0)
    c["exceed_at_first_entry"] = (torch.tensor([[1, 1, 1], [11, 1, 1], [1, 1, 11]], dtype=f32), True, 0)
    t = rnd(8, 5); t[1, 1] = 9.0; t[6, 3] = 4.0
    c["explicit_threshold_3p5"] = (t, True, 3.5)
    c["explicit_threshold_tensor"] = (t.clone(), True, torch.tensor(8.0))
    c["single_env"] = (torch.tensor([[1, 1, 1, 30, 1]], dtype=f32), True, 0)
    t = rnd(10, 4)
    t[::2, 2:] = 0  # out-of-sequence tails assigned zero before the call (dp_model.py:778)
    t[7, 1] = 100.0
    c["outseq_tails_zero"] = (t, True, 0)
    t = rnd(10, 4); t[0, 1:] = 0
    c["env0_single_positive"] = (t.clone().index_put_((torch.tensor([5]), torch.tensor([3])), 10.0 * t[0, 0] * 1.001), True, 0)
    t = rnd(12, 4); t[3, 0] = 80; t[4, 1] = 80; t[5, 2] = 80; t[6, 3] = 80; t[7] = 80; t[8, 0] = 1e30
    c["many_clipped"] = (t, True, 0)
    c["noclip_mixed_signs"] = (torch.tensor([[1, -2, 3], [0, 0, 5]], dtype=f32), False, 0)
    c["noclip_sum_negative"] = (torch.tensor([[1, -2, -3], [0, 0, 1]], dtype=f32), False, 0)
    t = rnd(4096, 4, 1e-3, 3e-3)
    idx = torch.randint(0, 4096, (97,), generator=g)
    t[idx, torch.randint(0, 4, (97,), generator=g)] = 0.5
    t[::5, 3] = 0
    c["headline_4096x4"] = (t, True, 0)
    t = rnd(4096, 4, 1e-3, 3e-3); t[0] = 0; t[9, 2] = 1.0
    c["headline_4096x4_env0_empty"] = (t, True, 0)
    t = rnd(10, 24, 1e-4, 4e-4); t[3, 17] = 0.2; t[9, 5] = 0.01
    c["window_10x24"] = (t, True, 0)
    c["float64_three_envs_clipped"] = (c["three_envs_clipped"][0].double(), True, 0)
    c["float64_env0_empty"] = (c["judge_env0_empty"][0].double(), True, 0)
    return c


def run_reduce_loss(rdu, out):
    names = []
    for name, (table, clip, th) in reduce_loss_cases().items():
        work = table.clone()
        with contextlib.redirect_stdout(io.StringIO()) as said:  # the reference prints "clipped env %d at %d"
            val = rdu.reduce_loss(work, clip=clip, th=th.clone() if torch.is_tensor(th) else th)
        out["rl/%s/table" % name] = table.numpy()
        out["rl/%s/clip" % name] = np.bool_(clip)
        out["rl/%s/th" % name] = np.float64(float(th))
        out["rl/%s/value" % name] = val.double().numpy()
        out["rl/%s/table_after" % name] = work.numpy()
        out["rl/%s/clipped_envs" % name] = np.asarray([int(l.split()[2]) for l in said.getvalue().splitlines()], dtype=np.int64)
        names.append(name)
        print("reduce_loss %-30s value %-12.6g clipped %s" % (name, float(val), out["rl/%s/clipped_envs" % name].tolist()[:8]))
    out["rl/names"] = np.asarray(names)


def run_small(rdu, rgu, out):
    g = torch.Generator().manual_seed(7)
    # remove_nan, both settings of clip (dp_utils.py:43-57), in place
    x = torch.randn(6, 11, generator=g) * 0.02
    x[1, 3] = float("nan"); x[4, 0] = float("nan"); x[2, 2] = float("inf"); x[3, 9] = -float("inf"); x[5, 5] = 0.5; x[0, 0] = -0.5
    for clip in (False, True):
        y = x.clone()
        rdu.remove_nan(y, 6, clip=clip)
        out["remove_nan/in"] = x.numpy()
        out["remove_nan/out_clip%d" % clip] = y.numpy()
    # rot_angle (geom_utils.py:37-46) on rotations incl. identity and a half turn, float32 and float64
    from scipy.spatial.transform import Rotation as R
    rs = np.random.RandomState(11)
    rv = rs.randn(40, 3)
    rv[0] = 0; rv[1] = [np.pi, 0, 0]; rv[2] = [1e-5, 0, 0]; rv[3] = [0, 3.1, 0]
    mats = R.from_rotvec(rv).as_matrix()
    out["rot_angle/mat"] = mats
    out["rot_angle/out_f64"] = rgu.rot_angle(torch.tensor(mats)).numpy()
    out["rot_angle/out_f32"] = rgu.rot_angle(torch.tensor(mats, dtype=torch.float32)).numpy()
    # fid_reindex (geom_utils.py:48-67)
    off = torch.tensor([0, 39, 72, 118, 229])
    fid = torch.tensor([0, 1, 38, 39, 40, 71, 72, 100, 117, 118, 228, 5, 80])
    vid, tid = rgu.fid_reindex(fid, 4, off)
    out["fid_reindex/fid"], out["fid_reindex/offset"] = fid.numpy(), off.numpy()
    out["fid_reindex/vid"], out["fid_reindex/tid"] = vid.numpy(), tid.numpy()
    fid2 = torch.arange(0, 760).reshape(10, 76) % 39
    vid2, tid2 = rgu.fid_reindex(fid2, 1, torch.tensor([0, 39]))
    out["fid_reindex/fid2"], out["fid_reindex/vid2"], out["fid_reindex/tid2"] = fid2.numpy(), vid2.numpy(), tid2.numpy()
    # quaternion_to_axis_angle / quaternion_invert (geom_utils.py:102-148), real part first
    q = torch.tensor(R.from_rotvec(rv).as_quat()[:, [3, 0, 1, 2]], dtype=torch.float32)
    q[5] = -q[5]
    out["quat/wxyz"] = q.numpy()
    out["quat/axis_angle"] = rgu.quaternion_to_axis_angle(q).numpy()
    out["quat/invert"] = rgu.quaternion_invert(q).numpy()
    # se3_vec2mat numpy branch (geom_utils.py:150-179) + se3_mat2rt
    vec = np.concatenate([rs.randn(12, 3), R.from_rotvec(rs.randn(12, 3)).as_quat()], -1).reshape(3, 4, 7)
    mat = rgu.se3_vec2mat(vec)
    out["se3/vec"], out["se3/mat"] = vec, mat
    rm, tm = rgu.se3_mat2rt(mat)
    out["se3/rmat"], out["se3/tmat"] = np.asarray(rm), np.asarray(tm)
    # compute_com (dp_utils.py:86-90)
    nb = 13
    body_q = np.concatenate([rs.randn(nb, 3), R.from_rotvec(rs.randn(nb, 3)).as_quat()], -1)
    part_com = rs.randn(nb, 3, 1) * 0.1
    part_mass = rs.rand(nb) + 0.2
    out["com/body_q"], out["com/part_com"], out["com/part_mass"] = body_q, part_com, part_mass
    out["com/out"] = rdu.compute_com(body_q, part_com, part_mass)
    # parse_rtk / project_bodies (dp_utils.py:185-216)
    rtk = torch.randn(2, 3, 4, 4, generator=g)
    rtk[..., 2, 3] += 5.0
    bodies = torch.randn(2, 3, 5, 7, generator=g)
    rt, km = rdu.parse_rtk(rtk)
    out["rtk/rtk"], out["rtk/bodies"] = rtk.numpy(), bodies.numpy()
    out["rtk/rtmat"], out["rtk/kmat"] = rt.numpy(), km.numpy()
    out["rtk/proj"] = rdu.project_bodies(bodies, rtk).numpy()


def run_mocap(rdu, rdl, out):
    """DataLoader + parse_amp + bullet2gl (dataloader.py:9-31, dp_utils.py:141-156) as phys_model.get_mocap_data composes them
    (dp_model.py:605-609) on whole sequences and on a (bs, T) window of interpolated frames."""
    here = os.getcwd()
    os.chdir(REF)  # the loader opens ./data/motion_sequences/...
    try:
        for seq in SEQS:
            dl = rdl.DataLoader({"seqname": seq})
            out["mocap/%s/frame_interval" % seq] = np.float64(dl.frame_interval)
            out["mocap/%s/offset" % seq] = np.asarray(dl.data_info["offset"])
            out["mocap/%s/n_frames" % seq] = np.int64(len(dl.amp_info))
            for in_bullet in (False, True):
                msm = rdl.parse_amp(dl.amp_info.copy())
                msm = {k: v.copy() for k, v in msm.items()}
                rdu.bullet2gl(msm, in_bullet)
                for k, v in msm.items():
                    out["mocap/%s/bullet%d/%s" % (seq, in_bullet, k)] = np.asarray(v)
            # a (bs, T) window like get_batch_input's: frames gathered to (2, 5, 85) first
            win = dl.amp_info[np.asarray([[0, 1, 2, 3, 4], [7, 9, 11, 13, 15]])]
            msm = {k: v.copy() for k, v in rdl.parse_amp(win.copy()).items()}
            rdu.bullet2gl(msm, False)
            for k in ("pos", "orn", "vel", "avel"):
                out["mocap/%s/window/%s" % (seq, k)] = np.asarray(msm[k])
    finally:
        os.chdir(here)


def run_timemlp(out):
    """the reference's TimeMLPWrapper as phys_model builds it (dp_model.py:292-315), at width 32 to keep the fixture small: the weights
    torch's default initialisation gives it under torch.manual_seed(0) (the constructor re-seeds with 8 at its end, so the SECOND module's
    weights follow from the first's), outputs at fractional frame ids, gradients of sum(y^2)"""
    import numpy as onp
    import diffphys.torch_utils as rtu

    assert os.path.realpath(rtu.__file__).startswith(os.path.realpath(REF))
    cfgs = {"root": dict(out_channels=6, D=8, skips=[4], time_scale=0.1, output_scale=0.5, W=32),
            "joint": dict(out_channels=12, W=32),
            "two_videos": dict(out_channels=5, W=32, output_scale=5.0,
                               frame_info={"frame_offset": onp.asarray([0, 39, 72]), "frame_mapping": list(range(72)),
                                           "frame_offset_raw": onp.asarray([0, 39, 72])})}
    torch.manual_seed(0)
    fid = torch.tensor([0.0, 0.5, 3.27, 17.0, 25.75, 38.0])
    for name, kw in cfgs.items():
        n = 72 if name == "two_videos" else 39
        m = rtu.TimeMLPWrapper(n, **kw)
        f = torch.cat([fid, torch.tensor([39.0, 55.5, 71.0])]) if name == "two_videos" else fid
        for k, v in m.state_dict().items():
            out["mlp/%s/state/%s" % (name, k)] = v.detach().numpy().copy()
        y = m(f)
        (y * y).sum().backward()
        out["mlp/%s/frame_id" % name] = f.numpy()
        out["mlp/%s/out" % name] = y.detach().numpy()
        for k, p_ in m.named_parameters():
            out["mlp/%s/grad/%s" % (name, k)] = p_.grad.detach().numpy().copy()
        print("TimeMLPWrapper %-10s %d state entries, out %s, |out| max %.4f" % (name, len(m.state_dict()), tuple(y.shape), float(y.abs().max())))
    out["mlp/names"] = onp.asarray(list(cfgs))
    # the two schedule helpers phys_model takes from lab4d_utils (dp_model.py:38-41,341,492-497)
    import diffphys.lab4d_utils as rl4

    xs = onp.asarray([-0.3, 0.0, 0.1, 0.25, 0.5, 0.77, 1.0, 1.4])
    out["interp/x2"] = xs
    out["interp/linear"] = onp.asarray([rl4.interp_wt((0, 0.5), (1, 0), float(t)) for t in xs])
    out["interp/linear_up"] = onp.asarray([rl4.interp_wt((0.2, 1.0), (0.01, 0.3), float(t), type="linear") for t in xs])
    out["interp/log"] = onp.asarray([rl4.interp_wt((0, 1), (1e-4, 1e-1), float(t), type="log") for t in xs])
    out["interp/exp"] = onp.asarray([rl4.interp_wt((1, 100), (0.0, 2.0), float(t), type="exp") for t in (0.5, 1.0, 3.0, 10.0, 100.0, 250.0)])
    lr = {"root_pose_mlp": 1e-4, "vel_mlp": 2e-4, "global_q": 1e-3}
    q = [("root_pose_mlp.head.0.weight", "startwith"), ("global_q", "startwith"), ("body_mass", "startwith"), ("x.vel_mlp.y", "with"), ("x.vel_mlp.y", "startwith")]
    out["match/result"] = onp.asarray([[float(a), float(b)] for a, b in (rl4.match_param_name(n, lr, t) for n, t in q)])
    try:
        rl4.match_param_name("root_pose_mlp.vel_mlp", lr, "with")
        out["match/multiple_raises"] = onp.bool_(False)
    except ValueError:
        out["match/multiple_raises"] = onp.bool_(True)


def main():
    rdu, rgu, rdl = import_reference()
    os.makedirs(OUT, exist_ok=True)
    note = ("outputs of the reference's own code (diffphys/dp_utils.py, geom_utils.py, dataloader.py imported from /root/reference "
            "with empty placeholder modules for dqtorch / cv2 / trimesh); generator scripts/make_ref_fixtures.py; torch %s numpy %s"
            % (torch.__version__, np.__version__))
    a = {"note": np.asarray(note)}
    run_reduce_loss(rdu, a)
    np.savez_compressed(os.path.join(OUT, "ref_host_reduce_loss.npz"), **a)
    b = {"note": np.asarray(note)}
    run_small(rdu, rgu, b)
    np.savez_compressed(os.path.join(OUT, "ref_host_small.npz"), **b)
    c = {"note": np.asarray(note)}
    run_mocap(rdu, rdl, c)
    np.savez_compressed(os.path.join(OUT, "ref_host_mocap.npz"), **c)
    d = {"note": np.asarray(note.replace("dp_utils.py, geom_utils.py, dataloader.py", "torch_utils.py, lab4d_utils.py"))}
    run_timemlp(d)
    np.savez_compressed(os.path.join(OUT, "ref_host_timemlp.npz"), **d)
    for f in ("ref_host_reduce_loss.npz", "ref_host_small.npz", "ref_host_mocap.npz", "ref_host_timemlp.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")


if __name__ == "__main__":
    main()
