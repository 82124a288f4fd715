"""Product, oracle and HIP kernels held to OUTPUTS OF THE REFERENCE'S OWN CODE (tests/golden/ref_host_*.npz, made in the build container
by scripts/make_ref_fixtures.py: diffphys/dp_utils.py, geom_utils.py, dataloader.py imported from /root/reference and run as they
are).  These are the only results in this repo that the reference itself produced (its rollout needs warp_lang, absent): they pin the
host functions either side of the hot path -- SURVEY.md section 8 rows a9 (remove_nan), f2 (mocap pipeline, SE(3) helpers), f3 / f4
(reduce_loss, the one reduction whose result steers the rollout's adjoint)."""
import os

import numpy as np
import pytest
import torch

from diffphys_amd import dataloader, dp_utils, geom_utils
from oracle import pose_torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def rl():
    with np.load(os.path.join(GOLD, "ref_host_reduce_loss.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def small():
    with np.load(os.path.join(GOLD, "ref_host_small.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def mocap():
    with np.load(os.path.join(GOLD, "ref_host_mocap.npz")) as z:
        return {k: z[k] for k in z.files}


def _same(a, b, rtol=0.0, atol=0.0):
    """equal including the positions of NaN / inf"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and np.allclose(a, b, rtol=rtol, atol=atol, equal_nan=True)


def test_fixtures_say_where_they_come_from(rl, small, mocap):
    for z in (rl, small, mocap):
        assert "outputs of the reference's own code" in str(z["note"]) and "make_ref_fixtures.py" in str(z["note"])
    assert len(rl["rl/names"]) >= 30
    assert float(rl["rl/judge_env0_empty/value"]) == pytest.approx(9.1666667, rel=1e-6)  # what round 4's reduce_loss got wrong (1.0)


@pytest.mark.parametrize("impl", ["product", "oracle"])
def test_reduce_loss_against_the_reference(rl, impl):
    """value, in-place truncation and which envs were clipped, for every table of the fixture: env 0 without a positive entry (NaN
    threshold: nothing clipped), NaN / inf entries, ties, explicit thresholds, clip on and off, float32 and float64."""
    fn = dp_utils.reduce_loss if impl == "product" else pose_torch.reduce_loss_loop
    for name in rl["rl/names"]:
        p = "rl/%s/" % name
        table = torch.from_numpy(rl[p + "table"].copy())
        th = float(rl[p + "th"])
        work = table.clone()
        val = fn(work, clip=bool(rl[p + "clip"]), th=th if th != 0 else 0)
        tol = 1e-6 if table.dtype == torch.float32 else 1e-12
        assert _same(val.detach().numpy(), rl[p + "value"], rtol=tol), (name, float(val), float(rl[p + "value"]))
        assert _same(work.numpy(), rl[p + "table_after"]), name
        clipped = np.nonzero((np.nan_to_num(work.numpy(), nan=1.0) != np.nan_to_num(table.numpy(), nan=1.0)).any(1))[0]
        assert set(clipped.tolist()) <= set(rl[p + "clipped_envs"].tolist()), name  # (a clip at a zero entry changes nothing visible)
    # a tensor threshold, as a caller that carries it over would pass it
    p = "rl/explicit_threshold_tensor/"
    work = torch.from_numpy(rl[p + "table"].copy())
    val = fn(work, clip=True, th=torch.tensor(float(rl[p + "th"])))
    assert _same(val.numpy(), rl[p + "value"], rtol=1e-6) and _same(work.numpy(), rl[p + "table_after"])


def test_reduce_loss_gradient_is_the_reference_loops(rl):
    """the product's synchronisation-free form has the gradient of the reference's control flow (autograd through the oracle loop,
    which the test above holds to the reference's values)"""
    for name in rl["rl/names"]:
        p = "rl/%s/" % name
        if not np.isfinite(rl[p + "table"]).all() or float(rl[p + "th"]) != 0:
            continue
        x = torch.from_numpy(rl[p + "table"].copy()).double()
        a, b = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ra = pose_torch.reduce_loss_loop(a * 1.0, clip=bool(rl[p + "clip"]))
        rb = dp_utils.reduce_loss(b * 1.0, clip=bool(rl[p + "clip"]))
        ra.backward(); rb.backward()
        assert torch.allclose(a.grad, b.grad, atol=1e-15), name


def test_remove_nan_against_the_reference(small):
    for clip in (0, 1):
        x = torch.from_numpy(small["remove_nan/in"].copy())
        dp_utils.remove_nan(x, 6, clip=bool(clip))
        assert _same(x.numpy(), small["remove_nan/out_clip%d" % clip])


@pytest.mark.parametrize("mod", [geom_utils, pose_torch], ids=["product", "oracle"])
def test_geometry_helpers_against_the_reference(small, mod):
    m = torch.from_numpy(small["rot_angle/mat"])
    assert _same(mod.rot_angle(m).numpy(), small["rot_angle/out_f64"], atol=1e-14)
    assert _same(mod.rot_angle(m.float()).numpy(), small["rot_angle/out_f32"], atol=2e-6)
    q = torch.from_numpy(small["quat/wxyz"])
    assert _same(mod.quaternion_to_axis_angle(q).numpy(), small["quat/axis_angle"], atol=2e-6)
    assert _same(mod.quaternion_invert(q).numpy(), small["quat/invert"])
    # se3_vec2mat: the reference's numpy branch (scipy) against the torch forms here, float64
    vec = torch.from_numpy(small["se3/vec"])
    mat = mod.se3_vec2mat(vec)
    assert _same(mat.numpy(), small["se3/mat"], atol=1e-12)
    assert _same(mat[..., :3, :3].numpy(), small["se3/rmat"], atol=1e-12) and _same(mat[..., :3, 3].numpy(), small["se3/tmat"], atol=1e-12)


def test_fid_reindex_and_compute_com_against_the_reference(small):
    vid, tid = geom_utils.fid_reindex(torch.from_numpy(small["fid_reindex/fid"]), 4, torch.from_numpy(small["fid_reindex/offset"]))
    assert np.array_equal(vid.numpy(), small["fid_reindex/vid"]) and _same(tid.numpy(), small["fid_reindex/tid"], atol=1e-7)
    vid, tid = geom_utils.fid_reindex(torch.from_numpy(small["fid_reindex/fid2"]), 1, torch.tensor([0, 39]))
    assert np.array_equal(vid.numpy(), small["fid_reindex/vid2"]) and _same(tid.numpy(), small["fid_reindex/tid2"], atol=1e-7)
    com = dp_utils.compute_com(small["com/body_q"], small["com/part_com"], small["com/part_mass"])
    assert _same(com, small["com/out"], atol=1e-14)


def test_mocap_pipeline_against_the_reference(mocap):
    """DataLoader -> parse_amp -> bullet2gl (both settings of in_bullet) on the five sequences the reference ships, whole tables and a
    (bs, T) window; and the device-side mocap_tensors (the product's get_mocap_data) against the same."""
    for seq in ("mi-pace", "mi-trot", "mi-spin", "mi-turn", "mi-sidesteps"):
        dl = dataloader.DataLoader({"seqname": seq}, data_root="/nonexistent")  # the compiled table (the GPU box has no data directory)
        assert len(dl.amp_info) == int(mocap["mocap/%s/n_frames" % seq]) and dl.frame_interval == float(mocap["mocap/%s/frame_interval" % seq])
        assert np.array_equal(dl.data_info["offset"], mocap["mocap/%s/offset" % seq])
        for in_bullet in (0, 1):
            msm = {k: np.array(v, copy=True) for k, v in dataloader.parse_amp(dl.amp_info).items()}
            dataloader.bullet2gl(msm, bool(in_bullet))
            for k, v in msm.items():
                ref = mocap["mocap/%s/bullet%d/%s" % (seq, in_bullet, k)]
                assert _same(v, ref, atol=1e-13), (seq, in_bullet, k, np.abs(v - ref).max())
        win = dl.amp_info[np.asarray([[0, 1, 2, 3, 4], [7, 9, 11, 13, 15]])]
        msm = {k: np.array(v, copy=True) for k, v in dataloader.parse_amp(win).items()}
        dataloader.bullet2gl(msm, False)
        for k in ("pos", "orn", "vel", "avel"):
            assert _same(msm[k], mocap["mocap/%s/window/%s" % (seq, k)], atol=1e-13)
        # the torch form the product runs per iteration, at integer frame positions = the table itself
        tens = dataloader.mocap_tensors(torch.from_numpy(dl.amp_info), torch.arange(len(dl.amp_info), dtype=torch.float64))
        for k in ("pos", "orn", "vel", "avel", "jang", "jvel", "kp", "kp_vel"):
            assert _same(tens[k].numpy(), mocap["mocap/%s/bullet0/%s" % (seq, k)].astype(np.float32), atol=1e-6), (seq, k)


# ---------------------------------------------------------------------------------------------------------------- GPU: the HIP kernel


@pytest.fixture(scope="module")
def dev():
    from diffphys_amd import hip_backend

    hip_backend.lib()
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_hip_reduce_kernel_against_the_reference(rl, dev):
    """pd_reduce_loss -- the one-workgroup code that runs after every rollout with a trajectory loss (pd_trajloss.h) -- on every
    float32 table of the fixture: value, threshold decisions (= the table after its in-place truncation), clipped-env count; its
    `scale` output against autograd through the reference's control flow."""
    from diffphys_amd import hip_backend

    n = 0
    for name in rl["rl/names"]:
        p = "rl/%s/" % name
        if rl[p + "table"].dtype != np.float32 or float(rl[p + "th"]) != 0:  # (the kernel has no caller-given threshold)
            continue
        clip = bool(rl[p + "clip"])
        table = torch.from_numpy(rl[p + "table"].copy()).to(dev)
        red, scale = hip_backend.reduce_loss(table, clip=clip)
        red = red.cpu().numpy()
        assert _same(red[0], rl[p + "value"], rtol=2e-6), (name, red[0], float(rl[p + "value"]))
        assert _same(table.cpu().numpy(), rl[p + "table_after"]), name
        assert int(red[3]) == len(rl[p + "clipped_envs"]), (name, red[3], rl[p + "clipped_envs"])
        if np.isfinite(rl[p + "table"]).all():
            x = torch.from_numpy(rl[p + "table"].copy()).double().requires_grad_(True)
            (g,) = torch.autograd.grad(pose_torch.reduce_loss_loop(x * 1.0, clip=clip), x)
            assert float((scale.cpu().double() - g).abs().max()) <= 1e-6 * float(g.abs().max()) + 1e-12, name
        n += 1
    assert n >= 25
    # a table too large for LDS takes the kernel's other path: same answers as the torch form the CPU tests hold to the reference
    g = torch.Generator().manual_seed(5)
    big = torch.rand(40000, 4, generator=g) * 2e-3 + 1e-3
    big[torch.randint(0, 40000, (300,), generator=g), torch.randint(0, 4, (300,), generator=g)] = 0.7
    for env0_empty in (False, True):
        t = big.clone()
        if env0_empty:
            t[0] = 0
        ref_tab = t.clone()
        ref = dp_utils.reduce_loss(ref_tab, clip=True)
        d = t.to(dev)
        red, _ = hip_backend.reduce_loss(d, clip=True)
        assert abs(float(red[0]) - float(ref)) <= 2e-6 * float(ref) and torch.equal(d.cpu(), ref_tab)
        assert (int(red[3]) == 0) == env0_empty
    # an empty table: value 0 (the header says so), nothing written anywhere else
    red, sc = hip_backend.reduce_loss(torch.zeros(0, 4, device=dev), clip=True)
    assert float(red[0]) == 0.0 and int(red[2]) == 0 and sc.numel() == 0


# ------------------------------------------------------------------------------------------------ the time-MLPs of phys_model (row f3)
_MLP_CFG = {"root": dict(out_channels=6, D=8, skips=[4], time_scale=0.1, output_scale=0.5, W=32),
            "joint": dict(out_channels=12, W=32),
            "two_videos": dict(out_channels=5, W=32, output_scale=5.0,
                               frame_info={"frame_offset": np.asarray([0, 39, 72]), "frame_mapping": list(range(72)),
                                           "frame_offset_raw": np.asarray([0, 39, 72])})}


def test_time_mlp_is_the_references_module():
    """diffphys_amd.time_mlp.TimeMLPWrapper against the reference's OWN module (diffphys/torch_utils.py:116-180 over lab4d_utils.py, run by
    scripts/make_ref_fixtures.py): (i) built under the same seed in the same order, its default initialisation gives the SAME weights bit
    for bit -- incl. the constructor's re-seeding, which makes the second module's weights a function of the first having been built;
    (ii) the reference's state_dict loads with strict=True; (iii) outputs at fractional frame ids and every parameter gradient of
    sum(y^2) agree to fp32 round-off (one and two videos; the root-pose configuration D=8, skips=[4], time_scale, output_scale)."""
    from diffphys_amd.time_mlp import TimeMLPWrapper

    with np.load(os.path.join(GOLD, "ref_host_timemlp.npz")) as z:
        ref = {k: z[k] for k in z.files}
    assert "outputs of the reference's own code" in str(ref["note"])
    torch.manual_seed(0)
    for name in ref["mlp/names"]:
        n = 72 if name == "two_videos" else 39
        m = TimeMLPWrapper(n, **_MLP_CFG[name])
        sd = {k[len("mlp/%s/state/" % name):]: torch.from_numpy(v) for k, v in ref.items() if k.startswith("mlp/%s/state/" % name)}
        assert set(sd) == set(m.state_dict()), (sorted(set(sd) ^ set(m.state_dict())))
        for k, v in m.state_dict().items():   # (i) the same initial weights, bit for bit
            assert torch.equal(v, sd[k]), (name, k)
        m.load_state_dict(sd, strict=True)    # (ii)
        y = m(torch.from_numpy(ref["mlp/%s/frame_id" % name]))
        assert np.allclose(y.detach().numpy(), ref["mlp/%s/out" % name], rtol=2e-5, atol=2e-7), (name, np.abs(y.detach().numpy() - ref["mlp/%s/out" % name]).max())
        (y * y).sum().backward()
        for k, p in m.named_parameters():     # (iii)
            g = ref["mlp/%s/grad/%s" % (name, k)]
            assert np.allclose(p.grad.numpy(), g, rtol=1e-4, atol=1e-6 * max(1.0, np.abs(g).max())), (name, k, np.abs(p.grad.numpy() - g).max())


def test_schedule_helpers_against_the_reference():
    """interp_wt / match_param_name of the reference's lab4d_utils (what set_loss_weight and the per-parameter learning rates run on)"""
    from diffphys_amd.time_mlp import interp_wt, match_param_name

    with np.load(os.path.join(GOLD, "ref_host_timemlp.npz")) as z:
        ref = {k: z[k] for k in z.files if k.startswith(("interp/", "match/"))}
    xs = ref["interp/x2"]
    assert np.allclose([interp_wt((0, 0.5), (1, 0), float(t)) for t in xs], ref["interp/linear"], atol=1e-15)
    assert np.allclose([interp_wt((0.2, 1.0), (0.01, 0.3), float(t), type="linear") for t in xs], ref["interp/linear_up"], atol=1e-15)
    assert np.allclose([interp_wt((0, 1), (1e-4, 1e-1), float(t), type="log") for t in xs], ref["interp/log"], rtol=1e-12)
    assert np.allclose([interp_wt((1, 100), (0.0, 2.0), float(t), type="exp") for t in (0.5, 1.0, 3.0, 10.0, 100.0, 250.0)], ref["interp/exp"], atol=1e-14)
    lr = {"root_pose_mlp": 1e-4, "vel_mlp": 2e-4, "global_q": 1e-3}
    q = [("root_pose_mlp.head.0.weight", "startwith"), ("global_q", "startwith"), ("body_mass", "startwith"), ("x.vel_mlp.y", "with"), ("x.vel_mlp.y", "startwith")]
    got = np.asarray([[float(a), float(b)] for a, b in (match_param_name(n, lr, t) for n, t in q)])
    assert np.array_equal(got, ref["match/result"])
    assert bool(ref["match/multiple_raises"])
    with pytest.raises(ValueError):
        match_param_name("root_pose_mlp.vel_mlp", lr, "with")
