"""CPU tests of the loss / frame utilities and MLP plumbing.  The SE(3) / loss formulas are pinned
to scipy here on BOTH restatements: the product's host helpers (diffphys_amd.geom_utils) and the checker of the HIP pose / loss
kernels (oracle/pose_torch.py); the kernels themselves are compared with that checker on the GPU (tests/test_gpu_pose.py)."""
import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation as R

from diffphys_amd import dp_utils, geom_utils
from oracle import pose_torch
from diffphys_amd.time_mlp import TimeMLPWrapper, interp_wt, match_param_name


@pytest.mark.parametrize("geom_utils", [geom_utils, pose_torch], ids=["product", "oracle"])
def test_quaternion_matrix_roundtrip_against_scipy(geom_utils):
    rng = np.random.RandomState(0)
    q_xyzw = rng.randn(50, 4)
    q_xyzw /= np.linalg.norm(q_xyzw, axis=1, keepdims=True)
    m_ref = R.from_quat(q_xyzw).as_matrix()
    q_wxyz = torch.tensor(q_xyzw[:, [3, 0, 1, 2]])
    m = geom_utils.quaternion_to_matrix(q_wxyz)
    assert np.allclose(m.numpy(), m_ref, atol=1e-12)
    back = geom_utils.matrix_to_quaternion(m)
    sign = torch.sign((back * q_wxyz).sum(-1, keepdim=True))
    assert torch.allclose(back * sign, q_wxyz, atol=1e-10)
    aa = torch.tensor(R.from_quat(q_xyzw).as_rotvec())
    assert np.allclose(geom_utils.axis_angle_to_matrix(aa).numpy(), m_ref, atol=1e-10)
    assert torch.allclose(geom_utils.quaternion_to_axis_angle(geom_utils.axis_angle_to_quaternion(aa)), aa, atol=1e-10)


@pytest.mark.parametrize("geom_utils", [geom_utils, pose_torch], ids=["product", "oracle"])
def test_se3_vec_mat_roundtrip_and_rotate_frame(geom_utils):
    rng = np.random.RandomState(1)
    v = torch.tensor(rng.randn(4, 3, 7))
    v[..., 3:] = v[..., 3:] / v[..., 3:].norm(dim=-1, keepdim=True)
    m = geom_utils.se3_vec2mat(v)
    assert torch.allclose(m[..., 3, :], torch.tensor([0.0, 0, 0, 1]).expand(4, 3, 4).to(m.dtype))
    v2 = geom_utils.se3_mat2vec(m)
    sign = torch.sign((v2[..., 3:] * v[..., 3:]).sum(-1, keepdim=True))
    assert torch.allclose(v2[..., :3], v[..., :3]) and torch.allclose(v2[..., 3:] * sign, v[..., 3:], atol=1e-10)
    g = torch.tensor([0.0, -0.3, 0.0, 0.0, 0.0, 0.0, 1.0], dtype=v.dtype)
    out = pose_torch.rotate_frame(g, v)
    assert torch.allclose(out[..., 1], v[..., 1] - 0.3) and torch.allclose(out[..., 0], v[..., 0])
    qd = torch.tensor(rng.randn(4, 3, 6))
    assert torch.allclose(pose_torch.rotate_frame_vel(g, qd), qd, atol=1e-12)  # identity rotation leaves twists alone
    # the product's compositions are HIP kernels and nothing else: CPU tensors are refused, not computed somewhere else
    for fn, args in ((dp_utils.rotate_frame, (g, v)), (dp_utils.rotate_frame_vel, (g, qd)), (dp_utils.compose_delta, (v, qd)), (dp_utils.se3_loss, (v, v))):
        with pytest.raises(TypeError, match="no CPU fallback"):
            fn(*args)


def test_se3_loss_and_reduce_loss():
    a = torch.tensor([[0.0, 0, 0, 0, 0, 0, 1.0]], dtype=torch.float64)
    ang = 0.4
    b = torch.tensor([[1.0, 2.0, 0, 0, np.sin(ang / 2), 0, np.cos(ang / 2)]], dtype=torch.float64)
    assert abs(float(pose_torch.se3_loss(a, b)) - (5.0 + 0.1 * ang)) < 1e-5
    nan = torch.tensor([[float("nan"), 0, 0, 0, 0, 0, 1.0]], dtype=torch.float64)
    assert float(pose_torch.se3_loss(nan, b)) == 0.0
    seq = torch.tensor([[1.0, 1.1, 0.9, 50.0, 1.0], [1.0, 1.0, 1.0, 1.0, 1.0]])
    red = dp_utils.reduce_loss(seq.clone(), clip=True)
    assert abs(float(red) - np.mean([1.0, 1.1, 0.9] + [1.0] * 5)) < 1e-6  # env 0 truncated where it first exceeds 10x median
    t = torch.tensor([1.0, float("nan"), 3.0])
    dp_utils.remove_nan(t)
    assert t.tolist() == [1.0, 0.0, 3.0]


def test_time_mlp_and_schedules():
    mlp = TimeMLPWrapper(39, out_channels=12, output_scale=0.5)
    y = mlp(torch.tensor([0.0, 3.5, 38.0]))
    assert y.shape == (3, 12) and torch.isfinite(y).all() and y.abs().max() < 1.0
    y.sum().backward()
    assert all(p.grad is not None for p in mlp.parameters())
    # the reference's constructor re-seeds torch at its END (torch_utils.py:160-161): every module built after another one starts from seed 8
    mlp2 = TimeMLPWrapper(39, out_channels=12, output_scale=0.5)
    mlp3 = TimeMLPWrapper(39, out_channels=12, output_scale=0.5)
    assert torch.equal(mlp2(torch.tensor([1.0])), mlp3(torch.tensor([1.0])))
    assert interp_wt((0, 0.5), (1, 0), 0.25) == 0.5 and interp_wt((0, 0.5), (1, 0), 0.9) == 0
    assert match_param_name("root_pose_mlp.base_quat", {"root_pose_mlp.base_quat": 1e-3}, "with") == (1, 1e-3)
    assert match_param_name("vel_mlp.head.weight", {"vel_mlp": 1e-4, "torque_mlp": 2.0}, "startwith") == (1, 1e-4)


def test_reduce_loss_matches_the_reference_loop():
    """reduce_loss is the reference's per-env loop (dp_utils.py:93-110) without its host synchronisations: same value, same
    in-place truncation, same gradient -- including the reference's quirk that the threshold comes from the first env only."""
    torch.manual_seed(0)
    for trial in range(120):
        bs, T = int(torch.randint(1, 9, (1,))), int(torch.randint(1, 12, (1,)))
        x = torch.rand(bs, T) ** 3
        if trial % 3 == 0:
            x[0] = 0
        if trial % 5 == 0:
            x[torch.rand(bs, T) < 0.4] = 0
        if trial % 7 == 0:
            x[int(torch.randint(0, bs, (1,))), int(torch.randint(0, T, (1,)))] = 50.0
        if trial % 11 == 0:
            x[:] = 0
        for clip in (False, True):
            a, b = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
            a2, b2 = a * 1.0, b * 1.0
            ra, rb = pose_torch.reduce_loss_loop(a2, clip=clip), dp_utils.reduce_loss(b2, clip=clip)
            assert torch.allclose(ra, rb, atol=1e-7) and torch.equal(a2.detach(), b2.detach())
            ra.backward(); rb.backward()
            assert torch.allclose(a.grad, b.grad, atol=1e-7)


def test_device_side_mocap_query_matches_the_numpy_pipeline():
    """dataloader.mocap_tensors (torch, no host round trip) against interp1d + parse_amp + bullet2gl (row f2), inside and outside
    the table's range (both extrapolate linearly); float32 outputs, agreement to 1e-6 relative."""
    import scipy.interpolate
    from diffphys_amd import dataloader

    dl = dataloader.DataLoader({"seqname": "mi-trot"})
    n = len(dl.amp_info)
    f = scipy.interpolate.interp1d(np.arange(n), dl.amp_info, kind="linear", fill_value="extrapolate", axis=0)
    steps = np.random.RandomState(0).uniform(-3.0, n + 3.0, size=(5, 17))
    steps[0, :4] = [0.0, n - 1.0, 1.0, n - 2.0]
    ref = {k: np.array(v, copy=True) for k, v in dataloader.parse_amp(f(steps)).items()}
    dataloader.bullet2gl(ref, False)
    got = dataloader.mocap_tensors(torch.as_tensor(np.asarray(dl.amp_info, dtype=np.float64)), torch.as_tensor(steps))
    assert set(got) == set(ref)
    for k in ref:
        assert got[k].dtype == torch.float32 and tuple(got[k].shape) == ref[k].shape
        assert np.abs(got[k].numpy() - ref[k].astype(np.float32)).max() <= 1e-6 * max(1.0, np.abs(ref[k]).max()), k


@pytest.mark.parametrize("steady", [False, True], ids=["parameters-come-and-go", "every-parameter-every-iteration"])
def test_batched_gradient_history_matches_the_reference_loop(steady):
    """grad_guard.GradHistory (one device tensor, one host transfer per iteration) against the reference's per-parameter
    lists (dp_model.py:965-1000 restated as a loop): same histories, same outlier decisions, same clipped gradients --
    with outliers injected; with a parameter that has no gradient in some iterations and one that appears late (the general, per-row
    path), and with every parameter present in every iteration (a real run: outliers then take the one-select path over the table)."""
    from diffphys_amd.grad_guard import GradHistory

    torch.manual_seed(0)
    shapes = [(3,), (4, 5), (7,), (2, 2, 2), (11,), (6,)]
    names = ["p%d" % i for i in range(len(shapes))]
    hist, queues = GradHistory(queue_length=10, scale_threshold=5.0), {}
    n_out = 0
    for it in range(60):
        active = [i for i in range(len(shapes)) if steady or (not (i == 2 and it % 7 == 3) and not (i == 5 and it < 9))]
        grads = [torch.randn(shapes[i]) * (0.5 + 0.1 * i) for i in active]
        if it in (25, 31, 32, 50):
            grads[it % len(grads)] *= 40.0  # an outlier
        if steady and it in (31, 44):
            grads[(it + 2) % len(grads)] *= 60.0  # two parameters flagged in one iteration
        ref = [g.clone() for g in grads]
        # reference loop
        ref_flags = []
        for k, i in enumerate(active):
            g = ref[k]
            norm = g.reshape(-1).norm(2, -1)
            q = queues.setdefault(names[i], [])
            flag = False
            if len(q) > 10:
                med = torch.stack(q[:-1]).median()
                if norm > 5.0 * med:
                    g.mul_((med / (norm + 1e-6)).clamp(max=1.0))
                    flag = True
                else:
                    q.append(norm); q.pop(0)
            else:
                q.append(norm)
            ref_flags.append(flag)
        # batched
        plan = hist.plan([names[i] for i in active], grads)
        any_out = plan["any_outlier"] is not None and bool(plan["any_outlier"] != 0)
        flags = hist.commit(plan, any_out)
        assert flags == ref_flags, it
        n_out += sum(flags)
        for a, b in zip(grads, ref):
            assert torch.allclose(a, b, rtol=1e-6, atol=0)
        for i in range(len(shapes)):
            if names[i] in queues:
                assert np.allclose(hist.history(names[i]), [float(x) for x in queues[names[i]]], rtol=1e-6), (it, i)
    assert n_out >= 3
    # plan() alone does not touch the histories (the global-norm guard returns before they are updated)
    before = hist.Q.clone(), list(hist.fill)
    hist.plan(names, [torch.randn(s) for s in shapes])
    assert torch.equal(hist.Q, before[0]) and hist.fill == before[1]


def test_time_mlp_linear_functions_match_torch_autograd():
    """time_mlp's custom autograd Functions (bias gradient by a sum the graph replay survives, ReLU in the GEMM, the shared time embedding)
    against plain torch on the CPU: outputs and all three gradients of relu(x W^T + b) and of x W^T + b, and a TimeMLPWrapper evaluated
    with and without the shared-embedding dict."""
    from diffphys_amd import time_mlp

    torch.manual_seed(3)
    x = torch.randn(37, 24, dtype=torch.float64, requires_grad=True)
    w = torch.randn(16, 24, dtype=torch.float64, requires_grad=True)
    b = torch.randn(16, dtype=torch.float64, requires_grad=True)
    up = torch.randn(37, 16, dtype=torch.float64)
    for fn, ref in ((time_mlp._LinearReluGemmBias.apply, lambda: torch.relu(torch.nn.functional.linear(x, w, b))),
                    (time_mlp._LinearGemmBias.apply, lambda: torch.nn.functional.linear(x, w, b))):
        got = fn(x, w, b)
        g = torch.autograd.grad(got, (x, w, b), up)
        want = ref()
        gr = torch.autograd.grad(want, (x, w, b), up)
        assert torch.allclose(got, want, rtol=1e-12, atol=1e-12)
        assert all(torch.allclose(a, c, rtol=1e-11, atol=1e-12) for a, c in zip(g, gr))
    torch.manual_seed(5)
    mlp = time_mlp.TimeMLPWrapper(90, out_channels=7)
    t = torch.linspace(0, 89, 41)
    a = mlp(t)
    shared = {}
    b1, b2 = mlp(t, shared), mlp(t, shared)   # the second call reads the cached mapping and Fourier features
    assert torch.equal(a, b1) and torch.equal(a, b2) and len(shared) == 2
