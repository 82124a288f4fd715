"""GPU tests of the pose-algebra kernels (C ABI pd_pose_op / pd_pose_op_vjp / pd_foot_height)
against the torch compositions they replace (oracle/pose_torch.py: test infrastructure, written after the reference's
diffphys/dp_utils.py:22-31,60-84 and diffphys/geom_utils.py:148-203) -- values in float32 and float64, gradients against
torch autograd of the float64 composition."""
import numpy as np
import pytest
import torch

from oracle import pose_torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _poses(rng, n, special=True):
    """(n, 7) poses with un-normalised quaternions; with `special`, the first rows are rotations by ~pi about x / y / z and
    about a diagonal (each of the four matrix->quaternion candidates is the best-conditioned one somewhere) and identity."""
    q = rng.randn(n, 4)
    q *= (0.5 + rng.rand(n, 1))  # |q| != 1, like linearly interpolated mocap rows
    if special:
        eps = 1e-3
        q[0] = [1, eps, -eps, eps]; q[1] = [eps, 1, eps, -eps]; q[2] = [-eps, eps, 1, eps]; q[3] = [0.7, 0.7, 0.1, 1e-4]
        q[4] = [0, 0, 0, 1]; q[5] = [0, 0, 0, -2.0]; q[6] = [0.5, 0.5, 0.5, 0.5]
    return np.concatenate([rng.randn(n, 3), q], 1)


def _deltas(rng, n):
    d = np.concatenate([rng.randn(n, 3) * 0.1, rng.randn(n, 3) * 0.5], 1)
    d[0, 3:] = 0.0                      # zero rotation: the series branch, zero subgradient of the norm
    d[1, 3:] = [3e-7, -2e-7, 1e-7]      # below the 1e-6 switch
    d[2, 3:] = [np.pi - 1e-3, 0, 0]     # almost a half turn
    d[3, 3:] = [2.0, -2.0, 1.5]         # more than pi
    return d


def _check(op_hip, op_torch, a64, b64, dev, grad_a=True, grad_b=True, tol=3e-6):
    a32 = torch.tensor(a64, dtype=torch.float32, device=dev, requires_grad=grad_a)
    b32 = torch.tensor(b64, dtype=torch.float32, device=dev, requires_grad=grad_b)
    out = op_hip(a32, b32)
    # float64 torch composition evaluated at the float32 inputs
    ad = a32.detach().double().cpu().requires_grad_(grad_a)
    bd = b32.detach().double().cpu().requires_grad_(grad_b)
    ref = op_torch(ad, bd)
    assert out.shape == ref.shape and out.dtype == torch.float32
    err = (out.detach().double().cpu() - ref.detach()).abs().max().item()
    assert err < tol * max(1.0, ref.detach().abs().max().item()), err
    # the float32 torch composition on the GPU is no closer to float64 than the kernel is (up to a factor)
    t32 = op_torch(a32.detach(), b32.detach())
    err32 = (t32.double().cpu() - ref.detach()).abs().max().item()
    assert err <= 4 * err32 + 1e-6, (err, err32)
    g = torch.tensor(np.random.RandomState(5).randn(*ref.shape), dtype=torch.float64)
    out.backward(g.float().to(dev))
    ref.backward(g)
    for x32, xd, need in ((a32, ad, grad_a), (b32, bd, grad_b)):
        if need:
            ge = (x32.grad.double().cpu() - xd.grad).abs().max().item()
            assert ge < 2e-5 * max(1.0, xd.grad.abs().max().item()), ge
    return err


def test_compose_delta_matches_the_torch_composition(dev):
    from diffphys_amd import dp_utils

    rng = np.random.RandomState(0)
    n = 4096
    tq, dl = _poses(rng, n), _deltas(rng, n)
    _check(dp_utils.compose_delta, pose_torch.compose_delta, tq, dl, dev)
    # (bs, T, .) operands as phys_model passes them, and no gradient needed for the target
    _check(dp_utils.compose_delta, pose_torch.compose_delta, tq.reshape(64, 64, 7), dl.reshape(64, 64, 6), dev, grad_a=False)


def test_rotate_frame_and_rotate_frame_vel_match_the_torch_compositions(dev):
    from diffphys_amd import dp_utils

    rng = np.random.RandomState(1)
    tq = _poses(rng, 64 * 50).reshape(64, 50, 7)
    qd = rng.randn(64, 50, 6)
    for gq in ([0.0, -0.3, 0.1, 0.0, 0.0, 0.0, 1.0], [0.2, 0.1, -0.4, 0.3, -0.5, 0.2, 0.9], [0.0, 0.0, 0.0, 1.0, 1e-3, 1e-3, 1e-3]):
        gq = np.asarray(gq)
        _check(dp_utils.rotate_frame, pose_torch.rotate_frame, gq, tq, dev)
        _check(dp_utils.rotate_frame, pose_torch.rotate_frame, gq, tq, dev, grad_b=False)  # as in phys_model: mocap rows need no gradient
        _check(dp_utils.rotate_frame_vel, pose_torch.rotate_frame_vel, gq, qd, dev)
        _check(dp_utils.rotate_frame_vel, pose_torch.rotate_frame_vel, gq, qd, dev, grad_b=False)


def test_pose_ops_reject_bad_operands(dev):
    from diffphys_amd import hip_backend

    a = torch.zeros(8, 7, device=dev)
    with pytest.raises(ValueError):
        hip_backend.pose_op(hip_backend.POSE_COMPOSE_DELTA, a, torch.zeros(8, 7, device=dev))  # delta must be 6 wide
    with pytest.raises(ValueError):
        hip_backend.pose_op(hip_backend.POSE_ROTATE_FRAME, torch.zeros(3, 7, device=dev), a)   # 3 globals against 8 targets
    with pytest.raises(TypeError):
        hip_backend.pose_op(hip_backend.POSE_ROTATE_FRAME, a.double(), a)
    with pytest.raises(ValueError):
        hip_backend.pose_op(hip_backend.POSE_ROTATE_FRAME, a.cpu(), a)
    out = hip_backend.pose_op(hip_backend.POSE_ROTATE_VEL, torch.tensor([0, 0, 0, 0, 0, 0, 1.0], device=dev), torch.zeros(0, 6, device=dev))
    assert out.shape == (0, 6)


def test_foot_height_matches_the_torch_gather(dev):
    """pd_foot_height against the torch gather over all candidates (oracle/pose_torch.py foot_height states the same formula): same heights, the same arg-min candidate wherever the
    minimum is not (nearly) tied, and the gradient torch.min routes to that candidate's body."""
    from diffphys_amd import hip_backend, robots

    tpl = robots.load_template("laikago")
    c_body = torch.tensor(tpl["contact_body"], dtype=torch.long, device=dev)
    c_point = torch.tensor(tpl["contact_point"], dtype=torch.float32, device=dev)
    c_dist = torch.tensor(tpl["contact_dist"], dtype=torch.float32, device=dev)
    rng = np.random.RandomState(2)
    nb = int(tpl["nb"])
    X = np.concatenate([rng.randn(32, 4, nb, 3) * 0.3, rng.randn(32, 4, nb, 4)], -1)
    X[..., 3:] /= np.linalg.norm(X[..., 3:], axis=-1, keepdims=True)
    x_hip = torch.tensor(X, dtype=torch.float32, device=dev, requires_grad=True)
    x_ref = x_hip.detach().clone().requires_grad_(True)

    def torch_height(state_body_q):
        Xc = state_body_q[..., c_body, :]
        q, p = Xc[..., 3:], Xc[..., :3]
        qv, w = q[..., :3], q[..., 3:]
        pt = c_point.expand(qv.shape)
        rot = pt * (2 * w * w - 1) + 2 * w * torch.cross(qv, pt, dim=-1) + 2 * qv * (qv * pt).sum(-1, keepdim=True)
        return (p[..., 1] + rot[..., 1] - c_dist).min(-1)

    h_ref, arg_ref = torch_height(x_ref)
    from diffphys_amd.phys_model import _FootHeightHip
    h = _FootHeightHip.apply(x_hip, c_body.int(), c_point, c_dist)
    assert h.shape == (32, 4) and torch.allclose(h, h_ref, rtol=0, atol=2e-6)
    _, arg = hip_backend.foot_height(x_hip.detach(), c_body.int(), c_point, c_dist)
    assert (arg.long() == arg_ref).float().mean().item() > 0.95  # ties / 1-ulp near-ties may pick a neighbour
    g = torch.tensor(rng.randn(32, 4), dtype=torch.float32, device=dev)
    h.backward(g)
    h_ref.backward(g)
    same = (arg.long() == arg_ref)[..., None, None].expand_as(x_ref.grad)
    assert torch.allclose(x_hip.grad[same], x_ref.grad[same], rtol=1e-5, atol=1e-6)
    assert ((x_hip.grad != 0).any(-1).sum(-1) == 1).all()  # exactly one body per pose set receives a gradient
    # a NaN pose propagates like torch.min
    bad = x_hip.detach().clone()
    bad[3, 1, 5, 4] = float("nan")
    hb, _ = hip_backend.foot_height(bad, c_body.int(), c_point, c_dist)
    assert torch.isnan(hb[3, 1]) and not torch.isnan(hb).sum().item() > 1


@pytest.mark.parametrize("n,k", [(0, 5), (1, 1), (7, 33), (760, 18), (7600, 256), (7600, 512), (25600, 257), (100003, 3)])
def test_colsum_against_float64_and_inside_a_replayed_graph(n, k, dev):
    """pd_colsum (bias gradients of the time-MLPs, gradient of a broadcast pose operand): column sums in a fixed order.  Against a float64
    sum (the error of n fp32 additions in 8 + 4 interleaved partial sums); bit-identical run to run; and -- the reason it exists -- the
    SECOND and third replay of a captured HIP graph return the sums of the data that is in the buffer THEN (torch's own sum(0) of such
    shapes returns the first replay's on this stack)."""
    from diffphys_amd import hip_backend

    g = torch.Generator(device="cpu").manual_seed(n * 1000 + k)
    x = (torch.randn(n, k, generator=g) * 3.0 + 0.5).to(dev)
    got = hip_backend.colsum(x)
    ref = x.double().sum(0)
    assert got.shape == (k,) and got.dtype == torch.float32
    scale = float(x.abs().double().sum(0).max()) if n else 1.0
    assert float((got.double() - ref).abs().max()) <= 2e-6 * max(scale, 1.0), float((got.double() - ref).abs().max())
    assert torch.equal(hip_backend.colsum(x), got)
    if n == 0:
        assert float(got.abs().max()) == 0.0
        return
    buf = x.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        hip_backend.colsum(buf)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            out = hip_backend.colsum(buf)
    torch.cuda.current_stream().wait_stream(side)
    for trial in range(3):
        fresh = (torch.randn(n, k, generator=g) * (trial + 1.0)).to(dev)
        buf.copy_(fresh)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, hip_backend.colsum(fresh)), trial


@pytest.mark.parametrize("n,m,kin", [(1, 128, 128), (2, 128, 128), (17, 128, 256), (119, 256, 128), (760, 256, 256), (7600, 256, 256), (7600, 256, 512), (1255, 128, 384),
                                     (30001, 256, 256)])
def test_linear_wgrad_on_the_matrix_cores_against_float64(n, m, kin, dev):
    """pd_linear_wgrad: gw = g^T x and gb = column sums of g for a linear layer's backward, v_mfma_f32_32x32x2_f32 tiles over sample slices
    then the slices in order.  Against float64 products (bar: 1e-6 of sum |g||x| per entry -- fp32 chains of n terms reach ~3e-7 of it);
    bit-identical run to run; the second and third replay of a captured graph give the products of the data in the buffers THEN; shapes the
    kernel does not serve return None (the caller keeps its BLAS path)."""
    from diffphys_amd import hip_backend

    gen = torch.Generator(device="cpu").manual_seed(n + 7 * m + 13 * kin)
    g = (torch.randn(n, m, generator=gen) * 0.7).to(dev)
    x = (torch.randn(n, kin, generator=gen) * 1.3 + 0.2).to(dev)
    gw, gb = hip_backend.linear_wgrad(g, x)
    assert gw.shape == (m, kin) and gb.shape == (m,)
    ref_w = g.double().t() @ x.double()
    ref_b = g.double().sum(0)
    bound_w = g.abs().double().t() @ x.abs().double()
    bound_b = g.abs().double().sum(0)
    assert float(((gw.double() - ref_w).abs() / (bound_w + 1e-30)).max()) < 1e-6
    assert float(((gb.double() - ref_b).abs() / (bound_b + 1e-30)).max()) < 1e-6
    gw2, gb2 = hip_backend.linear_wgrad(g, x)
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)
    gw3, none_b = hip_backend.linear_wgrad(g, x, want_bias=False)
    assert none_b is None and torch.equal(gw3, gw)
    assert hip_backend.linear_wgrad(g[:, :100].contiguous(), x) is None and hip_backend.linear_wgrad(g, x[:, :21].contiguous()) is None
    gbuf, xbuf = g.clone(), x.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        hip_backend.linear_wgrad(gbuf, xbuf)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            ow, ob = hip_backend.linear_wgrad(gbuf, xbuf)
    torch.cuda.current_stream().wait_stream(side)
    for trial in range(3):
        g2 = (torch.randn(n, m, generator=gen) * (trial + 0.5)).to(dev)
        x2 = torch.randn(n, kin, generator=gen).to(dev)
        gbuf.copy_(g2); xbuf.copy_(x2)
        graph.replay()
        torch.cuda.synchronize()
        ew, eb = hip_backend.linear_wgrad(g2, x2)
        assert torch.equal(ow, ew) and torch.equal(ob, eb), trial
