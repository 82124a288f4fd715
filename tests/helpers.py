"""Shared helpers for the test-suite: hand-built templates and error metrics."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ppr-diffphys_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from diffphys_amd import sim  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
INPUT_NAMES = ("q_init", "qd_init", "torques", "res_f", "refs", "target_ke", "target_kd", "body_mass", "body_inv_mass",
               "body_inertia", "body_inv_inertia")


def relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def build_template(builder, attach_ke=1000.0, attach_kd=10.0, kp=0.0, kd=0.0):
    top = sim.ModelBuilder()
    top.add_rigid_articulation(builder)
    env = top.finalize("cpu")
    env.joint_attach_ke, env.joint_attach_kd = attach_ke, attach_kd
    env.collide(None)
    tpl = env.template()
    tpl["kp"], tpl["kd"] = np.float32(kp), np.float32(kd)
    return tpl


def free_body(shape="sphere", radius=0.1, half=(0.1, 0.1, 0.1), mat=(1.0e4, 0.0, 1.0e2, 1.0)):
    """One FREE body with one collision shape at its origin."""
    b = sim.ModelBuilder()
    b.add_articulation()
    body = b.add_body(origin=sim.transform_identity(), parent=-1, joint_type=sim.JOINT_FREE, joint_armature=0.01)
    ke, kd, kf, mu = mat
    if shape == "sphere":
        b.add_shape_sphere(body, radius=radius, density=1000.0, ke=ke, kd=kd, kf=kf, mu=mu)
    elif shape == "box":
        b.add_shape_box(body, hx=half[0], hy=half[1], hz=half[2], density=1000.0, ke=ke, kd=kd, kf=kf, mu=mu)
    return b


def chain2(joint_type, axis=(1.0, 0.0, 0.0), origin=(0.0, -0.3, 0.0)):
    """FREE root + one child (revolute / compound / fixed); boxes so masses are well conditioned."""
    b = sim.ModelBuilder()
    b.add_articulation()
    root = b.add_body(origin=sim.transform_identity(), parent=-1, joint_type=sim.JOINT_FREE, joint_armature=0.01)
    b.add_shape_box(root, hx=0.1, hy=0.1, hz=0.1, density=1000.0, ke=1e4, kd=0.0, kf=1e2, mu=1.0)
    kw = {}
    if joint_type == sim.JOINT_COMPOUND:
        kw = dict(joint_target_ke=[0.0] * 3, joint_target_kd=[0.0] * 3, joint_limit_lower=[-1e3] * 3, joint_limit_upper=[1e3] * 3,
                  joint_limit_ke=0.0, joint_limit_kd=0.0)
    else:
        kw = dict(joint_axis=axis, joint_limit_ke=0.0, joint_limit_kd=0.0)
    child = b.add_body(origin=sim.transform_identity(), parent=root, joint_xform=sim.transform(origin, sim.quat_rpy(0.1, -0.2, 0.3)),
                       joint_type=joint_type, joint_armature=0.01, **kw)
    b.add_shape_box(child, pos=(0.0, -0.1, 0.02), hx=0.05, hy=0.1, hz=0.05, density=1000.0, ke=1e4, kd=0.0, kf=1e2, mu=1.0)
    return b


def default_inputs(tpl, bs, nsteps, dtype=np.float64, seed=0):
    """Inputs with template masses, zero controls; q_init from the template's joint_q."""
    nb, nq, nqd = int(tpl["nb"]), int(tpl["nq"]), int(tpl["nqd"])
    mass = np.tile(tpl["body_mass"].astype(np.float64), bs)
    inertia = np.tile(tpl["body_inertia"].astype(np.float64), (bs, 1, 1))
    d = dict(
        q_init=np.tile(tpl["joint_q"].astype(np.float64), bs), qd_init=np.zeros(bs * nqd), torques=np.zeros((nsteps, bs * nqd)),
        res_f=np.zeros((nsteps, bs * nb, 6)), refs=np.zeros((nsteps, bs * nqd)), target_ke=np.zeros(bs * nqd),
        target_kd=np.zeros(bs * nqd), body_mass=mass, body_inv_mass=1.0 / mass, body_inertia=inertia,
        body_inv_inertia=np.linalg.inv(inertia),
    )
    return {k: np.ascontiguousarray(v.astype(dtype)) for k, v in d.items()}


def load_golden(name):
    with np.load(os.path.join(GOLDEN, "rollout_%s.npz" % name)) as z:
        return {k: z[k] for k in z.files}


def golden_inputs(g):
    inp = {k: g["in_" + k] for k in INPUT_NAMES + ("adj_pos", "adj_vel")}
    inp["frame2step"] = [int(x) for x in g["frame2step"]]
    inp["nsteps"], inp["dt"] = int(g["nsteps"]), float(g["dt"])
    return inp


def tight_inputs(tpl, robot, bs, T, seed):
    """Non-singular inputs for the short-horizon tight-tolerance tests: feet a few mm in the ground, joint angles away
    from their references (no acos at 1), non-zero twists, torques, residual wrenches, perturbed gains / masses, and
    frames at states 0 and T (so every gradient is exercised through exactly T steps)."""
    from diffphys_amd import synth

    inp = synth.make_inputs(tpl, robot, bs=bs, nsteps=T, seed=seed, steps_per_frame=max(T, 1), penetration=0.004)
    rng = np.random.RandomState(seed + 1)
    nb, nqd = int(tpl["nb"]), int(tpl["nqd"])
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    inp["qd_init"] = f32(rng.randn(bs * nqd) * 0.2)
    inp["torques"] = f32(rng.randn(T, bs * nqd) * 0.5)
    inp["res_f"] = f32(rng.randn(T, bs * nb, 6) * 0.5)
    inp["refs"] = f32(inp["refs"] + rng.uniform(0.05, 0.15, inp["refs"].shape) * np.sign(rng.randn(*inp["refs"].shape)))
    inp["refs"].reshape(T, bs, nqd)[:, :, :6] = 0
    inp["target_ke"] = f32(inp["target_ke"] * rng.uniform(0.8, 1.2, bs * nqd))
    inp["target_kd"] = f32(inp["target_kd"] * rng.uniform(0.8, 1.2, bs * nqd))
    scale = rng.uniform(0.8, 1.25, bs * nb)
    mass = inp["body_mass"].astype(np.float64) * scale
    inertia = inp["body_inertia"].astype(np.float64) * scale[:, None, None]
    inp.update(body_mass=f32(mass), body_inv_mass=f32(1.0 / mass), body_inertia=f32(inertia),
               body_inv_inertia=f32(np.linalg.inv(inertia)))
    inp["frame2step"] = [0, T]
    inp["adj_pos"] = f32(rng.randn(2, bs * nb, 7))
    inp["adj_vel"] = f32(rng.randn(2, bs * nb, 6))
    return inp


def per_env(a, bs, lead):
    """[lead..., bs*n, ...] -> [bs, -1]: groups a flat env-major tensor by env (lead = number of leading step/frame axes)."""
    a = np.asarray(a, np.float64)
    if lead == 0:
        return a.reshape(bs, -1)
    t = a.shape[0]
    return np.moveaxis(a.reshape(t, bs, -1), 1, 0).reshape(bs, -1)


GRAD_LEAD = dict(q_init=0, qd_init=0, torques=1, res_f=1, refs=1, target_ke=0, target_kd=0, body_inv_mass=0, body_inertia=0,
                 body_inv_inertia=0)


def grad_env_errors(g, r, bs, abs_floor=0.0):
    """Per gradient tensor: per-env max |g - r| relative to that env's max |r| (envs whose reference gradient is
    identically zero are compared absolutely against the tensor's max).  abs_floor: the smallest scale an env's tensor is measured
    on (scripts/gpu_stress.py: a 2-step rollout that starts ON its joint targets has d loss / d target_ke ~ 1e-18 in float64 and
    ~1e-11 of rounding noise in fp32 -- "relative error 1e8" of a gradient that is zero for every purpose)."""
    out = {}
    for k, lead in GRAD_LEAD.items():
        a, b = per_env(g[k], bs, lead), per_env(r[k], bs, lead)
        scale = np.abs(b).max(1)
        scale = np.where(scale > 0, scale, np.abs(b).max() + 1e-30)
        out[k] = np.abs(a - b).max(1) / np.maximum(scale, abs_floor)
    return out



def first_branch_difference(rc64, st64, traj, inp, bs, height_tol=5e-7, coulomb_tol=1e-3, force_clamp_tol=2e-2):
    """First step (per env, or nsteps when there is none) at which the GPU rollout took a discrete decision the float64 oracle
    did not: a different number of touching contact candidates on some body, a different number of them on the sliding branch of
    the Coulomb min, a different velocity-clamp mask (the kernel's own, stored with its trajectory) -- or a candidate within
    `height_tol` of the ground in the kernel's state, where its fp32 height can fall on either side, or a touching candidate whose
    two friction bounds (kf |vt| and -mu (fn + fd), integrator_euler.py:158-165) are within `coulomb_tol` newtons of each other, where the
    kernel's fp32 comparison can pick the other branch of the min (kf = 100 N s/m: 1e-5 m/s of fp32 noise in a point velocity is 1e-3 N;
    found by the round-3 stress sweep: seed 9001 case 3933, one env 0.27 off in its gradients with poses equal to 1e-7, |a - b| =
    2e-4 N at step 4, float64 slides where the fp32 arithmetic sticks).  The oracle's branch log is
    evaluated in float64 on the oracle's trajectory and on the kernel's saved trajectory (oracle/ref_c.py branch_log).
    Envs WITHOUT such a step differentiate the same smooth function as the oracle."""
    nsteps = inp["nsteps"]
    nb = rc64.nb
    ref = rc64.branch_log(st64)
    gpu = rc64.branch_log(dict(states_q=traj["states_q"], states_qd=traj["states_qd"], states_f=traj["states_f"]), inp)
    clamp_gpu = traj["clamp"].reshape(nsteps, bs, nb) & 63
    diff = (ref["touch"] != gpu["touch"]).any(2) | (ref["slide"] != gpu["slide"]).any(2) | (ref["clamp"] != clamp_gpu).any(2)
    st_g = dict(states_q=traj["states_q"], states_qd=traj["states_qd"], states_f=traj["states_f"], _inputs={k: rc64._c(inp[k]) for k in
                ("body_inv_mass", "body_inertia", "body_inv_inertia")}, _bs=bs, _nsteps=nsteps, _dt=inp["dt"])
    st_g = {k: (rc64._c(v) if k.startswith("states") else v) for k, v in st_g.items()}
    probe = rc64.singularity_probe(st_g)
    diff |= probe[:, :, 0] < height_tol
    diff |= probe[:, :, 2] < coulomb_tol
    # ... or a contact force component within `force_clamp_tol` newtons of the +-500 N clamp (integrator_euler.py:172-175: 4e-5 of 500 N; found by
    # the stress sweep's seed 31337 case 9764: a robot dropped exactly 0.05 m into the ground, ke = 1e4 N/m => fn = 500 N on every point)
    diff |= probe[:, :, 4] < force_clamp_tol
    first = np.where(diff.any(0), diff.argmax(0), nsteps)
    return first



def oracle_bundle(tpl, inp, bs):
    """Everything the explained per-env gradient bars need: the float64 C oracle (trajectory, gradients), the fp32 C oracle's
    gradients, and the gradients of the float64 oracle with its stored states rounded to fp32 (two samples: to nearest, and to an
    adjacent fp32 number) -- the least any fp32 implementation does to a rollout (oracle/ref_c/diffphys_ref.c:
    ref_set_state_rounding).  `cond`: per env, the largest worst-tensor distance of these from the float64 gradients = how
    ill-conditioned the env is; `e_round`: the same from the rounded-state runs alone (no fp32 arithmetic at all)."""
    from oracle.ref_c import RefC

    rc64, rc32 = RefC(tpl, np.float64), RefC(tpl, np.float32)
    T, f2s, dt = inp["nsteps"], inp["frame2step"], inp["dt"]
    st64 = rc64.rollout_forward(inp, T, f2s, dt)
    g64 = rc64.rollout_backward(st64, inp["adj_pos"], inp["adj_vel"])
    st32 = rc32.rollout_forward(inp, T, f2s, dt)
    g32 = rc32.rollout_backward(st32, inp["adj_pos"], inp["adj_vel"])
    worst = lambda g: np.max(np.stack([v for v in grad_env_errors(g, g64, bs).values()]), axis=0)
    rounded = []
    try:
        for mode in (1, 2):
            rc64.set_state_rounding(mode)
            st_r = rc64.rollout_forward(inp, T, f2s, dt)
            rounded.append(worst(rc64.rollout_backward(st_r, inp["adj_pos"], inp["adj_vel"])))
    finally:
        rc64.set_state_rounding(0)
    e_round = np.maximum(rounded[0], rounded[1])
    # cond = the rounded-state runs ALONE (the fp32 oracle's literal acos is 1e-3 .. 1 off in most 100-step Laikago
    # rollouts and bounded little); cond_fp32 keeps the old, looser scale for printing
    return dict(rc64=rc64, st64=st64, g64=g64, st32=st32, g32=g32, cond=e_round, cond_fp32=np.maximum(worst(g32), e_round), e_round=e_round)


def own_trajectory_check(dm, tpl, inp, dev, hitlog_check=True, abs_floor=0.0, literal=False):
    """The airtight gradient comparison: the kernel's gradients against the float64 C oracle's adjoint OF THE
    KERNEL'S OWN saved trajectory -- same linearisation point, so the chaos of a 100-step rollout is not in the comparison -- with the
    kernel's discrete decisions: its stored velocity-clamp masks, and "this candidate touches" by its pinned fp32 height test
    (oracle/ref_c/diffphys_ref.c: touch_pinned_fp32, cross-checked here against the hit log the forward kernel wrote).  What stays
    re-decided in float64 -- the Coulomb min and the +-500 N force clamp -- is probed on the kernel's states (singularity_probe).
    Returns dict(worst [bs] = per env the largest per-tensor relative error (grad_env_errors), errors per tensor, probe minima
    per env: coulomb, force_clamp, height; grads of the oracle)."""
    import torch

    from oracle.ref_c import RefC

    bs = inp["q_init"].size // dm.nq
    T, f2s = inp["nsteps"], list(inp["frame2step"])
    FWD = ("q_init", "qd_init", "torques", "res_f", "refs", "target_ke", "target_kd", "body_inv_mass", "body_inertia", "body_inv_inertia")
    BWD = ("q_init", "qd_init", "torques", "refs", "target_ke", "target_kd", "body_inv_mass", "body_inertia", "body_inv_inertia")
    t = {k: torch.from_numpy(np.ascontiguousarray(inp[k], dtype=np.float32)).to(dev) for k in INPUT_NAMES + ("adj_pos", "adj_vel")}
    pos, vel, grf, jaf, ws = dm.rollout_forward(bs, T, inp["dt"], *[t[k] for k in FWD], frame2step=f2s)
    g = dm.rollout_backward(bs, T, inp["dt"], *[t[k] for k in BWD], f2s, ws, t["adj_pos"], t["adj_vel"])
    grads = {k: v.cpu().numpy() for k, v in g.items()}
    bq, bqd, bf, mask = dm.saved_trajectory(ws, bs, T)
    traj = dict(states_q=bq.cpu().numpy(), states_qd=bqd.cpu().numpy(), states_f=bf.cpu().numpy())
    mask = mask.cpu().numpy()
    rc = RefC(tpl, np.float64)
    # A model with FIXED joints: their angular error is evaluated scale-invariantly by the kernels (pd_math.h fixed_ang_h); the float64
    # reference must do the same (ref_set_twist_eval(1)) -- its literal form turns the fp32 norm error of the stored quaternions into
    # spurious angles (oracle/ref_c/diffphys_ref.c fixed_ang_h).  For the other joint types the switch changes nothing in float64.
    # literal=True: the device model runs under PD_NUM_LITERAL -- the kernels evaluate the reference's text, so the float64 reference does too
    has_fixed = 3 in set(int(t) for t in np.asarray(tpl["joint_type"]))
    rc.set_twist_eval(has_fixed and not literal)
    try:
        st = rc.trajectory_state(traj, inp)
        g64 = rc.rollout_backward_forced(st, inp["adj_pos"], inp["adj_vel"], clamp_mask=mask, pinned_touch=True)
        errs = grad_env_errors(grads, g64, bs, abs_floor)
        worst = np.max(np.stack([errs[k] for k in GRAD_LEAD]), axis=0)
        # conditioning of the adjoint ON THIS FIXED TRAJECTORY: how far the float64 gradients move when every stored fp32 value (state and
        # total wrench) is replaced by an adjacent fp32 number, all decisions held (two random sign patterns).  No forward pass is re-run, so
        # no chaos enters: it is what the last stored bit is worth -- the precision at which ANY fp32 evaluation sees the joint gaps
        # (x_err = a difference of ~0.5 m positions, 1e-5 .. 1e-4 m long on Laikago's 16 kN/m springs)
        tl = rc.touch_fp32(st, cap=64)
        cond = np.zeros(bs)
        for seed in (0, 1):
            rng = np.random.RandomState(1234 + seed)
            jit = {}
            for k, v in traj.items():
                v = np.asarray(v, np.float32)
                jit[k] = np.nextafter(v, np.where(rng.rand(*v.shape) < 0.5, -np.inf, np.inf).astype(np.float32))
            g_j = rc.rollout_backward_forced(rc.trajectory_state(jit, inp), inp["adj_pos"], inp["adj_vel"], clamp_mask=mask, touch_list=tl)
            e_j = grad_env_errors(g_j, g64, bs, abs_floor)
            cond = np.maximum(cond, np.max(np.stack([e_j[k] for k in GRAD_LEAD]), axis=0))
    finally:
        rc.set_twist_eval(False)
    # a plain fp32 evaluation of the same adjoint on the same trajectory with the same decisions: the fp32 build of the C oracle,
    # once with the literal twist angle (2 acos(twist.w): what an fp32 tape of the reference's text does) and once through atan2 (the
    # kernels' evaluation of that function) -- the yardstick for "as accurate as fp32 arithmetic allows"
    rc32 = RefC(tpl, np.float32)
    st32 = rc32.trajectory_state(traj, inp)
    e32 = {}
    try:
        for tag, sw in (("acos", False), ("atan2", True)):
            rc32.set_twist_eval(sw)
            g32 = rc32.rollout_backward_forced(st32, inp["adj_pos"], inp["adj_vel"], clamp_mask=mask, touch_list=tl)
            e = grad_env_errors(g32, g64, bs, abs_floor)
            e32[tag] = np.max(np.stack([e[k] for k in GRAD_LEAD]), axis=0)
    finally:
        rc32.set_twist_eval(False)
    probe = rc.singularity_probe(st)  # [T, bs, 5] on the kernel's own states
    out = dict(worst=worst, errs=errs, cond=cond, fp32_acos=e32["acos"], fp32_atan2=e32["atan2"], touch_counts=tl[..., 0].copy(), g64=g64, grads=grads, height=probe[:, :, 0].min(0), coulomb=probe[:, :, 2].min(0),
               force_clamp=probe[:, :, 4].min(0), clamp_dist=probe[:, :, 1].min(0), traj=traj, mask=mask)
    if hitlog_check:
        # the restated touch decision against the kernel's own record: every candidate it says touches is in the hit log of that
        # env-step (the log is the exact list in speculated steps and a superset -- cull survivors -- where the exact sweep ran)
        log = dm.saved_hit_log(ws, bs, T)            # [T, bs, 32] template indices
        tch = rc.touch_fp32(st, cap=32)               # [T, bs, 32]
        ok = log[..., 0] >= 0
        n_t = tch[..., 0]
        missing = 0
        for j in range(31):
            has = (j < n_t) & ok
            if not has.any():
                break
            k = tch[..., 1 + j]
            missing += int((has & ~(log[..., 1:] == k[..., None]).any(-1)).sum())
        out.update(hitlog_missing=missing, hitlog_overflow=int((~ok).sum()), touches=int(n_t.sum()),
                   log_entries=int(np.maximum(log[..., 0], 0).sum()))
    return out


def toy_template(tmp_path, attach_ke=8000.0, attach_kd=200.0):
    """The toy robot of tests/test_host.py (free + revolute + compound + fixed joints; box / sphere / mesh / capsule contacts) through the
    model compiler: the GENERIC kernel instantiation, which none of the three shipped robots uses."""
    from test_host import OBJ, URDF
    from diffphys_amd import sim
    from diffphys_amd.import_urdf import parse_urdf

    (tmp_path / "toy.urdf").write_text(URDF)
    (tmp_path / "tet.obj").write_text(OBJ)
    b = sim.ModelBuilder()
    parse_urdf(str(tmp_path / "toy.urdf"), b, xform=sim.transform((0, 0.5, 0), sim.quat_identity()), floating=True, density=1000.0,
               armature=0.01, stiffness=220.0, damping=2.0, shape_ke=1e4, shape_kd=10.0, shape_kf=1e2, shape_mu=0.7, limit_ke=50.0, limit_kd=1.0)
    return build_template(b, attach_ke=attach_ke, attach_kd=attach_kd)


def toy_inputs(tpl, bs, T, frame2step, seed=0):
    """float32 rollout inputs for the toy robot: feet in the ground, joints off their references, non-zero twists / torques / residuals"""
    nb, nq, nqd = int(tpl["nb"]), int(tpl["nq"]), int(tpl["nqd"])
    rng = np.random.RandomState(seed)
    q = np.tile(tpl["joint_q"].astype(np.float64), (bs, 1))
    q[:, 1] = 0.13 + rng.rand(bs) * 0.02
    q[:, 7:] = rng.uniform(-0.4, 0.4, (bs, nq - 7))
    mass = np.tile(tpl["body_mass"].astype(np.float64), bs)
    inertia = np.tile(tpl["body_inertia"].astype(np.float64), (bs, 1, 1))
    ke = np.tile(np.r_[np.zeros(6), np.full(nqd - 6, 60.0)], bs)
    F = len(frame2step)
    inp = dict(q_init=q.reshape(-1), qd_init=rng.randn(bs * nqd) * 0.2, torques=rng.randn(T, bs * nqd) * 0.3,
               res_f=rng.randn(T, bs * nb, 6) * 0.3, refs=rng.uniform(-0.3, 0.3, (T, bs * nqd)), target_ke=ke, target_kd=ke * 0.02,
               body_mass=mass, body_inv_mass=1 / mass, body_inertia=inertia, body_inv_inertia=np.linalg.inv(inertia),
               adj_pos=rng.randn(F, bs * nb, 7) * 1e-3, adj_vel=rng.randn(F, bs * nb, 6) * 1e-3)
    inp = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in inp.items()}
    inp.update(frame2step=list(frame2step), nsteps=T, dt=5e-4)
    return inp
