"""Seeded synthetic workloads of SURVEY.md section 8(d) (host side, numpy only).

Produces the 11 flat, env-major input tensors of ``ForwardWarp.apply``
(/root/reference/diffphys/dp_model.py:733-746) for a robot template:
Laikago envs track the AMP mocap joint angles (the compiled fixture
``templates/mocap_laikago.npz``); human / quad use the sinusoidal references
the survey prescribes.  The root height is set so the lowest ground-contact
candidate touches y = 0 in the initial pose.
"""
import numpy as np

from . import sim
from .dataloader import DataLoader, bullet2gl, parse_amp

DT = 5e-4  # /root/reference/diffphys/dp_model.py:57


def _qmul(a, b):
    ax, ay, az, aw = np.moveaxis(a, -1, 0)
    bx, by, bz, bw = np.moveaxis(b, -1, 0)
    return np.stack(
        [aw * bx + bw * ax + ay * bz - az * by, aw * by + bw * ay + az * bx - ax * bz,
         aw * bz + bw * az + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz], -1)


def _qrot(q, v):
    qv, w = q[..., :3], q[..., 3:]
    return v * (2 * w * w - 1) + 2 * w * np.cross(qv, v) + 2 * qv * (qv * v).sum(-1, keepdims=True)


def _qaa(axis, ang):
    return np.concatenate([axis * np.sin(ang * 0.5)[..., None], np.cos(ang * 0.5)[..., None]], -1)


def fk_pose_np(tpl, joint_q):
    """Body poses [bs,nb,7] from joint coords [bs,nq] (float64 numpy; poses only)."""
    nb = int(tpl["nb"])
    bs = joint_q.shape[0]
    out = np.zeros((bs, nb, 7))
    e = np.eye(3)
    for i in range(nb):
        par, ty, qs = int(tpl["joint_parent"][i]), int(tpl["joint_type"][i]), int(tpl["joint_q_start"][i])
        if par >= 0:
            p_wp, q_wp = out[:, par, :3], out[:, par, 3:]
        else:
            p_wp, q_wp = np.zeros((bs, 3)), np.tile([0.0, 0.0, 0.0, 1.0], (bs, 1))
        p_jc = np.zeros((bs, 3))
        if ty == sim.JOINT_REVOLUTE:
            q_jc = _qaa(np.tile(tpl["joint_axis"][i].astype(np.float64), (bs, 1)), joint_q[:, qs])
        elif ty == sim.JOINT_FREE:
            p_jc, q_jc = joint_q[:, qs:qs + 3], joint_q[:, qs + 3:qs + 7]
        elif ty == sim.JOINT_COMPOUND:
            q0 = _qaa(np.tile(e[0], (bs, 1)), joint_q[:, qs])
            a1 = _qrot(q0, np.tile(e[1], (bs, 1)))
            q1 = _qaa(a1, joint_q[:, qs + 1])
            a2 = _qrot(_qmul(q1, q0), np.tile(e[2], (bs, 1)))
            q2 = _qaa(a2, joint_q[:, qs + 2])
            q_jc = _qmul(q2, _qmul(q1, q0))
        elif ty == sim.JOINT_FIXED:
            q_jc = np.tile([0.0, 0.0, 0.0, 1.0], (bs, 1))
        else:
            raise NotImplementedError(ty)
        Xp = tpl["joint_X_p"][i].astype(np.float64)
        p_wj = p_wp + _qrot(q_wp, np.tile(Xp[:3], (bs, 1)))
        q_wj = _qmul(q_wp, np.tile(Xp[3:], (bs, 1)))
        out[:, i, :3] = p_wj + _qrot(q_wj, p_jc)
        out[:, i, 3:] = _qmul(q_wj, q_jc)
    return out


def lowest_contact_y(tpl, joint_q):
    """min over contact candidates of world y, per env (only the y-row of each body's rotation is needed)."""
    bq = fk_pose_np(tpl, joint_q)
    x, y, z, w = np.moveaxis(bq[..., 3:], -1, 0)
    Ry = np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1)  # [bs, nb, 3]
    cb = tpl["contact_body"].astype(np.int64)
    pts = tpl["contact_point"].astype(np.float64)
    out = np.full(bq.shape[0], np.inf)
    for b in np.unique(cb):
        sel = cb == b
        yb = bq[:, b, 1][:, None] + Ry[:, b] @ pts[sel].T - tpl["contact_dist"][sel][None]
        out = np.minimum(out, yb.min(-1))
    return out


def make_inputs(tpl, robot, bs, nsteps, seed=0, seqs=("mi-pace",), steps_per_frame=33, dt=DT, dtype=np.float32,
                penetration=0.0):
    """Returns a dict with the 11 inputs (numpy), ``frame2step`` and the upstream
    gradient seeds ``adj_pos`` / ``adj_vel`` of SURVEY.md section 8(d).  ``penetration`` lowers the root by
    that many metres below the "lowest point touches y = 0" pose (parity tests use a few mm so that the
    initial contacts are not sitting exactly on the non-differentiable c = 0 boundary)."""
    rng = np.random.RandomState(seed)
    nb, nq, nqd = int(tpl["nb"]), int(tpl["nq"]), int(tpl["nqd"])
    ndof = nqd - 6
    frame2step = list(range(0, nsteps, steps_per_frame))
    t = np.arange(nsteps)
    q_init = np.zeros((bs, nq))
    yaw = rng.uniform(-0.1, 0.1, size=bs)
    root_q = np.stack([np.zeros(bs), np.sin(yaw / 2), np.zeros(bs), np.cos(yaw / 2)], -1)  # about +y
    refs = np.zeros((nsteps, bs, nqd))
    if robot == "laikago":
        amps = {s: DataLoader({"seqname": s}).amp_info for s in set(seqs)}
        for e in range(bs):
            amp = amps[seqs[e % len(seqs)]]
            nfr = len(amp)
            span = (nsteps - 1) / steps_per_frame
            f0 = rng.randint(0, max(1, int(nfr - span - 1)))
            fr = f0 + t / steps_per_frame
            lo = np.clip(np.floor(fr).astype(int), 0, nfr - 2)
            a = (fr - lo)[:, None]
            rows = amp[lo] * (1 - a) + amp[lo + 1] * a  # linear interpolation of AMP rows (dp_model.py:421-427)
            msm = parse_amp(rows)
            refs[:, e, 6:] = msm["jang"]
            q_init[e, 7:] = msm["jang"][0]
    else:
        q_init[:, 7:] = rng.uniform(-0.2, 0.2, size=(bs, ndof))
        phase = rng.uniform(0, 2 * np.pi, size=(bs, ndof))
        refs[:, :, 6:] = q_init[None, :, 7:] + 0.3 * np.sin(2 * np.pi * 2.0 * (t * dt)[:, None, None] + phase[None])
    q_init[:, 3:7] = root_q
    h = -lowest_contact_y(tpl, q_init)
    q_init[:, 1] = h - penetration
    qd_init = np.zeros((bs, nqd))
    torques = np.zeros((nsteps, bs * nqd))
    res_f = np.zeros((nsteps, bs * nb, 6))
    kp, kd = float(tpl["kp"]), float(tpl["kd"])
    target_ke = np.tile(np.r_[np.zeros(6), np.full(ndof, kp)], bs)
    target_kd = np.tile(np.r_[np.zeros(6), np.full(ndof, kd)], bs)
    mass = np.tile(tpl["body_mass"].astype(np.float64), bs)
    inertia = np.tile(tpl["body_inertia"].astype(np.float64), (bs, 1, 1)) * mass[:, None, None]
    inv_inertia = np.linalg.inv(inertia)
    F = len(frame2step)
    out = dict(
        q_init=q_init.reshape(-1), qd_init=qd_init.reshape(-1), torques=torques, res_f=res_f,
        refs=refs.reshape(nsteps, bs * nqd), target_ke=target_ke, target_kd=target_kd,
        body_mass=mass, body_inv_mass=1.0 / mass, body_inertia=inertia, body_inv_inertia=inv_inertia,
        adj_pos=rng.randn(F, bs * nb, 7) * 1e-3, adj_vel=rng.randn(F, bs * nb, 6) * 1e-3,
    )
    out = {k: np.ascontiguousarray(v.astype(dtype)) for k, v in out.items()}
    out["frame2step"] = frame2step
    out["nsteps"] = nsteps
    out["dt"] = dt
    return out


def make_env_inputs(tpl, robot, env_ids, nsteps, seed=0, seqs=("mi-pace",), steps_per_frame=33, dt=DT, dtype=np.float32,
                    penetration=0.0):
    """Same workload as :func:`make_inputs`, but every env draws from its OWN generator seeded by (seed, global env id):
    the inputs of env e do not depend on which other envs are built with it.  A rank of a multi-GPU run builds exactly
    its contiguous slice of the global batch (bench.py) and the concatenation of the slices is
    the global batch, bit for bit, for any world size."""
    env_ids = np.asarray(env_ids, dtype=np.int64)
    bs = len(env_ids)
    nb, nq, nqd = int(tpl["nb"]), int(tpl["nq"]), int(tpl["nqd"])
    ndof = nqd - 6
    frame2step = list(range(0, nsteps, steps_per_frame))
    F = len(frame2step)
    t = np.arange(nsteps)
    q_init = np.zeros((bs, nq))
    refs = np.zeros((nsteps, bs, nqd))
    adj_pos = np.zeros((F, bs, nb, 7))
    adj_vel = np.zeros((F, bs, nb, 6))
    amps = {s: DataLoader({"seqname": s}).amp_info for s in set(seqs)} if robot == "laikago" else {}
    for i, e in enumerate(env_ids):
        rng = np.random.RandomState([int(seed) & 0x7fffffff, int(e)])
        yaw = rng.uniform(-0.1, 0.1)
        q_init[i, 3:7] = [0.0, np.sin(yaw / 2), 0.0, np.cos(yaw / 2)]
        if robot == "laikago":
            amp = amps[seqs[int(e) % len(seqs)]]
            nfr = len(amp)
            span = (nsteps - 1) / steps_per_frame
            f0 = rng.randint(0, max(1, int(nfr - span - 1)))
            fr = f0 + t / steps_per_frame
            lo = np.clip(np.floor(fr).astype(int), 0, nfr - 2)
            a = (fr - lo)[:, None]
            msm = parse_amp(amp[lo] * (1 - a) + amp[lo + 1] * a)
            refs[:, i, 6:] = msm["jang"]
            q_init[i, 7:] = msm["jang"][0]
        else:
            q_init[i, 7:] = rng.uniform(-0.2, 0.2, size=ndof)
            phase = rng.uniform(0, 2 * np.pi, size=ndof)
            refs[:, i, 6:] = q_init[i, 7:][None] + 0.3 * np.sin(2 * np.pi * 2.0 * (t * dt)[:, None] + phase[None])
        adj_pos[:, i] = rng.randn(F, nb, 7) * 1e-3
        adj_vel[:, i] = rng.randn(F, nb, 6) * 1e-3
    q_init[:, 1] = -lowest_contact_y(tpl, q_init) - penetration if bs else 0.0
    kp, kd = float(tpl["kp"]), float(tpl["kd"])
    mass = np.tile(tpl["body_mass"].astype(np.float64), bs)
    inertia = np.tile(tpl["body_inertia"].astype(np.float64), (bs, 1, 1)) * mass[:, None, None]
    out = dict(
        q_init=q_init.reshape(-1), qd_init=np.zeros(bs * nqd), torques=np.zeros((nsteps, bs * nqd)),
        res_f=np.zeros((nsteps, bs * nb, 6)), refs=refs.reshape(nsteps, bs * nqd),
        target_ke=np.tile(np.r_[np.zeros(6), np.full(ndof, kp)], bs), target_kd=np.tile(np.r_[np.zeros(6), np.full(ndof, kd)], bs),
        body_mass=mass, body_inv_mass=1.0 / mass, body_inertia=inertia,
        body_inv_inertia=np.linalg.inv(inertia) if bs else inertia,
        adj_pos=adj_pos.reshape(F, bs * nb, 7), adj_vel=adj_vel.reshape(F, bs * nb, 6),
    )
    out = {k: np.ascontiguousarray(v.astype(dtype)) for k, v in out.items()}
    out.update(frame2step=frame2step, nsteps=nsteps, dt=dt)
    return out


def concat_envs(parts, nb):
    """Concatenates per-slice input / output dicts along the env axis (env-major flat layouts of SURVEY.md row a7)."""
    out = {}
    for k, v in parts[0].items():
        if not isinstance(v, np.ndarray):
            out[k] = v
        elif k in ("torques", "refs", "res_f", "adj_pos", "adj_vel", "wp_pos", "wp_vel", "grf", "jaf"):  # [T or F, bs*n, ...]
            out[k] = np.concatenate([p[k] for p in parts], axis=1)
        else:
            out[k] = np.concatenate([p[k] for p in parts], axis=0)
    return out


INPUT_NAMES = ("q_init", "qd_init", "torques", "res_f", "refs", "target_ke", "target_kd",
               "body_mass", "body_inv_mass", "body_inertia", "body_inv_inertia")
