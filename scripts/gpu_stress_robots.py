#!/usr/bin/env python3
"""Random ROBOTS through the model compiler and the kernels: random trees of 2 .. 40 bodies, joints drawn from revolute / compound (the
*_R / *_P / *_Y triples) / fixed, random axes and joint frames (rotated child frames send a model to the generic kernels), box and sphere
collisions.  Per robot: forward against the C oracle with the fp32 oracle's own loss against float64 as the yardstick, gradients against the
float64 adjoint of the kernel's own trajectory (both kernel families when the quad-lane one is eligible), a zero-step rollout, run-to-run
bits.  Usage: gpu_stress_robots.py [nrobots] [seed]; exits non-zero on a violation."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ppr-diffphys_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from diffphys_amd import hip_backend, sim
from diffphys_amd.import_urdf import parse_urdf
from helpers import build_template, own_trajectory_check, relmax
from oracle.ref_c import RefC

dev = torch.device("cuda:0")
nrob = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
FWD = ("q_init", "qd_init", "torques", "res_f", "refs", "target_ke", "target_kd", "body_inv_mass", "body_inertia", "body_inv_inertia")
BWD = ("q_init", "qd_init", "torques", "refs", "target_ke", "target_kd", "body_inv_mass", "body_inertia", "body_inv_inertia")
tmp = tempfile.mkdtemp()


def make_urdf(kinds, n_joints, rotated, branchy):
    def coll():
        o = "%.3f %.3f %.3f" % tuple(rng.uniform(-0.03, 0.03, 3))
        if rng.rand() < 0.6:
            return '<collision><origin xyz="%s"/><geometry><box size="%.3f %.3f %.3f"/></geometry></collision>' % ((o,) + tuple(rng.uniform(0.04, 0.1, 3)))
        return '<collision><origin xyz="%s"/><geometry><sphere radius="%.3f"/></geometry></collision>' % (o, rng.uniform(0.02, 0.05))
    links = ['<link name="base">%s</link>' % coll()]
    joints, bodies = [], ["base"]
    for i in range(n_joints):
        par = bodies[rng.randint(len(bodies))] if branchy else bodies[-1]
        kind = kinds[rng.randint(len(kinds))]
        xyz = "%.3f %.3f %.3f" % tuple(rng.uniform(-0.12, 0.12, 3))
        rpy = "%.3f %.3f %.3f" % tuple(rng.uniform(-0.6, 0.6, 3)) if rotated else "0 0 0"
        ax = rng.randn(3); ax /= np.linalg.norm(ax)
        if not rotated:
            ax = np.eye(3)[rng.randint(3)] * (1 if rng.rand() < 0.5 else -1)
        axs = "%.4f %.4f %.4f" % tuple(ax)
        if kind == "rev":
            links.append('<link name="L%d">%s</link>' % (i, coll()))
            joints.append('<joint name="j%d" type="continuous"><parent link="%s"/><child link="L%d"/><axis xyz="%s"/><origin xyz="%s" rpy="%s"/><limit effort="1" velocity="1"/></joint>' % (i, par, i, axs, xyz, rpy))
            bodies.append("L%d" % i)
        elif kind == "fix":
            links.append('<link name="F%d">%s</link>' % (i, coll()))
            joints.append('<joint name="j%d" type="fixed"><parent link="%s"/><child link="F%d"/><origin xyz="%s" rpy="%s"/></joint>' % (i, par, i, xyz, rpy))
            bodies.append("F%d" % i)
        else:
            links.append('<link name="c%d_R"/><link name="c%d_P"/><link name="c%d_Y">%s</link>' % (i, i, i, coll()))
            joints.append('<joint name="c%d_R" type="revolute"><parent link="%s"/><child link="c%d_R"/><axis xyz="1 0 0"/><origin xyz="%s" rpy="%s"/><limit lower="-1.5" upper="1.5" effort="1" velocity="1"/></joint>'
                          '<joint name="c%d_P" type="revolute"><parent link="c%d_R"/><child link="c%d_P"/><axis xyz="0 1 0"/></joint>'
                          '<joint name="c%d_Y" type="revolute"><parent link="c%d_P"/><child link="c%d_Y"/><axis xyz="0 0 1"/></joint>' % (i, par, i, xyz, rpy, i, i, i, i, i, i))
            bodies.append("c%d_Y" % i)
    return '<?xml version="1.0"?>\n<robot name="r">\n' + "\n".join(links) + "\n" + "\n".join(joints) + "\n</robot>\n"


bad = 0
for r in range(nrob):
    flavour = rng.randint(5)
    kinds = (["rev"], ["cmp"], ["rev", "cmp"], ["rev", "cmp", "fix"], ["rev"])[flavour]
    rotated = flavour >= 3 and rng.rand() < 0.6   # (flavour 4: revolute-only robots with rotated joint origins and general axes)
    n_joints = int(rng.choice([1, 2, 3, 5, 8, 12, 15, 20, 30, 39]))
    branchy = rng.rand() < 0.7
    path = os.path.join(tmp, "r%d.urdf" % r)
    open(path, "w").write(make_urdf(kinds, n_joints, rotated, branchy))
    b = sim.ModelBuilder()
    limit_ke = float(rng.choice([0.0, 40.0]))
    parse_urdf(path, b, xform=sim.transform((0, 0.3, 0), sim.quat_identity()), floating=True, density=1000.0, armature=0.002, stiffness=30.0, damping=0.3,
               shape_ke=1e4, shape_kd=float(rng.choice([0.0, 5.0])), shape_kf=1e2, shape_mu=float(rng.choice([1.0, 0.5])), limit_ke=limit_ke, limit_kd=limit_ke / 40.0)
    tpl = build_template(b, attach_ke=3000.0, attach_kd=30.0)
    nb, nq, nqd = int(tpl["nb"]), int(tpl["nq"]), int(tpl["nqd"])
    has_fixed = 3 in set(int(t) for t in tpl["joint_type"])
    bs, T = int(rng.choice([1, 3, 6, 9])), int(rng.choice([4, 10, 20]))
    q = np.tile(tpl["joint_q"].astype(np.float64), (bs, 1))
    # drop the robot so that its lowest contact candidate of the rest pose is a few mm in the ground
    rc64 = RefC(tpl, np.float64)
    q[:, 7:] = rng.uniform(-0.3, 0.3, (bs, nq - 7))
    yaw = rng.uniform(-0.5, 0.5, bs)
    q[:, 3:7] = np.stack([0 * yaw, np.sin(yaw / 2), 0 * yaw, np.cos(yaw / 2)], -1)
    mass = np.tile(tpl["body_mass"].astype(np.float64), bs)
    inertia = np.tile(tpl["body_inertia"].astype(np.float64), (bs, 1, 1))
    ke = np.tile(np.r_[np.zeros(6), np.full(nqd - 6, 30.0)], bs)
    f2s = sorted(set([0, T // 2, T]))
    F = len(f2s)
    inp = dict(q_init=q.reshape(-1), qd_init=rng.randn(bs * nqd) * 0.1, torques=rng.randn(T, bs * nqd) * 0.05, res_f=rng.randn(T, bs * nb, 6) * 0.05,
               refs=rng.uniform(-0.2, 0.2, (T, bs * nqd)), target_ke=ke, target_kd=ke * 0.01, body_mass=mass, body_inv_mass=1 / mass, body_inertia=inertia,
               body_inv_inertia=np.linalg.inv(inertia), adj_pos=rng.randn(F, bs * nb, 7) * 1e-3, adj_vel=rng.randn(F, bs * nb, 6) * 1e-3)
    inp = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in inp.items()}
    inp.update(frame2step=f2s, nsteps=T, dt=5e-4)
    # lowest point of every env at y = -3 mm: one FK on the oracle, then shift the root
    st0 = rc64.rollout_forward(dict(inp, torques=inp["torques"][:0], res_f=inp["res_f"][:0], refs=inp["refs"][:0]), 0, [0], 5e-4)
    pos = st0["wp_pos"].reshape(bs, nb, 7)
    cb, cp, cd = tpl["contact_body"], tpl["contact_point"], tpl["contact_dist"]
    low = np.full(bs, 1e9)
    for e in range(bs):
        for c in range(len(cb)):
            p, qq = pos[e, cb[c], :3], pos[e, cb[c], 3:]
            v = cp[c]; u = qq[:3]
            rot = v * (2 * qq[3] ** 2 - 1) + 2 * qq[3] * np.cross(u, v) + 2 * u * np.dot(u, v)
            low[e] = min(low[e], p[1] + rot[1] - cd[c])
    qi = inp["q_init"].reshape(bs, nq).copy(); qi[:, 1] -= (low + 0.003).astype(np.float32); inp["q_init"] = qi.reshape(-1)
    st0 = rc64.rollout_forward(dict(inp, torques=inp["torques"][:0], res_f=inp["res_f"][:0], refs=inp["refs"][:0]), 0, [0], 5e-4)   # FK of the dropped robot
    why = []
    tag = "nb=%-2d joints=%s rot=%d branchy=%d bs=%d T=%-2d" % (nb, "+".join(kinds), rotated, branchy, bs, T)
    try:
        dm = hip_backend.DeviceModel(tpl)
        fams = [1, 2] if (kinds == ["rev"] and nb <= 16) else [0]   # revolute-only, <= 16 bodies: quad-lane eligible (rotated joint origins and general axes included)
        rc32 = RefC(tpl, np.float32)
        try:   # FIXED joints: the scale-invariant evaluation the kernels use (oracle switch; changes nothing in float64 for the others)
            rc32.set_twist_eval(has_fixed); rc64.set_twist_eval(has_fixed)
            st32 = rc32.rollout_forward(inp, T, f2s, 5e-4)
            st64 = rc64.rollout_forward(inp, T, f2s, 5e-4)
        finally:
            rc32.set_twist_eval(False); rc64.set_twist_eval(False)
        t = {k: torch.from_numpy(inp[k]).to(dev) for k in FWD + ("adj_pos", "adj_vel")}
        worst_all = 0.0
        for fam in fams:
            dm.set_kernel_family(fam)
            o = dm.rollout_forward(bs, T, 5e-4, *[t[k] for k in FWD], frame2step=f2s)
            g = dm.rollout_backward(bs, T, 5e-4, *[t[k] for k in BWD], f2s, o[4], t["adj_pos"], t["adj_vel"])
            o2 = dm.rollout_forward(bs, T, 5e-4, *[t[k] for k in FWD], frame2step=f2s)
            g2 = dm.rollout_backward(bs, T, 5e-4, *[t[k] for k in BWD], f2s, o2[4], t["adj_pos"], t["adj_vel"])
            if not (all(torch.equal(a, b_) for a, b_ in zip(o[:4], o2[:4])) and all(torch.equal(g[k], g2[k]) for k in g)): why.append("fam%d not repeatable" % fam)
            out = dict(wp_pos=o[0].cpu().numpy(), wp_vel=o[1].cpu().numpy(), grf=o[2].cpu().numpy(), jaf=o[3].cpu().numpy())
            for k, floor in (("wp_pos", 2e-5), ("wp_vel", 5e-3), ("grf", 1e-2), ("jaf", 2e-2)):
                e_, y_ = relmax(out[k], st64[k]), relmax(st32[k], st64[k])
                if not e_ < max(floor, 4.0 * y_): why.append("fam%d %s %.1e (fp32 oracle %.1e)" % (fam, k, e_, y_))
            if not all(bool(torch.isfinite(v).all()) for v in g.values()): why.append("fam%d non-finite gradient" % fam)
            own = own_trajectory_check(dm, tpl, inp, dev, hitlog_check=False, abs_floor=1e-7)
            w = own["worst"]
            lim = np.maximum(1e-3, 4.0 * own["fp32_atan2"])
            okg = bool((w <= lim).all())
            if not okg: why.append("fam%d own-trajectory %s (fp32 %s)" % (fam, np.array2string(w, precision=1), np.array2string(own["fp32_atan2"], precision=1)))
            worst_all = max(worst_all, float(np.median(w)))
            # zero steps
            z = dm.rollout_forward(bs, 0, 5e-4, t["q_init"], t["qd_init"], t["torques"][:0], t["res_f"][:0], t["refs"][:0], *[t[k] for k in FWD[5:]], frame2step=[0])
            gz = dm.rollout_backward(bs, 0, 5e-4, t["q_init"], t["qd_init"], t["torques"][:0], t["refs"][:0], *[t[k] for k in BWD[4:]], [0], z[4], t["adj_pos"][:1], t["adj_vel"][:1])
            if relmax(z[0].cpu().numpy(), st0["wp_pos"]) > 2e-6: why.append("fam%d zero-step FK" % fam)
            if not bool(torch.isfinite(gz["q_init"]).all()): why.append("fam%d zero-step gradient" % fam)
        torch.cuda.synchronize()
    except Exception as e:
        if "more than 8 children per body" in repr(e):   # a stated limit of the library: refused loudly at model creation
            print("ok  %s  REFUSED: %s" % (tag, e), flush=True)
            continue
        why.append("EXC %r" % (e,))
    bad += bool(why)
    print("%s %s segw=%s contacts %.1f  own-traj median %.1e  %s" % ("BAD" if why else "ok ", tag, dm.segment_width() if "dm" in dir() else "?", np.abs(st64["grf"]).max() if "st64" in dir() else -1, worst_all if "worst_all" in dir() else -1, "; ".join(why)), flush=True)
print("robot stress: %d robots, %d failures" % (nrob, bad))
sys.exit(1 if bad else 0)
