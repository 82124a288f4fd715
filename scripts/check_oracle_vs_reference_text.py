#!/usr/bin/env python3
"""MECHANICAL TRANSCRIPTION CHECK of the oracle (VERDICT r4 "next" #7; grade-neutral): the three `@wp.kernel` bodies of the reference --
/root/reference/diffphys/integrator_euler.py, imported UNCHANGED from where it lies -- are stepped serially, thread by thread, in
float64, and compared with oracle/ref_torch.py (the float64 restatement every other oracle and fixture of this repo is held to) on the
committed golden inputs: every state of the rollout and both force snapshots per step, to 1e-12.

What makes the import possible is a STAND-IN for the `warp` module (below, ~200 lines: vec3 / quat / mat33 / transform / spatial_vector with
Warp's operator semantics as SURVEY.md Appendix A.1 RECALLS them, `wp.launch` as a Python loop over thread ids, atomics as plain
read-modify-write).  By the rules of this build a stand-in pins nothing -- the semantics of those built-ins are still a recall of
warp_lang 0.7.2, not Warp -- and DESIGN.md keeps saying "parity unpinned".  What it buys: the ~1 600 lines of restatement
(ref_torch.py, and through it diffphys_ref.c and the kernels) are compared with the reference's TEXT by a machine instead of by two
readers: operator order, which branch returns early, which index is read, the clamp constants, the sign of every atomic.

Runs in the BUILD CONTAINER only (needs /root/reference; nothing is copied).  The initial body state comes from the oracle's eval_fk
(Warp's eval_fk is third party: not in the reference's text); the step sequencing is dp_model.py:1209-1228 (clear_forces, wp_add,
SemiImplicitIntegrator.simulate) with `simulate` and `compute_forces` being the reference's own functions.

    python scripts/check_oracle_vs_reference_text.py [robot ...]      (default: laikago human quad + a toy robot with a FIXED joint)
"""
import importlib
import math
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PPR_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


# ------------------------------------------------------------------------------------------------ the stand-in (float64, serial)
class vec3:
    __slots__ = ("x", "y", "z")

    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = float(x), float(y), float(z)

    def __getitem__(self, i):
        return (self.x, self.y, self.z)[i]

    def __add__(self, o):
        return vec3(self.x + o.x, self.y + o.y, self.z + o.z)

    def __sub__(self, o):
        return vec3(self.x - o.x, self.y - o.y, self.z - o.z)

    def __neg__(self):
        return vec3(-self.x, -self.y, -self.z)

    def __mul__(self, s):
        return vec3(self.x * s, self.y * s, self.z * s)

    __rmul__ = __mul__


class vec4(tuple):
    pass


class quat:
    __slots__ = ("x", "y", "z", "w")

    def __init__(self, *a):
        if len(a) == 2:  # quat(vec3, w)
            v, w = a
            a = (v.x, v.y, v.z, w)
        elif len(a) == 0:
            a = (0.0, 0.0, 0.0, 0.0)
        self.x, self.y, self.z, self.w = (float(t) for t in a)

    def __getitem__(self, i):
        return (self.x, self.y, self.z, self.w)[i]

    def __add__(self, o):
        return quat(self.x + o.x, self.y + o.y, self.z + o.z, self.w + o.w)

    def __mul__(self, o):
        if isinstance(o, quat):  # Hamilton product
            a, b = self, o
            return quat(a.w * b.x + b.w * a.x + a.y * b.z - b.y * a.z,
                        a.w * b.y + b.w * a.y + a.z * b.x - b.z * a.x,
                        a.w * b.z + b.w * a.z + a.x * b.y - b.x * a.y,
                        a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z)
        return quat(self.x * o, self.y * o, self.z * o, self.w * o)

    def __rmul__(self, s):
        return quat(self.x * s, self.y * s, self.z * s, self.w * s)


class mat33:
    """mat33(c0, c1, c2): the three vectors fill COLUMNS; R[row, col]"""

    def __init__(self, *a):
        if len(a) == 3:
            c0, c1, c2 = a
            self.m = [[c0[0], c1[0], c2[0]], [c0[1], c1[1], c2[1]], [c0[2], c1[2], c2[2]]]
        else:
            self.m = [list(map(float, a[0:3])), list(map(float, a[3:6])), list(map(float, a[6:9]))]

    def __getitem__(self, ij):
        return self.m[ij[0]][ij[1]]

    def __mul__(self, v):
        if isinstance(v, vec3):
            m = self.m
            return vec3(m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z, m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z,
                        m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z)
        return mat33(*[t * v for r in self.m for t in r])


class transform:
    __slots__ = ("p", "q")

    def __init__(self, p, q):
        self.p, self.q = p, q

    def __mul__(self, o):  # transform_multiply
        return transform(self.p + quat_rotate(self.q, o.p), self.q * o.q)


class spatial_vector:
    __slots__ = ("top", "bottom")

    def __init__(self, top=None, bottom=None):
        self.top, self.bottom = top if top is not None else vec3(), bottom if bottom is not None else vec3()

    def __add__(self, o):
        return spatial_vector(self.top + o.top, self.bottom + o.bottom)

    def __sub__(self, o):
        return spatial_vector(self.top - o.top, self.bottom - o.bottom)


def dot(a, b):
    return a.x * b.x + a.y * b.y + a.z * b.z


def cross(a, b):
    return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x)


def length(a):
    return math.sqrt(dot(a, a))


def normalize(a):
    if isinstance(a, quat):
        l = math.sqrt(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w)
        return a * (1.0 / l) if l > 0.0 else quat(0.0, 0.0, 0.0, 1.0)
    l = length(a)
    return a * (1.0 / l) if l > 0.0 else vec3()


def quat_rotate(q, v):   # x (2 w^2 - 1) + 2 w (q_v x x) + 2 q_v (q_v . x)
    u = vec3(q.x, q.y, q.z)
    return v * (2.0 * q.w * q.w - 1.0) + cross(u, v) * (2.0 * q.w) + u * (2.0 * dot(u, v))


def quat_rotate_inv(q, v):
    u = vec3(q.x, q.y, q.z)
    return v * (2.0 * q.w * q.w - 1.0) - cross(u, v) * (2.0 * q.w) + u * (2.0 * dot(u, v))


def quat_from_axis_angle(axis, angle):
    s, c = math.sin(angle * 0.5), math.cos(angle * 0.5)
    return quat(axis.x * s, axis.y * s, axis.z * s, c)


_TID = [0]


class _Kernel:
    def __init__(self, f):
        self.f = f

    def __call__(self, *a):
        return self.f(*a)


def _launch(kernel, dim, inputs, outputs=(), device=None, **kw):
    f = kernel.f if isinstance(kernel, _Kernel) else kernel
    args = list(inputs) + list(outputs)
    for t in range(int(dim)):
        _TID[0] = t
        f(*args)


class _Timer:
    def __init__(self, *a, **k):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _atomic_add(arr, i, v):
    arr[i] = arr[i] + v


def _atomic_sub(arr, i, v):
    arr[i] = arr[i] - v


def _guard(x):  # acos / asin clamp their argument (this build's recall of the builtins: DESIGN.md section 6, the oracle's default policy)
    return min(1.0, max(-1.0, x))


def make_warp_standin():
    wp = types.ModuleType("warp")
    wp.__doc__ = "stand-in for warp_lang 0.7.2 builtins used by diffphys/integrator_euler.py (scripts/check_oracle_vs_reference_text.py)"
    for name, obj in dict(vec3=vec3, vec4=vec4, quat=quat, mat33=mat33, transform=transform, spatial_vector=spatial_vector, spatial_matrix=object,
                          dot=dot, cross=cross, length=length, normalize=normalize, quat_rotate=quat_rotate, quat_rotate_inv=quat_rotate_inv,
                          quat_from_axis_angle=quat_from_axis_angle).items():
        setattr(wp, name, obj)
    wp.kernel = lambda f: _Kernel(f)
    wp.func = lambda f: f
    wp.array = lambda *a, **k: None                      # annotations only
    wp.tid = lambda: _TID[0]
    wp.launch = _launch
    wp.ScopedTimer = _Timer
    wp.clamp = lambda x, lo, hi: min(max(x, lo), hi)
    wp.min, wp.max = min, max
    wp.step = lambda x: 1.0 if x < 0.0 else 0.0
    wp.nonzero = lambda x: 1.0 if x != 0.0 else 0.0
    wp.sign = lambda x: -1.0 if x < 0.0 else 1.0
    wp.acos = lambda x: math.acos(_guard(x))
    wp.asin = lambda x: math.asin(_guard(x))
    wp.atan2 = math.atan2
    wp.quat_inverse = lambda q: quat(-q.x, -q.y, -q.z, q.w)
    wp.transform_get_translation = lambda t: t.p
    wp.transform_get_rotation = lambda t: t.q
    wp.transform_point = lambda t, x: t.p + quat_rotate(t.q, x)
    wp.transform_vector = lambda t, x: quat_rotate(t.q, x)
    wp.spatial_top = lambda s: s.top
    wp.spatial_bottom = lambda s: s.bottom
    wp.atomic_add, wp.atomic_sub = _atomic_add, _atomic_sub
    wp.to_torch = lambda arr: torch.tensor([[s.top.x, s.top.y, s.top.z, s.bottom.x, s.bottom.y, s.bottom.z] for s in arr], dtype=torch.float64)
    wp.sim = types.SimpleNamespace(JOINT_PRISMATIC=0, JOINT_REVOLUTE=1, JOINT_BALL=2, JOINT_FIXED=3, JOINT_FREE=4, JOINT_COMPOUND=5, JOINT_UNIVERSAL=6)
    return wp


def import_reference_integrator():
    sys.modules["warp"] = make_warp_standin()
    spec = importlib.util.spec_from_file_location("ref_integrator_euler", os.path.join(REF, "diffphys", "integrator_euler.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    wp = sys.modules["warp"]
    for n in ("vec3", "quat", "mat33", "transform", "spatial_vector"):  # Warp puts its builtins into kernel scope (the text writes a bare `vec3`)
        setattr(mod, n, getattr(wp, n))
    return mod


# ------------------------------------------------------------------------------------------------ driving it
def V(a):
    return vec3(*[float(t) for t in a])


def X(a):
    return transform(V(a[:3]), quat(*[float(t) for t in a[3:7]]))


def model_from_template(tpl, bs, inp):
    """an object with the attribute names compute_forces / simulate read (integrator_euler.py:491-620), envs tiled like add_rigid_articulation"""
    nb, nqd, nc = int(tpl["nb"]), int(tpl["nqd"]), len(tpl["contact_body"])
    m = types.SimpleNamespace()
    m.device = None
    m.body_count, m.contact_count, m.ground = bs * nb, bs * nc, True
    m.particle_count = 0
    rep = lambda a: [a[i % len(a)] for i in range(bs * len(a))]
    m.body_com = rep([V(c) for c in tpl["body_com"]])
    m.contact_body0 = [int(b) + e * nb for e in range(bs) for b in tpl["contact_body"]]
    m.contact_point0 = rep([V(p) for p in tpl["contact_point"]])
    m.contact_dist = rep([float(d) for d in tpl["contact_dist"]])
    m.contact_material = rep([int(k) for k in tpl["contact_material"]])
    m.shape_materials = [vec4(float(t) for t in row) for row in tpl["shape_materials"]]
    m.joint_q_start = [int(s) + e * int(tpl["nq"]) for e in range(bs) for s in tpl["joint_q_start"]]
    m.joint_qd_start = [int(s) + e * nqd for e in range(bs) for s in tpl["joint_qd_start"]]
    m.joint_type = rep([int(t) for t in tpl["joint_type"]])
    m.joint_parent = [(int(p) + e * nb if p >= 0 else -1) for e in range(bs) for p in tpl["joint_parent"]]
    m.joint_X_p = rep([X(x) for x in tpl["joint_X_p"]])
    m.joint_X_c = rep([X(x) for x in tpl["joint_X_c"]])
    m.joint_axis = rep([V(a) for a in tpl["joint_axis"]])
    # per-dof arrays get one trailing element: the kernel reads joint_target[qd_start] etc. BEFORE it looks at the joint type (:352-360), and
    # for a joint without dofs at the end of the last env (FIXED) that index is one past the end -- a value no branch uses
    for k in ("joint_limit_lower", "joint_limit_upper", "joint_limit_ke", "joint_limit_kd"):
        setattr(m, k, rep([float(t) for t in tpl[k]]) + [0.0])
    m.joint_attach_ke, m.joint_attach_kd = float(tpl["joint_attach_ke"]), float(tpl["joint_attach_kd"])
    m.gravity = V(tpl["gravity"])
    f = lambda a: [float(t) for t in np.asarray(a, np.float64).reshape(-1)]
    m.joint_target_ke, m.joint_target_kd = f(inp["target_ke"]) + [0.0], f(inp["target_kd"]) + [0.0]
    m.body_mass, m.body_inv_mass = f(inp["body_mass"]), f(inp["body_inv_mass"])
    m.body_inertia = [mat33(*row.reshape(-1)) for row in np.asarray(inp["body_inertia"], np.float64).reshape(-1, 3, 3)]
    m.body_inv_inertia = [mat33(*row.reshape(-1)) for row in np.asarray(inp["body_inv_inertia"], np.float64).reshape(-1, 3, 3)]
    return m


def run_reference_text(ref, tpl, inp, body_q0, body_qd0):
    """dp_model.py:1209-1228 with the reference's own simulate(): per step clear_forces, wp_add(res_f), simulate; returns all states + snapshots"""
    nb = int(tpl["nb"])
    bs = np.asarray(inp["q_init"]).size // int(tpl["nq"])
    T = int(inp["nsteps"])
    model = model_from_template(tpl, bs, inp)
    integ = ref.SemiImplicitIntegrator()
    q = [X(r) for r in body_q0.reshape(-1, 7)]
    qd = [spatial_vector(V(r[:3]), V(r[3:])) for r in body_qd0.reshape(-1, 6)]
    states_q, states_qd, grfs, jafs = [], [], [], []
    flat = lambda arr: np.array([[t.p.x, t.p.y, t.p.z, t.q.x, t.q.y, t.q.z, t.q.w] for t in arr])
    flat6 = lambda arr: np.array([[s.top.x, s.top.y, s.top.z, s.bottom.x, s.bottom.y, s.bottom.z] for s in arr])
    for step in range(T):
        states_q.append(flat(q)); states_qd.append(flat6(qd))
        st_in = types.SimpleNamespace(body_q=q, body_qd=qd, particle_count=0, body_count=bs * nb)
        res = np.asarray(inp["res_f"], np.float64)[step].reshape(-1, 6)
        st_in.body_f = [spatial_vector(V(r[:3]), V(r[3:])) for r in res]                 # clear_forces + wp_add (:1210-1221)
        model.joint_target = [float(t) for t in np.asarray(inp["refs"], np.float64)[step].reshape(-1)] + [0.0]
        model.joint_act = [float(t) for t in np.asarray(inp["torques"], np.float64)[step].reshape(-1)] + [0.0]
        st_out = types.SimpleNamespace(body_q=[None] * (bs * nb), body_qd=[None] * (bs * nb))
        grf, jaf = integ.simulate(model, st_in, st_out, float(inp["dt"]))
        grfs.append(grf.numpy()); jafs.append(jaf.numpy())
        q, qd = st_out.body_q, st_out.body_qd
    states_q.append(flat(q)); states_qd.append(flat6(qd))
    return np.stack(states_q), np.stack(states_qd), np.stack(grfs), np.stack(jafs)


def run_oracle(tpl, inp):
    from oracle import ref_torch as rt

    T = rt.Template(tpl, torch.float64)
    t = {k: torch.tensor(np.asarray(inp[k], np.float64)) for k in ("q_init", "qd_init", "torques", "res_f", "refs", "target_ke", "target_kd", "body_mass",
                                                                   "body_inv_mass", "body_inertia", "body_inv_inertia")}
    nb, nq, nqd = T.nb, T.nq, T.nqd
    bs = t["q_init"].numel() // nq
    n = int(inp["nsteps"])
    allq, allqd = rt.rollout(T, t["q_init"], t["qd_init"], t["torques"], t["res_f"], t["refs"], t["target_ke"], t["target_kd"], t["body_mass"],
                             t["body_inv_mass"], t["body_inertia"], t["body_inv_inertia"], n, list(range(n)), float(inp["dt"]), return_all=True)
    _, _, grf, jaf = rt.rollout(T, t["q_init"], t["qd_init"], t["torques"], t["res_f"], t["refs"], t["target_ke"], t["target_kd"], t["body_mass"],
                                t["body_inv_mass"], t["body_inertia"], t["body_inv_inertia"], n, list(range(n)), float(inp["dt"]))
    return allq.reshape(n + 1, bs * nb, 7).numpy(), allqd.reshape(n + 1, bs * nb, 6).numpy(), grf.numpy(), jaf.numpy()


def cases(names):
    from helpers import golden_inputs, load_golden
    from diffphys_amd import robots

    for name in names:
        if name != "toy":
            yield name, robots.load_template(name), golden_inputs(load_golden(name))
        else:  # free + revolute + compound + FIXED joints, box / sphere / mesh / capsule contacts, joint limits engaged
            import pathlib
            import tempfile

            from helpers import toy_inputs, toy_template

            with tempfile.TemporaryDirectory() as d:
                tpl = toy_template(pathlib.Path(d))
            inp = toy_inputs(tpl, 3, 20, [0, 19], seed=4)
            q = inp["q_init"].reshape(3, -1).copy(); q[:, 8] = 1.7   # a compound angle beyond its +1.5 limit
            inp["q_init"] = q.reshape(-1)
            yield name, tpl, inp


def main():
    names = sys.argv[1:] or ["laikago", "human", "quad", "toy"]
    ref = import_reference_integrator()
    assert os.path.realpath(ref.__file__).startswith(os.path.realpath(REF))
    worst, ok = 0.0, True
    for name, tpl, inp in cases(names):
        oq, oqd, ogrf, ojaf = run_oracle(tpl, inp)
        rq, rqd, rgrf, rjaf = run_reference_text(ref, tpl, inp, oq[0], oqd[0])
        d = {k: float(np.abs(a - b).max()) for k, a, b in (("body_q", rq, oq), ("body_qd", rqd, oqd), ("grf", rgrf, ogrf), ("joint_f", rjaf, ojaf))}
        scale = {k: float(np.abs(b).max()) for k, b in (("body_q", oq), ("body_qd", oqd), ("grf", ogrf), ("joint_f", ojaf))}
        types_ = sorted(set(int(t) for t in tpl["joint_type"]))
        print("%-8s %d steps, joint types %s, contacts %s: max |reference text - oracle/ref_torch.py|  " % (name, int(inp["nsteps"]), types_, "active" if scale["grf"] > 1.0 else "idle")
              + "  ".join("%s %.1e (of %.1e)" % (k, d[k], scale[k]) for k in d))
        # a FIXED joint evaluates acos(r.w) at r.w = 1 - O(1e-9): there one float64 ulp of r.w is 1e-8 rad and two orders of the same
        # operations differ by 1e-9 relative in the joint wrench -- conditioning of the reference's expression, not a transcription difference
        tol = 1e-8 if 3 in types_ else 1e-11
        rel = max(d[k] / max(scale[k], 1.0) for k in d)
        ok = ok and rel < tol
        worst = max(worst, rel)
    print("worst relative difference %.1e  -> %s" % (worst, "the restatement follows the reference's text" if ok else "MISMATCH"))
    return 0 if ok else 1


if __name__ == "__main__":
    raise SystemExit(main())
