"""ORACLE (test infrastructure): ctypes binding of oracle/ref_c/diffphys_ref.c.

``RefC(tpl, dtype)`` wraps one compiled precision (np.float32 / np.float64) and
exposes rollout forward / backward and FK forward / backward on numpy arrays in
the reference's flat env-major layouts.  See the C file's header for what it
restates and why parity is unpinned.
"""
import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_c")


def build(force=False):
    libs = [os.path.join(_DIR, "libdiffphys_ref_f32.so"), os.path.join(_DIR, "libdiffphys_ref_f64.so")]
    src = os.path.join(_DIR, "diffphys_ref.c")
    stale = force or any((not os.path.exists(l)) or os.path.getmtime(l) < os.path.getmtime(src) for l in libs)
    if stale:
        subprocess.check_call(["make", "-C", _DIR, "-s", "-B"] if force else ["make", "-C", _DIR, "-s"])
    return libs


class RefC:
    def __init__(self, tpl, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        name = "libdiffphys_ref_f32.so" if self.dtype == np.float32 else "libdiffphys_ref_f64.so"
        path = os.path.join(_DIR, name)
        if not os.path.exists(path):
            build()
        self.lib = ctypes.CDLL(path)
        self.real = ctypes.c_float if self.dtype == np.float32 else ctypes.c_double
        assert self.lib.ref_sizeof_real() == self.dtype.itemsize
        self.lib.ref_template_create.restype = ctypes.c_void_p
        self.nb, self.nq, self.nqd = int(tpl["nb"]), int(tpl["nq"]), int(tpl["nqd"])
        f = lambda k: np.ascontiguousarray(tpl[k], dtype=self.dtype)
        i = lambda k: np.ascontiguousarray(tpl[k], dtype=np.int32)
        self._keep = [
            i("joint_type"), i("joint_parent"), i("joint_q_start"), i("joint_qd_start"), f("joint_X_p"), f("joint_X_c"),
            f("joint_axis"), f("body_com"), f("joint_limit_lower"), f("joint_limit_upper"), f("joint_limit_ke"),
            f("joint_limit_kd"), i("contact_body"), f("contact_point"), f("contact_dist"), i("contact_material"),
            f("shape_materials"), f("gravity"),
        ]
        nc, nmat = len(tpl["contact_body"]), len(tpl["shape_materials"])
        args = [ctypes.c_int(v) for v in (self.nb, self.nq, self.nqd, nc, nmat)]
        args += [a.ctypes.data_as(ctypes.c_void_p) for a in self._keep]
        args += [self.real(float(tpl["joint_attach_ke"])), self.real(float(tpl["joint_attach_kd"]))]
        self.h = ctypes.c_void_p(self.lib.ref_template_create(*args))

    def __del__(self):
        try:
            self.lib.ref_template_destroy(self.h)
        except Exception:
            pass

    def set_acos_policy(self, unguarded):
        """False: clamped acos/asin with the guarded adjoint (default, what the kernels do); True: unguarded, as SURVEY.md
        Appendix A.1 recalls Warp.  Process-wide switch of this precision's library: reset it after use."""
        self.lib.ref_set_acos_policy(1 if unguarded else 0)

    def set_twist_eval(self, use_atan2):
        """True: the revolute twist angle through atan2 (the same function as the literal 2 acos(twist.w) sign(..), see
        ref_set_twist_eval in diffphys_ref.c).  Process-wide switch of this precision's library: reset it after use."""
        self.lib.ref_set_twist_eval(1 if use_atan2 else 0)

    def set_state_rounding(self, on):
        """1 / True: rollout_forward rounds every stored state to fp32; 2: to an adjacent fp32 number (ref_set_state_rounding in diffphys_ref.c).  Process-wide switch
        of this precision's library: reset it after use."""
        self.lib.ref_set_state_rounding(int(on))

    def num_threads(self):
        return int(self.lib.ref_num_threads())

    def _p(self, a):
        return a.ctypes.data_as(ctypes.c_void_p)

    def _c(self, a):
        return np.ascontiguousarray(a, dtype=self.dtype)

    def rollout_forward(self, inp, nsteps, frame2step, dt):
        nb = self.nb
        bs = inp["q_init"].size // self.nq
        F = len(frame2step)
        a = {k: self._c(inp[k]) for k in ("q_init", "qd_init", "torques", "res_f", "refs", "target_ke", "target_kd",
                                          "body_inv_mass", "body_inertia", "body_inv_inertia")}
        f2s = np.ascontiguousarray(frame2step, dtype=np.int32)
        z = lambda *s: np.zeros(s, dtype=self.dtype)
        st = dict(states_q=z(nsteps + 1, bs * nb, 7), states_qd=z(nsteps + 1, bs * nb, 6), states_f=z(nsteps, bs * nb, 6),
                  wp_pos=z(F, bs * nb, 7), wp_vel=z(F, bs * nb, 6), grf=z(F, bs * nb, 6), jaf=z(F, bs * nb, 6))
        self.lib.ref_rollout_forward(
            self.h, ctypes.c_int(bs), ctypes.c_int(nsteps), self.real(dt), self._p(a["q_init"]), self._p(a["qd_init"]),
            self._p(a["torques"]), self._p(a["res_f"]), self._p(a["refs"]), self._p(a["target_ke"]), self._p(a["target_kd"]),
            self._p(a["body_inv_mass"]), self._p(a["body_inertia"]), self._p(a["body_inv_inertia"]), ctypes.c_int(F),
            self._p(f2s), self._p(st["states_q"]), self._p(st["states_qd"]), self._p(st["states_f"]), self._p(st["wp_pos"]),
            self._p(st["wp_vel"]), self._p(st["grf"]), self._p(st["jaf"]))
        st["_inputs"] = a
        st["_f2s"] = f2s
        st["_bs"], st["_nsteps"], st["_dt"] = bs, nsteps, dt
        return st

    def rollout_backward(self, st, adj_pos, adj_vel):
        a, bs, nsteps = st["_inputs"], st["_bs"], st["_nsteps"]
        nb, nq, nqd = self.nb, self.nq, self.nqd
        z = lambda *s: np.zeros(s, dtype=self.dtype)
        g = dict(q_init=z(bs * nq), qd_init=z(bs * nqd), torques=z(nsteps, bs * nqd), res_f=z(nsteps, bs * nb, 6),
                 refs=z(nsteps, bs * nqd), target_ke=z(bs * nqd), target_kd=z(bs * nqd), body_mass=z(bs * nb),
                 body_inv_mass=z(bs * nb), body_inertia=z(bs * nb, 3, 3), body_inv_inertia=z(bs * nb, 3, 3))
        ap, av = self._c(adj_pos), self._c(adj_vel)
        self.lib.ref_rollout_backward(
            self.h, ctypes.c_int(bs), ctypes.c_int(nsteps), self.real(st["_dt"]), self._p(a["q_init"]), self._p(a["qd_init"]),
            self._p(a["torques"]), self._p(a["refs"]), self._p(a["target_ke"]), self._p(a["target_kd"]),
            self._p(a["body_inv_mass"]), self._p(a["body_inertia"]), self._p(a["body_inv_inertia"]),
            ctypes.c_int(len(st["_f2s"])), self._p(st["_f2s"]), self._p(st["states_q"]), self._p(st["states_qd"]),
            self._p(st["states_f"]), self._p(ap), self._p(av), self._p(g["q_init"]), self._p(g["qd_init"]),
            self._p(g["torques"]), self._p(g["res_f"]), self._p(g["refs"]), self._p(g["target_ke"]), self._p(g["target_kd"]),
            self._p(g["body_inv_mass"]), self._p(g["body_inertia"]), self._p(g["body_inv_inertia"]))
        return g

    def trajectory_state(self, traj, inp):
        """Wraps a trajectory ANOTHER implementation saved (dict with states_q [T, bs*nb, 7], states_qd [T, bs*nb, 6], states_f
        [T, bs*nb, 6]: the HIP kernels' DeviceModel.saved_trajectory) into what rollout_backward* / singularity_probe take: the
        stored values cast to this precision, the inputs of `inp`.  The reverse sweep never reads state T."""
        a = {k: self._c(inp[k]) for k in ("q_init", "qd_init", "torques", "res_f", "refs", "target_ke", "target_kd",
                                          "body_inv_mass", "body_inertia", "body_inv_inertia")}
        nsteps = int(np.asarray(traj["states_f"]).shape[0])
        st = dict(states_q=self._c(traj["states_q"]), states_qd=self._c(traj["states_qd"]), states_f=self._c(traj["states_f"]))
        assert st["states_q"].shape[0] >= nsteps
        st.update(_inputs=a, _f2s=np.ascontiguousarray(inp["frame2step"], dtype=np.int32), _bs=a["q_init"].size // self.nq,
                  _nsteps=nsteps, _dt=inp["dt"])
        return st

    def rollout_backward_forced(self, st, adj_pos, adj_vel, clamp_mask=None, pinned_touch=False, touch_list=None):
        """rollout_backward with the discrete decisions of the implementation that saved `st` (see ref_rollout_backward_forced in
        diffphys_ref.c): clamp_mask int [T, bs*nb] = its stored velocity-clamp masks, pinned_touch = contacts touch where its pinned
        fp32 height test says so; touch_list int [T, bs, cap] (what touch_fp32 returned for the UNPERTURBED states) names the touching
        candidates outright.  None given: identical to rollout_backward."""
        a, bs, nsteps = st["_inputs"], st["_bs"], st["_nsteps"]
        nb, nq, nqd = self.nb, self.nq, self.nqd
        z = lambda *s: np.zeros(s, dtype=self.dtype)
        g = dict(q_init=z(bs * nq), qd_init=z(bs * nqd), torques=z(nsteps, bs * nqd), res_f=z(nsteps, bs * nb, 6),
                 refs=z(nsteps, bs * nqd), target_ke=z(bs * nqd), target_kd=z(bs * nqd), body_mass=z(bs * nb),
                 body_inv_mass=z(bs * nb), body_inertia=z(bs * nb, 3, 3), body_inv_inertia=z(bs * nb, 3, 3))
        ap, av = self._c(adj_pos), self._c(adj_vel)
        tl = None if touch_list is None else np.ascontiguousarray(touch_list, dtype=np.int32)
        mk = None if clamp_mask is None else np.ascontiguousarray(np.asarray(clamp_mask).reshape(nsteps, bs * nb) & 63, dtype=np.int32)
        self.lib.ref_rollout_backward_forced(
            self.h, ctypes.c_int(bs), ctypes.c_int(nsteps), self.real(st["_dt"]), self._p(a["q_init"]), self._p(a["qd_init"]),
            self._p(a["torques"]), self._p(a["refs"]), self._p(a["target_ke"]), self._p(a["target_kd"]),
            self._p(a["body_inv_mass"]), self._p(a["body_inertia"]), self._p(a["body_inv_inertia"]),
            ctypes.c_int(len(st["_f2s"])), self._p(st["_f2s"]), self._p(st["states_q"]), self._p(st["states_qd"]),
            self._p(st["states_f"]), self._p(ap), self._p(av), self._p(g["q_init"]), self._p(g["qd_init"]),
            self._p(g["torques"]), self._p(g["res_f"]), self._p(g["refs"]), self._p(g["target_ke"]), self._p(g["target_kd"]),
            self._p(g["body_inv_mass"]), self._p(g["body_inertia"]), self._p(g["body_inv_inertia"]),
            self._p(mk) if mk is not None else ctypes.c_void_p(0), ctypes.c_int(1 if pinned_touch else 0),
            self._p(tl) if tl is not None else ctypes.c_void_p(0), ctypes.c_int(tl.shape[-1] if tl is not None else 0))
        return g

    def touch_fp32(self, st, cap=32):
        """int32 [T, bs, cap]: per env-step the count and the template indices (ascending) of the contact candidates that touch by
        the kernels' pinned fp32 height test on the stored state (ref_touch_fp32)."""
        bs, nsteps = st["_bs"], st["_nsteps"]
        out = np.zeros((nsteps, bs, cap), dtype=np.int32)
        self.lib.ref_touch_fp32(self.h, ctypes.c_int(bs), ctypes.c_int(nsteps), self._p(st["states_q"]), self._p(out), ctypes.c_int(cap))
        return out

    def singularity_probe(self, st):
        """[nsteps, bs, 5] for a trajectory returned by rollout_forward: min |contact height|, min distance of a pre-clamp
        velocity component from +-10, min |a - b| of the Coulomb switch over touching points, min 1 - |twist.w| over revolute
        joints, min distance of a contact force component from the +-500 N clamp (see ref_singularity_probe in diffphys_ref.c)."""
        a, bs, nsteps = st["_inputs"], st["_bs"], st["_nsteps"]
        out = np.zeros((nsteps, bs, 5), dtype=self.dtype)
        self.lib.ref_singularity_probe(self.h, ctypes.c_int(bs), ctypes.c_int(nsteps), self.real(st["_dt"]), self._p(st["states_q"]),
                                       self._p(st["states_qd"]), self._p(st["states_f"]), self._p(a["body_inv_mass"]),
                                       self._p(a["body_inertia"]), self._p(a["body_inv_inertia"]), self._p(out))
        return out

    def branch_log(self, st, inp=None):
        """Discrete decisions per (step, env, body) of a trajectory: touching contact candidates, how many of them slide on the
        kf |vt| friction branch, velocity clamp mask (see ref_branch_log in diffphys_ref.c).  st: what rollout_forward returned,
        or any dict with states_q / states_qd / states_f (e.g. a trajectory the GPU kernel saved) plus `inp` for the inputs."""
        a = st["_inputs"] if "_inputs" in st else {k: self._c(inp[k]) for k in ("body_inv_mass", "body_inertia", "body_inv_inertia")}
        sq, sd, sf = self._c(st["states_q"]), self._c(st["states_qd"]), self._c(st["states_f"])
        nsteps = sf.shape[0]
        bs = sf.shape[1] // self.nb
        dt = st["_dt"] if "_dt" in st else inp["dt"]
        out = [np.zeros((nsteps, bs, self.nb), dtype=np.int32) for _ in range(3)]
        self.lib.ref_branch_log(self.h, ctypes.c_int(bs), ctypes.c_int(nsteps), self.real(dt), self._p(sq), self._p(sd), self._p(sf),
                                self._p(a["body_inv_mass"]), self._p(a["body_inertia"]), self._p(a["body_inv_inertia"]),
                                self._p(out[0]), self._p(out[1]), self._p(out[2]))
        return dict(touch=out[0], slide=out[1], clamp=out[2])

    def fk_forward(self, joint_q, joint_qd):
        """joint_q [n,nq], joint_qd [n,nqd] -> body_q [n,nb,7], body_qd [n,nb,6]"""
        jq, jqd = self._c(joint_q), self._c(joint_qd)
        n = jq.shape[0]
        bq, bqd = np.zeros((n, self.nb, 7), self.dtype), np.zeros((n, self.nb, 6), self.dtype)
        self.lib.ref_fk_forward(self.h, ctypes.c_int(n), self._p(jq), self._p(jqd), self._p(bq), self._p(bqd))
        return bq, bqd

    def fk_backward(self, joint_q, joint_qd, body_q, adj_body_q, adj_body_qd):
        jq, jqd, bq = self._c(joint_q), self._c(joint_qd), self._c(body_q)
        n = jq.shape[0]
        gq, gqd = np.zeros((n, self.nq), self.dtype), np.zeros((n, self.nqd), self.dtype)
        self.lib.ref_fk_backward(self.h, ctypes.c_int(n), self._p(jq), self._p(jqd), self._p(bq),
                                 self._p(self._c(adj_body_q)), self._p(self._c(adj_body_qd)), self._p(gq), self._p(gqd))
        return gq, gqd
