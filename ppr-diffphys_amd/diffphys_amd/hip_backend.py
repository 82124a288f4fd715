"""ctypes binding of ``lib/libpprdiffphys_hip.so`` (C ABI: include/ppr_diffphys.h).

PyTorch is used here only as plumbing: device memory (``tensor.data_ptr()``) and
the current HIP stream.  There is NO fallback: if the library is missing or a
tensor is not a contiguous float32 CUDA(HIP) tensor, an exception is raised.
"""
import ctypes
import os

import numpy as np
import torch

# PPR_DIFFPHYS_LIB lets scripts/gpu_stamps.py load the -DPD_STAMPS diagnostic build; the product never sets it.
_LIB_PATH = os.environ.get("PPR_DIFFPHYS_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libpprdiffphys_hip.so")
_lib = None

NUM_STABLE, NUM_LITERAL = 0, 1  # PD_NUM_* of include/ppr_diffphys.h (DeviceModel.set_numeric_policy)
ABI_VERSION = 9  # PD_ABI_VERSION of include/ppr_diffphys.h this binding was written against

_fp = ctypes.POINTER(ctypes.c_float)
_ip = ctypes.POINTER(ctypes.c_int)


class _Desc(ctypes.Structure):
    _fields_ = [
        ("nb", ctypes.c_int), ("nq", ctypes.c_int), ("nqd", ctypes.c_int), ("nc", ctypes.c_int), ("nmat", ctypes.c_int),
        ("joint_type", _ip), ("joint_parent", _ip), ("joint_q_start", _ip), ("joint_qd_start", _ip),
        ("joint_X_p", _fp), ("joint_X_c", _fp), ("joint_axis", _fp), ("body_com", _fp),
        ("joint_limit_lower", _fp), ("joint_limit_upper", _fp), ("joint_limit_ke", _fp), ("joint_limit_kd", _fp),
        ("contact_body", _ip), ("contact_point", _fp), ("contact_dist", _fp), ("contact_material", _ip),
        ("shape_materials", _fp), ("gravity", ctypes.c_float * 3),
        ("joint_attach_ke", ctypes.c_float), ("joint_attach_kd", ctypes.c_float),
    ]


class _FkRide(ctypes.Structure):  # pd_fk_ride (include/ppr_diffphys.h)
    _fields_ = [("n", ctypes.c_int), ("bs", ctypes.c_int)] + [(k, ctypes.c_void_p) for k in (
        "joint_q", "joint_qd", "body_q", "body_qd", "adj_body_q", "adj_body_qd", "g_joint_q", "g_joint_qd")]


def lib_path():
    return _LIB_PATH


def lib():
    """Loads the HIP library; raises if it has not been built (``__graft_entry__.build()``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                "libpprdiffphys_hip.so not found at %s -- build it with `make -C ppr-diffphys_amd/csrc` "
                "(there is no CPU fallback for the product path)" % _LIB_PATH
            )
        L = ctypes.CDLL(_LIB_PATH)
        L.pd_last_error.restype = ctypes.c_char_p
        L.pd_rollout_workspace_floats.restype = ctypes.c_size_t
        L.pd_rollout_workspace_floats.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.pd_model_create.argtypes = [ctypes.POINTER(_Desc), ctypes.POINTER(ctypes.c_void_p)]
        L.pd_model_destroy.argtypes = [ctypes.c_void_p]
        L.pd_model_set_segment_width.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.pd_model_get_segment_width.argtypes = [ctypes.c_void_p]
        if hasattr(L, "pd_model_set_kernel_family"):
            L.pd_model_set_kernel_family.argtypes = [ctypes.c_void_p, ctypes.c_int]
            L.pd_model_get_kernel_family.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        L.pd_model_set_numeric_policy.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.pd_model_get_numeric_policy.argtypes = [ctypes.c_void_p]
        L.pd_last_kernel_ms.restype = ctypes.c_float
        L.pd_last_kernel_ms.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.pd_model_set_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.pd_last_launch_info.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int * 4)]
        if hasattr(L, "pd_model_contact_order"):  # (absent from older A/B builds loaded through PPR_DIFFPHYS_LIB)
            L.pd_model_contact_order.argtypes = [ctypes.c_void_p, _ip, ctypes.c_int]
        L.pd_model_bind_joint_X_p.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
        L.pd_rollout_forward.argtypes = [vp, ci, ci, cf] + [vp] * 10 + [ci, _ip] + [vp] * 5 + [vp]
        L.pd_rollout_backward.argtypes = [vp, ci, ci, cf] + [vp] * 9 + [ci, _ip] + [vp] * 3 + [vp] * 10 + [vp]
        if hasattr(L, "pd_rollout_forward_traj_loss"):
            L.pd_rollout_forward_traj_loss.argtypes = [vp, ci, ci, cf] + [vp] * 10 + [ci, _ip] + [vp] * 5 + [vp, vp, cf] + [vp] * 5 + [vp]
            L.pd_rollout_backward_traj_loss.argtypes = [vp, ci, ci, cf] + [vp] * 9 + [ci, _ip] + [vp] * 3 + [vp] * 4 + [vp] * 10 + [vp]
        if hasattr(L, "pd_rollout_forward_traj_loss_fk"):
            L.pd_rollout_forward_traj_loss_fk.argtypes = L.pd_rollout_forward_traj_loss.argtypes[:-1] + [ctypes.POINTER(_FkRide), vp]
            L.pd_rollout_backward_traj_loss_fk.argtypes = L.pd_rollout_backward_traj_loss.argtypes[:-1] + [ctypes.POINTER(_FkRide), vp]
        L.pd_fk_forward.argtypes = [vp, ci] + [vp] * 4 + [vp]
        L.pd_fk_backward.argtypes = [vp, ci] + [vp] * 6 + [vp]
        L.pd_se3_loss.argtypes = [ci, ci, vp, vp, cf, vp, vp, vp, vp]
        L.pd_reduce_loss.argtypes = [ci, ci, vp, ci, vp, vp, vp]
        L.pd_pose_op.argtypes = [ci, ci, vp, ci, vp, vp, vp]
        L.pd_pose_op_vjp.argtypes = [ci, ci, vp, ci, vp, vp, vp, vp, vp]
        L.pd_foot_height.argtypes = [ci, ci, ci] + [vp] * 6 + [vp]
        L.pd_foot_height_vjp.argtypes = [ci, ci] + [vp] * 6 + [vp]
        L.pd_colsum.argtypes = [ci, ci, vp, vp, vp, vp]
        L.pd_linear_wgrad_workspace_floats.restype = ctypes.c_size_t
        L.pd_linear_wgrad_workspace_floats.argtypes = [ci, ci, ci]
        L.pd_linear_wgrad.argtypes = [ci, ci, ci, vp, vp, vp, vp, vp, vp]
        if L.pd_abi_version() != ABI_VERSION and not (os.environ.get("PPR_DIFFPHYS_LIB") and os.environ.get("PPR_DIFFPHYS_ANY_ABI")):  # scripts/ab_time.sh times older builds
            raise RuntimeError("libpprdiffphys_hip.so ABI mismatch: library %d, binding %d" % (L.pd_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise RuntimeError("ppr_diffphys: " + lib().pd_last_error().decode())


def _dev(t, name, shape_numel=None):
    """Device pointer of a contiguous float32 GPU tensor (an int: ctypes converts it for the c_void_p parameters); anything else raises
    -- no silent copy, no CPU fallback.  The checks run on every call of every entry point: one fast path, messages on the slow one."""
    try:
        if t.is_cuda and t.dtype is torch.float32 and t.is_contiguous() and (shape_numel is None or t.numel() == shape_numel):
            return t.data_ptr()
    except AttributeError:
        pass
    if not torch.is_tensor(t):
        raise TypeError("%s must be a torch tensor" % name)
    if not t.is_cuda:
        raise ValueError("%s must live on the GPU (got %s)" % (name, t.device))
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32 (got %s)" % (name, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    raise ValueError("%s has %d elements, expected %d" % (name, t.numel(), shape_numel))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """torch's current HIP stream as a raw handle (torch.cuda.current_stream() builds a Stream object: ~7 us per call)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def build_id():
    """pd_build_id() of the loaded library: "<git HEAD at build time>+<source hash>"."""
    L = lib()
    L.pd_build_id.restype = ctypes.c_char_p
    return L.pd_build_id().decode()


def source_hash():
    """Hash of the library's sources as they sit in this tree (csrc/Makefile SRCS order; scripts/source_hash.py), or None when the
    sources are not there (an installed copy without csrc/)."""
    import hashlib

    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
    srcs = ["pd_kernels.hip", "pd_host.hip", "pd_loss.hip", "pd_pose.hip", "pd_mlp.hip", "pd_math.h", "pd_device.h", "pd_args.h", "pd_se3.h", "pd_quad.h", "pd_trajloss.h", "../../include/ppr_diffphys.h", "Makefile"]
    h = hashlib.sha256()
    try:
        for f in srcs:
            with open(os.path.join(csrc, f), "rb") as fh:
                h.update(fh.read())
    except OSError:
        return None
    return h.hexdigest()[:16]


def check_build_matches_sources():
    """Raises when the loaded library was not built from the sources beside it (a stale .so that travelled with the tree)."""
    want, have = source_hash(), build_id()
    if want is not None and not have.endswith("+" + want):
        raise RuntimeError("libpprdiffphys_hip.so is stale: built from sources %s, the tree's hash is %s (run make -C ppr-diffphys_amd/csrc)" % (have, want))
    return have


class DeviceModel:
    """Device copy of one articulation template (``pd_model``)."""

    def __init__(self, tpl):
        L = lib()
        f = lambda k: np.ascontiguousarray(tpl[k], dtype=np.float32)
        i = lambda k: np.ascontiguousarray(tpl[k], dtype=np.int32)
        self.nb, self.nq, self.nqd = int(tpl["nb"]), int(tpl["nq"]), int(tpl["nqd"])
        keep = dict(
            joint_type=i("joint_type"), joint_parent=i("joint_parent"), joint_q_start=i("joint_q_start"),
            joint_qd_start=i("joint_qd_start"), joint_X_p=f("joint_X_p"), joint_X_c=f("joint_X_c"),
            joint_axis=f("joint_axis"), body_com=f("body_com"), joint_limit_lower=f("joint_limit_lower"),
            joint_limit_upper=f("joint_limit_upper"), joint_limit_ke=f("joint_limit_ke"),
            joint_limit_kd=f("joint_limit_kd"), contact_body=i("contact_body"), contact_point=f("contact_point"),
            contact_dist=f("contact_dist"), contact_material=i("contact_material"), shape_materials=f("shape_materials"),
        )
        d = _Desc()
        d.nb, d.nq, d.nqd = self.nb, self.nq, self.nqd
        d.nc, d.nmat = len(keep["contact_body"]), len(keep["shape_materials"])
        for k, a in keep.items():
            setattr(d, k, a.ctypes.data_as(_ip if a.dtype == np.int32 else _fp))
        g = np.asarray(tpl["gravity"], dtype=np.float32)
        d.gravity = (ctypes.c_float * 3)(float(g[0]), float(g[1]), float(g[2]))
        d.joint_attach_ke = float(tpl["joint_attach_ke"])
        d.joint_attach_kd = float(tpl["joint_attach_kd"])
        h = ctypes.c_void_p()
        _check(L.pd_model_create(ctypes.byref(d), ctypes.byref(h)))
        self.h = h
        self._xp = None
        self._nc_keep = keep["contact_body"]

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().pd_model_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_segment_width(self, lanes):
        _check(lib().pd_model_set_segment_width(self.h, int(lanes)))

    def segment_width(self):
        return int(lib().pd_model_get_segment_width(self.h))

    def set_kernel_family(self, family):
        """0 automatic (by batch size), 1 lane per body always, 2 quad-lane (four lanes per body) wherever the robot is eligible."""
        _check(lib().pd_model_set_kernel_family(self.h, int(family)))

    def set_numeric_policy(self, policy):
        """NUM_STABLE (default) / NUM_LITERAL: how the revolute twist angle and the FIXED joint's angular error are evaluated by the rollout
        launches (``pd_model_set_numeric_policy``; LITERAL = the reference's acos forms, for side-by-side runs against Warp)."""
        _check(lib().pd_model_set_numeric_policy(self.h, int(policy)))

    def numeric_policy(self):
        return int(lib().pd_model_get_numeric_policy(self.h))

    def kernel_family(self):
        """(setting, eligible): eligible = the robot has quad-lane kernels (revolute-only, at most 16 bodies)."""
        el = ctypes.c_int(0)
        return int(lib().pd_model_get_kernel_family(self.h, ctypes.byref(el))), bool(el.value)

    def workspace_floats(self, bs, nsteps):
        return int(lib().pd_rollout_workspace_floats(self.h, bs, nsteps))

    # -- per-env joint_X_p (dp_interface.py:465 of the reference rebinds env.joint_X_p before every rollout) ----------
    def bind_joint_X_p(self, joint_X_p):
        """[n_envs*nb, 7] float32 GPU tensor (kept alive by this object) or None for the template's joint_X_p.
        A pointer swap on the host: no copy, no synchronisation, no rebuild."""
        if joint_X_p is None:
            _check(lib().pd_model_bind_joint_X_p(self.h, None, 0))
            self._xp = None
            return
        n = joint_X_p.numel() // (self.nb * 7)
        _check(lib().pd_model_bind_joint_X_p(self.h, _dev(joint_X_p, "joint_X_p", n * self.nb * 7), n))
        self._xp = joint_X_p

    # -- timing / launch geometry (bench.py) -------------------------------------------------------------------
    def set_timing(self, on):
        _check(lib().pd_model_set_timing(self.h, 1 if on else 0))

    def last_kernel_ms(self, kind):
        return float(lib().pd_last_kernel_ms(self.h, int(kind)))

    def last_launch_info(self, kind):
        out = (ctypes.c_int * 4)()
        _check(lib().pd_last_launch_info(self.h, int(kind), ctypes.byref(out)))
        return dict(workgroups=out[0], threads_per_wg=out[1], lds_bytes_per_wg=out[2], envs_per_wg=out[3])

    # -- rollout ----------------------------------------------------------------
    def alloc_rollout(self, bs, nsteps, nframes, device, want_forces=True, backward=True):
        """Workspace, frame outputs and gradient buffers of one (bs, nsteps, nframes) rollout, for callers that reuse them
        across iterations (pass as ``out=`` to rollout_forward / rollout_backward)."""
        nb, nq, nqd = self.nb, self.nq, self.nqd
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)
        o = dict(ws=e(self.workspace_floats(bs, nsteps)), wp_pos=e(nframes, bs * nb, 7), wp_vel=e(nframes, bs * nb, 6))
        if want_forces:
            o.update(grf=e(nframes, bs * nb, 6), jaf=e(nframes, bs * nb, 6))
        if backward:
            o["grads"] = self._alloc_grads(bs, nsteps, device)
        return o

    def _alloc_grads(self, bs, nsteps, device):
        nb, nq, nqd = self.nb, self.nq, self.nqd
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)
        return dict(q_init=e(bs * nq), qd_init=e(bs * nqd), torques=e(nsteps, bs * nqd), res_f=e(nsteps, bs * nb, 6),
                    refs=e(nsteps, bs * nqd), target_ke=e(bs * nqd), target_kd=e(bs * nqd), body_inv_mass=e(bs * nb),
                    body_inertia=e(bs * nb, 3, 3), body_inv_inertia=e(bs * nb, 3, 3))

    @staticmethod
    def _f2s(frame2step):
        f = [int(x) for x in frame2step]
        return (ctypes.c_int * max(len(f), 1))(*f), len(f)

    def rollout_forward(self, bs, nsteps, dt, q_init, qd_init, torques, res_f, refs, target_ke, target_kd, body_inv_mass,
                        body_inertia, body_inv_inertia, frame2step, want_forces=True, out=None):
        """-> wp_pos [F, bs*nb, 7], wp_vel [F, bs*nb, 6], grf, jaf [F, bs*nb, 6] (or None), workspace.
        frame2step: host sequence of F distinct ints in 0..nsteps (validated by the library before the launch)."""
        nb, nq, nqd = self.nb, self.nq, self.nqd
        dev = q_init.device
        f2s, nframes = self._f2s(frame2step)
        if out is None:
            out = self.alloc_rollout(bs, nsteps, nframes, dev, want_forces, backward=False)
        ws, wp_pos, wp_vel = out["ws"], out["wp_pos"], out["wp_vel"]
        grf, jaf = (out["grf"], out["jaf"]) if want_forces else (None, None)
        p = lambda t, name, n: _dev(t, name, n) if t.numel() else None  # empty tensors have a null data_ptr: the library accepts it
        _check(lib().pd_rollout_forward(
            self.h, bs, nsteps, float(dt), p(q_init, "q_init", bs * nq), p(qd_init, "qd_init", bs * nqd),
            p(torques, "torques", nsteps * bs * nqd), p(res_f, "res_f", nsteps * bs * nb * 6),
            p(refs, "refs", nsteps * bs * nqd), p(target_ke, "target_ke", bs * nqd),
            p(target_kd, "target_kd", bs * nqd), p(body_inv_mass, "body_inv_mass", bs * nb),
            p(body_inertia, "body_inertia", bs * nb * 9), p(body_inv_inertia, "body_inv_inertia", bs * nb * 9),
            nframes, f2s, p(ws, "workspace", self.workspace_floats(bs, nsteps)), p(wp_pos, "wp_pos", nframes * bs * nb * 7),
            p(wp_vel, "wp_vel", nframes * bs * nb * 6), p(grf, "grf", nframes * bs * nb * 6) if want_forces else None,
            p(jaf, "jaf", nframes * bs * nb * 6) if want_forces else None, _stream()))
        return wp_pos, wp_vel, grf, jaf, ws

    # -- rollout with the trajectory loss evaluated at the frame states (C ABI v5) ------------
    def rollout_forward_traj_loss(self, bs, nsteps, dt, q_init, qd_init, torques, res_f, refs, target_ke, target_kd, body_inv_mass,
                                  body_inertia, body_inv_inertia, frame2step, target_pos, outseq=None, rot_ratio=0.1, want_forces=True,
                                  want_seed_gt=True, out=None, fk=None):
        """``pd_rollout_forward_traj_loss``: rollout_forward plus, in the same rollout launch, se3_loss of every frame pose against
        target_pos [bs, F, nb, 7] and reduce_loss(clip=True) of the per-frame means.  -> (wp_pos, wp_vel, grf, jaf, ws, tl) with tl a dict:
        reduced [4] = (loss_traj, clip threshold, positive entries left, clipped envs), table [bs, F], scale [bs, F], seed_pos
        [F, bs*nb, 7], seed_gt [bs, F, nb, 7] or None.  outseq: bool / uint8 [bs, F] (entries the loss ignores) or None.
        fk = (joint_q [Ff, bs_f, nq], joint_qd [Ff, bs_f, nqd]): the FK of the control reference rides on the reduce_loss launch
        (``pd_rollout_forward_traj_loss_fk``); tl then also holds fk_body_q [bs_f, Ff, nb, 7] and fk_body_qd [bs_f, Ff, nb, 6]."""
        nb, nq, nqd = self.nb, self.nq, self.nqd
        dev = q_init.device
        f2s, nframes = self._f2s(frame2step)
        if out is None:
            out = self.alloc_rollout(bs, nsteps, nframes, dev, want_forces, backward=False)
        ws, wp_pos, wp_vel = out["ws"], out["wp_pos"], out["wp_vel"]
        grf, jaf = (out["grf"], out["jaf"]) if want_forces else (None, None)
        e = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
        tl = dict(reduced=e(4), table=e(bs, nframes), scale=e(bs, nframes), seed_pos=e(nframes, bs * nb, 7),
                  seed_gt=e(bs, nframes, nb, 7) if want_seed_gt else None)
        if outseq is not None:
            if not (outseq.is_cuda and outseq.dtype in (torch.bool, torch.uint8) and outseq.is_contiguous() and outseq.numel() == bs * nframes):
                raise ValueError("outseq must be a contiguous bool / uint8 GPU tensor of bs * nframes entries")
        p = lambda t, name, n: _dev(t, name, n) if (t is not None and t.numel()) else None
        ride = None
        if fk is not None:
            jq, jqd = fk
            Ff, bsf = int(jq.shape[0]), int(jq.shape[1])
            tl["fk_body_q"], tl["fk_body_qd"] = e(bsf, Ff, nb, 7), e(bsf, Ff, nb, 6)
            ride = _FkRide(Ff * bsf, bsf, p(jq, "fk joint_q", Ff * bsf * nq), p(jqd, "fk joint_qd", Ff * bsf * nqd),
                           p(tl["fk_body_q"], "fk body_q", None), p(tl["fk_body_qd"], "fk body_qd", None), None, None, None, None)
        entry = lib().pd_rollout_forward_traj_loss if ride is None else lib().pd_rollout_forward_traj_loss_fk
        _check(entry(
            self.h, bs, nsteps, float(dt), p(q_init, "q_init", bs * nq), p(qd_init, "qd_init", bs * nqd),
            p(torques, "torques", nsteps * bs * nqd), p(res_f, "res_f", nsteps * bs * nb * 6),
            p(refs, "refs", nsteps * bs * nqd), p(target_ke, "target_ke", bs * nqd),
            p(target_kd, "target_kd", bs * nqd), p(body_inv_mass, "body_inv_mass", bs * nb),
            p(body_inertia, "body_inertia", bs * nb * 9), p(body_inv_inertia, "body_inv_inertia", bs * nb * 9),
            nframes, f2s, p(ws, "workspace", self.workspace_floats(bs, nsteps)), p(wp_pos, "wp_pos", nframes * bs * nb * 7),
            p(wp_vel, "wp_vel", nframes * bs * nb * 6), p(grf, "grf", nframes * bs * nb * 6) if want_forces else None,
            p(jaf, "jaf", nframes * bs * nb * 6) if want_forces else None,
            p(target_pos, "target_pos", bs * nframes * nb * 7),
            ctypes.c_void_p(outseq.data_ptr()) if (outseq is not None and outseq.numel()) else None, ctypes.c_float(float(rot_ratio)),
            p(tl["seed_pos"], "seed_pos", nframes * bs * nb * 7), p(tl["seed_gt"], "seed_gt", nframes * bs * nb * 7),
            p(tl["table"], "loss_table", bs * nframes), _dev(tl["reduced"], "reduced", 4), p(tl["scale"], "scale", bs * nframes),
            *(() if ride is None else (ctypes.byref(ride),)), _stream()))
        return wp_pos, wp_vel, grf, jaf, ws, tl

    def rollout_backward_traj_loss(self, bs, nsteps, dt, q_init, qd_init, torques, refs, target_ke, target_kd, body_inv_mass,
                                   body_inertia, body_inv_inertia, frame2step, ws, tl, g_loss, adj_pos=None, adj_vel=None, out=None, fk=None):
        """``pd_rollout_backward_traj_loss``: the adjoint rollout seeded with g_loss (a 0-dim / 1-element GPU tensor: the upstream
        gradient of loss_traj) x scale / nb x seed_pos, plus adj_pos / adj_vel when given.
        fk = (joint_q [Ff, bs_f, nq], joint_qd [Ff, bs_f, nqd], adj_body_q [bs_f, Ff, nb, 7], adj_body_qd [bs_f, Ff, nb, 6]): the FK
        adjoint rides on the seeds launch (``pd_rollout_backward_traj_loss_fk``); g then also holds fk_joint_q [Ff, bs_f, nq] and
        fk_joint_qd [Ff, bs_f, nqd] (with ForwardKinematics.backward's post-processing)."""
        nb, nq, nqd = self.nb, self.nq, self.nqd
        dev = q_init.device
        f2s, nframes = self._f2s(frame2step)
        g = out["grads"] if out is not None else self._alloc_grads(bs, nsteps, dev)
        work = tl.get("work")
        if work is None:  # scratch for the seeds of this sweep (adj_pos / adj_vel layout), kept with the forward's outputs
            work = tl["work"] = torch.empty(nframes * bs * nb * 13, dtype=torch.float32, device=dev)
        p = lambda t, name, n=None: _dev(t, name, n) if (t is not None and t.numel()) else None
        ride = None
        if fk is not None:
            jq, jqd, aq, aqd = fk
            Ff, bsf = int(jq.shape[0]), int(jq.shape[1])
            g = dict(g)
            g["fk_joint_q"] = torch.empty(Ff, bsf, nq, dtype=torch.float32, device=dev)
            g["fk_joint_qd"] = torch.empty(Ff, bsf, nqd, dtype=torch.float32, device=dev)
            ride = _FkRide(Ff * bsf, bsf, p(jq, "fk joint_q", Ff * bsf * nq), p(jqd, "fk joint_qd", Ff * bsf * nqd), None, None,
                           p(aq, "fk adj_body_q", Ff * bsf * nb * 7), p(aqd, "fk adj_body_qd", Ff * bsf * nb * 6),
                           p(g["fk_joint_q"], "g"), p(g["fk_joint_qd"], "g"))
        entry = lib().pd_rollout_backward_traj_loss if ride is None else lib().pd_rollout_backward_traj_loss_fk
        _check(entry(
            self.h, bs, nsteps, float(dt), p(q_init, "q_init", bs * nq), p(qd_init, "qd_init", bs * nqd),
            p(torques, "torques", nsteps * bs * nqd), p(refs, "refs", nsteps * bs * nqd),
            p(target_ke, "target_ke", bs * nqd), p(target_kd, "target_kd", bs * nqd),
            p(body_inv_mass, "body_inv_mass", bs * nb), p(body_inertia, "body_inertia", bs * nb * 9),
            p(body_inv_inertia, "body_inv_inertia", bs * nb * 9), nframes, f2s,
            p(ws, "workspace", self.workspace_floats(bs, nsteps)), p(adj_pos, "adj_pos", nframes * bs * nb * 7),
            p(adj_vel, "adj_vel", nframes * bs * nb * 6), p(tl["seed_pos"], "seed_pos", nframes * bs * nb * 7),
            p(tl["scale"], "scale", bs * nframes), _dev(g_loss, "g_loss", 1), p(work, "seed_work", nframes * bs * nb * 13),
            p(g["q_init"], "g"), p(g["qd_init"], "g"),
            p(g["torques"], "g"), p(g["res_f"], "g"), p(g["refs"], "g"), p(g["target_ke"], "g"),
            p(g["target_kd"], "g"), p(g["body_inv_mass"], "g"), p(g["body_inertia"], "g"),
            p(g["body_inv_inertia"], "g"), *(() if ride is None else (ctypes.byref(ride),)), _stream()))
        return g

    def saved_trajectory(self, ws, bs, nsteps):
        """The trajectory a forward rollout saved for its adjoint, unpacked from the workspace (inspection / tests): states of
        steps 0 .. nsteps-1 in the reference's layouts -- body_q [T, bs*nb, 7], body_qd [T, bs*nb, 6] (angular first), the total
        body wrench body_f [T, bs*nb, 6] (torque first) -- and the 6-bit mask of the velocity components each step's
        integration clamped [T, bs*nb] (csrc/pd_kernels.hip: PD_TRAJ_G planes of float4)."""
        N = bs * self.nb
        pl = ws[: nsteps * 20 * N].view(nsteps, 5, N, 4)
        q, wv, pv, vt, fm = pl[:, 0], pl[:, 1], pl[:, 2], pl[:, 3], pl[:, 4]
        body_q = torch.cat([pv[..., :3], q], dim=-1)
        body_qd = torch.cat([wv[..., :3], wv[..., 3:4], pv[..., 3:4], vt[..., 0:1]], dim=-1)
        body_f = torch.cat([vt[..., 1:4], fm[..., :3]], dim=-1)
        mask = fm[..., 3].contiguous().view(torch.int32)
        return body_q, body_qd, body_f, mask

    def contact_order(self):
        """order[i] = index into the template's contact_* arrays of entry i of the device contact table (``pd_model_contact_order``)."""
        n = len(self._nc_keep)
        out = (ctypes.c_int * max(n, 1))()
        _check(lib().pd_model_contact_order(self.h, out, n))
        return np.frombuffer(out, dtype=np.int32, count=n).copy()

    def saved_hit_log(self, ws, bs, nsteps):
        """The contact hit log behind the trajectory planes (inspection / tests): int32 [T, bs, 32] -- [..., 0] the number of
        candidates that touched in that env-step (-1: more than 31), then that many TEMPLATE contact indices (decoded through
        contact_order(); the packed material / body bits are dropped), -1 padded."""
        N = bs * self.nb
        raw = ws[nsteps * 20 * N:].view(torch.int32)[: nsteps * bs * 32].view(nsteps, bs, 32).cpu().numpy()
        order = self.contact_order()
        cnt = raw[..., 0]
        valid = np.arange(31)[None, None, :] < np.maximum(cnt, 0)[..., None]
        pts = np.where(valid, raw[..., 1:] & 0xFFFF, 0)
        out = np.full(raw.shape, -1, np.int32)
        out[..., 0] = cnt
        out[..., 1:] = np.where(valid, order[pts] if len(order) else 0, -1)
        return out

    def rollout_backward(self, bs, nsteps, dt, q_init, qd_init, torques, refs, target_ke, target_kd, body_inv_mass,
                         body_inertia, body_inv_inertia, frame2step, ws, adj_pos, adj_vel, out=None):
        nb, nq, nqd = self.nb, self.nq, self.nqd
        dev = q_init.device
        f2s, nframes = self._f2s(frame2step)
        g = out["grads"] if out is not None else self._alloc_grads(bs, nsteps, dev)
        p = lambda t, name, n=None: _dev(t, name, n) if t.numel() else None
        _check(lib().pd_rollout_backward(
            self.h, bs, nsteps, float(dt), p(q_init, "q_init", bs * nq), p(qd_init, "qd_init", bs * nqd),
            p(torques, "torques", nsteps * bs * nqd), p(refs, "refs", nsteps * bs * nqd),
            p(target_ke, "target_ke", bs * nqd), p(target_kd, "target_kd", bs * nqd),
            p(body_inv_mass, "body_inv_mass", bs * nb), p(body_inertia, "body_inertia", bs * nb * 9),
            p(body_inv_inertia, "body_inv_inertia", bs * nb * 9), nframes, f2s,
            p(ws, "workspace", self.workspace_floats(bs, nsteps)), p(adj_pos, "adj_pos", nframes * bs * nb * 7),
            p(adj_vel, "adj_vel", nframes * bs * nb * 6), p(g["q_init"], "g"), p(g["qd_init"], "g"),
            p(g["torques"], "g"), p(g["res_f"], "g"), p(g["refs"], "g"), p(g["target_ke"], "g"),
            p(g["target_kd"], "g"), p(g["body_inv_mass"], "g"), p(g["body_inertia"], "g"),
            p(g["body_inv_inertia"], "g"), _stream()))
        return g

    # -- FK -----------------------------------------------------------------------
    def fk_forward(self, joint_q, joint_qd):
        n = joint_q.numel() // self.nq
        dev = joint_q.device
        body_q = torch.empty(n, self.nb, 7, dtype=torch.float32, device=dev)
        body_qd = torch.empty(n, self.nb, 6, dtype=torch.float32, device=dev)
        if n == 0:
            return body_q, body_qd
        _check(lib().pd_fk_forward(self.h, n, _dev(joint_q, "joint_q", n * self.nq), _dev(joint_qd, "joint_qd", n * self.nqd),
                                   _dev(body_q, "body_q"), _dev(body_qd, "body_qd"), _stream()))
        return body_q, body_qd

    def fk_backward(self, joint_q, joint_qd, adj_body_q, adj_body_qd):
        n = joint_q.numel() // self.nq
        dev = joint_q.device
        gq = torch.empty(n, self.nq, dtype=torch.float32, device=dev)
        gqd = torch.empty(n, self.nqd, dtype=torch.float32, device=dev)
        if n == 0:
            return gq, gqd
        _check(lib().pd_fk_backward(self.h, n, _dev(joint_q, "joint_q", n * self.nq), _dev(joint_qd, "joint_qd", n * self.nqd),
                                    _dev(adj_body_q, "adj_body_q", n * self.nb * 7),
                                    _dev(adj_body_qd, "adj_body_qd", n * self.nb * 6), _dev(gq, "g"), _dev(gqd, "g"), _stream()))
        return gq, gqd


def se3_loss(pred, gt, rot_ratio, want_grads=True):
    """Fused se3_loss (``pd_se3_loss``): (..., 7) or (..., 6) float32 GPU tensors -> loss (...), d loss/d pred, d loss/d gt."""
    dim = pred.shape[-1]
    if dim not in (6, 7) or gt.shape != pred.shape:
        raise ValueError("se3_loss: pred and gt must both be (..., 7) or (..., 6); got %s and %s" % (tuple(pred.shape), tuple(gt.shape)))
    n = pred.numel() // dim
    loss = torch.empty(pred.shape[:-1], device=pred.device, dtype=torch.float32)
    gp = torch.empty_like(pred) if want_grads else None
    gg = torch.empty_like(gt) if want_grads else None
    null = ctypes.c_void_p(0)
    rc = lib().pd_se3_loss(n, dim, _dev(pred, "pred"), _dev(gt, "gt"), ctypes.c_float(float(rot_ratio)), _dev(loss, "loss"),
                           _dev(gp, "g_pred") if want_grads else null, _dev(gg, "g_gt") if want_grads else null, _stream())
    if rc != 0:
        raise RuntimeError("pd_se3_loss failed (rc %d)" % rc)
    return loss, gp, gg


def reduce_loss(table, clip=False, want_scale=True):
    """reduce_loss of the reference (dp_utils.py:93-110) as ONE one-workgroup launch (``pd_reduce_loss``) on a float32 GPU table
    (bs, F): with clip the table is truncated IN PLACE like the reference's argument.  Returns reduced [4] = (value, threshold, positive
    entries left, clipped envs) and scale (bs, F) = d value / d entry (None unless want_scale)."""
    if table.dim() != 2:
        raise ValueError("reduce_loss: a (bs, F) table; got %s" % (tuple(table.shape),))
    bs, F = table.shape
    reduced = torch.empty(4, device=table.device, dtype=torch.float32)
    scale = torch.empty_like(table) if want_scale else None
    rc = lib().pd_reduce_loss(bs, F, _dev(table, "table") if table.numel() else ctypes.c_void_p(0), 1 if clip else 0, _dev(reduced, "reduced"),
                              _dev(scale, "scale") if want_scale and table.numel() else ctypes.c_void_p(0), _stream())
    if rc != 0:
        raise RuntimeError("pd_reduce_loss failed (rc %d)" % rc)
    return reduced, scale


POSE_COMPOSE_DELTA, POSE_ROTATE_FRAME, POSE_ROTATE_VEL = 0, 1, 2
_POSE_DIMS = {POSE_COMPOSE_DELTA: (7, 6, 7), POSE_ROTATE_FRAME: (7, 7, 7), POSE_ROTATE_VEL: (7, 6, 6)}


def _pose_n(op, a, b):
    na, nb_, no = _POSE_DIMS[op]
    if b.shape[-1] != nb_ or a.shape[-1] != na:
        raise ValueError("pose op %d: operands must end in %d and %d floats; got %s and %s" % (op, na, nb_, tuple(a.shape), tuple(b.shape)))
    n = b.numel() // nb_
    bcast = a.numel() == na and n != 1
    if not bcast and a.numel() != n * na:
        raise ValueError("pose op %d: %d poses against %d operands" % (op, a.numel() // na, n))
    return n, bcast, no


def pose_op(op, a, b):
    """``pd_pose_op``: compose_delta(a = target (..., 7), b = delta (..., 6)), rotate_frame(a = global (7,) or (..., 7),
    b = target (..., 7)) or rotate_frame_vel(a = global, b = (..., 6)) in one launch; float32 contiguous GPU tensors."""
    n, bcast, no = _pose_n(op, a, b)
    out = torch.empty(b.shape[:-1] + (no,), device=b.device, dtype=torch.float32)
    rc = lib().pd_pose_op(op, n, _dev(a, "a"), int(bcast), _dev(b, "b"), _dev(out, "out"), _stream())
    if rc != 0:
        raise RuntimeError("pd_pose_op failed (rc %d)" % rc)
    return out


COLSUM_SLICES = 32   # PD_COLSUM_SLICES of the header


def colsum(x):
    """x [n, k] -> [k], the column sums as ONE launch of the library's own kernel (``pd_colsum``: fixed summation order, no atomics, no
    second stage) instead of torch's reduction: torch's multi-block reductions of this shape (many rows, few columns) replay STALE inside a
    captured HIP graph on this stack (scripts/micro/torch_graph_replay2.py), and everything on phys_model's iteration path must survive
    ``phys_model.capture_iteration``.  Bias gradients of the time-MLPs, the gradient of a broadcast pose operand."""
    n, k = int(x.shape[0]), int(x.shape[1])
    out = torch.empty(k, dtype=torch.float32, device=x.device)
    ws = torch.empty(COLSUM_SLICES * k, dtype=torch.float32, device=x.device) if n > 1024 else None   # row slices first, then the slices in order
    rc = lib().pd_colsum(n, k, _dev(x, "x", n * k) if n * k else None, _dev(out, "out") if k else None, _dev(ws, "ws") if ws is not None else None,
                         _stream())
    if rc != 0:
        raise RuntimeError("pd_colsum failed (rc %d)" % rc)
    return out


def linear_wgrad(g, x, want_bias=True):
    """``pd_linear_wgrad``: (g^T x [m, kin], column sums of g [m] or None) for g [n, m] = dL/dy and x [n, kin] -- the weight and bias
    gradient of a linear layer on the fp32 matrix cores, fixed summation order.  Returns None when the shape is not served (m or kin not a
    multiple of 128): the caller keeps its BLAS path."""
    n, m = int(g.shape[0]), int(g.shape[1])
    kin = int(x.shape[1])
    ws_floats = lib().pd_linear_wgrad_workspace_floats(n, m, kin)
    if ws_floats == 0 or int(x.shape[0]) != n:
        return None
    gw = torch.empty(m, kin, dtype=torch.float32, device=g.device)
    gb = torch.empty(m, dtype=torch.float32, device=g.device) if want_bias else None
    ws = torch.empty(ws_floats, dtype=torch.float32, device=g.device)
    rc = lib().pd_linear_wgrad(n, m, kin, _dev(g, "g", n * m), _dev(x, "x", n * kin), _dev(gw, "gw"), _dev(gb, "gb") if want_bias else None, _dev(ws, "ws"), _stream())
    if rc != 0:
        raise RuntimeError("pd_linear_wgrad failed (rc %d)" % rc)
    return gw, gb


def pose_op_vjp(op, a, b, g_out, need_a=True, need_b=True):
    """``pd_pose_op_vjp``: (g_a, g_b) for the upstream gradient g_out; g_a is already summed when ``a`` was broadcast."""
    n, bcast, no = _pose_n(op, a, b)
    na = _POSE_DIMS[op][0]
    g_a = torch.empty(b.shape[:-1] + (na,), device=b.device, dtype=torch.float32) if need_a else None
    g_b = torch.empty_like(b) if need_b else None
    null = ctypes.c_void_p(0)
    rc = lib().pd_pose_op_vjp(op, n, _dev(a, "a"), int(bcast), _dev(b, "b"), _dev(g_out, "g_out", n * no),
                              _dev(g_a, "g_a") if need_a else null, _dev(g_b, "g_b") if need_b else null, _stream())
    if rc != 0:
        raise RuntimeError("pd_pose_op_vjp failed (rc %d)" % rc)
    if need_a:
        g_a = colsum(g_a.reshape(-1, na)).reshape(a.shape) if bcast else g_a.reshape(a.shape)
    return g_a, g_b


def _dev_int(t, name):
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()):
        raise TypeError("%s must be a contiguous int32 GPU tensor" % name)
    return ctypes.c_void_p(t.data_ptr())


def foot_height(body_q, c_body, c_point, c_dist):
    """``pd_foot_height``: body_q (..., nb, 7) -> (height (...), arg-min candidate (...) int32)."""
    nb = body_q.shape[-2]
    n = body_q.numel() // (nb * 7)
    h = torch.empty(body_q.shape[:-2], device=body_q.device, dtype=torch.float32)
    arg = torch.empty(body_q.shape[:-2], device=body_q.device, dtype=torch.int32)
    rc = lib().pd_foot_height(n, nb, c_body.numel(), _dev(body_q, "body_q"), _dev_int(c_body, "c_body"), _dev(c_point, "c_point", c_body.numel() * 3),
                              _dev(c_dist, "c_dist", c_body.numel()), _dev(h, "height"), _dev_int(arg, "arg"), _stream())
    if rc != 0:
        raise RuntimeError("pd_foot_height failed (rc %d)" % rc)
    return h, arg


def foot_height_vjp(body_q, c_body, c_point, arg, g_h):
    nb = body_q.shape[-2]
    n = body_q.numel() // (nb * 7)
    g = torch.empty_like(body_q)
    rc = lib().pd_foot_height_vjp(n, nb, _dev(body_q, "body_q"), _dev_int(c_body, "c_body"), _dev(c_point, "c_point"), _dev_int(arg, "arg"),
                                  _dev(g_h, "g_height", n), _dev(g, "g_body_q"), _stream())
    if rc != 0:
        raise RuntimeError("pd_foot_height_vjp failed (rc %d)" % rc)
    return g


def device_model(env):
    """DeviceModel of a :class:`diffphys_amd.sim.Model`, cached on the model."""
    if env._handle is None:
        env._handle = DeviceModel(env.template())
    return env._handle
