"""Host-side logic: model compiler, URDF/mesh readers, dataloader, synthetic workloads, C-ABI surface."""
import ctypes
import os
import re

import numpy as np
import pytest

from helpers import ROOT
from diffphys_amd import dataloader, robots, sim, synth
from diffphys_amd.import_urdf import parse_urdf

REF_DATA = "/root/reference/data/urdf_templates"

URDF = """<?xml version="1.0"?>
<robot name="toy">
  <link name="base"><collision><origin xyz="0 0.1 0" rpy="0 0 0"/><geometry><box size="0.4 0.2 0.2"/></geometry></collision></link>
  <link name="arm"><collision><origin xyz="0 -0.1 0" rpy="0 0 0"/><geometry><sphere radius="0.05"/></geometry></collision></link>
  <link name="leg_R"><collision><geometry><sphere radius="0.01"/></geometry></collision></link>
  <link name="leg_P"><collision><geometry><sphere radius="0.01"/></geometry></collision></link>
  <link name="leg_Y"><collision><origin xyz="0 -0.2 0"/><geometry><mesh filename="tet.obj"/></geometry></collision></link>
  <link name="tip"><collision><geometry><cylinder radius="0.02" length="0.1"/></geometry></collision></link>
  <joint name="j_arm" type="continuous"><parent link="base"/><child link="arm"/><axis xyz="0 0 1"/><origin xyz="0.2 0 0" rpy="0 0 0.5"/><limit effort="1" velocity="1"/></joint>
  <joint name="j_leg_R" type="revolute"><parent link="base"/><child link="leg_R"/><axis xyz="1 0 0"/><origin xyz="-0.2 0 0" rpy="0.1 0 0"/><limit lower="-1.5" upper="1.5" effort="1" velocity="1"/></joint>
  <joint name="j_leg_P" type="revolute"><parent link="leg_R"/><child link="leg_P"/><axis xyz="0 1 0"/></joint>
  <joint name="j_leg_Y" type="revolute"><parent link="leg_P"/><child link="leg_Y"/><axis xyz="0 0 1"/></joint>
  <joint name="j_tip" type="fixed"><parent link="leg_Y"/><child link="tip"/><origin xyz="0 -0.4 0"/></joint>
</robot>
"""
OBJ = "v 0 0 0\nv 0.1 0 0\nv 0 0.1 0\nv 0 0 0.1\nv 0 0 0\nf 1 3 2\nf 1 2 4\nf 2 3 4\nf 1 4 3\n"


@pytest.fixture()
def toy(tmp_path):
    (tmp_path / "toy.urdf").write_text(URDF)
    (tmp_path / "tet.obj").write_text(OBJ)
    b = sim.ModelBuilder()
    parse_urdf(str(tmp_path / "toy.urdf"), b, xform=sim.transform((0, 0.5, 0), sim.quat_identity()), floating=True, density=1000.0,
               armature=0.01, stiffness=220.0, damping=2.0, shape_ke=1e4, shape_kd=0.0, shape_kf=1e2, shape_mu=1.0, limit_ke=0.0, limit_kd=0.0)
    return b


def test_parse_urdf_joint_mapping(toy):
    b = toy
    # root FREE, continuous->REVOLUTE, *_R -> COMPOUND with the *_Y link as child, *_P/*_Y skipped, fixed->FIXED
    assert b.joint_type == [sim.JOINT_FREE, sim.JOINT_REVOLUTE, sim.JOINT_COMPOUND, sim.JOINT_FIXED]
    assert b.joint_parent == [-1, 0, 0, 2]
    assert b.joint_q_start == [0, 7, 8, 11] and b.joint_qd_start == [0, 6, 7, 10]
    assert b.joint_coord_count == 11 and b.joint_dof_count == 10
    assert b.joint_q[:7] == [0.0, 0.5, 0.0, 0.0, 0.0, 0.0, 1.0]
    # limits: continuous keeps +-1e3, the compound joint takes the _R joint's limits on all three axes
    assert b.joint_limit_lower[6] == -1e3 and b.joint_limit_upper[6] == 1e3
    assert b.joint_limit_lower[7:10] == [-1.5] * 3 and b.joint_limit_upper[7:10] == [1.5] * 3
    assert np.allclose(b.joint_X_p[1].p, [0.2, 0, 0]) and np.allclose(b.joint_X_p[1].q, sim.quat_rpy(0, 0, 0.5))


def test_mass_properties_and_contacts(toy):
    b = toy
    # box 0.4 x 0.2 x 0.2 at density 1000, com = collision origin, inertia = armature + box inertia
    m = 1000 * 0.4 * 0.2 * 0.2
    assert abs(b.body_mass[0] - m) < 1e-9 and np.allclose(b.body_com[0], [0, 0.1, 0])
    assert np.allclose(np.diag(b.body_inertia[0]), 0.01 + m / 12 * np.array([0.2**2 + 0.2**2, 0.4**2 + 0.2**2, 0.4**2 + 0.2**2]))
    # tetrahedron (duplicate vertex merged): volume 0.1^3/6
    assert abs(b.body_mass[2] - 1000 * 0.1**3 / 6) < 1e-9
    assert len(b.shape_geo_src[2].vertices) == 4
    top = sim.ModelBuilder()
    top.add_rigid_articulation(b)
    top.add_rigid_articulation(b)
    env = top.finalize("cpu")
    env.collide(None)
    # box 8 corners + sphere 1 + mesh 4 vertices + capsule 2
    assert len(env.t_contact_body) == 8 + 1 + 4 + 2 and env.contact_count == 2 * 15
    assert list(env.t_contact_body) == [0] * 8 + [1] + [2] * 4 + [3] * 2
    assert np.allclose(env.t_contact_point[8], [0, -0.1, 0]) and abs(env.t_contact_dist[8] - 0.05) < 1e-7
    assert np.allclose(sorted(env.t_contact_point[:8, 1]), [0.0] * 4 + [0.2] * 4)
    assert env.body_count == 8 and env.num_envs == 2 and env.body_com.shape == (8, 3)
    with pytest.raises(NotImplementedError):
        top.add_rigid_articulation(sim.ModelBuilder())


@pytest.mark.parametrize("name,nb,nq,nqd,nc", [("laikago", 13, 19, 18, 3838), ("human", 19, 61, 60, 152), ("quad", 26, 82, 81, 208)])
def test_compiled_templates(name, nb, nq, nqd, nc):
    """Committed templates have the sizes of SURVEY.md section 8; if the reference data is mounted, recompiling reproduces them."""
    tpl = robots.load_template(name)
    assert (int(tpl["nb"]), int(tpl["nq"]), int(tpl["nqd"]), len(tpl["contact_body"])) == (nb, nq, nqd, nc)
    assert tpl["joint_type"][0] == sim.JOINT_FREE and tpl["joint_parent"][0] == -1
    assert np.all(tpl["joint_parent"][1:] < np.arange(1, nb))
    if os.path.isdir(REF_DATA):
        env, _, info = robots.make_env(name, REF_DATA, 1, device="cpu")
        fresh = env.template()
        for k, v in fresh.items():
            assert np.allclose(np.asarray(v, np.float64), np.asarray(tpl[k], np.float64), rtol=0, atol=1e-7), k
    if name == "human":
        assert np.allclose(tpl["body_mass"], 1.0)
    if name == "laikago":
        assert abs(float(tpl["body_mass"].sum()) - 7.7) < 0.1 and float(tpl["joint_attach_ke"]) == 16000.0


def test_dataloader_and_bullet2gl():
    dl = dataloader.DataLoader({"seqname": "mi-pace"})
    assert dl.amp_info.shape == (39, 85) and abs(dl.frame_interval - 0.01667) < 1e-9
    assert list(dl.data_info["offset"]) == [0, 39]
    msm = dataloader.parse_amp(dl.amp_info[:2].copy())
    pos0, orn0 = msm["pos"].copy(), msm["orn"].copy()
    dataloader.bullet2gl(msm, False)
    assert np.allclose(msm["pos"], pos0[:, [1, 2, 0]]) and np.allclose(msm["orn"][:, :3], orn0[:, [1, 2, 0]])
    assert np.allclose(msm["orn"][:, 3], orn0[:, 3]) and msm["jang"].shape == (2, 12)


def test_synth_inputs_shapes_and_ground_touch():
    tpl = robots.load_template("laikago")
    inp = synth.make_inputs(tpl, "laikago", bs=5, nsteps=67, seed=0, seqs=("mi-trot", "mi-spin"))
    assert inp["frame2step"] == [0, 33, 66]
    assert inp["refs"].shape == (67, 5 * 18) and inp["res_f"].shape == (67, 5 * 13, 6) and inp["body_inertia"].shape == (65, 3, 3)
    assert np.all(inp["refs"].reshape(67, 5, 18)[:, :, :6] == 0)
    low = synth.lowest_contact_y(tpl, inp["q_init"].reshape(5, -1).astype(np.float64))
    assert np.abs(low).max() < 1e-6
    again = synth.make_inputs(tpl, "laikago", bs=5, nsteps=67, seed=0, seqs=("mi-trot", "mi-spin"))
    assert all(np.array_equal(inp[k], again[k]) for k in synth.INPUT_NAMES)


def test_c_abi_exports_every_declared_symbol():
    """The built library loads without a GPU and exports exactly what include/ppr_diffphys.h declares."""
    from diffphys_amd import hip_backend

    hdr = open(os.path.join(ROOT, "include", "ppr_diffphys.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(pd_[A-Za-z0-9_]+)\s*\(", hdr)))
    assert "pd_rollout_forward" in names and "pd_fk_backward" in names and "pd_model_bind_joint_X_p" in names and "pd_pose_op_vjp" in names and "pd_build_id" in names and "pd_model_contact_order" in names and "pd_rollout_backward_traj_loss" in names and "pd_model_set_kernel_family" in names and "pd_rollout_backward_traj_loss_fk" in names and "pd_reduce_loss" in names and "pd_model_set_numeric_policy" in names and "pd_colsum" in names and "pd_linear_wgrad" in names and len(names) == 34
    lib = hip_backend.lib()
    for n in names:
        assert hasattr(lib, n), "missing symbol " + n
    # ... and nothing else: the A/B variants and debug hooks of earlier rounds live in -DPD_EXPERIMENT / -DPD_STAMPS builds only
    import subprocess
    exported = {l.split()[-1] for l in subprocess.check_output(["nm", "-D", "--defined-only", hip_backend.lib_path()]).decode().splitlines() if " T pd_" in l}
    assert exported == set(names), sorted(exported ^ set(names))
    # the binary was built from the sources beside it (pd_build_id = "<git HEAD>+<source hash>")
    assert hip_backend.check_build_matches_sources().endswith("+" + hip_backend.source_hash())
    assert lib.pd_abi_version() == 9 == hip_backend.ABI_VERSION
    assert int(re.search(r"#define PD_ABI_VERSION (\d+)", hdr).group(1)) == hip_backend.ABI_VERSION == 9
    lib.pd_rollout_workspace_floats.restype = ctypes.c_size_t
    assert lib.pd_rollout_workspace_floats(None, 4, 10) == 0


def test_docs_name_the_headers_abi_version_and_symbol_count():
    """INTEGRATION.md told a maintainer `PD_ABI_VERSION` is 6 while header and library were at 9.  The two places that
    state the number -- and the symbol count -- are held to include/ppr_diffphys.h."""
    hdr = open(os.path.join(ROOT, "include", "ppr_diffphys.h")).read()
    version = int(re.search(r"#define PD_ABI_VERSION (\d+)", hdr).group(1))
    nsym = len(set(re.findall(r"\b(pd_[A-Za-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", hdr, flags=re.S))))
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    readme = open(os.path.join(ROOT, "README.md")).read()
    m = re.search(r"`PD_ABI_VERSION` is (\d+) \(`pd_abi_version\(\)`; (\d+) exported symbols", integ)
    assert m and (int(m.group(1)), int(m.group(2))) == (version, nsym), (m and m.groups(), version, nsym)
    m = re.search(r"C ABI `include/ppr_diffphys.h`, \*\*version (\d+)\*\*", design)
    assert m and int(m.group(1)) == version
    m = re.search(r"types; (\d+) symbols = what the header declares", design)
    assert m and int(m.group(1)) == nsym
    m = re.search(r"\(ABI v(\d+)\)", readme)
    assert m and int(m.group(1)) == version


def test_main_refuses_the_flags_it_does_not_build_and_sets_progress_before_the_eval_pass():
    """VERDICT r5 missing #6 / weak #7.  `--pos_distill_wt > 0` enters get_distilled_kinematics in the reference (dp_model.py:800-804, lab4d):
    refused loudly here, as is `--reg_root_wt`; and main.py sets `model.progress` BEFORE the evaluation pass as the reference's main.py:64."""
    import importlib.util

    path = os.path.join(ROOT, "ppr-diffphys_amd", "main.py")
    spec = importlib.util.spec_from_file_location("pd_main_cpu", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.get_opts([])["pos_distill_wt"] == 0.0
    for flag in ("--pos_distill_wt", "--reg_root_wt"):
        with pytest.raises(NotImplementedError, match="lab4d"):
            m.get_opts([flag, "0.1"])
    src = open(path).read()
    loop = src[src.index("for it in range(model.total_iters)"):]
    assert loop.index("model.progress = it /") < loop.index("reinit_envs(1, frames_per_wdw=model.total_frames")


def test_c_abi_argument_errors_without_a_gpu():
    """Every entry point refuses a null model / bad arguments with a non-zero code and a message (no compute, no GPU
    needed); creating a model on a box without a GPU fails loudly instead of falling back to anything."""
    from diffphys_amd import hip_backend

    lib = hip_backend.lib()
    err = lambda: lib.pd_last_error().decode()
    f2s = (ctypes.c_int * 2)(0, 1)
    assert lib.pd_rollout_forward(None, 1, 1, ctypes.c_float(5e-4), *([None] * 10), 2, f2s, *([None] * 5), None) != 0
    assert "null model" in err()
    assert lib.pd_rollout_backward(None, 1, 1, ctypes.c_float(5e-4), *([None] * 9), 2, f2s, *([None] * 13), None) != 0
    assert lib.pd_rollout_forward_traj_loss(None, 1, 1, ctypes.c_float(5e-4), *([None] * 10), 2, f2s, *([None] * 5), None, None, ctypes.c_float(0.1), *([None] * 5), None) != 0
    assert lib.pd_rollout_backward_traj_loss(None, 1, 1, ctypes.c_float(5e-4), *([None] * 9), 2, f2s, *([None] * 17), None) != 0 and "null model" in err()
    assert lib.pd_rollout_forward_traj_loss_fk(None, 1, 1, ctypes.c_float(5e-4), *([None] * 10), 2, f2s, *([None] * 5), None, None, ctypes.c_float(0.1), *([None] * 5), None, None) != 0
    assert lib.pd_rollout_backward_traj_loss_fk(None, 1, 1, ctypes.c_float(5e-4), *([None] * 9), 2, f2s, *([None] * 17), None, None) != 0 and "null model" in err()
    assert lib.pd_fk_forward(None, 1, None, None, None, None, None) != 0 and "null model" in err()
    # pose algebra / foot height: bad op, negative count, null operands are refused before any launch; n = 0 is a no-op
    assert lib.pd_pose_op(7, 1, None, 0, None, None, None) != 0 and lib.pd_pose_op(0, -1, None, 0, None, None, None) != 0
    assert lib.pd_pose_op(0, 4, None, 0, None, None, None) != 0 and lib.pd_pose_op(0, 0, None, 0, None, None, None) == 0
    assert lib.pd_pose_op_vjp(1, 4, None, 1, None, None, None, None, None) != 0 and lib.pd_pose_op_vjp(1, 0, None, 1, None, None, None, None, None) == 0
    assert lib.pd_foot_height(4, 13, 0, *([None] * 7)) != 0 and lib.pd_foot_height(4, 13, 8, *([None] * 7)) != 0 and lib.pd_foot_height(0, 13, 8, *([None] * 7)) == 0
    assert lib.pd_foot_height_vjp(4, 13, *([None] * 7)) != 0 and lib.pd_foot_height_vjp(0, 13, *([None] * 7)) == 0
    lib.pd_linear_wgrad_workspace_floats.restype = ctypes.c_size_t
    assert lib.pd_linear_wgrad_workspace_floats(7600, 256, 512) == 32 * (256 * 512 + 256) and lib.pd_linear_wgrad_workspace_floats(7600, 256, 21) == 0
    assert lib.pd_linear_wgrad(7600, 256, 21, *([None] * 6)) != 0 and lib.pd_linear_wgrad(7600, 256, 256, *([None] * 6)) != 0
    assert lib.pd_colsum(4, 8, None, None, None, None) != 0 and lib.pd_colsum(-1, 8, None, None, None, None) != 0 and lib.pd_colsum(4, 0, None, None, None, None) == 0
    assert lib.pd_model_bind_joint_X_p(None, None, 0) != 0
    assert lib.pd_model_contact_order(None, None, 0) != 0
    assert lib.pd_model_set_kernel_family(None, 2) != 0 and lib.pd_model_get_kernel_family(None, None) == 0
    assert lib.pd_model_set_timing(None, 1) != 0
    assert lib.pd_last_kernel_ms(None, 0) < 0
    info = (ctypes.c_int * 4)()
    assert lib.pd_last_launch_info(None, 0, ctypes.byref(info)) != 0
    assert lib.pd_model_set_segment_width(None, 16) != 0 and lib.pd_model_get_segment_width(None) == 0
    import torch

    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="ppr_diffphys"):
            hip_backend.DeviceModel(robots.load_template("laikago"))


def test_boundary_rejects_bad_tensors():
    """No silent fallback: CPU tensors / wrong dtype raise instead of computing somewhere else."""
    import torch
    from diffphys_amd import hip_backend

    with pytest.raises(ValueError):
        hip_backend._dev(torch.zeros(4), "x")
    with pytest.raises(TypeError):
        hip_backend._dev(np.zeros(4), "x")


def test_convert_and_lazy_host_frames():
    from diffphys_amd.dp_model import HostFrames, convert_ppr_warp
    import torch

    hf = HostFrames(torch.arange(2 * 3 * 7.0).view(2, 3, 7))
    assert len(hf) == 2 and hf._host is None          # nothing copied until somebody looks
    assert hf[1].shape == (3, 7) and hf[1][0, 0] == 21.0 and np.stack(hf, 0).shape == (2, 3, 7)
    x = torch.arange(8.0)
    assert convert_ppr_warp(x).tolist() == [3, 4, 5, 0, 1, 2, 6, 7]


def test_traj_loss_fk_function_bookkeeping_with_a_fake_backend(monkeypatch):
    """The autograd plumbing of ForwardWarpTrajLoss / ForwardWarpTrajLossFK (16 inputs, 6 outputs, 16 gradient slots, which tensors are
    saved and handed to which backend call) with the device model replaced by a recorder on CPU tensors -- no kernel runs; the numerics
    are the GPU tests' business."""
    import torch
    from diffphys_amd import dp_model, hip_backend

    bs, T, F, nb, nq, nqd = 3, 5, 2, 4, 10, 9
    calls = {}

    class Fake:
        def __init__(self):
            self.nb, self.nq, self.nqd = nb, nq, nqd

        def rollout_forward_traj_loss(self, bs_, nsteps, dt, *inp, frame2step, target_pos, outseq, rot_ratio, want_seed_gt, fk=None):
            calls["fwd"] = dict(n_inp=len(inp), fk=None if fk is None else [tuple(x.shape) for x in fk], tgt=tuple(target_pos.shape), f2s=list(frame2step))
            z = lambda *s: torch.zeros(*s)
            tl = dict(reduced=torch.tensor([0.5, 1.0, 2.0, 0.0]), table=z(bs, F), scale=torch.full((bs, F), 0.25), seed_pos=z(F, bs * nb, 7),
                      seed_gt=torch.ones(bs, F, nb, 7) if want_seed_gt else None)
            if fk is not None:
                tl["fk_body_q"], tl["fk_body_qd"] = z(bs, F, nb, 7).requires_grad_(False), z(bs, F, nb, 6)
            return z(F, bs * nb, 7), z(F, bs * nb, 6), z(F, bs * nb, 6), z(F, bs * nb, 6), z(7), tl

        def rollout_backward_traj_loss(self, bs_, nsteps, dt, *args, fk=None):
            calls["bwd"] = dict(n_args=len(args), gl=float(args[-1]), fk=None if fk is None else [tuple(x.shape) for x in fk])
            g = dict(q_init=torch.ones(bs * nq), qd_init=torch.ones(bs * nqd), torques=torch.ones(T, bs * nqd), res_f=torch.ones(T, bs * nb, 6),
                     refs=torch.ones(T, bs * nqd), target_ke=torch.ones(bs * nqd), target_kd=torch.ones(bs * nqd), body_inv_mass=torch.ones(bs * nb),
                     body_inertia=torch.ones(bs * nb, 3, 3), body_inv_inertia=torch.ones(bs * nb, 3, 3))
            if fk is not None:
                g["fk_joint_q"], g["fk_joint_qd"] = torch.full((F, bs, nq), 2.0), torch.full((F, bs, nqd), 3.0)
            return g

    monkeypatch.setattr(hip_backend, "device_model", lambda env: Fake())

    class Host:
        pass

    h = Host()
    h.env, h.num_envs, h.steps_idx, h.frame2step, h.dt = object(), bs, range(T), [0, 4], 5e-4
    shapes = [(bs * nq,), (bs * nqd,), (T, bs * nqd), (T, bs * nb, 6), (T, bs * nqd), (bs * nqd,), (bs * nqd,), (bs * nb,), (bs * nb,), (bs * nb, 3, 3), (bs * nb, 3, 3)]
    args = [torch.zeros(*s, requires_grad=True) for s in shapes]
    tgt = torch.zeros(bs, F, nb, 7, requires_grad=True)
    outseq = torch.zeros(bs, F, dtype=torch.bool)
    qq, qqd = torch.zeros(F, bs, nq, requires_grad=True), torch.zeros(F, bs, nqd, requires_grad=True)
    loss, pos, vel, qp, qv, pid = dp_model.ForwardWarpTrajLossFK.apply(*args, tgt, outseq, qq, qqd, h)
    assert calls["fwd"] == dict(n_inp=10, fk=[(F, bs, nq), (F, bs, nqd)], tgt=(bs, F, nb, 7), f2s=[0, 4])
    assert float(loss.detach()) == 0.5 and not pos.requires_grad and not vel.requires_grad and qp.requires_grad and qv.requires_grad
    assert qp.shape == (bs, F, nb, 7) and qv.shape == (bs, F, nb, 6) and len(pid) == F and len(h.grfs) == 2
    (loss * 0.3 + qp.sum() * 2.0).backward()           # qv unused: its adjoint arrives as zeros
    assert calls["bwd"]["fk"] == [(F, bs, nq), (F, bs, nqd), (bs, F, nb, 7), (bs, F, nb, 6)] and abs(calls["bwd"]["gl"] - 0.3) < 1e-7
    assert all(a.grad is not None and a.grad.shape == a.shape for a in args)
    assert float(args[7].grad.abs().max()) == 0.0 and float(args[8].grad.min()) == 1.0      # body_mass: zero gradient; inv_mass: the backend's
    assert float(qq.grad.min()) == 2.0 and float(qqd.grad.min()) == 3.0
    assert torch.allclose(tgt.grad, torch.full_like(tgt, 0.25 * 0.3 / nb))                   # seed_gt x scale x g / nb
    # the plain form: no FK arguments reach the backend, 14 gradient slots
    for a in args + [tgt]:
        a.grad = None
    loss2, _, _ = dp_model.ForwardWarpTrajLoss.apply(*args, tgt, outseq, h)
    loss2.backward()
    assert calls["fwd"]["fk"] is None and calls["bwd"]["fk"] is None and all(a.grad is not None for a in args)
