#!/usr/bin/env python3
"""The reference's `phys_model.forward` TEXT, executed (build container only; grade-neutral like scripts/check_oracle_vs_reference_text.py):
/root/reference/diffphys/dp_model.py is imported UNCHANGED and its `forward` -- window and out-of-sequence bookkeeping, the mocap pipeline
(`get_mocap_data` -> `parse_amp` -> `bullet2gl`), `rotate_frame` / `rotate_frame_vel` / `compose_delta`, the five time-MLPs (the reference's
own modules), `rearrange_pred`, the init noise, `convert_ppr_warp` on the flat vectors, gains / masses / inertias, the loss assembly
(`se3_loss`, `reduce_loss`, weights) -- runs on the CPU with what cannot run here replaced by STAND-INS:

    ForwardWarp / ForwardKinematics (Warp)   ->  oracle/ref_torch.py (float64; the restatement the whole repo is held to)
    dqtorch's three quaternion kernels       ->  diffphys_amd.geom_utils (pytorch3d's conventions, pinned to scipy in the CPU tests)
    get_foot_height (posed URDF meshes)      ->  zeros (its weight, reg_foot_wt, is 0 in main.py's flags)
    warp / urdfpy / cv2 / trimesh            ->  import-time placeholders; the model object is built without `__init__` (which needs Warp's
                                                ModelBuilder and urdfpy) from the same template, gains and masses this package's phys_model uses

Output: tests/golden/ref_text_phys_model_forward.npz -- seeds, window starts, global_q, the loss terms the reference's text produced and, from
`loss.backward()` through the same stand-ins, the gradient of every parameter (norms; the small tensors in full).
tests/test_gpu_workload.py builds THIS package's phys_model under the same seeds on the GPU (its time-MLPs reproduce the reference's initial
weights bit for bit) and must reproduce them -- measured: loss terms to 1e-5, all 105 gradient tensors to 5e-5.  By the rules of this build stand-ins pin nothing ("parity unpinned" stays); what
this buys is that the ~500 lines of mirrored host plumbing -- incl. the harness quirks SURVEY N4 (i)-(iv) -- are compared with the
reference's text by a machine.

    python scripts/check_phys_model_vs_reference_text.py
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PPR_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
OUT = os.path.join(ROOT, "tests", "golden", "ref_text_phys_model_forward.npz")   # (--out FILE: tests/test_reference_text.py regenerates and compares)
if "--out" in sys.argv:
    OUT = sys.argv[sys.argv.index("--out") + 1]

CASES = [dict(seq="mi-pace", num_envs=3, frames_per_wdw=2, seed=11, frame_start=[0, 7, 30]),
         dict(seq="mi-trot", num_envs=4, frames_per_wdw=3, seed=12, frame_start=[2, 11, 29, 30]),
         dict(seq="mi-spin", num_envs=2, frames_per_wdw=2, seed=13, frame_start=[44, 5])]   # 44 + 1 = the clip's last frame
GLOBAL_Q = [0.0, 0.012, 0.0, 0.0, 0.0, 0.0, 1.0]
OPTS = dict(traj_wt=0.01, pos_state_wt=0.01, vel_state_wt=1e-4, pos_distill_wt=0.0, reg_torque_wt=0.0, reg_res_f_wt=0.0, reg_foot_wt=0.0, noise_std=2e-3)


def install_standins():
    from check_oracle_vs_reference_text import make_warp_standin
    from diffphys_amd import geom_utils as gu

    wp = make_warp_standin()
    wp.init = lambda: None
    art, mdl = types.ModuleType("warp.sim.articulation"), types.ModuleType("warp.sim.model")
    art.eval_fk = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("eval_fk is Warp's: replaced by the oracle in this check"))
    mdl.Mesh = object
    sim = types.ModuleType("warp.sim")
    for k, v in vars(wp.sim).items():
        setattr(sim, k, v)
    sim.articulation, sim.model = art, mdl
    wp.sim = sim
    sys.modules.update({"warp": wp, "warp.sim": sim, "warp.sim.articulation": art, "warp.sim.model": mdl})
    dq = types.ModuleType("dqtorch")
    dq.quaternion_to_matrix, dq.matrix_to_quaternion, dq.axis_angle_to_quaternion = gu.quaternion_to_matrix, gu.matrix_to_quaternion, gu.axis_angle_to_quaternion
    up = types.ModuleType("urdfpy")
    up.URDF = object
    sys.modules.update({"dqtorch": dq, "urdfpy": up})
    for name in ("cv2", "trimesh"):
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                sys.modules[name] = types.ModuleType(name)


def reference_model(rdm, rdl, tpl, case):
    """a phys_model of the reference without its __init__ (Warp's ModelBuilder, urdfpy): the attributes forward() reads, set the way
    __init__ / reinit_envs would (dp_model.py:56-250, 354-405) from the template this package compiles from the same URDF"""
    from oracle import ref_torch as rt

    m = object.__new__(rdm.phys_model)
    torch.nn.Module.__init__(m)
    here = os.getcwd()
    os.chdir(REF)
    try:
        loader = rdl.DataLoader({"seqname": case["seq"]})
    finally:
        os.chdir(here)
    m.opts = dict(OPTS)
    m.dt, m.noise_std, m.progress, m.device, m.in_bullet = 5e-4, OPTS["noise_std"], 0.0, "cpu", False
    m.preset_data(loader)                                                       # the reference's own
    m.n_dof, m.n_links = int(tpl["nq"]) - 7, int(tpl["nb"])
    kp, kd, nqd = float(tpl["kp"]), float(tpl["kd"]), int(tpl["nqd"])
    m.target_ke = torch.nn.Parameter(torch.tensor([0.0] * 6 + [kp] * (nqd - 6), dtype=torch.float32))
    m.target_kd = torch.nn.Parameter(torch.tensor([0.0] * 6 + [kd] * (nqd - 6), dtype=torch.float32))
    m.body_mass = torch.nn.Parameter(torch.tensor(tpl["body_mass"], dtype=torch.float32))
    m.norm_body_inertia = torch.tensor(tpl["body_inertia"], dtype=torch.float32)
    torch.manual_seed(case["seed"])
    m.add_nn_modules()                                                          # the reference's own: five TimeMLPWrappers, its order
    m.global_q = torch.nn.Parameter(torch.tensor(GLOBAL_Q, dtype=torch.float32))
    # reinit_envs (dp_model.py:354-366) without the Warp states
    m.num_envs, m.frames_per_wdw = case["num_envs"], case["frames_per_wdw"]
    m.steps_idx = range(m.steps_per_fr_interval * (m.frames_per_wdw - 1) + 1)
    m.steps_idx_fr = torch.LongTensor(list(m.steps_idx)) / m.steps_per_fr_interval
    m.frame2step = [i for i in range(len(m.steps_idx)) if i % m.steps_per_fr_interval == 0]
    m.env = types.SimpleNamespace(oracle=rt.Template(tpl, torch.float64))
    m.get_foot_height = lambda body_q: torch.zeros(body_q.shape[:2])           # (weight 0; the reference poses URDF meshes here)
    m.train()
    return m


def patch_boundary(rdm):
    """ForwardWarp / ForwardKinematics of the reference (its two Warp-backed autograd Functions) -> the float64 oracle, same signatures"""
    from oracle import ref_torch as rt

    class ForwardWarp:
        @staticmethod
        def apply(q_init, qd_init, torques, res_f, refs, target_ke, target_kd, body_mass, body_inv_mass, body_inertia, body_inv_inertia, self):
            d = lambda t: t.double()
            pos, vel, grf, jaf = rt.rollout(self.env.oracle, d(q_init), d(qd_init), d(torques), d(res_f), d(refs), d(target_ke), d(target_kd), d(body_mass),
                                            d(body_inv_mass), d(body_inertia), d(body_inv_inertia), len(self.steps_idx), list(self.frame2step), self.dt)
            self.grfs, self.jafs = list(grf.float()), list(jaf.float())
            self.sim_trajs = [p[: self.env.oracle.nb].detach().numpy() for p in pos]
            return pos.float(), vel.float()

    def fk_post(g):  # ForwardKinematics.backward's post-processing of the gradients it returns (dp_model.py:1109-1110, 1122-1123)
        g = torch.where(g.isnan(), torch.zeros_like(g), g)
        return torch.where(g > 1, torch.ones_like(g), g)

    class ForwardKinematics:
        @staticmethod
        def apply(rj_q, rj_qd, env):
            for t in (rj_q, rj_qd):
                if t.requires_grad:
                    t.register_hook(fk_post)
            bq, bqd = rt.fk_frames(env.oracle, rj_q.double(), rj_qd.double())
            return bq.float().contiguous(), bqd.float().contiguous(), [b.detach().numpy() for b in bq[0]]

    rdm.ForwardWarp, rdm.ForwardKinematics = ForwardWarp, ForwardKinematics


def main():
    install_standins()
    sys.path.insert(0, REF)
    import diffphys.dataloader as rdl
    import diffphys.dp_model as rdm
    from diffphys_amd import robots

    assert os.path.realpath(rdm.__file__).startswith(os.path.realpath(REF))
    patch_boundary(rdm)
    tpl = robots.load_template("laikago")
    out = {"note": np.asarray("loss terms of the reference's phys_model.forward TEXT (diffphys/dp_model.py:664-838, imported unchanged) over stand-ins: rollouts / FK by "
                              "oracle/ref_torch.py, dqtorch by diffphys_amd.geom_utils, foot height 0; generator scripts/check_phys_model_vs_reference_text.py; "
                              "a stand-in pins nothing; torch %s" % torch.__version__),
           "global_q": np.asarray(GLOBAL_Q, np.float64), "n_cases": np.int64(len(CASES))}
    for k, v in OPTS.items():
        out["opts/" + k] = np.float64(v)
    for i, case in enumerate(CASES):
        m = reference_model(rdm, rdl, tpl, case)
        np.random.seed(1000 + case["seed"])
        res = m.forward(frame_start=torch.tensor(case["frame_start"], dtype=torch.long))
        p = "case%d/" % i
        for k, v in case.items():
            out[p + k] = np.asarray(v)
        for k, v in res.items():
            out[p + k] = np.float64(float(v.detach()))
        # ... and loss.backward() (dp_model.py:840-841) through the same stand-ins: every parameter's gradient norm, the small ones in full
        res["total_loss"].backward()
        names = []
        for n_, q_ in m.named_parameters():
            g_ = q_.grad if q_.grad is not None else torch.zeros_like(q_)
            out[p + "gradnorm/" + n_] = np.float64(float(g_.double().norm()))
            names.append(n_)
            if g_.numel() <= 64:
                out[p + "grad/" + n_] = g_.detach().double().numpy()
        out[p + "param_names"] = np.asarray(names)
        out[p + "sim_env0"] = np.stack(m.sim_trajs, 0)
        out[p + "n_steps"] = np.int64(len(m.steps_idx))
        print("%s  %d envs x %d steps, %d frames:" % (case["seq"], case["num_envs"], len(m.steps_idx), case["frames_per_wdw"]),
              "  ".join("%s %.6e" % (k, float(v.detach())) for k, v in res.items()))
    # ---- the training loop of main.py:62-105 (progress, forward, backward, update = check_grad + AdamW + OneCycleLR) for a few iterations,
    # the reference's text throughout: add_optimizer / get_optimizable_param_list / get_lr_dict / update / check_grad (dp_model.py:407-520, 904-1000)
    case = CASES[0]
    m = reference_model(rdm, rdl, tpl, case)
    m.opts.update(phys_learning_rate=1e-4, num_rounds=5, iters_per_round=20)
    m.total_iters = 101                      # int(num_rounds * iters_per_round * ratio_phys_cycle) + warmup_iters + 1   (dp_model.py:60-66)
    m.grad_queue, m.model_cache, m.optimizer_cache, m.scheduler_cache = {}, [None, None], [None, None], [None, None]
    m.add_optimizer(m.opts)
    np.random.seed(2000 + case["seed"])
    K = 8
    starts = [[(3 * it + 5 * e) % (m.total_frames - m.frames_per_wdw) for e in range(m.num_envs)] for it in range(K)]
    hist = {k: [] for k in ("total_loss", "loss_traj", "loss_pos_state", "loss_vel_state")}
    for it in range(K):
        m.progress = it / (m.opts["num_rounds"] * m.opts["iters_per_round"])     # main.py:64
        ld = m.forward(frame_start=torch.tensor(starts[it], dtype=torch.long))
        m.backward(ld["total_loss"])
        m.update()
        for k in hist:
            hist[k].append(float(ld[k].detach()))
    out["train/frame_starts"] = np.asarray(starts)
    for k, v in hist.items():
        out["train/" + k] = np.asarray(v)
    out["train/lr_last"] = np.asarray(sorted(set(g["lr"] for g in m.optimizer.param_groups)))
    for n_ in ("global_q", "target_kd", "body_mass"):
        out["train/param/" + n_] = getattr(m, n_).detach().double().numpy()
    for n_, q_ in m.named_parameters():
        out["train/paramnorm/" + n_] = np.float64(float(q_.detach().double().norm()))
    print("training loop, %d iterations: total_loss" % K, " ".join("%.6e" % v for v in hist["total_loss"]))
    run_reference_main_text(rdm, rdl, tpl, out)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


MAIN_OPTS = dict(seqname="mi-pace", logname="reftext", logroot="/tmp/pprdp_reftext/", num_rounds=2, iters_per_round=3, accu_steps=1, phys_learning_rate=1e-4)
MAIN_CLIP_FRAMES = 5      # the mocap clip truncated to its first 5 frames: the evaluation pass (1 env over the WHOLE clip, main.py:77) is 133 steps
MAIN_TRAIN_SHAPE = (3, 2)  # stands where main.py:86 hard-codes reinit_envs(10, frames_per_wdw=24) -- 10 x 760 float64 autograd steps per iteration on the CPU otherwise
MAIN_SEED = 31


def run_reference_main_text(rdm, rdl, tpl, out):
    """/root/reference/main.py's `main()` TEXT executed: progress -> every iters_per_round iterations save_checkpoint, the
    evaluation pass (reinit_envs(1, total_frames, is_eval=True), forward(), query(), vis.show), reinit_envs for training -> forward, backward,
    update, write_log; 7 iterations with iters_per_round = 3, i.e. evaluation passes at 0, 3, 6.  Stand-ins: absl (flags -> MAIN_OPTS), the
    renderer (`diffphys.vis.PhysVisualizer`: records), `phys_model(opts, dataloader)` -> the model object built as above, its `reinit_envs`
    without Warp states and with the training shape MAIN_TRAIN_SHAPE, `query` -> {}.  Recorded per forward() call, in call order: the
    progress at the call, the scale of the init noise it drew, its window starts, its loss terms -- the evaluation passes' init noise reads
    `progress`, so the order of main.py:64 and :73-79 is visible in the numbers."""
    import importlib.util

    absl, app, flags = types.ModuleType("absl"), types.ModuleType("absl.app"), types.ModuleType("absl.flags")
    for n in ("DEFINE_integer", "DEFINE_string", "DEFINE_float", "DEFINE_bool", "DEFINE_boolean"):
        setattr(flags, n, lambda *a, **k: None)
    flags.FLAGS = types.SimpleNamespace(flag_values_dict=lambda: dict(MAIN_OPTS, **OPTS))
    app.run = lambda f: None
    absl.app, absl.flags = app, flags
    shown, logged = [], []

    class PhysVisualizer:
        def __init__(self, save_dir):
            self.save_dir = save_dir

        def show(self, it, data, fps=None):
            shown.append(int(it))

        def write_log(self, loss_dict, it):
            logged.append((int(it), {k: float(v.detach()) if torch.is_tensor(v) else float(v) for k, v in loss_dict.items()}))

    vis = types.ModuleType("diffphys.vis")
    vis.PhysVisualizer = PhysVisualizer
    sys.modules.update({"absl": absl, "absl.app": app, "absl.flags": flags, "diffphys.vis": vis})
    spec = importlib.util.spec_from_file_location("ref_main_text", os.path.join(REF, "main.py"))
    rmain = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rmain)

    case = dict(seq=MAIN_OPTS["seqname"], num_envs=MAIN_TRAIN_SHAPE[0], frames_per_wdw=MAIN_TRAIN_SHAPE[1], seed=MAIN_SEED)
    real_loader = rdl.DataLoader

    class TruncatedLoader(real_loader):
        def __init__(self, opts):
            real_loader.__init__(self, opts)
            self.amp_info = self.amp_info[:MAIN_CLIP_FRAMES]
            self.data_info["offset"] = np.asarray([0, MAIN_CLIP_FRAMES])

    rdl.DataLoader = TruncatedLoader
    try:
        m = reference_model(rdm, rdl, tpl, case)
    finally:
        rdl.DataLoader = real_loader
    assert m.total_frames == MAIN_CLIP_FRAMES
    m.opts.update(MAIN_OPTS)
    m.total_iters = int(MAIN_OPTS["num_rounds"] * MAIN_OPTS["iters_per_round"]) + 1   # dp_model.py:60-66 with ratio_phys_cycle 1, no warm-up
    m.save_dir = os.path.join(MAIN_OPTS["logroot"], "%s-%s" % (MAIN_OPTS["seqname"], MAIN_OPTS["logname"]))
    os.makedirs(m.save_dir, exist_ok=True)
    m.grad_queue, m.model_cache, m.optimizer_cache, m.scheduler_cache = {}, [None, None], [None, None], [None, None]
    m.add_optimizer(m.opts)

    def reinit_envs(num_envs, frames_per_wdw, is_eval=False, overwrite=False):   # dp_model.py:354-366 without the Warp states
        if not is_eval:
            num_envs, frames_per_wdw = MAIN_TRAIN_SHAPE
        m.num_envs, m.frames_per_wdw = num_envs, frames_per_wdw
        m.steps_idx = range(m.steps_per_fr_interval * (m.frames_per_wdw - 1) + 1)
        m.steps_idx_fr = torch.LongTensor(list(m.steps_idx)) / m.steps_per_fr_interval
        m.frame2step = [i for i in range(len(m.steps_idx)) if i % m.steps_per_fr_interval == 0]

    calls = []
    ref_forward, np_normal = m.forward, np.random.normal

    def forward(frame_start=None):
        rec = dict(progress=float(m.progress), num_envs=int(m.num_envs), scale=float("nan"))

        def normal(*a, **k):
            rec["scale"] = float(k["scale"])
            return np_normal(*a, **k)

        fs_of = m.compute_frame_start

        def compute_frame_start():
            rec["frame_start"] = fs_of()
            return rec["frame_start"]

        np.random.normal, m.compute_frame_start = normal, compute_frame_start
        try:
            res = ref_forward(frame_start)
        finally:
            np.random.normal = np_normal
            del m.compute_frame_start
        rec["losses"] = {k: float(v.detach()) for k, v in res.items()}
        calls.append(rec)
        return res

    m.reinit_envs, m.forward, m.query, m.cuda = reinit_envs, forward, (lambda: {}), (lambda: m)
    rmain.phys_model = lambda opts, dataloader: m
    rmain.DataLoader = lambda opts: None
    np.random.seed(3000 + MAIN_SEED)
    rmain.main(None)

    assert shown == [0, 3, 6] and [it for it, _ in logged] == list(range(m.total_iters)), (shown, [it for it, _ in logged])
    out["main/n_forward"] = np.int64(len(calls))
    out["main/clip_frames"], out["main/train_shape"], out["main/seed"] = np.int64(MAIN_CLIP_FRAMES), np.asarray(MAIN_TRAIN_SHAPE), np.int64(MAIN_SEED)
    for k in ("num_rounds", "iters_per_round"):
        out["main/" + k] = np.int64(MAIN_OPTS[k])
    out["main/progress"] = np.asarray([c["progress"] for c in calls])
    out["main/num_envs"] = np.asarray([c["num_envs"] for c in calls])
    out["main/noise_scale"] = np.asarray([c["scale"] for c in calls])
    out["main/frame_start"] = np.asarray([list(c["frame_start"].numpy()) + [-1] * (8 - c["num_envs"]) for c in calls])
    for k in ("total_loss", "loss_traj", "loss_pos_state", "loss_vel_state"):
        out["main/" + k] = np.asarray([c["losses"][k] for c in calls])
    out["main/logged_loss"] = np.asarray([d["loss"] for _, d in logged])
    out["main/lr_last"] = np.asarray(sorted(set(g["lr"] for g in m.optimizer.param_groups)))
    for n_, q_ in m.named_parameters():
        out["main/paramnorm/" + n_] = np.float64(float(q_.detach().double().norm()))
    print("reference main() text: %d forward() calls (evaluation passes shown at %s); progress / noise scale per call:" % (len(calls), shown))
    for c in calls:
        print("   envs %2d  progress %.4f  noise scale %.6e  total_loss %.6e" % (c["num_envs"], c["progress"], c["scale"], c["losses"]["total_loss"]))


if __name__ == "__main__":
    main()
