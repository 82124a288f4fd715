"""Host plumbing of the motion-imitation optimisation on top of the HIP boundary
(SURVEY.md section 8 row f3).  Mirrors ``phys_model`` of the reference
(/root/reference/diffphys/dp_model.py:56-1011): same attributes, method names,
tensor flow, loss terms, optimiser / scheduler, gradient guards and checkpoint
queue, so that ``main.py`` of the reference can drive it with the same flags.

Not reproduced (out of scope): rendering / mesh posing
(``query`` returns trajectories only; the foot-height regulariser uses the
ground-contact candidates instead of the visual meshes), lab4d coupling.

Harness quirks of the reference that are kept on purpose:
  (i)  ``convert_ppr_warp`` is applied to the FLAT qd_init vector, so only env 0's root twist is swapped;
  (ii) torques / res_f are reshaped [bs,T,.] -> [T,.] without the permute the other tensors get (both are zero);
  (iii) the model stays in train() mode during "eval" rollouts, so init noise is added there too;
  (iv) mocap rows (incl. quaternions) are interpolated linearly and not re-normalised.
"""
import os
from copy import deepcopy

import numpy as np
import scipy.interpolate
import torch
import torch.nn as nn

from . import robots, sim
from .dataloader import mocap_tensors, bullet2gl, parse_amp
from .dp_model import ForwardKinematics, ForwardWarp, ForwardWarpTrajLossFK, convert_ppr_warp
from .dp_utils import compose_delta, reduce_loss, reduce_loss_masked, rotate_frame, rotate_frame_vel, se3_loss
from .geom_utils import fid_reindex
from .grad_guard import GradHistory
from .time_mlp import TimeMLPWrapper, interp_wt, match_param_name


def get_local_rank():
    try:
        return int(os.environ["LOCAL_RANK"])
    except Exception:
        return 0


def _mean_sq(x):
    """x.pow(2).mean() as a dot product (rocBLAS): torch's multi-block full reduction of a tensor this size replays stale inside a captured
    HIP graph on this stack (scripts/micro/torch_graph_replay2.py; hip_backend.colsum has the column-sum case)."""
    v = x.reshape(-1)
    return torch.dot(v, v) / v.numel()


class _FootHeightHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, body_q, c_body, c_point, c_dist):
        from . import hip_backend

        body_q = body_q.detach().contiguous()
        h, arg = hip_backend.foot_height(body_q, c_body, c_point, c_dist)
        ctx.save_for_backward(body_q, c_body, c_point, arg)
        return h

    @staticmethod
    def backward(ctx, g):
        from . import hip_backend

        body_q, c_body, c_point, arg = ctx.saved_tensors
        return hip_backend.foot_height_vjp(body_q, c_body, c_point, arg, g.contiguous()), None, None, None


class phys_model(nn.Module):
    def __init__(self, opts, dataloader, dt=5e-4, device="cuda", urdf_root=None):
        super().__init__()
        self.opts = opts
        self.save_dir = os.path.join(opts["logroot"], "%s-%s" % (opts["seqname"], opts["logname"]))
        self.total_iters = int(opts["num_rounds"] * opts["iters_per_round"] * opts["ratio_phys_cycle"]) + opts["warmup_iters"] + 1
        self.progress = 0
        self.dt = dt
        self.noise_std = opts["noise_std"]
        self.device = device
        self.preset_data(dataloader)

        name = opts["urdf_template"]
        if name not in robots.PRESETS:
            raise NotImplementedError(name)
        self.in_bullet = False
        # articulation template: compiled from the URDF when its directory is given, else the committed npz
        if urdf_root is not None:
            env, _, info = robots.make_env(name, urdf_root, 1, device="cpu")
            self.template = env.template()
            self.template["kp"], self.template["kd"] = np.float32(info["kp"]), np.float32(info["kd"])
        else:
            self.template = robots.load_template(name)
        tpl = self.template
        self.joint_attach_ke, self.joint_attach_kd = float(tpl["joint_attach_ke"]), float(tpl["joint_attach_kd"])
        self.n_dof = int(tpl["nq"]) - 7
        self.n_links = int(tpl["nb"])
        kp, kd = float(tpl["kp"]), float(tpl["kd"])
        nqd = int(tpl["nqd"])
        self.target_ke = nn.Parameter(torch.tensor([0.0] * 6 + [kp] * (nqd - 6), dtype=torch.float32))
        self.target_kd = nn.Parameter(torch.tensor([0.0] * 6 + [kd] * (nqd - 6), dtype=torch.float32))
        self.body_mass = nn.Parameter(torch.tensor(tpl["body_mass"], dtype=torch.float32))
        self.register_buffer("norm_body_inertia", torch.tensor(tpl["body_inertia"], dtype=torch.float32))
        self.register_buffer("c_body", torch.tensor(tpl["contact_body"], dtype=torch.long), persistent=False)
        self.register_buffer("c_body_i32", torch.tensor(tpl["contact_body"], dtype=torch.int32), persistent=False)
        self.register_buffer("c_point", torch.tensor(tpl["contact_point"], dtype=torch.float32), persistent=False)
        self.register_buffer("c_dist", torch.tensor(tpl["contact_dist"], dtype=torch.float32), persistent=False)

        self.add_nn_modules()
        self.to(device)
        self.init_global_q()
        self.add_optimizer(opts)
        self.model_cache, self.optimizer_cache, self.scheduler_cache = [None, None], [None, None], [None, None]
        self.grad_history = GradHistory(queue_length=10, scale_threshold=5.0)

    # ------------------------------------------------------------------ setup
    def preset_data(self, dataloader):
        amp_info = dataloader.amp_info
        self.frame_offset_raw = dataloader.data_info["offset"]
        self.frame_interval = dataloader.frame_interval
        self.frame_info = None
        self.total_frames = len(amp_info)
        self.steps_per_fr_interval = int(self.frame_interval / self.dt)
        self.amp_info_func = scipy.interpolate.interp1d(np.arange(self.total_frames), amp_info, kind="linear",
                                                        fill_value="extrapolate", axis=0)
        self._amp_dev = {}  # device copies of the AMP table for get_mocap_tensors

    def add_nn_modules(self):
        n = self.total_frames
        self.root_pose_mlp = TimeMLPWrapper(n, out_channels=6, D=8, skips=[4], time_scale=0.1, output_scale=0.5)
        self.joint_angle_mlp = TimeMLPWrapper(n, out_channels=self.n_dof)
        self.vel_mlp = TimeMLPWrapper(n, out_channels=6 + self.n_dof, output_scale=5.0)
        self.torque_mlp = TimeMLPWrapper(n, out_channels=self.n_dof)
        self.residual_f_mlp = TimeMLPWrapper(n, out_channels=6 * self.n_links)

    def _make_env(self, num_envs):
        env = sim.Model.from_template(self.template, num_envs, self.device)
        env.ground = True
        env.joint_attach_ke, env.joint_attach_kd = self.joint_attach_ke, self.joint_attach_kd
        return env

    def init_global_q(self):
        self.frame2step = [0]
        self.num_envs = 1
        self.env = self._make_env(1)
        self.global_q = torch.tensor([0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0], device=self.device)
        steps_fr = torch.tensor([[0.0]], device=self.device)
        with torch.no_grad():
            _, _, queried_q, queried_qd, _, _ = self.get_batch_input(steps_fr)
            pos, _, _ = ForwardKinematics.apply(queried_q[:, None], queried_qd[:, None], self.env)
            foot_height = float(self.get_foot_height(pos)[0, 0])
        self.global_q = nn.Parameter(torch.tensor([0.0, -foot_height, 0.0, 0.0, 0.0, 0.0, 1.0], dtype=torch.float32, device=self.device))

    def set_progress(self, num_iters):
        self.progress = num_iters / self.total_iters
        if "reg_cam_prior_wt" in self.opts:
            self.set_loss_weight("reg_cam_prior_wt", (0, 0.5), (1, 0), self.progress)

    def set_loss_weight(self, loss_name, anchor_x, anchor_y, current_steps, type="linear"):
        if "%s_init" % loss_name not in self.opts:
            self.opts["%s_init" % loss_name] = self.opts[loss_name]
        self.opts[loss_name] = self.opts["%s_init" % loss_name] * interp_wt(anchor_x, anchor_y, current_steps, type=type)

    def reinit_envs(self, num_envs, frames_per_wdw, is_eval=False, overwrite=False):
        self.num_envs, self.frames_per_wdw = num_envs, frames_per_wdw
        self.steps_idx = range(self.steps_per_fr_interval * (frames_per_wdw - 1) + 1)
        self.frame2step = [i for i in range(len(self.steps_idx)) if i % self.steps_per_fr_interval == 0]
        # device tensors of this window shape, made once per shape: main.py alternates between the evaluation and the training shape, and a
        # captured iteration (capture_iteration) reads them BY ADDRESS -- they must neither move nor be freed
        cache = self.__dict__.setdefault("_wdw_cache", {})
        if frames_per_wdw not in cache:
            cache[frames_per_wdw] = (torch.tensor(list(self.steps_idx), device=self.device) / self.steps_per_fr_interval,
                                     torch.tensor(self.frame2step, dtype=torch.long, device=self.device))
        self.steps_idx_fr, self._f2s_t = cache[frames_per_wdw]  # (_f2s_t: device copy of frame2step for indexing, no upload per use)
        env_name = "eval_env" if is_eval else "train_env"
        if hasattr(self, env_name) and not overwrite and getattr(self, env_name).num_envs == num_envs:
            self.env = getattr(self, env_name)
        else:
            self.env = self._make_env(num_envs)  # the trajectory store lives in the autograd ctx, not in per-step States
            setattr(self, env_name, self.env)

    # -------------------------------------------------------------- optimiser
    def get_lr_dict(self):
        lr_base = self.opts["phys_learning_rate"]
        lr_explicit = lr_base * 10
        startwith = {k: lr_explicit for k in ("global_q", "target_ke", "target_kd", "attach_ke", "attach_kd", "body_mass")}
        startwith.update({k: lr_base for k in ("root_pose_mlp", "joint_angle_mlp", "vel_mlp", "torque_mlp", "residual_f_mlp")})
        return startwith, {"root_pose_mlp.base_quat": lr_explicit}

    def get_optimizable_param_list(self):
        startwith, with_ = self.get_lr_dict()
        refs, params, lrs = [], [], []
        for name, p in self.named_parameters():
            m_loose, lr_loose = match_param_name(name, with_, type="with")
            m_strict, lr_strict = match_param_name(name, startwith, type="startwith")
            lr = lr_loose if m_loose > 0 else (lr_strict if m_strict > 0 else 0.0)
            if lr > 0:
                refs.append({name: p})
                params.append({"params": p})
                lrs.append(lr)
        return refs, params, lrs

    def add_optimizer(self, opts):
        self.params_ref_list, params_list, lr_list = self.get_optimizable_param_list()
        # same AdamW as the reference (dp_model.py:914-918); on the GPU the fused implementation (one launch for all
        # parameters instead of a few per parameter, identical update rule) where this torch build has it
        kw = {}
        if torch.cuda.is_available() and str(self.device).startswith("cuda"):
            try:
                torch.optim.AdamW([torch.zeros(1, device=self.device, requires_grad=True)], fused=True)
                kw["fused"] = True
            except Exception:
                kw = {}
        # the reference makes one parameter group per parameter; AdamW treats every parameter independently, so grouping the
        # parameters by learning rate is the same update with 2 fused launches instead of 80 (params_ref_list keeps the names)
        by_lr = {}
        for g, lr in zip(params_list, lr_list):
            by_lr.setdefault(lr, []).append(g["params"])
        params_list, lr_list = [{"params": ps} for ps in by_lr.values()], list(by_lr.keys())
        self.optimizer = torch.optim.AdamW(params_list, lr=opts["phys_learning_rate"], weight_decay=1e-4, **kw)
        total_iters = max(2, self.total_iters)
        self.scheduler = torch.optim.lr_scheduler.OneCycleLR(
            self.optimizer, lr_list, total_iters, pct_start=2.0 / total_iters, cycle_momentum=False,
            anneal_strategy="linear", final_div_factor=1e2, div_factor=25)

    def update(self):
        grad_dict = self.check_grad()  # also raises on the NaN loss deferred from forward()
        self.optimizer.step()
        self.scheduler.step()
        self.optimizer.zero_grad()
        return grad_dict

    def check_grad(self, thresh=10.0):
        """global-norm guard with rollback to the model cached two rounds ago, then per-parameter median-based clipping
        (dp_model.py:936-1000).  Same decisions and values as the reference's per-parameter loop (grad_guard.GradHistory;
        tests/test_host_plumbing.py compares them), with ONE host transfer per iteration -- the deferred NaN check of the
        loss, the global norm and "is any parameter an outlier" travel together -- instead of one per parameter."""
        named = [(name, p) for d in self.params_ref_list for name, p in d.items() if p.requires_grad and p.grad is not None]
        # a 0-dim bool tensor: some forward() since the last check produced a NaN total_loss (accumulated over the accu_steps
        # forward() calls of one update; never set by eval / no_grad forwards) -- read and cleared here
        pending, self._pending_nan = getattr(self, "_pending_nan", None), None
        if not named:
            if pending is not None and bool(pending):
                raise FloatingPointError("total_loss is NaN")
            return {}
        params_list = [p for _, p in named]
        grad_norm = torch.nn.utils.clip_grad_norm_(params_list, thresh)
        plan = self.grad_history.plan([n for n, _ in named], [p.grad for p in params_list])  # norms AFTER the global clip, like the reference
        zero = grad_norm.new_zeros(())
        host = torch.stack([pending.to(grad_norm.dtype) if pending is not None else zero, grad_norm,
                            plan["any_outlier"].to(grad_norm.dtype) if plan["any_outlier"] is not None else zero]).tolist()
        if host[0] != 0:
            raise FloatingPointError("total_loss is NaN")  # deferred from forward(), see there
        if not (host[1] <= thresh):  # too large OR not finite (NaN > thresh is False: a NaN norm must not reach the optimiser)
            self.optimizer.zero_grad()
            self._rezero_grad_bufs()   # clip_grad_norm_ with a NaN norm has just written 0 * NaN into the cached zero gradients
            if get_local_rank() == 0:
                print("large grad: %.2f, clear gradients" % host[1])
            if self.model_cache[0] is not None:
                self.load_state_dict(self.model_cache[0])
                self.optimizer.load_state_dict(self.optimizer_cache[0])
                self.scheduler.load_state_dict(self.scheduler_cache[0])
            return {}
        flags = self.grad_history.commit(plan, host[2] != 0)
        grad_dict = {"grad/" + name: g for (name, _), g in zip(named, plan["norms"].unbind())}
        if plan["med"] is not None:
            for (name, _), m, has in zip(named, plan["med"].unbind(), plan["has_med"]):
                if has:
                    grad_dict["grad_med/" + name] = m
        if get_local_rank() == 0 and any(flags):
            host_norms = plan["norms"].tolist()   # (one transfer for all the flagged parameters' messages)
            for (name, _), f, nrm in zip(named, flags, host_norms):
                if f:
                    print("large grad: %.2f, clear %s" % (nrm, name))
        return grad_dict

    def clear_grad(self):
        self.optimizer.zero_grad()
        if self.model_cache[0] is not None:
            self.load_state_dict(self.model_cache[0])
            self.optimizer.load_state_dict(self.optimizer_cache[0])
            self.scheduler.load_state_dict(self.scheduler_cache[0])

    def save_checkpoint(self, steps_count):
        for c in (self.model_cache, self.optimizer_cache, self.scheduler_cache):
            c[0] = c[1]
        self.model_cache[1] = deepcopy(self.state_dict())
        self.optimizer_cache[1] = deepcopy(self.optimizer.state_dict())
        self.scheduler_cache[1] = deepcopy(self.scheduler.state_dict())
        if get_local_rank() == 0:
            os.makedirs(self.save_dir, exist_ok=True)
            path = "%s/ckpt_phys_%04d.pth" % (self.save_dir, steps_count)
            torch.save(self.model_cache[1], path)
            torch.save(self.model_cache[1], "%s/ckpt_phys_latest.pth" % self.save_dir)

    def load_checkpoint(self, model_path):
        self.load_state_dict(torch.load(model_path, map_location="cpu"), strict=False)

    # ------------------------------------------------------------ data query
    def get_mocap_data(self, steps_fr):
        msm = parse_amp(self.amp_info_func(steps_fr))
        msm = {k: np.array(v, copy=True) for k, v in msm.items()}
        bullet2gl(msm, self.in_bullet)
        return msm

    def get_mocap_tensors(self, steps_fr):
        """get_mocap_data on the device of ``steps_fr``: the same linear inter/extrapolation of the AMP table (float64, like
        scipy's interp1d), column map and axis change, without the round trip through numpy and six uploads per iteration.
        Returns float32 tensors.  (The in_bullet variant goes through scipy's Rotation and keeps the numpy path.)"""
        if self.in_bullet:
            msm = self.get_mocap_data(steps_fr.detach().cpu().numpy())
            return {k: torch.tensor(v, dtype=torch.float32, device=steps_fr.device) for k, v in msm.items()}
        dev = steps_fr.device
        tab = self._amp_dev.get(dev)
        if tab is None:
            tab = self._amp_dev[dev] = torch.as_tensor(np.asarray(self.amp_info_func.y, dtype=np.float64), device=dev)
        return mocap_tensors(tab, steps_fr)

    def get_net_pred(self, steps_fr):
        bs, nstep = steps_fr.shape
        t = steps_fr.reshape(-1)
        if getattr(self, "skip_zeroed_mlps", True):
            # The reference evaluates torque_mlp and residual_f_mlp and then multiplies both outputs by zero (`torques *= 0`, `res_f *= 0`,
            # dp_model.py:526-536): the rollout sees zeros and every parameter of the two MLPs gets an exactly-zero gradient (AdamW then
            # only applies its weight decay to them).  Same values without the two MLPs' forward and backward -- 2 of the 5 time-MLPs,
            # ~40 % of an iteration's GEMMs: zeros here, zero gradients attached in backward() / iteration() (_attach_zero_grads).
            # skip_zeroed_mlps = False evaluates them as the reference does (tests compare the two bit for bit).
            torques = torch.zeros(bs, nstep, 6 + self.n_dof, dtype=torch.float32, device=steps_fr.device)
            res_f = torch.zeros(bs, nstep, 6 * self.n_links, dtype=torch.float32, device=steps_fr.device)
        else:
            shared = {}
            torques = self.torque_mlp(t, shared)
            torques = torch.cat([torch.zeros_like(torques[:, :1].repeat(1, 6)), torques], 1).view(bs, nstep, -1) * 0
            res_f = self.residual_f_mlp(t, shared).view(bs, nstep, -1, 6)
            res_f = torch.cat([res_f[..., :3] * 10, res_f[..., 3:]], -1).view(bs, nstep, -1) * 0
        shared = {}   # frame -> time mapping and Fourier features, computed once for the MLPs that agree on them (time_mlp.TimeEmbedding)
        delta_root = self.root_pose_mlp(t, shared).view(bs, nstep, -1)
        delta_ja_ref = self.joint_angle_mlp(t, shared).view(bs, nstep, -1)
        state_qd = self.vel_mlp(t, shared).view(bs, nstep, -1)
        return torques, delta_root, delta_ja_ref, state_qd, res_f

    def _attach_zero_grads(self):
        """the exactly-zero gradients of the two MLPs whose outputs the reference multiplies by zero (get_net_pred), from buffers made once"""
        if not getattr(self, "skip_zeroed_mlps", True):
            return
        z = self.__dict__.setdefault("_zero_grad_bufs", {})
        for mlp in (self.torque_mlp, self.residual_f_mlp):
            for p in mlp.parameters():
                if p.requires_grad and p.grad is None:
                    b = z.get(p)
                    if b is None or b.shape != p.shape or b.device != p.device:
                        b = z[p] = torch.zeros_like(p)
                    p.grad = b   # (update()'s zero_grad(set_to_none=True) detaches it again; the guards scale it in place: still zero)

    def _rezero_grad_bufs(self):
        """The cached zero gradients are SHARED across iterations, and the guards scale gradients in place: a finite factor leaves zeros
        zero, a NaN clip coefficient (some other gradient was NaN) does not -- and zero_grad() only detaches the buffer.  Called on the
        guard's clear / roll-back path, so that the NaN iteration cannot poison every later global norm."""
        bufs = list(self.__dict__.get("_zero_grad_bufs", {}).values())
        if bufs:
            torch._foreach_zero_(bufs)

    @staticmethod
    def rearrange_pred(queried_q, queried_ja, queried_qd, torques, res_f):
        bs, nstep, _ = queried_q.shape
        queried_q = torch.cat([queried_q, queried_ja], -1).permute(1, 0, 2).reshape(nstep, -1)
        queried_qd = queried_qd.permute(1, 0, 2).reshape(nstep, -1)
        ref_ja = torch.cat([torch.zeros_like(queried_ja[..., :1].repeat(1, 1, 6)), queried_ja], -1).permute(1, 0, 2).reshape(nstep, -1)
        return ref_ja, queried_q, queried_qd, torques.reshape(nstep, -1), res_f.reshape(nstep, -1, 6)  # quirk (ii)

    def get_foot_height(self, state_body_q):
        """lowest ground-contact candidate per (env, frame); the reference poses the visual meshes instead (dp_model.py:574-579).
        float32 GPU poses: one HIP launch (``pd_foot_height``) and one for the gradient, which reaches the body of the
        lowest candidate only -- the torch composition's gather over 3 838 candidates has a 3.3 ms index_put backward."""
        if not (state_body_q.is_cuda and state_body_q.dtype == torch.float32):
            raise TypeError("get_foot_height needs float32 GPU poses (the product path has no CPU fallback; oracle/pose_torch.py foot_height is the checker)")
        return _FootHeightHip.apply(state_body_q, self.c_body_i32, self.c_point, self.c_dist)

    def compute_frame_start(self, rng=np.random):
        fs = torch.tensor(rng.rand(self.num_envs), device=self.device)
        return (fs * (self.total_frames - self.frames_per_wdw)).round().long()

    def fk_pos_vel(self, target_q, target_ja, target_qd, target_jad):
        q = torch.cat([target_q, target_ja], -1).permute(1, 0, 2).contiguous()
        qd = convert_ppr_warp(torch.cat([target_qd, target_jad], -1).permute(1, 0, 2).contiguous())
        body_q, body_qd, msm = ForwardKinematics.apply(q, qd, self.env)
        return body_q, convert_ppr_warp(body_qd), msm

    def get_batch_input(self, steps_fr):
        device = steps_fr.device
        msm = self.get_mocap_tensors(steps_fr)
        t = lambda k: msm[k]
        target_ja, target_jad = t("jang"), t("jvel")
        target_q = torch.cat([t("pos"), t("orn")], -1)
        target_qd = torch.cat([t("vel"), t("avel")], -1)
        target_q = rotate_frame(self.global_q, target_q)
        target_qd = rotate_frame_vel(self.global_q, target_qd)
        fr = lambda x: self._frames_of(x, 1)
        target_position, target_velocity, self.target_trajs = self.fk_pos_vel(fr(target_q), fr(target_ja), fr(target_qd), fr(target_jad))
        torques, delta_q, delta_ja, queried_qd, res_f = self.get_net_pred(steps_fr)
        queried_q = compose_delta(target_q, delta_q)
        queried_ja = target_ja + delta_ja
        ref_ja, queried_q, queried_qd, torques, res_f = self.rearrange_pred(queried_q, queried_ja, queried_qd, torques, res_f)
        return target_position, ref_ja, queried_q, queried_qd, torques, res_f

    # ---------------------------------------------------------------- forward
    def make_q_init_noise(self, rng=np.random):
        """The init noise of this iteration (dp_model.py:702-712): N(0, noise_std * ratio), none on the root translation, x5 on
        the root rotation; None when the model is not training or the noise is off.  A separate method so that a captured
        iteration (capture_iteration) can feed it through a static buffer."""
        if not (self.training and self.noise_std > 0):
            return None
        noise_ratio = np.clip(1 - 1.5 * self.progress, 0, 1)
        nq = self.n_dof + 7
        noise = torch.tensor(rng.normal(size=self.num_envs * nq, scale=self.noise_std * noise_ratio), dtype=torch.float32,
                             device=self.device).view(self.num_envs, -1)
        noise[:, :3] = 0
        noise[:, 3:7] *= 5
        return noise.reshape(-1)

    def forward(self, frame_start=None, q_init_noise=None):
        """``q_init_noise``: the tensor make_q_init_noise() would draw (a captured iteration passes its static buffer)."""
        frame_start = self.compute_frame_start() if frame_start is None else frame_start[: self.num_envs]
        steps_fr = frame_start[:, None] + self.steps_idx_fr[None]
        if len(self.frame_offset_raw) - 1 == 1:
            # one video: no frame of a window lies in another video than its first -- the mask the reference computes (fid_reindex, then
            # vidid[:, :1] != vidid; dp_model.py:673-676) is all False; a cached constant instead of ~15 launches per forward()
            key = (int(steps_fr.shape[0]), self.frames_per_wdw, str(steps_fr.device))
            cache = self.__dict__.setdefault("_no_outseq", {})
            outseq_idx = cache.get(key)
            if outseq_idx is None:
                if len(cache) > 8:
                    cache.clear()
                outseq_idx = cache[key] = torch.zeros(self._frames_of(steps_fr, 1).shape, dtype=torch.bool, device=steps_fr.device)
        else:
            vidid, _ = fid_reindex(self._frames_of(steps_fr, 1), len(self.frame_offset_raw) - 1, self.frame_offset_raw)
            outseq_idx = (vidid[:, :1] - vidid) != 0
        target_position, ref_ja, queried_q, queried_qd, torques, res_f = self.get_batch_input(steps_fr)

        res_fin = res_f.clone()
        q_init = queried_q[0].reshape(-1)  # a VIEW of queried_q, as in the reference
        qd_init = queried_qd[0]
        if q_init_noise is None:
            q_init_noise = self.make_q_init_noise()  # quirk (iii): also during "eval" rollouts
        if q_init_noise is not None:
            # in place, like the reference's `q_init += q_init_noise` (dp_model.py:712): the noise therefore also enters
            # queried_q[frame2step][0], i.e. the control-reference FK and the pos_state loss of frame 0
            q_init += q_init_noise
        n = self.num_envs
        target_ke = self.target_ke[None].repeat(n, 1).view(-1)
        target_kd = self.target_kd[None].repeat(n, 1).view(-1)
        body_mass = self.body_mass[None].repeat(n, 1).view(-1)
        body_inv_mass = 1.0 / body_mass
        body_inertia = self.norm_body_inertia[None].repeat(n, 1, 1, 1).view(-1, 3, 3) * body_mass[..., None, None]
        # body_inertia.inverse() of the reference (dp_model.py:730) as inverse(norm_inertia) / m: inverse(N m) = inverse(N) / m exactly, the
        # constant factor inverted once (LU, eagerly) -- torch's batched LU is a host synchronisation per forward() with its singular-matrix
        # check and is not capturable in a HIP graph; the eager and the captured iteration run this one formula (bit-identical losses)
        body_inv_inertia = (self._inv_norm_inertia()[None].repeat(n, 1, 1, 1).view(-1, 3, 3) / body_mass[..., None, None]).contiguous()
        qd_init = convert_ppr_warp(qd_init)  # quirk (i): flat vector
        res_fin = convert_ppr_warp(res_fin)
        F_ = self.frames_per_wdw
        # loss_traj is the ONE term that back-propagates through the rollout (dp_model.py:777-779; the others use sim_position.detach()).
        # fuse_traj_loss (default): the rollout evaluates it where the frame poses are produced and the adjoint seeds itself
        # (dp_model.ForwardWarpTrajLoss, C ABI pd_rollout_*_traj_loss); False: the reference's sequence
        # ForwardWarp -> se3_loss -> reduce_loss through torch (kept: tests compare the two)
        fused = bool(getattr(self, "fuse_traj_loss", True)) and q_init.is_cuda
        queried_q = self._frames_of(queried_q, 0).reshape(F_, n, -1)
        queried_qd = convert_ppr_warp(self._frames_of(queried_qd, 0).reshape(F_, n, -1))
        if fused:
            # ... and the FK of the control reference (dp_model.py:758) rides on the two small launches of that path: no FK launch,
            # no permuted copies (dp_model.ForwardWarpTrajLossFK, C ABI pd_rollout_*_traj_loss_fk)
            loss_traj_fused, sim_position, sim_velocity, queried_position, queried_velocity, self.pid_ref = ForwardWarpTrajLossFK.apply(
                q_init, qd_init, torques, res_fin, ref_ja, target_ke, target_kd, body_mass, body_inv_mass, body_inertia, body_inv_inertia,
                target_position.reshape(n, F_, -1, 7), outseq_idx, queried_q, queried_qd, self)
        else:
            sim_position, sim_velocity = ForwardWarp.apply(q_init, qd_init, torques, res_fin, ref_ja, target_ke, target_kd, body_mass,
                                                           body_inv_mass, body_inertia, body_inv_inertia, self)
            queried_position, queried_velocity, self.pid_ref = ForwardKinematics.apply(queried_q, queried_qd, self.env)
        sim_velocity = convert_ppr_warp(sim_velocity)
        queried_velocity = convert_ppr_warp(queried_velocity)
        foot_height = self.get_foot_height(queried_position)

        target_position = target_position.reshape(n, F_, -1, 7)
        sim_position = sim_position.reshape(F_, n, -1, 7).permute(1, 0, 2, 3)
        sim_velocity = sim_velocity.reshape(F_, n, -1, 6).permute(1, 0, 2, 3)

        loss_dict = {}
        if fused:
            loss_dict["traj"] = loss_traj_fused
        else:
            loss_traj = se3_loss(sim_position, target_position).mean(-1)
            loss_traj = torch.where(outseq_idx, torch.zeros_like(loss_traj), loss_traj)
            loss_dict["traj"] = reduce_loss(loss_traj, clip=True)
        loss_pos = se3_loss(queried_position, sim_position.detach()).mean(-1)
        loss_dict["pos_state"] = reduce_loss_masked(loss_pos, outseq_idx)
        loss_vel = se3_loss(queried_velocity, sim_velocity.detach()).mean(-1)
        loss_dict["vel_state"] = reduce_loss_masked(loss_vel, outseq_idx)
        loss_dict["reg_torque"] = _mean_sq(torques)
        loss_dict["reg_res_f"] = _mean_sq(res_f)
        loss_dict["reg_foot"] = foot_height.pow(2).mean()

        total_loss = 0
        for k, v in loss_dict.items():
            total_loss = total_loss + v * self.opts[k + "_wt"]
        out = {"loss_" + k: v for k, v in loss_dict.items()}
        # the reference drops into pdb on a NaN loss right here (dp_model.py:832): one host synchronisation per forward().
        # The check is kept but deferred to update(), which has to talk to the host anyway (gradient-norm guard), so that
        # nothing between forward() and backward() waits for the device
        # nothing between forward() and backward() waits for the device.  The flag ACCUMULATES: main.py calls forward() accu_steps
        # times per update() and sums the losses, a NaN in any of those windows must stop the update, not only one in the last.
        # Evaluation forwards (no_grad, or the module in eval mode) leave no flag behind for the next training update.
        if torch.is_grad_enabled() and self.training:
            nan_now = total_loss.detach().isnan()
            prev = getattr(self, "_pending_nan", None)
            self._pending_nan = nan_now if prev is None else (prev | nan_now)
        out["total_loss"] = total_loss
        return out

    def _frames_of(self, x, dim):
        """x[frame2step] along `dim`.  frame2step is every steps_per_fr_interval-th step (reinit_envs), so this is a strided
        VIEW whose backward is a strided copy; the reference's list indexing (dp_model.py:621-626, 747-752) has an
        index_put backward that costs ~1 ms per use on the GPU (3 ms per iteration)."""
        f2s = list(self.frame2step)
        k = self.steps_per_fr_interval
        if f2s == list(range(0, x.shape[dim], k))[: len(f2s)] and len(f2s) == len(range(0, x.shape[dim], k)):
            return x[(slice(None),) * dim + (slice(None, None, k),)]
        return x.index_select(dim, self._frame_index())

    def _frame_index(self):
        t = getattr(self, "_f2s_t", None)
        if t is None or t.numel() != len(self.frame2step):  # frame2step set by hand (init_global_q, tests)
            t = self._f2s_t = torch.tensor(list(self.frame2step), dtype=torch.long, device=self.device)
        return t

    def _inv_norm_inertia(self):
        """inverse(norm_body_inertia), made once per VALUE of that buffer: keyed on its storage, version counter and device, so that
        load_state_dict / load_checkpoint (an in-place copy: the version moves) and .to(device) (another storage) invalidate it."""
        nbi = self.norm_body_inertia
        key = (nbi.data_ptr(), nbi._version, str(nbi.device))
        cache = getattr(self, "_inv_norm_cache", None)
        if cache is None or getattr(self, "_inv_norm_key", None) != key:
            inv = nbi.inverse().contiguous()
            if cache is not None and cache.device == inv.device and cache.shape == inv.shape:
                cache.copy_(inv)   # in place: a captured iteration reads this tensor BY ADDRESS (iteration() re-checks the key before a replay)
            else:
                self._inv_norm_cache = inv
            self._inv_norm_key = key
        return self._inv_norm_cache

    def backward(self, loss):
        loss.backward()
        self._attach_zero_grads()

    # ------------------------------------------------------- one iteration as ONE HIP graph
    # main.py:96-103 of the reference at accu_steps = 1 is  forward() -> backward() -> update().  forward() + backward() are ~620 of
    # the iteration's ~650 launches (five time-MLPs, pose algebra, the two rollout launches, the loss terms and autograd's backward of
    # all that) on 10 envs x 760 steps: host-bound at 16 ms.  Nothing in them talks to the host (rounds 2-4), so they are captured once
    # and replayed; update() -- the gradient guards with their ONE host transfer, AdamW, the LR schedule -- stays as it is, eager,
    # ~25 launches.  The window starts and the init noise are drawn on the host exactly as before (same random stream as the eager
    # path) and reach the graph through two static buffers.
    def capture_iteration(self, validate=True, verbose=False):
        """Captures forward() + backward() at the current window shape (reinit_envs) into a HIP graph.  Returns True when the graph is in
        use afterwards.  ``validate``: two replays on fresh window starts / noise are compared BIT FOR BIT (total loss, every loss term,
        every parameter gradient) with eager forward() + backward() on the same inputs; on any difference the graph is dropped and
        iteration() stays eager (torch's own multi-block reductions replay stale on this stack: scripts/micro/torch_graph_replay2.py --
        everything on this path goes through GEMMs / the library's kernels instead, and this check is what holds that)."""
        self._graph = None
        self._graph_wanted = True   # iteration() captures again (a few times at most) when the window shape, the env or a loss weight changes
        self._graph_captures = getattr(self, "_graph_captures", 0) + 1
        if not (self.training and str(self.device).startswith("cuda") and torch.cuda.is_available()):
            return False
        n, nq = self.num_envs, self.n_dof + 7
        dev = self.device
        g_fs = torch.zeros(n, dtype=torch.long, device=dev)
        g_noise = torch.zeros(n * nq, dtype=torch.float32, device=dev)
        self._inv_norm_inertia(); self._frame_index()
        rng = np.random.RandomState(20261003)   # (a private stream: capturing must not move the run's own random numbers)
        keep_pending = getattr(self, "_pending_nan", None)
        params = [p for p in self.parameters() if p.requires_grad]
        weights = {k: v for k, v in self.opts.items() if k.endswith("_wt")}

        def draw():
            g_fs.copy_(self.compute_frame_start(rng))
            noise = self.make_q_init_noise(rng)
            g_noise.copy_(noise) if noise is not None else g_noise.zero_()

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        mt = torch.autograd.is_multithreading_enabled()
        torch.autograd.set_multithreading_enabled(False)   # (the capture must see every backward launch on the capturing stream)
        quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if quiet is not None:  # the parameters' AccumulateGrad nodes were made on the default stream (eager iterations, the constructor's FK):
            quiet(False)       # intended here -- the warm-up below runs them on the capturing stream before anything is captured
        ok, st, failure = True, None, None
        try:
            with torch.cuda.stream(side):
                for _ in range(3):   # warm-up on the capturing stream: allocator, frame tables of the rollout, rocBLAS handles
                    draw()
                    self.optimizer.zero_grad(set_to_none=True)
                    self._pending_nan = None
                    self.forward(frame_start=g_fs, q_init_noise=g_noise)["total_loss"].backward()
                self.optimizer.zero_grad(set_to_none=True)
                self._pending_nan = None
                side.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    out = self.forward(frame_start=g_fs, q_init_noise=g_noise)
                    out["total_loss"].backward()
            torch.cuda.current_stream().wait_stream(side)
            st = dict(graph=graph, fs=g_fs, noise=g_noise, out=out, nan=self._pending_nan, grads=[(p, p.grad) for p in params if p.grad is not None],
                      side=dict(grfs=self.grfs, jafs=self.jafs, sim=self.sim_trajs._dev, tgt=self.target_trajs._dev, pid=self.pid_ref._dev,
                                info=self.traj_loss_info), shape=(n, self.frames_per_wdw), weights=weights, env=self.env, replays=0)
            if validate:
                for trial in range(2):
                    draw()
                    graph.replay()
                    got = {k: v.detach().clone() for k, v in out.items()}
                    got_g = [g.clone() for _, g in st["grads"]]
                    for p in params:
                        p.grad = None
                    self._pending_nan = None
                    ref = self.forward(frame_start=g_fs.clone(), q_init_noise=g_noise.clone())
                    ref["total_loss"].backward()
                    bad = [k for k in got if not torch.equal(got[k], ref[k].detach())]
                    names = {id(p): nme for nme, p in self.named_parameters()}
                    bad += [names[id(p)] for (p, _), g in zip(st["grads"], got_g) if p.grad is None or not torch.equal(g, p.grad)]
                    if verbose or bad:
                        print("capture_iteration: replay %d vs eager: %s" % (trial, "bit-identical (%d loss terms, %d gradients)" % (len(got), len(got_g)) if not bad
                                                                             else "DIFFERENT in %s" % bad[:12]))
                    ok = ok and not bad
        except Exception as e:   # an op that cannot be captured, the graph pool out of memory, an allocator error: stay eager, do not abort training
            ok, st, failure = False, None, "%s: %s" % (type(e).__name__, e)
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
        finally:
            torch.autograd.set_multithreading_enabled(mt)
            if quiet is not None:
                quiet(True)   # (torch's default)
        for p in params:
            p.grad = None
        self._pending_nan = keep_pending
        if ok:
            self._graph = st
        elif get_local_rank() == 0:
            print("capture_iteration: %s -- staying eager" % ("capture failed (%s)" % failure if failure is not None else
                                                             "the captured iteration does not replay bit for bit on this stack"))
        return ok

    def _graph_usable(self):
        st = getattr(self, "_graph", None)
        return (st is not None and self.training and st["shape"] == (self.num_envs, self.frames_per_wdw) and st["env"] is self.env
                and all(self.opts.get(k) == v for k, v in st["weights"].items()))

    def iteration(self):
        """forward() + backward() of one optimisation iteration; returns forward()'s dict.  The replay of the captured graph when there is
        one for the current window shape and loss weights (capture_iteration), eager otherwise -- same random numbers, same kernels on
        the same data, bit-identical results either way (tests/test_gpu_workload.py)."""
        if not self._graph_usable() and getattr(self, "_graph_wanted", False) and self.training and getattr(self, "_graph_captures", 0) < 8:
            self.capture_iteration()   # another window shape / env (reinit_envs(overwrite=True)) / loss weight than the graph was captured with
        if not self._graph_usable():
            out = self.forward()
            self.backward(out["total_loss"])
            return out
        self._attach_zero_grads()
        self._inv_norm_inertia()   # (a host-side key check; refreshes the tensor the graph reads when norm_body_inertia was reloaded)
        st = self._graph
        st["fs"].copy_(self.compute_frame_start(), non_blocking=True)
        noise = self.make_q_init_noise()
        st["noise"].copy_(noise, non_blocking=True) if noise is not None else st["noise"].zero_()
        st["graph"].replay()
        st["replays"] += 1
        for p, g in st["grads"]:   # (update() detaches them with zero_grad(set_to_none=True); the buffers are the graph's)
            p.grad = g
        prev = getattr(self, "_pending_nan", None)
        self._pending_nan = st["nan"] if prev is None else (prev | st["nan"])
        side = st["side"]
        from .dp_model import HostFrames
        self.grfs, self.jafs, self.traj_loss_info = side["grfs"], side["jafs"], side["info"]
        self.sim_trajs, self.target_trajs, self.pid_ref = HostFrames(side["sim"]), HostFrames(side["tgt"]), HostFrames(side["pid"])
        return st["out"]

    @torch.no_grad()
    def query(self, img_size=None):
        """Env 0 of the last forward(), frame by frame, in the reference's keys (dp_model.py:843-902): the simulated, target and
        control-reference robots POSED -- the reference hands trimesh objects to its renderer (articulate_robot_rbrt:
        collision-mesh vertices rotated and translated by each body's pose); here each is the array of those posed vertices
        [F, n_vertices, 3] (the collision vertices of all bodies in template order = the contact candidates, `vertex_body`
        says which body each belongs to; triangle indices, colours and force arrows are the renderer's and out of scope) --
        plus the centre of mass of the target (`com_k`) and simulated (`com`) robot per frame (dp_utils.py:86-90 with the env's
        template masses, as the reference), `max_w`, the raw body poses (`*_poses` [F, nb, 7]) and the frame forces."""
        sim = np.stack(list(self.sim_trajs), 0).astype(np.float64)
        tgt = np.stack(list(self.target_trajs), 0).astype(np.float64)
        ref = np.stack(list(self.pid_ref), 0).astype(np.float64)
        pts = np.asarray(self.template["contact_point"], np.float64)
        cb = np.asarray(self.template["contact_body"], np.int64)
        part_com = np.asarray(self.template["body_com"], np.float64)
        part_mass = np.asarray(self.template["body_mass"], np.float64)

        def rotm(q):  # [..., 4] (x, y, z, w), unit -> [..., 3, 3]
            x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
            return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                             2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                             2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(q.shape[:-1] + (3, 3))

        def pose_vertices(traj):  # [F, nb, 7] -> [F, nc, 3]
            qn = traj[..., 3:] / np.linalg.norm(traj[..., 3:], axis=-1, keepdims=True)
            R, t = rotm(qn), traj[..., :3]
            return np.einsum("fcij,cj->fci", R[:, cb], pts) + t[:, cb]

        def com_of(traj):  # dp_utils.py:86-90
            qn = traj[..., 3:] / np.linalg.norm(traj[..., 3:], axis=-1, keepdims=True)
            body_com = np.einsum("fbij,bj->fbi", rotm(qn), part_com) + traj[..., :3]
            return (body_com * part_mass[None, :, None]).sum(1) / part_mass.sum()

        data = {"sim_traj": pose_vertices(sim), "target_traj": pose_vertices(tgt), "control_ref": pose_vertices(ref),
                "vertex_body": cb, "sim_poses": sim, "target_poses": tgt, "control_ref_poses": ref,
                "com": com_of(sim), "com_k": list(com_of(tgt)), "body_mass": self.body_mass.detach().cpu().numpy(),
                "grf": torch.stack(self.grfs, 0).cpu().numpy(), "jaf": torch.stack(self.jafs, 0).cpu().numpy()}
        # max_w (dp_model.py:897-899: 3 x the robot's horizontal extent in its rest pose); the rest pose here is the first target frame
        rest = data["target_traj"][0] - data["target_traj"][0].mean(0, keepdims=True)
        data["max_w"] = float(3 * np.abs(rest[:, [0, 2]]).max())
        if img_size is not None:
            data["img_size"] = img_size
        return data
