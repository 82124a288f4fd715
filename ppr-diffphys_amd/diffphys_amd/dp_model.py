"""The two autograd boundaries of the reference, backed by the HIP library.

``ForwardWarp`` and ``ForwardKinematics`` keep the reference's names, argument
order, tensor layouts, side effects and gradient post-processing
(/root/reference/diffphys/dp_model.py:1022-1130 and :1145-1400) so that
``phys_model.forward`` can call them unchanged; see INTEGRATION.md.

Differences that are deliberate:
  * one kernel launch per rollout instead of a Python loop of Warp launches;
  * ``self.state_steps`` is not needed (the trajectory lives in a workspace
    tensor owned by the autograd ctx);
  * ``self.sim_trajs`` / ``body_q_numpy`` are lazy host views (:class:`HostFrames`): the env-0 copy the reference makes
    eagerly for visualisation happens on first access, so neither ``forward`` synchronises the host;
  * ``body_mass`` gets a zero gradient: the reference kernel loads it and never
    uses it (/root/reference/diffphys/integrator_euler.py:43).
"""
import numpy as np
import torch

from . import hip_backend


def convert_ppr_warp(tensor):
    """[linear, angular] <=> [angular, linear] on the last axis (dp_model.py:1014-1019)."""
    return torch.cat([tensor[..., 3:6], tensor[..., 0:3], tensor[..., 6:]], -1)


class HostFrames:
    """Env-0 poses of every frame as a list of numpy arrays -- what the reference builds eagerly for visualisation
    (``self.sim_trajs`` dp_model.py:1237-1244, ``body_q_numpy`` :1066-1072) and ``query()`` consumes (:855-860).  Here the
    device-to-host copy happens on first access, so ``forward()`` itself never synchronises the host (SURVEY section 8(b):
    "no host sync inside the op except the optional env-0 trajectory copy")."""

    def __init__(self, frames_dev):  # [F, nb, 7] device tensor (a detached view; nothing writes it afterwards)
        self._dev, self._host = frames_dev.detach(), None

    def _get(self):
        if self._host is None:
            self._host = list(self._dev.detach().cpu().numpy())
            self._dev = None
        return self._host

    def __len__(self):
        return len(self._host) if self._host is not None else int(self._dev.shape[0])

    def __getitem__(self, i):
        return self._get()[i]

    def __iter__(self):
        return iter(self._get())


def _f32c(t):
    """detached, float32, contiguous -- without touching a tensor that already is (the usual case; three tensor ops saved per input)"""
    t = t.detach()
    return t if (t.dtype is torch.float32 and t.is_contiguous()) else t.to(torch.float32).contiguous()


class ForwardKinematics(torch.autograd.Function):
    """rj_q [T,bs,nq], rj_qd [T,bs,nqd], env -> body_q [bs,T,nb,7], body_qd [bs,T,nb,6], body_q_numpy."""

    @staticmethod
    def forward(ctx, rj_q, rj_qd, env):
        is_cuda = rj_q.is_cuda
        if not is_cuda:  # the reference moves CPU inputs to the GPU (dp_model.py:1030-1035)
            rj_q, rj_qd = rj_q.cuda(), rj_qd.cuda()
        num_frames, bs, _ = rj_q.shape
        dm = hip_backend.device_model(env)
        jq = rj_q.detach().to(torch.float32).contiguous()
        jqd = rj_qd.detach().to(torch.float32).contiguous()
        body_q, body_qd = dm.fk_forward(jq.view(num_frames * bs, -1), jqd.view(num_frames * bs, -1))
        ctx.dm, ctx.shape, ctx.is_cuda = dm, (num_frames, bs), is_cuda
        ctx.save_for_backward(jq, jqd)
        body_q = body_q.view(num_frames, bs, dm.nb, 7).permute(1, 0, 2, 3).contiguous()
        body_qd = body_qd.view(num_frames, bs, dm.nb, 6).permute(1, 0, 2, 3).contiguous()
        body_q_numpy = HostFrames(body_q[0])  # env 0, one array per frame (visualisation); copied to the host on first use
        if not is_cuda:
            body_q, body_qd = body_q.cpu(), body_qd.cpu()
        return body_q, body_qd, body_q_numpy

    @staticmethod
    def backward(ctx, adj_body_qs, adj_body_qd, _):
        jq, jqd = ctx.saved_tensors
        num_frames, bs = ctx.shape
        dm = ctx.dm
        aq = adj_body_qs.to(jq.device, torch.float32).permute(1, 0, 2, 3).contiguous()
        aqd = adj_body_qd.to(jq.device, torch.float32).permute(1, 0, 2, 3).contiguous()
        gq, gqd = dm.fk_backward(jq.view(num_frames * bs, -1), jqd.view(num_frames * bs, -1), aq, aqd)

        def post(g, last):  # the reference's post-processing (dp_model.py:1109-1110,1122-1123: NaN -> 0, values > 1 -> 1, upper
            g = g.view(num_frames, bs, last)  # clamp only) is applied by the kernel where it stores the gradients (pd_fk_backward)
            return g if ctx.is_cuda else g.cpu()

        return post(gq, dm.nq), post(gqd, dm.nqd), None


class ForwardWarp(torch.autograd.Function):
    """ForwardWarp.apply(q_init, qd_init, torques, res_f, refs, target_ke, target_kd, body_mass,
    body_inv_mass, body_inertia, body_inv_inertia, self) -> (wp_pos [F,bs*nb,7], wp_vel [F,bs*nb,6]).

    Read from ``self``: ``env``, ``steps_idx``, ``frame2step``, ``dt``, ``num_envs``.
    Written to ``self``: ``grfs``, ``jafs`` (lists of F tensors [bs*nb,6]), ``sim_trajs`` (F numpy [nb,7], env 0)."""

    @staticmethod
    def forward(ctx, q_init, qd_init, torques, res_f, refs, target_ke, target_kd, body_mass, body_inv_mass, body_inertia,
                body_inv_inertia, self):
        env = self.env
        dm = hip_backend.device_model(env)
        bs = int(self.num_envs)
        nsteps = len(self.steps_idx)
        frame2step = list(self.frame2step)
        dev = q_init.device
        c = _f32c
        inp = [c(t) for t in (q_init, qd_init, torques, res_f, refs, target_ke, target_kd, body_inv_mass, body_inertia,
                              body_inv_inertia)]
        frame2step = [int(s) for s in frame2step]
        wp_pos, wp_vel, grf, jaf, ws = dm.rollout_forward(bs, nsteps, self.dt, *inp, frame2step=frame2step)
        ctx.dm, ctx.meta = dm, (bs, nsteps, float(self.dt), frame2step)
        ctx.save_for_backward(ws, *inp)
        ctx.mass_shape = body_mass.shape
        # side outputs consumed by phys_model.query() (dp_model.py:855-860); a frame at state `nsteps` has no force
        # snapshot in the reference (:1225-1228 append for step in steps_idx only): keep its list lengths
        has_f = [f for f, s in enumerate(frame2step) if s < nsteps]
        self.grfs = [grf[f] for f in has_f]
        self.jafs = [jaf[f] for f in has_f]
        self.sim_trajs = HostFrames(wp_pos[:, : dm.nb])
        return wp_pos, wp_vel

    @staticmethod
    def backward(ctx, adj_body_qs, adj_body_qd):
        ws, q_init, qd_init, torques, res_f, refs, ke, kd, inv_m, inertia, inv_inertia = ctx.saved_tensors
        bs, nsteps, dt, frame2step = ctx.meta
        g = ctx.dm.rollout_backward(bs, nsteps, dt, q_init, qd_init, torques, refs, ke, kd, inv_m, inertia, inv_inertia,
                                    frame2step, ws, adj_body_qs.to(torch.float32).contiguous(),
                                    adj_body_qd.to(torch.float32).contiguous())
        # remove_nan (dp_model.py:1294-1384: NaN -> 0 on every returned gradient, inf kept) is applied by the adjoint kernel where it
        # stores the gradients (pd_rollout_backward): no pass over the tensors here
        return (g["q_init"], g["qd_init"], g["torques"].view_as(torques), g["res_f"].view_as(res_f),
                g["refs"].view_as(refs), g["target_ke"], g["target_kd"],
                torch.zeros(ctx.mass_shape, dtype=torch.float32, device=ws.device), g["body_inv_mass"],
                g["body_inertia"].view_as(inertia), g["body_inv_inertia"].view_as(inv_inertia), None)


class ForwardWarpTrajLoss(torch.autograd.Function):
    """ForwardWarp with the one loss term that back-propagates through the rollout evaluated INSIDE it (SURVEY section 8 row f4;
    reference: dp_model.py:733-779 -- ForwardWarp.apply, then loss_traj = reduce_loss(se3_loss(sim_position, target_position).mean(-1)
    with outseq entries zeroed, clip=True); pos_state / vel_state / distill use sim_position.detach(), :795,802):

        loss_traj, wp_pos, wp_vel = ForwardWarpTrajLoss.apply(<the 11 inputs of ForwardWarp>, target_position [bs,F,nb,7],
                                                              outseq_idx [bs,F] bool, self)

    loss_traj is the reduced scalar; wp_pos / wp_vel come back DETACHED (non-differentiable outputs: what the other loss terms and
    query() consume).  Forward = ONE rollout launch that also evaluates se3_loss and its gradients at the frame states + one
    one-workgroup launch for reduce_loss; backward = a few-microsecond launch that builds the seeds from what the forward left, scaled
    by the upstream gradient of loss_traj read on the device, then the adjoint rollout launch -- no pose, seed or per-frame loss goes
    through a torch op in between.
    Side outputs on ``self`` as ForwardWarp, plus ``self.traj_loss_info`` = (loss, clip threshold, positives left, clipped envs)."""

    @staticmethod
    def forward(ctx, q_init, qd_init, torques, res_f, refs, target_ke, target_kd, body_mass, body_inv_mass, body_inertia,
                body_inv_inertia, target_position, outseq_idx, self):
        return _traj_loss_forward(ctx, (q_init, qd_init, torques, res_f, refs, target_ke, target_kd, body_mass, body_inv_mass, body_inertia,
                                        body_inv_inertia), target_position, outseq_idx, self, None)

    @staticmethod
    def backward(ctx, g_loss, _gp, _gv):
        return _traj_loss_backward(ctx, g_loss, None) + (None,)


class ForwardWarpTrajLossFK(torch.autograd.Function):
    """ForwardWarpTrajLoss plus the FK of the control reference (dp_model.py:758 of the reference:
    ``ForwardKinematics.apply(queried_q, queried_qd, self.env)``) in the SAME launches -- the second half of SURVEY section 8 row f4:

        loss_traj, wp_pos, wp_vel, queried_position [bs,F,nb,7], queried_velocity [bs,F,nb,6], pid_ref =
            ForwardWarpTrajLossFK.apply(<the 11 inputs of ForwardWarp>, target_position, outseq_idx, queried_q [F,bs,nq], queried_qd [F,bs,nqd], self)

    The FK chains are extra workgroups of the reduce_loss launch that follows the rollout launch, their adjoint (with
    ForwardKinematics.backward's NaN -> 0 / > 1 -> 1 post-processing) extra workgroups of the seeds launch in front of the adjoint
    rollout; the kernels write / read the [bs, F, ...] layout directly (the reference permutes and copies, :1093-1094).  Values and
    gradients are those of ForwardKinematics.apply bit for bit (same device code)."""

    @staticmethod
    def forward(ctx, q_init, qd_init, torques, res_f, refs, target_ke, target_kd, body_mass, body_inv_mass, body_inertia,
                body_inv_inertia, target_position, outseq_idx, queried_q, queried_qd, self):
        return _traj_loss_forward(ctx, (q_init, qd_init, torques, res_f, refs, target_ke, target_kd, body_mass, body_inv_mass, body_inertia,
                                        body_inv_inertia), target_position, outseq_idx, self, (queried_q, queried_qd))

    @staticmethod
    def backward(ctx, g_loss, _gp, _gv, g_qpos, g_qvel, _pid):
        g = _traj_loss_backward(ctx, g_loss, (g_qpos, g_qvel))
        return g[:13] + g[13:] + (None,)


def _traj_loss_forward(ctx, rollout_inputs, target_position, outseq_idx, self, queried):
    (q_init, qd_init, torques, res_f, refs, target_ke, target_kd, body_mass, body_inv_mass, body_inertia, body_inv_inertia) = rollout_inputs
    dm = hip_backend.device_model(self.env)
    bs, nsteps = int(self.num_envs), len(self.steps_idx)
    frame2step = [int(s) for s in self.frame2step]
    c = _f32c
    inp = [c(t) for t in (q_init, qd_init, torques, res_f, refs, target_ke, target_kd, body_inv_mass, body_inertia,
                          body_inv_inertia)]
    tgt = c(target_position).view(bs, len(frame2step), dm.nb, 7)
    outseq = None if outseq_idx is None else outseq_idx.detach().to(torch.bool).contiguous()
    need_gt = target_position.requires_grad
    fk = None if queried is None else (c(queried[0]), c(queried[1]))
    wp_pos, wp_vel, grf, jaf, ws, tl = dm.rollout_forward_traj_loss(bs, nsteps, self.dt, *inp, frame2step=frame2step, target_pos=tgt,
                                                                    outseq=outseq, rot_ratio=0.1, want_seed_gt=need_gt, fk=fk)
    # What the backward needs of tl -- WITHOUT the Function's outputs: an output kept on ctx closes the cycle ctx -> tensor -> grad_fn ->
    # ctx through C++ references that Python's collector cannot see, and every iteration would then pin its buffers for good (ADVICE r4:
    # ~30 MB per iteration at 4096 envs).  The FK outputs' shapes are all the backward wants of them.
    ctx.dm, ctx.meta = dm, (bs, nsteps, float(self.dt), frame2step)
    ctx.tl = {k: v for k, v in tl.items() if k not in ("fk_body_q", "fk_body_qd", "reduced")}
    ctx.fk_shapes = None if fk is None else (tuple(tl["fk_body_q"].shape), tuple(tl["fk_body_qd"].shape))
    ctx.save_for_backward(ws, *inp, *(fk or ()))
    ctx.mass_shape, ctx.tgt_shape = body_mass.shape, target_position.shape
    has_f = [f for f, s in enumerate(frame2step) if s < nsteps]
    self.grfs = [grf[f] for f in has_f]
    self.jafs = [jaf[f] for f in has_f]
    self.sim_trajs = HostFrames(wp_pos[:, : dm.nb])
    self.traj_loss_info = tl["reduced"]
    ctx.mark_non_differentiable(wp_pos, wp_vel)
    out = (tl["reduced"][0].clone(), wp_pos, wp_vel)
    if fk is not None:
        body_q, body_qd = tl["fk_body_q"], tl["fk_body_qd"]
        out += (body_q, body_qd, HostFrames(body_q[0]))  # env 0, one array per frame (visualisation), as ForwardKinematics
    return out


def _traj_loss_backward(ctx, g_loss, g_queried):
    ws, q_init, qd_init, torques, res_f, refs, ke, kd, inv_m, inertia, inv_inertia = ctx.saved_tensors[:11]
    bs, nsteps, dt, frame2step = ctx.meta
    tl = ctx.tl
    gl = g_loss.detach().to(torch.float32).reshape(1).contiguous()
    fk = None
    if g_queried is not None and (g_queried[0] is not None or g_queried[1] is not None):
        jq, jqd = ctx.saved_tensors[11:13]
        z = lambda sh: torch.zeros(sh, dtype=torch.float32, device=ws.device)
        aq = z(ctx.fk_shapes[0]) if g_queried[0] is None else g_queried[0].to(torch.float32).contiguous()
        aqd = z(ctx.fk_shapes[1]) if g_queried[1] is None else g_queried[1].to(torch.float32).contiguous()
        fk = (jq, jqd, aq, aqd)
    g = ctx.dm.rollout_backward_traj_loss(bs, nsteps, dt, q_init, qd_init, torques, refs, ke, kd, inv_m, inertia, inv_inertia,
                                          frame2step, ws, tl, gl, fk=fk)
    g_tgt = None
    if ctx.needs_input_grad[11] and tl["seed_gt"] is not None:  # d loss_traj / d target pose = g x share / nb x d se3 / d gt
        # a zero share is an ASSIGNMENT in the reference (loss_seq[i, idx:] = 0, loss_traj[outseq_idx] = 0): nothing flows there, not 0 * inf
        k = (tl["scale"] * (gl / ctx.dm.nb))[:, :, None, None]
        g_tgt = torch.where(k != 0, tl["seed_gt"] * k, torch.zeros_like(tl["seed_gt"])).view(ctx.tgt_shape)
    out = (g["q_init"], g["qd_init"], g["torques"].view_as(torques), g["res_f"].view_as(res_f),
           g["refs"].view_as(refs), g["target_ke"], g["target_kd"],
           torch.zeros(ctx.mass_shape, dtype=torch.float32, device=ws.device), g["body_inv_mass"],
           g["body_inertia"].view_as(inertia), g["body_inv_inertia"].view_as(inv_inertia), g_tgt, None)
    if g_queried is not None:
        out += (g.get("fk_joint_q"), g.get("fk_joint_qd"))
    ctx.tl = None  # the sweep's buffers go with it (a second backward through the same graph is not supported, as with saved tensors)
    return out
