"""ORACLE (test infrastructure, not product code): float64/float32 PyTorch
restatement of the reference's hot path, gradients by autograd.

PARITY UNPINNED: the reference's arithmetic lives partly in warp_lang==0.7.2
(/root/reference/requirements.txt:12), which is neither vendored under
/root/reference nor installable here, and the reference ships no tests or
golden vectors.  This file therefore restates

  * /root/reference/diffphys/integrator_euler.py:21-91    integrate_bodies
  * /root/reference/diffphys/integrator_euler.py:93-179   eval_body_contacts
  * /root/reference/diffphys/integrator_euler.py:234-286  quat_twist / quat_decompose / eval_joint_force
  * /root/reference/diffphys/integrator_euler.py:289-451  eval_body_joints
  * /root/reference/diffphys/integrator_euler.py:491-620  compute_forces / simulate (sequencing, grf/jaf)
  * /root/reference/diffphys/dp_model.py:1133-1249        wp_add + ForwardWarp.forward rollout
  * warp.sim.articulation.eval_fk (third party; SURVEY.md Appendix A.3, from recall)

line by line, with Warp built-ins as in SURVEY.md Appendix A.1.  It is pinned
by analytic known-answer tests (tests/test_oracle_known_answers.py), autograd vs
central finite differences, and agreement with the independent C restatement
(oracle/ref_c).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.

Layouts are the reference's: quat (x,y,z,w); transform (p, q); twist (w, v);
wrench (tau, f); every per-body array is env-major [bs*nb, ...].
"""
import torch

JOINT_REVOLUTE, JOINT_FIXED, JOINT_FREE, JOINT_COMPOUND = 1, 3, 4, 5


# ---------------------------------------------------------------- quat / vec
def q_mul(a, b):
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack(
        [
            aw * bx + bw * ax + ay * bz - az * by,
            aw * by + bw * ay + az * bx - ax * bz,
            aw * bz + bw * az + ax * by - ay * bx,
            aw * bw - ax * bx - ay * by - az * bz,
        ],
        -1,
    )


def q_conj(q):
    return torch.cat([-q[..., :3], q[..., 3:]], -1)


def q_rot(q, v):
    """x(2w^2-1) + 2w(qv x x) + 2 qv (qv . x)   """
    qv, w = q[..., :3], q[..., 3:]
    return v * (2.0 * w * w - 1.0) + 2.0 * w * torch.cross(qv, v, dim=-1) + 2.0 * qv * (qv * v).sum(-1, keepdim=True)


def q_rot_inv(q, v):
    qv, w = q[..., :3], q[..., 3:]
    return v * (2.0 * w * w - 1.0) - 2.0 * w * torch.cross(qv, v, dim=-1) + 2.0 * qv * (qv * v).sum(-1, keepdim=True)


def q_axis_angle(axis, angle):
    half = angle * 0.5
    return torch.cat([axis * torch.sin(half)[..., None], torch.cos(half)[..., None]], -1)


def safe_normalize(v):
    """wp.normalize: v/|v|, and 0 (with zero gradient) for |v| == 0."""
    l = torch.sqrt((v * v).sum(-1, keepdim=True))
    ok = l > 0
    return torch.where(ok, v / torch.where(ok, l, torch.ones_like(l)), torch.zeros_like(v))


def safe_length(v):
    s = (v * v).sum(-1)
    ok = s > 0
    return torch.where(ok, torch.sqrt(torch.where(ok, s, torch.ones_like(s))), torch.zeros_like(s))


def q_normalize(q):
    return q / torch.sqrt((q * q).sum(-1, keepdim=True))


def cross(a, b):
    return torch.cross(a, b, dim=-1)


def dot(a, b):
    return (a * b).sum(-1)


class _GuardedArc(torch.autograd.Function):
    """acos / asin with the argument clamped to [-1, 1] and an adjoint that drops the contribution (instead of
    inf / NaN) where sqrt(1-x^2) is not > 0, like Warp's builtins.  POLICY, see DESIGN.md section 6."""

    @staticmethod
    def forward(ctx, x, is_acos):
        ctx.save_for_backward(x)
        ctx.is_acos = is_acos
        xc = x.clamp(-1.0, 1.0)  # Warp's builtins clamp the argument (recall)
        return torch.acos(xc) if is_acos else torch.asin(xc)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        d = torch.sqrt(1.0 - x * x)
        ok = d > 0
        inv = torch.where(ok, 1.0 / torch.where(ok, d, torch.ones_like(d)), torch.zeros_like(d))
        return (-g * inv if ctx.is_acos else g * inv), None


def acos_g(x):
    return _GuardedArc.apply(x, True)


def asin_g(x):
    return _GuardedArc.apply(x, False)


# ------------------------------------------------------------------- template
class Template:
    """Template arrays as torch tensors of a chosen dtype (ints stay python lists)."""

    def __init__(self, tpl, dtype=torch.float64):
        self.dtype = dtype
        self.nb, self.nq, self.nqd = int(tpl["nb"]), int(tpl["nq"]), int(tpl["nqd"])
        self.joint_type = [int(x) for x in tpl["joint_type"]]
        self.joint_parent = [int(x) for x in tpl["joint_parent"]]
        self.q_start = [int(x) for x in tpl["joint_q_start"]]
        self.qd_start = [int(x) for x in tpl["joint_qd_start"]]
        t = lambda k: torch.tensor(tpl[k], dtype=dtype)
        self.X_p, self.X_c, self.axis, self.com = t("joint_X_p"), t("joint_X_c"), t("joint_axis"), t("body_com")
        self.limit_lower, self.limit_upper = t("joint_limit_lower"), t("joint_limit_upper")
        self.limit_ke, self.limit_kd = t("joint_limit_ke"), t("joint_limit_kd")
        self.c_body = torch.tensor(tpl["contact_body"], dtype=torch.long)
        self.c_point, self.c_dist = t("contact_point"), t("contact_dist")
        self.c_mat = torch.tensor(tpl["contact_material"], dtype=torch.long)
        self.materials = t("shape_materials")
        self.gravity = t("gravity")
        self.attach_ke, self.attach_kd = float(tpl["joint_attach_ke"]), float(tpl["joint_attach_kd"])


# ------------------------------------------------------------------------- FK
def eval_fk(T, joint_q, joint_qd):
    """joint_q [bs,nq], joint_qd [bs,nqd] -> body_q [bs,nb,7], body_qd [bs,nb,6].  SURVEY.md A.3."""
    bs = joint_q.shape[0]
    zero3 = torch.zeros(bs, 3, dtype=T.dtype)
    ex = torch.tensor([1.0, 0.0, 0.0], dtype=T.dtype).expand(bs, 3)
    ey = torch.tensor([0.0, 1.0, 0.0], dtype=T.dtype).expand(bs, 3)
    ez = torch.tensor([0.0, 0.0, 1.0], dtype=T.dtype).expand(bs, 3)
    body_q, body_qd = [], []
    for i in range(T.nb):
        par = T.joint_parent[i]
        if par >= 0:
            p_wp, q_wp = body_q[par][:, :3], body_q[par][:, 3:]
            v_wp = body_qd[par]
        else:
            p_wp, q_wp = zero3, torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=T.dtype).expand(bs, 4)
            v_wp = torch.zeros(bs, 6, dtype=T.dtype)
        ty, qs, qds = T.joint_type[i], T.q_start[i], T.qd_start[i]
        axis = T.axis[i].expand(bs, 3)
        if ty == JOINT_REVOLUTE:
            p_jc, q_jc = zero3, q_axis_angle(axis, joint_q[:, qs])
            w_jc, v_jc = axis * joint_qd[:, qds : qds + 1], zero3
        elif ty == JOINT_FIXED:
            p_jc, q_jc = zero3, torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=T.dtype).expand(bs, 4)
            w_jc, v_jc = zero3, zero3
        elif ty == JOINT_FREE:
            p_jc, q_jc = joint_q[:, qs : qs + 3], joint_q[:, qs + 3 : qs + 7]
            w_jc, v_jc = joint_qd[:, qds : qds + 3], joint_qd[:, qds + 3 : qds + 6]
        elif ty == JOINT_COMPOUND:
            q_off = T.X_c[i, 3:].expand(bs, 4)
            a0 = q_rot(q_off, ex)
            q0 = q_axis_angle(a0, joint_q[:, qs + 0])
            a1 = q_rot(q_mul(q0, q_off), ey)
            q1 = q_axis_angle(a1, joint_q[:, qs + 1])
            a2 = q_rot(q_mul(q1, q_mul(q0, q_off)), ez)
            q2 = q_axis_angle(a2, joint_q[:, qs + 2])
            p_jc, q_jc = zero3, q_mul(q2, q_mul(q1, q0))
            w_jc = a0 * joint_qd[:, qds : qds + 1] + a1 * joint_qd[:, qds + 1 : qds + 2] + a2 * joint_qd[:, qds + 2 : qds + 3]
            v_jc = zero3
        else:
            raise NotImplementedError("joint type %d" % ty)
        # X_wj = X_wp * X_pj ; X_wc = X_wj * X_jc
        p_pj, q_pj = T.X_p[i, :3].expand(bs, 3), T.X_p[i, 3:].expand(bs, 4)
        p_wj = p_wp + q_rot(q_wp, p_pj)
        q_wj = q_mul(q_wp, q_pj)
        p_wc = p_wj + q_rot(q_wj, p_jc)
        q_wc = q_mul(q_wj, q_jc)
        ang = q_rot(q_wj, w_jc)
        lin = q_rot(q_wj, v_jc)
        v_wc = v_wp + torch.cat([ang, lin + cross(ang, T.com[i].expand(bs, 3))], -1)
        body_q.append(torch.cat([p_wc, q_wc], -1))
        body_qd.append(v_wc)
    return torch.stack(body_q, 1), torch.stack(body_qd, 1)


# ------------------------------------------------------------------- contacts
def eval_body_contacts(T, body_q, body_qd, body_f):
    """integrator_euler.py:93-179.  body_* are [bs,nb,*]; returns body_f minus contact wrenches."""
    if T.c_body.numel() == 0:
        return body_f
    bs = body_q.shape[0]
    cb = T.c_body
    X = body_q[:, cb]  # [bs,Nc,7]
    p, q = X[..., :3], X[..., 3:]
    w, v = body_qd[:, cb, :3], body_qd[:, cb, 3:]
    n = torch.tensor([0.0, 1.0, 0.0], dtype=T.dtype)
    cp = p + q_rot(q, T.c_point[None].expand(bs, -1, 3)) - n * T.c_dist[None, :, None]  # :121
    r = cp - (p + q_rot(q, T.com[cb][None].expand(bs, -1, 3)))  # :124
    dpdt = v + cross(w, r)  # :127
    c = cp[..., 1]  # :130
    active = ~(c > 0.0)  # :132
    mat = T.materials[T.c_mat]  # [Nc,4]
    ke, kd, kf, mu = mat[:, 0], mat[:, 1], mat[:, 2], mat[:, 3]
    vn = dpdt[..., 1]
    vt = dpdt - n * vn[..., None]
    fn = c * ke
    step_c = (c < 0.0).to(T.dtype)
    fd = torch.minimum(vn, torch.zeros_like(vn)) * kd * step_c  # :150
    # wp.min(a, b) routes the adjoint to b on ties; torch.where reproduces that
    a_, b_ = kf * safe_length(vt), 0.0 - mu * (fn + fd)
    ft = safe_normalize(vt) * torch.where(a_ < b_, a_, b_)[..., None]  # :165
    f_total = n * (fn + fd)[..., None] + ft
    f_total = torch.clamp(f_total, -500.0, 500.0)  # :172-175
    t_total = cross(r, f_total)
    wrench = torch.cat([t_total, f_total], -1) * active[..., None].to(T.dtype)
    wrench = torch.where(active[..., None], wrench, torch.zeros_like(wrench))
    out = body_f.index_add(1, cb, -wrench)  # atomic_sub :179
    return out


# --------------------------------------------------------------------- joints
def eval_joint_force(q, qd, target, ke, kd, act, lower, upper, lke, lkd, axis):
    """integrator_euler.py:261-286"""
    zero = torch.zeros_like(q)
    limit_f = zero
    limit_f = torch.where(q < lower, lke * (lower - q) - lkd * torch.minimum(qd, zero), limit_f)
    limit_f = torch.where(q > upper, lke * (upper - q) - lkd * torch.maximum(qd, zero), limit_f)
    return (ke * (q - target) + kd * qd + act - limit_f)[..., None] * axis


def quat_decompose(q):
    """integrator_euler.py:245-258; R = rotation matrix of q, R[row, col]."""
    bs = q.shape[0]
    e = torch.eye(3, dtype=q.dtype)
    c0, c1, c2 = q_rot(q, e[0].expand(bs, 3)), q_rot(q, e[1].expand(bs, 3)), q_rot(q, e[2].expand(bs, 3))
    # mat33(c0,c1,c2) fills columns: R[r,c] = c_c[r]
    phi = torch.atan2(c2[:, 1], c2[:, 2])
    theta = asin_g(-c2[:, 0])
    psi = torch.atan2(c1[:, 0], c0[:, 0])
    return -torch.stack([phi, theta, psi], -1)


def eval_body_joints(T, body_q, body_qd, body_f, target, act, target_ke, target_kd):
    """integrator_euler.py:289-451.  target/act/target_ke/target_kd are [bs,nqd]."""
    bs = body_q.shape[0]
    out = [body_f[:, i] for i in range(T.nb)]
    for i in range(T.nb):
        ty = T.joint_type[i]
        if ty == JOINT_FREE:
            continue  # :382
        par = T.joint_parent[i]
        p_pj, q_pj = T.X_p[i, :3].expand(bs, 3), T.X_p[i, 3:].expand(bs, 4)
        x_p, q_p = p_pj, q_pj
        r_p = torch.zeros(bs, 3, dtype=T.dtype)
        w_p = torch.zeros(bs, 3, dtype=T.dtype)
        v_p = torch.zeros(bs, 3, dtype=T.dtype)
        if par >= 0:  # :326-333
            pp, qp = body_q[:, par, :3], body_q[:, par, 3:]
            x_p = pp + q_rot(qp, p_pj)
            q_p = q_mul(qp, q_pj)
            r_p = x_p - (pp + q_rot(qp, T.com[par].expand(bs, 3)))
            w_p, v_p = body_qd[:, par, :3], body_qd[:, par, 3:]
        x_c, q_c = body_q[:, i, :3], body_q[:, i, 3:]
        r_c = x_c - (x_c + q_rot(q_c, T.com[i].expand(bs, 3)))  # :338
        w_c, v_c = body_qd[:, i, :3], body_qd[:, i, 3:]
        qds = T.qd_start[i]
        x_err = x_c - x_p
        r_err = q_mul(q_conj(q_p), q_c)
        v_err = v_c - v_p
        w_err = w_c - w_p
        ake, akd = T.attach_ke, T.attach_kd
        ads = 0.01
        t_total = torch.zeros(bs, 3, dtype=T.dtype)
        f_total = torch.zeros(bs, 3, dtype=T.dtype)
        if ty == JOINT_FIXED:  # :385-390
            ang_err = safe_normalize(r_err[:, :3]) * (acos_g(r_err[:, 3]) * 2.0)[:, None]
            f_total = f_total + x_err * ake + v_err * akd
            t_total = t_total + q_rot(q_p, ang_err) * ake + w_err * akd * ads
        elif ty == JOINT_REVOLUTE:  # :392-409
            axis = T.axis[i].expand(bs, 3)
            axis_p = q_rot(q_p, axis)
            axis_c = q_rot(q_c, axis)
            a = dot(r_err[:, :3], axis)[:, None] * axis
            twist = q_normalize(torch.cat([a, r_err[:, 3:]], -1))
            sgn = torch.where(dot(axis, twist[:, :3]) < 0, -1.0, 1.0).to(T.dtype)
            q = acos_g(twist[:, 3]) * 2.0 * sgn
            qd = dot(w_err, axis_p)
            sl = slice(qds, qds + 1)
            t_total = eval_joint_force(
                q, qd, target[:, qds], target_ke[:, qds], target_kd[:, qds], act[:, qds],
                T.limit_lower[qds], T.limit_upper[qds], T.limit_ke[qds], T.limit_kd[qds], axis_p,
            )
            swing_err = cross(axis_p, axis_c)
            f_total = f_total + x_err * ake + v_err * akd
            t_total = t_total + swing_err * ake + (w_err - qd[:, None] * axis_p) * akd * ads
        elif ty == JOINT_COMPOUND:  # :411-445
            q_off = T.X_c[i, 3:].expand(bs, 4)
            q_pc = q_mul(q_mul(q_mul(q_conj(q_off), q_conj(q_p)), q_c), q_off)
            angles = quat_decompose(q_pc)
            e = torch.eye(3, dtype=T.dtype)
            axis_0 = e[0].expand(bs, 3)
            q_0 = q_axis_angle(axis_0, angles[:, 0])
            axis_1 = q_rot(q_0, e[1].expand(bs, 3))
            q_1 = q_axis_angle(axis_1, angles[:, 1])
            axis_2 = q_rot(q_mul(q_1, q_0), e[2].expand(bs, 3))
            q_w = q_mul(q_p, q_off)
            t_total = torch.zeros(bs, 3, dtype=T.dtype)
            for k, ax in enumerate((axis_0, axis_1, axis_2)):
                axw = q_rot(q_w, ax)
                j = qds + k
                t_total = t_total + eval_joint_force(
                    angles[:, k], dot(axw, w_err), target[:, j], target_ke[:, j], target_kd[:, j], act[:, j],
                    T.limit_lower[j], T.limit_upper[j], T.limit_ke[j], T.limit_kd[j], axw,
                )
            t_total = torch.clamp(t_total, -1.0e4, 1.0e4)
            f_sub = torch.clamp(x_err * ake + v_err * akd, -1.0e4, 1.0e4)
            f_total = f_total + f_sub
        else:
            raise NotImplementedError("joint type %d" % ty)
        if par >= 0:  # :448-449
            out[par] = out[par] + torch.cat([t_total + cross(r_p, f_total), f_total], -1)
        out[i] = out[i] - torch.cat([t_total + cross(r_c, f_total), f_total], -1)  # :451
    return torch.stack(out, 1)


# ------------------------------------------------------------------ integrate
def integrate_bodies(T, body_q, body_qd, body_f, inv_m, I, inv_I, dt):
    """integrator_euler.py:21-91.  inv_m [bs,nb], I / inv_I [bs,nb,3,3]."""
    x0, r0 = body_q[..., :3], body_q[..., 3:]
    w0, v0 = body_qd[..., :3], body_qd[..., 3:]
    t0, f0 = body_f[..., :3], body_f[..., 3:]
    com = T.com[None].expand_as(x0)
    x_com = x0 + q_rot(r0, com)
    nz = (inv_m != 0).to(T.dtype)
    v1 = v0 + (f0 * inv_m[..., None] + T.gravity * nz[..., None]) * dt
    x1 = x_com + v1 * dt
    wb = q_rot_inv(r0, w0)
    Iwb = (I @ wb[..., None])[..., 0]
    tb = q_rot_inv(r0, t0) - cross(wb, Iwb)
    w1 = q_rot(r0, wb + (inv_I @ tb[..., None])[..., 0] * dt)
    w1q = torch.cat([w1, torch.zeros_like(w1[..., :1])], -1)
    r1 = q_normalize(r0 + q_mul(w1q, r0) * 0.5 * dt)
    w1 = w1 * (1.0 - 0.1 * dt)
    w1 = torch.clamp(w1, -10.0, 10.0)
    v1 = torch.clamp(v1, -10.0, 10.0)
    q_new = torch.cat([x1 - q_rot(r1, com), r1], -1)
    qd_new = torch.cat([w1, v1], -1)
    return q_new, qd_new


# -------------------------------------------------------------------- rollout
def simulate_step(T, body_q, body_qd, res_f_t, target, act, target_ke, target_kd, inv_m, I, inv_I, dt):
    """clear_forces + wp_add + compute_forces + integrate (dp_model.py:1210-1228).
    Returns (q_new, qd_new, grf, jaf)."""
    body_f = res_f_t  # zero + res_f
    body_f = eval_body_contacts(T, body_q, body_qd, body_f)
    grf = body_f
    body_f = eval_body_joints(T, body_q, body_qd, body_f, target, act, target_ke, target_kd)
    jaf = body_f - grf
    q_new, qd_new = integrate_bodies(T, body_q, body_qd, body_f, inv_m, I, inv_I, dt)
    return q_new, qd_new, grf, jaf


def rollout(T, q_init, qd_init, torques, res_f, refs, target_ke, target_kd, body_mass, body_inv_mass,
            body_inertia, body_inv_inertia, nsteps, frame2step, dt, return_all=False):
    """ForwardWarp.forward (dp_model.py:1146-1249) on flat, env-major inputs:
    q_init [bs*nq], qd_init [bs*nqd], torques/refs [T,bs*nqd], res_f [T,bs*nb,6], gains [bs*nqd],
    masses [bs*nb], inertias [bs*nb,3,3].  Returns wp_pos [F,bs*nb,7], wp_vel [F,bs*nb,6], grfs, jafs."""
    nb, nq, nqd = T.nb, T.nq, T.nqd
    bs = q_init.numel() // nq
    body_q, body_qd = eval_fk(T, q_init.view(bs, nq), qd_init.view(bs, nqd))
    ke, kd = target_ke.view(bs, nqd), target_kd.view(bs, nqd)
    inv_m = body_inv_mass.view(bs, nb)
    I, inv_I = body_inertia.view(bs, nb, 3, 3), body_inv_inertia.view(bs, nb, 3, 3)
    _ = body_mass  # loaded but unused by the kernel (integrator_euler.py:43)
    pos, vel, grfs, jafs, allq, allqd = [], [], [], [], [], []
    for step in range(nsteps):
        if return_all:
            allq.append(body_q)
            allqd.append(body_qd)
        if step in frame2step:
            pos.append(body_q.reshape(bs * nb, 7))
            vel.append(body_qd.reshape(bs * nb, 6))
        q_new, qd_new, grf, jaf = simulate_step(
            T, body_q, body_qd, res_f[step].view(bs, nb, 6), refs[step].view(bs, nqd), torques[step].view(bs, nqd),
            ke, kd, inv_m, I, inv_I, dt,
        )
        if step in frame2step:
            grfs.append(grf.reshape(bs * nb, 6))
            jafs.append(jaf.reshape(bs * nb, 6))
        body_q, body_qd = q_new, qd_new
    if return_all:
        allq.append(body_q)
        allqd.append(body_qd)
        return torch.stack(allq, 0), torch.stack(allqd, 0)
    return torch.stack(pos, 0), torch.stack(vel, 0), torch.stack(grfs, 0), torch.stack(jafs, 0)


def fk_frames(T, rj_q, rj_qd):
    """ForwardKinematics.forward (dp_model.py:1022-1084): rj_q [F,bs,nq], rj_qd [F,bs,nqd]
    -> body_q [bs,F,nb,7], body_qd [bs,F,nb,6]."""
    F, bs, _ = rj_q.shape
    bq, bqd = eval_fk(T, rj_q.reshape(F * bs, -1), rj_qd.reshape(F * bs, -1))
    return bq.view(F, bs, T.nb, 7).permute(1, 0, 2, 3), bqd.view(F, bs, T.nb, 6).permute(1, 0, 2, 3)
