"""Model compiler: the subset of ``warp.sim`` the reference's host code drives.

Replaces (call sites in the reference):
  * ``wp.sim.ModelBuilder`` / ``add_body`` / ``add_shape_*`` / ``add_rigid_articulation``
    / ``finalize``  -- /root/reference/diffphys/dp_model.py:126-146,247-250,384-389,
    /root/reference/diffphys/import_urdf.py:36-103,145-284
  * ``Model.state`` / ``Model.collide`` / ``State.clear_forces``
    -- /root/reference/diffphys/dp_model.py:396-401,1046,1210

warp_lang 0.7.2 is not vendored in the reference and not installable here, so
the construction rules follow SURVEY.md Appendix A.2 (marked [RECALL] there).

Design difference from Warp: the finalized :class:`Model` keeps ONE articulation
template plus ``num_envs``; nothing is tiled per env.  The HIP kernels index
``env = segment id`` and read the shared template from LDS / L2.
"""
import math

import numpy as np

# joint / geometry type codes (Warp values)
JOINT_PRISMATIC = 0
JOINT_REVOLUTE = 1
JOINT_BALL = 2
JOINT_FIXED = 3
JOINT_FREE = 4
JOINT_COMPOUND = 5
JOINT_UNIVERSAL = 6

GEO_SPHERE = 0
GEO_BOX = 1
GEO_CAPSULE = 2
GEO_MESH = 3

_DOF_COORD = {
    JOINT_PRISMATIC: (1, 1),
    JOINT_REVOLUTE: (1, 1),
    JOINT_BALL: (3, 4),
    JOINT_FIXED: (0, 0),
    JOINT_FREE: (6, 7),
    JOINT_COMPOUND: (3, 3),
    JOINT_UNIVERSAL: (2, 2),
}


# ----------------------------------------------------------------------------
# small numpy helpers with Warp conventions: quat = (x, y, z, w), transform = (p, q)
# ----------------------------------------------------------------------------
def quat_identity():
    return np.array([0.0, 0.0, 0.0, 1.0])


def quat_from_axis_angle(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    s = math.sin(angle * 0.5)
    return np.array([axis[0] * s, axis[1] * s, axis[2] * s, math.cos(angle * 0.5)])


def quat_rpy(roll, pitch, yaw):
    cy, sy = math.cos(yaw * 0.5), math.sin(yaw * 0.5)
    cr, sr = math.cos(roll * 0.5), math.sin(roll * 0.5)
    cp, sp = math.cos(pitch * 0.5), math.sin(pitch * 0.5)
    w = cy * cr * cp + sy * sr * sp
    x = cy * sr * cp - sy * cr * sp
    y = cy * cr * sp + sy * sr * cp
    z = sy * cr * cp - cy * sr * sp
    return np.array([x, y, z, w])


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array(
        [
            aw * bx + bw * ax + ay * bz - az * by,
            aw * by + bw * ay + az * bx - ax * bz,
            aw * bz + bw * az + ax * by - ay * bx,
            aw * bw - ax * bx - ay * by - az * bz,
        ]
    )


def quat_to_matrix(q):
    x, y, z, w = q
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
        ]
    )


def quat_rotate(q, v):
    return quat_to_matrix(q) @ np.asarray(v, dtype=np.float64)


class transform:
    """(p, q) pair; mirrors ``wp.transform`` enough for parse_urdf's ``xform.p`` / ``xform.q``."""

    def __init__(self, p=(0.0, 0.0, 0.0), q=(0.0, 0.0, 0.0, 1.0)):
        self.p = np.asarray(p, dtype=np.float64).copy()
        self.q = np.asarray(q, dtype=np.float64).copy()

    def as7(self):
        return np.concatenate([self.p, self.q])


def transform_identity():
    return transform()


def transform_point(t, x):
    return t.p + quat_rotate(t.q, x)


def _transform_inertia(m, I, p, q):
    R = quat_to_matrix(q)
    p = np.asarray(p, dtype=np.float64)
    return R @ I @ R.T + m * (np.dot(p, p) * np.eye(3) - np.outer(p, p))


class Mesh:
    """Triangle mesh with density-1 mass properties (SURVEY.md Appendix A.2:
    4-point tetrahedral quadrature about the vertex mean)."""

    def __init__(self, vertices, indices, compute_inertia=True):
        self.vertices = np.asarray(vertices, dtype=np.float64).reshape(-1, 3)
        self.indices = np.asarray(indices, dtype=np.int64).reshape(-1)
        self.mass = 1.0
        self.com = np.zeros(3)
        self.I = np.eye(3)
        if compute_inertia and len(self.indices) > 0:
            com = self.vertices.mean(0)
            tri = self.vertices[self.indices.reshape(-1, 3)]  # [F,3,3]
            p, q, r = tri[:, 0], tri[:, 1], tri[:, 2]
            vol = np.linalg.det(np.stack([p - com, q - com, r - com], -1)) / 6.0  # [F]
            mid = (com[None] + p + q + r) / 4.0
            alpha = math.sqrt(5.0) / 5.0
            I = np.zeros((3, 3))
            for v in (p, q, r, np.broadcast_to(com, p.shape)):
                d = mid + (v - mid) * alpha - com[None]
                dd = (d * d).sum(-1)
                I += 0.25 * (
                    (vol * dd).sum() * np.eye(3) - np.einsum("f,fi,fj->ij", vol, d, d)
                )
            self.I = I
            self.mass = float(vol.sum())
            self.com = com


# ----------------------------------------------------------------------------
class ModelBuilder:
    """List-based builder; the lists the reference mutates in place
    (``body_mass``, ``body_inertia``, ``shape_geo_scale``, ``joint_target_ke/kd``,
    /root/reference/diffphys/dp_model.py:166-205) are plain Python lists here too."""

    def __init__(self):
        self.body_mass = []
        self.body_inertia = []
        self.body_com = []
        self.body_q = []
        self.body_qd = []
        self.joint_type = []
        self.joint_parent = []
        self.joint_X_p = []
        self.joint_X_c = []
        self.joint_axis = []
        self.joint_armature = []
        self.joint_q = []
        self.joint_qd = []
        self.joint_act = []
        self.joint_target = []
        self.joint_target_ke = []
        self.joint_target_kd = []
        self.joint_limit_lower = []
        self.joint_limit_upper = []
        self.joint_limit_ke = []
        self.joint_limit_kd = []
        self.joint_q_start = []
        self.joint_qd_start = []
        self.joint_dof_count = 0
        self.joint_coord_count = 0
        self.shape_body = []
        self.shape_transform = []
        self.shape_geo_type = []
        self.shape_geo_scale = []
        self.shape_geo_src = []
        self.shape_materials = []
        self.articulation_start = []
        # articulation replication (add_rigid_articulation)
        self._template = None
        self._num_envs = 0

    # -- articulation / bodies ------------------------------------------------
    def add_articulation(self):
        self.articulation_start.append(len(self.joint_type))

    def add_body(
        self,
        origin,
        parent=-1,
        joint_xform=None,
        joint_xform_child=None,
        joint_axis=(0.0, 0.0, 0.0),
        joint_type=JOINT_FREE,
        joint_target_ke=0.0,
        joint_target_kd=0.0,
        joint_limit_ke=100.0,
        joint_limit_kd=10.0,
        joint_limit_lower=-1.0e3,
        joint_limit_upper=1.0e3,
        joint_armature=0.0,
        com=np.zeros(3),
        I_m=np.zeros((3, 3)),
        m=0.0,
    ):
        child = len(self.body_mass)
        self.body_inertia.append(np.asarray(I_m, dtype=np.float64) + np.eye(3) * joint_armature)
        self.body_mass.append(float(m))
        self.body_com.append(np.asarray(com, dtype=np.float64).copy())
        self.body_q.append(origin if origin is not None else transform_identity())
        self.body_qd.append(np.zeros(6))

        self.joint_type.append(int(joint_type))
        self.joint_parent.append(int(parent))
        self.joint_X_p.append(joint_xform if joint_xform is not None else transform_identity())
        self.joint_X_c.append(joint_xform_child if joint_xform_child is not None else transform_identity())
        self.joint_armature.append(joint_armature)
        self.joint_axis.append(np.asarray(joint_axis, dtype=np.float64).copy())

        dof_count, coord_count = _DOF_COORD[int(joint_type)]
        rs = lambda v: np.resize(np.atleast_1d(np.asarray(v, dtype=np.float64)), dof_count)
        ke, kd = rs(joint_target_ke), rs(joint_target_kd)
        lke, lkd = rs(joint_limit_ke), rs(joint_limit_kd)
        lo, up = rs(joint_limit_lower), rs(joint_limit_upper)
        for _ in range(coord_count):
            self.joint_q.append(0.0)
        for i in range(dof_count):
            self.joint_qd.append(0.0)
            self.joint_act.append(0.0)
            self.joint_target.append(0.0)
            self.joint_target_ke.append(float(ke[i]))
            self.joint_target_kd.append(float(kd[i]))
            self.joint_limit_lower.append(float(lo[i]))
            self.joint_limit_upper.append(float(up[i]))
            self.joint_limit_ke.append(float(lke[i]))
            self.joint_limit_kd.append(float(lkd[i]))
        if joint_type in (JOINT_FREE, JOINT_BALL):
            self.joint_q[-1] = 1.0
        self.joint_q_start.append(self.joint_coord_count)
        self.joint_qd_start.append(self.joint_dof_count)
        self.joint_dof_count += dof_count
        self.joint_coord_count += coord_count
        return child

    # -- shapes -----------------------------------------------------------------
    def add_shape_sphere(self, body, pos=(0, 0, 0), rot=(0, 0, 0, 1), radius=1.0, density=1000.0, ke=1e5, kd=1e3, kf=1e3, mu=0.5):
        self._add_shape(body, pos, rot, GEO_SPHERE, (radius, 0.0, 0.0, 0.0), None, density, ke, kd, kf, mu)

    def add_shape_box(self, body, pos=(0, 0, 0), rot=(0, 0, 0, 1), hx=0.5, hy=0.5, hz=0.5, density=1000.0, ke=1e5, kd=1e3, kf=1e3, mu=0.5):
        self._add_shape(body, pos, rot, GEO_BOX, (hx, hy, hz, 0.0), None, density, ke, kd, kf, mu)

    def add_shape_capsule(self, body, pos=(0, 0, 0), rot=(0, 0, 0, 1), radius=1.0, half_width=0.5, density=1000.0, ke=1e5, kd=1e3, kf=1e3, mu=0.5):
        self._add_shape(body, pos, rot, GEO_CAPSULE, (radius, half_width, 0.0, 0.0), None, density, ke, kd, kf, mu)

    def add_shape_mesh(self, body, pos=(0, 0, 0), rot=(0, 0, 0, 1), mesh=None, scale=(1.0, 1.0, 1.0), density=1000.0, ke=1e5, kd=1e3, kf=1e3, mu=0.5):
        self._add_shape(body, pos, rot, GEO_MESH, (scale[0], scale[1], scale[2], 0.0), mesh, density, ke, kd, kf, mu)

    def _add_shape(self, body, pos, rot, type, scale, src, density, ke, kd, kf, mu):
        self.shape_body.append(body)
        self.shape_transform.append(transform(pos, rot))
        self.shape_geo_type.append(type)
        self.shape_geo_scale.append((scale[0], scale[1], scale[2]))
        self.shape_geo_src.append(src)
        self.shape_materials.append((ke, kd, kf, mu))
        m, I = self._compute_shape_mass(type, scale, src, density)
        self._update_body_mass(body, m, I, np.asarray(pos, dtype=np.float64), np.asarray(rot, dtype=np.float64))

    @staticmethod
    def _compute_shape_mass(type, scale, src, density):
        if density == 0.0:
            return 0.0, np.zeros((3, 3))
        if type == GEO_SPHERE:
            r = scale[0]
            m = density * (4.0 / 3.0) * math.pi * r ** 3
            return m, np.eye(3) * (2.0 / 5.0) * m * r * r
        if type == GEO_BOX:
            w, h, d = scale[0] * 2.0, scale[1] * 2.0, scale[2] * 2.0
            m = density * w * h * d
            return m, np.diag([m / 12.0 * (h * h + d * d), m / 12.0 * (w * w + d * d), m / 12.0 * (w * w + h * h)])
        if type == GEO_CAPSULE:
            r, l = scale[0], scale[1] * 2.0
            ms = density * (4.0 / 3.0) * math.pi * r ** 3
            mc = density * math.pi * r * r * l
            Ia = mc * (0.25 * r * r + l * l / 12.0) + ms * (0.4 * r * r + 0.375 * r * l + 0.25 * l * l)
            Ib = (mc * 0.5 + ms * 0.4) * r * r
            return ms + mc, np.diag([Ib, Ia, Ia])
        if type == GEO_MESH:
            s = scale[0]
            return density * src.mass * s ** 3, density * src.I * s ** 5
        raise ValueError(type)

    def _update_body_mass(self, i, m, I, p, q):
        if i == -1:
            return
        new_mass = self.body_mass[i] + m
        if new_mass == 0.0:
            return
        new_com = (self.body_com[i] * self.body_mass[i] + p * m) / new_mass
        com_offset = new_com - self.body_com[i]
        shape_offset = new_com - p
        new_inertia = _transform_inertia(self.body_mass[i], self.body_inertia[i], com_offset, quat_identity()) + _transform_inertia(
            m, I, shape_offset, q
        )
        self.body_mass[i] = new_mass
        self.body_inertia[i] = new_inertia
        self.body_com[i] = new_com

    # -- replication / finalize -----------------------------------------------
    def add_rigid_articulation(self, articulation):
        """The reference tiles the template ``num_envs`` times
        (/root/reference/diffphys/dp_model.py:385-386); here it is counted."""
        if self._template is None:
            self._template = articulation
        elif self._template is not articulation:
            raise NotImplementedError("all envs must share one articulation template")
        self._num_envs += 1

    def finalize(self, device="cuda"):
        tpl, n = (self, 1) if self._template is None else (self._template, self._num_envs)
        return Model(tpl, n, device)


# ----------------------------------------------------------------------------
class State:
    """Per-step maximal-coordinate state (row a1 of SURVEY.md section 8):
    ``body_q [bs*nb,7]``, ``body_qd [bs*nb,6]`` (w,v), ``body_f [bs*nb,6]`` (tau,f)."""

    def __init__(self, model, requires_grad=False):
        import torch

        n = model.body_count
        self.body_count = n
        self.particle_count = 0
        dev = model.device
        self.body_q = torch.zeros(n, 7, dtype=torch.float32, device=dev)
        self.body_q[:, 6] = 1.0
        self.body_qd = torch.zeros(n, 6, dtype=torch.float32, device=dev)
        self.body_f = torch.zeros(n, 6, dtype=torch.float32, device=dev)
        self.requires_grad = requires_grad

    def clear_forces(self):
        self.body_f.zero_()


class Model:
    """One articulation template + ``num_envs``.

    Flat template arrays (float32 / int32 numpy, see :meth:`template`) are what
    the C-ABI consumes (include/ppr_diffphys.h).  Per-env attributes the
    reference's host code reads (``body_com``, ``body_mass``; dp_model.py:852-853)
    are exposed tiled on demand.
    """

    def __init__(self, b, num_envs, device):
        self.device = device
        self.num_envs = int(num_envs)
        nb = len(b.body_mass)
        self.nb = nb
        self.nq = b.joint_coord_count
        self.nqd = b.joint_dof_count
        self.body_count = nb * self.num_envs
        self.articulation_count = self.num_envs
        self.ground = True
        self.gravity = np.array([0.0, -9.80665, 0.0], dtype=np.float32)
        self.joint_attach_ke = 1.0e3
        self.joint_attach_kd = 1.0e2

        f32 = lambda x: np.ascontiguousarray(np.asarray(x, dtype=np.float32))
        i32 = lambda x: np.ascontiguousarray(np.asarray(x, dtype=np.int32))
        self.t_joint_type = i32(b.joint_type)
        self.t_joint_parent = i32(b.joint_parent)
        self.t_joint_q_start = i32(b.joint_q_start)
        self.t_joint_qd_start = i32(b.joint_qd_start)
        self.t_joint_X_p = f32([t.as7() for t in b.joint_X_p]).reshape(nb, 7)
        self.t_joint_X_c = f32([t.as7() for t in b.joint_X_c]).reshape(nb, 7)
        self.t_joint_axis = f32(b.joint_axis).reshape(nb, 3)
        self.t_body_com = f32(b.body_com).reshape(nb, 3)
        self.t_body_mass = f32(b.body_mass)
        self.t_body_inertia = f32(b.body_inertia).reshape(nb, 3, 3)
        self.t_joint_q = f32(b.joint_q)
        self.t_joint_target_ke = f32(b.joint_target_ke)
        self.t_joint_target_kd = f32(b.joint_target_kd)
        self.t_joint_limit_lower = f32(b.joint_limit_lower)
        self.t_joint_limit_upper = f32(b.joint_limit_upper)
        self.t_joint_limit_ke = f32(b.joint_limit_ke)
        self.t_joint_limit_kd = f32(b.joint_limit_kd)
        for i in range(nb):
            p = int(self.t_joint_parent[i])
            if p >= i:
                raise ValueError("parents must precede children (body %d has parent %d)" % (i, p))

        self._shapes = dict(
            body=list(b.shape_body),
            xform=[transform(t.p, t.q) for t in b.shape_transform],
            type=list(b.shape_geo_type),
            scale=[tuple(s) for s in b.shape_geo_scale],
            src=list(b.shape_geo_src),
            materials=f32(b.shape_materials).reshape(-1, 4),
        )
        self.t_shape_materials = self._shapes["materials"]
        self.t_contact_body = np.zeros(0, np.int32)
        self.t_contact_point = np.zeros((0, 3), np.float32)
        self.t_contact_dist = np.zeros(0, np.float32)
        self.t_contact_material = np.zeros(0, np.int32)
        self.contact_count = 0
        self._handle = None  # device-side template, owned by diffphys_amd.hip_backend

    # -- Warp-compatible surface ----------------------------------------------
    def state(self, requires_grad=False):
        return State(self, requires_grad)

    def collide(self, state=None):
        """Ground-contact candidates in body frame:
        sphere 1 point (dist=radius), capsule 2, box 8 corners, mesh one per vertex."""
        s = self._shapes
        body, point, dist, mat = [], [], [], []

        def add(b, t, p, d, m):
            body.append(b)
            point.append(transform_point(t, np.asarray(p, dtype=np.float64)))
            dist.append(d)
            mat.append(m)

        for i in range(len(s["body"])):
            X_bs, ty, sc = s["xform"][i], s["type"][i], s["scale"][i]
            if ty == GEO_SPHERE:
                add(s["body"][i], X_bs, (0.0, 0.0, 0.0), sc[0], i)
            elif ty == GEO_CAPSULE:
                add(s["body"][i], X_bs, (-sc[1], 0.0, 0.0), sc[0], i)
                add(s["body"][i], X_bs, (sc[1], 0.0, 0.0), sc[0], i)
            elif ty == GEO_BOX:
                for sz in (-1.0, 1.0):
                    for sy in (-1.0, 1.0):
                        for sx in (-1.0, 1.0):
                            add(s["body"][i], X_bs, (sx * sc[0], sy * sc[1], sz * sc[2]), 0.0, i)
            elif ty == GEO_MESH:
                for v in s["src"][i].vertices:
                    add(s["body"][i], X_bs, (v[0] * sc[0], v[1] * sc[1], v[2] * sc[2]), 0.0, i)
        self.t_contact_body = np.asarray(body, dtype=np.int32)
        self.t_contact_point = np.asarray(point, dtype=np.float32).reshape(-1, 3)
        self.t_contact_dist = np.asarray(dist, dtype=np.float32)
        self.t_contact_material = np.asarray(mat, dtype=np.int32)
        self.contact_count = len(body) * self.num_envs
        self._handle = None

    # -- template export --------------------------------------------------------
    def template(self):
        """Dict of flat numpy arrays; the serialised form lives in
        ``diffphys_amd/templates/*.npz`` and is what the oracle and the C-ABI take."""
        return dict(
            nb=np.int32(self.nb),
            nq=np.int32(self.nq),
            nqd=np.int32(self.nqd),
            joint_type=self.t_joint_type,
            joint_parent=self.t_joint_parent,
            joint_q_start=self.t_joint_q_start,
            joint_qd_start=self.t_joint_qd_start,
            joint_X_p=self.t_joint_X_p,
            joint_X_c=self.t_joint_X_c,
            joint_axis=self.t_joint_axis,
            body_com=self.t_body_com,
            body_mass=self.t_body_mass,
            body_inertia=self.t_body_inertia,
            joint_q=self.t_joint_q,
            joint_target_ke=self.t_joint_target_ke,
            joint_target_kd=self.t_joint_target_kd,
            joint_limit_lower=self.t_joint_limit_lower,
            joint_limit_upper=self.t_joint_limit_upper,
            joint_limit_ke=self.t_joint_limit_ke,
            joint_limit_kd=self.t_joint_limit_kd,
            contact_body=self.t_contact_body,
            contact_point=self.t_contact_point,
            contact_dist=self.t_contact_dist,
            contact_material=self.t_contact_material,
            shape_materials=self.t_shape_materials,
            gravity=self.gravity,
            joint_attach_ke=np.float32(self.joint_attach_ke),
            joint_attach_kd=np.float32(self.joint_attach_kd),
        )

    @staticmethod
    def from_template(tpl, num_envs, device="cuda"):
        """Rebuild a Model from a dict produced by :meth:`template` (e.g. an npz)."""
        m = Model.__new__(Model)
        m.device = device
        m.num_envs = int(num_envs)
        m.nb, m.nq, m.nqd = int(tpl["nb"]), int(tpl["nq"]), int(tpl["nqd"])
        m.body_count = m.nb * m.num_envs
        m.articulation_count = m.num_envs
        m.ground = True
        m.gravity = np.asarray(tpl["gravity"], dtype=np.float32)
        m.joint_attach_ke = float(tpl["joint_attach_ke"])
        m.joint_attach_kd = float(tpl["joint_attach_kd"])
        for k in (
            "joint_type joint_parent joint_q_start joint_qd_start joint_X_p joint_X_c joint_axis body_com "
            "body_mass body_inertia joint_q joint_target_ke joint_target_kd joint_limit_lower joint_limit_upper "
            "joint_limit_ke joint_limit_kd contact_body contact_point contact_dist contact_material shape_materials"
        ).split():
            setattr(m, "t_" + k, np.ascontiguousarray(tpl[k]))
        m._shapes = None
        m.contact_count = len(m.t_contact_body) * m.num_envs
        m._handle = None
        return m

    # -- tiled views the reference's host code reads ---------------------------
    def _tiled(self, arr):
        import torch

        t = torch.from_numpy(np.tile(arr, (self.num_envs,) + (1,) * (arr.ndim - 1)))
        return t

    @property
    def body_com(self):
        return self._tiled(self.t_body_com)

    @property
    def body_mass(self):
        return self._tiled(self.t_body_mass)

    @property
    def body_inertia(self):
        return self._tiled(self.t_body_inertia)
