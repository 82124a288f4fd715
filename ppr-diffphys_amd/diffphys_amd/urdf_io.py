"""Minimal URDF reader on xml.etree (stands in for urdfpy, which the reference
uses at /root/reference/diffphys/import_urdf.py:122 and is not installable here).

Only what ``parse_urdf`` consumes is kept: link order, per-link collision
elements (origin + geometry), joint order, joint type / parent / child / axis /
origin / limits / damping.  Link and joint order is file order, as in urdfpy.
"""
import os
import xml.etree.ElementTree as ET

import numpy as np

from .mesh_io import load_mesh


def _floats(text, n, default):
    if text is None:
        return np.asarray(default, dtype=np.float64)
    vals = [float(t) for t in text.split()]
    assert len(vals) == n, "expected %d floats, got %r" % (n, text)
    return np.asarray(vals, dtype=np.float64)


class Origin:
    def __init__(self, node):
        self.xyz = np.zeros(3)
        self.rpy = np.zeros(3)
        if node is not None:
            self.xyz = _floats(node.get("xyz"), 3, (0, 0, 0))
            self.rpy = _floats(node.get("rpy"), 3, (0, 0, 0))


class Collision:
    """kind in {box, sphere, cylinder, mesh}."""

    def __init__(self, node, base_dir):
        self.origin = Origin(node.find("origin"))
        geo = node.find("geometry")
        self.kind = None
        self.size = None
        self.radius = None
        self.length = None
        self.mesh_file = None
        self.mesh_scale = np.ones(3)
        self._base_dir = base_dir
        self._mesh = None
        if geo is None:
            return
        if geo.find("box") is not None:
            self.kind = "box"
            self.size = _floats(geo.find("box").get("size"), 3, (0, 0, 0))
        elif geo.find("sphere") is not None:
            self.kind = "sphere"
            self.radius = float(geo.find("sphere").get("radius"))
        elif geo.find("cylinder") is not None:
            self.kind = "cylinder"
            self.radius = float(geo.find("cylinder").get("radius"))
            self.length = float(geo.find("cylinder").get("length"))
        elif geo.find("mesh") is not None:
            self.kind = "mesh"
            m = geo.find("mesh")
            self.mesh_file = m.get("filename")
            if m.get("scale") is not None:
                self.mesh_scale = _floats(m.get("scale"), 3, (1, 1, 1))

    def mesh(self):
        """(vertices, faces) with the URDF <mesh scale> applied.  (urdfpy keeps `scale` as an attribute beside the meshes and the reference's
        parse_urdf reads the meshes only, import_urdf.py:77-90; every URDF the reference ships says scale="1 1 1", where the two readings
        agree -- a user file with another scale gets the geometry it describes.)"""
        if self._mesh is None:
            fn = self.mesh_file
            if fn.startswith("package://"):
                fn = fn[len("package://"):]
            v, f = load_mesh(os.path.join(self._base_dir, fn))
            self._mesh = (v * self.mesh_scale[None], f)
        return self._mesh


class Link:
    def __init__(self, node, base_dir):
        self.name = node.get("name")
        self.collisions = [Collision(c, base_dir) for c in node.findall("collision")]
        inertial = node.find("inertial")
        self.inertial_origin = Origin(inertial.find("origin") if inertial is not None else None)
        self.mass = 0.0
        self.inertia = np.zeros((3, 3))
        if inertial is not None:
            if inertial.find("mass") is not None:
                self.mass = float(inertial.find("mass").get("value"))
            i = inertial.find("inertia")
            if i is not None:
                g = lambda k: float(i.get(k, 0.0))
                self.inertia = np.array(
                    [[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]]
                )


class Joint:
    def __init__(self, node):
        self.name = node.get("name")
        self.joint_type = node.get("type")
        self.parent = node.find("parent").get("link")
        self.child = node.find("child").get("link")
        self.origin = Origin(node.find("origin"))
        ax = node.find("axis")
        self.axis = _floats(ax.get("xyz"), 3, (1, 0, 0)) if ax is not None else np.array([1.0, 0.0, 0.0])
        lim = node.find("limit")
        self.limit_lower = None
        self.limit_upper = None
        if lim is not None:
            # absent lower/upper stay None (urdfpy does the same), so parse_urdf keeps +-1e3
            self.limit_lower = float(lim.get("lower")) if lim.get("lower") is not None else None
            self.limit_upper = float(lim.get("upper")) if lim.get("upper") is not None else None
        dyn = node.find("dynamics")
        self.damping = float(dyn.get("damping", 0.0)) if dyn is not None else None


class URDF:
    def __init__(self, links, joints, path):
        self.links = links
        self.joints = joints
        self.path = path
        self.link_map = {l.name: l for l in links}
        children = {j.child for j in joints}
        roots = [l for l in links if l.name not in children]
        self.base_link = roots[0] if roots else links[0]

    @staticmethod
    def load(path):
        base_dir = os.path.dirname(os.path.abspath(path))
        root = ET.parse(path).getroot()
        links = [Link(n, base_dir) for n in root.findall("link")]
        joints = [Joint(n) for n in root.findall("joint")]
        return URDF(links, joints, path)
