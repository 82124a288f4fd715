"""Minimal OBJ / STL readers for collision geometry (no trimesh dependency).

The reference reaches these files through urdfpy -> trimesh
(/root/reference/diffphys/import_urdf.py:78-92); only vertex positions and
triangle indices are consumed there, so that is all that is read here.
Duplicate vertex positions are merged (trimesh's default ``process=True``
does the same), which is what makes Laikago's collision set 3 838 points
(SURVEY.md section 8 robot table).
"""
import struct

import numpy as np


def _merge_vertices(vertices, faces):
    vertices = np.asarray(vertices, dtype=np.float64).reshape(-1, 3)
    faces = np.asarray(faces, dtype=np.int64).reshape(-1, 3)
    if len(vertices) == 0:
        return vertices, faces
    # first-occurrence order is kept so the contact order is stable
    _, first, inverse = np.unique(vertices, axis=0, return_index=True, return_inverse=True)
    inverse = np.asarray(inverse).reshape(-1)
    order = np.argsort(first)
    rank = np.empty_like(order)
    rank[order] = np.arange(len(order))
    new_vertices = vertices[first[order]]
    new_faces = rank[inverse[faces]]
    # drop degenerate triangles created by the merge
    keep = (
        (new_faces[:, 0] != new_faces[:, 1])
        & (new_faces[:, 1] != new_faces[:, 2])
        & (new_faces[:, 0] != new_faces[:, 2])
    )
    return new_vertices, new_faces[keep]


def load_obj(path):
    verts, faces = [], []
    with open(path, "r", errors="replace") as f:
        for line in f:
            if line.startswith("v "):
                p = line.split()
                verts.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("f "):
                idx = []
                for tok in line.split()[1:]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                for k in range(1, len(idx) - 1):  # fan-triangulate polygons
                    faces.append((idx[0], idx[k], idx[k + 1]))
    return np.asarray(verts, dtype=np.float64), np.asarray(faces, dtype=np.int64)


def load_stl(path):
    with open(path, "rb") as f:
        data = f.read()
    if len(data) >= 84:
        (ntri,) = struct.unpack_from("<I", data, 80)
        if 84 + 50 * ntri == len(data):
            rec = np.frombuffer(data, dtype=np.uint8, offset=84).reshape(ntri, 50)
            tri = rec[:, 12:48].copy().view("<f4").reshape(ntri, 3, 3)
            verts = tri.reshape(-1, 3).astype(np.float64)
            faces = np.arange(3 * ntri, dtype=np.int64).reshape(ntri, 3)
            return verts, faces
    # ascii STL
    verts = []
    for line in data.decode("ascii", errors="replace").splitlines():
        p = line.split()
        if len(p) == 4 and p[0] == "vertex":
            verts.append((float(p[1]), float(p[2]), float(p[3])))
    verts = np.asarray(verts, dtype=np.float64)
    faces = np.arange(len(verts), dtype=np.int64).reshape(-1, 3)
    return verts, faces


def load_mesh(path, merge=True):
    """Returns (vertices [V,3] f64, faces [F,3] i64)."""
    low = path.lower()
    if low.endswith(".obj"):
        v, f = load_obj(path)
    elif low.endswith(".stl"):
        v, f = load_stl(path)
    else:
        raise ValueError("unsupported mesh format: %s" % path)
    if merge:
        v, f = _merge_vertices(v, f)
    return v, f
