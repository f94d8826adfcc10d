"""Mesh ingest for the model compiler: binary STL -> scaled triangles -> volume,
centre of mass, principal inertia, convex hull + vertex adjacency graph.

Restates what MuJoCo 2.1.0's compiler does to a `<mesh file="*.stl">` asset
(the dialect is fixed by reference src/Mujoco_XML.py:73,195
and output/DexterousDynamos.xml:7,10-25).  MuJoCo itself is not available
(SURVEY.md §8c), so every convention here is [EXT] and behind a named switch:

* ``inertia_mode="legacy"`` (default): MuJoCo <= 2.1.1 pyramid algorithm —
  pyramids from each face to the area-weighted face centroid, |volume| taken
  per pyramid, which over-counts for non-convex meshes.
* ``inertia_mode="exact"``: signed tetrahedra (exact for any closed mesh).
"""
import os
import struct

import numpy as np

from .mathutil import principal_axes, quat_to_mat


def read_stl(path):
    """Return float64 triangles [n,3,3] from a binary (or ASCII) STL file."""
    with open(path, "rb") as f:
        raw = f.read()
    if len(raw) >= 84:
        (ntri,) = struct.unpack_from("<I", raw, 80)
        if 84 + 50 * ntri == len(raw):
            rec = np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("attr", "<u2")])
            arr = np.frombuffer(raw, dtype=rec, count=ntri, offset=84)
            return arr["v"].astype(np.float64)
    # ASCII fallback
    txt = raw.decode("ascii", errors="replace")
    verts = []
    for line in txt.splitlines():
        s = line.strip().split()
        if len(s) == 4 and s[0] == "vertex":
            verts.append([float(s[1]), float(s[2]), float(s[3])])
    if not verts or len(verts) % 3:
        raise ValueError(f"{path}: not a valid STL file")
    return np.asarray(verts, dtype=np.float64).reshape(-1, 3, 3)


def write_binary_stl(path, tris):
    """Write float32 binary STL (used by tests and the synthetic fixtures)."""
    tris = np.asarray(tris, dtype=np.float32).reshape(-1, 3, 3)
    n = np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0])
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    n = np.where(ln > 0, n / np.maximum(ln, 1e-30), 0).astype(np.float32)
    rec = np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("attr", "<u2")])
    arr = np.zeros(len(tris), dtype=rec)
    arr["n"] = n
    arr["v"] = tris
    with open(path, "wb") as f:
        f.write(b"mujoco_amd synthetic STL".ljust(80, b" "))
        f.write(struct.pack("<I", len(tris)))
        f.write(arr.tobytes())


def _products_of_inertia(D, E, F, vol):
    """Sum over tetrahedra (origin, D, E, F) of the second-moment integrals."""
    P = np.zeros(6)
    pairs = [(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)]
    for j, (a, b) in enumerate(pairs):
        t = (2.0 * (D[:, a] * D[:, b] + E[:, a] * E[:, b] + F[:, a] * F[:, b])
             + D[:, a] * E[:, b] + D[:, b] * E[:, a]
             + D[:, a] * F[:, b] + D[:, b] * F[:, a]
             + E[:, a] * F[:, b] + E[:, b] * F[:, a])
        P[j] = np.sum(vol * t) / 20.0
    return P


def mesh_inertia(tris, mode="legacy"):
    """Volume, centre of mass and unit-density inertia tensor about the COM.

    Returns (volume, com[3], I[3,3]).
    """
    v0, v1, v2 = tris[:, 0], tris[:, 1], tris[:, 2]
    nrm = np.cross(v1 - v0, v2 - v0)
    nlen = np.linalg.norm(nrm, axis=1)
    area = 0.5 * nlen
    good = nlen > 0
    nrm = np.where(good[:, None], nrm / np.maximum(nlen, 1e-300)[:, None], 0.0)
    cen = (v0 + v1 + v2) / 3.0
    tot_area = area.sum()
    if tot_area <= 0:
        raise ValueError("mesh has zero surface area")
    facecen = (area[:, None] * cen).sum(axis=0) / tot_area

    if mode == "exact":
        apex = np.zeros(3)
    elif mode == "legacy":
        apex = facecen
    else:
        raise ValueError(f"unknown inertia_mode {mode!r}")

    vol = area * np.einsum("ij,ij->i", cen - apex, nrm) / 3.0
    if mode == "legacy":
        vol = np.abs(vol)
    volume = vol.sum()
    if volume <= 1e-15 * max(1.0, tot_area ** 1.5):
        raise ValueError("mesh volume is too small (open or degenerate surface)")
    # centroid of pyramid (apex, face) = 3/4 face centre + 1/4 apex
    com = (vol[:, None] * (0.75 * cen + 0.25 * apex)).sum(axis=0) / volume

    D, E, F = v0 - com, v1 - com, v2 - com
    cen_c = cen - com
    vol2 = area * np.einsum("ij,ij->i", cen_c, nrm) / 3.0
    if mode == "legacy":
        vol2 = np.abs(vol2)
    P = _products_of_inertia(D, E, F, vol2)
    I = np.array([
        [P[1] + P[2], -P[3], -P[4]],
        [-P[3], P[0] + P[2], -P[5]],
        [-P[4], -P[5], P[0] + P[1]],
    ])
    return float(volume), com, I


def convex_hull(points):
    """Convex hull of a point set: (hull_verts[nh,3], faces[nf,3] (hull-local ids),
    adj_start[nh+1], adj[...]) — CSR vertex adjacency for hill-climbing support."""
    from scipy.spatial import ConvexHull

    pts = np.unique(np.ascontiguousarray(points, dtype=np.float64), axis=0)
    hull = ConvexHull(pts)
    vid = np.sort(hull.vertices)
    remap = -np.ones(len(pts), dtype=np.int64)
    remap[vid] = np.arange(len(vid))
    faces = remap[hull.simplices]
    assert (faces >= 0).all()
    nh = len(vid)
    edges = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], axis=0)
    edges = np.concatenate([edges, edges[:, ::-1]], axis=0)
    edges = np.unique(edges, axis=0)
    counts = np.bincount(edges[:, 0], minlength=nh)
    adj_start = np.zeros(nh + 1, dtype=np.int32)
    np.cumsum(counts, out=adj_start[1:])
    adj = edges[:, 1].astype(np.int32)
    # orient faces outward (useful for debugging / plane tests)
    hv = pts[vid]
    c = hv.mean(axis=0)
    n = np.cross(hv[faces[:, 1]] - hv[faces[:, 0]], hv[faces[:, 2]] - hv[faces[:, 0]])
    flip = np.einsum("ij,ij->i", n, hv[faces[:, 0]] - c) < 0
    faces[flip] = faces[flip][:, ::-1]
    return hv, faces.astype(np.int32), adj_start, adj


class CompiledMesh:
    """A mesh asset after compilation: expressed in its own principal frame
    centred at the COM; `pos`/`quat` give that frame in the mesh's original
    (file * scale) coordinates so geoms can compensate."""

    def __init__(self, name, tris, inertia_mode="legacy", need_hull=True):
        self.name = name
        volume, com, I = mesh_inertia(tris, inertia_mode)
        moments, quat = principal_axes(I)
        R = quat_to_mat(quat)
        self.volume = volume
        self.pos = com
        self.quat = quat
        self.inertia_unit = moments  # principal moments at unit density
        local = (tris.reshape(-1, 3) - com) @ R  # rows: R^T (v - com)
        self.ntri = len(tris)
        self.aabb_min = local.min(axis=0)
        self.aabb_max = local.max(axis=0)
        if need_hull:
            hv, faces, adj_start, adj = convex_hull(local)
            self.hull_vert = hv
            self.hull_face = faces
            self.hull_adj_start = adj_start
            self.hull_adj = adj
            self.rbound = float(np.sqrt((hv * hv).sum(axis=1).max()))
        else:
            self.hull_vert = np.zeros((0, 3))
            self.hull_face = np.zeros((0, 3), dtype=np.int32)
            self.hull_adj_start = np.zeros(1, dtype=np.int32)
            self.hull_adj = np.zeros(0, dtype=np.int32)
            self.rbound = float(np.sqrt((local * local).sum(axis=1).max()))


def load_mesh(name, path, scale, inertia_mode="legacy", need_hull=True):
    if not os.path.exists(path):
        raise FileNotFoundError(f"mesh '{name}': file not found: {path}")
    tris = read_stl(path) * np.asarray(scale, dtype=np.float64)
    if np.prod(scale) < 0:
        tris = tris[:, ::-1, :]
    return CompiledMesh(name, tris, inertia_mode, need_hull)


def fibonacci_directions(n):
    """n roughly uniform unit vectors (Fibonacci lattice)."""
    k = np.arange(n) + 0.5
    phi = np.arccos(1 - 2 * k / n)
    theta = np.pi * (1 + 5 ** 0.5) * k
    return np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], axis=1)


def simplify_hull(hull_verts, max_verts=128):
    """Inner approximation of a convex hull with at most `max_verts` vertices: the support vertices
    of the hull along `max_verts` evenly spread directions (plus the six axis extremes), re-hulled.
    Every kept vertex is a vertex of the original, so the simplified hull lies inside it: contacts
    can only appear later / be shallower, never earlier.  Returns (verts, faces, adj_start, adj)."""
    hv = np.asarray(hull_verts, dtype=np.float64)
    if len(hv) <= max_verts:
        return convex_hull(hv)
    n = max_verts
    while True:
        dirs = np.concatenate([fibonacci_directions(max(n - 6, 4)), np.eye(3), -np.eye(3)])
        keep = np.unique(np.argmax(hv @ dirs.T, axis=0))
        out = convex_hull(hv[keep])
        if len(out[0]) <= max_verts or n <= 8:
            return out
        n -= max(1, (len(out[0]) - max_verts))


def simplify_model_hulls(model, max_verts=128):
    """SURVEY.md §8 f-3 (the GPU-side analogue of the reference's `reduce_mesh`, src/utils.py:8-74):
    an OPT-IN accuracy/throughput knob.  Replaces every collision hull of `model` by its
    `simplify_hull`; inertias, frames and bounding radii stay those of the full meshes.  Changes the
    contact geometry, so it is off for every parity run and reported separately."""
    m = model.copy()
    hv, hs, ha, hf = [], [], [], []
    vadr = aadr = fadr = 0
    m.mesh_vertadr, m.mesh_vertnum = m.mesh_vertadr.copy(), m.mesh_vertnum.copy()
    m.mesh_adjadr, m.mesh_faceadr, m.mesh_facenum = m.mesh_adjadr.copy(), m.mesh_faceadr.copy(), m.mesh_facenum.copy()
    for i in range(len(model.mesh_vertadr)):
        a, n = int(model.mesh_vertadr[i]), int(model.mesh_vertnum[i])
        v, f, s0, adj = simplify_hull(model.hull_vert[a:a + n], max_verts)
        m.mesh_vertadr[i], m.mesh_vertnum[i], m.mesh_adjadr[i] = vadr, len(v), aadr
        m.mesh_faceadr[i], m.mesh_facenum[i] = fadr, len(f)
        hv.append(v); hs.append(s0[:-1] + aadr); ha.append(adj); hf.append(f)
        vadr += len(v); aadr += len(adj); fadr += len(f)
    m.hull_vert = np.concatenate(hv) if hv else np.zeros((0, 3))
    m.hull_adj_start = np.concatenate(hs + [np.array([aadr])]).astype(np.int32)
    m.hull_adj = np.concatenate(ha).astype(np.int32) if ha else np.zeros(0, dtype=np.int32)
    m.hull_face = np.concatenate(hf).astype(np.int32) if hf else np.zeros((0, 3), dtype=np.int32)
    return m
