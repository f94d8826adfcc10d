"""MJCF-subset parser + model compiler.

Parses exactly the dialect the reference's exporter emits
(reference src/Mujoco_XML.py:37-52 template, :69-75 defaults, :195 mesh asset,
:212 body, :221 geom, :255 joint, :261 actuator, :267 exclude, :274 joint
equality; instance output/DexterousDynamos.xml:1-167) plus the few extras the
build needs (``<option>``, default classes, primitive/plane geoms, slide joints,
motor/velocity actuators) and flattens it into the `Model` arrays the device
kernels consume.  Ids follow MJCF depth-first document order.

MuJoCo 2.1.0 itself cannot be run here: defaults and compile
conventions are [EXT] — restated from MuJoCo's public documentation.
"""
import os
import xml.etree.ElementTree as ET

import numpy as np

from . import meshproc
from .mathutil import (axis_angle_quat, principal_axes, quat_mul, quat_normalize,
                       quat_to_mat, rot_vec_quat)
from .model import (Model, GEOM_PLANE, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID,
                    GEOM_CYLINDER, GEOM_BOX, GEOM_MESH, JNT_SLIDE, JNT_HINGE,
                    INT_EULER, INT_RK4, SOL_PGS, SOL_CG, SOL_NEWTON,
                    DSBL_CONTACT, DSBL_CONSTRAINT, DSBL_EQUALITY, DSBL_FRICTIONLOSS,
                    DSBL_LIMIT, DSBL_GRAVITY, DSBL_CLAMPCTRL, DSBL_WARMSTART,
                    DSBL_PASSIVE, DSBL_ACTUATION, DSBL_REFSAFE, DSBL_FILTERPARENT,
                    MINVAL)

_GEOM_TYPES = {"plane": GEOM_PLANE, "sphere": GEOM_SPHERE, "capsule": GEOM_CAPSULE,
               "ellipsoid": GEOM_ELLIPSOID, "cylinder": GEOM_CYLINDER, "box": GEOM_BOX,
               "mesh": GEOM_MESH}
_FLAGS = {"contact": DSBL_CONTACT, "constraint": DSBL_CONSTRAINT, "equality": DSBL_EQUALITY,
          "frictionloss": DSBL_FRICTIONLOSS, "limit": DSBL_LIMIT, "gravity": DSBL_GRAVITY,
          "clampctrl": DSBL_CLAMPCTRL, "warmstart": DSBL_WARMSTART, "passive": DSBL_PASSIVE,
          "actuation": DSBL_ACTUATION, "refsafe": DSBL_REFSAFE, "filterparent": DSBL_FILTERPARENT}

# MuJoCo built-in defaults [EXT] for the attributes this dialect can carry
_BUILTIN = {
    "geom": dict(type="sphere", size="0 0 0", pos="0 0 0", quat="1 0 0 0", contype="1",
                 conaffinity="1", condim="3", friction="1 0.005 0.0001", margin="0", gap="0",
                 solref="0.02 1", solimp="0.9 0.95 0.001 0.5 2", density="1000", solmix="1",
                 priority="0"),
    "joint": dict(type="hinge", pos="0 0 0", axis="0 0 1", limited="false", range="0 0",
                  damping="0", armature="0", margin="0", frictionloss="0", stiffness="0",
                  ref="0", springref="0", solreflimit="0.02 1",
                  solimplimit="0.9 0.95 0.001 0.5 2", solreffriction="0.02 1",
                  solimpfriction="0.9 0.95 0.001 0.5 2"),
    "mesh": dict(scale="1 1 1"),
    "general": dict(ctrllimited="false", forcelimited="false", ctrlrange="0 0",
                    forcerange="0 0", gear="1", kp="1", kv="1"),
    "equality": dict(active="true", solref="0.02 1", solimp="0.9 0.95 0.001 0.5 2"),
}
_ACT_TAGS = ("position", "motor", "velocity")


def _floats(s, n=None):
    v = np.array([float(x) for x in str(s).split()], dtype=np.float64)
    if n is not None and len(v) != n:
        if len(v) < n:
            v = np.concatenate([v, np.zeros(n - len(v))])
        else:
            raise ValueError(f"expected {n} numbers, got {s!r}")
    return v


def _bool(s):
    s = str(s).strip().lower()
    if s in ("true", "1"):
        return True
    if s in ("false", "0"):
        return False
    raise ValueError(f"bad boolean {s!r}")


class _Defaults:
    """Default-class tree: class name -> {tag -> attrib dict}, with inheritance."""

    def __init__(self):
        self.classes = {"main": {}}
        self.parent = {"main": None}

    def load(self, elem, cls="main"):
        for child in elem:
            if child.tag == "default":
                name = child.get("class")
                if not name:
                    raise ValueError("nested <default> needs a class attribute")
                self.classes[name] = {}
                self.parent[name] = child.get("parent", cls)
                self.load(child, name)
            else:
                self.classes[cls].setdefault(child.tag, {}).update(child.attrib)

    def resolve(self, tag, cls):
        chain = []
        c = cls or "main"
        if c not in self.classes:
            raise ValueError(f"unknown default class {c!r}")
        while c is not None:
            chain.append(c)
            c = self.parent[c]
        out = dict(_BUILTIN["general" if tag in _ACT_TAGS else tag]) if (tag in _BUILTIN or tag in _ACT_TAGS) else {}
        for c in reversed(chain):
            out.update(self.classes[c].get(tag, {}))
        return out


def _primitive_inertia(gtype, size, density):
    """(mass, principal inertia[3]) of a primitive geom about its centre [EXT]."""
    if gtype == GEOM_SPHERE:
        r = size[0]
        m = density * 4.0 / 3.0 * np.pi * r ** 3
        return m, np.full(3, 0.4 * m * r * r)
    if gtype == GEOM_BOX:
        m = density * 8.0 * size[0] * size[1] * size[2]
        return m, m / 3.0 * np.array([size[1] ** 2 + size[2] ** 2, size[0] ** 2 + size[2] ** 2,
                                      size[0] ** 2 + size[1] ** 2])
    if gtype == GEOM_CYLINDER:
        r, h = size[0], size[1]
        m = density * np.pi * r * r * 2 * h
        ixy = m * (3 * r * r + (2 * h) ** 2) / 12.0
        return m, np.array([ixy, ixy, 0.5 * m * r * r])
    if gtype == GEOM_CAPSULE:
        r, h = size[0], size[1]
        mc = density * np.pi * r * r * 2 * h
        ms = density * 4.0 / 3.0 * np.pi * r ** 3
        m = mc + ms
        izz = 0.5 * mc * r * r + 0.4 * ms * r * r
        ixx = (mc * (3 * r * r + (2 * h) ** 2) / 12.0
               + ms * (0.4 * r * r + h * h + 0.75 * r * h))
        return m, np.array([ixx, ixx, izz])
    if gtype == GEOM_ELLIPSOID:
        m = density * 4.0 / 3.0 * np.pi * size[0] * size[1] * size[2]
        return m, 0.2 * m * np.array([size[1] ** 2 + size[2] ** 2, size[0] ** 2 + size[2] ** 2,
                                      size[0] ** 2 + size[1] ** 2])
    return 0.0, np.zeros(3)


def _primitive_rbound(gtype, size):
    if gtype == GEOM_SPHERE:
        return size[0]
    if gtype == GEOM_CAPSULE:
        return size[0] + size[1]
    if gtype == GEOM_CYLINDER:
        return float(np.hypot(size[0], size[1]))
    if gtype in (GEOM_BOX,):
        return float(np.linalg.norm(size))
    if gtype == GEOM_ELLIPSOID:
        return float(np.max(size))
    return 0.0  # plane: unbounded, flagged by rbound == 0


def load_model_from_path(path, **kw):
    """mujoco-py `load_model_from_path` counterpart (BASELINE.json north_star;
    reference call site src/Mujoco_XML.py:341 uses the `_from_xml` sibling)."""
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    if path.endswith(".npz"):
        return Model.load(path)
    with open(path, "r") as f:
        txt = f.read()
    return load_model_from_xml(txt, base_dir=os.path.dirname(os.path.abspath(path)), **kw)


def load_model_from_xml(xml_string, base_dir=None, inertia_mode="legacy", add_ground_plane=None):
    """Compile an MJCF string.  Mesh `file=` paths are resolved relative to
    `base_dir` (the XML's directory; reference output/DexterousDynamos.xml:10-25
    uses paths relative to the XML, no meshdir) unless absolute
    (src/Mujoco_XML.py:339 rewrites them to /tmp/assets/...).

    add_ground_plane: optional z height; appends the build's own plane geom to
    the world body (BASELINE.json config 3 — NOT part of the reference XML).
    """
    root = ET.fromstring(xml_string)
    if root.tag != "mujoco":
        raise ValueError("root element must be <mujoco>")
    base_dir = base_dir or os.getcwd()

    # ---- compiler / option ------------------------------------------------
    angle_scale = np.pi / 180.0  # MJCF default angle unit is degree [EXT]
    meshdir = ""
    for c in root.findall("compiler"):
        if c.get("angle", "degree") == "radian":
            angle_scale = 1.0
        meshdir = c.get("meshdir", meshdir)
        if c.get("coordinate", "local") != "local":
            raise ValueError("only coordinate='local' is supported")
    m = Model()
    m.name = root.get("model", "model")
    for o in root.findall("option"):
        if "timestep" in o.attrib:
            m.opt_timestep = float(o.get("timestep"))
        if "gravity" in o.attrib:
            m.opt_gravity = _floats(o.get("gravity"), 3)
        if "integrator" in o.attrib:
            m.opt_integrator = {"Euler": INT_EULER, "RK4": INT_RK4}[o.get("integrator")]
        if "solver" in o.attrib:
            m.opt_solver = {"PGS": SOL_PGS, "CG": SOL_CG, "Newton": SOL_NEWTON}[o.get("solver")]
        if "iterations" in o.attrib:
            m.opt_iterations = int(o.get("iterations"))
        if "tolerance" in o.attrib:
            m.opt_tolerance = float(o.get("tolerance"))
        if "ls_iterations" in o.attrib:
            m.opt_ls_iterations = int(o.get("ls_iterations"))
        if "ls_tolerance" in o.attrib:
            m.opt_ls_tolerance = float(o.get("ls_tolerance"))
        if "impratio" in o.attrib:
            m.opt_impratio = float(o.get("impratio"))
        if "mpr_iterations" in o.attrib:
            m.opt_mpr_iterations = int(o.get("mpr_iterations"))
        if "mpr_tolerance" in o.attrib:
            m.opt_mpr_tolerance = float(o.get("mpr_tolerance"))
        if o.get("cone", "pyramidal") != "pyramidal":
            raise ValueError("only cone='pyramidal' is supported")
        for fl in o.findall("flag"):
            for k, v in fl.attrib.items():
                if k in _FLAGS and v == "disable":
                    m.opt_disableflags |= _FLAGS[k]

    defaults = _Defaults()
    for d in root.findall("default"):
        defaults.load(d)

    # ---- assets -------------------------------------------------------------
    mesh_specs = {}
    for a in root.findall("asset"):
        for me in a.findall("mesh"):
            att = defaults.resolve("mesh", me.get("class"))
            att.update(me.attrib)
            name = att.get("name") or os.path.splitext(os.path.basename(att["file"]))[0]
            fpath = att["file"]
            if not os.path.isabs(fpath):
                fpath = os.path.join(base_dir, meshdir, fpath)
            mesh_specs[name] = (fpath, _floats(att["scale"], 3))

    # ---- body tree ------------------------------------------------------------
    bodies = [dict(name="world", parent=0, pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]),
                   joints=[], geoms=[], childclass=None)]

    def walk(elem, parent_id, childclass):
        for child in elem:
            if child.tag == "body":
                cc = child.get("childclass", childclass)
                b = dict(name=child.get("name", f"body{len(bodies)}"), parent=parent_id,
                         pos=_floats(child.get("pos", "0 0 0"), 3),
                         quat=quat_normalize(_floats(child.get("quat", "1 0 0 0"), 4)),
                         joints=[], geoms=[], childclass=cc)
                bodies.append(b)
                bid = len(bodies) - 1
                # joints / geoms of this body come in document order regardless of
                # where child bodies sit (reference XML puts them after children)
                for sub in child:
                    if sub.tag == "joint":
                        att = defaults.resolve("joint", sub.get("class", cc))
                        att.update(sub.attrib)
                        b["joints"].append(att)
                    elif sub.tag == "geom":
                        att = defaults.resolve("geom", sub.get("class", cc))
                        att.update(sub.attrib)
                        b["geoms"].append(att)
                    elif sub.tag in ("inertial", "freejoint", "site", "camera", "light"):
                        if sub.tag in ("inertial", "freejoint"):
                            raise ValueError(f"<{sub.tag}> is outside the supported MJCF subset")
                walk(child, bid, cc)

    wb = root.find("worldbody")
    if wb is None:
        raise ValueError("missing <worldbody>")
    for sub in wb:
        if sub.tag == "geom":
            att = defaults.resolve("geom", sub.get("class"))
            att.update(sub.attrib)
            bodies[0]["geoms"].append(att)
    walk(wb, 0, None)
    if add_ground_plane is not None:
        att = defaults.resolve("geom", None)
        att.update(dict(type="plane", size="0 0 1", pos=f"0 0 {float(add_ground_plane)!r}",
                        name="ground_plane"))
        bodies[0]["geoms"].append(att)

    nbody = len(bodies)
    used_meshes = []
    for b in bodies:
        for g in b["geoms"]:
            if g.get("type") == "mesh" or ("mesh" in g and g.get("type", "mesh") == "mesh"):
                g["type"] = "mesh"
                if g["mesh"] not in mesh_specs:
                    raise ValueError(f"geom references unknown mesh {g['mesh']!r}")
                if g["mesh"] not in used_meshes:
                    used_meshes.append(g["mesh"])

    # compile only meshes that a geom uses (four *-Assembly meshes in the
    # reference XML are declared but unused: output/DexterousDynamos.xml:11-13,17)
    meshes = {}
    for name in used_meshes:
        fpath, scale = mesh_specs[name]
        meshes[name] = meshproc.load_mesh(name, fpath, scale, inertia_mode)

    m.mesh_names = list(used_meshes)
    m.mesh_vertadr = np.zeros(len(used_meshes), dtype=np.int32)
    m.mesh_vertnum = np.zeros(len(used_meshes), dtype=np.int32)
    m.mesh_adjadr = np.zeros(len(used_meshes), dtype=np.int32)
    m.mesh_faceadr = np.zeros(len(used_meshes), dtype=np.int32)
    m.mesh_facenum = np.zeros(len(used_meshes), dtype=np.int32)
    hv, hs, ha, hf = [], [], [], []
    vadr = aadr = fadr = 0
    for i, name in enumerate(used_meshes):
        cm = meshes[name]
        m.mesh_vertadr[i] = vadr
        m.mesh_vertnum[i] = len(cm.hull_vert)
        m.mesh_adjadr[i] = aadr
        m.mesh_faceadr[i] = fadr
        m.mesh_facenum[i] = len(cm.hull_face)
        hv.append(cm.hull_vert)
        hs.append(cm.hull_adj_start[:-1] + aadr)  # absolute start per vertex
        ha.append(cm.hull_adj)                     # hull-local neighbour ids
        hf.append(cm.hull_face)
        vadr += len(cm.hull_vert)
        aadr += len(cm.hull_adj)
        fadr += len(cm.hull_face)
    m.hull_vert = np.concatenate(hv) if hv else np.zeros((0, 3))
    m.hull_adj_start = (np.concatenate(hs + [np.array([aadr])]).astype(np.int32)
                        if hs else np.zeros(1, dtype=np.int32))
    m.hull_adj = np.concatenate(ha).astype(np.int32) if ha else np.zeros(0, dtype=np.int32)
    m.hull_face = np.concatenate(hf).astype(np.int32) if hf else np.zeros((0, 3), dtype=np.int32)

    # ---- flatten ----------------------------------------------------------------
    m.body_names = [b["name"] for b in bodies]
    if len(set(m.body_names)) != nbody:
        raise ValueError("repeated body name")
    m.body_parentid = np.array([b["parent"] for b in bodies], dtype=np.int32)
    m.body_pos = np.array([b["pos"] for b in bodies])
    m.body_quat = np.array([b["quat"] for b in bodies])
    m.body_rootid = np.zeros(nbody, dtype=np.int32)
    m.body_weldid = np.zeros(nbody, dtype=np.int32)
    m.body_jntadr = -np.ones(nbody, dtype=np.int32)
    m.body_jntnum = np.zeros(nbody, dtype=np.int32)
    m.body_dofadr = -np.ones(nbody, dtype=np.int32)
    m.body_dofnum = np.zeros(nbody, dtype=np.int32)
    m.body_geomadr = -np.ones(nbody, dtype=np.int32)
    m.body_geomnum = np.zeros(nbody, dtype=np.int32)

    jn, jt, jb, jpos, jax, jlim, jrng, jmar, jstf, jref, jsref = [], [], [], [], [], [], [], [], [], [], []
    jsolref, jsolimp = [], []
    darm, ddmp, dfl, dsolref, dsolimp = [], [], [], [], []
    for bid, b in enumerate(bodies):
        p = b["parent"]
        if bid > 0:
            m.body_rootid[bid] = bid if p == 0 else m.body_rootid[p]
            m.body_weldid[bid] = bid if b["joints"] else m.body_weldid[p]
        if b["joints"]:
            m.body_jntadr[bid] = len(jn)
            m.body_jntnum[bid] = len(b["joints"])
            m.body_dofadr[bid] = len(jn)
            m.body_dofnum[bid] = len(b["joints"])
        for j in b["joints"]:
            t = j["type"]
            if t not in ("hinge", "slide"):
                raise ValueError(f"joint type {t!r} is outside the supported subset (hinge, slide)")
            jn.append(j.get("name", f"joint{len(jn)}"))
            jt.append(JNT_HINGE if t == "hinge" else JNT_SLIDE)
            jb.append(bid)
            jpos.append(_floats(j["pos"], 3))
            ax = _floats(j["axis"], 3)
            n = np.linalg.norm(ax)
            if n < MINVAL:
                raise ValueError("zero joint axis")
            jax.append(ax / n)
            lim = _bool(j["limited"])
            rng = _floats(j["range"], 2)
            ref = float(j["ref"])
            sref = float(j["springref"])
            if t == "hinge":
                rng = rng * angle_scale
                ref *= angle_scale
                sref *= angle_scale
            if lim and rng[0] > rng[1]:
                raise ValueError(f"joint {jn[-1]}: range[0] > range[1]")
            jlim.append(1 if lim else 0)
            jrng.append(rng)
            jmar.append(float(j["margin"]))
            jstf.append(float(j["stiffness"]))
            jref.append(ref)
            jsref.append(sref)
            jsolref.append(_floats(j["solreflimit"], 2))
            jsolimp.append(_floats(j["solimplimit"], 5))
            darm.append(float(j["armature"]))
            ddmp.append(float(j["damping"]))
            dfl.append(float(j["frictionloss"]))
            dsolref.append(_floats(j["solreffriction"], 2))
            dsolimp.append(_floats(j["solimpfriction"], 5))

    njnt = len(jn)
    m.joint_names = jn
    if len(set(jn)) != njnt:
        raise ValueError("repeated joint name")
    m.jnt_type = np.array(jt, dtype=np.int32).reshape(njnt)
    m.jnt_bodyid = np.array(jb, dtype=np.int32).reshape(njnt)
    m.jnt_qposadr = np.arange(njnt, dtype=np.int32)
    m.jnt_dofadr = np.arange(njnt, dtype=np.int32)
    m.jnt_pos = np.array(jpos).reshape(njnt, 3)
    m.jnt_axis = np.array(jax).reshape(njnt, 3)
    m.jnt_limited = np.array(jlim, dtype=np.int32).reshape(njnt)
    m.jnt_range = np.array(jrng).reshape(njnt, 2)
    m.jnt_margin = np.array(jmar).reshape(njnt)
    m.jnt_stiffness = np.array(jstf).reshape(njnt)
    m.jnt_solref = np.array(jsolref).reshape(njnt, 2)
    m.jnt_solimp = np.array(jsolimp).reshape(njnt, 5)
    m.qpos0 = np.array(jref).reshape(njnt)
    m.qpos_spring = np.array(jsref).reshape(njnt)
    m.dof_bodyid = m.jnt_bodyid.copy()
    m.dof_jntid = np.arange(njnt, dtype=np.int32)
    m.dof_armature = np.array(darm).reshape(njnt)
    m.dof_damping = np.array(ddmp).reshape(njnt)
    m.dof_frictionloss = np.array(dfl).reshape(njnt)
    m.dof_solref = np.array(dsolref).reshape(njnt, 2)
    m.dof_solimp = np.array(dsolimp).reshape(njnt, 5)
    # dof tree: parent = previous dof of the same body, else last dof of the
    # nearest ancestor that has dofs
    m.dof_parentid = -np.ones(njnt, dtype=np.int32)
    for d in range(njnt):
        b = m.dof_bodyid[d]
        if d > m.body_dofadr[b]:
            m.dof_parentid[d] = d - 1
        else:
            a = m.body_parentid[b]
            while a > 0 and m.body_dofnum[a] == 0:
                a = m.body_parentid[a]
            if a > 0:
                m.dof_parentid[d] = m.body_dofadr[a] + m.body_dofnum[a] - 1
    m.dof_Madr = np.zeros(njnt, dtype=np.int32)
    nM = 0
    for d in range(njnt):
        m.dof_Madr[d] = nM
        k = d
        while k >= 0:
            nM += 1
            k = m.dof_parentid[k]
    m.nM = nM

    # ---- geoms + body inertial frames -----------------------------------------
    gname, gtype, gbody, gpos, gquat, gsize, gmesh = [], [], [], [], [], [], []
    gcontype, gconaff, gcondim, gfric, gmargin, ggap, gsolref, gsolimp, gsolmix, gprio, grb = (
        [], [], [], [], [], [], [], [], [], [], [])
    m.body_mass = np.zeros(nbody)
    m.body_ipos = np.zeros((nbody, 3))
    m.body_iquat = np.tile(np.array([1.0, 0, 0, 0]), (nbody, 1))
    m.body_inertia = np.zeros((nbody, 3))
    for bid, b in enumerate(bodies):
        parts = []  # (mass, pos, quat, principal inertia)
        if b["geoms"]:
            m.body_geomadr[bid] = len(gname)
            m.body_geomnum[bid] = len(b["geoms"])
        for g in b["geoms"]:
            t = _GEOM_TYPES.get(g["type"])
            if t is None:
                raise ValueError(f"geom type {g['type']!r} not supported")
            pos = _floats(g["pos"], 3)
            quat = quat_normalize(_floats(g["quat"], 4))
            size = _floats(g["size"], 3)
            density = float(g["density"])
            mesh_id = -1
            if t == GEOM_MESH:
                cm = meshes[g["mesh"]]
                mesh_id = used_meshes.index(g["mesh"])
                # geom frame = mesh principal frame seen from the body
                pos = pos + rot_vec_quat(cm.pos, quat)
                quat = quat_normalize(quat_mul(quat, cm.quat))
                mass = density * cm.volume
                inertia = density * cm.inertia_unit
                rb = cm.rbound
                size = 0.5 * (cm.aabb_max - cm.aabb_min)
            else:
                mass, inertia = _primitive_inertia(t, size, density)
                rb = _primitive_rbound(t, size)
            if "mass" in g:
                gm = float(g["mass"])
                if mass > 0:
                    inertia = inertia * (gm / mass)
                mass = gm
            gname.append(g.get("name", f"geom{len(gname)}"))
            gtype.append(t)
            gbody.append(bid)
            gpos.append(pos)
            gquat.append(quat)
            gsize.append(size)
            gmesh.append(mesh_id)
            gcontype.append(int(g["contype"]))
            gconaff.append(int(g["conaffinity"]))
            gcondim.append(int(g["condim"]))
            gfric.append(_floats(g["friction"], 3))
            gmargin.append(float(g["margin"]))
            ggap.append(float(g["gap"]))
            gsolref.append(_floats(g["solref"], 2))
            gsolimp.append(_floats(g["solimp"], 5))
            gsolmix.append(float(g["solmix"]))
            gprio.append(int(g["priority"]))
            grb.append(rb)
            if t != GEOM_PLANE and mass > 0:
                parts.append((mass, pos, quat, inertia))
        if bid == 0:
            continue
        if len(parts) == 1:
            m.body_mass[bid], m.body_ipos[bid], m.body_iquat[bid], m.body_inertia[bid] = parts[0]
        elif len(parts) > 1:
            mass = sum(p[0] for p in parts)
            com = sum(p[0] * p[1] for p in parts) / mass
            I = np.zeros((3, 3))
            for pm, pp, pq, pi in parts:
                R = quat_to_mat(pq)
                d = pp - com
                I += R @ np.diag(pi) @ R.T + pm * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
            mom, q = principal_axes(I)
            m.body_mass[bid], m.body_ipos[bid], m.body_iquat[bid], m.body_inertia[bid] = mass, com, q, mom
        if b["joints"]:
            if m.body_mass[bid] < MINVAL or m.body_inertia[bid].min() < MINVAL:
                # MuJoCo's compile error for massless moving bodies [EXT]
                raise ValueError(f"body {b['name']!r}: mass and inertia of moving bodies must be "
                                 f"larger than mjMINVAL")
    ngeom = len(gname)
    m.geom_names = gname
    m.geom_type = np.array(gtype, dtype=np.int32).reshape(ngeom)
    m.geom_bodyid = np.array(gbody, dtype=np.int32).reshape(ngeom)
    m.geom_pos = np.array(gpos).reshape(ngeom, 3)
    m.geom_quat = np.array(gquat).reshape(ngeom, 4)
    m.geom_size = np.array(gsize).reshape(ngeom, 3)
    m.geom_dataid = np.array(gmesh, dtype=np.int32).reshape(ngeom)
    m.geom_contype = np.array(gcontype, dtype=np.int32).reshape(ngeom)
    m.geom_conaffinity = np.array(gconaff, dtype=np.int32).reshape(ngeom)
    m.geom_condim = np.array(gcondim, dtype=np.int32).reshape(ngeom)
    m.geom_friction = np.array(gfric).reshape(ngeom, 3)
    m.geom_margin = np.array(gmargin).reshape(ngeom)
    m.geom_gap = np.array(ggap).reshape(ngeom)
    m.geom_solref = np.array(gsolref).reshape(ngeom, 2)
    m.geom_solimp = np.array(gsolimp).reshape(ngeom, 5)
    m.geom_solmix = np.array(gsolmix).reshape(ngeom)
    m.geom_priority = np.array(gprio, dtype=np.int32).reshape(ngeom)
    m.geom_rbound = np.array(grb).reshape(ngeom)
    m.body_subtreemass = m.body_mass.copy()
    for bid in range(nbody - 1, 0, -1):
        m.body_subtreemass[m.body_parentid[bid]] += m.body_subtreemass[bid]

    # ---- actuators ------------------------------------------------------------------
    aname, atrn, aclim, aflim, acr, afr, agear, again, abias = [], [], [], [], [], [], [], [], []
    for a in root.findall("actuator"):
        for e in a:
            if e.tag not in _ACT_TAGS:
                raise ValueError(f"actuator <{e.tag}> not supported")
            att = defaults.resolve(e.tag, e.get("class"))
            att.update(e.attrib)
            if "joint" not in att:
                raise ValueError("only joint transmissions are supported")
            if att["joint"] not in jn:
                raise ValueError(f"actuator references unknown joint {att['joint']!r}")
            jid = jn.index(att["joint"])
            aname.append(att.get("name", f"actuator{len(aname)}"))
            atrn.append(jid)
            aclim.append(1 if _bool(att["ctrllimited"]) else 0)
            aflim.append(1 if _bool(att["forcelimited"]) else 0)
            acr.append(_floats(att["ctrlrange"], 2))
            afr.append(_floats(att["forcerange"], 2))
            agear.append(_floats(att["gear"])[0])
            if e.tag == "position":
                kp = float(att["kp"])
                again.append([kp, 0, 0])
                abias.append([0, -kp, 0])
            elif e.tag == "velocity":
                kv = float(att["kv"])
                again.append([kv, 0, 0])
                abias.append([0, 0, -kv])
            else:
                again.append([1, 0, 0])
                abias.append([0, 0, 0])
    nu = len(aname)
    m.actuator_names = aname
    m.actuator_trnid = np.array(atrn, dtype=np.int32).reshape(nu)
    m.actuator_ctrllimited = np.array(aclim, dtype=np.int32).reshape(nu)
    m.actuator_forcelimited = np.array(aflim, dtype=np.int32).reshape(nu)
    m.actuator_ctrlrange = np.array(acr, dtype=np.float64).reshape(nu, 2)
    m.actuator_forcerange = np.array(afr, dtype=np.float64).reshape(nu, 2)
    m.actuator_gear = np.array(agear, dtype=np.float64).reshape(nu)
    m.actuator_gainprm = np.array(again, dtype=np.float64).reshape(nu, 3)
    m.actuator_biasprm = np.array(abias, dtype=np.float64).reshape(nu, 3)

    # ---- contact excludes ----------------------------------------------------------
    ex = []
    for c in root.findall("contact"):
        for e in c.findall("exclude"):
            b1, b2 = e.get("body1"), e.get("body2")
            if b1 not in m.body_names or b2 not in m.body_names:
                raise ValueError(f"exclude references unknown body {b1!r}/{b2!r}")
            ex.append([m.body_names.index(b1), m.body_names.index(b2)])
    m.exclude_pair = np.array(ex, dtype=np.int32).reshape(len(ex), 2)

    # ---- equalities -----------------------------------------------------------------
    e1, e2, edata, eact, esolref, esolimp = [], [], [], [], [], []
    for q in root.findall("equality"):
        for e in q:
            if e.tag != "joint":
                raise ValueError(f"equality <{e.tag}> not supported (joint only)")
            att = defaults.resolve("equality", e.get("class"))
            att.update(e.attrib)
            if att["joint1"] not in jn:
                raise ValueError(f"equality references unknown joint {att['joint1']!r}")
            e1.append(jn.index(att["joint1"]))
            j2 = att.get("joint2")
            if j2 is not None and j2 not in jn:
                raise ValueError(f"equality references unknown joint {j2!r}")
            e2.append(jn.index(j2) if j2 is not None else -1)
            edata.append(_floats(att.get("polycoef", "0 1 0 0 0"), 5))
            eact.append(1 if _bool(att["active"]) else 0)
            esolref.append(_floats(att["solref"], 2))
            esolimp.append(_floats(att["solimp"], 5))
    neq = len(e1)
    m.eq_obj1id = np.array(e1, dtype=np.int32).reshape(neq)
    m.eq_obj2id = np.array(e2, dtype=np.int32).reshape(neq)
    m.eq_data = np.array(edata, dtype=np.float64).reshape(neq, 5)
    m.eq_active = np.array(eact, dtype=np.int32).reshape(neq)
    m.eq_solref = np.array(esolref, dtype=np.float64).reshape(neq, 2)
    m.eq_solimp = np.array(esolimp, dtype=np.float64).reshape(neq, 5)

    m.finalize()
    return m
