"""Exporter restatement: `fusion_info.json` + STL folder -> MJCF of the dialect
`mujoco_amd.mjcf` parses, i.e. what the reference's `Fusion_to_Mujoco(...).export_xml()` writes.

Follows, without pyquaternion / termcolor / mujoco-py:
* the component and joint trees and the transform algebra of `src/Fusion_Model.py:60-142, 144-244`
  (component pose = product of the `transformation` entries down from "Root"; a joint component's
  body frame is its component's absolute pose relative to its *joint* parent; joint axis / origin
  are rotated into the body frame);
* `Fusion_to_Mujoco._recursive_add_component` (`src/Fusion_to_Mujoco.py:91-113`): one body with one
  mesh geom per joint component, a hinge per rotating component, a position actuator per joint
  except the "M-DP" ones, which get a joint equality `polycoef="0 70/120 0 0 0"` to the parent's
  joint, and a contact exclude for every parent/child pair ("root" included);
* the defaults and the document order of `src/Mujoco_XML.py:37-75, 187-274`: every insertion lands
  directly after the parent's opening tag, so inside a body the children come first (last added
  first), then the joint, then the geom — MuJoCo numbers bodies in document order, so ids follow.

Parity is **structural only**: `output/DexterousDynamos.xml` was written from a JSON that is not in
the reference; `tests/test_exporter.py` checks topology, names and numbers against it through the
closest input the reference holds.  The asset folders of the reference are incomplete
(`.MISSING_LARGE_BLOBS`); `mesh_dirs` lists further folders to look a mesh up in and
`substitutions` maps a mesh that exists nowhere to a stand-in — both are declared in the XML as
comments so a report can state them.
"""
import json
import os

import numpy as np

from .mathutil import quat_conj, quat_mul, quat_normalize, rot_vec_quat


class Component:
    def __init__(self, cid, name, stlname):
        self.id, self.name, self.stlname = cid, name, stlname
        self.quat, self.trans = np.array([1.0, 0, 0, 0]), np.zeros(3)          # transform to the parent component
        self.abs_quat, self.abs_trans = np.array([1.0, 0, 0, 0]), np.zeros(3)  # pose in the Fusion world
        self.rel_quat, self.rel_trans = np.array([1.0, 0, 0, 0]), np.zeros(3)  # pose in the joint parent's frame
        self.parent, self.children, self.joint = None, [], None


class Joint:
    def __init__(self, name, axis, origin, jrange):
        self.name, self.axis, self.origin, self.range = name, np.asarray(axis, float), np.asarray(origin, float), list(jrange)
        self.rel_axis, self.rel_pos = np.zeros(3), np.zeros(3)


class FusionModel:
    """Component tree + joint tree of one `fusion_info.json` (`ExtractFusionInfo.py` schema)."""

    def __init__(self, json_path):
        with open(json_path) as f:
            data = json.load(f)
        self.components, self.joint_components = [], []
        self._build_components(data["components"])
        self._build_joints(data["joints"])
        self._transforms()

    def _build_components(self, items):
        by_id = {"Root": Component("Root", "Root", None)}
        self.components.append(by_id["Root"])
        for it in items:
            cid = it["component"]["id"]
            c = Component(cid, f'{it["component"]["name"]}_{cid}', it["stl_file"])
            by_id[cid] = c
            self.components.append(c)
        for it in items:
            c, p = by_id[it["component"]["id"]], by_id[it["parent"]["id"]]
            c.parent = p
            p.children.append(c)
            c.quat = quat_normalize(np.asarray(it["transformation"]["quaternion"], float))
            c.trans = np.asarray(it["transformation"]["translation"], float)
        self._by_id = by_id

    def _build_joints(self, items):
        jc, parent_id = {}, {}
        for it in items:  # one joint component per rotating component, in file order
            rid = it["component_rotating"]["id"]
            if rid in jc:
                continue
            name = f'{it["component_rotating"]["name"]}_{rid}'
            c = Component(rid, name, self._by_id[rid].stlname.split(".stl")[0])
            t = it["transformation"]
            c.joint = Joint(name + "_joint", t["joint_axis"], t["joint_origin"], t["joint_range"])
            jc[rid] = c
            parent_id[rid] = it["component_base"]["id"]
            self.joint_components.append(c)
        root = None
        for it in items:  # the one base that never rotates is the root of the joint tree
            bid = it["component_base"]["id"]
            if bid in jc:
                continue
            if root is not None:
                raise ValueError("Multiple root components found in joint data. This sort of model is not supported - "
                                 "all joints must lead to a single root component.")
            root = Component(bid, f'{it["component_base"]["name"]}_{bid}', self._by_id[bid].stlname.split(".stl")[0])
            jc[bid] = root
        if root is None:
            raise ValueError("joint data has no root component")
        self.joint_components.insert(0, root)
        for c in self.joint_components:
            if c is root:
                continue
            c.parent = jc[parent_id[c.id]]
            c.parent.children.append(c)
        self._jc = jc

    def _transforms(self):
        for c in self.components:  # absolute pose: compose the chain of transforms up to "Root"
            q, t, p = c.quat.copy(), c.trans.copy(), c.parent
            chain = []
            while p is not None:
                chain.append(p)
                p = p.parent
            for a in chain:
                t = rot_vec_quat(t, a.quat) + a.trans
            for a in chain:
                q = quat_mul(a.quat, q)
            c.abs_quat, c.abs_trans = q, t
            if c.id in self._jc:
                self._jc[c.id].abs_quat, self._jc[c.id].abs_trans = q, t
        for c in self.joint_components:
            if c.parent is not None:
                pq = c.parent.abs_quat
                if not np.isclose(np.linalg.norm(pq), 1):
                    pq = quat_normalize(pq)
                c.rel_trans = rot_vec_quat(c.abs_trans - c.parent.abs_trans, quat_conj(pq))
                c.rel_quat = quat_mul(quat_conj(pq), c.abs_quat)
            if c.joint is not None:
                inv = quat_conj(c.abs_quat)
                c.joint.rel_axis = rot_vec_quat(c.joint.axis, inv)
                c.joint.rel_pos = rot_vec_quat(c.joint.origin - c.abs_trans, inv)


def _fmt(v):
    return " ".join(repr(float(x)) for x in v)


def find_latest_folder(asset_folder):
    """Newest `fusion_export_<date>_<time>` folder by its name (`src/utils.py:88-122`)."""
    cands = sorted(d for d in os.listdir(asset_folder)
                   if d.startswith("fusion_export_") and os.path.isdir(os.path.join(asset_folder, d)))
    if not cands:
        raise FileNotFoundError(f"no fusion_export_* folder in {asset_folder}")
    return os.path.join(asset_folder, cands[-1])


def export_xml(export_folder, model_name="DexterousDynamos", mesh_dirs=(), substitutions=None,
               json_filename="fusion_info.json"):
    """MJCF text for one Fusion export folder.  Meshes are looked up in the folder itself, then in
    `mesh_dirs`; `substitutions` = {mesh name: file path} for meshes that exist nowhere."""
    fm = FusionModel(os.path.join(export_folder, json_filename))
    substitutions = dict(substitutions or {})
    meshes, notes = {}, []
    for root, _, files in os.walk(export_folder):  # every STL of the folder is an asset, used or not
        for fn in sorted(files):
            if fn.lower().endswith(".stl"):
                meshes[os.path.splitext(fn)[0]] = os.path.abspath(os.path.join(root, fn))
    for c in fm.joint_components:
        if c.stlname in meshes:
            continue
        if c.stlname in substitutions:
            meshes[c.stlname] = os.path.abspath(substitutions[c.stlname])
            notes.append(f"mesh {c.stlname}: SUBSTITUTED by {substitutions[c.stlname]}")
            continue
        for d in mesh_dirs:
            p = os.path.join(d, c.stlname + ".stl")
            if os.path.exists(p):
                meshes[c.stlname] = os.path.abspath(p)
                notes.append(f"mesh {c.stlname}: missing from the export folder, taken from {d}")
                break
        else:
            raise FileNotFoundError(f"mesh '{c.stlname}.stl' of body '{c.name}' not found in {export_folder} or {list(mesh_dirs)}")
    out = [f'<mujoco model="{model_name}">', '\t<compiler angle="radian"/>']
    out += [f"\t<!-- {n} -->" for n in notes]
    out += ['\t<default>',
            '\t\t<geom rgba="1 1 1 1" type="mesh" friction="1 0.005 0.001" condim="3" margin="0.0005" contype="1" conaffinity="1"/>',
            '\t\t<joint type="hinge" limited="true" damping="0.1" armature="0.001" margin="0.01" frictionloss="0.001"/>',
            '\t\t<position ctrllimited="true" forcelimited="true" forcerange="-1 1" kp="2.0"/>',
            '\t\t<mesh scale="0.001 0.001 0.001"/>', '\t</default>', '\t<asset>']
    out += [f'\t\t<mesh name="{n}" file="{p}"/>' for n, p in meshes.items()]
    out += ['\t</asset>', '\t<worldbody>', '\t\t<body name="root" quat="1.0 0.0 0.0 0.0">']
    actuators, excludes, equalities = [], [], []

    def visit(c):  # order in which the reference ADDS things (actuators, excludes, equalities keep it)
        excludes.append((c.parent.name if c.parent is not None else "root", c.name))
        if c.joint is not None:
            if "M-DP" in c.name:
                equalities.append((c.joint.name, c.parent.joint.name, 70 / 120))
            else:
                actuators.append((c.joint.name + "_actuator", c.joint.name, c.joint.range))
        for ch in c.children:
            visit(ch)

    def emit(c, depth):  # document order: children (last added first), joint, geom
        ind = "\t" * depth
        out.append(f'{ind}<body name="{c.name}" pos="{_fmt(c.rel_trans)}" quat="{_fmt(c.rel_quat)}">')
        for ch in reversed(c.children):
            emit(ch, depth + 1)
        if c.joint is not None:
            out.append(f'{ind}\t<joint name="{c.joint.name}" pos="{_fmt(c.joint.rel_pos)}" axis="{_fmt(c.joint.rel_axis)}" '
                       f'range="{_fmt(c.joint.range)}"/>')
        out.append(f'{ind}\t<geom mesh="{c.stlname}"/>')
        out.append(f"{ind}</body>")

    visit(fm.joint_components[0])
    emit(fm.joint_components[0], 3)
    out += ['\t\t</body>', '\t</worldbody>', '\t<actuator>']
    out += [f'\t\t<position name="{n}" joint="{j}" ctrlrange="{_fmt(r)}"/>' for n, j, r in actuators]
    out += ['\t</actuator>', '\t<contact>']
    out += [f'\t\t<exclude body1="{a}" body2="{b}"/>' for a, b in excludes]
    out += ['\t</contact>', '\t<equality>']
    out += [f'\t\t<joint joint1="{a}" joint2="{b}" polycoef="0 {f} 0 0 0"/>' for a, b, f in equalities]
    out += ['\t</equality>', '</mujoco>']
    return "\n".join(out)


def export_model(export_folder, **kw):
    """`export_xml` followed by the model compiler (`load_model_from_xml`)."""
    from .mjcf import load_model_from_xml
    return load_model_from_xml(export_xml(export_folder, **kw), base_dir=export_folder)
