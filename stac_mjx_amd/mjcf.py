"""MJCF-subset model compiler: XML -> flat kinematic tables (no mujoco needed).

The reference obtains its model tables from ``mujoco.MjSpec.from_file(...).compile()``
(``stac_mjx/stac.py:185-235``) after inserting one site per keypoint
(``stac.py:192-205``) and rescaling with ``rescale.dm_scale_spec``
(``stac_mjx/rescale.py:6-46``).  MuJoCo is not available to this engine, so this
module restates the part of the MJCF compile that forward kinematics depends on:

* bodies in depth-first document order (world = 0), ``pos`` and orientation
  (``quat`` / ``euler`` / ``axisangle`` / ``xyaxes`` / ``zaxis``),
* nested ``<default class=...>`` resolution incl. ``childclass``,
* joints (``free`` / ``ball`` / ``slide`` / ``hinge``) grouped per body in
  document order, ``axis`` normalised, ``pos``, ``range``, ``ref`` -> ``qpos0``,
* ``compiler angle=degree|radian`` and ``eulerseq``,
* keypoint sites (inserted *before* scaling and never scaled, as in the reference),
* the ``dm_scale_spec`` rule: only ``pos`` of bodies *below* the first top-level
  body is scaled; joint ``pos`` and site ``pos`` are not (``rescale.py:21-45``).

All arithmetic is float64 (like the MuJoCo compiler); tables are cast to float32
at the end (like ``mjx.put_model``).
"""

from __future__ import annotations

import math
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from pathlib import Path
from typing import Mapping, Sequence

import numpy as np

# MuJoCo mjtJoint enum values (mujoco/mjmodel.h) -- kept so tables are interchangeable.
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
JNT_QPOS_DIMS = {JNT_FREE: 7, JNT_BALL: 4, JNT_SLIDE: 1, JNT_HINGE: 1}
_JNT_TYPE_BY_NAME = {"free": JNT_FREE, "ball": JNT_BALL, "slide": JNT_SLIDE, "hinge": JNT_HINGE}


class MjcfError(ValueError):
    """Raised for MJCF constructs outside the supported kinematic subset."""


@dataclass
class ModelTables:
    """Flat kinematic model. Arrays are C-contiguous."""

    nbody: int
    njnt: int
    nq: int
    nsite: int  # number of *fit* sites (one per keypoint), K
    body_parentid: np.ndarray  # [nbody] i32
    body_pos: np.ndarray  # [nbody,3] f32
    body_quat: np.ndarray  # [nbody,4] f32 (w,x,y,z), unit
    body_jntadr: np.ndarray  # [nbody] i32 (-1 if none)
    body_jntnum: np.ndarray  # [nbody] i32
    body_depth: np.ndarray  # [nbody] i32 (world = 0)
    jnt_type: np.ndarray  # [njnt] i32 (mjtJoint)
    jnt_qposadr: np.ndarray  # [njnt] i32
    jnt_bodyid: np.ndarray  # [njnt] i32
    jnt_pos: np.ndarray  # [njnt,3] f32
    jnt_axis: np.ndarray  # [njnt,3] f32, unit
    jnt_range: np.ndarray  # [njnt,2] f32 (radians for hinge/ball)
    qpos0: np.ndarray  # [nq] f32
    site_bodyid: np.ndarray  # [K] i32
    site_pos: np.ndarray  # [K,3] f32  (mutable: the marker offsets)
    body_names: list[str] = field(default_factory=list)
    jnt_names: list[str] = field(default_factory=list)
    site_names: list[str] = field(default_factory=list)
    timestep: float = 0.002

    def copy(self) -> "ModelTables":
        kw = {}
        for k, v in self.__dict__.items():
            kw[k] = v.copy() if isinstance(v, np.ndarray) else (list(v) if isinstance(v, list) else v)
        return ModelTables(**kw)

    # -- (de)serialisation used for committed fixtures ------------------------------
    def to_npz_dict(self) -> dict:
        d = {}
        for k, v in self.__dict__.items():
            if isinstance(v, np.ndarray):
                d[k] = v
            elif isinstance(v, list):
                d[k] = np.array(v, dtype=np.str_)
            else:
                d[k] = np.array(v)
        return d

    @staticmethod
    def from_npz_dict(d: Mapping[str, np.ndarray]) -> "ModelTables":
        kw = {}
        for k in ModelTables.__dataclass_fields__:
            v = d[k]
            if k in ("body_names", "jnt_names", "site_names"):
                kw[k] = [str(s) for s in v.tolist()]
            elif k in ("nbody", "njnt", "nq", "nsite"):
                kw[k] = int(v)
            elif k == "timestep":
                kw[k] = float(v)
            else:
                kw[k] = np.ascontiguousarray(v)
        return ModelTables(**kw)

    def save(self, path) -> None:
        np.savez_compressed(path, **self.to_npz_dict())

    @staticmethod
    def load(path) -> "ModelTables":
        with np.load(path, allow_pickle=False) as d:
            return ModelTables.from_npz_dict(d)


# ----------------------------------------------------------------------------------
# small float64 quaternion helpers (w, x, y, z)
# ----------------------------------------------------------------------------------
def _qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array(
        [
            aw * bw - ax * bx - ay * by - az * bz,
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
        ]
    )


def _axis_angle_quat(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    n = np.linalg.norm(axis)
    if n < 1e-14:
        return np.array([1.0, 0.0, 0.0, 0.0])
    axis = axis / n
    s = math.sin(0.5 * angle)
    return np.array([math.cos(0.5 * angle), axis[0] * s, axis[1] * s, axis[2] * s])


def _mat_to_quat(m):
    """Rotation matrix (columns = frame axes) -> unit quaternion."""
    t = m[0, 0] + m[1, 1] + m[2, 2]
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = [0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s]
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = math.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = [(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s]
    elif m[1, 1] > m[2, 2]:
        s = math.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = [(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s]
    else:
        s = math.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = [(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s]
    q = np.array(q)
    return q / np.linalg.norm(q)


def _floats(s, n=None):
    if s is None:
        return None
    if isinstance(s, (list, tuple, np.ndarray)):
        v = np.array([float(x) for x in s], dtype=np.float64)
    else:
        v = np.array([float(x) for x in str(s).split()], dtype=np.float64)
    if n is not None and v.size != n:
        raise MjcfError(f"expected {n} numbers, got {v.size}: {s!r}")
    return v


class _Compiler:
    def __init__(self, root: ET.Element):
        self.angle_deg = True  # MuJoCo default is degrees
        self.eulerseq = "xyz"
        for c in root.findall("compiler"):
            if "angle" in c.attrib:
                self.angle_deg = c.attrib["angle"].strip().lower() == "degree"
            if "eulerseq" in c.attrib:
                self.eulerseq = c.attrib["eulerseq"].strip()
            if c.attrib.get("coordinate", "local").strip().lower() != "local":
                raise MjcfError("compiler coordinate='global' is not supported")
        self.timestep = 0.002
        for o in root.findall("option"):
            if "timestep" in o.attrib:
                self.timestep = float(o.attrib["timestep"])

    def ang(self, v):
        return np.deg2rad(v) if self.angle_deg else v

    def orientation(self, attrib: Mapping[str, str]) -> np.ndarray:
        """Resolve quat / euler / axisangle / xyaxes / zaxis to a unit quaternion."""
        if "quat" in attrib:
            q = _floats(attrib["quat"], 4)
            n = np.linalg.norm(q)
            if n < 1e-14:
                raise MjcfError("zero quaternion")
            return q / n
        if "euler" in attrib:
            e = self.ang(_floats(attrib["euler"], 3))
            q = np.array([1.0, 0.0, 0.0, 0.0])
            for i, ch in enumerate(self.eulerseq):
                ax = np.zeros(3)
                ax["xyz".index(ch.lower())] = 1.0
                r = _axis_angle_quat(ax, e[i])
                # lower case: rotate about the moving (intrinsic) axes -> post-multiply;
                # upper case: fixed (extrinsic) axes -> pre-multiply.
                q = _qmul(q, r) if ch.islower() else _qmul(r, q)
            return q / np.linalg.norm(q)
        if "axisangle" in attrib:
            a = _floats(attrib["axisangle"], 4)
            return _axis_angle_quat(a[:3], float(self.ang(a[3])))
        if "xyaxes" in attrib:
            a = _floats(attrib["xyaxes"], 6)
            x = a[:3] / np.linalg.norm(a[:3])
            y = a[3:] - x * np.dot(x, a[3:])
            y = y / np.linalg.norm(y)
            z = np.cross(x, y)
            return _mat_to_quat(np.stack([x, y, z], axis=1))
        if "zaxis" in attrib:
            z = _floats(attrib["zaxis"], 3)
            z = z / np.linalg.norm(z)
            z0 = np.array([0.0, 0.0, 1.0])
            ax = np.cross(z0, z)
            s = np.linalg.norm(ax)
            ang = math.atan2(s, float(np.dot(z0, z)))
            if s < 1e-10:
                ax = np.array([1.0, 0.0, 0.0])
            return _axis_angle_quat(ax, ang)
        return np.array([1.0, 0.0, 0.0, 0.0])


def _collect_defaults(root: ET.Element) -> dict[str, dict[str, dict[str, str]]]:
    """class name -> {element tag -> attribute dict}, with nested inheritance."""
    classes: dict[str, dict[str, dict[str, str]]] = {"main": {}}

    def walk(node: ET.Element, inherited: dict[str, dict[str, str]], top: bool):
        name = node.attrib.get("class", "main" if top else None)
        if name is None:
            raise MjcfError("nested <default> without class")
        if name == "main":
            cur = classes["main"]  # several top-level <default> blocks merge into main
        else:
            cur = {tag: dict(attrs) for tag, attrs in inherited.items()}
        for child in node:
            if child.tag != "default":
                cur.setdefault(child.tag, {}).update(child.attrib)
        classes[name] = cur
        for child in node:
            if child.tag == "default":
                walk(child, cur, False)

    for d in root.findall("default"):
        walk(d, classes["main"], True)
    return classes


def compile_mjcf(
    xml: str | Path,
    *,
    sites: Mapping[str, tuple[str, Sequence[float]]] | None = None,
    scale: float = 1.0,
    legacy_joint_pos_scale: bool = False,
    from_string: bool = False,
) -> ModelTables:
    """Compile an MJCF file (or string) to :class:`ModelTables`.

    Args:
        xml: path to an MJCF file, or the XML text when ``from_string``.
        sites: ordered ``{site_name: (body_name, local_pos)}`` -- the keypoint sites the
            reference adds with ``parent.add_site`` (``stac.py:192-205``).  When ``None``
            the model's own ``<site>`` elements become the fit sites (used by the
            ``test_m_opt``-style toy models).
        scale: ``SCALE_FACTOR`` applied with the ``dm_scale_spec`` rule (``rescale.py``).
        legacy_joint_pos_scale: test-only switch reproducing the scaling rule of the build
            that produced ``demos/demo_viz.p`` (joint ``pos`` scaled only where the attribute
            is explicit in the XML); the *current* reference scales no joint ``pos``.
    """
    root = ET.fromstring(xml) if from_string else ET.parse(str(xml)).getroot()
    if root.tag != "mujoco":
        raise MjcfError("root element must be <mujoco>")
    for bad in ("include", "frame", "replicate", "attach"):
        if next(root.iter(bad), None) is not None:
            raise MjcfError(f"<{bad}> is not supported by the kinematic MJCF subset")
    comp = _Compiler(root)
    defaults = _collect_defaults(root)
    world = root.find("worldbody")
    if world is None:
        raise MjcfError("missing <worldbody>")

    body_parent = [0]
    body_pos = [np.zeros(3)]
    body_quat = [np.array([1.0, 0.0, 0.0, 0.0])]
    body_names = ["world"]
    body_depth = [0]
    body_scaled = [False]  # whether dm_scale_spec touches this body's pos
    body_joints: list[list[dict]] = [[]]
    native_sites: list[tuple[str, int, np.ndarray]] = []

    def resolved(tag: str, elem: ET.Element, childclass: str | None) -> tuple[dict, set]:
        cls = elem.attrib.get("class", childclass or "main")
        if cls not in defaults:
            raise MjcfError(f"unknown default class {cls!r}")
        attrs = dict(defaults[cls].get(tag, {}))
        attrs.update(elem.attrib)
        return attrs, set(elem.attrib)

    def visit(elem: ET.Element, parent_id: int, childclass: str | None, depth: int, scaled: bool):
        bid = len(body_parent)
        body_parent.append(parent_id)
        body_names.append(elem.attrib.get("name", f"body{bid}"))
        body_pos.append(_floats(elem.attrib.get("pos", "0 0 0"), 3))
        body_quat.append(comp.orientation(elem.attrib))
        body_depth.append(depth)
        body_scaled.append(scaled)
        body_joints.append([])
        if elem.attrib.get("mocap", "false").lower() == "true":
            raise MjcfError("mocap bodies are not supported")
        cc = elem.attrib.get("childclass", childclass)
        first_child_of_world = parent_id == 0
        for child in elem:
            if child.tag in ("joint", "freejoint"):
                if child.tag == "freejoint":
                    a, explicit = dict(child.attrib), set(child.attrib)
                    a["type"] = "free"
                else:
                    a, explicit = resolved("joint", child, cc)
                jtype = _JNT_TYPE_BY_NAME.get(a.get("type", "hinge").strip().lower())
                if jtype is None:
                    raise MjcfError(f"unknown joint type {a.get('type')!r}")
                axis = _floats(a.get("axis", "0 0 1"), 3)
                n = np.linalg.norm(axis)
                axis = axis / n if n > 1e-14 else np.array([0.0, 0.0, 1.0])
                rng = _floats(a.get("range", "0 0"), 2)
                ref = float(a.get("ref", 0.0))
                if jtype in (JNT_HINGE, JNT_BALL):
                    rng = comp.ang(rng)
                    ref = float(comp.ang(ref))
                body_joints[bid].append(
                    dict(
                        name=a.get("name", f"joint{bid}_{len(body_joints[bid])}"),
                        type=jtype,
                        pos=_floats(a.get("pos", "0 0 0"), 3),
                        pos_explicit="pos" in explicit,
                        axis=axis,
                        range=rng,
                        ref=ref,
                    )
                )
            elif child.tag == "site":
                a, _ = resolved("site", child, cc)
                native_sites.append((a.get("name", ""), bid, _floats(a.get("pos", "0 0 0"), 3)))
        for child in elem:
            if child.tag == "body":
                # dm_scale_spec: scale_bodies(worldbody.first_body()) scales the pos of every body
                # strictly below the FIRST top-level body (rescale.py:21-33,45).
                child_scaled = scaled or (first_child_of_world and bid == 1)
                visit(child, bid, cc, depth + 1, child_scaled)

    world_cc = world.attrib.get("childclass")
    for child in world:
        if child.tag == "body":
            visit(child, 0, world_cc, 1, False)
        elif child.tag == "site":
            a, _ = resolved("site", child, world_cc)
            native_sites.append((a.get("name", ""), 0, _floats(a.get("pos", "0 0 0"), 3)))

    nbody = len(body_parent)
    # -- scaling (before qpos0 is derived, like spec.compile() after dm_scale_spec) ----------
    if scale != 1.0:
        for b in range(nbody):
            if body_scaled[b]:
                body_pos[b] = body_pos[b] * scale
            if legacy_joint_pos_scale and b >= 1:
                for j in body_joints[b]:
                    if j["pos_explicit"]:
                        j["pos"] = j["pos"] * scale

    # -- joints, qpos layout --------------------------------------------------------------
    jnt_type, jnt_qposadr, jnt_bodyid, jnt_pos, jnt_axis, jnt_range, jnt_names = [], [], [], [], [], [], []
    body_jntadr = np.full(nbody, -1, np.int32)
    body_jntnum = np.zeros(nbody, np.int32)
    qpos0: list[float] = []
    for b in range(nbody):
        js = body_joints[b]
        if js:
            body_jntadr[b] = len(jnt_type)
            body_jntnum[b] = len(js)
        for j in js:
            if j["type"] == JNT_FREE and (len(js) != 1 or body_parent[b] != 0):
                raise MjcfError("free joint must be the only joint of a top-level body")
            jnt_type.append(j["type"])
            jnt_qposadr.append(len(qpos0))
            jnt_bodyid.append(b)
            jnt_pos.append(j["pos"])
            jnt_axis.append(j["axis"])
            jnt_range.append(j["range"])
            jnt_names.append(j["name"])
            if j["type"] == JNT_FREE:
                qpos0.extend(body_pos[b].tolist() + body_quat[b].tolist())
            elif j["type"] == JNT_BALL:
                qpos0.extend([1.0, 0.0, 0.0, 0.0])
            else:
                qpos0.append(j["ref"])
    njnt = len(jnt_type)

    # -- fit sites ------------------------------------------------------------------------
    site_names, site_bodyid, site_pos = [], [], []
    if sites is None:
        for name, bid, pos in native_sites:
            site_names.append(name)
            site_bodyid.append(bid)
            site_pos.append(pos)
    else:
        for name, (bname, pos) in sites.items():
            if bname not in body_names:
                raise MjcfError(f"site {name!r}: unknown body {bname!r}")
            site_names.append(name)
            site_bodyid.append(body_names.index(bname))
            site_pos.append(_floats(pos, 3))

    def f32(a, shape):
        return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(shape).astype(np.float32))

    return ModelTables(
        nbody=nbody,
        njnt=njnt,
        nq=len(qpos0),
        nsite=len(site_names),
        body_parentid=np.asarray(body_parent, np.int32),
        body_pos=f32(body_pos, (nbody, 3)),
        body_quat=f32(body_quat, (nbody, 4)),
        body_jntadr=body_jntadr,
        body_jntnum=body_jntnum,
        body_depth=np.asarray(body_depth, np.int32),
        jnt_type=np.asarray(jnt_type, np.int32).reshape(njnt),
        jnt_qposadr=np.asarray(jnt_qposadr, np.int32).reshape(njnt),
        jnt_bodyid=np.asarray(jnt_bodyid, np.int32).reshape(njnt),
        jnt_pos=f32(jnt_pos, (njnt, 3)),
        jnt_axis=f32(jnt_axis, (njnt, 3)),
        jnt_range=f32(jnt_range, (njnt, 2)),
        qpos0=f32(qpos0, (len(qpos0),)),
        site_bodyid=np.asarray(site_bodyid, np.int32).reshape(len(site_names)),
        site_pos=f32(site_pos, (len(site_names), 3)),
        body_names=body_names,
        jnt_names=jnt_names,
        site_names=site_names,
        timestep=comp.timestep,
    )


# ----------------------------------------------------------------------------------
# bounds (reference: stac.py:23-88)
# ----------------------------------------------------------------------------------
def align_joint_dims(types, ranges, names):
    """Per-qpos lower/upper bounds and joint names (restates ``_align_joint_dims``, stac.py:54-88).

    free -> [-inf x3, -1 x4] / [+inf x3, +1 x4]; other types use ``range`` repeated ``dims``
    times, ``range == (0, 0)`` meaning unconstrained (ball +-1, slide +-inf, hinge +-2pi);
    finally ``lb = min(lb, 0)`` (stac.py:88).
    """
    inf = np.float32(np.inf)
    unconstrained = {
        JNT_FREE: ([-inf] * 3 + [-1.0] * 4, [inf] * 3 + [1.0] * 4),
        JNT_BALL: ([-1.0] * 4, [1.0] * 4),
        JNT_SLIDE: ([-inf], [inf]),
        JNT_HINGE: ([np.float32(-2 * np.pi)], [np.float32(2 * np.pi)]),
    }
    lb, ub, part_names = [], [], []
    for t, r, name in zip(types, ranges, names):
        t = int(t)
        dims = JNT_QPOS_DIMS[t]
        if t == JNT_FREE:
            lo, hi = unconstrained[t]
        else:
            lo_v, hi_v = float(r[0]), float(r[1])
            if lo_v == 0 and hi_v == 0:
                lo, hi = unconstrained[t]
            else:
                lo, hi = [lo_v] * dims, [hi_v] * dims
        lb.extend(lo)
        ub.extend(hi)
        part_names += [name] * dims
    lb = np.minimum(np.asarray(lb, np.float32), np.float32(0.0))
    return lb, np.asarray(ub, np.float32), part_names
