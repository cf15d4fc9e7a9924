"""MJCF subset parser for the four cosim robots.

The reference hands its (rewritten) MJCF to MuJoCo's compiler through
``MujocoEnv.__init__`` (reference ``envs/flamingo_light_v1/flamingo_light_v1.py:81-87``);
this module is the host-side replacement for the *parsing* half of that call.  It covers
exactly the elements/attributes the four robot files use: compiler,
option, nested default classes, body / inertial / joint / geom / site, mesh + hfield
assets, ``connect`` equalities, contact excludes, ``motor`` actuators and the IMU sensors.

Output is a plain ``dict`` ("spec") in MuJoCo's depth-first document order; numeric
compilation (inertia frames, qpos0 constants, invweight0 ...) lives in ``compile.py``.
"""
from __future__ import annotations

import math
import os
import xml.etree.ElementTree as ET
from typing import Dict, List, Optional

import numpy as np

# MuJoCo enum values (mjtJoint / mjtGeom) so that the blob reads like an mjModel
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = range(8)
GEOM_TYPES = {"plane": GEOM_PLANE, "hfield": GEOM_HFIELD, "sphere": GEOM_SPHERE, "capsule": GEOM_CAPSULE,
              "ellipsoid": GEOM_ELLIPSOID, "cylinder": GEOM_CYLINDER, "box": GEOM_BOX, "mesh": GEOM_MESH}

DEFAULT_SOLREF = (0.02, 1.0)
DEFAULT_SOLIMP = (0.9, 0.95, 0.001, 0.5, 2.0)


def _floats(s: Optional[str], n: Optional[int] = None, default=None) -> Optional[np.ndarray]:
    if s is None:
        return None if default is None else np.asarray(default, dtype=np.float64)
    v = np.asarray([float(x) for x in s.split()], dtype=np.float64)
    if n is not None and len(v) != n:
        raise ValueError(f"expected {n} numbers, got '{s}'")
    return v


def _partial(s: Optional[str], default) -> np.ndarray:
    """MuJoCo lets solref/solimp/friction give a prefix; the rest keeps the default."""
    out = np.asarray(default, dtype=np.float64).copy()
    if s is not None:
        v = [float(x) for x in s.split()]
        out[:len(v)] = v
    return out


def _tristate(s: Optional[str]) -> Optional[bool]:
    if s is None or s == "auto":
        return None
    return s == "true"


class _Defaults:
    """Resolved default classes: class name -> {element tag -> attrib dict}."""

    def __init__(self, root: ET.Element):
        self.classes: Dict[str, Dict[str, Dict[str, str]]] = {"main": {}}
        top = root.find("default")
        if top is not None:
            self._walk(top, "main", {})

    def _walk(self, node: ET.Element, name: str, inherited: Dict[str, Dict[str, str]]):
        cur = {k: dict(v) for k, v in inherited.items()}
        for child in node:
            if child.tag != "default":
                cur.setdefault(child.tag, {}).update(child.attrib)
        self.classes[name] = cur
        for child in node:
            if child.tag == "default":
                self._walk(child, child.attrib["class"], cur)

    def resolve(self, elem: ET.Element, childclass: Optional[str], tag: Optional[str] = None) -> Dict[str, str]:
        cls = elem.attrib.get("class", childclass or "main")
        if cls not in self.classes:
            raise ValueError(f"unknown default class '{cls}'")
        tag = tag or elem.tag
        out = dict(self.classes[cls].get(tag, {}))
        # 'motor' also inherits from <general> defaults in MuJoCo; the robots do not use them
        out.update(elem.attrib)
        return out


def parse_mjcf(path: str, root: Optional[ET.Element] = None) -> dict:
    """Parse an MJCF file (or an already-edited element tree ``root``) into a spec dict."""
    if root is None:
        root = ET.parse(path).getroot()
    if root.tag != "mujoco":
        raise ValueError(f"{path}: not an MJCF file")

    comp = root.find("compiler")
    comp_attr = {}
    for c in root.findall("compiler"):  # w4_p_v2 carries a second <compiler> (reference w4_p_v2.xml:216)
        comp_attr.update(c.attrib)
    angle_scale = 1.0 if comp_attr.get("angle", "degree") == "radian" else math.pi / 180.0
    autolimits = comp_attr.get("autolimits", "true") == "true"
    balanceinertia = comp_attr.get("balanceinertia", "false") == "true"
    del comp

    opt_e = root.find("option")
    oa = opt_e.attrib if opt_e is not None else {}
    option = {
        "timestep": float(oa.get("timestep", 0.002)),
        "gravity": _floats(oa.get("gravity"), 3, (0.0, 0.0, -9.81)),
        "iterations": int(oa.get("iterations", 100)),
        "ls_iterations": int(oa.get("ls_iterations", 50)),
        "tolerance": float(oa.get("tolerance", 1e-8)),
        "ls_tolerance": float(oa.get("ls_tolerance", 0.01)),
        "impratio": float(oa.get("impratio", 1.0)),
        "solver": oa.get("solver", "Newton"),
        "cone": oa.get("cone", "pyramidal"),
        "integrator": oa.get("integrator", "Euler"),
        "jacobian": oa.get("jacobian", "auto"),
    }

    defaults = _Defaults(root)

    bodies: List[dict] = [dict(name="world", parent=-1, pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]),
                               inertial=None, joints=[], childclass=None)]
    joints: List[dict] = []
    geoms: List[dict] = []
    sites: List[dict] = []

    def add_geom(e: ET.Element, body_id: int, childclass: Optional[str]):
        a = defaults.resolve(e, childclass)
        gtype = a.get("type", "mesh" if "mesh" in a else "sphere")
        if gtype not in GEOM_TYPES:
            raise ValueError(f"unsupported geom type '{gtype}'")
        if "fromto" in a or "euler" in a or "axisangle" in a:
            raise ValueError("geom fromto/euler/axisangle are not used by the cosim robots and not supported")
        geoms.append(dict(
            name=a.get("name", f"geom{len(geoms)}"), body=body_id, type=gtype, type_id=GEOM_TYPES[gtype],
            pos=_floats(a.get("pos"), 3, (0, 0, 0)), quat=_floats(a.get("quat"), 4, (1, 0, 0, 0)),
            size=_partial(a.get("size"), (0.0, 0.0, 0.0)),
            contype=int(a.get("contype", 1)), conaffinity=int(a.get("conaffinity", 1)),
            condim=int(a.get("condim", 3)), priority=int(a.get("priority", 0)),
            friction=_partial(a.get("friction"), (1.0, 0.005, 0.0001)),
            solref=_partial(a.get("solref"), DEFAULT_SOLREF), solimp=_partial(a.get("solimp"), DEFAULT_SOLIMP),
            solmix=float(a.get("solmix", 1.0)), margin=float(a.get("margin", 0.0)), gap=float(a.get("gap", 0.0)),
            mesh=a.get("mesh"), hfield=a.get("hfield"), group=int(a.get("group", 0)),
        ))

    def walk(e: ET.Element, parent_id: int, childclass: Optional[str]):
        for child in e:
            if child.tag == "geom":
                add_geom(child, parent_id, childclass)
            elif child.tag == "site":
                a = defaults.resolve(child, childclass)
                sites.append(dict(name=a.get("name", f"site{len(sites)}"), body=parent_id,
                                  pos=_floats(a.get("pos"), 3, (0, 0, 0)), quat=_floats(a.get("quat"), 4, (1, 0, 0, 0))))
            elif child.tag == "body":
                cc = child.attrib.get("childclass", childclass)
                bid = len(bodies)
                if "euler" in child.attrib or "axisangle" in child.attrib:
                    raise ValueError("body euler/axisangle not supported")
                body = dict(name=child.attrib.get("name", f"body{bid}"), parent=parent_id,
                            pos=_floats(child.attrib.get("pos"), 3, (0, 0, 0)),
                            quat=_floats(child.attrib.get("quat"), 4, (1, 0, 0, 0)),
                            inertial=None, joints=[], childclass=cc)
                bodies.append(body)
                ine = child.find("inertial")
                if ine is not None:
                    ia = ine.attrib
                    body["inertial"] = dict(
                        pos=_floats(ia.get("pos"), 3, (0, 0, 0)), mass=float(ia["mass"]),
                        quat=_floats(ia.get("quat"), 4, (1, 0, 0, 0)),
                        diaginertia=_floats(ia.get("diaginertia"), 3),
                        fullinertia=_floats(ia.get("fullinertia"), 6))
                for j in child:
                    if j.tag in ("joint", "freejoint"):
                        a = defaults.resolve(j, cc, tag="joint") if j.tag == "joint" else dict(j.attrib, type="free")
                        jtype = a.get("type", "hinge")
                        if jtype not in ("free", "hinge"):
                            raise ValueError(f"joint type '{jtype}' not supported (cosim robots use free + hinge)")
                        rng = _floats(a.get("range"), 2, (0, 0))
                        if jtype == "hinge":
                            rng = rng * angle_scale
                        limited = _tristate(a.get("limited"))
                        if limited is None:
                            limited = bool(autolimits and rng[0] < rng[1])
                        frcrange = _floats(a.get("actuatorfrcrange"), 2, (0, 0))
                        frclimited = _tristate(a.get("actuatorfrclimited"))
                        if frclimited is None:
                            frclimited = bool(autolimits and frcrange[0] < frcrange[1])
                        jd = dict(name=a.get("name", f"joint{len(joints)}"), body=bid, type=jtype,
                                  type_id=JNT_FREE if jtype == "free" else JNT_HINGE,
                                  pos=_floats(a.get("pos"), 3, (0, 0, 0)), axis=_floats(a.get("axis"), 3, (0, 0, 1)),
                                  range=rng, limited=limited, ref=float(a.get("ref", 0.0)) * angle_scale,
                                  damping=float(a.get("damping", 0.0)), stiffness=float(a.get("stiffness", 0.0)),
                                  frictionloss=float(a.get("frictionloss", 0.0)), armature=float(a.get("armature", 0.0)),
                                  margin=float(a.get("margin", 0.0)),
                                  solreflimit=_partial(a.get("solreflimit"), DEFAULT_SOLREF),
                                  solimplimit=_partial(a.get("solimplimit"), DEFAULT_SOLIMP),
                                  solreffriction=_partial(a.get("solreffriction"), DEFAULT_SOLREF),
                                  solimpfriction=_partial(a.get("solimpfriction"), DEFAULT_SOLIMP),
                                  actuatorfrclimited=frclimited, actuatorfrcrange=frcrange,
                                  cls=a.get("class", cc))
                        if jd["stiffness"] != 0.0:
                            raise ValueError("joint stiffness != 0 not supported")
                        body["joints"].append(len(joints))
                        joints.append(jd)
                walk(child, bid, cc)

    wb = root.find("worldbody")
    if wb is None:
        raise ValueError("MJCF has no <worldbody>")
    walk(wb, 0, None)

    # MuJoCo lists geoms/sites body by body (mjCModel::IndexAssets); bodies are already in DFS order
    geoms.sort(key=lambda g: g["body"])
    sites.sort(key=lambda s: s["body"])

    mesh_files, hfields = {}, {}
    for asset in root.findall("asset"):
        for m in asset.findall("mesh"):
            mesh_files[m.attrib.get("name", os.path.splitext(os.path.basename(m.attrib["file"]))[0])] = m.attrib["file"]
        for h in asset.findall("hfield"):
            hfields[h.attrib["name"]] = dict(file=h.attrib.get("file"), size=_floats(h.attrib["size"], 4),
                                             nrow=int(h.attrib.get("nrow", 0)), ncol=int(h.attrib.get("ncol", 0)))

    equalities = []
    for eq in root.findall("equality"):
        for c in eq:
            if c.tag != "connect":
                raise ValueError(f"equality '{c.tag}' not supported (cosim robots use connect only)")
            equalities.append(dict(body1=c.attrib["body1"], body2=c.attrib.get("body2", "world"),
                                   anchor=_floats(c.attrib["anchor"], 3),
                                   solref=_partial(c.attrib.get("solref"), DEFAULT_SOLREF),
                                   solimp=_partial(c.attrib.get("solimp"), DEFAULT_SOLIMP),
                                   active=c.attrib.get("active", "true") == "true"))

    excludes = []
    for ct in root.findall("contact"):
        for ex in ct.findall("exclude"):
            excludes.append((ex.attrib["body1"], ex.attrib["body2"]))

    actuators = []
    for act in root.findall("actuator"):
        for mtr in act:
            if mtr.tag != "motor":
                raise ValueError(f"actuator '{mtr.tag}' not supported (cosim robots use motor only)")
            a = defaults.resolve(mtr, None)
            ctrlrange = _floats(a.get("ctrlrange"), 2, (0, 0))
            ctrllimited = _tristate(a.get("ctrllimited"))
            if ctrllimited is None:
                ctrllimited = bool(autolimits and ctrlrange[0] < ctrlrange[1])
            actuators.append(dict(name=a.get("name", f"motor{len(actuators)}"), joint=a["joint"],
                                  gear=_partial(a.get("gear"), (1.0,))[0], ctrllimited=ctrllimited, ctrlrange=ctrlrange))

    sensors = []
    for sn in root.findall("sensor"):
        for s in sn:
            sensors.append(dict(type=s.tag, name=s.attrib.get("name"), site=s.attrib.get("site", s.attrib.get("objname")),
                                cutoff=float(s.attrib.get("cutoff", 0.0))))

    return dict(path=path, model=root.attrib.get("model", ""), option=option, balanceinertia=balanceinertia,
                bodies=bodies, joints=joints, geoms=geoms, sites=sites, mesh_files=mesh_files, hfields=hfields,
                equalities=equalities, excludes=excludes, actuators=actuators, sensors=sensors)
