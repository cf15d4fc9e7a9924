"""Model compiler: MJCF spec + config -> ModelBlob (``cosim_model_t``) + per-env parameters.

Host-side replacement for the numeric half of ``MjModel.from_xml_path`` (reference call
site ``envs/flamingo_light_v1/flamingo_light_v1.py:81-87``) and for the constants the
robot-env constructors derive from ``config`` (``:22-42,68-98``).  What MuJoCo's compiler
computes and the step uses is restated here in fp64 numpy [upstream user_model.cc /
engine_setconst.c]: inertial frames (``fullinertia`` -> principal axes), ``qpos0``, the
dof tree, ``connect`` anchors in both body frames, the contact-filter result per geom,
bounding spheres, and the ``qpos0``-dependent constants ``dof_invweight0``,
``body_invweight0`` and ``stat.meaninertia``.  The latter depend on the body masses and
are therefore produced *per environment* when mass noise is on (the reference recompiles
the MJCF for every construction, so every env instance has its own).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import mjcf
from .model import CosimModel, DEFINES, set_field
from .robots import ROBOTS
from .xml_manager import ASSET_DIR, XMLManager

MJ_MINVAL = 1e-15


# ----------------------------------------------------------------------------- small math
def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw])


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def mat_to_quat(m):
    """Rotation matrix -> unit quaternion (w >= 0 branch where possible)."""
    t = np.trace(m)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(m)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + m[i, i] - m[j, j] - m[k, k]) * 2
        q = np.zeros(4)
        q[0] = (m[k, j] - m[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (m[j, i] + m[i, j]) / s
        q[1 + k] = (m[k, i] + m[i, k]) / s
    return q / np.linalg.norm(q)


def axis_angle_quat(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    return np.concatenate([[np.cos(angle / 2)], axis * np.sin(angle / 2)])


def _normalize(v):
    n = np.linalg.norm(v)
    return v / n if n > 0 else v


# ----------------------------------------------------------------------------- compile
def _hull_centroid(v: np.ndarray) -> np.ndarray:
    """Volume centroid of the convex hull of `v` (tetrahedra fanned from the vertex mean)."""
    from scipy.spatial import ConvexHull
    hull = ConvexHull(v)
    o = v.mean(0)
    a, b, c = (v[hull.simplices[:, k]] - o for k in range(3))
    vol = np.abs(np.einsum("ij,ij->i", a, np.cross(b, c)))           # 6 x tetra volume
    return o + ((a + b + c) / 4.0 * vol[:, None]).sum(0) / vol.sum()


class CompiledModel:
    """ModelBlob plus the side arrays (hull vertices/graph, hfield) and name maps."""

    def __init__(self):
        self.blob = CosimModel()
        self.hull_vert = np.zeros((0, 3), dtype=np.float32)
        self.hull_adr = np.zeros((1,), dtype=np.int32)     # CSR over *global* hull vertex ids
        self.hull_nbr = np.zeros((0,), dtype=np.int32)     # neighbour ids local to the geom's hull
        self.hfield = np.zeros((0, 0), dtype=np.float32)
        self.body_names: List[str] = []
        self.joint_names: List[str] = []
        self.geom_names: List[str] = []
        self.spec: dict = {}
        self.const: dict = {}   # mass-independent pieces for per-env constants (see env_constants)


def _inertial(body: dict, balance: bool) -> Tuple[float, np.ndarray, np.ndarray, np.ndarray]:
    ine = body["inertial"]
    if ine is None:
        raise ValueError(f"body '{body['name']}' has no <inertial>; geom-inferred inertia is not supported")
    mass = ine["mass"]
    if ine["fullinertia"] is not None:
        xx, yy, zz, xy, xz, yz = ine["fullinertia"]
        full = np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])
        w, v = np.linalg.eigh(full)
        # principal moments in descending order like mju_eig3; right-handed frame
        order = np.argsort(-w)
        w, v = w[order], v[:, order]
        if np.linalg.det(v) < 0:
            v[:, 2] = -v[:, 2]
        iquat = mat_to_quat(v)
        inertia = w
    else:
        inertia = ine["diaginertia"].copy()
        iquat = _normalize(ine["quat"].copy())
    if np.any(inertia <= 0):
        raise ValueError(f"body '{body['name']}': inertia must be positive definite")
    a, b, c = inertia
    if a + b < c or a + c < b or b + c < a:
        if balance:   # compiler balanceinertia: replace by the mean [upstream user_objects.cc]
            inertia = np.full(3, inertia.mean())
        else:
            raise ValueError(f"body '{body['name']}': inertia violates A + B >= C (set balanceinertia)")
    return mass, ine["pos"].copy(), iquat, inertia


def forward_kinematics(m: dict, qpos: np.ndarray) -> dict:
    """FK on the numpy model dict (engine_core_smooth.c mj_kinematics, restated for free + hinge)."""
    nb = m["nbody"]
    xpos = np.zeros((nb, 3))
    xquat = np.zeros((nb, 4))
    xquat[0, 0] = 1
    xmat = np.zeros((nb, 3, 3))
    xmat[0] = np.eye(3)
    xanchor = np.zeros((m["njnt"], 3))
    xaxis = np.zeros((m["njnt"], 3))
    for b in range(1, nb):
        p = m["body_parentid"][b]
        jn, ja = m["body_jntnum"][b], m["body_jntadr"][b]
        if jn == 1 and m["jnt_type"][ja] == mjcf.JNT_FREE:
            qa = m["jnt_qposadr"][ja]
            xpos[b] = qpos[qa:qa + 3]
            xquat[b] = _normalize(qpos[qa + 3:qa + 7])
            xanchor[ja] = xpos[b]
            xaxis[ja] = m["jnt_axis"][ja]
        else:
            xpos[b] = xpos[p] + xmat[p] @ m["body_pos"][b]
            xquat[b] = quat_mul(xquat[p], m["body_quat"][b])
            for j in range(ja, ja + jn):
                rot = quat_to_mat(xquat[b])
                xanchor[j] = xpos[b] + rot @ m["jnt_pos"][j]
                xaxis[j] = rot @ m["jnt_axis"][j]
                ang = qpos[m["jnt_qposadr"][j]] - m["qpos0"][m["jnt_qposadr"][j]]
                xquat[b] = quat_mul(xquat[b], axis_angle_quat(m["jnt_axis"][j], ang))
                xpos[b] = xanchor[j] - quat_to_mat(xquat[b]) @ m["jnt_pos"][j]
        xquat[b] = _normalize(xquat[b])
        xmat[b] = quat_to_mat(xquat[b])
    xipos = np.array([xpos[b] + xmat[b] @ m["body_ipos"][b] for b in range(nb)])
    ximat = np.array([quat_to_mat(quat_mul(xquat[b], m["body_iquat"][b])) for b in range(nb)])
    return dict(xpos=xpos, xquat=xquat, xmat=xmat, xipos=xipos, ximat=ximat, xanchor=xanchor, xaxis=xaxis)


def body_jacobian(m: dict, fk: dict, body: int, point: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """(jacp, jacr) of a world point attached to ``body`` (engine_core_smooth.c mj_jac)."""
    nv = m["nv"]
    jacp = np.zeros((3, nv))
    jacr = np.zeros((3, nv))
    b = body
    while b > 0:
        for j in range(m["body_jntadr"][b], m["body_jntadr"][b] + m["body_jntnum"][b]):
            d = m["jnt_dofadr"][j]
            if m["jnt_type"][j] == mjcf.JNT_FREE:
                jacp[:, d:d + 3] = np.eye(3)
                for k in range(3):
                    ax = fk["xmat"][b][:, k]
                    jacr[:, d + 3 + k] = ax
                    jacp[:, d + 3 + k] = np.cross(ax, point - fk["xpos"][b])
            else:
                ax = fk["xaxis"][j]
                jacr[:, d] = ax
                jacp[:, d] = np.cross(ax, point - fk["xanchor"][j])
        b = m["body_parentid"][b]
    return jacp, jacr


def _mass_matrix_parts(m: dict, fk: dict):
    """M(qpos0) = Mconst + sum_b mass_b * Mb[b]  (mass enters only the translational part)."""
    nv, nb = m["nv"], m["nbody"]
    mconst = np.diag(m["dof_armature"][:nv].astype(np.float64))
    mb = np.zeros((nb, nv, nv))
    jps, jrs = [None] * nb, [None] * nb
    for b in range(1, nb):
        jp, jr = body_jacobian(m, fk, b, fk["xipos"][b])
        jps[b], jrs[b] = jp, jr
        iw = fk["ximat"][b] @ np.diag(m["body_inertia"][b]) @ fk["ximat"][b].T
        mconst = mconst + jr.T @ iw @ jr
        mb[b] = jp.T @ jp
    return mconst, mb, jps, jrs


def env_constants(cm: CompiledModel, body_mass: np.ndarray) -> Dict[str, np.ndarray]:
    """qpos0 constants for a batch of mass vectors (engine_setconst.c set0, restated).

    ``body_mass``: ``[N, nbody]``.  Returns ``dof_invweight0 [N, nv]``,
    ``body_invweight0 [N, nbody, 2]`` and ``meaninertia [N]``.
    """
    c = cm.const
    m = c["m"]
    nv, nb = m["nv"], m["nbody"]
    body_mass = np.atleast_2d(np.asarray(body_mass, dtype=np.float64))
    M = c["mconst"][None] + np.einsum("nb,bij->nij", body_mass, c["mb"])
    Minv = np.linalg.inv(M)
    dinv = np.einsum("nii->ni", Minv).copy()
    for j in range(m["njnt"]):
        if m["jnt_type"][j] == mjcf.JNT_FREE:   # free joint: average the 3 translational / 3 rotational entries
            d = m["jnt_dofadr"][j]
            dinv[:, d:d + 3] = dinv[:, d:d + 3].mean(axis=1, keepdims=True)
            dinv[:, d + 3:d + 6] = dinv[:, d + 3:d + 6].mean(axis=1, keepdims=True)
    binv = np.zeros((len(body_mass), nb, 2))
    for b in range(1, nb):
        jp, jr = c["jps"][b], c["jrs"][b]
        binv[:, b, 0] = np.einsum("ij,njk,ik->n", jp, Minv, jp) / 3.0
        binv[:, b, 1] = np.einsum("ij,njk,ik->n", jr, Minv, jr) / 3.0
    mean = np.einsum("nii->n", M) / nv
    return dict(dof_invweight0=dinv, body_invweight0=binv, meaninertia=mean)


def _load_hfield(path: str) -> np.ndarray:
    """PNG -> elevation in [0,1], row 0 = -y edge (MuJoCo flips image rows) [upstream user_objects.cc]."""
    from PIL import Image  # host-side asset loading only
    img = Image.open(path)
    a = np.asarray(img.convert("L"), dtype=np.float64)
    a = a[::-1].copy()
    lo, hi = a.min(), a.max()
    a = (a - lo) / (hi - lo) if hi > lo else np.zeros_like(a)
    return a.astype(np.float32)


def compile_model(config: dict, model_path: Optional[str] = None) -> CompiledModel:
    """Build the ModelBlob for ``config["env"]["id"]`` with ``XMLManager``'s edits applied."""
    env_id = config["env"]["id"]
    xm = XMLManager(config, model_path)
    root = xm.get_model_tree()
    spec = mjcf.parse_mjcf(xm.model_path, root)
    robot = ROBOTS[env_id]
    opt = spec["option"]
    if opt["integrator"] != "implicitfast":
        raise ValueError(f"integrator '{opt['integrator']}' not supported (reference models use implicitfast)")
    if opt["cone"] != "pyramidal":
        raise ValueError("only the pyramidal friction cone is supported (reference default)")
    if opt["solver"] not in ("Newton", "PGS"):
        raise ValueError(f"solver '{opt['solver']}' not supported")

    bodies, joints = spec["bodies"], spec["joints"]
    nb, nj = len(bodies), len(joints)
    if nb > DEFINES["CS_MAXBODY"] or nj > DEFINES["CS_MAXJNT"]:
        raise ValueError("model exceeds blob capacity")
    bname = {b["name"]: i for i, b in enumerate(bodies)}
    jname = {j["name"]: i for i, j in enumerate(joints)}

    m: dict = dict(nbody=nb, njnt=nj)
    m["body_parentid"] = np.array([max(b["parent"], 0) for b in bodies], dtype=np.int32)
    m["body_pos"] = np.array([b["pos"] for b in bodies])
    m["body_quat"] = np.array([_normalize(b["quat"]) for b in bodies])
    m["body_mass"] = np.zeros(nb)
    m["body_ipos"] = np.zeros((nb, 3))
    m["body_iquat"] = np.tile(np.array([1.0, 0, 0, 0]), (nb, 1))
    m["body_inertia"] = np.zeros((nb, 3))
    for i in range(1, nb):
        mass, ipos, iquat, inertia = _inertial(bodies[i], spec["balanceinertia"])
        m["body_mass"][i], m["body_ipos"][i], m["body_iquat"][i], m["body_inertia"][i] = mass, ipos, iquat, inertia
    rootid = np.zeros(nb, dtype=np.int32)
    for i in range(1, nb):
        rootid[i] = i if bodies[i]["parent"] == 0 else rootid[bodies[i]["parent"]]
    m["body_rootid"] = rootid

    # joints / dofs
    m["jnt_type"] = np.array([j["type_id"] for j in joints], dtype=np.int32)
    m["jnt_bodyid"] = np.array([j["body"] for j in joints], dtype=np.int32)
    m["jnt_pos"] = np.array([j["pos"] for j in joints]).reshape(nj, 3)
    m["jnt_axis"] = np.array([_normalize(j["axis"]) for j in joints]).reshape(nj, 3)
    m["jnt_range"] = np.array([j["range"] for j in joints]).reshape(nj, 2)
    m["jnt_limited"] = np.array([int(j["limited"]) for j in joints], dtype=np.int32)
    m["jnt_margin"] = np.array([j["margin"] for j in joints])
    m["jnt_solref"] = np.array([j["solreflimit"] for j in joints]).reshape(nj, 2)
    m["jnt_solimp"] = np.array([j["solimplimit"] for j in joints]).reshape(nj, 5)
    m["jnt_actfrclimited"] = np.array([int(j["actuatorfrclimited"]) for j in joints], dtype=np.int32)
    m["jnt_actfrcrange"] = np.array([j["actuatorfrcrange"] for j in joints]).reshape(nj, 2)
    qadr, dadr = [], []
    nq = nv = 0
    for j in joints:
        qadr.append(nq)
        dadr.append(nv)
        nq += 7 if j["type"] == "free" else 1
        nv += 6 if j["type"] == "free" else 1
    if nv > DEFINES["CS_MAXDOF"] or nq > DEFINES["CS_MAXQ"]:
        raise ValueError("model exceeds blob capacity (dofs)")
    m["nq"], m["nv"] = nq, nv
    m["jnt_qposadr"] = np.array(qadr, dtype=np.int32)
    m["jnt_dofadr"] = np.array(dadr, dtype=np.int32)
    m["body_jntnum"] = np.array([len(b["joints"]) for b in bodies], dtype=np.int32)
    m["body_jntadr"] = np.array([b["joints"][0] if b["joints"] else -1 for b in bodies], dtype=np.int32)
    body_dofnum = np.zeros(nb, dtype=np.int32)
    body_dofadr = -np.ones(nb, dtype=np.int32)
    for ji, j in enumerate(joints):
        n = 6 if j["type"] == "free" else 1
        if body_dofadr[j["body"]] < 0:
            body_dofadr[j["body"]] = dadr[ji]
        body_dofnum[j["body"]] += n
        if j["type"] == "free" and (bodies[j["body"]]["parent"] != 0 or len(bodies[j["body"]]["joints"]) != 1):
            raise ValueError("free joint must be the only joint of a top-level body")
    m["body_dofnum"], m["body_dofadr"] = body_dofnum, body_dofadr
    dof_bodyid = np.zeros(nv, dtype=np.int32)
    dof_jntid = np.zeros(nv, dtype=np.int32)
    dof_parentid = -np.ones(nv, dtype=np.int32)
    for ji, j in enumerate(joints):
        n = 6 if j["type"] == "free" else 1
        for k in range(n):
            d = dadr[ji] + k
            dof_bodyid[d], dof_jntid[d] = j["body"], ji
    for d in range(nv):
        b = dof_bodyid[d]
        if d > body_dofadr[b]:
            dof_parentid[d] = d - 1
        else:
            p = bodies[b]["parent"]
            while p > 0 and body_dofnum[p] == 0:
                p = bodies[p]["parent"]
            dof_parentid[d] = body_dofadr[p] + body_dofnum[p] - 1 if p > 0 else -1
    m["dof_bodyid"], m["dof_jntid"], m["dof_parentid"] = dof_bodyid, dof_jntid, dof_parentid
    per_dof = lambda key: np.array([joints[dof_jntid[d]][key] for d in range(nv)])
    m["dof_armature"] = per_dof("armature")
    m["dof_damping"] = per_dof("damping")
    m["dof_frictionloss"] = per_dof("frictionloss")
    m["dof_solref"] = np.array([joints[dof_jntid[d]]["solreffriction"] for d in range(nv)]).reshape(nv, 2)
    m["dof_solimp"] = np.array([joints[dof_jntid[d]]["solimpfriction"] for d in range(nv)]).reshape(nv, 5)
    # note: <joint type="free"> inherits class defaults (flamingo_p_v3.xml:24 gives the base joint armature 0.01 and
    # frictionloss 0.1 on all six dofs); only the <freejoint/> shortcut ignores defaults

    qpos0 = np.zeros(nq)
    for ji, j in enumerate(joints):
        if j["type"] == "free":
            qpos0[qadr[ji]:qadr[ji] + 3] = bodies[j["body"]]["pos"]
            qpos0[qadr[ji] + 3:qadr[ji] + 7] = _normalize(bodies[j["body"]]["quat"])
        else:
            qpos0[qadr[ji]] = j["ref"]
    m["qpos0"] = qpos0

    fk0 = forward_kinematics(m, qpos0)

    # geoms: ground + collision-enabled robot geoms
    ground = None
    rgeoms = []
    for g in spec["geoms"]:
        if g["body"] == 0:
            if g["name"] == "ground":
                ground = g
            elif g["contype"] or g["conaffinity"]:
                raise ValueError(f"static geom '{g['name']}' besides the ground is not supported")
        elif g["contype"] or g["conaffinity"]:
            rgeoms.append(g)
    if ground is None:
        raise ValueError("MJCF has no geom named 'ground' on the world body")
    if len(rgeoms) > DEFINES["CS_MAXGEOM"]:
        raise ValueError("too many collision geoms for the blob")
    ng = len(rgeoms)
    hulls = np.load(os.path.join(ASSET_DIR, env_id, "hulls.npz")) if any(g["type"] == "mesh" for g in rgeoms) else None
    hv, hadr, hnbr = [], [0], []
    g_hulladr, g_hullnum = np.zeros(ng, dtype=np.int32), np.zeros(ng, dtype=np.int32)
    g_rbound, g_rcenter = np.zeros(ng), np.zeros((ng, 3))
    g_aabb = np.zeros((ng, 6))
    g_center = np.zeros((ng, 3))
    for i, g in enumerate(rgeoms):
        rot = quat_to_mat(_normalize(g["quat"]))
        if g["type"] == "mesh":
            v = hulls[f"{g['mesh']}/vert"].astype(np.float64)
            adr, nbr = hulls[f"{g['mesh']}/adr"], hulls[f"{g['mesh']}/nbr"]
            vb = (v @ rot.T + g["pos"]).astype(np.float32)        # hull vertices in the *body* frame
            g_hulladr[i], g_hullnum[i] = sum(len(x) for x in hv), len(vb)
            base = len(hnbr)
            hadr.extend((base + adr[1:]).tolist())
            hnbr.extend(nbr.tolist())
            hv.append(vb)
            ctr = 0.5 * (vb.astype(np.float64).min(0) + vb.astype(np.float64).max(0))
            g_rcenter[i] = ctr
            g_rbound[i] = np.linalg.norm(vb.astype(np.float64) - ctr, axis=1).max()
            g_aabb[i, :3] = ctr
            g_aabb[i, 3:] = 0.5 * (vb.astype(np.float64).max(0) - vb.astype(np.float64).min(0))
            g_center[i] = _hull_centroid(vb.astype(np.float64))
        else:
            g_rcenter[i] = g["pos"]
            s = g["size"]
            if g["type"] == "sphere":
                g_rbound[i] = s[0]
            elif g["type"] == "cylinder":
                g_rbound[i] = np.hypot(s[0], s[1])
            elif g["type"] == "box":
                g_rbound[i] = np.linalg.norm(s)
            elif g["type"] == "capsule":
                g_rbound[i] = s[0] + s[1]
            else:
                raise ValueError(f"collision geom type '{g['type']}' not supported")
            g_aabb[i, :3] = g["pos"]
            ar = np.abs(rot)                                     # body-frame box around the oriented primitive
            if g["type"] == "box":
                g_aabb[i, 3:] = ar @ np.asarray(s[:3], dtype=np.float64)
            elif g["type"] == "cylinder":
                ax = rot[:, 2]
                g_aabb[i, 3:] = s[1] * np.abs(ax) + s[0] * np.sqrt(np.maximum(0.0, 1.0 - ax * ax))
            elif g["type"] == "capsule":
                g_aabb[i, 3:] = s[1] * np.abs(rot[:, 2]) + s[0]
            else:
                g_aabb[i, 3:] = g_rbound[i]
            g_center[i] = g["pos"]

    def can_collide(a, b):
        return bool((a["contype"] & b["conaffinity"]) or (b["contype"] & a["conaffinity"]))

    excl = {(bname[a], bname[b]) for a, b in spec["excludes"]} | {(bname[b], bname[a]) for a, b in spec["excludes"]}
    pairs = []
    for i in range(ng):
        for k in range(i + 1, ng):
            b1, b2 = rgeoms[i]["body"], rgeoms[k]["body"]
            if b1 == b2 or (b1, b2) in excl or not can_collide(rgeoms[i], rgeoms[k]):
                continue
            if bodies[b1]["parent"] == b2 or bodies[b2]["parent"] == b1:   # filterparent
                continue
            pairs.append((i, k))

    # equality connect: anchor2 so that both anchors coincide at qpos0 [upstream user_objects.cc mjCEquality]
    eqs = [e for e in spec["equalities"] if e["active"]]
    if len(eqs) > DEFINES["CS_MAXEQ"]:
        raise ValueError("too many equalities")
    eq_b1 = np.array([bname[e["body1"]] for e in eqs], dtype=np.int32)
    eq_b2 = np.array([bname[e["body2"]] for e in eqs], dtype=np.int32)
    eq_a1 = np.array([e["anchor"] for e in eqs]).reshape(len(eqs), 3)
    eq_a2 = np.zeros((len(eqs), 3))
    for i in range(len(eqs)):
        glob = fk0["xpos"][eq_b1[i]] + fk0["xmat"][eq_b1[i]] @ eq_a1[i]
        eq_a2[i] = fk0["xmat"][eq_b2[i]].T @ (glob - fk0["xpos"][eq_b2[i]])

    # mass-dependent constants at the nominal masses
    cm = CompiledModel()
    mconst, mb, jps, jrs = _mass_matrix_parts(m, fk0)
    cm.const = dict(m=m, mconst=mconst, mb=mb, jps=jps, jrs=jrs, fk0=fk0)
    nominal = env_constants(cm, m["body_mass"][None])

    # ---- fill the blob
    B = cm.blob
    sf = lambda k, v: set_field(B, k, v)
    sf("magic", DEFINES["CS_MODEL_MAGIC"])
    sf("magic_end", DEFINES["CS_MODEL_MAGIC"])
    for k in ("nq", "nv", "nbody", "njnt"):
        sf(k, m[k])
    sf("ngeom", ng)
    sf("neq", len(eqs))
    sf("npair", len(pairs))
    level = config["random"]["precision"]
    ptab = config["random_table"]["precision"][level]
    sf("solver", DEFINES["CS_SOLVER_NEWTON"] if opt["solver"] == "Newton" else DEFINES["CS_SOLVER_PGS"])
    sf("iterations", opt["iterations"])
    sf("ls_iterations", opt["ls_iterations"])
    sf("frame_skip", int(ptab["frame_skip"]))
    sf("timestep", opt["timestep"])
    sf("tolerance", opt["tolerance"])
    sf("ls_tolerance", opt["ls_tolerance"])
    sf("impratio", opt["impratio"])
    sf("gravity", opt["gravity"])
    sf("meaninertia", nominal["meaninertia"][0])
    for k in ("body_parentid", "body_rootid", "body_jntnum", "body_jntadr", "body_dofnum", "body_dofadr", "body_pos",
              "body_quat", "body_ipos", "body_iquat", "body_mass", "body_inertia",
              "jnt_type", "jnt_qposadr", "jnt_dofadr", "jnt_bodyid", "jnt_limited", "jnt_actfrclimited", "jnt_pos",
              "jnt_axis", "jnt_range", "jnt_margin", "jnt_solref", "jnt_solimp", "jnt_actfrcrange", "qpos0",
              "dof_bodyid", "dof_jntid", "dof_parentid", "dof_armature", "dof_damping", "dof_frictionloss",
              "dof_solref", "dof_solimp"):
        sf(k, m[k])
    sf("body_invweight0", nominal["body_invweight0"][0])
    sf("dof_invweight0", nominal["dof_invweight0"][0])

    sf("ground_type", DEFINES["CS_GEOM_PLANE"] if ground["type"] == "plane" else DEFINES["CS_GEOM_HFIELD"])
    if ground["type"] not in ("plane", "hfield"):
        raise ValueError("ground must be a plane or an hfield")
    sf("ground_contype", ground["contype"])
    sf("ground_conaffinity", ground["conaffinity"])
    sf("ground_condim", ground["condim"])
    sf("ground_friction", ground["friction"])
    sf("ground_solref", ground["solref"])
    sf("ground_solimp", ground["solimp"])
    sf("ground_solmix", ground["solmix"])
    sf("ground_margin", ground["margin"])
    sf("ground_gap", ground["gap"])
    sf("ground_pos", ground["pos"])
    if ground["type"] == "hfield":
        hf = spec["hfields"][ground["hfield"]]
        png = os.path.join(ASSET_DIR, "terrain", os.path.basename(hf["file"]))
        cm.hfield = _load_hfield(png)   # the PNG's own resolution overrides nrow/ncol [upstream]
        sf("hfield_nrow", cm.hfield.shape[0])
        sf("hfield_ncol", cm.hfield.shape[1])
        sf("hfield_size", hf["size"])
    if ng:
        sf("geom_type", np.array([g["type_id"] for g in rgeoms], dtype=np.int32))
        sf("geom_bodyid", np.array([g["body"] for g in rgeoms], dtype=np.int32))
        sf("geom_contype", np.array([g["contype"] for g in rgeoms], dtype=np.int32))
        sf("geom_conaffinity", np.array([g["conaffinity"] for g in rgeoms], dtype=np.int32))
        sf("geom_condim", np.array([g["condim"] for g in rgeoms], dtype=np.int32))
        sf("geom_ground", np.array([int(can_collide(ground, g)) for g in rgeoms], dtype=np.int32))
        sf("geom_hulladr", g_hulladr)
        sf("geom_hullnum", g_hullnum)
        sf("geom_pos", np.array([g["pos"] for g in rgeoms]))
        sf("geom_quat", np.array([_normalize(g["quat"]) for g in rgeoms]))
        sf("geom_size", np.array([g["size"] for g in rgeoms]))
        sf("geom_friction", np.array([g["friction"] for g in rgeoms]))
        sf("geom_solref", np.array([g["solref"] for g in rgeoms]))
        sf("geom_solimp", np.array([g["solimp"] for g in rgeoms]))
        sf("geom_solmix", np.array([g["solmix"] for g in rgeoms]))
        sf("geom_margin", np.array([g["margin"] for g in rgeoms]))
        sf("geom_gap", np.array([g["gap"] for g in rgeoms]))
        sf("geom_rbound", g_rbound)
        sf("geom_rcenter", g_rcenter)
        sf("geom_center", g_center)
        sf("geom_aabb", g_aabb)
    if len(pairs) > DEFINES["CS_MAXPAIR"]:
        raise ValueError(f"{len(pairs)} self-collision geom pairs exceed the blob capacity")
    if pairs:
        sf("pair_geom1", np.array([p[0] for p in pairs], dtype=np.int32))
        sf("pair_geom2", np.array([p[1] for p in pairs], dtype=np.int32))
    if eqs:
        sf("eq_body1", eq_b1)
        sf("eq_body2", eq_b2)
        sf("eq_anchor1", eq_a1)
        sf("eq_anchor2", eq_a2)
        sf("eq_solref", np.array([e["solref"] for e in eqs]))
        sf("eq_solimp", np.array([e["solimp"] for e in eqs]))

    acts = spec["actuators"]
    nu = len(acts)
    if nu > DEFINES["CS_MAXU"]:
        raise ValueError("too many actuators")
    sf("nu", nu)
    for a in acts:
        if joints[jname[a["joint"]]]["type"] != "hinge":
            raise ValueError("motors must act on hinge joints")
    sf("act_jntid", np.array([jname[a["joint"]] for a in acts], dtype=np.int32))
    sf("act_dofid", np.array([dadr[jname[a["joint"]]] for a in acts], dtype=np.int32))
    sf("act_ctrllimited", np.array([int(a["ctrllimited"]) for a in acts], dtype=np.int32))
    sf("act_gear", np.array([a["gear"] for a in acts]))
    sf("act_ctrlrange", np.array([a["ctrlrange"] for a in acts]).reshape(nu, 2))

    # IMU: the site the gyro sits on (all four robots put framequat/gyro/velocimeter on one site)
    gyro = next(s for s in spec["sensors"] if s["type"] == "gyro")
    vel = next(s for s in spec["sensors"] if s["type"] == "velocimeter")
    site = next(s for s in spec["sites"] if s["name"] == gyro["site"])
    sf("imu_bodyid", site["body"])
    sf("imu_pos", site["pos"])
    sf("imu_quat", _normalize(site["quat"]))
    sf("gyro_cutoff", gyro["cutoff"])
    sf("velocimeter_cutoff", vel["cutoff"])

    # ---- robot-env layer
    ctl = robot["control"](config["hardware"])["actuators"]
    if [c["joint"] for c in ctl] != [a["joint"] for a in acts]:
        raise ValueError("robot table and MJCF actuator order disagree")
    sf("ctl_velmode", np.array([int(c["vel"]) for c in ctl], dtype=np.int32))
    sf("ctl_qadr", np.array([qadr[jname[c["joint"]]] for c in ctl], dtype=np.int32))
    sf("ctl_dadr", np.array([dadr[jname[c["joint"]]] for c in ctl], dtype=np.int32))
    for key, fld in (("kp", "ctl_kp"), ("kd", "ctl_kd"), ("scale", "ctl_scale"), ("gear", "ctl_gear"),
                     ("gamma", "ctl_gamma"), ("maxtq", "ctl_maxtq")):
        sf(fld, np.array([float(c[key]) for c in ctl]))
    gear = float(config["hardware"].get("gear_ratio", 1.0))
    gv = lambda x: gear if x == "gear" else float(x)
    if len(robot["obs_vel"]) > DEFINES["CS_MAXOBSJ"]:
        raise ValueError("too many observed joints")
    sf("nobs_pos", len(robot["obs_pos"]))
    sf("nobs_vel", len(robot["obs_vel"]))
    sf("obs_qadr", np.array([qadr[jname[j]] for j, _ in robot["obs_pos"]], dtype=np.int32))
    sf("obs_dadr", np.array([dadr[jname[j]] for j, _ in robot["obs_vel"]], dtype=np.int32))
    sf("obs_qgear", np.array([gv(g) for _, g in robot["obs_pos"]]))
    sf("obs_dgear", np.array([gv(g) for _, g in robot["obs_vel"]]))
    info = robot["info_state"]
    sf("ninfo_state", len(info))
    sf("info_kind", np.array([0 if k == "pos" else 1 for k, _ in info], dtype=np.int32))
    sf("info_adr", np.array([qadr[jname[robot["obs_pos"][i][0]]] if k == "pos" else dadr[jname[robot["obs_vel"][i][0]]]
                             for k, i in info], dtype=np.int32))
    geared = robot.get("info_state_geared", False)
    sf("info_gear", np.array([(gv(robot["obs_pos"][i][1]) if k == "pos" else gv(robot["obs_vel"][i][1])) if geared else 1.0
                              for k, i in info]))
    init_qpos = np.zeros(nq)
    init_qpos[2] = robot["init_height"]
    init_qpos[3] = 1.0
    sf("init_qpos", init_qpos)
    if robot["init_noise_joints"] == "all_hinge":
        noise_q = [qadr[i] for i, j in enumerate(joints) if j["type"] == "hinge"]
    else:
        noise_q = [qadr[jname[j]] for j in robot["init_noise_joints"]]
    sf("init_noise_nq", len(noise_q))
    sf("init_noise_qadr", np.array(noise_q, dtype=np.int32))
    sf("term_mode", robot["term_mode"])
    sf("nterm_body", len(robot["term_bodies"]))
    if robot["term_bodies"]:
        sf("term_body", np.array([bname[b] for b in robot["term_bodies"]], dtype=np.int32))
    sf("heightmap_miss", robot["heightmap_miss"])

    if hv:
        cm.hull_vert = np.concatenate(hv).astype(np.float32)
        cm.hull_adr = np.asarray(hadr, dtype=np.int32)
        cm.hull_nbr = np.asarray(hnbr, dtype=np.int32)
    sf("nhullvert", len(cm.hull_vert))
    sf("nhulledge", len(cm.hull_nbr))
    cm.body_names = [b["name"] for b in bodies]
    cm.joint_names = [j["name"] for j in joints]
    cm.geom_names = [g["name"] for g in rgeoms]
    cm.spec = spec
    cm.xml_manager = xm
    cm.xml_root = root
    return cm
