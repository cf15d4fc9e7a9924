#!/usr/bin/env python3
"""Opportunistic cross-check of the CPU oracle against MuJoCo itself.

Nothing in this build can import ``mujoco`` (no wheel, no network), so agreement with ``mj_step`` is UNPINNED.  This
script closes that gap on any machine that happens to have ``mujoco`` installed: it installs nothing, and exits with
status 77 ("skipped") when the import fails.

What it does: builds the same MJCF the engine compiles (``cosim_amd.xml_manager.XMLManager`` edits applied in
memory), replaces every ``<mesh file=...>`` by the committed convex-hull vertices (``<mesh vertex=...>``: MuJoCo takes
the hull of the vertices, which is the hull the engine uses; the raw STLs do not ship), drops the visual geoms, loads
it with ``mujoco.MjModel.from_xml_string``, and runs N control steps of the robot env's PD law (zero action and a
sinusoid) side by side with ``oracle.Oracle``.  Reported: per-step max |dqpos| / |dqvel|, contact counts, and the
RMS joint divergence over the run (BASELINE north-star: < 1e-3 rad over 1000 steps).

usage: python tools/crosscheck_mujoco.py [--env flamingo_light_v1] [--terrain flat] [--steps 1000] [--amp 0.0]
"""
import argparse
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def build_xml(config) -> str:
    """The engine's MJCF as one self-contained string (hull vertices inline, terrain PNG by absolute path)."""
    from cosim_amd.xml_manager import ASSET_DIR, XMLManager
    xm = XMLManager(config)
    root = xm.get_model_tree()
    hulls = np.load(os.path.join(ASSET_DIR, config["env"]["id"], "hulls.npz"))
    asset = root.find("asset")
    for mesh in list(asset.findall("mesh")):
        name = mesh.attrib.get("name") or os.path.splitext(os.path.basename(mesh.attrib["file"]))[0]
        key = os.path.basename(mesh.attrib.get("file", name))
        key = key if f"{key}/vert" in hulls else name
        if f"{key}/vert" not in hulls:
            asset.remove(mesh)          # visual-only mesh: its geoms are dropped below
            continue
        v = hulls[f"{key}/vert"].astype(np.float64)
        mesh.attrib.pop("file", None)
        mesh.attrib["name"] = name
        mesh.attrib["vertex"] = " ".join(f"{x:.9g}" for x in v.ravel())
    kept = {m.attrib["name"] for m in asset.findall("mesh")}
    terrain_dir = os.path.join(ASSET_DIR, "terrain")
    for hf in asset.findall("hfield"):
        hf.attrib["file"] = os.path.join(terrain_dir, os.path.basename(hf.attrib["file"]))
    for parent in root.iter():
        for g in list(parent.findall("geom")):
            if g.attrib.get("class") == "visual" or (g.attrib.get("type") == "mesh" and g.attrib.get("mesh") not in kept):
                parent.remove(g)
    comp = root.find("compiler")
    if comp is not None:
        comp.attrib.pop("meshdir", None)
    return ET.tostring(root, encoding="unicode")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="flamingo_light_v1")
    ap.add_argument("--terrain", default="flat")
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--amp", type=float, default=0.0)
    args = ap.parse_args()
    try:
        import mujoco
    except Exception as e:  # noqa: BLE001
        print(f"SKIPPED: mujoco is not importable on this host ({type(e).__name__}: {e}); nothing was installed")
        return 77
    from cosim_amd.compile import compile_model
    from cosim_amd.config import PARITY_RANDOM, make_config
    from cosim_amd.model import get_field
    from oracle.envlayer import pd_controller  # noqa: F401  (documentation: the PD law below is oracle_control_step's)
    from oracle.oracle import Oracle

    cfg = make_config(args.env, terrain=args.terrain, random=PARITY_RANDOM)
    cm = compile_model(cfg)
    b = cm.blob
    model = mujoco.MjModel.from_xml_string(build_xml(cfg))
    data = mujoco.MjData(model)
    assert model.nq == b.nq and model.nv == b.nv and model.nu == b.nu, "model sizes differ: the MJCF subsets disagree"
    q0 = np.array(get_field(b, "init_qpos")[:b.nq])
    o = Oracle(cm)
    o.reset(q0)
    mujoco.mj_resetData(model, data)
    data.qpos[:] = q0
    data.qvel[:] = 0
    mujoco.mj_forward(model, data)
    kp, kd = np.array(get_field(b, "ctl_kp")[:b.nu]), np.array(get_field(b, "ctl_kd")[:b.nu])
    scale, gear = np.array(get_field(b, "ctl_scale")[:b.nu]), np.array(get_field(b, "ctl_gear")[:b.nu])
    gamma, maxtq = np.array(get_field(b, "ctl_gamma")[:b.nu]), np.array(get_field(b, "ctl_maxtq")[:b.nu])
    vel = np.array(get_field(b, "ctl_velmode")[:b.nu]).astype(bool)
    qadr, dadr = np.array(get_field(b, "ctl_qadr")[:b.nu]), np.array(get_field(b, "ctl_dadr")[:b.nu])
    phi = np.random.default_rng(0).uniform(0, 2 * np.pi, b.nu)
    worst_q = worst_v = 0.0
    sq = 0.0
    for t in range(args.steps):
        a = np.clip(args.amp * np.sin(2 * np.pi * 0.5 * t * 0.02 + phi), -1, 1)
        # reference flamingo_light_v1.py:135-154 (and siblings): PD torque once per control step, held over frame_skip
        q, qd = data.qpos[qadr] * gear, data.qvel[dadr] * gear
        tq = np.where(vel, kd * (a * scale - qd), kp * (a * scale - q) + kd * (0.0 - qd)) * gamma
        data.ctrl[:] = np.clip(tq, -maxtq, maxtq)
        mujoco.mj_step(model, data, nstep=int(b.frame_skip))
        o.control_step(a)
        dq, dv = np.abs(data.qpos - o.qpos).max(), np.abs(data.qvel - o.qvel).max()
        worst_q, worst_v = max(worst_q, dq), max(worst_v, dv)
        sq += float(np.mean((data.qpos[7:] - o.qpos[7:]) ** 2))
        if t % max(1, args.steps // 10) == 0:
            print(f"step {t:5d}: max|dqpos| {dq:.3e} max|dqvel| {dv:.3e}  ncon mujoco {data.ncon} oracle {o.ncon}  nefc {data.nefc}/{o.nefc}")
    rms = np.sqrt(sq / args.steps)
    print(f"{args.env} {args.terrain} amp {args.amp}: {args.steps} control steps, worst |dqpos| {worst_q:.3e}, worst |dqvel| {worst_v:.3e}, "
          f"joint RMS divergence {rms:.3e} rad (north-star < 1e-3)")
    return 0 if rms < 1e-3 else 1


if __name__ == "__main__":
    sys.exit(main())
