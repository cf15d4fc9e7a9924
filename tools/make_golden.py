#!/usr/bin/env python3
"""Capture golden vectors from the reference's importable pure-Python pieces.

Runs ONLY in the build container: it puts /root/reference on sys.path and drives the reference's own
``envs/wrappers.py``, ``envs/*/manager/control_manager.py``, ``envs/*/manager/xml_manager.py``,
``envs/*/utils/math_utils.py`` and ``envs/*/utils/noise_generator_utils.py`` (none of them needs
mujoco / gymnasium).  Outputs are data only — inputs and expected outputs — written to tests/golden/.
The reference Python itself is never copied.
"""
import copy
import importlib
import json
import os
import random
import shutil
import sys
import tempfile
import warnings

import numpy as np

REPO = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)
OUT = os.path.join(REPO, "tests", "golden")

from cosim_amd.config import make_config  # noqa: E402  (only used to build the config dicts)
from cosim_amd.robots import ROBOTS, obs_to_dim  # noqa: E402


class FakeEnv:
    """Scripted stand-in for the robot env: returns pre-drawn observation dicts."""

    def __init__(self, env_id, cfg, obs_seq, qpos_seq=None, done_at=None):
        self.id = env_id
        self.obs_to_dim = obs_to_dim(env_id, cfg)
        self.action_dim = self.obs_to_dim["last_action"]
        self.control_freq = 50.0
        self.obs_seq = obs_seq
        self.qpos_seq = qpos_seq
        self.t = 0
        self.done_at = done_at

    def _obs(self):
        return {k: v[self.t] for k, v in self.obs_seq.items()}

    def reset(self):
        self.t = 0
        return self._obs(), {"dt": 0.02}

    def step(self, action):
        self.t += 1
        term = self.done_at is not None and self.t == self.done_at
        return self._obs(), term, False, {"dt": 0.02}

    def event(self, event, value):
        pass

    def get_data(self):
        class D:
            pass
        d = D()
        d.qpos = self.qpos_seq[self.t]
        return d

    def render(self):
        pass

    def close(self):
        pass


def golden_wrappers():
    from envs.wrappers import CommandWrapper, StateBuildWrapper, TimeLimitWrapper
    rng = np.random.default_rng(20240601)
    T = 40
    out = {}
    meta = {}
    variants = []
    for env_id in ROBOTS:
        variants.append((f"{env_id}_default", env_id, {}, False))
    variants.append(("flamingo_light_v1_freq", "flamingo_light_v1", {"freq": {"dof_vel": 10, "ang_vel": 25}}, False))
    variants.append(("flamingo_light_v1_stack1", "flamingo_light_v1", {"stack_size": 1}, False))
    variants.append(("flamingo_light_v1_stack5", "flamingo_light_v1", {"stack_size": 5}, False))
    variants.append(("flamingo_light_v1_cmdstacked", "flamingo_light_v1", {"stacked_add": ["command"], "non_stacked": ["lin_vel"]}, False))
    variants.append(("flamingo_light_v1_poscmd", "flamingo_light_v1", {"command_dim": 2}, True))
    variants.append(("flamingo_light_v1_short", "flamingo_light_v1", {"max_duration": 0.2}, False))
    for name, env_id, mod, poscmd in variants:
        cfg = make_config(env_id, max_duration=mod.get("max_duration", 120.0), position_command=poscmd)
        ob = cfg["observation"]
        for k, f in mod.get("freq", {}).items():
            ob[k]["freq"] = f
        if "stack_size" in mod:
            ob["stack_size"] = mod["stack_size"]
        if "stacked_add" in mod:
            ob["stacked_obs_order"] = ob["stacked_obs_order"] + mod["stacked_add"]
        if "non_stacked" in mod:
            ob["non_stacked_obs_order"] = mod["non_stacked"]
            for n in mod["non_stacked"]:
                ob[n] = {"freq": 50, "scale": 2.0}
        if "command_dim" in mod:
            ob["command_dim"] = mod["command_dim"]
            ob["command_scales"] = {str(i): ob["command_scales"][str(i)] for i in range(mod["command_dim"])}
        dims = obs_to_dim(env_id, cfg)
        obs_seq = {k: rng.normal(size=(T + 1, d)) for k, d in dims.items() if k not in ("command", "height_map") and d > 0}
        obs_seq["height_map"] = [None] * (T + 1)
        qpos_seq = rng.normal(size=(T + 1, 19))
        qpos_seq[:, 3:7] /= np.linalg.norm(qpos_seq[:, 3:7], axis=1, keepdims=True)
        env = CommandWrapper(TimeLimitWrapper(StateBuildWrapper(FakeEnv(env_id, cfg, obs_seq, qpos_seq), cfg), cfg), cfg)
        cmds = rng.uniform(-1, 1, size=(T + 1, 6))
        states, flags = [], []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            env.receive_user_command(cmds[0][:ob["command_dim"]].copy())
            s, _ = env.reset()
            states.append(s.copy())
            applied = [env.applied_command.copy()]
            steps = T if "max_duration" not in mod else int(mod["max_duration"] * 50)
            for t in range(1, steps + 1):
                env.receive_user_command(cmds[t][:ob["command_dim"]].copy())
                s, term, trunc, info = env.step(np.zeros(dims["last_action"]))
                states.append(s.copy())
                flags.append((term, trunc))
                applied.append(env.applied_command.copy())
        out[f"{name}/states"] = np.array(states, dtype=np.float32)
        out[f"{name}/flags"] = np.array(flags, dtype=np.uint8)
        out[f"{name}/applied"] = np.array(applied)
        out[f"{name}/cmds"] = cmds
        out[f"{name}/qpos"] = qpos_seq
        for k, v in obs_seq.items():
            if k != "height_map":
                out[f"{name}/obs/{k}"] = v
        meta[name] = dict(env_id=env_id, mod=mod, position_command=poscmd, state_dim=int(env.state_dim),
                          cmd_slices=[[s.start, s.stop] for s in env.cmd_slices],
                          max_sim_step=int(env.env.max_sim_step), info_keys=sorted(info.keys()))
    np.savez_compressed(os.path.join(OUT, "wrappers.npz"), **out)
    json.dump(meta, open(os.path.join(OUT, "wrappers_meta.json"), "w"), indent=1, sort_keys=True)
    print("wrappers:", {k: v["state_dim"] for k, v in meta.items()})


def golden_control():
    out = {}
    for env_id in ROBOTS:
        mod = importlib.import_module(f"envs.{env_id}.manager.control_manager")
        for prob in (0.0, 0.05, 0.5, 1.0):
            cm = mod.ControlManager({"random": {"action_delay_prob": prob}})
            random.seed(7)
            acts = np.arange(30, dtype=np.float64)[:, None] * np.ones((1, 3))
            outs, us = [], []
            st = random.getstate()
            for a in acts:
                outs.append(np.array(cm.delay_filter(a.copy())))
            random.setstate(st)
            us = [random.uniform(0, 1) for _ in acts]
            out[f"{env_id}/delay_p{prob}/in"] = acts
            out[f"{env_id}/delay_p{prob}/out"] = np.array(outs)
            out[f"{env_id}/delay_p{prob}/u"] = np.array(us)
        rng = np.random.default_rng(3)
        args = rng.normal(size=(50, 6))
        out[f"{env_id}/pd/args"] = args
        out[f"{env_id}/pd/out"] = np.array([mod.ControlManager.pd_controller(*a) for a in args])
    np.savez_compressed(os.path.join(OUT, "control.npz"), **out)
    print("control: ok")


def golden_xml():
    import xml.etree.ElementTree as ET
    res = {}
    for env_id in ROBOTS:
        mod = importlib.import_module(f"envs.{env_id}.manager.xml_manager")
        tmp = tempfile.mkdtemp()
        try:
            os.makedirs(os.path.join(tmp, "assets", "xml"))
            os.makedirs(os.path.join(tmp, "manager"))
            src = os.path.join(REF, "envs", env_id, "assets", "xml", f"{env_id}.xml")
            shutil.copy(src, os.path.join(tmp, "assets", "xml"))
            for terrain, rnd in (("flat", dict(mass_noise=0.05, load=1.0, sliding_friction=0.6, torsional_friction=0.03,
                                               rolling_friction=0.02, friction_loss=0.2, precision="high")),
                                 ("rocky_hard", dict(mass_noise=0.0, load=0.0))):
                cfg = make_config(env_id, terrain=terrain, random=rnd)
                m = mod.XMLManager(cfg)
                m.cur_dir = os.path.join(tmp, "manager")
                np.random.seed(0)
                path = m.get_model_path()
                root = ET.parse(path).getroot()
                rec = {"masses": {}, "geom_friction": {}, "default_frictionloss": {}}
                for body in root.findall(".//body"):
                    for ine in body.findall("inertial"):
                        rec["masses"][body.attrib["name"]] = float(ine.attrib["mass"])
                    for g in body.findall("geom"):
                        if "friction" in g.attrib and "name" in g.attrib:
                            rec["geom_friction"][g.attrib["name"]] = g.attrib["friction"]
                for g in root.findall(".//geom"):
                    if g.attrib.get("name") == "ground":
                        rec["ground"] = {k: g.attrib.get(k) for k in ("type", "size", "hfield", "friction")}
                opt = root.find("option")
                rec["option"] = {"timestep": opt.attrib["timestep"], "iterations": opt.attrib["iterations"]}
                for d in root.findall(".//default"):
                    for j in d.findall("joint"):
                        if "frictionloss" in j.attrib:
                            rec["default_frictionloss"][d.attrib.get("class", "main")] = j.attrib["frictionloss"]
                res[f"{env_id}/{terrain}"] = rec
        finally:
            shutil.rmtree(tmp)
    json.dump(res, open(os.path.join(OUT, "xml.json"), "w"), indent=1, sort_keys=True)
    print("xml: base mass light_v1 =", res["flamingo_light_v1/flat"]["masses"]["base_link"])


def golden_math():
    from envs.flamingo_light_v1.utils.math_utils import MathUtils
    rng = np.random.default_rng(11)
    q = rng.normal(size=(64, 4))
    pg = np.array([MathUtils.quat_to_base_vel(x, np.array([0, 0, -1.0])) for x in q])       # quat in xyzw, normalised by scipy
    qn = q / np.linalg.norm(q, axis=1, keepdims=True)
    rm = np.array([MathUtils.quat_to_rot_matrix(x) for x in qn])                              # quat in wxyz, NOT normalised
    rm_raw = np.array([MathUtils.quat_to_rot_matrix(x) for x in q])
    np.savez_compressed(os.path.join(OUT, "math.npz"), quat=q, projected_gravity_xyzw=pg, rotmat_wxyz_unit=rm, rotmat_wxyz_raw=rm_raw)
    print("math: ok")


def golden_noise():
    from envs.flamingo_light_v1.utils.noise_generator_utils import truncated_gaussian_noisy_data
    import yaml
    table = yaml.safe_load(open(os.path.join(REF, "config", "random_table.yaml")))["random_table"]["sensor_noise"]
    np.random.seed(5)
    res = {}
    for level, fields in table.items():
        for field, p in fields.items():
            x = truncated_gaussian_noisy_data(np.zeros(200000), **p)
            res[f"{level}/{field}"] = dict(params=p, mean=float(x.mean()), std=float(x.std()), min=float(x.min()), max=float(x.max()),
                                           q=[float(v) for v in np.quantile(x, [0.05, 0.25, 0.5, 0.75, 0.95])])
    json.dump(res, open(os.path.join(OUT, "noise_moments.json"), "w"), indent=1, sort_keys=True)
    print("noise: ok", res["low/dof_pos"]["std"])


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    golden_wrappers()
    golden_control()
    golden_xml()
    golden_math()
    golden_noise()
