"""Build the ``config`` dict the reference's GUI hands to ``Tester`` / ``build_env``.

The reference has no CLI: ``ui/main_window.py:709-788`` (``_gather_config``) assembles one
nested dict from ``config/env_table.yaml``, ``config/random_table.yaml`` and widget
values, and every consumer indexes into it (schema: SURVEY.md App. C).  ``make_config``
reproduces that dict with the GUI's default widget values (``ui/main_window.py:422,
434-438,483-519``) so that headless callers get exactly what a GUI user would.
Extra keys understood by this engine only: ``config["engine"] = {num_envs, device, seed}``.
"""
from __future__ import annotations

import copy
import os
from typing import Optional

import yaml

CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "config")

OBS_TYPES = ["dof_pos", "dof_vel", "ang_vel", "lin_vel_x", "lin_vel_y", "lin_vel_z",
             "projected_gravity", "height_map", "last_action"]   # ui/main_window.py:28

GUI_RANDOM_DEFAULTS = dict(precision="medium", sensor_noise="low", init_noise=0.05, sliding_friction=0.80,
                           torsional_friction=0.02, rolling_friction=0.01, friction_loss=0.10,
                           action_delay_prob=0.05, mass_noise=0.05, load=0.0)   # ui/main_window.py:483-519

PARITY_RANDOM = dict(precision="medium", sensor_noise="none", init_noise=0.0, sliding_friction=0.80,
                     torsional_friction=0.02, rolling_friction=0.01, friction_loss=0.10,
                     action_delay_prob=0.0, mass_noise=0.0, load=0.0)   # all stochastic knobs at zero


def _to_float(v, default=None):
    try:
        return float(v)
    except (TypeError, ValueError):
        return default if default is not None else v


def load_tables():
    with open(os.path.join(CONFIG_DIR, "env_table.yaml")) as f:
        env_table = yaml.safe_load(f)
    with open(os.path.join(CONFIG_DIR, "random_table.yaml")) as f:
        random_table = yaml.safe_load(f)["random_table"]
    return env_table, random_table


def observation_defaults(env_cfg: dict) -> dict:
    """``MainWindow._make_observation_defaults`` (ui/main_window.py:100-152)."""
    cmd_cfg = env_cfg.get("command", {}) or {}
    obs_scales = env_cfg.get("obs_scales", {}) or {}
    command_scales_cfg = {str(k): _to_float(v, 1.0) for k, v in (env_cfg.get("command_scales", {}) or {}).items()}
    stacked = list(env_cfg.get("stacked_obs_order", []) or [])
    non_stacked = list(env_cfg.get("non_stacked_obs_order", []) or [])
    obs = {}
    for name in stacked + non_stacked:
        if name != "command":
            obs[name] = {"freq": 50, "scale": _to_float(obs_scales.get(name, 1.0), 1.0)}
    for name in OBS_TYPES:
        obs.setdefault(name, None)
    cmd_dim = int(cmd_cfg.get("command_dim", 6))
    command_scales = {str(i): _to_float(command_scales_cfg.get(str(i), 1.0), 1.0) for i in range(cmd_dim)}
    if "height_map" in stacked or "height_map" in non_stacked:
        hm = env_cfg.get("height_map", {}) or {}
        height_map = {"size_x": float(hm.get("size_x", 1.0)), "size_y": float(hm.get("size_y", 0.6)),
                      "res_x": int(hm.get("res_x", 15)), "res_y": int(hm.get("res_y", 9)), "freq": 50, "scale": 1.0}
    else:
        height_map = None
    out = {"stacked_obs_order": stacked, "non_stacked_obs_order": non_stacked,
           "stack_size": int(env_cfg.get("stack_size", 3)), "command_dim": cmd_dim,
           "command_scales": command_scales, "height_map": height_map}
    out.update(obs)
    if height_map is not None:
        out["height_map"] = height_map
    return out


def make_config(env_id: str, terrain: str = "flat", random: Optional[dict] = None, max_duration: float = 120.0,
                position_command: bool = False, num_envs: int = 1, seed: int = 0, device: int = 0,
                height_map: bool = False) -> dict:
    """The dict ``_gather_config`` would return for ``env_id`` with default widgets."""
    env_table, random_table = load_tables()
    if env_id not in env_table:
        raise NameError(f"Please select a valid environment id. Received '{env_id}'.")
    env_cfg = copy.deepcopy(env_table[env_id])
    if height_map and "height_map" not in env_cfg["non_stacked_obs_order"]:
        env_cfg["non_stacked_obs_order"] = list(env_cfg["non_stacked_obs_order"]) + ["height_map"]
    hardware = {}
    for k, v in (env_cfg.get("hardware", {}) or {}).items():
        hardware[k] = {kk: _to_float(vv, vv) for kk, vv in v.items()} if isinstance(v, dict) else _to_float(v, v)
    rnd = dict(GUI_RANDOM_DEFAULTS)
    if random:
        rnd.update(random)
    return {
        "env": {"id": env_id, "terrain": terrain, "max_duration": float(max_duration),
                "position_command": bool(position_command)},
        "observation": observation_defaults(env_cfg),
        "policy": {"use_lstm": False, "h_in_dim": 256, "c_in_dim": 256, "onnx_file": ""},
        "random": rnd,
        "hardware": hardware,
        "random_table": random_table,
        "engine": {"num_envs": int(num_envs), "seed": int(seed), "device": int(device)},
    }
