"""``BatchedEnv`` — N environment instances on one MI355X behind the reference's env API.

Host-side mirror of the reference's wrapper stack for the batched case: one object plays
``CommandWrapper(TimeLimitWrapper(StateBuildWrapper(<Robot>(config))))`` (reference
``envs/build.py:8-24``) for ``num_envs`` instances at once.  Method names, argument meaning
and error behaviour follow ``envs/wrappers.py``; tensors carry a leading ``[N]`` dimension and
live on the GPU (torch is used for device memory and streams only).  All per-step arithmetic —
command transform, delay filter, PD law, physics substeps, observation build — runs inside
``cosim_step`` (one HIP kernel launch per control step).
"""
from __future__ import annotations

from typing import Dict, Optional

import os

import numpy as np

from . import rng as crng
from .compile import CompiledModel, compile_model, env_constants
from .engine import Engine, make_obs_config
from .model import get_field
from .robots import ROBOTS, obs_to_dim as robot_obs_to_dim


def _cmd_slices(stacked, non_stacked, dims, stack_size, command_dim):
    """``StateBuildWrapper._get_cmd_index_cache`` (wrappers.py:129-158)."""
    out = []
    if command_dim <= 0:
        return out
    stacked_dim = sum(dims[n] for n in stacked)
    off, starts = 0, []
    for n in stacked:
        if n == "command":
            starts.append(off)
        off += dims[n]
    for k in range(stack_size):
        for s in starts:
            out.append(slice(k * stacked_dim + s, k * stacked_dim + s + command_dim))
    base, off = stack_size * stacked_dim, 0
    for n in non_stacked:
        if n == "command":
            out.append(slice(base + off, base + off + command_dim))
        off += dims[n]
    return out


def draw_env_params(config: dict, cm: CompiledModel, seed: int, gids, gain_noise: float = 0.0) -> Dict[str, np.ndarray]:
    """Per-env domain randomisation as pure functions of (seed, global env id): body masses (XMLManager step 3, reference
    manager/xml_manager.py:43-55: ``mass += U(-m k, +m k)`` on the listed bodies, ``+ load`` on the base) and PD gains
    (table value x U(1 - g, 1 + g)).  Shared by ``BatchedEnv`` and by the CPU twin of a fleet env (oracle/fleet.py)."""
    blob = cm.blob
    gids = np.asarray(gids, dtype=np.uint64)
    N, nb, nu = len(gids), blob.nbody, blob.nu
    mass = np.tile(np.array(get_field(blob, "body_mass")[:nb]), (N, 1))
    k, load = config["random"]["mass_noise"], config["random"]["load"]
    robot = ROBOTS[config["env"]["id"]]
    for name in robot["mass_bodies"]:
        b = cm.body_names.index(name)
        m0 = mass[:, b].copy()
        u = crng.uniform(seed, gids, 0, crng.PURPOSE_MASS, b).astype(np.float64)
        mass[:, b] = m0 + (2.0 * u - 1.0) * m0 * k
        if name == robot["base_body"]:
            mass[:, b] += load
    kp = np.tile(np.array(get_field(blob, "ctl_kp")[:nu]), (N, 1))
    kd = np.tile(np.array(get_field(blob, "ctl_kd")[:nu]), (N, 1))
    if gain_noise > 0:
        idx = np.arange(nu)[None, :]
        up = crng.uniform(seed, gids[:, None], 0, crng.PURPOSE_GAIN, idx)
        ud = crng.uniform(seed, gids[:, None], 1, crng.PURPOSE_GAIN, idx)
        kp = kp * (1.0 + gain_noise * (2.0 * up - 1.0))
        kd = kd * (1.0 + gain_noise * (2.0 * ud - 1.0))
    return {"body_mass": mass, "kp": kp, "kd": kd}


class _Data:
    """What ``get_data()`` hands out: batched ``qpos`` / ``qvel`` views (reference: the MjData object)."""

    def __init__(self, qpos, qvel):
        self.qpos, self.qvel = qpos, qvel


class BatchedEnv:
    def __init__(self, config: dict, num_envs: Optional[int] = None, device: Optional[int] = None, seed: Optional[int] = None,
                 auto_reset: bool = True, env_id0: int = 0, gain_noise: float = 0.0, compiled: Optional[CompiledModel] = None,
                 ranges: Optional[int] = None, deferred_join: Optional[bool] = None):
        """``ranges`` > 1: ``step()`` issues the fleet as that many launches over contiguous env ranges on engine-owned HIP streams
        (``cosim_set_param "ranges"``).  With ``deferred_join`` the caller's stream is NOT made to wait for them inside ``step()``:
        call ``join()`` before consuming ``state`` / ``terminated`` / ``info`` on the current stream (``get_data``, ``reset``,
        ``event``, ``set_state`` and ``solver_stats`` join by themselves).  That is what lets a range's next control step overlap
        the tail of the others' current one; it fits callers whose next action does not need the whole fleet's last state (an
        action table; a policy evaluated per range on ``range_streams``).  With a deferred join the ``action`` tensor of a step must
        stay untouched until that step has run (at most two steps are in flight: an action table or three rotating buffers), and
        ``receive_user_command`` joins first.  Defaults: ``config["engine"]`` / 1 / False."""
        import torch  # plumbing only

        eng_cfg = config.get("engine", {})
        self.config = config
        self.id = config["env"]["id"]
        if self.id not in ROBOTS:
            raise NameError(f"Please select a valid environment id. Received '{self.id}'.")
        self.num_envs = int(num_envs if num_envs is not None else eng_cfg.get("num_envs", 1))
        self.seed = int(seed if seed is not None else eng_cfg.get("seed", 0))
        dev = int(device if device is not None else eng_cfg.get("device", 0))
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedEnv needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.torch = torch
        self.device = torch.device(f"cuda:{dev}")
        self.env_id0 = int(env_id0)

        # robot-env constructor checks (flamingo_light_v1.py:36-42)
        level = config["random"]["precision"]
        ptab = config["random_table"]["precision"][level]
        self.dt_ = ptab["timestep"]
        self.frame_skip = ptab["frame_skip"]
        self.control_freq = 1 / (self.dt_ * self.frame_skip)
        assert self.control_freq == 50, "Currently, only control frequency of 50 is supported."

        self.cm = compiled if compiled is not None else compile_model(config)
        blob = self.cm.blob
        self.action_dim = blob.nu
        self.nq, self.nv = blob.nq, blob.nv
        self.obs_to_dim = robot_obs_to_dim(self.id, config)
        ob = config["observation"]
        self.command_dim = ob["command_dim"]
        assert self.command_dim >= 0, "command_dim must be equal or greater than 0."
        self.stack_size = int(ob["stack_size"])
        self.stacked_obs_order = list(ob["stacked_obs_order"])
        self.non_stacked_obs_order = list(ob["non_stacked_obs_order"])
        self._stacked_obs_dim = sum(self.obs_to_dim[n] for n in self.stacked_obs_order)
        self._non_stacked_obs_dim = sum(self.obs_to_dim[n] for n in self.non_stacked_obs_order)
        self.state_dim = self.stack_size * self._stacked_obs_dim + self._non_stacked_obs_dim
        self.cmd_slices = _cmd_slices(self.stacked_obs_order, self.non_stacked_obs_order, self.obs_to_dim,
                                      self.stack_size, self.command_dim)
        self.max_sim_step = int(config["env"]["max_duration"] * self.control_freq)
        self.auto_reset = bool(auto_reset)

        obs_cfg = make_obs_config(config, self.obs_to_dim, self.control_freq, self.auto_reset)
        self.engine = Engine(self.cm, obs_cfg, self.num_envs, dev, self.seed, self.env_id0)
        assert self.engine.query("state_dim") == self.state_dim
        self.info_dim = self.engine.query("info_dim")
        # kernel variant: COSIM_ENVS_PER_WAVE=1|2 overrides the engine's choice where the variant exists (A/B runs)
        prio = os.environ.get("COSIM_WAVE_PRIORITY")           # "base,t1,t2,t3" (tuning runs)
        if prio:
            self.engine.set_param("wave_priority", np.array([float(x) for x in prio.split(",")]))
        if os.environ.get("COSIM_CONTACT_TWIST") == "1":       # dense-row kernel -> its contact-twist variant (A/B runs)
            try:
                self.engine.set_param("contact_twist", np.array([1.0]))
            except (ValueError, RuntimeError):
                pass
        if os.environ.get("COSIM_LS_SCALE"):                   # line-search gradient tolerance multiplier (tuning runs)
            self.engine.set_param("ls_tolerance_scale", np.array([float(os.environ["COSIM_LS_SCALE"])]))
        if os.environ.get("COSIM_PAIR_MODE"):                  # "1": hull pairs wave-cooperative (A/B runs)
            self.engine.set_param("pair_mode", np.array([float(os.environ["COSIM_PAIR_MODE"])]))
        for var, name in (("COSIM_SPLIT", "split"), ("COSIM_NARROW_WAVES", "narrow_waves"), ("COSIM_NARROW_OCC", "narrow_occupancy")):
            if os.environ.get(var):                            # split pipeline of the heightfield humanoid kernels (A/B and tuning runs)
                try:
                    self.engine.set_param(name, np.array([float(os.environ[var])]))
                except (ValueError, RuntimeError):
                    pass
        epw = os.environ.get("COSIM_ENVS_PER_WAVE")
        if epw:
            try:
                self.engine.set_param("envs_per_wave", np.array([float(epw)]))
            except (ValueError, RuntimeError):
                pass

        self.ranges = int(ranges if ranges is not None else os.environ.get("COSIM_RANGES", eng_cfg.get("ranges", 1)))
        self.ranges = max(1, min(self.ranges, self.num_envs, 16))
        self.deferred_join = bool(deferred_join if deferred_join is not None else eng_cfg.get("deferred_join", False))
        self.range_list = [(0, self.num_envs)]
        self.range_streams = [None]
        if self.ranges > 1:
            self.engine.set_param("ranges", np.array([float(self.ranges)]))
            self.engine.set_param("deferred_join", np.array([float(self.deferred_join)]))
            if os.environ.get("COSIM_INFLIGHT"):               # steps the host may run ahead of each range stream (tuning runs)
                self.engine.set_param("inflight", np.array([float(os.environ["COSIM_INFLIGHT"])]))
            rl = [self.engine.range(i) for i in range(self.ranges)]
            self.range_list = [(f, c) for f, c, _ in rl]
            self.range_streams = [torch.cuda.ExternalStream(st, device=self.device) for _, _, st in rl]

        self._randomise(gain_noise)

        N, t = self.num_envs, torch
        f32 = dict(dtype=t.float32, device=self.device)
        self.state = t.zeros((N, self.state_dim), **f32)
        self.terminated = t.zeros((N,), dtype=t.uint8, device=self.device)
        self.truncated = t.zeros((N,), dtype=t.uint8, device=self.device)
        self.info_buf = t.zeros((N, self.info_dim), **f32)
        self.user_command = t.zeros((N, max(self.command_dim, 1)), **f32)
        self._qpos = t.zeros((N, self.nq), **f32)
        self._qvel = t.zeros((N, self.nv), **f32)
        self.reset_flag = False

    # ------------------------------------------------------------------ domain randomisation (XMLManager step 3)
    def _randomise(self, gain_noise: float):
        gids = np.arange(self.env_id0, self.env_id0 + self.num_envs, dtype=np.uint64)
        p = draw_env_params(self.config, self.cm, self.seed, gids, gain_noise)
        self.body_mass, self.kp, self.kd = p["body_mass"], p["kp"], p["kd"]
        c = env_constants(self.cm, self.body_mass)
        self.engine.set_param("body_mass", self.body_mass)
        self.engine.set_param("body_invweight0", c["body_invweight0"][:, :, 0])
        self.engine.set_param("dof_invweight0", c["dof_invweight0"])
        self.engine.set_param("meaninertia", c["meaninertia"])
        if gain_noise > 0:
            self.engine.set_param("kp", self.kp)
            self.engine.set_param("kd", self.kd)

    # ------------------------------------------------------------------ BaseEnv API (wrappers.py:8-85), batched
    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def _cmd_ptr(self):
        return self.user_command.data_ptr() if self.command_dim > 0 else None

    def receive_user_command(self, user_command):
        """Store the raw user command(s); scaling / position-mode transform happen in the next kernel launch
        from the pre-step pose, exactly where ``CommandWrapper.receive_user_command`` reads ``get_data()``."""
        t = self.torch
        if not t.is_tensor(user_command):
            # the same command as last time (the reference's loop re-sends it every step, core/tester.py:68): nothing to write
            host = np.ascontiguousarray(user_command, dtype=np.float32)
            last = getattr(self, "_last_cmd_host", None)
            if last is not None and last.shape == host.shape and np.array_equal(last, host):
                return
            self._last_cmd_host = host.copy()
        else:
            self._last_cmd_host = None
        uc = t.as_tensor(user_command, dtype=t.float32, device=self.device)
        if uc.ndim == 1:
            uc = uc[None, :].expand(self.num_envs, -1)
        if self.config["env"]["position_command"] is not False:
            assert self.command_dim == 2, f"Currently, position command only support 2 dimenstion, but got {self.command_dim}."
        if self.ranges > 1 and self.deferred_join:
            self.engine.join(self._stream())          # steps in flight still read the command buffer: write it behind them
        self.user_command[:, :self.command_dim] = uc[:, :self.command_dim]

    def reset(self, mask=None):
        t = self.torch
        mptr = None
        if mask is not None:
            mask = t.as_tensor(mask, device=self.device).to(t.uint8).contiguous()
            mptr = mask.data_ptr()
        self.engine.reset(mptr, self._cmd_ptr(), self.state.data_ptr(), self._stream())
        self.reset_flag = True
        return self.state, {}

    def step(self, action):
        assert self.reset_flag is True, "Call 'reset()' before calling 'step()'."
        t = self.torch
        a = t.as_tensor(action, dtype=t.float32, device=self.device)
        if a.shape != (self.num_envs, self.action_dim):
            raise ValueError(f"Action dimension mismatch. Expected {(self.num_envs, self.action_dim)}, found {tuple(a.shape)}")
        a = a.contiguous()
        self.engine.step(a.data_ptr(), self._cmd_ptr(), self.state.data_ptr(), self.terminated.data_ptr(),
                         self.truncated.data_ptr(), self.info_buf.data_ptr(), self._stream())
        if self.command_dim < 0 or self.command_dim > 6:
            raise ValueError(f"Invalid 'command_dim': expected 0> or <7; but got {self.command_dim}.")
        return self.state, self.terminated, self.truncated, self._info(a)

    def rollout(self, actions, info: bool = True):
        """``len(actions)`` control steps of the whole fleet under an action table ``[K, N, action_dim]`` in one launch per range
        (``cosim_rollout``: the reference's ``Tester.test`` loop, core/tester.py:66-97, with the policy replaced by a lookup).
        Returns ``(states [K, N, state_dim], terminated [K, N], truncated [K, N], info_buf [K, N, info_dim] or None)``: row k is what
        ``step(actions[k])`` would have returned (auto-reset included).  ``self.state`` etc. keep the last row."""
        assert self.reset_flag is True, "Call 'reset()' before calling 'step()'."
        t = self.torch
        a = t.as_tensor(actions, dtype=t.float32, device=self.device).contiguous()
        if a.ndim != 3 or tuple(a.shape[1:]) != (self.num_envs, self.action_dim):
            raise ValueError(f"Action table mismatch. Expected [K, {self.num_envs}, {self.action_dim}], found {tuple(a.shape)}")
        K = a.shape[0]
        states = t.empty((K, self.num_envs, self.state_dim), dtype=t.float32, device=self.device)
        term = t.empty((K, self.num_envs), dtype=t.uint8, device=self.device)
        trunc = t.empty((K, self.num_envs), dtype=t.uint8, device=self.device)
        inf = t.empty((K, self.num_envs, self.info_dim), dtype=t.float32, device=self.device) if info else None
        self.engine.rollout(K, a.data_ptr(), self._cmd_ptr(), states.data_ptr(), term.data_ptr(), trunc.data_ptr(),
                            inf.data_ptr() if info else None, self._stream())
        self.state.copy_(states[-1]); self.terminated.copy_(term[-1]); self.truncated.copy_(trunc[-1])
        if info:
            self.info_buf.copy_(inf[-1])
        return states, term, trunc, inf

    def step_range(self, first: int, count: int, action):
        """One control step of envs ``[first, first + count)`` on the current stream (``cosim_step_range``): ``action`` is the
        whole fleet's ``[N, action_dim]`` tensor; outputs land in the fleet's ``state`` / ``terminated`` / ``truncated`` /
        ``info_buf`` rows of that range.  Stepping the fleet as S such shards on S streams pipelines control steps across shards."""
        assert self.reset_flag is True, "Call 'reset()' before calling 'step()'."
        if tuple(action.shape) != (self.num_envs, self.action_dim) or not action.is_contiguous():
            raise ValueError(f"Action dimension mismatch. Expected contiguous {(self.num_envs, self.action_dim)}, found {tuple(action.shape)}")
        self.engine.step_range(first, count, action.data_ptr(), self._cmd_ptr(), self.state.data_ptr(), self.terminated.data_ptr(),
                               self.truncated.data_ptr(), self.info_buf.data_ptr(), self._stream())

    def join(self):
        """Deferred join: make the current stream wait for every range stream's work so far (``cosim_join``)."""
        self.engine.join(self._stream())

    def range_mark(self, i: int):
        """After enqueuing work of your own on ``range_streams[i]``: the next ``join()`` waits for it too (``cosim_range_mark``)."""
        self.engine.range_mark(i)

    def _info(self, action) -> Dict[str, object]:
        """Batched ``_get_info`` + ``user_command_i`` (flamingo_light_v1.py:166-183; wrappers.py:399-400)."""
        info = getattr(self, "_info_views", None)
        if info is None:   # views of the fleet's fixed buffers: built once, handed out every step (only "action" changes)
            nu, b = self.action_dim, self.info_buf
            info = {
                "dt": self.dt_ * self.frame_skip,
                "action": action,
                "action_diff_RMSE": b[:, 0],
                "lin_vel_x": b[:, 1],
                "lin_vel_y": b[:, 2],
                "ang_vel_yaw": b[:, 3],
                "torque": b[:, 4:4 + nu],
                "set_points": b[:, 4 + nu:4 + 2 * nu],
                "state": b[:, 4 + 2 * nu:],
            }
            for i in range(self.command_dim):
                info[f"user_command_{i}"] = self.user_command[:, i]
            self._info_views = info
        info = dict(info)
        info["action"] = action
        return info

    def event(self, event: str, value, mask=None):
        if event == "push":
            t = self.torch
            v = t.as_tensor(value, dtype=t.float32, device=self.device).reshape(-1, 3)
            if v.shape[0] == 1:
                v = v.expand(self.num_envs, 3)
            v = v.contiguous()
            mptr = None
            if mask is not None:
                mask = t.as_tensor(mask, device=self.device).to(t.uint8).contiguous()
                mptr = mask.data_ptr()
            self.engine.push(v.data_ptr(), mptr, self._stream())
        else:
            raise NotImplementedError(f"event:{event} is not supported.")

    def get_data(self):
        self.engine.get("qpos", self._qpos.data_ptr(), self._stream())
        self.engine.get("qvel", self._qvel.data_ptr(), self._stream())
        return _Data(self._qpos, self._qvel)

    def set_state(self, qpos=None, qvel=None, qacc_warmstart=None):
        """Test / checkpoint hook: overwrite the physics state of all envs."""
        t = self.torch
        for name, v in (("qpos", qpos), ("qvel", qvel), ("qacc_warmstart", qacc_warmstart)):
            if v is not None:
                x = t.as_tensor(v, dtype=t.float32, device=self.device).contiguous()
                self.engine.set(name, x.data_ptr(), self._stream())
                t.cuda.synchronize(self.device)

    def solver_stats(self):
        """Cumulative solver counters since creation (fleet sums): control steps, constraint rows (summed over
        substeps), Newton iterations, line-search evaluations, Hessian factorisations, non-finite resets."""
        t = self.torch
        buf = t.zeros((self.num_envs, 16), dtype=t.float32, device=self.device)
        self.engine.get("meta", buf.data_ptr(), self._stream())
        t.cuda.synchronize(self.device)
        mi = buf.view(t.int32).to(t.int64)
        m = mi.sum(dim=0).cpu().numpy()
        # dropped_*: contacts / limit rows that found no slot (0 unless an env was stepped with a truncated constraint set);
        # max_contacts: most contacts detected in one substep by any env of the fleet
        return {"step_count": int(m[1]), "rows": int(m[3]), "nan_resets": int(m[4]), "newton_iters": int(m[5]),
                "ls_evals": int(m[6]), "factorisations": int(m[7]), "dropped_contacts": int(m[8]), "dropped_limit_rows": int(m[9]),
                "max_contacts": int(mi[:, 10].max().item()), "episodes_ended": int(m[11]),
                # control steps redone by the large-capacity kernel because their contacts did not fit the fleet kernel's slots
                "fixup_steps": int(m[12]),
                # heightfield: geoms whose prism walk hit the 32768-prism bound (counted in dropped_contacts too)
                "truncated_walks": int(m[13])}

    def render(self):
        pass  # headless

    def close(self):
        self.engine.close()
