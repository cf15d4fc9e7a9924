"""CPU twin of ONE environment of a ``BatchedEnv`` fleet.  TEST INFRASTRUCTURE (only ``tests/`` and ``bench.py``'s
``cpu_baseline`` leg import this).

``FleetEnvTwin(cfg, cm, seed, gid, gain_noise)`` steps the fp64 oracle through exactly what fleet env ``gid`` goes through
on the GPU, physics-wise: the same randomised masses and PD gains (``batched_env.draw_env_params``), the same init-noise draws
at every (auto-)reset and the same action-delay decisions (host twin of the device's Philox streams, ``cosim_amd.rng``), the
robot env's termination rule and the time limit, with gym-style auto-reset.  Sensor noise, commands and the height map only
feed the observation, which an open-loop action table never reads back, so they are not part of the twin.

Reference lines followed: manager/control_manager.py:14-23 (delay filter), flamingo_light_v1.py:131-154 (PD + do_simulation,
inside ``oracle_control_step``), :209-232 (reset_model / initial_qpos), flamingo_p_v3.py:225-233 (_is_done), envs/wrappers.py:316-318
(time limit).
"""
from __future__ import annotations

import numpy as np

from cosim_amd import rng as crng
from cosim_amd.batched_env import draw_env_params
from cosim_amd.model import get_field, set_field

from .oracle import Oracle


class FleetEnvTwin:
    def __init__(self, cfg: dict, cm, seed: int, gid: int, gain_noise: float = 0.0, auto_reset: bool = True):
        self.cfg, self.cm, self.seed, self.gid, self.auto_reset = cfg, cm, int(seed), int(gid), auto_reset
        p = draw_env_params(cfg, cm, self.seed, np.array([gid], dtype=np.uint64), gain_noise)
        self.o = Oracle(cm, body_mass=p["body_mass"][0])
        set_field(self.o.model, "ctl_kp", p["kp"][0])
        set_field(self.o.model, "ctl_kd", p["kd"][0])
        b = cm.blob
        self.nu = b.nu
        self.q0 = np.array(get_field(b, "init_qpos")[:b.nq])
        self.noise_qadr = np.array(get_field(b, "init_noise_qadr")[:b.init_noise_nq], dtype=np.int64)
        self.init_noise = float(cfg["random"]["init_noise"])
        self.delay_prob = float(cfg["random"]["action_delay_prob"])
        ptab = cfg["random_table"]["precision"][cfg["random"]["precision"]]
        self.max_sim_step = int(cfg["env"]["max_duration"] * (1.0 / (ptab["timestep"] * ptab["frame_skip"])))
        self.counter = 0          # the device's per-env launch counter (meta[1]): one per reset launch, one per control step
        self.sim_step = 0
        self.prev = np.zeros(self.nu)
        self.has_prev = False
        self.episodes_ended = 0
        self.bad_resets = 0

    def _reset_state(self, counter: int):
        q = self.q0.copy()
        if self.noise_qadr.size:
            u = crng.uniform(self.seed, np.uint64(self.gid), counter, crng.PURPOSE_INIT, np.arange(self.noise_qadr.size)).astype(np.float64)
            # the device adds fp32 noise to an fp32 qpos: init_noise * (2 u - 1) evaluated in fp32
            q[self.noise_qadr] = (q[self.noise_qadr].astype(np.float32) +
                                  np.float32(self.init_noise) * (np.float32(2.0) * u.astype(np.float32) - np.float32(1.0))).astype(np.float64)
        self.o.reset(q)
        self.sim_step, self.has_prev = 0, False

    def reset(self):
        """``env.reset()``: one launch of its own (counter + 1), init noise drawn under that launch's counter."""
        self._reset_state(self.counter)
        self.counter += 1

    def filtered(self, raw: np.ndarray) -> np.ndarray:
        """The delay filter over the next len(raw) control steps, assuming no reset in between (control_manager.py:14-23)."""
        n = len(raw)
        u = crng.uniform(self.seed, np.uint64(self.gid), self.counter + np.arange(n), crng.PURPOSE_DELAY, 0)
        delayed = np.float32(self.delay_prob) > u
        if not self.has_prev:
            delayed[0] = False
        prevs = np.concatenate([self.prev[None], raw[:-1]], axis=0)
        return np.where(delayed[:, None], prevs, raw)

    def rollout(self, raw_actions: np.ndarray) -> int:
        """len(raw_actions) control steps (auto-reset inside, like the fleet); returns the steps done."""
        raw = np.asarray(raw_actions, dtype=np.float64)
        done_total = 0
        while done_total < len(raw):
            chunk = raw[done_total:]
            if self.max_sim_step > self.sim_step:
                chunk = chunk[:self.max_sim_step - self.sim_step]
            n, terminated = self.o.rollout_env(self.filtered(chunk))
            bad = bool(self.o.bad)
            if n == 0 and not bad:
                break
            self.counter += n
            self.sim_step += n
            done_total += n
            if n > 0:
                self.prev, self.has_prev = chunk[n - 1].copy(), True
            truncated = self.sim_step == self.max_sim_step
            if bad or terminated or truncated:
                self.episodes_ended += int(terminated or truncated)
                self.bad_resets += int(bad)
                if not (self.auto_reset or bad):
                    break
                self._reset_state(self.counter - 1 if n > 0 else self.counter)   # the reset happens inside the step that ended the episode
        return done_total

    @property
    def qpos(self):
        return self.o.qpos

    @property
    def qvel(self):
        return self.o.qvel
