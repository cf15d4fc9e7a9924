"""Fleet reporter: what the reference's ``Reporter`` distils from the ``info`` stream (core/reporter.py:210-218 write_info,
:257 dt, :380-382 set_points vs state, :429-442 torque / action_diff_RMSE, :506-508 command tracking), reduced over N envs.

``write_info(info)`` takes the batched ``info`` dict of ``BatchedEnv.step`` and keeps sufficient statistics on the device
(``distributed.MetricsAccumulator``: one small all-reduce across GPUs when ``summary()`` is called); ``trace_env`` keeps
the full per-step series of one env as plain numpy / floats — the dict stream the reference's single-env ``Reporter``
consumes, so its PDF code can be fed from it unchanged.
"""
from __future__ import annotations

import json
from typing import Optional

import numpy as np

from .distributed import MetricsAccumulator


class FleetReporter:
    def __init__(self, env, trace_env: Optional[int] = None):
        self.env, self.trace_env = env, trace_env
        nu, cd = env.action_dim, env.command_dim
        self.names = (["action_diff_RMSE", "lin_vel_x", "lin_vel_y", "ang_vel_yaw"] + [f"abs_torque_{i}" for i in range(nu)] +
                      [f"tracking_err_{i}" for i in range(min(cd, 3))])
        self.acc = MetricsAccumulator(self.names, device=env.device)
        self.trace = []
        self._row = None
        self._fast = hasattr(env, "info_buf") and hasattr(env, "user_command")   # BatchedEnv: the info dict is views of these
        self._lib = None
        if self._fast and str(env.device).startswith("cuda") and len(self.names) <= 32:
            import ctypes
            from .engine import load_library
            self._lib = load_library()
            self._lib.cosim_fleet_stats.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                                    ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
            self._lib.cosim_last_error.restype = ctypes.c_char_p
        self.steps = 0
        self.episodes_ended = 0

    def write_info_range(self, first: int, count: int):
        """GPU fast path for a fleet stepped as several ranges on streams of their own (``BatchedEnv.step_range``): reduce the info
        rows of envs [first, first + count) on the CURRENT stream -- the stream that range's step was launched on -- into the shared
        accumulator (double atomics), so sampling a step needs no cross-stream wait.  Call once per range; ``steps`` counts one
        sampled step when the last range (first + count == num_envs) is written."""
        if not (self._fast and self._lib is not None):
            raise RuntimeError("write_info_range needs the BatchedEnv GPU path (libcosim_hip.so)")
        t = self.env.torch
        e = self.env
        nu, cd = e.action_dim, min(e.command_dim, 3)
        if first < 0 or count <= 0 or first + count > e.num_envs:
            raise ValueError("write_info_range: range outside the fleet")
        rc = self._lib.cosim_fleet_stats(e.info_buf.data_ptr() + first * e.info_buf.shape[1] * 4, count, e.info_buf.shape[1], nu,
                                         e.user_command.data_ptr() + first * e.user_command.shape[1] * 4, e.user_command.shape[1], cd,
                                         self.acc.buf.data_ptr(), t.cuda.current_stream(e.device).cuda_stream)
        if rc != 0:
            raise RuntimeError(self._lib.cosim_last_error().decode())
        if first + count == e.num_envs:
            self.steps += 1

    def write_info(self, info):
        """``info``: the dict of ``BatchedEnv.step`` (on the GPU fast path it is only a token: the statistics are reduced from
        the env's own ``info_buf`` / ``user_command`` buffers, which the dict's entries are views of)."""
        t = self.env.torch
        nu, cd = self.env.action_dim, min(self.env.command_dim, 3)
        # one row per env: [action_diff_RMSE, lin_vel_x, lin_vel_y, ang_vel_yaw, |torque|..., |command - measured|...]
        # (command tracking as in reporter.py:506-508: applied command 0, 1 vs base linear velocity, 2 vs yaw rate)
        if self._fast and self._lib is not None:
            # BatchedEnv on a GPU: the info dict is views of info_buf / user_command -> one launch of the engine's reducer
            e = self.env
            rc = self._lib.cosim_fleet_stats(e.info_buf.data_ptr(), e.num_envs, e.info_buf.shape[1], nu, e.user_command.data_ptr(),
                                             e.user_command.shape[1], cd, self.acc.buf.data_ptr(),
                                             t.cuda.current_stream(e.device).cuda_stream)
            if rc != 0:
                raise RuntimeError(self._lib.cosim_last_error().decode())
        else:
            if self._row is None:
                self._row = t.empty((self.env.num_envs, len(self.names)), dtype=t.float32, device=self.env.device)
            self._row[:, :4] = t.cat([info["action_diff_RMSE"][:, None], info["lin_vel_x"][:, None], info["lin_vel_y"][:, None],
                                      info["ang_vel_yaw"][:, None]], dim=1)
            t.abs(info["torque"], out=self._row[:, 4:4 + nu])
            if cd:
                cmd = t.stack([info[f"user_command_{i}"] for i in range(cd)], dim=1)
                t.abs(cmd - self._row[:, 1:1 + cd], out=self._row[:, 4 + nu:4 + nu + cd])
            self.acc.update(self._row)
        self.steps += 1
        if self.trace_env is not None:
            i = self.trace_env
            row = {}
            for k, v in info.items():
                if hasattr(v, "shape") and len(v.shape) >= 1 and v.shape[0] == self.env.num_envs:
                    x = v[i].detach().cpu().numpy()
                    row[k] = x.astype(np.float64) if x.ndim else float(x)
                else:
                    row[k] = v
            self.trace.append(row)

    def note_done(self, terminated, truncated):
        self.episodes_ended += int((terminated | truncated).sum().item())

    def summary(self) -> dict:
        return {"control_steps": self.steps, "envs": self.env.num_envs, "episodes_ended": self.episodes_ended, "metrics": self.acc.reduce()}

    def save(self, path: str):
        out = self.summary()
        if self.trace:
            out["trace_env"] = self.trace_env
            out["trace"] = {k: np.asarray([r[k] for r in self.trace]).tolist() for k in self.trace[0]}
        with open(path, "w") as f:
            json.dump(out, f)
        return out
