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


NBINS = 512


class FleetReporter:
    def __init__(self, env, trace_env: Optional[int] = None, percentiles: bool = False):
        """``percentiles``: also keep one 512-bin histogram per metric column (magnitudes over [0, range): action-RMSE 0..2, base
        velocities 0..8, |torque| 0..max torque of the actuator, tracking error 0..8), summed over the sampled steps and, in
        ``summary()``, over ranks: p5 / p50 / p95 of tracking error, |torque| and action-RMSE come out of it to one bin width."""
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
        self.hist = None
        if percentiles:
            if self._lib is None:
                raise RuntimeError("FleetReporter(percentiles=True) needs the BatchedEnv GPU path (libcosim_hip.so)")
            import ctypes
            from .model import get_field
            t = env.torch
            blob = env.cm.blob
            maxtq = [float(x) for x in get_field(blob, "ctl_maxtq")[:nu]]
            self.hist_hi = np.array([2.0, 8.0, 8.0, 8.0] + [max(1e-3, m) for m in maxtq] + [8.0] * min(cd, 3), dtype=np.float32)
            self._hist_hi_dev = t.tensor(self.hist_hi, device=env.device)
            self.hist = t.zeros((len(self.names), NBINS), dtype=t.float64, device=env.device)
            self._lib.cosim_fleet_hist.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                                   ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]

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
        self._hist_rows(first, count)
        if first + count == e.num_envs:
            self.steps += 1

    def _hist_rows(self, first: int, count: int):
        if self.hist is None:
            return
        e, t = self.env, self.env.torch
        rc = self._lib.cosim_fleet_hist(e.info_buf.data_ptr() + first * e.info_buf.shape[1] * 4, count, e.info_buf.shape[1], e.action_dim,
                                        e.user_command.data_ptr() + first * e.user_command.shape[1] * 4, e.user_command.shape[1],
                                        min(e.command_dim, 3), self._hist_hi_dev.data_ptr(), NBINS, self.hist.data_ptr(),
                                        t.cuda.current_stream(e.device).cuda_stream)
        if rc != 0:
            raise RuntimeError(self._lib.cosim_last_error().decode())

    def percentiles(self, qs=(0.05, 0.5, 0.95)) -> dict:
        """p5 / p50 / p95 (or ``qs``) per metric column from the histograms, all-reduced over ranks; linear inside a bin."""
        import torch.distributed as dist
        h = self.hist.clone()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
        h = h.cpu().numpy()
        out = {}
        for i, name in enumerate(self.names):
            c = np.cumsum(h[i])
            n = c[-1]
            width = float(self.hist_hi[i]) / NBINS
            row = {}
            for q in qs:
                if n <= 0:
                    row[f"p{int(round(100 * q))}"] = float("nan")
                    continue
                b = int(np.searchsorted(c, q * n, side="left"))
                b = min(b, NBINS - 1)
                below = c[b - 1] if b > 0 else 0.0
                frac = (q * n - below) / max(h[i, b], 1e-300)
                row[f"p{int(round(100 * q))}"] = (b + min(max(frac, 0.0), 1.0)) * width
            row["bin_width"] = width
            out[name] = row
        return out

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
            self._hist_rows(0, e.num_envs)
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
        out = {"control_steps": self.steps, "envs": self.env.num_envs, "episodes_ended": self.episodes_ended, "metrics": self.acc.reduce()}
        if self.hist is not None:
            out["percentiles"] = self.percentiles()
        return out

    def save(self, path: str):
        out = self.summary()
        if self.trace:
            out["trace_env"] = self.trace_env
            out["trace"] = {k: np.asarray([r[k] for r in self.trace]).tolist() for k in self.trace[0]}
        with open(path, "w") as f:
            json.dump(out, f)
        return out
