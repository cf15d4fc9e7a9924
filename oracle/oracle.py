"""ctypes loader for the CPU oracle (``oracle/cosim_oracle.c``).  TEST INFRASTRUCTURE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this
module.  It also holds the numpy restatement of the reference's pure-Python per-step pieces
(delay filter, observation assembly, wrappers) that the HIP path is checked against; each
function cites the reference lines it follows and is itself pinned by ``tests/golden/``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_DIR, "libcosim_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_DIR, "cosim_oracle.c")
    hdr = os.path.join(_DIR, "..", "include", "cosim_model.h")
    stale = (not os.path.isfile(_LIB_PATH)) or any(
        os.path.isfile(p) and os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _DIR, "-B", "libcosim_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.oracle_new.restype = ctypes.c_void_p
        L.oracle_new.argtypes = [ctypes.c_void_p] * 5
        L.oracle_free.argtypes = [ctypes.c_void_p]
        L.oracle_ptr.restype = ctypes.POINTER(ctypes.c_double)
        L.oracle_ptr.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        L.oracle_int.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        L.oracle_model.restype = ctypes.c_void_p
        L.oracle_model.argtypes = [ctypes.c_void_p]
        L.oracle_forward.argtypes = [ctypes.c_void_p]
        L.oracle_step.argtypes = [ctypes.c_void_p]
        L.oracle_cfrc_ext.argtypes = [ctypes.c_void_p]
        L.oracle_reset.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.oracle_contact.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.oracle_control_step.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.oracle_rollout.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.oracle_rollout_env.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.oracle_mpr_pair.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.oracle_mpr_prims.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p]
        L.oracle_set_self_collision.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.oracle_set_boxbox_mpr.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.oracle_box_box.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p,
                                     ctypes.c_void_p]
        L.oracle_ray_down.restype = ctypes.c_double
        L.oracle_ray_down.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double]
        _lib = L
    return _lib


_SHAPES = {
    "qpos": ("nq",), "qvel": ("nv",), "qacc": ("nv",), "qacc_warmstart": ("nv",), "ctrl": ("nu",),
    "xpos": ("nbody", 3), "xquat": ("nbody", 4), "xmat": ("nbody", 9), "xipos": ("nbody", 3),
    "subtree_com": ("nbody", 3), "cinert": ("nbody", 10), "cvel": ("nbody", 6), "cfrc_ext": ("nbody", 6),
    "qfrc_bias": ("nv",), "qfrc_passive": ("nv",), "qfrc_actuator": ("nv",), "qfrc_smooth": ("nv",),
    "qacc_smooth": ("nv",), "qfrc_constraint": ("nv",),
    "sensor_quat": (4,), "sensor_gyro": (3,), "sensor_vel": (3,),
}


class Oracle:
    """One environment instance of the fp64 restatement."""

    def __init__(self, compiled, body_mass: Optional[np.ndarray] = None):
        from cosim_amd.compile import env_constants
        from cosim_amd.model import CosimModel
        self.L = lib()
        assert self.L.oracle_model_sizeof() == ctypes.sizeof(CosimModel), "cosim_model_t layout mismatch"
        self.cm = compiled
        self._hv = np.ascontiguousarray(compiled.hull_vert, dtype=np.float32)
        self._ha = np.ascontiguousarray(compiled.hull_adr, dtype=np.int32)
        self._hn = np.ascontiguousarray(compiled.hull_nbr, dtype=np.int32)
        self._hf = np.ascontiguousarray(compiled.hfield, dtype=np.float32)
        self.h = self.L.oracle_new(ctypes.addressof(compiled.blob), self._hv.ctypes.data, self._ha.ctypes.data,
                                   self._hn.ctypes.data, self._hf.ctypes.data)
        if not self.h:
            raise RuntimeError("oracle_new rejected the model blob")
        from cosim_amd.model import CosimModel as CM
        self.model = CM.from_address(self.L.oracle_model(self.h))   # the oracle's private (editable) copy
        self.nq, self.nv, self.nu, self.nbody = (self.model.nq, self.model.nv, self.model.nu, self.model.nbody)
        self._nvmax = self.L.oracle_int(self.h, b"nvmax")
        if body_mass is not None:
            self.set_body_mass(body_mass)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oracle_free(self.h)
            self.h = None

    # ---- parameters
    def set_body_mass(self, body_mass: np.ndarray):
        """Per-env mass randomisation: masses plus the qpos0 constants that depend on them."""
        from cosim_amd.compile import env_constants
        from cosim_amd.model import set_field
        body_mass = np.asarray(body_mass, dtype=np.float64)
        c = env_constants(self.cm, body_mass[None])
        set_field(self.model, "body_mass", body_mass)
        set_field(self.model, "body_invweight0", c["body_invweight0"][0])
        set_field(self.model, "dof_invweight0", c["dof_invweight0"][0])
        set_field(self.model, "meaninertia", c["meaninertia"][0])

    # ---- raw views
    def view(self, name: str) -> np.ndarray:
        p = self.L.oracle_ptr(self.h, name.encode())
        if not p:
            raise KeyError(name)
        if name in ("M",):
            a = np.ctypeslib.as_array(p, shape=(self._nvmax, self._nvmax))
            return a[:self.nv, :self.nv]
        if name == "J":
            a = np.ctypeslib.as_array(p, shape=(self.L.oracle_int(self.h, b"maxefc"), self._nvmax))
            return a[:self.nefc, :self.nv]
        if name == "cdof":
            return np.ctypeslib.as_array(p, shape=(self._nvmax, 6))[:self.nv]
        if name == "efc_KBIP":
            return np.ctypeslib.as_array(p, shape=(self.L.oracle_int(self.h, b"maxefc"), 4))[:self.nefc]
        if name.startswith("efc_"):
            return np.ctypeslib.as_array(p, shape=(self.L.oracle_int(self.h, b"maxefc"),))[:self.nefc]
        dims = tuple(getattr(self, s) if isinstance(s, str) else s for s in _SHAPES[name])
        return np.ctypeslib.as_array(p, shape=dims)

    def __getattr__(self, name):
        if name in ("ncon", "nefc", "ne", "nf", "nl", "solver_niter", "ls_total", "bad", "contact_overflow"):
            return self.L.oracle_int(self.h, name.encode())
        if name in _SHAPES or name in ("M", "J", "cdof") or name.startswith("efc_"):
            return self.view(name)
        raise AttributeError(name)

    def contacts(self) -> np.ndarray:
        out = np.zeros((self.ncon, 10))   # dist, pos3, normal3, geom2, efc_address, geom1 (-1: ground)
        for i in range(self.ncon):
            self.L.oracle_contact(self.h, i, out[i].ctypes.data)
        return out

    def set_self_collision(self, on: bool):
        self.L.oracle_set_self_collision(self.h, int(on))

    def set_boxbox_mpr(self, on: bool):
        """True: box-box pairs through MPR (one contact per pair) instead of the mjc_BoxBox restatement."""
        self.L.oracle_set_boxbox_mpr(self.h, int(on))

    def mpr_pair(self, g1: int, g2: int):
        """One MPR query between robot geoms g1, g2 at the current pose: (hit, depth, dir[3] g1->g2, pos[3])."""
        out = np.zeros(7)
        rc = self.L.oracle_mpr_pair(self.h, g1, g2, out.ctypes.data)
        return rc == 0, out[0], out[1:4].copy(), out[4:7].copy()

    # ---- stepping
    def reset(self, qpos: Optional[np.ndarray] = None, qvel: Optional[np.ndarray] = None):
        qp = None if qpos is None else np.ascontiguousarray(qpos, dtype=np.float64)
        qv = None if qvel is None else np.ascontiguousarray(qvel, dtype=np.float64)
        self.L.oracle_reset(self.h, None if qp is None else qp.ctypes.data, None if qv is None else qv.ctypes.data)

    def forward(self):
        self.L.oracle_forward(self.h)

    def step(self):
        self.L.oracle_step(self.h)

    def control_step(self, filtered_action: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(filtered_action, dtype=np.float64)
        tq = np.zeros(self.nu)
        self.L.oracle_control_step(self.h, a.ctypes.data, tq.ctypes.data)
        return tq

    def rollout(self, actions: np.ndarray) -> int:
        """len(actions) control steps inside one C call; returns the steps done (stops early on a bad state)."""
        a = np.ascontiguousarray(actions, dtype=np.float64)
        assert a.ndim == 2 and a.shape[1] == self.cm.blob.nu
        return int(self.L.oracle_rollout(self.h, a.ctypes.data, a.shape[0]))

    def rollout_env(self, actions: np.ndarray):
        """Like ``rollout`` but with the robot env's ``_is_done``: stops after the control step that terminates (flamingo_p_v3's
        cfrc_ext rule) or on a bad state.  Returns (steps done, terminated)."""
        a = np.ascontiguousarray(actions, dtype=np.float64)
        assert a.ndim == 2 and a.shape[1] == self.cm.blob.nu
        term = ctypes.c_int(0)
        n = int(self.L.oracle_rollout_env(self.h, a.ctypes.data, a.shape[0], ctypes.byref(term)))
        return n, bool(term.value)

    def ray_down(self, x: float, y: float, z0: float) -> float:
        return self.L.oracle_ray_down(self.h, x, y, z0)


def box_box(pos1, mat1, size1, pos2, mat2, size2, margin=0.0):
    """mjc_BoxBox restatement on two free boxes: (points [n, 3], dist [n], normal[3] box1 -> box2)."""
    L = lib()
    p1 = np.concatenate([np.asarray(pos1, float), np.asarray(mat1, float).reshape(9)])
    p2 = np.concatenate([np.asarray(pos2, float), np.asarray(mat2, float).reshape(9)])
    s1, s2 = np.ascontiguousarray(size1, dtype=np.float64), np.ascontiguousarray(size2, dtype=np.float64)
    out, nrm = np.zeros(32), np.zeros(3)
    n = L.oracle_box_box(p1.ctypes.data, s1.ctypes.data, p2.ctypes.data, s2.ctypes.data, float(margin), out.ctypes.data, nrm.ctypes.data)
    out = out.reshape(8, 4)[:n]
    return out[:, :3].copy(), out[:, 3].copy(), nrm
