"""ctypes mirror of ``include/cosim_model.h`` (the ModelBlob).

The struct layout is *derived from the header text* at import time, so the Python and C
sides cannot drift: every ``int``/``double`` member (scalars and fixed arrays whose
bounds are ``CS_*`` defines or literals) becomes a ctypes field of the same name, order
and shape.  ``tests/test_abi.py`` additionally checks ``sizeof`` against the built
libraries.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

import numpy as np

_HEADER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "cosim_model.h")


def _parse_header(path: str):
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    defines: Dict[str, int] = {}
    for m in re.finditer(r"^\s*#define\s+(CS_\w+)\s+(0x[0-9a-fA-F]+|\d+)\s*$", text, flags=re.M):
        defines[m.group(1)] = int(m.group(2), 0)
    body = re.search(r"typedef struct cosim_model \{(.*?)\} cosim_model_t;", text, flags=re.S).group(1)
    fields: List[Tuple[str, str, Tuple[int, ...]]] = []
    for stmt in body.split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        m = re.match(r"(int|double)\s+(.*)$", stmt, flags=re.S)
        if not m:
            raise ValueError(f"cosim_model.h: cannot parse member '{stmt}'")
        ctype = m.group(1)
        for decl in m.group(2).split(","):
            decl = decl.strip()
            dm = re.match(r"(\w+)((?:\[\w+\])*)$", decl)
            if not dm:
                raise ValueError(f"cosim_model.h: cannot parse declarator '{decl}'")
            dims = tuple(defines[d] if d in defines else int(d) for d in re.findall(r"\[(\w+)\]", dm.group(2)))
            fields.append((dm.group(1), ctype, dims))
    return defines, fields


DEFINES, FIELDS = _parse_header(_HEADER)
globals().update(DEFINES)


def _ctype(ctype: str, dims: Tuple[int, ...]):
    t = ctypes.c_int if ctype == "int" else ctypes.c_double
    for d in reversed(dims):
        t = t * d
    return t


class CosimModel(ctypes.Structure):
    """``cosim_model_t``; fill through :func:`set_field` / read through :func:`get_field`."""
    _fields_ = [(name, _ctype(ct, dims)) for name, ct, dims in FIELDS]


_FIELD_INFO = {name: (ct, dims) for name, ct, dims in FIELDS}


def field_array(model: CosimModel, name: str) -> np.ndarray:
    """Writable numpy view of an array member (shares memory with the struct)."""
    ct, dims = _FIELD_INFO[name]
    if not dims:
        raise ValueError(f"{name} is a scalar")
    dtype = np.int32 if ct == "int" else np.float64
    off = getattr(CosimModel, name).offset
    buf = (ctypes.c_char * ctypes.sizeof(model)).from_address(ctypes.addressof(model))
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(dims)), offset=off).reshape(dims)


def set_field(model: CosimModel, name: str, value) -> None:
    ct, dims = _FIELD_INFO[name]
    if not dims:
        setattr(model, name, int(value) if ct == "int" else float(value))
        return
    arr = field_array(model, name)
    v = np.asarray(value)
    if v.ndim != len(dims):
        raise ValueError(f"{name}: expected {len(dims)}-d data, got shape {v.shape}")
    if any(s > d for s, d in zip(v.shape, dims)):
        raise ValueError(f"{name}: data of shape {v.shape} exceeds the blob capacity {dims}")
    arr[tuple(slice(0, s) for s in v.shape)] = v


def get_field(model: CosimModel, name: str):
    ct, dims = _FIELD_INFO[name]
    if not dims:
        return getattr(model, name)
    return field_array(model, name)


def model_to_dict(model: CosimModel) -> dict:
    out = {}
    for name, _, dims in FIELDS:
        v = get_field(model, name)
        out[name] = np.array(v) if dims else v
    return out
