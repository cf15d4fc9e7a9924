"""Batched on-device policies: the counterpart of the reference's ``core/policy.py:5-54`` for N environments.

The reference runs an ONNX file through onnxruntime on the CPU, one state at a time (``MLPPolicy.get_action`` :11-21,
``LSTMPolicy.get_action`` :34-47, ``build_policy`` :49-53).  Here the same contract — ``get_action(state) -> action``
clipped to [-1, 1], LSTM ``h``/``c`` carried between calls — is kept over ``[N, state_dim]`` device tensors, so the
rollout loop never leaves the GPU.  onnxruntime and the ``onnx`` package are not available in this
build, so the file is read with a small protobuf wire-format decoder (``read_onnx``) and the graph is run by a tiny
interpreter over the operator subset RL policy exports use (Gemm / MatMul / Add / activations / LSTM / shape glue).
The GEMMs go through torch (rocBLAS / hipBLASLt): plain library matrix products, not part of the §8 hot path.

PARITY UNPINNED against onnxruntime (no policy file ships with the reference; onnxruntime is absent): the
tests compare with a numpy evaluation of the same graph.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Optional

import numpy as np

# ---------------------------------------------------------------------------------------------- protobuf wire format
_DT = {1: np.float32, 6: np.int32, 7: np.int64, 10: np.float16, 11: np.float64}


def _varint(b: bytes, i: int):
    r, s = 0, 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        if c < 0x80:
            return r, i
        s += 7


def _fields(b: bytes):
    """(field number, wire type, value) over one message; length-delimited values come back as bytes."""
    i, n = 0, len(b)
    while i < n:
        key, i = _varint(b, i)
        f, w = key >> 3, key & 7
        if w == 0:
            v, i = _varint(b, i)
        elif w == 1:
            v, i = b[i:i + 8], i + 8
        elif w == 2:
            ln, i = _varint(b, i)
            v, i = b[i:i + ln], i + ln
        elif w == 5:
            v, i = b[i:i + 4], i + 4
        else:
            raise ValueError(f"unsupported protobuf wire type {w}")
        yield f, w, v


def _packed_varints(v) -> List[int]:
    if isinstance(v, int):
        return [v]
    out, i = [], 0
    while i < len(v):
        x, i = _varint(v, i)
        out.append(x)
    return out


def _signed(x: int) -> int:
    return x - (1 << 64) if x >= (1 << 63) else x


def _tensor(b: bytes):
    dims, dt, name, raw, fdata, idata = [], 1, "", None, [], []
    for f, w, v in _fields(b):
        if f == 1:
            dims += [_signed(x) for x in _packed_varints(v)]
        elif f == 2:
            dt = v
        elif f == 8:
            name = v.decode()
        elif f == 9:
            raw = v
        elif f == 4:
            fdata.append(np.frombuffer(v, dtype="<f4") if w == 2 else np.frombuffer(v, dtype="<f4", count=1))
        elif f == 7:
            idata += [_signed(x) for x in _packed_varints(v)]
    if dt not in _DT:
        raise NotImplementedError(f"ONNX tensor data type {dt} ({name})")
    if raw is not None:
        a = np.frombuffer(raw, dtype=np.dtype(_DT[dt]).newbyteorder("<"))
    elif fdata:
        a = np.concatenate(fdata)
    else:
        a = np.asarray(idata, dtype=_DT[dt])
    return name, a.astype(_DT[dt]).reshape(dims)


def _attribute(b: bytes):
    name, val, ints, floats = "", None, [], []
    for f, w, v in _fields(b):
        if f == 1:
            name = v.decode()
        elif f == 2:
            val = struct.unpack("<f", v)[0]
        elif f == 3:
            val = _signed(v)
        elif f == 4:
            val = v.decode(errors="replace")
        elif f == 5:
            val = _tensor(v)[1]
        elif f == 7:
            floats += list(np.frombuffer(v, dtype="<f4")) if w == 2 else [struct.unpack("<f", v)[0]]
        elif f == 8:
            ints += [_signed(x) for x in _packed_varints(v)]
    if ints:
        val = ints
    elif floats:
        val = floats
    return name, val


def read_onnx(path: str) -> dict:
    """ModelProto -> {"nodes": [{op, inputs, outputs, attrs}], "init": {name: ndarray}, "inputs": [...], "outputs": [...]}."""
    data = open(path, "rb").read()
    graph = None
    for f, w, v in _fields(data):
        if f == 7:
            graph = v
    if graph is None:
        raise ValueError(f"{path}: no graph in the ONNX model")
    nodes, init, inputs, outputs = [], {}, [], []
    for f, w, v in _fields(graph):
        if f == 1:
            n = {"op": "", "inputs": [], "outputs": [], "attrs": {}}
            for g, _, x in _fields(v):
                if g == 1:
                    n["inputs"].append(x.decode())
                elif g == 2:
                    n["outputs"].append(x.decode())
                elif g == 4:
                    n["op"] = x.decode()
                elif g == 5:
                    k, a = _attribute(x)
                    n["attrs"][k] = a
            nodes.append(n)
        elif f == 5:
            k, a = _tensor(v)
            init[k] = a
        elif f in (11, 12):
            name = next((x.decode() for g, _, x in _fields(v) if g == 1), "")
            (inputs if f == 11 else outputs).append(name)
    return {"nodes": nodes, "init": init, "inputs": [i for i in inputs if i not in init], "outputs": outputs}


# ---------------------------------------------------------------------------------------------- graph interpreter
class OnnxGraph:
    """Runs the operator subset of RL policy exports on torch tensors (weights resident on ``device``)."""

    def __init__(self, model: dict, device):
        import torch
        self.torch, self.device = torch, device
        self.nodes, self.inputs, self.outputs = model["nodes"], model["inputs"], model["outputs"]
        self.const = {k: torch.as_tensor(np.ascontiguousarray(v), device=device) for k, v in model["init"].items()}

    def run(self, feeds: Dict[str, "object"]) -> List["object"]:
        t = self.torch
        env = dict(self.const)
        env.update(feeds)
        for n in self.nodes:
            x = [env[i] if i else None for i in n["inputs"]]
            a, op = n["attrs"], n["op"]
            if op == "Gemm":
                A = x[0].transpose(-1, -2) if a.get("transA", 0) else x[0]
                B = x[1].transpose(-1, -2) if a.get("transB", 0) else x[1]
                al, be = a.get("alpha", 1.0), a.get("beta", 1.0)
                if len(x) > 2 and x[2] is not None and A.dim() == 2 and x[2].dim() <= 2:
                    y = t.addmm(x[2], A, B, beta=be, alpha=al)      # one library GEMM with the bias folded in
                else:
                    y = al * (A @ B)
                    if len(x) > 2 and x[2] is not None:
                        y = y + be * x[2]
            elif op == "MatMul":
                y = x[0] @ x[1]
            elif op in ("Add", "Sub", "Mul", "Div"):
                y = {"Add": t.add, "Sub": t.sub, "Mul": t.mul, "Div": t.div}[op](x[0], x[1])
            elif op == "Relu":
                y = t.relu(x[0])
            elif op == "Tanh":
                y = t.tanh(x[0])
            elif op == "Sigmoid":
                y = t.sigmoid(x[0])
            elif op == "Elu":
                y = t.nn.functional.elu(x[0], alpha=a.get("alpha", 1.0))
            elif op == "LeakyRelu":
                y = t.nn.functional.leaky_relu(x[0], negative_slope=a.get("alpha", 0.01))
            elif op == "Softplus":
                y = t.nn.functional.softplus(x[0])
            elif op == "Identity":
                y = x[0]
            elif op == "Clip":
                lo = x[1] if len(x) > 1 and x[1] is not None else a.get("min")
                hi = x[2] if len(x) > 2 and x[2] is not None else a.get("max")
                y = t.clamp(x[0], min=None if lo is None else float(lo), max=None if hi is None else float(hi))
            elif op == "Flatten":
                ax = a.get("axis", 1)
                y = x[0].reshape(int(np.prod(x[0].shape[:ax])) if ax else 1, -1)
            elif op == "Concat":
                y = t.cat(x, dim=a.get("axis", 0))
            elif op in ("Squeeze", "Unsqueeze"):
                axes = a.get("axes")
                if axes is None:
                    axes = [int(v) for v in x[1].tolist()] if len(x) > 1 and x[1] is not None else None
                y = x[0]
                if op == "Squeeze":
                    for ax in sorted([ax % y.dim() for ax in axes], reverse=True) if axes is not None else []:
                        y = y.squeeze(ax)
                    if axes is None:
                        y = y.squeeze()
                else:
                    for ax in sorted(axes):
                        y = y.unsqueeze(ax)
            elif op == "Reshape":
                shape = [int(v) for v in x[1].tolist()]
                shape = [x[0].shape[i] if s == 0 else s for i, s in enumerate(shape)]
                y = x[0].reshape(shape)
            elif op == "Transpose":
                y = x[0].permute(a["perm"]) if "perm" in a else x[0].permute(*reversed(range(x[0].dim())))
            elif op == "Constant":
                y = t.as_tensor(np.ascontiguousarray(a["value"]), device=self.device)
            elif op == "LSTM":
                outs = self._lstm(x, a)
                for name, val in zip(n["outputs"], outs):
                    if name:
                        env[name] = val
                continue
            else:
                raise NotImplementedError(f"ONNX operator '{op}' is not in the policy subset of cosim_amd.policy")
            env[n["outputs"][0]] = y
        return [env[o] for o in self.outputs]

    def _lstm(self, x, a):
        """ONNX LSTM, one direction, default activations: X [T,B,I], W [1,4H,I] (i o f c), R [1,4H,H], B [1,8H]."""
        t = self.torch
        if a.get("direction", "forward") != "forward" or a.get("layout", 0):
            raise NotImplementedError("LSTM: only direction=forward, layout=0")
        X, W, R = x[0], x[1][0], x[2][0]
        H = R.shape[1]
        Bv = x[3][0] if len(x) > 3 and x[3] is not None else t.zeros(8 * H, device=X.device)
        h = x[5][0] if len(x) > 5 and x[5] is not None else t.zeros((X.shape[1], H), device=X.device)
        c = x[6][0] if len(x) > 6 and x[6] is not None else t.zeros((X.shape[1], H), device=X.device)
        if X.is_cuda and X.shape[0] == 1 and X.dtype == t.float32:
            # one step for all envs: the engine's fused cell (gates on the matrix pipe + activations, one launch)
            out = self._lstm_cell_hip(X[0], h, c, W, R, Bv)
            if out is not None:
                hn, cn = out
                return hn.unsqueeze(0).unsqueeze(1), hn.unsqueeze(0), cn.unsqueeze(0)
        bias = Bv[:4 * H] + Bv[4 * H:]
        ys = []
        for s in range(X.shape[0]):
            g = X[s] @ W.T + h @ R.T + bias
            i, o, f, cc = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
            c = t.sigmoid(f) * c + t.sigmoid(i) * t.tanh(cc)
            h = t.sigmoid(o) * t.tanh(c)
            ys.append(h)
        return t.stack(ys).unsqueeze(1), h.unsqueeze(0), c.unsqueeze(0)


def _lstm_cell_hip(self, x, h, c, W, R, Bv):
    """``cosim_lstm_cell`` (csrc/cosim_mlp.hip): None when the library is not available (the interpreter's own ops then run)."""
    import ctypes
    t = self.torch
    if getattr(self, "_lstm_lib", None) is None:
        try:
            from .engine import load_library
            L = load_library()
            L.cosim_lstm_cell.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 6
            L.cosim_lstm_cell.restype = ctypes.c_int
            L.cosim_last_error.restype = ctypes.c_char_p
            self._lstm_lib = L
        except Exception:  # noqa: BLE001
            self._lstm_lib = False
    if not self._lstm_lib:
        return None
    x, h, c, W, R, Bv = (a.contiguous() for a in (x, h, c, W, R, Bv))
    hn, cn = t.empty_like(h), t.empty_like(c)
    rc = self._lstm_lib.cosim_lstm_cell(x.data_ptr(), h.data_ptr(), c.data_ptr(), x.shape[0], x.shape[1], h.shape[1], W.data_ptr(), R.data_ptr(),
                                        Bv.data_ptr(), hn.data_ptr(), cn.data_ptr(), t.cuda.current_stream(x.device).cuda_stream)
    if rc == -1:      # COSIM_EINVAL: a shape the fused cell does not take (in_dim + hidden > 1200): the interpreter's own ops run
        return None
    if rc != 0:
        raise RuntimeError(self._lstm_lib.cosim_last_error().decode())
    return hn, cn


OnnxGraph._lstm_cell_hip = _lstm_cell_hip


# ---------------------------------------------------------------------------------------------- the reference's classes, batched
_ACT = {"Relu": 1, "Tanh": 2, "Elu": 3, "Sigmoid": 4, "LeakyRelu": 5}


def _mlp_chain(model: dict):
    """The graph as a plain actor MLP -- Gemm(transB=1, alpha=beta=1) [+ activation] ... -- or None if it is anything else."""
    nodes, init = model["nodes"], model["init"]
    if len(model["inputs"]) != 1 or len(model["outputs"]) != 1:
        return None
    layers, cur, i = [], model["inputs"][0], 0
    while i < len(nodes):
        n = nodes[i]
        a = n["attrs"]
        if n["op"] != "Gemm" or n["inputs"][0] != cur or a.get("transB", 0) != 1 or a.get("transA", 0) or \
                a.get("alpha", 1.0) != 1.0 or a.get("beta", 1.0) != 1.0 or n["inputs"][1] not in init:
            return None
        w = init[n["inputs"][1]]
        b = init.get(n["inputs"][2]) if len(n["inputs"]) > 2 and n["inputs"][2] else None
        if w.ndim != 2 or w.dtype != np.float32 or (b is not None and (b.shape != (w.shape[0],) or b.dtype != np.float32)):
            return None
        cur, act, alpha = n["outputs"][0], 0, 1.0
        i += 1
        if i < len(nodes) and nodes[i]["op"] in _ACT and nodes[i]["inputs"] == [cur]:
            act = _ACT[nodes[i]["op"]]
            alpha = float(nodes[i]["attrs"].get("alpha", 1.0 if act == 3 else 0.01))
            cur = nodes[i]["outputs"][0]
            i += 1
        layers.append((w, b, act, alpha))
    if cur != model["outputs"][0] or not 1 <= len(layers) <= 6 or any(max(w.shape) > 512 for w, *_ in layers):
        return None
    return layers


class MLPPolicy:
    """``core/policy.py:5-21``: ``get_action(state[N, state_dim]) -> action[N, action_dim]`` in [-1, 1].

    A graph that is a plain Gemm / activation chain runs as ONE launch of the engine's fused MFMA kernel
    (``cosim_mlp_forward``, csrc/cosim_mlp.hip) when the policy lives on a GPU; anything else goes through the interpreter."""

    def __init__(self, policy_path: str, device=None, fused: Optional[bool] = None):
        import ctypes
        import torch
        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        model = read_onnx(policy_path)
        self.graph = OnnxGraph(model, self.device)
        self.input_name = self.graph.inputs[0]
        chain = _mlp_chain(model) if (fused is not False and self.device.type == "cuda") else None
        if fused and chain is None:
            raise ValueError("fused=True needs a Gemm/activation chain on a GPU device")
        self._fused = None
        if chain is not None:
            from .engine import load_library
            L = load_library()
            L.cosim_mlp_forward.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
            L.cosim_last_error.restype = ctypes.c_char_p
            ws = [torch.as_tensor(np.ascontiguousarray(w), device=self.device) for w, *_ in chain]
            bs = [None if b is None else torch.as_tensor(np.ascontiguousarray(b), device=self.device) for _, b, *_ in chain]
            nl = len(chain)
            self._fused = dict(
                L=L, nl=nl, keep=(ws, bs),
                dims=(ctypes.c_int * (nl + 1))(chain[0][0].shape[1], *[w.shape[0] for w, *_ in chain]),
                w=(ctypes.c_void_p * nl)(*[w.data_ptr() for w in ws]),
                b=(ctypes.c_void_p * nl)(*[None if b is None else b.data_ptr() for b in bs]),
                act=(ctypes.c_int * nl)(*[a for *_, a, _ in chain]),
                alpha=(ctypes.c_float * nl)(*[al for *_, al in chain]),
                out_dim=chain[-1][0].shape[0], in_dim=chain[0][0].shape[1], out=None)

    graph_safe = True     # stateless: get_action is pure device work

    def get_action(self, state):
        t = self.torch
        s = t.as_tensor(state, dtype=t.float32, device=self.device)
        single = s.dim() == 1
        x = s.unsqueeze(0) if single else s
        f = self._fused
        if f is not None and x.shape[1] == f["in_dim"]:
            x = x.contiguous()
            if f["out"] is None or f["out"].shape[0] != x.shape[0]:
                f["out"] = t.empty((x.shape[0], f["out_dim"]), dtype=t.float32, device=self.device)
            rc = f["L"].cosim_mlp_forward(x.data_ptr(), x.shape[0], f["nl"], f["dims"], f["w"], f["b"], f["act"], f["alpha"], 1.0,
                                          f["out"].data_ptr(), t.cuda.current_stream(self.device).cuda_stream)
            if rc != 0:
                raise RuntimeError(f["L"].cosim_last_error().decode())
            out = f["out"]
        else:
            out = self.graph.run({self.input_name: x})[0].clamp(-1.0, 1.0)
        return out.squeeze(0) if single else out


    def get_action_into(self, state_rows, out_rows):
        """``out_rows[:] = get_action(state_rows)`` on the CURRENT stream without allocating: for callers that evaluate the policy per
        env range on the range's own stream (``Runner.test_pipelined``).  Both are contiguous row slices of ``[N, ...]`` tensors."""
        f = self._fused
        if f is None or state_rows.shape[1] != f["in_dim"] or not (state_rows.is_contiguous() and out_rows.is_contiguous()):
            out_rows.copy_(self.get_action(state_rows))
            return
        rc = f["L"].cosim_mlp_forward(state_rows.data_ptr(), state_rows.shape[0], f["nl"], f["dims"], f["w"], f["b"], f["act"], f["alpha"], 1.0,
                                      out_rows.data_ptr(), self.torch.cuda.current_stream(self.device).cuda_stream)
        if rc != 0:
            raise RuntimeError(f["L"].cosim_last_error().decode())


class LSTMPolicy:
    """``core/policy.py:24-47``: inputs (state, "h_in", "c_in"), outputs (action, h_out, c_out); h/c kept per env."""

    def __init__(self, config: dict, policy_path: str, num_envs: int = 1, device=None):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.graph = OnnxGraph(read_onnx(policy_path), self.device)
        names = self.graph.inputs
        assert len(names) >= 3 and names[1] == "h_in" and names[2] == "c_in", \
            "The input names of ONNX policy must include 'h_in' and 'c_in'"
        self.input_name = names[0]
        self.h_in = torch.zeros((1, num_envs, config["policy"]["h_in_dim"]), dtype=torch.float32, device=self.device)
        self.c_in = torch.zeros((1, num_envs, config["policy"]["c_in_dim"]), dtype=torch.float32, device=self.device)

    graph_safe = True     # get_action is pure device work on persistent tensors (Runner.test_graphed)

    def reset(self, mask=None):
        """Zero the recurrent state (of the masked envs): what re-creating the policy does in the reference
        (core/tester.py builds a fresh policy per episode).  Branch-free on the device, so it can sit inside a captured graph."""
        if mask is None:
            self.h_in.zero_(); self.c_in.zero_()
        else:
            # masked_fill, not a multiply by a 0 / 1 mask: NaN * 0 is NaN, and a non-finite recurrent state must not survive the
            # reset of its env (it would emit NaN actions and be NaN-reset every step from then on)
            done = (self.torch.as_tensor(mask, device=self.device) != 0)[None, :, None]
            self.h_in.masked_fill_(done, 0.0)
            self.c_in.masked_fill_(done, 0.0)

    def get_action(self, state):
        t = self.torch
        s = t.as_tensor(state, dtype=t.float32, device=self.device)
        single = s.dim() == 1
        action, h_out, c_out = self.graph.run({self.input_name: s.unsqueeze(0) if single else s, "h_in": self.h_in, "c_in": self.c_in})[:3]
        # in place: the recurrent state lives in two persistent tensors, so a HIP-graph replay of this call feeds it back
        self.h_in.copy_(h_out.reshape(self.h_in.shape))
        self.c_in.copy_(c_out.reshape(self.c_in.shape))
        action = action.reshape(-1, action.shape[-1]).clamp(-1.0, 1.0)
        return action.squeeze(0) if single else action


def build_policy(config: dict, policy_path: str, num_envs: int = 1, device=None):
    """``core/policy.py:49-53``."""
    if config["policy"]["use_lstm"]:
        return LSTMPolicy(config, policy_path, num_envs=num_envs, device=device)
    return MLPPolicy(policy_path, device=device)


# ---------------------------------------------------------------------------------------------- writer (tests / synthetic policies)
def _enc_varint(x: int) -> bytes:
    x &= (1 << 64) - 1
    out = bytearray()
    while True:
        c = x & 0x7F
        x >>= 7
        out.append(c | (0x80 if x else 0))
        if not x:
            return bytes(out)


def _enc(field: int, wire: int, payload) -> bytes:
    key = _enc_varint((field << 3) | wire)
    if wire == 0:
        return key + _enc_varint(payload)
    if wire == 2:
        return key + _enc_varint(len(payload)) + payload
    return key + payload


def write_onnx(path: str, nodes: List[dict], init: Dict[str, np.ndarray], inputs: List[str], outputs: List[str]):
    """Minimal ONNX writer for the same subset (no policy ships with the reference: synthetic policies and the tests use it)."""
    def tensor(name, a):
        a = np.ascontiguousarray(a)
        dt = {np.dtype(np.float32): 1, np.dtype(np.int64): 7}[a.dtype]
        return b"".join(_enc(1, 0, int(d)) for d in a.shape) + _enc(2, 0, dt) + _enc(8, 2, name.encode()) + _enc(9, 2, a.tobytes())

    def attr(k, v):
        b = _enc(1, 2, k.encode())
        if isinstance(v, float):
            return b + _enc(2, 5, struct.pack("<f", v)) + _enc(20, 0, 1)
        if isinstance(v, int):
            return b + _enc(3, 0, v) + _enc(20, 0, 2)
        if isinstance(v, str):
            return b + _enc(4, 2, v.encode()) + _enc(20, 0, 3)
        return b + b"".join(_enc(8, 0, int(x)) for x in v) + _enc(20, 0, 7)

    g = b""
    for n in nodes:
        nb = b"".join(_enc(1, 2, i.encode()) for i in n["inputs"]) + b"".join(_enc(2, 2, o.encode()) for o in n["outputs"])
        nb += _enc(4, 2, n["op"].encode()) + b"".join(_enc(5, 2, attr(k, v)) for k, v in n.get("attrs", {}).items())
        g += _enc(1, 2, nb)
    g += _enc(2, 2, b"cosim_amd_policy")
    for k, a in init.items():
        g += _enc(5, 2, tensor(k, a))
    for name in inputs:
        g += _enc(11, 2, _enc(1, 2, name.encode()))
    for name in outputs:
        g += _enc(12, 2, _enc(1, 2, name.encode()))
    model = _enc(1, 0, 8) + _enc(7, 2, g) + _enc(8, 2, _enc(2, 0, 17))
    with open(path, "wb") as f:
        f.write(model)


def write_random_mlp(path: str, state_dim: int, action_dim: int, hidden=(256, 128), seed: int = 0, activation: str = "Elu"):
    """A random-weight actor with the usual export shape (Gemm + activation ... Gemm): stands in for the missing policy files."""
    rng = np.random.default_rng(seed)
    dims = [state_dim, *hidden, action_dim]
    nodes, init, x = [], {}, "obs"
    for li in range(len(dims) - 1):
        w = (rng.standard_normal((dims[li + 1], dims[li])) / np.sqrt(dims[li])).astype(np.float32)
        init[f"w{li}"], init[f"b{li}"] = w, np.zeros(dims[li + 1], dtype=np.float32)
        y = f"h{li}" if li < len(dims) - 2 else "actions"
        nodes.append({"op": "Gemm", "inputs": [x, f"w{li}", f"b{li}"], "outputs": [y + "_lin" if li < len(dims) - 2 else y], "attrs": {"transB": 1}})
        if li < len(dims) - 2:
            nodes.append({"op": activation, "inputs": [y + "_lin"], "outputs": [y]})
        x = y
    write_onnx(path, nodes, init, ["obs"], ["actions"])
