"""Batched policy module: ONNX reader + interpreter against numpy evaluations of the same graphs.
PARITY UNPINNED against onnxruntime (absent here; no policy file ships with the reference)."""
import numpy as np
import pytest

from cosim_amd.policy import LSTMPolicy, MLPPolicy, build_policy, read_onnx, write_onnx, write_random_mlp


def test_mlp_policy_matches_numpy_and_clips(tmp_path):
    p = str(tmp_path / "actor.onnx")
    write_random_mlp(p, state_dim=52, action_dim=4, hidden=(64, 32), seed=3, activation="Elu")
    m = read_onnx(p)
    assert m["inputs"] == ["obs"] and m["outputs"] == ["actions"] and [n["op"] for n in m["nodes"]] == ["Gemm", "Elu", "Gemm", "Elu", "Gemm"]
    assert m["nodes"][0]["attrs"]["transB"] == 1 and m["init"]["w0"].shape == (64, 52)
    pol = MLPPolicy(p, device="cpu")
    x = (3.0 * np.random.default_rng(0).standard_normal((7, 52))).astype(np.float32)
    h = x
    for li in range(3):
        h = h @ m["init"][f"w{li}"].T + m["init"][f"b{li}"]
        if li < 2:
            h = np.where(h > 0, h, np.exp(np.minimum(h, 0)) - 1)
    got = pol.get_action(x).numpy()
    np.testing.assert_allclose(got, np.clip(h, -1, 1), atol=2e-5)
    assert np.abs(h).max() > 1.0 and np.abs(got).max() <= 1.0          # the clip of core/policy.py:20 is exercised
    single = pol.get_action(x[2])                                        # single-state call keeps the reference's shape
    assert single.shape == (4,) and np.allclose(single.numpy(), got[2], atol=1e-6)


def test_lstm_policy_carries_state_per_env(tmp_path):
    rng = np.random.default_rng(1)
    I, H, A = 10, 6, 3
    W = (0.4 * rng.standard_normal((1, 4 * H, I))).astype(np.float32)
    R = (0.4 * rng.standard_normal((1, 4 * H, H))).astype(np.float32)
    B = (0.1 * rng.standard_normal((1, 8 * H))).astype(np.float32)
    Wo = (0.5 * rng.standard_normal((A, H))).astype(np.float32)
    bo = np.zeros(A, dtype=np.float32)
    nodes = [{"op": "Unsqueeze", "inputs": ["obs"], "outputs": ["x3"], "attrs": {"axes": [0]}},
             {"op": "LSTM", "inputs": ["x3", "W", "R", "B", "", "h_in", "c_in"], "outputs": ["Y", "h_out", "c_out"], "attrs": {"hidden_size": H}},
             {"op": "Squeeze", "inputs": ["h_out"], "outputs": ["hs"], "attrs": {"axes": [0]}},
             {"op": "Gemm", "inputs": ["hs", "Wo", "bo"], "outputs": ["actions"], "attrs": {"transB": 1}}]
    p = str(tmp_path / "lstm.onnx")
    write_onnx(p, nodes, {"W": W, "R": R, "B": B, "Wo": Wo, "bo": bo}, ["obs", "h_in", "c_in"], ["actions", "h_out", "c_out"])
    cfg = {"policy": {"use_lstm": True, "h_in_dim": H, "c_in_dim": H}}
    pol = build_policy(cfg, p, num_envs=5, device="cpu")
    assert isinstance(pol, LSTMPolicy)
    sig = lambda v: 1 / (1 + np.exp(-v))
    h = np.zeros((5, H)); c = np.zeros((5, H))
    for t in range(4):
        x = rng.standard_normal((5, I)).astype(np.float32)
        g = x @ W[0].T + h @ R[0].T + B[0, :4 * H] + B[0, 4 * H:]
        i, o, f, cc = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
        c = sig(f) * c + sig(i) * np.tanh(cc)
        h = sig(o) * np.tanh(c)
        np.testing.assert_allclose(pol.get_action(x).numpy(), np.clip(h @ Wo.T + bo, -1, 1), atol=2e-5)
    pol.reset(mask=np.array([1, 0, 0, 0, 1]))
    assert float(pol.h_in[0, 0].abs().max()) == 0.0 and float(pol.h_in[0, 1].abs().max()) > 0.0
    with pytest.raises(AssertionError, match="h_in"):                     # core/policy.py:28-29
        write_onnx(p, nodes, {"W": W, "R": R, "B": B, "Wo": Wo, "bo": bo}, ["obs", "hidden", "cell"], ["actions", "h_out", "c_out"])
        LSTMPolicy(cfg, p, num_envs=1, device="cpu")


def test_unknown_operator_fails_loudly(tmp_path):
    p = str(tmp_path / "bad.onnx")
    write_onnx(p, [{"op": "Einsum", "inputs": ["obs"], "outputs": ["actions"]}], {}, ["obs"], ["actions"])
    with pytest.raises(NotImplementedError, match="Einsum"):
        MLPPolicy(p, device="cpu").get_action(np.zeros((1, 3), dtype=np.float32))
