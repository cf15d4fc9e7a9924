"""The CPU restatements (oracle/envlayer.py, cosim_amd host mirrors) against the golden vectors that
tools/make_golden.py captured from the reference's own importable modules."""
import json
import os

import numpy as np
import pytest

from cosim_amd.config import make_config
from cosim_amd.robots import ROBOTS, obs_to_dim
from cosim_amd.xml_manager import XMLManager
from oracle import envlayer


@pytest.fixture(scope="module")
def wrappers(golden_dir):
    return np.load(os.path.join(golden_dir, "wrappers.npz")), json.load(open(os.path.join(golden_dir, "wrappers_meta.json")))


def _variant_config(meta):
    mod, env_id = meta["mod"], meta["env_id"]
    cfg = make_config(env_id, max_duration=mod.get("max_duration", 120.0), position_command=meta["position_command"])
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
    return cfg


VARIANTS = ["flamingo_light_v1_default", "flamingo_p_v3_default", "w4_p_v2_default", "humanoid_p_v0_default",
            "flamingo_light_v1_freq", "flamingo_light_v1_stack1", "flamingo_light_v1_stack5",
            "flamingo_light_v1_cmdstacked", "flamingo_light_v1_poscmd", "flamingo_light_v1_short"]


@pytest.mark.parametrize("name", VARIANTS)
def test_wrapper_stack_matches_reference(wrappers, name):
    data, meta = wrappers
    m = meta[name]
    cfg = _variant_config(m)
    dims = obs_to_dim(m["env_id"], cfg)
    w = envlayer.WrapperOracle(cfg, dims)
    assert w.state_dim == m["state_dim"]
    assert [[s.start, s.stop] for s in w.cmd_slices] == m["cmd_slices"]
    assert w.max_sim_step == m["max_sim_step"]
    states, flags = data[f"{name}/states"], data[f"{name}/flags"]
    cmds, qpos = data[f"{name}/cmds"], data[f"{name}/qpos"]
    obs_keys = [k.split("/")[-1] for k in data.files if k.startswith(f"{name}/obs/")]

    def obs_at(t):
        return {k: data[f"{name}/obs/{k}"][t] for k in obs_keys}

    cd = cfg["observation"]["command_dim"]
    w.receive_user_command(cmds[0][:cd], qpos[0])
    s = w.reset(obs_at(0))
    np.testing.assert_array_equal(s, states[0])          # bit-exact: fp32 cast, scale, stack, command overwrite
    np.testing.assert_allclose(w.applied_command, data[f"{name}/applied"][0], rtol=0, atol=1e-15)
    for t in range(1, len(states)):
        # the reference reads get_data().qpos of the wrapped env at its current time index (t-1 before the step)
        w.receive_user_command(cmds[t][:cd], qpos[t - 1])
        s, term, trunc = w.step(obs_at(t))
        np.testing.assert_array_equal(s, states[t])
        assert (int(term), int(trunc)) == tuple(int(x) for x in flags[t - 1])
        np.testing.assert_allclose(w.applied_command, data[f"{name}/applied"][t], rtol=0, atol=1e-12)
    if name.endswith("_short"):
        assert flags[-1][1] == 1 and len(flags) == 10     # truncated exactly at int(0.2 * 50) steps


def test_reporter_info_contract(wrappers):
    _, meta = wrappers
    keys = set(meta["flamingo_light_v1_default"]["info_keys"])
    assert {f"user_command_{i}" for i in range(4)} <= keys and "dt" in keys


def test_delay_filter_and_pd(golden_dir):
    g = np.load(os.path.join(golden_dir, "control.npz"))
    for env_id in ROBOTS:
        for prob in (0.0, 0.05, 0.5, 1.0):
            k = f"{env_id}/delay_p{prob}"
            out = envlayer.delay_filter_sequence(g[f"{k}/in"], g[f"{k}/u"], prob)
            np.testing.assert_array_equal(out, g[f"{k}/out"])
        a = g[f"{env_id}/pd/args"]
        np.testing.assert_allclose(envlayer.pd_controller(*a.T), g[f"{env_id}/pd/out"], rtol=1e-15, atol=0)


def test_projected_gravity_and_rotation(golden_dir):
    g = np.load(os.path.join(golden_dir, "math.npz"))
    q = g["quat"]
    for i in range(len(q)):
        x, y, z, w = q[i]     # the golden call passed the 4 numbers as xyzw
        np.testing.assert_allclose(envlayer.projected_gravity([w, x, y, z]), g["projected_gravity_xyzw"][i], atol=1e-14)
        np.testing.assert_allclose(envlayer.rot_matrix_wxyz(q[i]), g["rotmat_wxyz_raw"][i], atol=1e-14)


def test_xml_manager_mirror_matches_reference(golden_dir):
    gold = json.load(open(os.path.join(golden_dir, "xml.json")))
    flat_rnd = dict(mass_noise=0.05, load=1.0, sliding_friction=0.6, torsional_friction=0.03, rolling_friction=0.02,
                    friction_loss=0.2, precision="high")
    for env_id in ROBOTS:
        for terrain, rnd in (("flat", flat_rnd), ("rocky_hard", dict(mass_noise=0.0, load=0.0))):
            rec = gold[f"{env_id}/{terrain}"]
            cfg = make_config(env_id, terrain=terrain, random=rnd)
            xm = XMLManager(cfg)
            root = xm.get_model_tree()
            np.random.seed(0)
            masses = xm.draw_masses(root)          # legacy np.random stream, document order: same draws as the reference
            nominal = xm.nominal_masses(root)
            for body, m_ref in rec["masses"].items():
                m = masses[body][0] if body in masses else nominal[body]
                assert m == pytest.approx(m_ref, rel=1e-15, abs=0), (env_id, terrain, body)
            ground = [g for g in root.findall(".//geom") if g.attrib.get("name") == "ground"][0]
            for k in ("type", "size", "hfield", "friction"):
                assert ground.attrib.get(k) == rec["ground"][k], (env_id, terrain, k)
            opt = root.find("option")
            assert opt.attrib["timestep"] == rec["option"]["timestep"] and opt.attrib["iterations"] == rec["option"]["iterations"]
            for body in root.findall(".//body"):
                for g in body.findall("geom"):
                    if "friction" in g.attrib and "name" in g.attrib:
                        assert g.attrib["friction"] == rec["geom_friction"][g.attrib["name"]]
            for d in root.findall(".//default"):
                for j in d.findall("joint"):
                    if "frictionloss" in j.attrib:
                        assert j.attrib["frictionloss"] == rec["default_frictionloss"][d.attrib.get("class", "main")]
    assert gold["flamingo_light_v1/flat"]["masses"]["base_link"] == pytest.approx(3.798051924914548, rel=1e-15)


def test_truncated_gaussian_moments_match_scipy_reference(golden_dir):
    """Distribution-level pin for A7: inverse-CDF sampling (what the HIP kernel does) against the reference's draws."""
    from scipy.stats import norm
    gold = json.load(open(os.path.join(golden_dir, "noise_moments.json")))
    rng = np.random.default_rng(0)
    for key in ("low/dof_pos", "low/dof_vel", "medium/ang_vel", "high/projected_gravity", "extreme/lin_vel"):
        p = gold[key]["params"]
        a, b = (p["lower"] - p["mean"]) / p["std"], (p["upper"] - p["mean"]) / p["std"]
        u = rng.uniform(size=200000)
        x = p["mean"] + p["std"] * norm.ppf(norm.cdf(a) + u * (norm.cdf(b) - norm.cdf(a)))
        assert x.min() >= p["lower"] - 1e-12 and x.max() <= p["upper"] + 1e-12
        assert abs(x.std() - gold[key]["std"]) < 0.02 * gold[key]["std"]
        assert abs(x.mean() - gold[key]["mean"]) < 0.02 * gold[key]["std"]
        np.testing.assert_allclose(np.quantile(x, [0.05, 0.25, 0.5, 0.75, 0.95]), gold[key]["q"], atol=0.03 * gold[key]["std"])
