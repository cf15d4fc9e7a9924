"""GPU parity of the randomised / per-robot env layer: per-env masses and gains (A1), observation and info VALUES of every robot
(A6, A9), height maps (A8), init-noise joint sets (A11), sensor-noise levels (A7, config 4), terminal-step info.

The checker is the fp64 oracle with per-env parameters plus the numpy restatement of the robot-env layer in oracle/envlayer.py
(joint lists quoted from the reference's robot files, addresses read from the MJCF text -- independent of cosim_amd.compile's tables).
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ASSETS = os.path.join(os.path.dirname(__file__), "..", "cosim_amd", "assets")


def _xml(env_id):
    return os.path.join(ASSETS, env_id, f"{env_id}.xml")


def _per_env_oracle(cm, env, e):
    """Oracle instance carrying env e's randomised masses (-> invweight0, meaninertia) and PD gains."""
    from cosim_amd.model import set_field
    from oracle.oracle import Oracle
    o = Oracle(cm, body_mass=env.body_mass[e])
    set_field(o.model, "ctl_kp", env.kp[e])
    set_field(o.model, "ctl_kd", env.kd[e])
    return o


def test_randomised_masses_loads_and_gains_match_per_env_oracles():
    """A1 (xml_manager.py:43-55: mass += U(-m k, m k), base += load) + per-env Kp/Kd: every env of a randomised fleet against ITS OWN
    oracle (same masses, the mass-dependent constants body_invweight0 / dof_invweight0 / meaninertia recomputed, same gains), one
    control step from settled, contact-rich states.  The second half shows the test bites: the same fleet with the constants left
    at their nominal values misses the oracle by far more than the tolerance."""
    import torch
    from cosim_amd.batched_env import BatchedEnv
    from cosim_amd.compile import compile_model
    from cosim_amd.config import PARITY_RANDOM, make_config
    from cosim_amd.model import get_field
    n = 64
    rnd = dict(PARITY_RANDOM, mass_noise=0.05, load=1.0)
    cfg = make_config("flamingo_light_v1", random=rnd, num_envs=n, seed=77)
    cm = compile_model(cfg)
    env = BatchedEnv(cfg, num_envs=n, auto_reset=False, compiled=cm, seed=77, env_id0=1000, gain_noise=0.1)
    b = cm.blob
    m0 = np.array(get_field(b, "body_mass")[:b.nbody])
    base = cm.body_names.index("base_link")
    # the draws are what XMLManager's are: listed bodies within +-5 %, the base additionally + load, the others nominal
    rel = env.body_mass[:, 1:] / m0[None, 1:] - 1.0                # (body 0 is the massless world)
    listed = [cm.body_names.index(x) for x in ("base_link", "left_shoulder_link", "right_shoulder_link", "left_wheel_link", "right_wheel_link")]
    others = [i for i in range(1, b.nbody) if i not in listed]
    assert np.abs(env.body_mass[:, others] - m0[None, others]).max() == 0.0
    assert np.all(np.abs(env.body_mass[:, base] - 1.0 - m0[base]) <= 0.05 * m0[base] + 1e-12)
    lr = rel[:, [i - 1 for i in listed[1:]]]
    assert np.abs(lr).max() <= 0.05 + 1e-12 and lr.std() > 0.02
    kp0, kd0 = np.array(get_field(b, "ctl_kp")[:b.nu]), np.array(get_field(b, "ctl_kd")[:b.nu])
    assert np.abs(env.kp - kp0).max() <= 0.1 * kp0.max() + 1e-9 and (np.abs(env.kd / kd0 - 1).max(axis=0) > 0.05).all()

    q0 = np.array(get_field(b, "init_qpos")[:b.nq])
    rng = np.random.default_rng(5)
    R = dict(qpos=[], qvel=[], warm=[], act=[], qpos1=[], qvel1=[], tq=[], ncon=[])
    for e in range(n):
        o = _per_env_oracle(cm, env, e)
        o.reset(q0)
        phase = rng.uniform(0, 2 * np.pi, size=4)
        for t in range(40 + e % 7):                                  # settle on wheels and casters, then rock a little
            o.control_step(0.15 * np.sin(0.3 * t + phase))
        a = 0.25 * np.sin(phase)
        R["qpos"].append(o.qpos.copy()); R["qvel"].append(o.qvel.copy()); R["warm"].append(o.qacc_warmstart.copy()); R["act"].append(a)
        R["tq"].append(o.control_step(a))
        R["qpos1"].append(o.qpos.copy()); R["qvel1"].append(o.qvel.copy()); R["ncon"].append(o.ncon)
    R = {k: np.array(v) for k, v in R.items()}
    assert R["ncon"].min() >= 1 and np.median(R["ncon"]) >= 4         # every env stands on something: contact rows are live

    def replay(e_):
        e_.reset()
        e_.set_state(R["qpos"], R["qvel"], R["warm"])
        _, _, _, info = e_.step(torch.tensor(R["act"], dtype=torch.float32, device=e_.device))
        d = e_.get_data()
        return (d.qpos.cpu().numpy().astype(np.float64), d.qvel.cpu().numpy().astype(np.float64), info["torque"].cpu().numpy().astype(np.float64))

    qp, qv, tq = replay(env)
    np.testing.assert_allclose(tq, R["tq"], atol=2e-4)                # per-env gains reach the PD law
    ev = np.abs(qv - R["qvel1"]).max(axis=1)
    assert np.abs(qp - R["qpos1"]).max() < 2e-5 and ev.max() < 1e-3 and np.median(ev) < 1e-4, (np.abs(qp - R["qpos1"]).max(), ev.max(), np.median(ev))

    # sabotage 1: nominal invweights / meaninertia with the randomised masses (what "forgetting env_constants" would do)
    env.engine.set_param("body_invweight0", np.tile(np.array(get_field(b, "body_invweight0"))[:b.nbody, 0], (n, 1)))
    env.engine.set_param("dof_invweight0", np.tile(np.array(get_field(b, "dof_invweight0"))[:b.nv], (n, 1)))
    env.engine.set_param("meaninertia", np.full((n,), b.meaninertia))
    _, qv_bad, _ = replay(env)
    ev_bad = np.abs(qv_bad - R["qvel1"]).max(axis=1)
    assert np.median(ev_bad) > 5 * max(np.median(ev), 2e-5), (np.median(ev_bad), np.median(ev))
    # sabotage 2: nominal masses
    env.engine.set_param("body_mass", np.tile(m0, (n, 1)))
    _, qv_bad2, _ = replay(env)
    assert np.median(np.abs(qv_bad2 - R["qvel1"]).max(axis=1)) > 20 * max(np.median(ev), 2e-5)
    env.close()


@pytest.mark.parametrize("env_id,steps,amp", [("flamingo_light_v1", 60, 0.25), ("flamingo_p_v3", 40, 0.3), ("w4_p_v2", 60, 0.3),
                                              ("humanoid_p_v0", 40, 0.3)])
def test_observation_and_info_values_of_every_robot(env_id, steps, amp):
    """A6 / A9: the newest observation frame (incl. the gear-scaled leg entries of flamingo_p_v3 / w4_p_v2) and every info value
    (action_diff_RMSE, torque, lin_vel_x/y, ang_vel_yaw, set_points, state) after one control step from states recorded along an
    oracle trajectory, against the reference's _get_obs / _get_info restated over the oracle (envlayer.RobotEnvOracle)."""
    import torch
    from cosim_amd.batched_env import BatchedEnv
    from cosim_amd.compile import compile_model
    from cosim_amd.config import PARITY_RANDOM, make_config
    from cosim_amd.model import get_field
    from cosim_amd.robots import obs_to_dim
    from oracle.envlayer import RobotEnvOracle, WrapperOracle
    from oracle.oracle import Oracle
    cfg = make_config(env_id, random=PARITY_RANDOM)
    cm = compile_model(cfg)
    b = cm.blob
    ro = RobotEnvOracle(env_id, _xml(env_id), cfg["hardware"])
    dims = obs_to_dim(env_id, cfg)
    o = Oracle(cm)
    q0 = np.array(get_field(b, "init_qpos")[:b.nq])
    o.reset(q0)
    rng = np.random.default_rng(3)
    phase = rng.uniform(0, 2 * np.pi, size=b.nu)
    R = dict(qpos=[], qvel=[], warm=[], act=[], frame=[], info=[])
    cmd = np.array([0.5, 0.0, 0.1, 0.2, 0.0, 0.0])[:cfg["observation"]["command_dim"]]
    for t in range(steps):
        a = np.clip(amp * np.sin(0.25 * t + phase), -1, 1)
        R["qpos"].append(o.qpos.copy()); R["qvel"].append(o.qvel.copy()); R["warm"].append(o.qacc_warmstart.copy()); R["act"].append(a)
        tq = o.control_step(a)
        w = WrapperOracle(cfg, dims)                                  # fresh wrapper: reset frame, then this one step
        w.receive_user_command(cmd)
        w.reset(ro.obs(o, np.zeros(b.nu)))
        s, _, _ = w.step(ro.obs(o, a))
        R["frame"].append(s)
        R["info"].append(ro.info(o, a, np.zeros(b.nu), tq))
        if o.bad:
            break
    n = len(R["frame"])
    env = BatchedEnv(cfg, num_envs=n, auto_reset=False, compiled=cm)
    env.receive_user_command(cmd.astype(np.float32))
    env.reset()
    env.set_state(np.array(R["qpos"]), np.array(R["qvel"]), np.array(R["warm"]))
    state, term, trunc, info = env.step(torch.tensor(np.array(R["act"]), dtype=torch.float32, device=env.device))
    got = state.cpu().numpy().astype(np.float64)
    ref = np.array(R["frame"], dtype=np.float64)
    sd = sum(dims[k] for k in cfg["observation"]["stacked_obs_order"])
    S = int(cfg["observation"]["stack_size"])
    new = np.r_[0:sd, S * sd:got.shape[1]]                            # newest stacked frame + the non-stacked part
    # one control step of fp32 physics behind every value; scaled observation units (dof_vel x 0.15, ang_vel x 0.25)
    err = np.abs(got[:, new] - ref[:, new])
    assert np.median(err.max(axis=1)) < 2e-4 and np.quantile(err.max(axis=1), 0.95) < 5e-3, (np.median(err.max(axis=1)), err.max())
    # exact pieces: dof_pos / dof_vel entries are gathers (x gear) of the post-step state the engine itself reports
    d = env.get_data()
    qp, qv = d.qpos.cpu().numpy().astype(np.float64), d.qvel.cpu().numpy().astype(np.float64)
    sc = cfg["observation"]
    off = 0
    for name in sc["stacked_obs_order"]:
        if name == "dof_pos":
            exp = qp[:, ro.q_idx].copy(); exp[:, ro.r["geared"]] *= ro.gear
            np.testing.assert_allclose(got[:, off:off + dims[name]], exp * sc[name]["scale"], atol=2e-6)
        if name == "dof_vel":
            exp = qv[:, ro.qd_idx].copy(); exp[:, ro.r["geared"]] *= ro.gear
            np.testing.assert_allclose(got[:, off:off + dims[name]], exp * sc[name]["scale"], rtol=1e-6, atol=2e-6)
        off += dims[name]
    if ro.gear != 1.0:
        assert abs(ro.gear) > 1.2                                     # the geared columns really are scaled (gear_ratio -1.5)
    # info values
    I = {k: info[k].cpu().numpy().astype(np.float64) for k in ("action_diff_RMSE", "lin_vel_x", "lin_vel_y", "ang_vel_yaw", "torque", "set_points", "state")}
    ri = R["info"]
    np.testing.assert_allclose(I["action_diff_RMSE"], [x["action_diff_RMSE"] for x in ri], atol=1e-6)
    np.testing.assert_allclose(I["torque"], [x["torque"] for x in ri], atol=5e-4, rtol=1e-5)
    for k in ("lin_vel_x", "lin_vel_y", "ang_vel_yaw"):
        e_ = np.abs(I[k] - np.array([x[k] for x in ri]))
        assert np.median(e_) < 2e-4 and np.quantile(e_, 0.95) < 2e-2, (k, np.median(e_), e_.max())
    e_ = np.abs(I["state"] - np.array([x["state"] for x in ri])).max(axis=1)
    assert np.median(e_) < 1e-4 and np.quantile(e_, 0.95) < 2e-2, (np.median(e_), e_.max())
    assert I["state"].shape[1] == len(ro.r["info_state"]) and I["set_points"].shape[1] == b.nu
    if ro.action_scale is not None:
        np.testing.assert_allclose(I["set_points"], np.array(R["act"]) * ro.action_scale[None], atol=1e-5, rtol=1e-6)
    env.close()


@pytest.mark.parametrize("env_id,res,size,miss", [("flamingo_p_v3", (12, 12), (0.8, 0.8), 1.0), ("humanoid_p_v0", (15, 9), (1.0, 0.6), 5.0)])
def test_height_map_grid_and_miss_value(env_id, res, size, miss):
    """A8: the 12 x 12 map of flamingo_p_v3 (env_table.yaml:81-85) and the humanoid's 15 x 9 map against the oracle's vertical ray
    (mj_rayHfield restated), at the reset pose dropped over scattered places -- and at the rim of the field, where rays miss the
    terrain: robot_z + 1.0, but + 5.0 for the humanoid (envs/humanoid_p_v0/utils/mujoco_utils.py:141)."""
    from cosim_amd.batched_env import BatchedEnv
    from cosim_amd.compile import compile_model
    from cosim_amd.config import PARITY_RANDOM, make_config
    from cosim_amd.model import get_field
    from oracle.oracle import Oracle
    cfg = make_config(env_id, terrain="rocky_hard", random=PARITY_RANDOM, height_map=True)
    ob = cfg["observation"]["height_map"]
    assert (ob["res_x"], ob["res_y"]) == res and (ob["size_x"], ob["size_y"]) == size
    cm = compile_model(cfg)
    b = cm.blob
    o = Oracle(cm)
    q0 = np.array(get_field(b, "init_qpos")[:b.nq])
    half = b.hfield_size[0]
    rng = np.random.default_rng(2)
    spots = [(rng.uniform(-0.6, 0.6) * half, rng.uniform(-0.6, 0.6) * half, rng.uniform(-np.pi, np.pi)) for _ in range(10)]
    spots += [(half - 0.15, 3.0, 0.3), (-half + 0.1, -7.0, 2.0), (5.0, half - 0.05, -1.0)]          # windows that stick out of the field
    Q = []
    for x, y, yaw in spots:
        q = q0.copy()
        q[0:2] = (x, y)
        q[3:7] = [np.cos(yaw / 2), 0.02, -0.03, np.sin(yaw / 2)]
        q[3:7] /= np.linalg.norm(q[3:7])
        q[2] = q0[2] + 0.3
        Q.append(q)
    Q = np.array(Q)
    n = len(Q)
    env = BatchedEnv(cfg, num_envs=n, auto_reset=False, compiled=cm)
    env.reset()
    env.set_state(Q, np.zeros((n, b.nv)), np.zeros((n, b.nv)))
    import torch
    state, _, _, _ = env.step(torch.zeros((n, b.nu), device=env.device))
    qp = env.get_data().qpos.cpu().numpy().astype(np.float64)
    rx, ry = res
    got = state[:, -rx * ry:].cpu().numpy().astype(np.float64)
    nmiss = 0
    for e in range(n):
        q = qp[e]
        w, x, y, z = q[3:7]                                           # raw quaternion, as quat_to_rot_matrix uses it (no normalisation)
        Rm = np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w],
                       [2 * x * y + 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * x * w],
                       [2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y]])
        xs = np.linspace(-size[0] / 2, size[0] / 2, rx)
        ys = np.linspace(-size[1] / 2, size[1] / 2, ry)
        exp = np.zeros((ry, rx))
        for i in range(ry):
            for j in range(rx):
                P = q[0:3] + Rm @ np.array([xs[j], ys[i], 0.0])
                dist = o.ray_down(P[0], P[1], P[2] + 10.0)
                hit = dist >= 0
                nmiss += not hit
                exp[i, j] = q[2] - (P[2] + 10.0 - dist) if hit else q[2] + miss
        np.testing.assert_allclose(got[e], exp.ravel(), atol=3e-4, err_msg=f"env {e}")
    assert nmiss >= 20                                               # the rim cases really exercise the miss value
    env.close()


@pytest.mark.parametrize("env_id", ["flamingo_light_v1", "flamingo_p_v3", "w4_p_v2", "humanoid_p_v0"])
def test_reset_draws_init_noise_on_the_reference_joint_set(env_id):
    """A11: initial_qpos() -- light_v1 perturbs shoulders and wheels only (flamingo_light_v1.py:229-231), the other robots every
    hinge (qpos[7:]) -- with the engine's Philox stream (purpose 2, index = position in the joint list), heights per robot."""
    from cosim_amd import rng as crng
    from cosim_amd.batched_env import BatchedEnv
    from cosim_amd.compile import compile_model
    from cosim_amd.config import PARITY_RANDOM, make_config
    from oracle.envlayer import RobotEnvOracle
    n, seed, id0 = 96, 4321, 5000
    cfg = make_config(env_id, random=dict(PARITY_RANDOM, init_noise=0.05))
    cm = compile_model(cfg)
    ro = RobotEnvOracle(env_id, _xml(env_id), cfg["hardware"])
    env = BatchedEnv(cfg, num_envs=n, auto_reset=False, compiled=cm, seed=seed, env_id0=id0)
    env.reset()
    q = env.get_data().qpos.cpu().numpy().astype(np.float64)
    exp = np.zeros((n, cm.blob.nq))
    exp[:, 2] = ro.r["init_height"]
    exp[:, 3] = 1.0
    gids = np.arange(id0, id0 + n)
    for i, adr in enumerate(ro.noise_qadr):                            # draw i belongs to the i-th joint of the reference's list
        u = crng.uniform(seed, gids, 0, crng.PURPOSE_INIT, i).astype(np.float64)
        exp[:, adr] = 0.05 * (2.0 * u - 1.0)
    np.testing.assert_allclose(q, exp, atol=2e-7)
    noisy = np.zeros(cm.blob.nq, bool); noisy[ro.noise_qadr] = True
    assert (np.abs(q[:, noisy]).max(axis=0) > 0.03).all() and np.abs(q[:, noisy]).max() <= 0.05 + 1e-7
    assert np.abs(q[:, 7:][:, ~noisy[7:]]).max() == 0.0 if (~noisy[7:]).any() else True
    # a second reset of half the fleet draws fresh values (step counter moved on), the rest stays
    mask = np.zeros(n, np.uint8); mask[::2] = 1
    env.reset(mask)
    q2 = env.get_data().qpos.cpu().numpy().astype(np.float64)
    assert np.abs(q2[1::2] - q[1::2]).max() == 0.0 and (np.abs(q2[::2] - q[::2]).max(axis=1) > 0).all()
    env.close()


@pytest.mark.parametrize("level", ["low", "medium", "high", "ultra", "extreme"])
def test_p_v3_full_randomisation_sensor_noise_levels(level, golden_dir):
    """Config 4's workload (flamingo_p_v3, full random_table.yaml domain randomisation + sensor noise): every noise level of
    random_table.yaml:24-210 on the device, checked against the moments of the reference's own sampler
    (tests/golden/noise_moments.json, 1e6 draws of truncated_gaussian_noisy_data per level and field)."""
    import torch
    from cosim_amd.batched_env import BatchedEnv
    from cosim_amd.compile import compile_model
    from cosim_amd.config import make_config
    from cosim_amd.robots import obs_to_dim
    gold = json.load(open(os.path.join(golden_dir, "noise_moments.json")))
    n = 8192
    cfg = make_config("flamingo_p_v3", random=dict(sensor_noise=level, init_noise=0.0), num_envs=n, seed=9)   # the rest: GUI defaults
    assert cfg["random"]["mass_noise"] == 0.05 and cfg["random"]["action_delay_prob"] == 0.05
    cm = compile_model(cfg)
    env = BatchedEnv(cfg, num_envs=n, auto_reset=True, compiled=cm, seed=9, gain_noise=0.1)
    s, _ = env.reset()
    x = s.cpu().numpy().astype(np.float64)
    dims, ob = obs_to_dim("flamingo_p_v3", cfg), cfg["observation"]
    off = 0
    for name in ob["stacked_obs_order"]:
        d = dims[name]
        if name in ("dof_pos", "dof_vel", "ang_vel"):                 # clean value 0 at the reset pose (zero velocity, zero joints)
            g = gold[f"{level}/{name}"]
            v = x[:, off:off + d] / ob[name]["scale"]
            assert v.min() >= g["params"]["lower"] - 1e-6 and v.max() <= g["params"]["upper"] + 1e-6, name
            assert abs(v.std() - g["std"]) < 0.03 * g["std"] and abs(v.mean() - g["mean"]) < 0.03 * g["std"], (name, v.std(), g["std"])
            qs = np.quantile(v.ravel(), [0.05, 0.25, 0.5, 0.75, 0.95])
            np.testing.assert_allclose(qs, g["q"], atol=0.05 * g["std"])
            assert abs(np.corrcoef(v[:, 0], v[:, 1])[0, 1]) < 0.05     # i.i.d. per element
        if name == "projected_gravity":
            g = gold[f"{level}/{name}"]
            v = x[:, off + 2] / ob[name]["scale"] + 1.0
            assert abs(v.std() - g["std"]) < 0.03 * g["std"]
        off += d
    # the randomised fleet steps and stays finite under this level, and the noise is redrawn every step
    act = torch.zeros((n, 8), device=env.device)
    s1 = env.step(act)[0].clone()
    s2 = env.step(act)[0].clone()
    assert torch.isfinite(s2).all() and float((s1[:, 0] - s2[:, 0]).abs().mean()) > 0.1 * gold[f"{level}/dof_pos"]["std"]
    assert env.solver_stats()["dropped_contacts"] == 0
    env.close()


def test_info_of_the_terminal_step_describes_that_step():
    """With auto-reset the step that truncates an episode still reports that step's info (torque, set_points, velocities, state),
    as the reference's _get_info does, while the returned state vector already belongs to the next episode."""
    import torch
    from cosim_amd.batched_env import BatchedEnv
    from cosim_amd.compile import compile_model
    from cosim_amd.config import PARITY_RANDOM, make_config
    cfg = make_config("flamingo_light_v1", random=PARITY_RANDOM, max_duration=0.1)     # 5 control steps
    cm = compile_model(cfg)
    a = BatchedEnv(cfg, num_envs=8, auto_reset=True, compiled=cm)
    m = BatchedEnv(cfg, num_envs=8, auto_reset=False, compiled=cm)
    a.reset(); m.reset()
    act = torch.full((8, 4), 0.5, device=a.device)
    for t in range(5):
        sa, ta, ca, ia = a.step(act)
        sm, tm, cm_, im = m.step(act)
        for k in ("action_diff_RMSE", "torque", "set_points", "state", "lin_vel_x", "ang_vel_yaw"):
            assert torch.equal(ia[k], im[k]), (t, k)                  # identical to the env that is NOT reset in that step
    assert bool(ca.all()) and bool(cm_.all())
    assert float(ia["torque"].abs().max()) > 0 and float(ia["set_points"].abs().max()) > 0
    assert float(sa[:, 12:16].abs().max()) == 0.0 and float(sm[:, 12:16].abs().max()) == 0.5   # state: next episode vs this one
    assert a.solver_stats()["episodes_ended"] == 8
    a.close(); m.close()


def test_step_range_shards_on_streams_equal_the_single_launch():
    """cosim_step_range: one fleet stepped as 4 contiguous ranges on 4 HIP streams (control steps of different ranges overlap on the
    chip) gives the bits of the one-launch step, for the randomised GUI-default workload with auto-reset."""
    import torch
    from cosim_amd.batched_env import BatchedEnv
    from cosim_amd.compile import compile_model
    from cosim_amd.config import make_config
    n, S, T = 256, 4, 25
    cfg = make_config("flamingo_light_v1", max_duration=0.3, num_envs=n, seed=3)
    cm = compile_model(cfg)
    a = BatchedEnv(cfg, num_envs=n, auto_reset=True, seed=3, compiled=cm, gain_noise=0.1)
    b = BatchedEnv(cfg, num_envs=n, auto_reset=True, seed=3, compiled=cm, gain_noise=0.1)
    acts = torch.tensor(np.random.default_rng(0).uniform(-1, 1, size=(T, n, 4)), dtype=torch.float32, device=a.device)
    a.reset(); b.reset()
    streams = [torch.cuda.Stream(device=a.device) for _ in range(S)]
    torch.cuda.synchronize()
    ns = n // S
    for t in range(T):
        a.step(acts[t])
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                b.step_range(i * ns, ns, acts[t])
    torch.cuda.synchronize()
    assert torch.equal(a.state, b.state) and torch.equal(a.info_buf, b.info_buf) and torch.equal(a.truncated, b.truncated)
    assert torch.equal(a.get_data().qpos, b.get_data().qpos)
    with pytest.raises(ValueError):
        b.engine.step_range(n - 8, 16, acts[0].data_ptr(), b._cmd_ptr(), b.state.data_ptr(), b.terminated.data_ptr(), b.truncated.data_ptr(),
                            b.info_buf.data_ptr(), None)
    a.close(); b.close()


@pytest.mark.parametrize("I,H,n", [(52, 256, 4096), (88, 64, 333), (10, 6, 5)])
def test_lstm_cell_kernel_matches_the_fp64_cell(tmp_path, I, H, n):
    """N1: the recurrent policy (core/policy.py:24-47) with its LSTM node as ONE launch of cosim_lstm_cell (gates i, o, f, c on the
    matrix pipe, activations fused), checked against the fp64 numpy cell over several steps with the state carried per env, plus the
    masked state reset and the in-place state update a graph replay relies on."""
    import torch
    from cosim_amd.policy import LSTMPolicy, build_policy, write_onnx
    rng = np.random.default_rng(1)
    A = 4
    W = (0.4 * rng.standard_normal((1, 4 * H, I)) / np.sqrt(I / 8)).astype(np.float32)
    R = (0.4 * rng.standard_normal((1, 4 * H, H)) / np.sqrt(H / 8)).astype(np.float32)
    B = (0.1 * rng.standard_normal((1, 8 * H))).astype(np.float32)
    Wo = (0.5 * rng.standard_normal((A, H)) / np.sqrt(H / 8)).astype(np.float32)
    bo = np.zeros(A, dtype=np.float32)
    nodes = [{"op": "Unsqueeze", "inputs": ["obs"], "outputs": ["x3"], "attrs": {"axes": [0]}},
             {"op": "LSTM", "inputs": ["x3", "W", "R", "B", "", "h_in", "c_in"], "outputs": ["Y", "h_out", "c_out"], "attrs": {"hidden_size": H}},
             {"op": "Squeeze", "inputs": ["h_out"], "outputs": ["hs"], "attrs": {"axes": [0]}},
             {"op": "Gemm", "inputs": ["hs", "Wo", "bo"], "outputs": ["actions"], "attrs": {"transB": 1}}]
    p = str(tmp_path / "lstm.onnx")
    write_onnx(p, nodes, {"W": W, "R": R, "B": B, "Wo": Wo, "bo": bo}, ["obs", "h_in", "c_in"], ["actions", "h_out", "c_out"])
    pol = build_policy({"policy": {"use_lstm": True, "h_in_dim": H, "c_in_dim": H}}, p, num_envs=n, device="cuda:0")
    assert isinstance(pol, LSTMPolicy)
    hptr = pol.h_in.data_ptr()
    sig = lambda v: 1 / (1 + np.exp(-v))
    h = np.zeros((n, H)); c = np.zeros((n, H))
    W64, R64, b64 = W[0].astype(np.float64), R[0].astype(np.float64), (B[0, :4 * H] + B[0, 4 * H:]).astype(np.float64)
    for t in range(4):
        x = rng.standard_normal((n, I)).astype(np.float32)
        g = x.astype(np.float64) @ W64.T + h @ R64.T + b64
        i, o, f, cc = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
        c = sig(f) * c + sig(i) * np.tanh(cc)
        h = sig(o) * np.tanh(c)
        got = pol.get_action(torch.tensor(x, device="cuda:0")).cpu().numpy()
        np.testing.assert_allclose(got, np.clip(h @ Wo.T.astype(np.float64) + bo, -1, 1), atol=3e-5)
        np.testing.assert_allclose(pol.h_in[0].cpu().numpy(), h, atol=2e-5)
        np.testing.assert_allclose(pol.c_in[0].cpu().numpy(), c, atol=2e-5)
    assert pol.h_in.data_ptr() == hptr                                # state updated in place (graph replay feeds it back)
    assert pol.graph._lstm_lib not in (None, False)                   # the HIP cell ran, not the interpreter's op chain
    mask = np.zeros(n, np.uint8); mask[::3] = 1
    pol.reset(mask)
    hh = pol.h_in[0].cpu().numpy()
    assert np.abs(hh[::3]).max() == 0.0 and np.abs(hh[1::3]).max() > 0.0


def test_range_launches_inside_a_captured_graph_equal_the_eager_loop():
    """cosim_step with engine-owned ranges is capturable: the fork (range streams wait for the capturing stream) and the join
    (``env.join()`` before the capture ends) tie the range streams into the graph; inside it the range chains of consecutive steps
    stay independent (deferred join).  Three control steps per replay, several replays: the bits of the eager loop."""
    import torch
    from cosim_amd.batched_env import BatchedEnv
    from cosim_amd.config import make_config
    cfg = make_config("flamingo_light_v1", num_envs=256, seed=9)
    n, K, R = 256, 3, 4
    acts = (0.3 * torch.randn((K, n, 4), device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(2))).contiguous()
    outs = []
    for graphed in (False, True):
        env = BatchedEnv(cfg, num_envs=n, seed=9, auto_reset=True, ranges=4, deferred_join=True)
        env.receive_user_command(np.array([0.5, 0.0, 0.0, 0.0], dtype=np.float32))
        env.reset()
        torch.cuda.synchronize()

        def chunk():
            for k in range(K):
                env.step(acts[k])
            env.join()
        if graphed:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                chunk()                              # warm-up outside the capture (one chunk of the R)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                chunk()
            for _ in range(R - 1):
                g.replay()
        else:
            for _ in range(R):
                chunk()
        torch.cuda.synchronize()
        outs.append((env.state.clone(), env.get_data().qpos.clone(), env.solver_stats()["step_count"]))
        env.close()
    assert outs[0][2] == outs[1][2] == n * (1 + K * R)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("env_id,terrain,ranges", [("flamingo_light_v1", "flat", 1), ("flamingo_light_v1", "flat", 4),
                                                    ("flamingo_p_v3", "flat", 2)])
def test_rollout_rows_equal_the_step_loop(env_id, terrain, ranges):
    """cosim_rollout (K control steps in one launch per range) against K cosim_step calls on a twin fleet: row k of the rollout
    is what step k returned -- bit for bit for every env whose steps all stayed in the fleet kernel.  An env the dense fleet kernel
    abandons at step k finishes the rollout in the large-capacity kernel (contact-twist rows instead of dense rows from k+1 on,
    other rounding): those envs are held to a tolerance until the first episode end after k, and must be few."""
    import torch
    from cosim_amd.batched_env import BatchedEnv
    from cosim_amd.config import make_config
    n, K = 192, 120
    cfg = make_config(env_id, terrain=terrain, num_envs=n, seed=21, max_duration=1.6)      # 80-step episodes: auto-reset inside the table
    cmd = np.array([0.6, 0.0, 0.2, 0.0], dtype=np.float32)
    a = BatchedEnv(cfg, num_envs=n, seed=21, auto_reset=True, gain_noise=0.1)
    acts = (0.4 * torch.randn((K, n, a.action_dim), device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(5))).clamp_(-1, 1).contiguous()
    b = BatchedEnv(cfg, num_envs=n, seed=21, auto_reset=True, gain_noise=0.1, ranges=ranges)
    assert b.engine.query("rollout") == 1
    for e in (a, b):
        e.receive_user_command(cmd)
        e.reset()
    rows = []
    for k in range(K):
        s, te, tr, info = a.step(acts[k])
        rows.append((s.clone(), te.clone(), tr.clone(), a.info_buf.clone()))
    fix_a = a.solver_stats()["fixup_steps"]
    S, TE, TR, INF = b.rollout(acts)
    torch.cuda.synchronize()
    sb = b.solver_stats()
    assert sb["step_count"] == a.solver_stats()["step_count"] == n * (1 + K)
    ref_s = torch.stack([r[0] for r in rows]); ref_te = torch.stack([r[1] for r in rows]); ref_tr = torch.stack([r[2] for r in rows])
    ref_inf = torch.stack([r[3] for r in rows])
    assert (ref_tr | ref_te).sum().item() >= n                     # every env ended an episode (fall or time limit) inside the table
    same = ((S == ref_s).all(dim=2) & (INF == ref_inf).all(dim=2) & (TE == ref_te) & (TR == ref_tr)).all(dim=0)     # per env
    odd = (~same).nonzero().flatten().tolist()
    if fix_a == 0 and sb["fixup_steps"] == 0:
        assert not odd
    assert len(odd) <= max(2, n // 16), (len(odd), fix_a, sb["fixup_steps"])
    for i in odd:                                                  # same trajectory up to rounding until the paths part
        first = int((~((S[:, i] == ref_s[:, i]).all(dim=1))).nonzero()[0])
        assert torch.equal(S[:first, i], ref_s[:first, i])
        np.testing.assert_allclose(S[first, i].cpu().numpy(), ref_s[first, i].cpu().numpy(), rtol=0, atol=5e-3)
    # the env keeps stepping afterwards, and its tensors hold the last row
    assert torch.equal(b.state, S[-1]) and torch.equal(b.terminated, TE[-1])
    b.step(acts[0])
    torch.cuda.synchronize()
    assert torch.isfinite(b.state).all()
    a.close(); b.close()


def test_rollout_is_refused_where_no_rollout_kernel_exists():
    import torch
    from cosim_amd.batched_env import BatchedEnv
    from cosim_amd.config import make_config
    cfg = make_config("humanoid_p_v0", terrain="stairs_up_hard", num_envs=8, seed=1)
    env = BatchedEnv(cfg, num_envs=8, seed=1)
    env.reset()
    assert env.engine.query("rollout") == 0
    with pytest.raises(ValueError):
        env.rollout(torch.zeros((2, 8, env.action_dim), device="cuda:0"))
    env.close()


@pytest.mark.parametrize("env_id,terrain", [("w4_p_v2", "rocky_hard"), ("flamingo_p_v3", "flat"), ("w4_p_v2", "flat"),
                                            ("humanoid_p_v0", "stairs_up_hard"), ("flamingo_p_v3", "rocky_easy")])
def test_support_maps_leave_every_bit_of_the_fleet_as_the_hull_scans_do(env_id, terrain):
    """Mesh support queries through the hulls' support maps (csrc/cosim_hullmap.h) against full scans of the hulls (what the
    reference's mjc_support does): the same vertex every time, so the same contacts and the same fleet, bit for bit -- prism walk,
    cooperative hull walk, plane-hull routine and robot-robot pairs included."""
    import torch
    from cosim_amd.batched_env import BatchedEnv
    from cosim_amd.config import make_config
    n, K = 128, 60
    cfg = make_config(env_id, terrain=terrain, num_envs=n, seed=33)
    outs = []
    for use_map in (1.0, 0.0):
        env = BatchedEnv(cfg, num_envs=n, seed=33, auto_reset=True, gain_noise=0.1)
        env.engine.set_param("support_map", np.array([use_map], dtype=np.float32))
        # (hulls with few prisms under them take the cooperative walk when they have no map: the same walk in both runs, so that the
        # contacts come in the same order)
        env.engine.set_param("coop_walk", np.array([1.0], dtype=np.float32))
        acts = (0.5 * torch.randn((K, n, env.action_dim), device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(8))).clamp_(-1, 1)
        env.reset()
        states = []
        for k in range(K):
            s, te, tr, _ = env.step(acts[k])
            states.append(s.clone())
        d = env.get_data()
        st = env.solver_stats()
        outs.append((torch.stack(states), d.qpos.clone(), d.qvel.clone(), st["rows"], st["newton_iters"]))
        env.close()
    assert outs[0][3] == outs[1][3] and outs[0][4] == outs[1][4]
    assert outs[0][3] > 0
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


@pytest.mark.parametrize("robot", ["humanoid_p_v0", "w4_p_v2", "flamingo_p_v3"])
def test_device_support_routines_agree_map_scan_lane_parallel_and_cooperative(robot):
    """The device's own support routines (cosim_debug_support) on every mesh hull: through the support map and by scanning the
    hull, lane-parallel and wave-cooperative -- four ways, one vertex, the one a host-side scan with the same roundings finds.
    Directions include near-ties (within 1e-4 of a hull face normal's axis), where a different rounding sequence per loop used to
    break the tie differently."""
    from cosim_amd.batched_env import BatchedEnv
    from cosim_amd.config import make_config
    from cosim_amd.model import get_field
    env = BatchedEnv(make_config(robot, num_envs=2, seed=1), num_envs=2, seed=1)
    b = env.cm.blob
    gnum = np.array(get_field(b, "geom_hullnum")[:b.ngeom]); gadr = np.array(get_field(b, "geom_hulladr")[:b.ngeom])
    rng = np.random.default_rng(0)
    checked = 0
    for g in range(b.ngeom):
        if gnum[g] < 32:
            continue
        V = env.cm.hull_vert[gadr[g]:gadr[g] + gnum[g]].astype(np.float32)
        D = rng.normal(size=(12000, 3))
        D[:2000, :2] *= 1e-4
        D[2000:4000, 1:] *= 1e-4
        D[4000:6000, ::2] *= 1e-4
        D = np.ascontiguousarray(D / np.linalg.norm(D, axis=1)[:, None], dtype=np.float32)
        outs = []
        for use_map in (1, 0):
            o = np.zeros((len(D), 6), dtype=np.float32)
            env.engine._check(env.engine.L.cosim_debug_support(env.engine.h, g, D.ctypes.data, len(D), o.ctypes.data, use_map))
            outs.append(o)
        assert np.array_equal(outs[0], outs[1])
        assert np.array_equal(outs[0][:, :3], outs[0][:, 3:])
        # host twin of hull_dot: fma(l2, z, fma(l1, y, l0 * x)) in fp32, emulated in fp64 (exact products and sums, rounded to fp32 each step)
        t = (D[:, 0:1].astype(np.float64) * V[None, :, 0]).astype(np.float32)
        t = (D[:, 1:2].astype(np.float64) * V[None, :, 1] + t).astype(np.float32)
        t = (D[:, 2:3].astype(np.float64) * V[None, :, 2] + t).astype(np.float32)
        assert np.array_equal(outs[0][:, :3], V[t.argmax(1)])
        checked += 1
    assert checked >= 2
    env.close()


def test_block_tests_of_the_prism_walk_remove_only_prisms_that_cannot_touch():
    """The narrowphase kernel (humanoid on 1 cm stairs) tests blocks of 8 prisms before their prisms.  A block is dropped only if none
    of its prisms can touch the geom, so the contacts -- and with them every bit of the fleet -- are those of the walk that sends every
    block on to the per-prism tests."""
    import torch
    from cosim_amd.batched_env import BatchedEnv
    from cosim_amd.config import make_config
    n, K = 192, 80
    cfg = make_config("humanoid_p_v0", terrain="stairs_up_hard", num_envs=n, seed=12)
    outs = []
    for cull in (1.0, 0.0):
        env = BatchedEnv(cfg, num_envs=n, seed=12, auto_reset=True, gain_noise=0.1)
        assert env.engine.query("split") > 0
        env.engine.set_param("block_cull", np.array([cull], dtype=np.float32))
        acts = (0.6 * torch.randn((K, n, env.action_dim), device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(3))).clamp_(-1, 1)
        env.reset()
        states = []
        for k in range(K):
            s, _, _, _ = env.step(acts[k])
            states.append(s.clone())
        d = env.get_data()
        st = env.solver_stats()
        outs.append((torch.stack(states), d.qpos.clone(), st["rows"], st["max_contacts"]))
        env.close()
    assert outs[0][2] == outs[1][2] and outs[0][3] == outs[1][3] and outs[0][3] > 60      # fallen humanoids on the steps: many contacts
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_hulls_on_coarse_terrain_staged_walk_finds_the_contacts_of_the_cooperative_walk():
    """w4_p_v2 on rocky_hard (55 cm cells): a wheel hull has a handful of prisms under it.  With support maps those prisms go through
    the staged lane-parallel walk with everybody else's (contacts in geom order, the reference's order); round 2 walked each such
    hull wave-cooperatively after the others (same contacts, appended last).  One control step from the same states: the same number of
    constraint rows, and states that differ by the rounding of a different row order only."""
    import torch
    from cosim_amd.batched_env import BatchedEnv
    from cosim_amd.config import make_config
    n = 256
    cfg = make_config("w4_p_v2", terrain="rocky_hard", num_envs=n, seed=5, height_map=True)
    src = BatchedEnv(cfg, num_envs=n, seed=5, auto_reset=True, gain_noise=0.1)
    acts = (0.5 * torch.randn((41, n, src.action_dim), device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(1))).clamp_(-1, 1)
    src.reset()
    for k in range(40):
        src.step(acts[k])
    d = src.get_data()
    q, v = d.qpos.clone(), d.qvel.clone()
    w = torch.zeros_like(v)                  # (cold start of the solver in both runs)
    src.close()
    outs = []
    for coop in (0.0, 1.0):
        env = BatchedEnv(cfg, num_envs=n, seed=5, auto_reset=False, gain_noise=0.1)
        env.engine.set_param("coop_walk", np.array([coop], dtype=np.float32))
        env.reset()
        env.set_state(q, v, w)
        r0 = env.solver_stats()["rows"]
        env.step(acts[40])
        dd = env.get_data()
        st = env.solver_stats()
        outs.append((dd.qpos.clone(), dd.qvel.clone(), st["rows"] - r0, st["max_contacts"], st["dropped_contacts"]))
        env.close()
    assert outs[0][2] == outs[1][2] and outs[0][2] > 4 * 8 * n          # same rows over the step's four substeps; wheels on the ground
    assert outs[0][3] == outs[1][3] and outs[0][4] == outs[1][4] == 0
    dq = (outs[0][0] - outs[1][0]).abs().max(dim=1).values.cpu().numpy()
    print("per-env max |dqpos| quantiles 50/90/99/100 %:", np.quantile(dq, [0.5, 0.9, 0.99, 1.0]))
    # a wheel hull resting on several prisms is a redundant contact set: where the solve is ill-conditioned (tests/test_gpu_parity.py,
    # the config-3 trajectory test, has the probes) another row order moves the iterate it stops at; everywhere else it is round-off
    assert np.quantile(dq, 0.9) < 2e-5 and dq.max() < 0.05, np.quantile(dq, [0.5, 0.9, 0.99, 1.0])
