"""Host-side logic that needs no GPU: config builder, model compiler, RNG, ABI surface, error behaviour."""
import ctypes
import json
import os

import numpy as np
import pytest

from cosim_amd import rng as crng
from cosim_amd.batched_env import _cmd_slices
from cosim_amd.compile import compile_model, env_constants
from cosim_amd.config import PARITY_RANDOM, make_config
from cosim_amd.engine import EXPORTS, ObsConfig, load_library, make_obs_config
from cosim_amd.model import CosimModel, get_field
from cosim_amd.robots import ROBOTS, obs_to_dim

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_philox_known_answer_vectors():
    # Random123 known-answer tests for philox4x32-10
    z = np.uint32(0)
    out = crng.philox4x32(z, z, z, z, z, z)
    assert [int(x) for x in out] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = np.uint32(0xFFFFFFFF)
    out = crng.philox4x32(f, f, f, f, f, f)
    assert [int(x) for x in out] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    u = crng.uniform(7, np.arange(1000), 3, crng.PURPOSE_DELAY, 0)
    assert u.min() > 0 and u.max() < 1 and abs(u.mean() - 0.5) < 0.05


def test_fleets_of_two_seeds_are_not_permutations_of_each_other():
    # seed and env id sit in separate Philox words: (seed 0, env 1) and (seed 1, env 0) must not share a stream, and the
    # fleet drawn under seed 1235 must not be the fleet of seed 1234 with its envs shuffled
    assert crng.uniform(0, np.array([1]), 0, 0, 0)[0] != crng.uniform(1, np.array([0]), 0, 0, 0)[0]
    g = np.arange(4096)
    a = np.sort(crng.uniform(1234, g, 5, crng.PURPOSE_MASS, 2))
    b = np.sort(crng.uniform(1235, g, 5, crng.PURPOSE_MASS, 2))
    assert (a == b).mean() < 0.01
    # distinct (purpose, index) pairs are distinct streams as well
    assert crng.uniform(3, g, 1, 1, 2)[7] != crng.uniform(3, g, 1, 2, 1)[7]


def test_config_matches_gui_defaults():
    cfg = make_config("flamingo_light_v1")
    assert cfg["random"] == dict(precision="medium", sensor_noise="low", init_noise=0.05, sliding_friction=0.8,
                                 torsional_friction=0.02, rolling_friction=0.01, friction_loss=0.1, action_delay_prob=0.05,
                                 mass_noise=0.05, load=0.0)
    ob = cfg["observation"]
    assert ob["stacked_obs_order"] == ["dof_pos", "dof_vel", "ang_vel", "projected_gravity", "last_action"]
    assert ob["dof_vel"] == {"freq": 50, "scale": 0.15} and ob["last_action"] == {"freq": 50, "scale": 1.0}
    assert ob["height_map"] is None and ob["lin_vel_x"] is None
    assert ob["command_scales"] == {"0": 2.0, "1": 1.0, "2": 0.25, "3": 1.0}
    assert cfg["hardware"]["Kp_shoulder"] == 15.0 and cfg["hardware"]["action_scales"]["wheel"] == 40.0
    with pytest.raises(NameError):
        make_config("no_such_robot")


def test_state_dims_and_cmd_slices_match_reference(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, "wrappers_meta.json")))
    for env_id in ROBOTS:
        cfg = make_config(env_id)
        dims = obs_to_dim(env_id, cfg)
        ob = cfg["observation"]
        sd = sum(dims[n] for n in ob["stacked_obs_order"]) * ob["stack_size"] + sum(dims[n] for n in ob["non_stacked_obs_order"])
        assert sd == meta[f"{env_id}_default"]["state_dim"]
        sl = _cmd_slices(ob["stacked_obs_order"], ob["non_stacked_obs_order"], dims, ob["stack_size"], ob["command_dim"])
        assert [[s.start, s.stop] for s in sl] == meta[f"{env_id}_default"]["cmd_slices"]


def test_compiled_model_facts_light_v1():
    cm = compile_model(make_config("flamingo_light_v1", random=PARITY_RANDOM))
    b = cm.blob
    assert (b.nq, b.nv, b.nu, b.nbody, b.neq, b.npair) == (19, 18, 4, 14, 2, 0)
    assert np.array(get_field(b, "body_mass")[:14]).sum() == pytest.approx(4.94905, abs=1e-5)
    assert list(get_field(b, "ctl_qadr")[:4]) == [7, 10, 9, 12]                        # sh 7,10; wheel 9,12
    assert b.timestep == 0.005 and b.frame_skip == 4 and b.iterations == 50
    assert np.allclose(get_field(b, "ground_friction"), [0.8, 0.02, 0.01])
    # connect anchors coincide at qpos0
    from cosim_amd.compile import forward_kinematics
    fk = forward_kinematics(cm.const["m"], np.array(get_field(b, "qpos0")[:19]))
    for e in range(2):
        b1, b2 = get_field(b, "eq_body1")[e], get_field(b, "eq_body2")[e]
        p1 = fk["xpos"][b1] + fk["xmat"][b1] @ get_field(b, "eq_anchor1")[e]
        p2 = fk["xpos"][b2] + fk["xmat"][b2] @ get_field(b, "eq_anchor2")[e]
        np.testing.assert_allclose(p1, p2, atol=1e-12)
    # frictionloss default classes joints / wheels were rewritten to friction_loss, casters stay at 0
    fl = np.array(get_field(b, "dof_frictionloss")[:18])
    assert np.allclose(fl[6:14], 0.1) and np.allclose(fl[14:], 0.0) and np.allclose(fl[:6], 0.0)


def _hinge_ranges(cm):
    b = cm.blob
    return {cm.joint_names[j]: tuple(get_field(b, "jnt_range")[j]) for j in range(b.njnt) if get_field(b, "jnt_limited")[j]}


def test_compiled_model_facts_flamingo_p_v3():
    """SURVEY App. A, column flamingo_p_v3 (facts read off envs/flamingo_p_v3/assets/xml/flamingo_p_v3.xml by hand)."""
    cm = compile_model(make_config("flamingo_p_v3", random=PARITY_RANDOM))
    b = cm.blob
    assert (b.nq, b.nv, b.nu, b.nbody - 1, b.neq) == (15, 14, 8, 9, 0)                 # 9 bodies + world; free + 8 hinges
    assert np.array(get_field(b, "body_mass")[:b.nbody]).sum() == pytest.approx(16.51937, abs=1e-5)
    # action order = XML actuator order: L/R hip, L/R shoulder, L/R leg, L/R wheel -> qpos 7, 11, 8, 12, 9, 13, 10, 14
    assert list(get_field(b, "ctl_qadr")[:8]) == [7, 11, 8, 12, 9, 13, 10, 14]
    assert list(get_field(b, "ctl_gear")[:8]) == [1, 1, 1, 1, -1.5, -1.5, 1, 1]          # legs geared -1.5 (flamingo_p_v3.py:161)
    assert list(get_field(b, "ctl_velmode")[:8]) == [0, 0, 0, 0, 0, 0, 1, 1]             # wheels are velocity-PD
    assert np.allclose(get_field(b, "gravity"), [0, 0, -9.807]) and b.timestep == 0.005
    assert np.array(get_field(b, "init_qpos")[:3]).tolist() == [0.0, 0.0, 0.61282]       # initial_qpos (:249-255)
    assert b.term_mode == 1 and b.nterm_body == 5 and b.init_noise_nq == 8               # cfrc_ext rule; noise on every hinge
    rng = _hinge_ranges(cm)
    assert rng["left_hip_joint"] == pytest.approx((-0.6, 0.6)) and rng["left_shoulder_joint"] == pytest.approx((-1.6, 1.53))
    assert rng["left_leg_joint"] == pytest.approx((-0.64, 1.35)) and "left_wheel_joint" not in rng      # range "0 0": unlimited
    # frictionloss: only the `wheels` default class is rewritten by XMLManager step 6; the hinges keep the 0.1 of
    # the XML's top-level joint default -- and so do the SIX dofs of the base: it is declared <joint type="free"> (flamingo_p_v3.xml:42),
    # not <freejoint>, so the joint defaults (frictionloss 0.1, armature 0.01) apply to it; only its damping is overridden to 0
    fl = np.array(get_field(b, "dof_frictionloss")[:b.nv])
    assert np.allclose(fl, 0.1) and np.allclose(get_field(b, "dof_armature")[:6], 0.01) and np.allclose(get_field(b, "dof_damping")[:6], 0.0)
    assert b.ngeom == 8 and b.npair > 0 and set(get_field(b, "geom_type")[:8]) == {7}    # eight hulls, self-collision on (masks 1 / 1)


def test_compiled_model_facts_w4_p_v2():
    """SURVEY App. A, column w4_p_v2."""
    cm = compile_model(make_config("w4_p_v2", terrain="rocky_hard", random=PARITY_RANDOM))
    b = cm.blob
    assert (b.nq, b.nv, b.nu, b.nbody - 1, b.neq) == (23, 22, 16, 17, 0)
    assert np.array(get_field(b, "body_mass")[:b.nbody]).sum() == pytest.approx(36.20476, abs=1e-5)
    # FL 7-10, FR 11-14, RL 15-18, RR 19-22 (hip, shoulder, leg, wheel); the XML's actuator order is leg by leg
    assert sorted(get_field(b, "ctl_qadr")[:16]) == list(range(7, 23))
    gear = np.array(get_field(b, "ctl_gear")[:16]); qadr = np.array(get_field(b, "ctl_qadr")[:16])
    assert sorted(qadr[gear == -1.5].tolist()) == [9, 13, 17, 21] and (gear[gear != -1.5] == 1).all()   # the four legs
    assert sorted(qadr[np.array(get_field(b, "ctl_velmode")[:16]) == 1].tolist()) == [10, 14, 18, 22]   # the four wheels
    assert np.allclose(get_field(b, "gravity"), [0, 0, -9.807])
    assert np.array(get_field(b, "init_qpos")[:3]).tolist() == [0.0, 0.0, 0.47957] and b.term_mode == 0
    assert b.ngeom == 17 and set(get_field(b, "geom_type")[:17]) == {7}
    # rocky_hard: 512 x 512 samples, half-extent 140 m, z 0.25 m, base 0.1 m (w4_p_v2.xml:179)
    assert (b.hfield_nrow, b.hfield_ncol) == (512, 512) and list(get_field(b, "hfield_size")) == pytest.approx([140, 140, 0.25, 0.1])
    assert cm.hfield.min() == 0.0 and cm.hfield.max() == 1.0 and len(np.unique(cm.hfield)) == 101      # 101 grey levels (App. A)


def test_compiled_model_facts_humanoid_p_v0():
    """SURVEY App. A, column humanoid_p_v0."""
    cm = compile_model(make_config("humanoid_p_v0", terrain="stairs_up_hard", random=PARITY_RANDOM))
    b = cm.blob
    assert (b.nq, b.nv, b.nu, b.nbody - 1, b.neq) == (30, 29, 23, 25, 0)                # 25 bodies incl. the jointless imu_link
    assert np.array(get_field(b, "body_mass")[:b.nbody]).sum() == pytest.approx(61.80264, abs=1e-5)
    assert sorted(get_field(b, "ctl_qadr")[:23]) == list(range(7, 30))                  # torso 7; arms 8-17; legs 18-29
    assert (np.array(get_field(b, "ctl_gear")[:23]) == 1).all() and (np.array(get_field(b, "ctl_velmode")[:23]) == 0).all()
    assert np.allclose(get_field(b, "gravity"), [0, 0, -9.807])
    assert np.array(get_field(b, "init_qpos")[:3]).tolist() == [0.0, 0.0, 1.105] and b.term_mode == 0
    gt = list(get_field(b, "geom_type")[:b.ngeom])
    assert (gt.count(6), gt.count(5), gt.count(7)) == (4, 16, 2)                        # 4 boxes, 16 cylinders, 2 mesh feet
    assert b.heightmap_miss == 5.0                                                      # utils/mujoco_utils.py:141 (z_min_world = -5)
    # stairs_up_hard: the 1024 x 1024 PNG overrides the XML's nrow / ncol = 512; half-extent 5 m, z 0.825 m (humanoid_p_v0.xml:220)
    assert (b.hfield_nrow, b.hfield_ncol) == (1024, 1024) and list(get_field(b, "hfield_size"))[:3] == pytest.approx([5, 5, 0.825])
    assert len(np.unique(cm.hfield)) == 6 and cm.hfield[512, 512] == 0.0                # six levels; the spawn point is at elevation 0


def test_hfield_png_conventions_known_answer(tmp_path):
    """N3 / A2 (SURVEY 8a A2 marks these "[upstream, verify]"): how a terrain PNG becomes elevation.  The convention ASSUMED here,
    and stated so that a MuJoCo cross-check can falsify it: (1) grey levels are normalised by the file's own min and max to [0, 1];
    (2) image rows are flipped -- the image's LAST row is hfield row 0, the row at y = -size_y; (3) column c sits at
    x = -size_x + c * 2 size_x / (ncol - 1), row r at y = -size_y + r * 2 size_y / (nrow - 1); (4) the surface height is
    ground_z + size_z * elevation; (5) the file's resolution overrides the XML's nrow / ncol.  A hand-built 4 x 4 file pins the
    loader; the oracle's vertical ray and the compiled field pin (3) and (4) on a shipped terrain."""
    from PIL import Image
    from cosim_amd.compile import _load_hfield
    from oracle.oracle import Oracle
    img = np.array([[10, 20, 30, 40],          # top row of the image
                    [50, 60, 70, 80],
                    [90, 100, 110, 120],
                    [130, 170, 210, 250]], dtype=np.uint8)     # bottom row of the image
    path = tmp_path / "hand.png"
    Image.fromarray(img, mode="L").save(path)
    h = _load_hfield(str(path))
    assert h.shape == (4, 4) and h.dtype == np.float32
    np.testing.assert_allclose(h[0], (np.array([130, 170, 210, 250]) - 10) / 240.0, rtol=1e-6)      # (2): bottom image row first
    np.testing.assert_allclose(h[3], (np.array([10, 20, 30, 40]) - 10) / 240.0, rtol=1e-6)
    assert h.min() == 0.0 and h.max() == 1.0                                                         # (1)
    flat = tmp_path / "flat.png"
    Image.fromarray(np.full((3, 5), 255, dtype=np.uint8), mode="L").save(flat)
    assert _load_hfield(str(flat)).shape == (3, 5) and (_load_hfield(str(flat)) == 0.0).all()        # constant image -> elevation 0
    # (3), (4), (5) on stairs_up_hard through the compiled model and the oracle's vertical ray (mj_rayHfield restatement)
    cm = compile_model(make_config("humanoid_p_v0", terrain="stairs_up_hard", random=PARITY_RANDOM))
    b = cm.blob
    sx, sy, sz, _ = get_field(b, "hfield_size")
    nr, nc = cm.hfield.shape
    assert (nr, nc) == (1024, 1024)
    o = Oracle(cm)
    gz = get_field(b, "ground_pos")[2]
    rng = np.random.default_rng(0)
    for r, c in rng.integers(1, 1023, size=(12, 2)):
        x, y = -sx + c * 2 * sx / (nc - 1), -sy + r * 2 * sy / (nr - 1)
        assert 10.0 - o.ray_down(x, y, 10.0) == pytest.approx(gz + sz * cm.hfield[r, c], abs=1e-9), (r, c)
    src = np.asarray(Image.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cosim_amd", "assets", "terrain",
                                             "stairs_up_hard.png")).convert("L"), dtype=np.float64)
    r, c = 100, 900
    assert cm.hfield[r, c] == pytest.approx((src[nr - 1 - r, c] - src.min()) / (src.max() - src.min()), abs=1e-7)


@pytest.mark.parametrize("env_id", ["flamingo_light_v1", "flamingo_p_v3"])
def test_env_constants_batched_equals_single(env_id):
    cm = compile_model(make_config(env_id, random=PARITY_RANDOM))
    nb = cm.blob.nbody
    m0 = np.array(get_field(cm.blob, "body_mass")[:nb])
    rng = np.random.default_rng(0)
    masses = m0[None] * (1 + 0.05 * rng.uniform(-1, 1, size=(5, nb)))
    c = env_constants(cm, masses)
    for i in range(5):
        ci = env_constants(cm, masses[i][None])
        for k in c:
            np.testing.assert_allclose(c[k][i], ci[k][0], rtol=1e-12)
    nominal = env_constants(cm, m0[None])
    np.testing.assert_allclose(nominal["dof_invweight0"][0], np.array(get_field(cm.blob, "dof_invweight0")[:cm.blob.nv]))
    assert nominal["meaninertia"][0] == pytest.approx(cm.blob.meaninertia)


def test_obs_config_errors_follow_reference():
    cfg = make_config("flamingo_light_v1")
    dims = obs_to_dim("flamingo_light_v1", cfg)
    c = make_obs_config(cfg, dims, 50.0, True)
    assert (c.stack_size, c.command_dim, c.n_stacked, c.n_non_stacked, c.max_sim_step) == (3, 4, 5, 1, 6000)
    assert list(c.field_interval)[:6] == [1, 1, 1, 1, 1, 1] and c.noise_enabled == 1
    bad = make_config("flamingo_light_v1")
    bad["observation"]["dof_vel"]["freq"] = 0
    with pytest.raises(ValueError):            # wrappers.py:186-187
        make_obs_config(bad, dims, 50.0, True)
    bad = make_config("flamingo_light_v1")
    bad["observation"]["stacked_obs_order"] = ["lin_vel_x"]     # offered by the GUI, provided by no env
    with pytest.raises(KeyError):              # wrappers.py:116
        make_obs_config(bad, dims, 50.0, True)
    slow = make_config("flamingo_light_v1")
    slow["observation"]["dof_vel"]["freq"] = 10
    assert make_obs_config(slow, dims, 50.0, True).field_interval[1] == 5


def test_abi_library_exports_and_layouts():
    L = load_library()
    for sym in EXPORTS:
        assert hasattr(L, sym), sym
    assert L.cosim_model_sizeof() == ctypes.sizeof(CosimModel)
    assert L.cosim_obs_config_sizeof() == ctypes.sizeof(ObsConfig)
    # every entry point declared in include/cosim.h is exported
    import re
    hdr = open(os.path.join(ROOT, "include", "cosim.h")).read()
    declared = set(re.findall(r"\b(cosim_[a-z_]+)\s*\(", hdr))
    assert declared <= set(EXPORTS) | {"cosim_engine"}, declared - set(EXPORTS)
    for sym in declared:
        assert hasattr(L, sym), sym


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cosim_amd.build import build_env
    with pytest.raises(RuntimeError):
        build_env(make_config("flamingo_light_v1"))
    with pytest.raises(NameError):
        build_env({"env": {"id": "nope"}})


def test_oracle_and_engine_share_the_model_layout():
    from oracle.oracle import lib
    assert lib().oracle_model_sizeof() == ctypes.sizeof(CosimModel)


def test_mujoco_crosscheck_is_opportunistic(tmp_path):
    """tools/crosscheck_mujoco.py: builds a self-contained MJCF (hull vertices inline, no STL files)
    and compares the oracle with mj_step where mujoco exists; here it must report 'skipped' (exit 77) and install nothing."""
    import subprocess
    import sys
    import xml.etree.ElementTree as ET
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from tools.crosscheck_mujoco import build_xml
    root = ET.fromstring(build_xml(make_config("w4_p_v2", terrain="rocky_hard", random=PARITY_RANDOM)))
    meshes = root.find("asset").findall("mesh")
    assert meshes and all("vertex" in m.attrib and "file" not in m.attrib for m in meshes)
    assert all(g.attrib.get("class") != "visual" for g in root.iter("geom"))
    try:
        import mujoco  # noqa: F401
        pytest.skip("mujoco is importable here: run tools/crosscheck_mujoco.py for the real comparison")
    except ImportError:
        pass
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "..", "tools", "crosscheck_mujoco.py"), "--steps", "5"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 77 and "SKIPPED" in r.stdout


def test_height_map_on_flat_terrain_is_refused():
    """mj_rayHfield (reference utils/mujoco_utils.py:169) needs an hfield ground; terrain 'flat' makes it a plane."""
    cfg = make_config("w4_p_v2", terrain="flat", height_map=True)
    from cosim_amd.robots import obs_to_dim
    with pytest.raises(ValueError, match="heightfield terrain"):
        make_obs_config(cfg, obs_to_dim("w4_p_v2", cfg), 50.0, True)
    ok = make_config("w4_p_v2", terrain="rocky_easy", height_map=True)
    make_obs_config(ok, obs_to_dim("w4_p_v2", ok), 50.0, True)


def test_one_env_info_stream_is_accepted_by_the_reference_reporter(tmp_path):
    """SURVEY §8b 'info contract (Reporter drop-in)' / §8f N2: the single-env dicts that Runner / FleetReporter cut out of the
    batched info are fed to the reference's own core/reporter.py (importable here: matplotlib only), which must build its PDF.
    Skipped where the reference tree is not mounted (the GPU box)."""
    ref = "/root/reference/core/reporter.py"
    if not os.path.exists(ref):
        pytest.skip("reference tree not mounted")
    import importlib.util
    import torch
    from cosim_amd.runner import Runner
    spec = importlib.util.spec_from_file_location("ref_reporter", ref)
    mod = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mod)
    except ImportError as e:
        pytest.skip(f"reference reporter not importable here: {e}")

    class FakeEnv:                       # the attributes Runner reads; info tensors shaped like BatchedEnv._info's
        num_envs, command_dim, auto_reset = 8, 4, False
    N, nu = 8, 4
    run = Runner.__new__(Runner)
    run.env = FakeEnv()
    cfg = make_config("flamingo_light_v1")
    rep = mod.Reporter(str(tmp_path / "report.pdf"), cfg)
    g = torch.Generator().manual_seed(0)
    for t in range(60):
        buf = torch.randn((N, 4 + 3 * nu), generator=g)
        info = {"dt": 0.02, "action": torch.randn((N, nu), generator=g), "action_diff_RMSE": buf[:, 0], "lin_vel_x": buf[:, 1],
                "lin_vel_y": buf[:, 2], "ang_vel_yaw": buf[:, 3], "torque": buf[:, 4:4 + nu], "set_points": buf[:, 4 + nu:4 + 2 * nu],
                "state": buf[:, 4 + 2 * nu:]}
        for i in range(4):
            info[f"user_command_{i}"] = torch.full((N,), 0.1 * i)
        one = run._one_env_info(info, 3)
        assert isinstance(one["lin_vel_x"], float) and one["torque"].shape == (nu,) and one["dt"] == 0.02
        rep.write_info(one)
    rep.generate_report()
    assert os.path.getsize(tmp_path / "report.pdf") > 10000


@pytest.mark.parametrize("robot", ["w4_p_v2", "flamingo_p_v3", "humanoid_p_v0"])
def test_hull_support_map_returns_the_vertex_of_the_full_scan(robot):
    """The reference's support function for a mesh geom scans every hull vertex (mjc_support: arg max of dir . vertex).  The engine
    looks the direction up in a cube map of candidate lists instead (csrc/cosim_hullmap.h); the answer must be the vertex the scan
    finds, ties included (lowest index).  Host-only hook, every hull the robots carry; random directions plus the ones that make
    ties or sit on cell borders: face normals of the hull, the coordinate axes, cube-map cell corners, and nudged copies."""
    from scipy.spatial import ConvexHull
    from cosim_amd.model import get_field
    L = load_library()
    cm = compile_model(make_config(robot, num_envs=1))
    b = cm.blob
    gadr = np.array(get_field(b, "geom_hulladr")[:b.ngeom]); gnum = np.array(get_field(b, "geom_hullnum")[:b.ngeom])
    rng = np.random.default_rng(4)
    seen, checked = set(), 0
    for g in range(b.ngeom):
        if gnum[g] < 32 or (gadr[g], gnum[g]) in seen:
            continue
        seen.add((gadr[g], gnum[g]))
        V = np.ascontiguousarray(cm.hull_vert[gadr[g]:gadr[g] + gnum[g]], dtype=np.float32)
        adr = np.ascontiguousarray(cm.hull_adr[gadr[g]:gadr[g] + gnum[g] + 1], dtype=np.int32)
        nbr = np.ascontiguousarray(cm.hull_nbr, dtype=np.int32)
        D = [rng.normal(size=(100000, 3))]
        D.append(ConvexHull(V.astype(np.float64)).equations[:, :3])                          # face normals: whole faces tie
        D.append(np.concatenate([np.eye(3), -np.eye(3)]))
        R = 16
        t = -1.0 + 2.0 * np.arange(R + 1) / R
        uu, vv = np.meshgrid(t, t, indexing="ij")
        for a in range(3):
            for s in (1.0, -1.0):
                P = np.zeros((uu.size, 3)); P[:, a] = s; P[:, (a + 1) % 3] = uu.ravel(); P[:, (a + 2) % 3] = vv.ravel()
                D.append(P)                                                                   # cell corners and borders
        D = np.concatenate(D)
        D = np.concatenate([D, D + 1e-6 * rng.normal(size=D.shape), D * 1e-3, V.astype(np.float64) - V.mean(0)])
        D = np.ascontiguousarray(D / np.maximum(np.linalg.norm(D, axis=1), 1e-30)[:, None] * rng.uniform(0.5, 2.0, size=(len(D), 1)), dtype=np.float32)
        mi = np.zeros(len(D), dtype=np.int32); si = np.zeros(len(D), dtype=np.int32); st = np.zeros(3, dtype=np.int32)
        rc = L.cosim_hull_support_check(V.ctypes.data, len(V), adr.ctypes.data, nbr.ctypes.data, D.ctypes.data, len(D), mi.ctypes.data,
                                        si.ctypes.data, st.ctypes.data)
        assert rc == 0
        assert np.array_equal(mi, si), (robot, g, int((mi != si).sum()))
        assert st[2] == 6 * R * R and st[0] / st[2] < 12.0 and st[1] < len(V), st         # a handful of candidates per cell
        checked += 1
    assert checked >= 2
