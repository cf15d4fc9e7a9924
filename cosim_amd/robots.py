"""Per-robot constants of the reference's robot-env classes, as data.

Each entry restates what one ``envs/<robot>/<robot>.py`` + ``manager/xml_manager.py``
pair hard-codes (reference file:line cited per field), so that one engine can serve all
robots: the control law of ``step()``, the joint gathers of ``_get_obs()`` /
``_get_info()``, the ``initial_qpos()`` recipe, the termination rule and the body lists
``XMLManager`` randomises.
"""
from __future__ import annotations

from typing import Dict, List


def _light_v1(hw: dict) -> dict:
    # reference envs/flamingo_light_v1/flamingo_light_v1.py:22-33,131-152
    s, w = hw["action_scales"]["shoulder"], hw["action_scales"]["wheel"]
    return dict(
        actuators=[  # action order == XML actuator order (flamingo_light_v1.xml:285-288)
            dict(joint="left_shoulder_joint", vel=False, kp=hw["Kp_shoulder"], kd=hw["Kd_shoulder"], scale=s,
                 gear=1.0, gamma=1.0, maxtq=hw["leg_max_torque"]),   # :149 clips shoulders with leg_max_torque
            dict(joint="right_shoulder_joint", vel=False, kp=hw["Kp_shoulder"], kd=hw["Kd_shoulder"], scale=s,
                 gear=1.0, gamma=1.0, maxtq=hw["leg_max_torque"]),
            dict(joint="left_wheel_joint", vel=True, kp=0.0, kd=hw["Kd_wheel"], scale=w, gear=1.0, gamma=1.0,
                 maxtq=hw["wheel_max_torque"]),
            dict(joint="right_wheel_joint", vel=True, kp=0.0, kd=hw["Kd_wheel"], scale=w, gear=1.0, gamma=1.0,
                 maxtq=hw["wheel_max_torque"]),
        ],
    )


def _p_v3(hw: dict) -> dict:
    # reference envs/flamingo_p_v3/flamingo_p_v3.py:23-48,150-188
    sc = hw["action_scales"]
    g, gam = hw["gear_ratio"], hw["gamma"]
    lm = hw["leg_max_torque"]  # hips, shoulders and legs are all clipped with leg_max_torque (:183-185)
    acts = []
    for name, scale, kp, kd, gear, gamma in (
            ("hip", sc["hip"], hw["Kp_hip"], hw["Kd_hip"], 1.0, 1.0),
            ("shoulder", sc["shoulder"], hw["Kp_shoulder"], hw["Kd_shoulder"], 1.0, 1.0),
            ("leg", sc["leg"], hw["Kp_leg"], hw["Kd_leg"], g, gam)):
        for side in ("left", "right"):
            acts.append(dict(joint=f"{side}_{name}_joint", vel=False, kp=kp, kd=kd, scale=scale, gear=gear,
                             gamma=gamma, maxtq=lm))
    for side in ("left", "right"):
        acts.append(dict(joint=f"{side}_wheel_joint", vel=True, kp=0.0, kd=hw["Kd_wheel"], scale=sc["wheel"],
                         gear=1.0, gamma=1.0, maxtq=hw["wheel_max_torque"]))
    return dict(actuators=acts)


def _w4(hw: dict) -> dict:
    # reference envs/w4_p_v2/w4_p_v2.py:22-45,151-187 (hips / shoulders / legs use their own max torque here)
    sc = hw["action_scales"]
    g, gam = hw["gear_ratio"], hw["gamma"]
    legs = ("FL", "FR", "RL", "RR")
    acts = []
    for name, kp, kd, gear, gamma, mx in (("hip", hw["Kp_hip"], hw["Kd_hip"], 1.0, 1.0, hw["hip_max_torque"]),
                                          ("shoulder", hw["Kp_shoulder"], hw["Kd_shoulder"], 1.0, 1.0, hw["shoulder_max_torque"]),
                                          ("leg", hw["Kp_leg"], hw["Kd_leg"], g, gam, hw["leg_max_torque"])):
        for leg in legs:
            acts.append(dict(joint=f"{leg}_{name}_joint", vel=False, kp=kp, kd=kd, scale=sc[name], gear=gear, gamma=gamma, maxtq=mx))
    for leg in legs:
        acts.append(dict(joint=f"{leg}_wheel_joint", vel=True, kp=0.0, kd=hw["Kd_wheel"], scale=sc["wheel"], gear=1.0,
                         gamma=1.0, maxtq=hw["wheel_max_torque"]))
    return dict(actuators=acts)


_HUMANOID_GROUPS = ["hip_pitch", "torso", "hip_roll", "shoulder_pitch", "hip_yaw", "shoulder_roll", "knee", "shoulder_yaw",
                    "ankle_pitch", "elbow_pitch", "ankle_roll", "elbow_yaw"]


def _humanoid_joints():
    # joint_names_in_order (reference envs/humanoid_p_v0/humanoid_p_v0.py:139-150) == XML actuator order (:157-179)
    out = []
    for g in _HUMANOID_GROUPS:
        out += [("torso_joint", g)] if g == "torso" else [(f"left_{g}_joint", g), (f"right_{g}_joint", g)]
    return out


def _humanoid(hw: dict) -> dict:
    # reference envs/humanoid_p_v0/humanoid_p_v0.py:22-90,186-250
    sc = hw["action_scales"]
    return dict(actuators=[dict(joint=j, vel=False, kp=hw[f"Kp_{g}"], kd=hw[f"Kd_{g}"], scale=sc[g], gear=1.0, gamma=1.0,
                                maxtq=hw[f"{g}_joint_max_torque"]) for j, g in _humanoid_joints()])


ROBOTS: Dict[str, dict] = {
    "flamingo_light_v1": dict(
        xml="flamingo_light_v1.xml",
        base_body="base_link",
        control=_light_v1,
        # _get_obs gathers (flamingo_light_v1.py:95-98); (joint, gear multiplier)
        obs_pos=[("left_shoulder_joint", 1.0), ("right_shoulder_joint", 1.0)],
        obs_vel=[("left_shoulder_joint", 1.0), ("right_shoulder_joint", 1.0),
                 ("left_wheel_joint", 1.0), ("right_wheel_joint", 1.0)],
        # info["state"] (flamingo_light_v1.py:171): [dof_pos[0], dof_pos[1], dof_vel[2], dof_vel[3]]
        info_state=[("pos", 0), ("pos", 1), ("vel", 2), ("vel", 3)],
        init_height=0.13,  # :227
        init_noise_joints=["left_shoulder_joint", "right_shoulder_joint", "left_wheel_joint", "right_wheel_joint"],  # :229
        term_mode=0, term_bodies=[],  # _is_done: empty body list -> never (:189-207)
        # XMLManager (manager/xml_manager.py:11-12,60)
        mass_bodies=["base_link", "left_shoulder_link", "right_shoulder_link", "left_wheel_link", "right_wheel_link"],
        friction_bodies=["left_wheel_link", "right_wheel_link"],
        heightmap_miss=1.0,
    ),
    "flamingo_p_v3": dict(
        xml="flamingo_p_v3.xml",
        base_body="base_link",
        control=_p_v3,
        obs_pos=[("left_hip_joint", 1.0), ("right_hip_joint", 1.0), ("left_shoulder_joint", 1.0),
                 ("right_shoulder_joint", 1.0), ("left_leg_joint", "gear"), ("right_leg_joint", "gear")],
        obs_vel=[("left_hip_joint", 1.0), ("right_hip_joint", 1.0), ("left_shoulder_joint", 1.0),
                 ("right_shoulder_joint", 1.0), ("left_leg_joint", "gear"), ("right_leg_joint", "gear"),
                 ("left_wheel_joint", 1.0), ("right_wheel_joint", 1.0)],
        # info["state"] reads the *ungeared* dof_pos/dof_vel (flamingo_p_v3.py:201-207)
        info_state=[("pos", 0), ("pos", 1), ("pos", 2), ("pos", 3), ("pos", 4), ("pos", 5), ("vel", 6), ("vel", 7)],
        info_state_geared=False,
        init_height=0.61282,  # :251
        init_noise_joints="all_hinge",  # qpos[7:15] (:253-254)
        term_mode=1, term_bodies=["base_link", "left_hip_link", "right_hip_link", "left_shoulder_link",
                                  "right_shoulder_link"],  # cfrc_ext rows 1,2,6,3,7 (:225-233)
        mass_bodies=["base_link", "left_hip_link", "right_hip_link", "left_shoulder_link", "right_shoulder_link",
                     "left_leg_link", "right_leg_link", "left_wheel_link", "right_wheel_link"],
        friction_bodies=["left_wheel_link", "right_wheel_link"],
        heightmap_miss=1.0,
    ),
    "w4_p_v2": dict(
        xml="w4_p_v2.xml",
        base_body="base_link",
        control=_w4,
        # _get_obs (w4_p_v2.py:104-120): hips, shoulders, legs (geared) ; dof_vel adds the wheels
        obs_pos=[(f"{l}_{n}_joint", "gear" if n == "leg" else 1.0) for n in ("hip", "shoulder", "leg") for l in ("FL", "FR", "RL", "RR")],
        obs_vel=[(f"{l}_{n}_joint", "gear" if n == "leg" else 1.0) for n in ("hip", "shoulder", "leg") for l in ("FL", "FR", "RL", "RR")]
        + [(f"{l}_wheel_joint", 1.0) for l in ("FL", "FR", "RL", "RR")],
        info_state=[("pos", i) for i in range(12)] + [("vel", i) for i in range(12, 16)],   # ungeared (:204-206)
        info_state_geared=False,
        init_height=0.47957,  # :246
        init_noise_joints="all_hinge",
        term_mode=0, term_bodies=[],  # _is_done returns False (:225-226)
        mass_bodies=["base_link", "FL_hip_link", "FR_hip_link", "RL_hip_link", "RR_hip_link", "FL_shoulder_link", "FR_shoulder_link",
                     "RL_shoulder_link", "RR_shoulder_link", "FL_leg_link", "FR_leg_link", "RL_leg_link", "RR_leg_link",
                     "FL_wheel_link", "FR_wheel_link", "RL_wheel_link", "RR_wheel_link"],
        friction_bodies=["FL_wheel_link", "FR_wheel_link", "RL_wheel_link", "RR_wheel_link"],
        heightmap_miss=1.0,
    ),
    "humanoid_p_v0": dict(
        xml="humanoid_p_v0.xml",
        base_body="pelvis_link",
        control=_humanoid,
        obs_pos=[(j, 1.0) for j, _ in _humanoid_joints()],
        obs_vel=[(j, 1.0) for j, _ in _humanoid_joints()],
        info_state=[("pos", i) for i in range(23)],   # joint_state = dof_pos (:268)
        info_state_geared=False,
        init_height=1.105,  # :308
        init_noise_joints="all_hinge",
        term_mode=0, term_bodies=[],
        mass_bodies=["pelvis_link", "torso_link",
                     "left_shoulder_pitch_link", "left_shoulder_roll_link", "left_shoulder_yaw_link",
                     "left_elbow_pitch_link", "left_elbow_yaw_link",
                     "right_shoulder_pitch_link", "right_shoulder_roll_link", "right_shoulder_yaw_link",
                     "right_elbow_pitch_link", "right_elbow_yaw_link",
                     "left_hip_pitch_link", "left_hip_roll_link", "left_hip_yaw_link",
                     "left_knee_link", "left_ankle_pitch_link", "left_ankle_roll_link",
                     "right_hip_pitch_link", "right_hip_roll_link", "right_hip_yaw_link",
                     "right_knee_link", "right_ankle_pitch_link", "right_ankle_roll_link"],
        friction_bodies=["left_ankle_roll_link", "right_ankle_roll_link"],
        heightmap_miss=5.0,   # z_min_world = -5.0 (envs/humanoid_p_v0/utils/mujoco_utils.py:141)
    ),
}


def obs_to_dim(env_id: str, config: dict) -> Dict[str, int]:
    """``obs_to_dim`` of the robot env (e.g. flamingo_light_v1.py:68-77)."""
    r = ROBOTS[env_id]
    hm = config["observation"].get("height_map")
    hm_dim = int(hm["res_x"] * hm["res_y"]) if hm is not None else 0
    nu = len(r["control"](config["hardware"])["actuators"])
    return {
        "dof_pos": len(r["obs_pos"]),
        "dof_vel": len(r["obs_vel"]),
        "ang_vel": 3,
        "lin_vel": 3,
        "projected_gravity": 3,
        "last_action": nu,
        "height_map": hm_dim,
        "command": config["observation"]["command_dim"],
    }


def robot_ids() -> List[str]:
    return list(ROBOTS)
