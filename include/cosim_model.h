/* cosim_model.h — the compiled-model blob ("ModelBlob") handed across the C ABI.
 *
 * This is the data format on the model side of the hot path: what MuJoCo's compiler
 * produces as an mjModel for the reference (reference call site
 * envs/flamingo_light_v1/flamingo_light_v1.py:81-87, MujocoEnv.__init__ ->
 * MjModel.from_xml_path) is produced here by cosim_amd/compile.py as one flat,
 * pointer-free struct of doubles and ints.  Field names follow mjModel where a field
 * has the same meaning.  The HIP engine converts it to fp32 device constants at
 * cosim_create(); the CPU oracle (oracle/, test infrastructure) reads it as is.
 *
 * The struct is mirrored field-for-field by cosim_amd/model.py (ctypes); the test
 * suite checks sizeof and field offsets through cosim_model_sizeof()/..._offsetof().
 */
#ifndef COSIM_MODEL_H
#define COSIM_MODEL_H

#ifdef __cplusplus
extern "C" {
#endif

#define CS_MAXBODY 32  /* incl. world body 0 */
#define CS_MAXJNT 32
#define CS_MAXDOF 32
#define CS_MAXQ 33
#define CS_MAXGEOM 40  /* collision-enabled robot geoms (ground is described apart) */
#define CS_MAXEQ 4
#define CS_MAXU 24
#define CS_MAXOBSJ 24  /* joints gathered into dof_pos / dof_vel */
#define CS_MAXPAIR 256 /* robot-robot geom pairs that pass the contype/conaffinity + exclude filter */
#define CS_MAXINFOSTATE 24

/* mjtJoint / mjtGeom values */
#define CS_JNT_FREE 0
#define CS_JNT_HINGE 3
#define CS_GEOM_PLANE 0
#define CS_GEOM_HFIELD 1
#define CS_GEOM_SPHERE 2
#define CS_GEOM_CAPSULE 3
#define CS_GEOM_CYLINDER 5
#define CS_GEOM_BOX 6
#define CS_GEOM_MESH 7

#define CS_SOLVER_NEWTON 0
#define CS_SOLVER_PGS 1

#define CS_MODEL_MAGIC 0x43534d31 /* "CSM1" */

typedef struct cosim_model {
  int magic;
  /* sizes */
  int nq, nv, nu, nbody, njnt, ngeom, neq, npair;
  int nhullvert, nhulledge; /* lengths of the hull arrays passed next to the blob */
  /* <option> */
  int solver, iterations, ls_iterations, frame_skip;
  double timestep, tolerance, ls_tolerance, impratio;
  double gravity[3];
  double meaninertia; /* stat.meaninertia at qpos0 (nominal masses) */

  /* bodies */
  int body_parentid[CS_MAXBODY], body_rootid[CS_MAXBODY];
  int body_jntnum[CS_MAXBODY], body_jntadr[CS_MAXBODY];
  int body_dofnum[CS_MAXBODY], body_dofadr[CS_MAXBODY];
  double body_pos[CS_MAXBODY][3], body_quat[CS_MAXBODY][4];
  double body_ipos[CS_MAXBODY][3], body_iquat[CS_MAXBODY][4];
  double body_mass[CS_MAXBODY], body_inertia[CS_MAXBODY][3];
  double body_invweight0[CS_MAXBODY][2];

  /* joints */
  int jnt_type[CS_MAXJNT], jnt_qposadr[CS_MAXJNT], jnt_dofadr[CS_MAXJNT], jnt_bodyid[CS_MAXJNT];
  int jnt_limited[CS_MAXJNT], jnt_actfrclimited[CS_MAXJNT];
  double jnt_pos[CS_MAXJNT][3], jnt_axis[CS_MAXJNT][3], jnt_range[CS_MAXJNT][2], jnt_margin[CS_MAXJNT];
  double jnt_solref[CS_MAXJNT][2], jnt_solimp[CS_MAXJNT][5];
  double jnt_actfrcrange[CS_MAXJNT][2];
  double qpos0[CS_MAXQ];

  /* dofs */
  int dof_bodyid[CS_MAXDOF], dof_jntid[CS_MAXDOF], dof_parentid[CS_MAXDOF];
  double dof_armature[CS_MAXDOF], dof_damping[CS_MAXDOF], dof_frictionloss[CS_MAXDOF];
  double dof_invweight0[CS_MAXDOF];
  double dof_solref[CS_MAXDOF][2], dof_solimp[CS_MAXDOF][5];

  /* ground geom (geom 0 of the reference models: "ground", plane when terrain == flat) */
  int ground_type, ground_contype, ground_conaffinity, ground_condim;
  double ground_friction[3], ground_solref[2], ground_solimp[5], ground_solmix, ground_margin, ground_gap;
  double ground_pos[3];
  /* hfield (ground_type == CS_GEOM_HFIELD): data passed next to the blob, row-major [nrow][ncol] in [0,1] */
  int hfield_nrow, hfield_ncol;
  double hfield_size[4]; /* half-x, half-y, elevation z, base */

  /* robot collision geoms */
  int geom_type[CS_MAXGEOM], geom_bodyid[CS_MAXGEOM], geom_contype[CS_MAXGEOM], geom_conaffinity[CS_MAXGEOM];
  int geom_condim[CS_MAXGEOM], geom_ground[CS_MAXGEOM]; /* geom_ground: passes the filter against the ground */
  int geom_hulladr[CS_MAXGEOM], geom_hullnum[CS_MAXGEOM]; /* mesh geoms: slice of the hull vertex array */
  double geom_pos[CS_MAXGEOM][3], geom_quat[CS_MAXGEOM][4], geom_size[CS_MAXGEOM][3];
  double geom_friction[CS_MAXGEOM][3], geom_solref[CS_MAXGEOM][2], geom_solimp[CS_MAXGEOM][5];
  double geom_solmix[CS_MAXGEOM], geom_margin[CS_MAXGEOM], geom_gap[CS_MAXGEOM];
  double geom_rbound[CS_MAXGEOM];     /* bounding-sphere radius about geom_rcenter (body frame) */
  double geom_rcenter[CS_MAXGEOM][3];
  double geom_aabb[CS_MAXGEOM][6];    /* body-frame box around the geom: centre xyz, half-extent xyz (broadphase) */
  double geom_center[CS_MAXGEOM][3];  /* body-frame interior point handed to MPR as the geom centre (mjccd_center: geom_xpos;
                                         for a mesh MuJoCo's geom frame sits at the mesh CoM = the hull's volume centroid) */
  int pair_geom1[CS_MAXPAIR], pair_geom2[CS_MAXPAIR]; /* robot-robot candidate pairs (self collision) */

  /* equality: connect */
  int eq_body1[CS_MAXEQ], eq_body2[CS_MAXEQ];
  double eq_anchor1[CS_MAXEQ][3], eq_anchor2[CS_MAXEQ][3];
  double eq_solref[CS_MAXEQ][2], eq_solimp[CS_MAXEQ][5];

  /* actuators: motors on hinge joints */
  int act_jntid[CS_MAXU], act_dofid[CS_MAXU], act_ctrllimited[CS_MAXU];
  double act_gear[CS_MAXU], act_ctrlrange[CS_MAXU][2];

  /* IMU site (framequat / gyro / velocimeter all sit on it) */
  int imu_bodyid;
  double imu_pos[3], imu_quat[4];
  double gyro_cutoff, velocimeter_cutoff;

  /* ---- robot-env layer (reference envs/<robot>/<robot>.py step/_get_obs/_get_info) ---- */
  /* torque_i = clip(gamma_i * (kp_i (a_i s_i - g_i q_i) + kd_i ((vel_i ? a_i s_i : 0) - g_i qd_i)), +-maxtq_i) */
  int ctl_velmode[CS_MAXU];  /* 1: velocity-PD (wheels) */
  int ctl_qadr[CS_MAXU], ctl_dadr[CS_MAXU];
  double ctl_kp[CS_MAXU], ctl_kd[CS_MAXU], ctl_scale[CS_MAXU], ctl_gear[CS_MAXU], ctl_gamma[CS_MAXU], ctl_maxtq[CS_MAXU];
  int nobs_pos, nobs_vel;
  int obs_qadr[CS_MAXOBSJ], obs_dadr[CS_MAXOBSJ];
  double obs_qgear[CS_MAXOBSJ], obs_dgear[CS_MAXOBSJ];
  int ninfo_state;                      /* info["state"]: entries are qpos (kind 0) or qvel (kind 1) reads */
  int info_kind[CS_MAXINFOSTATE], info_adr[CS_MAXINFOSTATE];
  double info_gear[CS_MAXINFOSTATE];
  int init_noise_nq, init_noise_qadr[CS_MAXQ]; /* qpos entries that get U(-init_noise, +init_noise) at reset */
  double init_qpos[CS_MAXQ];                    /* reference initial_qpos() without noise */
  int term_mode;                                /* 0: never (light/w4/humanoid), 1: cfrc_ext > 1 on listed bodies (p_v3) */
  int nterm_body, term_body[CS_MAXBODY];
  double heightmap_miss;                        /* robot_z - z_min_world on a ray miss: 1.0 (humanoid 5.0) */
  int magic_end;
} cosim_model_t;

#ifdef __cplusplus
}
#endif
#endif /* COSIM_MODEL_H */
