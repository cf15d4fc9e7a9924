// cosim_dev.h — device-side model/state layout of the MI355X rollout engine (internal to libcosim_hip.so).
//
// One environment per wavefront (64 lanes).  Lane l plays body l, dof l, geom l, actuator l and constraint row l in
// the phases where that object kind is processed; per-env intermediates live in LDS, per-env persistent state lives
// in one contiguous HBM record per env (coalesced: lane l reads rec[l], rec[l+64], ...).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cosim.h"

namespace cosim {

constexpr int MAXB = CS_MAXBODY;
constexpr int MAXD = CS_MAXDOF;
constexpr int MAXG = CS_MAXGEOM;
constexpr int MAXU = CS_MAXU;
constexpr int MAXEQ = CS_MAXEQ;
constexpr int MAXROW = 64;        // constraint rows per env == lanes per wave
constexpr int MAXFRAME = 256;     // single-frame observation elements (stacked + non-stacked)
constexpr int MAXTRI = MAXD * (MAXD + 1) / 2;

// fp32 device copy of the ModelBlob plus derived tables
struct DevModel {
  int nq, nv, nu, nbody, njnt, ngeom, neq, nfric;
  int frame_skip, iterations, ls_iterations, maxdepth;
  int ground_type, hfield_nrow, hfield_ncol, nhullvert;
  int imu_body, term_mode, nterm_body, ntri;
  float timestep, tolerance, ls_tolerance, impratio;
  float gravity[3];
  float ground_pos[3];
  float hfield_size[4];
  float imu_pos[3], imu_quat[4], gyro_cutoff, vel_cutoff, heightmap_miss;
  // bodies (one joint per body at most)
  int body_parent[MAXB], body_level[MAXB], body_jtype[MAXB] /* -1 none, 0 free, 3 hinge */, body_qadr[MAXB], body_dadr[MAXB];
  int body_lastdof[MAXB];      // last dof of the body or of its nearest ancestor with dofs (-1: none)
  unsigned body_subtree[MAXB]; // bit c set: body c is in the subtree of this body (incl. itself)
  float body_pos[MAXB][3], body_quat[MAXB][4], body_ipos[MAXB][3], body_iquat[MAXB][4], body_inertia[MAXB][3];
  float jnt_pos[MAXB][3], jnt_axis[MAXB][3], jnt_q0[MAXB];
  int jnt_limited[MAXB];
  float jnt_range[MAXB][2], jnt_margin[MAXB], jnt_solref[MAXB][2], jnt_solimp[MAXB][5];
  // dofs
  int dof_body[MAXD], dof_parent[MAXD], dof_frclimited[MAXD];
  float dof_armature[MAXD], dof_damping[MAXD], dof_solref[MAXD][2], dof_solimp[MAXD][5], dof_frcrange[MAXD][2];
  int dof_act[MAXD];  // actuator driving this dof (-1: none)
  int fric_dof[MAXD]; // dofs that carry a frictionloss row in the model (nfric entries)
  // robot collision geoms against the ground (contact parameters already mixed with the ground's)
  int geom_type[MAXG], geom_body[MAXG], geom_ground[MAXG], geom_hulladr[MAXG], geom_hullnum[MAXG], geom_condim[MAXG];
  float geom_pos[MAXG][3], geom_quat[MAXG][4], geom_size[MAXG][3], geom_rbound[MAXG], geom_rcenter[MAXG][3];
  float geom_solref[MAXG][2], geom_solimp[MAXG][5], geom_margin[MAXG], geom_includemargin[MAXG];
  // equality connect
  int eq_body1[MAXEQ], eq_body2[MAXEQ];
  float eq_anchor1[MAXEQ][3], eq_anchor2[MAXEQ][3], eq_solref[MAXEQ][2], eq_solimp[MAXEQ][5];
  // actuators + robot-env control law
  int act_dof[MAXU], act_ctrllimited[MAXU], ctl_velmode[MAXU], ctl_qadr[MAXU], ctl_dadr[MAXU];
  float act_gear[MAXU], act_ctrlrange[MAXU][2];
  float ctl_scale[MAXU], ctl_gear[MAXU], ctl_gamma[MAXU], ctl_maxtq[MAXU];
  // robot-env observation / info / reset tables
  int nobs_pos, nobs_vel, ninfo_state, init_noise_nq;
  int obs_qadr[CS_MAXOBSJ], obs_dadr[CS_MAXOBSJ], info_kind[CS_MAXINFOSTATE], info_adr[CS_MAXINFOSTATE];
  int init_noise_qadr[CS_MAXQ];
  float obs_qgear[CS_MAXOBSJ], obs_dgear[CS_MAXOBSJ], info_gear[CS_MAXINFOSTATE], init_qpos[CS_MAXQ];
  int term_body[MAXB];
  // packed lower-triangle index -> (row, col)
  unsigned char tri_row[MAXTRI], tri_col[MAXTRI];
};

// wrapper layer, expanded per single-frame element
struct DevObs {
  int stack_size, command_dim, stacked_dim, non_stacked_dim, state_dim, frame_dim /* stacked_dim + non_stacked_dim */;
  int position_command, max_sim_step, auto_reset, noise_enabled, info_dim;
  float action_delay_prob, init_noise;
  float command_scales[CS_MAXCMD];
  int hm_res_x, hm_res_y;
  float hm_size_x, hm_size_y;
  // per frame element e (stacked frame first, then the non-stacked part)
  unsigned char el_field[MAXFRAME];  // CS_OBS_*
  unsigned short el_index[MAXFRAME]; // index inside the field
  unsigned char el_interval[MAXFRAME];
  float el_scale[MAXFRAME];
  float noise_mean[8], noise_std[8], noise_lower[8], noise_upper[8];
};

// per-env HBM record layouts (float offsets)
struct Layout {
  // state record
  int s_qpos, s_qvel, s_warm, s_delay, s_lastact, s_cache, s_stack, s_meta, s_stride;
  // meta words (int bits): [0] sim_step [1] step_count [2] has_prev [3] episode [4] nan_resets
  // parameter record
  int p_mass, p_binvw, p_dinvw, p_floss, p_gmu, p_kp, p_kd, p_mean, p_stride;
};

}  // namespace cosim
