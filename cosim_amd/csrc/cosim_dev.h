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

// fp32 device copy of the ModelBlob.  Everything a lane needs about "its" body / dof / geom / actuator / table row sits
// in ONE 512-byte record per lane index, so every model read is <record base> + <immediate offset>: no per-array
// address registers (a flat struct of arrays cost ~100 VGPRs of hoisted 64-bit addresses and spilled to scratch).
struct LaneRec {
  // ---- body (index = body id); one joint per body at most
  int b_parent, b_level, b_jtype /* -1 none, 0 free, 3 hinge */, b_qadr, b_dadr, b_lastdof;
  unsigned b_subtree;  // bit c: body c is in this body's subtree (incl. itself)
  unsigned b_dofmask;  // bit d: dof d is on the chain from the root to this body
  float b_pos[3], b_quat[4], b_ipos[3], b_iquat[4], b_inertia[3], j_pos[3], j_axis[3], j_q0;
  int j_limited;
  float j_range[2], j_margin, j_solref[2] /* all *_solref: (K, B) of mj_makeImpedance, filled by the host */, j_solimp[5];
  // ---- dof (index = dof id)
  int d_body, d_parent, d_frclimited, d_act, d_fric /* i-th dof that carries a frictionloss row */;
  unsigned d_ancmask;  // ancestors of this dof incl. itself
  float d_armature, d_damping, d_solref[2], d_solimp[5], d_frcrange[2];
  // ---- robot collision geom (index = geom id); contact parameters already mixed with the ground's
  int g_type, g_body, g_ground, g_hulladr, g_hullnum;
  float g_pos[3], g_quat[4], g_size[3], g_rbound, g_rcenter[3], g_solref[2], g_solimp[5], g_margin, g_incmargin;
  // ---- actuator + robot-env control law (index = actuator id)
  int a_dof, a_ctrllimited, a_velmode, a_qadr, a_dadr;
  float a_gear, a_ctrlrange[2], a_scale, a_cgear, a_gamma, a_maxtq;
  // ---- observation / info / reset tables (index = table row)
  int o_qadr, o_dadr, i_kind, i_adr, n_qadr, t_body;
  float o_qgear, o_dgear, i_gear, init_qpos;
  // ---- equality connect (index = equality id)
  int e_body1, e_body2;
  float e_anchor1[3], e_anchor2[3], e_solref[2], e_solimp[5];
  float g_half[3];  // half-extents of the body-frame box around the geom (centre = g_rcenter)
};
static_assert(sizeof(LaneRec) == 512, "LaneRec must stay 512 bytes (immediate-offset addressing)");

struct DevModel {
  int nq, nv, nu, nbody, njnt, ngeom, neq, nfric, npair;
  int frame_skip, iterations, ls_iterations, maxdepth;
  int ground_type, hfield_nrow, hfield_ncol, nhullvert;
  int imu_body, term_mode, nterm_body, ntri;
  int nobs_pos, nobs_vel, ninfo_state, init_noise_nq;
  int any_jpos;            // some hinge joint has an anchor away from its body's origin (none of the cosim robots does)
  unsigned imu_dofmask;
  unsigned term_bodymask;  // bodies whose cfrc_ext ends the episode (term_mode 1)
  float timestep, tolerance, ls_tolerance, impratio;
  float gravity[3];
  float ground_pos[3];
  float hfield_size[4];
  float imu_pos[3], imu_quat[4], gyro_cutoff, vel_cutoff, heightmap_miss;
  LaneRec rec[64];
  // packed lower-triangle index -> (row, col)
  unsigned char tri_row[MAXTRI], tri_col[MAXTRI];
  int g_hullmap[64];   // per geom: first cell of its hull's support map (cosim_hullmap.h) in KArgs::hull_cell, or -1 (no map: scan)
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
  float noise_ca[8], noise_cb[8];  // Phi((lower - mean) / std), Phi((upper - mean) / std): constants of the inverse-CDF sampler
};

// per-env HBM record layouts (float offsets)
struct Layout {
  // state record
  int s_qpos, s_qvel, s_warm, s_delay, s_lastact, s_cache, s_stack, s_meta, s_stride;
  // meta words (int bits): [0] sim_step [1] step_count [2] has_prev [3] sum nefc [4] nan_resets [5] Newton iterations
  // [6] line-search evaluations [7] Hessian factorisations (cumulative since creation; [3] per substep) [8] contacts left out for
  // lack of slots / rows [9] limit rows left out [10] most contacts detected in one substep [11] episodes ended
  // [12] control steps redone by the large-capacity kernel (env_fixup_kernel) [13] heightfield prism walks cut short (also in [8])
  static constexpr int NMETA = 16;
  // parameter record
  int p_mass, p_binvw, p_dinvw, p_floss, p_gmu, p_kp, p_kd, p_mean, p_stride;
};

}  // namespace cosim
