// cosim_engine.hip — host side of libcosim_hip.so: the C ABI declared in include/cosim.h.
//
// Converts the fp64 ModelBlob into fp32 device tables, owns the per-env HBM records (state + randomised parameters),
// and launches the one-wave-per-env kernel (cosim_kernels.hip) specialised for the model's (nv, nbody).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <string.h>

#include <string>
#include <vector>

#include "cosim_kernels.hip"
#include "cosim_mlp.hip"

using namespace cosim;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) return fail(COSIM_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

struct cosim_engine {
  int n_envs = 0, device = 0;
  cosim_model_t model;
  cosim_obs_config_t obs_cfg;
  DevModel hm;  // host copies
  DevObs ho;
  Layout lay;
  DevModel* d_model = nullptr;
  DevObs* d_obs = nullptr;
  float *d_state = nullptr, *d_params = nullptr, *d_hull_vert = nullptr, *d_hfield = nullptr, *d_hfield_mip = nullptr, *d_dbg = nullptr;
  int *d_hull_adr = nullptr, *d_hull_nbr = nullptr;
  float4* d_hull_cell = nullptr;    // support maps of the hulls (cosim_hullmap.h)
  float4* d_hull_cand = nullptr;
  int hullmap_of_geom[64];          // what DevModel::g_hullmap holds while "support_map" is on
  unsigned* d_pairs = nullptr;   // robot-robot candidate pairs (geom1 | geom2 << 16)
  float4* d_gext = nullptr;      // per geom: MPR centre (body frame), raw sliding friction
  std::vector<float> h_params;
  bool params_dirty = true;
  uint64_t seed = 0;
  int64_t env_id0 = 0;
  float tol32 = 1e-6f;
  float ls_scale = 1.f;
  int max_newton = 50;
  int max_ls = 24;
  int nsub_override = 0;
  int pair_coop = 1;
  int timing_stride = 1;   // kernel timing: an event pair around every n-th launch (the events themselves cost ~4 % of a short run at 1)
  unsigned launch_seq = 0;
  int block_cull = 1;
  int coop_walk = 0;
  int pair_boxbox = 1;
  int prio[4] = {3, 0, 2, 4};   // wave priority by solver lag (see the kernel): usual iterations per substep, lag thresholds
  // timing
  bool timing = false;
  std::vector<hipEvent_t> ev;  // event pairs (start, stop) of timed launches not yet read back
  int ev_used = 0;
  double t_accum_ms = 0.0;
  int t_launches = 0;
  void (*launch)(cosim_engine*, const KArgs&, int grid, hipStream_t) = nullptr;
  void (*launch_prof)(cosim_engine*, const KArgs&, int grid, hipStream_t) = nullptr;  // diagnostic build (light_v1 flat only)
  void (*launch2)(cosim_engine*, const KArgs&, int grid, hipStream_t) = nullptr;      // two environments per wave (reset / step)
  void (*launch_prof2)(cosim_engine*, const KArgs&, int grid, hipStream_t) = nullptr;
  void (*launch_ct)(cosim_engine*, const KArgs&, int grid, hipStream_t) = nullptr;       // contact-twist variant of a dense-row kernel
  void (*launch_ct_prof)(cosim_engine*, const KArgs&, int grid, hipStream_t) = nullptr;
  int ct_lds_bytes = 0, ct_contact_slots = 0;
  int epw = 1;   // environments per wave of the reset / step launches
  int lds_bytes = 0;
  int geom_stage = 64;   // plane kernels: geom lanes that can stage their contacts
  int contact_slots = 0, pair_slots = 0;   // ground-contact / robot-robot contact capacity of the selected kernel
  // large-capacity kernel behind the fleet kernel (env_fixup_kernel): redoes the control step of envs whose contacts did not fit
  void (*launch_fix)(cosim_engine*, const KArgs&, int grid, hipStream_t) = nullptr;
  int fix_contact_slots = 0;
  // split pipeline (env_narrow_kernel + env_step_kernel, one pair of launches per substep) where the model / terrain has one
  void (*launch_narrow)(cosim_engine*, const KArgs&, int grid, hipStream_t) = nullptr;
  void (*launch_stepx)(cosim_engine*, const KArgs&, int grid, hipStream_t) = nullptr;
  bool split = false;
  int narrow_waves = 6;
  int narrow_occ = 2;   // narrowphase kernel variant: waves per SIMD its registers are allocated for (0: diagnostic build)
  float *d_xcon = nullptr, *d_xstate = nullptr;
  int* d_xcnt = nullptr;
  int* d_ovf = nullptr;   // [n_envs] flags, set by the fleet kernel, cleared by the fix-up kernel
  // rollout launches (cosim_rollout): K control steps per launch where the variant has such a kernel
  void (*launch_roll)(cosim_engine*, const KArgs&, int grid, hipStream_t) = nullptr;
  void (*launch_roll_fix)(cosim_engine*, const KArgs&, int grid, hipStream_t) = nullptr;
  // range launches: cosim_step issues the fleet as n_ranges launches over contiguous env ranges on engine-owned streams, so that a
  // range's next control step fills the tail of the others' launches (a launch ends with its slowest env)
  int n_ranges = 1;
  bool deferred_join = false;   // true: cosim_step does not make the caller's stream wait for the range streams (cosim_join does)
  bool join_pending = false;
  std::vector<hipStream_t> rstream;
  std::vector<hipEvent_t> rdone;   // one per range: recorded after the range's last launch
  hipEvent_t ev_in = nullptr;      // recorded on the caller's stream, waited on by the range streams: the step's inputs are ready
  std::vector<int> rfirst, rcount;
  // flow control of the range launches: the host stays at most `inflight` control steps ahead of each range (a ring of events per
  // range; cosim_step blocks on the oldest).  Deep queues are slow on this runtime: with the host hundreds of steps ahead the four
  // range chains step at 12.0 M env-steps/s, held to 2 ... 16 steps ahead at 13.6 ... 13.7 M (1: 13.35, 64: 13.3; MI355X, ROCm 7.2).
  // Short runs gain most from a shallow queue (20 timed steps: 2 -> 12.9 M, 4 -> 12.4, 8 -> 12.2, 16 -> 11.4, unbounded 10.9).
  int inflight = 2;   // 0: unbounded
  std::vector<hipEvent_t> ring;   // [n_ranges][inflight]
  long ring_pos = 0;
};

template <int NV, int NB, int RPL, bool HF, int GTM, bool SC, int MCT>
static void launch_t(cosim_engine* e, const KArgs& a, int grid, hipStream_t s) {
  hipLaunchKernelGGL((env_kernel<NV, NB, RPL, HF, GTM, SC, false, 1, MCT>), dim3(grid), dim3(64), 0, s, a);
}
template <int NV, int NB, int RPL, bool HF, int GTM, bool SC, int MCT>
static void launch_prof_t(cosim_engine* e, const KArgs& a, int grid, hipStream_t s) {
  hipLaunchKernelGGL((env_kernel<NV, NB, RPL, HF, GTM, SC, true, 1, MCT>), dim3(grid), dim3(64), 0, s, a);
}
template <int NV, int NB, int RPL, bool HF, int GTM, bool SC, int MCT>
static void launch_fix_t(cosim_engine* e, const KArgs& a, int grid, hipStream_t s) {   // grid = envs of the range
  hipLaunchKernelGGL((env_fixup_kernel<NV, NB, RPL, HF, GTM, SC, MCT>), dim3((grid + 63) / 64), dim3(64), 0, s, a);
}
template <int NV, int NB, int RPL, bool HF, int GTM, bool SC, int MCT>
static void launch_narrow_t(cosim_engine* e, const KArgs& a, int grid, hipStream_t s) {   // grid = envs x waves per env
  if (e->narrow_occ >= 4) hipLaunchKernelGGL((env_narrow_kernel<NV, NB, RPL, HF, GTM, SC, MCT, 4>), dim3(grid), dim3(64), 0, s, a);
  else if (e->narrow_occ == 3) hipLaunchKernelGGL((env_narrow_kernel<NV, NB, RPL, HF, GTM, SC, MCT, 3>), dim3(grid), dim3(64), 0, s, a);
  else if (e->narrow_occ == 2) hipLaunchKernelGGL((env_narrow_kernel<NV, NB, RPL, HF, GTM, SC, MCT, 2>), dim3(grid), dim3(64), 0, s, a);
  else hipLaunchKernelGGL((env_narrow_kernel<NV, NB, RPL, HF, GTM, SC, MCT, 2, true>), dim3(grid), dim3(64), 0, s, a);   // narrow_occupancy 0: diagnostic build
}
template <int NV, int NB, int RPL, bool HF, int GTM, bool SC, int MCT>
static void launch_stepx_t(cosim_engine* e, const KArgs& a, int grid, hipStream_t s) {
  hipLaunchKernelGGL((env_step_kernel<NV, NB, RPL, HF, GTM, SC, MCT>), dim3(grid), dim3(64), 0, s, a);
}
template <int NV, int NB, int RPL, bool HF, int GTM, bool SC, int MCT>
static void launch_roll_t(cosim_engine* e, const KArgs& a, int grid, hipStream_t s) {
  hipLaunchKernelGGL((env_rollout_kernel<NV, NB, RPL, HF, GTM, SC, MCT>), dim3(grid), dim3(64), 0, s, a);
}
template <int NV, int NB, int RPL, bool HF, int GTM, bool SC, int MCT>
static void launch_roll_fix_t(cosim_engine* e, const KArgs& a, int grid, hipStream_t s) {   // grid = envs of the range
  hipLaunchKernelGGL((env_rollout_fix_kernel<NV, NB, RPL, HF, GTM, SC, MCT>), dim3((grid + 63) / 64), dim3(64), 0, s, a);
}
template <int NV, int NB, int GTM>
static void launch2_t(cosim_engine* e, const KArgs& a, int grid, hipStream_t s) {   // grid = number of envs
  hipLaunchKernelGGL((env_kernel<NV, NB, 2, false, GTM, false, false, 2>), dim3((grid + 1) / 2), dim3(64), 0, s, a);
}
template <int NV, int NB, int GTM>
static void launch_prof2_t(cosim_engine* e, const KArgs& a, int grid, hipStream_t s) {
  hipLaunchKernelGGL((env_kernel<NV, NB, 2, false, GTM, false, true, 2>), dim3((grid + 1) / 2), dim3(64), 0, s, a);
}
// MCT_FLAT / MCT_HF / MCT_HFC: ground-contact slots of the contact-twist kernels on a plane / on a heightfield / on a COARSE
// heightfield (cells of 10 cm or more: the reference's rocky_* and slope_* fields have 55 cm cells, a geom lies over a handful of
// prisms, and the slots that stairs with 1 cm cells need would only cost resident waves; 0: dense contact rows)
template <int NV, int NB, int RPL, int GTM, bool SC, int MCT_FLAT, int MCT_HF, int MCT_HFC = MCT_HF>
static void select_t(cosim_engine* e, bool hf, bool coarse = false) {
  using LH = typename KTraits<NV, NB, RPL, true, SC, 1, MCT_HF>::L;
  using LC = typename KTraits<NV, NB, RPL, true, SC, 1, MCT_HFC>::L;
  using LF = typename KTraits<NV, NB, RPL, false, SC, 1, MCT_FLAT>::L;
  if (hf && coarse && MCT_HFC != MCT_HF) {
    e->launch = launch_t<NV, NB, RPL, true, GTM, SC, MCT_HFC>;
    e->lds_bytes = (int)sizeof(LC); e->contact_slots = LC::MC; e->pair_slots = LC::MCP; e->geom_stage = 64;
    return;
  }
  e->launch = hf ? launch_t<NV, NB, RPL, true, GTM, SC, MCT_HF> : launch_t<NV, NB, RPL, false, GTM, SC, MCT_FLAT>;
  e->lds_bytes = hf ? (int)sizeof(LH) : (int)sizeof(LF);
  e->contact_slots = hf ? LH::MC : LF::MC;
  e->pair_slots = hf ? LH::MCP : LF::MCP;
  e->geom_stage = hf ? 64 : LF::NGS;
}

static int round_up(int x, int m) { return (x + m - 1) / m * m; }

static int build_dev_model(cosim_engine* e) {
  const cosim_model_t& m = e->model;
  DevModel& d = e->hm;
  memset(&d, 0, sizeof d);
  d.nq = m.nq; d.nv = m.nv; d.nu = m.nu; d.nbody = m.nbody; d.njnt = m.njnt; d.ngeom = m.ngeom; d.neq = m.neq; d.npair = m.npair;
  d.frame_skip = m.frame_skip; d.iterations = m.iterations; d.ls_iterations = m.ls_iterations;
  d.ground_type = m.ground_type; d.hfield_nrow = m.hfield_nrow; d.hfield_ncol = m.hfield_ncol; d.nhullvert = m.nhullvert;
  d.imu_body = m.imu_bodyid; d.term_mode = m.term_mode; d.nterm_body = m.nterm_body;
  d.timestep = (float)m.timestep; d.tolerance = (float)m.tolerance; d.ls_tolerance = (float)m.ls_tolerance; d.impratio = (float)m.impratio;
  for (int k = 0; k < 3; k++) { d.gravity[k] = (float)m.gravity[k]; d.ground_pos[k] = (float)m.ground_pos[k]; d.imu_pos[k] = (float)m.imu_pos[k]; }
  for (int k = 0; k < 4; k++) { d.hfield_size[k] = (float)m.hfield_size[k]; d.imu_quat[k] = (float)m.imu_quat[k]; }
  d.gyro_cutoff = (float)m.gyro_cutoff; d.vel_cutoff = (float)m.velocimeter_cutoff; d.heightmap_miss = (float)m.heightmap_miss;
  if (m.solver != CS_SOLVER_NEWTON) return fail(COSIM_EINVAL, "only solver=\"Newton\" (the reference models' setting) is implemented");
  if (m.ground_type != CS_GEOM_PLANE && (m.hfield_nrow < 2 || m.hfield_ncol < 2)) return fail(COSIM_EINVAL, "heightfield ground without elevation data");
  if (m.nbody > 32 || m.nv > 32 || m.ngeom > 32 || m.nq > 64) return fail(COSIM_EINVAL, "model exceeds the per-lane record capacity");
  if (m.neq > MAXEQ) return fail(COSIM_EINVAL, "too many equalities");
  if (m.ngeom > 24 || m.nu > m.nv - 6 || m.nq != m.nv + 1) return fail(COSIM_EINVAL, "model exceeds the per-env LDS tables (geoms <= 24, nu <= nv - 6, one free joint)");
  int maxdepth = 0;
  for (int j = 0; j < m.njnt; j++)
    if (m.jnt_type[j] == CS_JNT_HINGE && (m.jnt_pos[j][0] != 0.0 || m.jnt_pos[j][1] != 0.0 || m.jnt_pos[j][2] != 0.0)) d.any_jpos = 1;
  // dof ancestor masks
  unsigned anc[MAXD];
  for (int i = 0; i < m.nv; i++) anc[i] = (1u << i) | (m.dof_parentid[i] >= 0 ? anc[m.dof_parentid[i]] : 0u);
  for (int b = 0; b < m.nbody; b++) {
    LaneRec& r = d.rec[b];
    r.b_parent = m.body_parentid[b];
    int lev = 0;
    for (int p = b; p > 0; p = m.body_parentid[p]) lev++;
    r.b_level = lev;
    if (lev > maxdepth) maxdepth = lev;
    if (m.body_jntnum[b] > 1) return fail(COSIM_EINVAL, "more than one joint per body is not supported");
    r.b_jtype = m.body_jntnum[b] == 1 ? m.jnt_type[m.body_jntadr[b]] : -1;
    r.b_qadr = m.body_jntnum[b] == 1 ? m.jnt_qposadr[m.body_jntadr[b]] : 0;
    r.b_dadr = m.body_jntnum[b] == 1 ? m.jnt_dofadr[m.body_jntadr[b]] : 0;
    int a = b;
    while (a > 0 && m.body_dofnum[a] == 0) a = m.body_parentid[a];
    r.b_lastdof = a > 0 ? m.body_dofadr[a] + m.body_dofnum[a] - 1 : -1;
    r.b_dofmask = r.b_lastdof >= 0 ? anc[r.b_lastdof] : 0u;
    unsigned mask = 0;
    for (int c = 0; c < m.nbody; c++) {
      int q = c;
      while (q > 0 && q != b) q = m.body_parentid[q];
      if (q == b && (b > 0 || c == 0)) mask |= 1u << c;
    }
    r.b_subtree = mask;
    for (int k = 0; k < 3; k++) { r.b_pos[k] = (float)m.body_pos[b][k]; r.b_ipos[k] = (float)m.body_ipos[b][k]; r.b_inertia[k] = (float)m.body_inertia[b][k]; }
    for (int k = 0; k < 4; k++) { r.b_quat[k] = (float)m.body_quat[b][k]; r.b_iquat[k] = (float)m.body_iquat[b][k]; }
    if (m.body_jntnum[b] == 1) {
      int j = m.body_jntadr[b];
      for (int k = 0; k < 3; k++) { r.j_pos[k] = (float)m.jnt_pos[j][k]; r.j_axis[k] = (float)m.jnt_axis[j][k]; }
      r.j_q0 = m.jnt_type[j] == CS_JNT_HINGE ? (float)m.qpos0[m.jnt_qposadr[j]] : 0.f;
      r.j_limited = m.jnt_limited[j];
      r.j_margin = (float)m.jnt_margin[j];
      for (int k = 0; k < 2; k++) { r.j_range[k] = (float)m.jnt_range[j][k]; r.j_solref[k] = (float)m.jnt_solref[j][k]; }
      for (int k = 0; k < 5; k++) r.j_solimp[k] = (float)m.jnt_solimp[j][k];
    }
  }
  d.maxdepth = maxdepth;
  d.imu_dofmask = d.rec[m.imu_bodyid].b_dofmask;
  int nfric = 0;
  for (int i = 0; i < m.nv; i++) {
    LaneRec& r = d.rec[i];
    r.d_body = m.dof_bodyid[i]; r.d_parent = m.dof_parentid[i]; r.d_ancmask = anc[i];
    r.d_armature = (float)m.dof_armature[i]; r.d_damping = (float)m.dof_damping[i];
    for (int k = 0; k < 2; k++) r.d_solref[k] = (float)m.dof_solref[i][k];
    for (int k = 0; k < 5; k++) r.d_solimp[k] = (float)m.dof_solimp[i][k];
    int j = m.dof_jntid[i];
    r.d_frclimited = m.jnt_type[j] == CS_JNT_HINGE ? m.jnt_actfrclimited[j] : 0;
    r.d_frcrange[0] = (float)m.jnt_actfrcrange[j][0]; r.d_frcrange[1] = (float)m.jnt_actfrcrange[j][1];
    r.d_act = -1;
    if (m.dof_frictionloss[i] > 0) d.rec[nfric++].d_fric = i;
  }
  d.nfric = nfric;
  for (int g = 0; g < m.ngeom; g++) {
    LaneRec& r = d.rec[g];
    r.g_type = m.geom_type[g]; r.g_body = m.geom_bodyid[g]; r.g_ground = m.geom_ground[g];
    r.g_hulladr = m.geom_hulladr[g]; r.g_hullnum = m.geom_hullnum[g];
    int condim = m.geom_condim[g] > m.ground_condim ? m.geom_condim[g] : m.ground_condim;
    if (m.geom_ground[g] && condim != 3) return fail(COSIM_EINVAL, "only condim 3 contacts are implemented");
    for (int k = 0; k < 3; k++) { r.g_pos[k] = (float)m.geom_pos[g][k]; r.g_size[k] = (float)m.geom_size[g][k]; r.g_rcenter[k] = (float)m.geom_rcenter[g][k]; }
    for (int k = 0; k < 4; k++) r.g_quat[k] = (float)m.geom_quat[g][k];
    r.g_rbound = (float)m.geom_rbound[g];
    for (int k = 0; k < 3; k++) r.g_half[k] = (float)m.geom_aabb[g][3 + k];
    for (int k = 0; k < 3; k++) if (fabs(m.geom_aabb[g][k] - m.geom_rcenter[g][k]) > 1e-12) return fail(COSIM_EINVAL, "geom_aabb centre must equal geom_rcenter");
    // mj_contactParam with equal priorities: solmix-weighted solref/solimp, margins by max
    double s1 = m.ground_solmix, s2 = m.geom_solmix[g], mix;
    if (s1 >= 1e-15 && s2 >= 1e-15) mix = s1 / (s1 + s2);
    else if (s1 < 1e-15 && s2 < 1e-15) mix = 0.5;
    else mix = s1 < 1e-15 ? 0.0 : 1.0;
    for (int k = 0; k < 2; k++)
      r.g_solref[k] = (m.ground_solref[0] > 0 && m.geom_solref[g][0] > 0)
                          ? (float)(mix * m.ground_solref[k] + (1 - mix) * m.geom_solref[g][k])
                          : (float)fmin(m.ground_solref[k], m.geom_solref[g][k]);
    for (int k = 0; k < 5; k++) r.g_solimp[k] = (float)(mix * m.ground_solimp[k] + (1 - mix) * m.geom_solimp[g][k]);
    double margin = fmax(m.ground_margin, m.geom_margin[g]), gap = fmax(m.ground_gap, m.geom_gap[g]);
    r.g_margin = (float)margin;
    r.g_incmargin = (float)(margin - gap);
  }
  for (int q = 0; q < m.neq; q++) {
    LaneRec& r = d.rec[q];
    r.e_body1 = m.eq_body1[q]; r.e_body2 = m.eq_body2[q];
    for (int k = 0; k < 3; k++) { r.e_anchor1[k] = (float)m.eq_anchor1[q][k]; r.e_anchor2[k] = (float)m.eq_anchor2[q][k]; }
    for (int k = 0; k < 2; k++) r.e_solref[k] = (float)m.eq_solref[q][k];
    for (int k = 0; k < 5; k++) r.e_solimp[k] = (float)m.eq_solimp[q][k];
  }
  // solref -> (K, B) of mj_makeImpedance once on the host (they depend on solref, solimp[1] and the timestep only): the
  // *_solref slots of the device records carry K and B from here on
  {
    auto kb = [&](float* solref, const float* solimp) {
      const double dmax = fmin(0.9999, fmax(0.0001, (double)solimp[1]));
      double K, B;
      if (solref[0] > 0.f) {
        const double tc = fmax((double)solref[0], 2.0 * m.timestep), dr = solref[1];   // refsafe
        K = 1.0 / fmax(1e-15, dmax * dmax * tc * tc * dr * dr);
        B = 2.0 / fmax(1e-15, dmax * tc);
      } else { K = -(double)solref[0] / fmax(1e-15, dmax * dmax); B = -(double)solref[1] / fmax(1e-15, dmax); }
      solref[0] = (float)K; solref[1] = (float)B;
    };
    for (int b = 0; b < m.nbody; b++) if (m.body_jntnum[b] == 1) kb(d.rec[b].j_solref, d.rec[b].j_solimp);
    for (int i = 0; i < m.nv; i++) kb(d.rec[i].d_solref, d.rec[i].d_solimp);
    for (int g = 0; g < m.ngeom; g++) kb(d.rec[g].g_solref, d.rec[g].g_solimp);
    for (int q = 0; q < m.neq; q++) kb(d.rec[q].e_solref, d.rec[q].e_solimp);
  }
  for (int u = 0; u < m.nu; u++) {
    LaneRec& r = d.rec[u];
    r.a_dof = m.act_dofid[u]; r.a_ctrllimited = m.act_ctrllimited[u]; r.a_gear = (float)m.act_gear[u];
    r.a_ctrlrange[0] = (float)m.act_ctrlrange[u][0]; r.a_ctrlrange[1] = (float)m.act_ctrlrange[u][1];
    if (d.rec[m.act_dofid[u]].d_act >= 0) return fail(COSIM_EINVAL, "two motors on one dof are not supported");
    d.rec[m.act_dofid[u]].d_act = u;
    r.a_velmode = m.ctl_velmode[u]; r.a_qadr = m.ctl_qadr[u]; r.a_dadr = m.ctl_dadr[u];
    r.a_scale = (float)m.ctl_scale[u]; r.a_cgear = (float)m.ctl_gear[u]; r.a_gamma = (float)m.ctl_gamma[u];
    r.a_maxtq = (float)m.ctl_maxtq[u];
  }
  d.nobs_pos = m.nobs_pos; d.nobs_vel = m.nobs_vel; d.ninfo_state = m.ninfo_state; d.init_noise_nq = m.init_noise_nq;
  for (int i = 0; i < CS_MAXOBSJ; i++) { d.rec[i].o_qadr = m.obs_qadr[i]; d.rec[i].o_dadr = m.obs_dadr[i]; d.rec[i].o_qgear = (float)m.obs_qgear[i]; d.rec[i].o_dgear = (float)m.obs_dgear[i]; }
  for (int i = 0; i < CS_MAXINFOSTATE; i++) { d.rec[i].i_kind = m.info_kind[i]; d.rec[i].i_adr = m.info_adr[i]; d.rec[i].i_gear = (float)m.info_gear[i]; }
  for (int i = 0; i < CS_MAXQ; i++) { d.rec[i].n_qadr = m.init_noise_qadr[i]; d.rec[i].init_qpos = (float)m.init_qpos[i]; }
  for (int i = 0; i < CS_MAXBODY; i++) d.rec[i].t_body = m.term_body[i];
  for (int i = 0; i < m.nterm_body; i++) d.term_bodymask |= 1u << m.term_body[i];
  d.ntri = m.nv * (m.nv + 1) / 2;
  for (int r = 0, e2 = 0; r < m.nv; r++)
    for (int c = 0; c <= r; c++, e2++) { d.tri_row[e2] = (unsigned char)r; d.tri_col[e2] = (unsigned char)c; }
  return COSIM_OK;
}

static int build_dev_obs(cosim_engine* e) {
  const cosim_obs_config_t& c = e->obs_cfg;
  DevObs& o = e->ho;
  memset(&o, 0, sizeof o);
  if (c.stack_size < 1 || c.command_dim < 0 || c.command_dim > CS_MAXCMD) return fail(COSIM_EINVAL, "bad stack_size / command_dim");
  if (c.n_stacked < 0 || c.n_stacked > CS_MAXFIELD || c.n_non_stacked < 0 || c.n_non_stacked > CS_MAXFIELD) return fail(COSIM_EINVAL, "bad field lists");
  o.stack_size = c.stack_size; o.command_dim = c.command_dim; o.position_command = c.position_command;
  o.max_sim_step = c.max_sim_step; o.auto_reset = c.auto_reset; o.noise_enabled = c.noise_enabled;
  o.action_delay_prob = c.action_delay_prob; o.init_noise = c.init_noise;
  for (int i = 0; i < CS_MAXCMD; i++) o.command_scales[i] = c.command_scales[i];
  o.hm_res_x = c.hm_res_x; o.hm_res_y = c.hm_res_y; o.hm_size_x = c.hm_size_x; o.hm_size_y = c.hm_size_y;
  for (int f = 0; f < 8; f++) {
    o.noise_mean[f] = c.noise_mean[f]; o.noise_std[f] = c.noise_std[f]; o.noise_lower[f] = c.noise_lower[f]; o.noise_upper[f] = c.noise_upper[f];
    const double sd = c.noise_std[f] > 0 ? c.noise_std[f] : 1.0;
    o.noise_ca[f] = (float)(0.5 * erfc(-((double)c.noise_lower[f] - c.noise_mean[f]) / sd / sqrt(2.0)));
    o.noise_cb[f] = (float)(0.5 * erfc(-((double)c.noise_upper[f] - c.noise_mean[f]) / sd / sqrt(2.0)));
  }
  int el = 0;
  for (int pass = 0; pass < 2; pass++) {
    const int* list = pass ? c.non_stacked_field : c.stacked_field;
    int n = pass ? c.n_non_stacked : c.n_stacked;
    for (int i = 0; i < n; i++) {
      int f = list[i];
      if (f < 0 || f > 7) return fail(COSIM_EINVAL, "unknown observation field id");
      if (f == CS_OBS_HEIGHT_MAP && (e->model.ground_type != CS_GEOM_HFIELD || c.hm_res_x * c.hm_res_y != c.field_dim[f] || c.field_dim[f] < 1))
        return fail(COSIM_EINVAL, "height_map observation needs a heightfield terrain (mj_rayHfield on a plane is an error in the reference too) and res_x * res_y elements");
      int dim = c.field_dim[f];
      if (c.field_interval[f] < 1 && f != CS_OBS_COMMAND) return fail(COSIM_EINVAL, "observation interval must be >= 1");
      for (int k = 0; k < dim; k++) {
        if (el >= MAXFRAME) return fail(COSIM_EINVAL, "observation frame too large");
        o.el_field[el] = (unsigned char)f; o.el_index[el] = (unsigned short)k;
        o.el_interval[el] = (unsigned char)(f == CS_OBS_COMMAND ? 1 : (c.field_interval[f] > 255 ? 255 : c.field_interval[f]));
        o.el_scale[el] = c.field_scale[f];
        el++;
      }
    }
    if (pass == 0) o.stacked_dim = el;
  }
  o.frame_dim = el;
  o.non_stacked_dim = el - o.stacked_dim;
  o.state_dim = o.stack_size * o.stacked_dim + o.non_stacked_dim;
  o.info_dim = 4 + 2 * e->model.nu + e->model.ninfo_state;
  return COSIM_OK;
}

static void build_layout(cosim_engine* e) {
  const cosim_model_t& m = e->model;
  Layout& L = e->lay;
  int o = 0;
  L.s_qpos = o; o += m.nq;
  L.s_qvel = o; o += m.nv;
  L.s_warm = o; o += m.nv;
  L.s_delay = o; o += m.nu;
  L.s_lastact = o; o += m.nu;
  L.s_meta = o; o += Layout::NMETA;
  L.s_cache = o; o += e->ho.frame_dim;
  L.s_stack = o; o += e->ho.stack_size * e->ho.stacked_dim;
  L.s_stride = round_up(o, 32);
  int p = 0;
  L.p_mass = p; p += m.nbody;
  L.p_binvw = p; p += m.nbody;
  L.p_dinvw = p; p += m.nv;
  L.p_floss = p; p += m.nv;
  L.p_gmu = p; p += m.ngeom;
  L.p_kp = p; p += m.nu;
  L.p_kd = p; p += m.nu;
  L.p_mean = p; p += 1;
  L.p_stride = round_up(p, 32);
}

static void default_params(cosim_engine* e) {
  const cosim_model_t& m = e->model;
  const Layout& L = e->lay;
  e->h_params.assign((size_t)e->n_envs * L.p_stride, 0.f);
  for (int n = 0; n < e->n_envs; n++) {
    float* p = e->h_params.data() + (size_t)n * L.p_stride;
    for (int b = 0; b < m.nbody; b++) { p[L.p_mass + b] = (float)m.body_mass[b]; p[L.p_binvw + b] = (float)m.body_invweight0[b][0]; }
    for (int i = 0; i < m.nv; i++) { p[L.p_dinvw + i] = (float)m.dof_invweight0[i]; p[L.p_floss + i] = (float)m.dof_frictionloss[i]; }
    for (int g = 0; g < m.ngeom; g++) p[L.p_gmu + g] = (float)fmax(1e-5, fmax(m.ground_friction[0], m.geom_friction[g][0]));
    for (int u = 0; u < m.nu; u++) { p[L.p_kp + u] = (float)m.ctl_kp[u]; p[L.p_kd + u] = (float)m.ctl_kd[u]; }
    p[L.p_mean] = (float)m.meaninertia;
  }
  e->params_dirty = true;
}

static int upload_params(cosim_engine* e) {
  if (!e->params_dirty) return COSIM_OK;
  HIP_TRY(hipMemcpy(e->d_params, e->h_params.data(), e->h_params.size() * sizeof(float), hipMemcpyHostToDevice));
  e->params_dirty = false;
  return COSIM_OK;
}

// n contiguous ranges of n_envs / n envs (the first n_envs % n one longer; even sizes for the two-envs-per-wave kernel), each with
// a non-blocking stream of its own and a "done" event
static int set_ranges(cosim_engine* e, int n) {
  if (n < 1 || n > 16 || n > e->n_envs) return fail(COSIM_EINVAL, "cosim_set_param: ranges must be 1..16 and at most n_envs");
  HIP_TRY(hipSetDevice(e->device));
  for (hipStream_t x : e->rstream) { HIP_TRY(hipStreamSynchronize(x)); HIP_TRY(hipStreamDestroy(x)); }
  for (hipEvent_t x : e->rdone) HIP_TRY(hipEventDestroy(x));
  e->rstream.clear(); e->rdone.clear(); e->rfirst.clear(); e->rcount.clear();
  e->join_pending = false;
  e->n_ranges = n;
  for (hipEvent_t x : e->ring) HIP_TRY(hipEventDestroy(x));
  e->ring.clear();
  e->ring_pos = 0;
  for (int i = 0; i < n * e->inflight; i++) { hipEvent_t ev; HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); e->ring.push_back(ev); }
  if (n == 1) return COSIM_OK;
  if (!e->ev_in) HIP_TRY(hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming));
  const int unit = e->epw == 2 ? 2 : 1, units = e->n_envs / unit;
  int first = 0;
  for (int i = 0; i < n; i++) {
    int cnt = (units / n + (i < units % n ? 1 : 0)) * unit;
    if (i == n - 1) cnt = e->n_envs - first;
    hipStream_t st; hipEvent_t ev;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    e->rstream.push_back(st); e->rdone.push_back(ev); e->rfirst.push_back(first); e->rcount.push_back(cnt);
    first += cnt;
  }
  return COSIM_OK;
}

// `stream` waits for everything the range streams have been given so far: the "done" events are recorded here, at join time (an
// event marks everything enqueued before it), not after every range launch -- a deferred-join caller pays no marker packets per step
static int join_ranges(cosim_engine* e, hipStream_t stream) {
  if (!e->join_pending) return COSIM_OK;
  for (int i = 0; i < e->n_ranges; i++) {
    HIP_TRY(hipEventRecord(e->rdone[i], e->rstream[i]));
    HIP_TRY(hipStreamWaitEvent(stream, e->rdone[i], 0));
  }
  e->join_pending = false;
  return COSIM_OK;
}

extern "C" {

// Fused actor MLP (cosim_mlp.hip): out = clip(act_L(... act_1(x W_1^T + b_1) ...)).  All pointers are device pointers; dims has
// n_layers + 1 entries; act / act_alpha one entry per layer (0 none, 1 relu, 2 tanh, 3 elu, 4 sigmoid, 5 leaky relu).
int cosim_mlp_forward(const float* x_dev, int n, int n_layers, const int* dims, const float* const* w_dev, const float* const* b_dev,
                      const int* act, const float* act_alpha, float clip, float* out_dev, void* stream) {
  if (!x_dev || !dims || !w_dev || !act || !out_dev || n <= 0) return fail(COSIM_EINVAL, "cosim_mlp_forward: bad argument");
  if (n_layers < 1 || n_layers > MLP_MAXL) return fail(COSIM_EINVAL, "cosim_mlp_forward: 1..6 layers");
  MlpArgs a;
  memset(&a, 0, sizeof a);
  int maxd = 0;
  for (int l = 0; l <= n_layers; l++) {
    if (dims[l] < 1 || dims[l] > MLP_MAXD) return fail(COSIM_EINVAL, "cosim_mlp_forward: layer width outside 1..512");
    a.dims[l] = dims[l];
    if (dims[l] > maxd) maxd = dims[l];
  }
  for (int l = 0; l < n_layers; l++) {
    if (!w_dev[l]) return fail(COSIM_EINVAL, "cosim_mlp_forward: null weight");
    a.w[l] = w_dev[l]; a.b[l] = b_dev ? b_dev[l] : nullptr; a.act[l] = act[l]; a.act_alpha[l] = act_alpha ? act_alpha[l] : 1.f;
  }
  a.x = x_dev; a.out = out_dev; a.nl = n_layers; a.n = n; a.clip = clip; a.ld = maxd | 1;
  const size_t lds = (size_t)2 * 32 * a.ld * sizeof(float);
  static size_t lds_allowed[64] = {0};   // per device: the attribute belongs to the function on the device it was set on
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return fail(COSIM_EINVAL, "cosim_mlp_forward: device index out of range");
  if (lds > lds_allowed[dev]) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_forward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_allowed[dev] = lds;
  }
  hipLaunchKernelGGL(mlp_forward_kernel, dim3((n + 31) / 32), dim3(256), lds, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return COSIM_OK;
}

int cosim_lstm_cell(const float* x_dev, const float* h_dev, const float* c_dev, int n, int in_dim, int hidden, const float* w_dev,
                    const float* r_dev, const float* b_dev, float* h_out_dev, float* c_out_dev, void* stream) {
  if (!x_dev || !h_dev || !c_dev || !w_dev || !r_dev || !h_out_dev || !c_out_dev || n <= 0) return fail(COSIM_EINVAL, "cosim_lstm_cell: bad argument");
  if (in_dim < 1 || hidden < 1 || in_dim + hidden > 1200) return fail(COSIM_EINVAL, "cosim_lstm_cell: in_dim + hidden outside 2..1200");
  LstmArgs a;
  a.x = x_dev; a.h = h_dev; a.c = c_dev; a.W = w_dev; a.R = r_dev; a.B = b_dev; a.h_out = h_out_dev; a.c_out = c_out_dev;
  a.n = n; a.I = in_dim; a.H = hidden; a.ld = (in_dim + hidden) | 1;
  const size_t lds = (size_t)32 * a.ld * sizeof(float);
  static size_t lds_allowed[64] = {0};   // per device
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return fail(COSIM_EINVAL, "cosim_lstm_cell: device index out of range");
  if (lds > lds_allowed[dev]) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_cell_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_allowed[dev] = lds;
  }
  hipLaunchKernelGGL(lstm_cell_kernel, dim3((n + 31) / 32), dim3(256), lds, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return COSIM_OK;
}

int cosim_fleet_stats(const float* info_dev, int n, int info_dim, int nu, const float* cmd_dev, int cmd_stride, int ncmd, double* acc_dev,
                      void* stream) {
  if (!info_dev || !acc_dev || n <= 0 || nu < 0 || ncmd < 0 || ncmd > 3 || 4 + nu + ncmd > 32 || info_dim < 4 + nu || (ncmd > 0 && !cmd_dev))
    return fail(COSIM_EINVAL, "cosim_fleet_stats: bad argument");
  const int blocks = n >= 8 * 64 ? 64 : (n + 7) / 8;
  hipLaunchKernelGGL(fleet_stats_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, info_dev, n, info_dim, nu, cmd_dev, cmd_stride, ncmd,
                     acc_dev);
  HIP_TRY(hipGetLastError());
  return COSIM_OK;
}

int cosim_fleet_hist(const float* info_dev, int n, int info_dim, int nu, const float* cmd_dev, int cmd_stride, int ncmd, const float* hi_dev,
                     int nbins, double* hist_dev, void* stream) {
  if (!info_dev || !hi_dev || !hist_dev || n <= 0 || nu < 0 || ncmd < 0 || ncmd > 3 || 4 + nu + ncmd > 32 || info_dim < 4 + nu || nbins < 2 ||
      (ncmd > 0 && !cmd_dev))
    return fail(COSIM_EINVAL, "cosim_fleet_hist: bad argument");
  const int blocks = n >= 8 * 64 ? 64 : (n + 7) / 8;
  hipLaunchKernelGGL(fleet_hist_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, info_dev, n, info_dim, nu, cmd_dev, cmd_stride, ncmd,
                     hi_dev, nbins, hist_dev);
  HIP_TRY(hipGetLastError());
  return COSIM_OK;
}

const char* cosim_last_error(void) { return g_err.c_str(); }
int cosim_model_sizeof(void) { return (int)sizeof(cosim_model_t); }
int cosim_obs_config_sizeof(void) { return (int)sizeof(cosim_obs_config_t); }

int cosim_create(const cosim_model_t* model, const float* hull_vert, const int* hull_adr, const int* hull_nbr, const float* hfield,
                 const cosim_obs_config_t* obs, int n_envs, int device, uint64_t seed, int64_t env_id0, cosim_engine_t** out) {
  if (!model || !obs || !out || n_envs < 1) return fail(COSIM_EINVAL, "cosim_create: null argument or n_envs < 1");
  if (model->magic != CS_MODEL_MAGIC || model->magic_end != CS_MODEL_MAGIC) return fail(COSIM_EINVAL, "cosim_create: model blob magic mismatch (layout drift?)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(COSIM_ENOGPU, "cosim_create: no HIP device available");
  if (device < 0 || device >= ndev) return fail(COSIM_EINVAL, "cosim_create: bad device index");
  HIP_TRY(hipSetDevice(device));
  cosim_engine* e = new cosim_engine();
  e->n_envs = n_envs; e->device = device; e->model = *model; e->obs_cfg = *obs; e->seed = seed; e->env_id0 = env_id0;
  int rc = build_dev_model(e);
  if (rc == COSIM_OK) rc = build_dev_obs(e);
  if (rc != COSIM_OK) { delete e; return rc; }
  build_layout(e);
  const int nv = model->nv, nb = model->nbody;
  // kernel instantiations: (nv, nbody) of the four cosim robots; RPL = constraint rows per lane
  const bool hf = model->ground_type == CS_GEOM_HFIELD;
  // coarse field: both cell edges at least 10 cm (see select_t)
  const bool coarse = hf && model->hfield_ncol > 1 && model->hfield_nrow > 1 && 2.0 * model->hfield_size[0] / (model->hfield_ncol - 1) >= 0.1 &&
                      2.0 * model->hfield_size[1] / (model->hfield_nrow - 1) >= 0.1;
  int gtm = 0;   // geom types present: the kernel is specialised on them (bit 0 sphere, 1 cylinder, 2 box, 3 mesh)
  std::vector<char> in_pair(model->ngeom > 0 ? model->ngeom : 1, 0);
  for (int p = 0; p < model->npair; p++) { in_pair[model->pair_geom1[p]] = 1; in_pair[model->pair_geom2[p]] = 1; }
  for (int g = 0; g < model->ngeom; g++) {
    if (!model->geom_ground[g] && !in_pair[g]) continue;
    switch (model->geom_type[g]) {
      case CS_GEOM_SPHERE: gtm |= GT_SPHERE; break;
      case CS_GEOM_CYLINDER: gtm |= GT_CYLINDER; break;
      case CS_GEOM_BOX: gtm |= GT_BOX; break;
      case CS_GEOM_MESH: gtm |= GT_MESH; break;
      default: delete e; return fail(COSIM_EINVAL, "cosim_create: collision geom type not implemented in the HIP engine");
    }
  }
  constexpr int G_LIGHT = GT_SPHERE | GT_CYLINDER | GT_MESH, G_MESH = GT_MESH, G_HUM = GT_BOX | GT_CYLINDER | GT_MESH;
  if (nv == 18 && nb <= 14 && (gtm & ~G_LIGHT) == 0) {   // flamingo_light_v1
    select_t<18, 14, 1, G_LIGHT, false, 0, 128, 48>(e, hf, coarse);
    if (!hf) {
      e->launch_prof = launch_prof_t<18, 14, 1, false, G_LIGHT, false, 0>; e->launch2 = launch2_t<18, 14, G_LIGHT>; e->launch_prof2 = launch_prof2_t<18, 14, G_LIGHT>;
      // the same robot on the plane with its ground contacts in twist space (32 slots instead of 12; cosim_set_param "contact_twist")
      e->launch_ct = launch_t<18, 14, 1, false, G_LIGHT, false, 32>;
      e->launch_ct_prof = launch_prof_t<18, 14, 1, false, G_LIGHT, false, 32>;
      e->ct_lds_bytes = (int)sizeof(typename KTraits<18, 14, 1, false, false, 1, 32>::L);
      e->ct_contact_slots = 32;
      // ... and with 40 slots (four per ground geom at most: 7 hulls, 2 cylinders, 4 spheres -> 40 is the most the plane narrowphase
      // can emit) as the kernel that redoes the rare control step with more than 14 contacts: nothing is ever left out
      e->launch_fix = launch_fix_t<18, 14, 1, false, G_LIGHT, false, 40>;
      e->fix_contact_slots = 40;
      e->launch_roll = launch_roll_t<18, 14, 1, false, G_LIGHT, false, 0>;
      e->launch_roll_fix = launch_roll_fix_t<18, 14, 1, false, G_LIGHT, false, 40>;
    }
  }
  else if (nv == 14 && nb <= 10 && (gtm & ~G_MESH) == 0) {   // flamingo_p_v3
    // one row per lane: with the ground contacts in twist space the dense rows are the robot-robot contacts (8 slots = 32 rows), 8
    // frictionloss rows and at most 8 limit rows (8 hinges)
    // (no coarse-cell variant: at one row per lane the 64-slot kernel already fits the 12 waves per CU its registers allow)
    select_t<14, 10, 1, G_MESH, true, 32, 64, 64>(e, hf, coarse);
    if (!hf) {
      // Plane: dense rows, like flamingo_light_v1.  With the friction-loss and limit rows in their dofs' lanes all 64 slots are contact
      // rows: 16 contacts, ground and robot-robot together (most seen in the bench: 16), and the dense solver iteration is cheaper
      // than the contact-twist one at these counts (kernel 0.325 ms against 0.400 per 1024 envs).  A control step with more is redone
      // by the contact-twist kernel (32 ground slots = four per geom, the narrowphase's maximum, + 8 pair slots) right behind it;
      // cosim_set_param "contact_twist" 1 makes that kernel the fleet kernel, as in round 2.
      e->launch_ct = e->launch; e->launch_ct_prof = launch_prof_t<14, 10, 1, false, G_MESH, true, 32>;
      e->ct_lds_bytes = e->lds_bytes; e->ct_contact_slots = e->contact_slots;
      using LD_ = typename KTraits<14, 10, 1, false, true, 1, 0>::L;
      e->launch = launch_t<14, 10, 1, false, G_MESH, true, 0>;
      e->launch_prof = launch_prof_t<14, 10, 1, false, G_MESH, true, 0>;
      e->lds_bytes = (int)sizeof(LD_); e->contact_slots = LD_::MC; e->pair_slots = 0;
      e->launch_fix = launch_fix_t<14, 10, 1, false, G_MESH, true, 32>;
      e->fix_contact_slots = 32;
      e->launch_roll = launch_roll_t<14, 10, 1, false, G_MESH, true, 0>;
      e->launch_roll_fix = launch_roll_fix_t<14, 10, 1, false, G_MESH, true, 32>;
    }
  }
  else if (nv == 22 && nb <= 18 && (gtm & ~G_MESH) == 0) {   // w4_p_v2
    // plane: at most 4 contacts per geom (17 geoms); 80 slots keep the env at 19 KB of LDS = the 8 waves per CU its 256 registers allow
    select_t<22, 18, 2, G_MESH, true, 80, 128, 48>(e, hf, coarse);
    if (hf && coarse) e->launch_prof = launch_prof_t<22, 18, 2, true, G_MESH, true, 48>;   // diagnostic build of the config-3 kernel
  }
  else if (nv == 29 && nb <= 26 && (gtm & ~G_HUM) == 0) {     // humanoid_p_v0
    // plane: at most 4 contacts per geom (22 geoms); 96 slots = 6 waves per CU instead of 5
    select_t<29, 26, 2, G_HUM, true, 96, 256>(e, hf);
    if (hf) {
      e->launch_prof = launch_prof_t<29, 26, 2, true, G_HUM, true, 256>;
      // the prism walk in a kernel of its own, several waves per env (default; cosim_set_param "split" 0 goes back to the fused kernel)
      e->launch_narrow = launch_narrow_t<29, 26, 2, true, G_HUM, true, 256>;
      e->launch_stepx = launch_stepx_t<29, 26, 2, true, G_HUM, true, 256>;
      e->split = true;
    }
  }
  else { delete e; return fail(COSIM_EINVAL, "cosim_create: no kernel instantiation for this (nv, nbody); add one in cosim_engine.hip"); }
  if (nv != 18 && model->neq > 0) { delete e; return fail(COSIM_EINVAL, "cosim_create: this robot's kernels keep no rows for connect equalities"); }
  if (model->ngeom > e->geom_stage) { delete e; return fail(COSIM_EINVAL, "cosim_create: more collision geoms than the plane kernel stages contacts for"); }
  {
    // support maps of the mesh geoms' hulls (geoms that share a hull slice share the map)
    std::vector<float> cells, cand;
    for (int g = 0; g < 64; g++) e->hullmap_of_geom[g] = -1;
    for (int g = 0; g < model->ngeom; g++) {
      if (model->geom_type[g] != CS_GEOM_MESH || model->geom_hullnum[g] < HM_MIN_VERTS || !hull_vert || !hull_adr || !hull_nbr) continue;
      for (int h = 0; h < g; h++)
        if (e->hullmap_of_geom[h] >= 0 && model->geom_hulladr[h] == model->geom_hulladr[g] && model->geom_hullnum[h] == model->geom_hullnum[g])
          e->hullmap_of_geom[g] = e->hullmap_of_geom[h];
      if (e->hullmap_of_geom[g] >= 0) continue;
      const int adr = model->geom_hulladr[g], num = model->geom_hullnum[g];
      if (adr < 0 || adr + num > model->nhullvert) { delete e; return fail(COSIM_EINVAL, "cosim_create: geom hull slice outside the hull vertex array"); }
      for (int v = adr; v < adr + num; v++)
        for (int k = hull_adr[v]; k < hull_adr[v + 1]; k++)
          if (k < 0 || k >= model->nhulledge || hull_nbr[k] < 0 || hull_nbr[k] >= num) { delete e; return fail(COSIM_EINVAL, "cosim_create: hull neighbour graph out of range"); }
      e->hullmap_of_geom[g] = (int)(cells.size() / (4 * HM_REC));
      build_support_map(hull_vert + 3 * (size_t)adr, num, hull_adr + adr, hull_nbr, cells, cand);
    }
    for (int g = 0; g < 64; g++) e->hm.g_hullmap[g] = e->hullmap_of_geom[g];
    if (cells.empty()) cells.assign(4 * HM_REC, 0.f);
    if (cand.empty()) cand.assign(4, 0.f);
    HIP_TRY(hipMalloc(&e->d_hull_cell, cells.size() * sizeof(float)));
    HIP_TRY(hipMalloc(&e->d_hull_cand, cand.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(e->d_hull_cell, cells.data(), cells.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->d_hull_cand, cand.data(), cand.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  HIP_TRY(hipMalloc(&e->d_model, sizeof(DevModel)));
  HIP_TRY(hipMalloc(&e->d_obs, sizeof(DevObs)));
  HIP_TRY(hipMemcpy(e->d_model, &e->hm, sizeof(DevModel), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->d_obs, &e->ho, sizeof(DevObs), hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&e->d_state, (size_t)n_envs * e->lay.s_stride * sizeof(float)));
  HIP_TRY(hipMemset(e->d_state, 0, (size_t)n_envs * e->lay.s_stride * sizeof(float)));
  HIP_TRY(hipMalloc(&e->d_params, (size_t)n_envs * e->lay.p_stride * sizeof(float)));
  HIP_TRY(hipMalloc(&e->d_dbg, 8192 * sizeof(float)));
  if (e->launch_stepx) {
    HIP_TRY(hipMalloc(&e->d_xcon, (size_t)n_envs * XG * XC * 8 * sizeof(float)));
    HIP_TRY(hipMalloc(&e->d_xcnt, (size_t)n_envs * XG * sizeof(int)));
    HIP_TRY(hipMalloc(&e->d_xstate, (size_t)n_envs * XS * sizeof(float)));
    HIP_TRY(hipMemset(e->d_xcnt, 0, (size_t)n_envs * XG * sizeof(int)));
    HIP_TRY(hipMemset(e->d_xstate, 0, (size_t)n_envs * XS * sizeof(float)));
  }
  HIP_TRY(hipMalloc(&e->d_ovf, (size_t)n_envs * sizeof(int)));
  HIP_TRY(hipMemset(e->d_ovf, 0, (size_t)n_envs * sizeof(int)));
  int nhv = model->nhullvert > 0 ? model->nhullvert : 1, nhe = model->nhulledge > 0 ? model->nhulledge : 1;
  HIP_TRY(hipMalloc(&e->d_hull_vert, (size_t)nhv * 4 * sizeof(float)));   // 16 bytes per vertex on the device: one load each
  HIP_TRY(hipMalloc(&e->d_hull_adr, (size_t)(nhv + 1) * sizeof(int)));
  HIP_TRY(hipMalloc(&e->d_hull_nbr, (size_t)nhe * sizeof(int)));
  if (model->nhullvert > 0) {
    if (!hull_vert || !hull_adr || !hull_nbr) return fail(COSIM_EINVAL, "cosim_create: model has mesh geoms but no hull arrays were passed");
    std::vector<float> hv4((size_t)model->nhullvert * 4, 0.f);
    for (int i = 0; i < model->nhullvert; i++) for (int k = 0; k < 3; k++) hv4[4 * (size_t)i + k] = hull_vert[3 * (size_t)i + k];
    HIP_TRY(hipMemcpy(e->d_hull_vert, hv4.data(), hv4.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->d_hull_adr, hull_adr, (size_t)(model->nhullvert + 1) * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->d_hull_nbr, hull_nbr, (size_t)model->nhulledge * sizeof(int), hipMemcpyHostToDevice));
  }
  {
    std::vector<unsigned> hp(model->npair > 0 ? model->npair : 1, 0u);
    for (int p = 0; p < model->npair; p++) hp[p] = (unsigned)model->pair_geom1[p] | ((unsigned)model->pair_geom2[p] << 16);
    std::vector<float4> hg(model->ngeom > 0 ? model->ngeom : 1);
    for (int g = 0; g < model->ngeom; g++)
      hg[g] = make_float4((float)model->geom_center[g][0], (float)model->geom_center[g][1], (float)model->geom_center[g][2], (float)model->geom_friction[g][0]);
    HIP_TRY(hipMalloc(&e->d_pairs, hp.size() * sizeof(unsigned)));
    HIP_TRY(hipMalloc(&e->d_gext, hg.size() * sizeof(float4)));
    HIP_TRY(hipMemcpy(e->d_pairs, hp.data(), hp.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->d_gext, hg.data(), hg.size() * sizeof(float4), hipMemcpyHostToDevice));
  }
  if (model->ground_type == CS_GEOM_HFIELD) {
    // any cell size: the narrowphase walks however many prisms lie under a geom (1 cm cells of the stairs_* terrains included);
    // contacts beyond the kernel variant's slots are counted (cosim_get "meta", word 8), never dropped silently
    if (!hfield) return fail(COSIM_EINVAL, "cosim_create: heightfield ground but no elevation data was passed");
    size_t nh = (size_t)model->hfield_nrow * model->hfield_ncol;
    HIP_TRY(hipMalloc(&e->d_hfield, nh * sizeof(float)));
    HIP_TRY(hipMemcpy(e->d_hfield, hfield, nh * sizeof(float), hipMemcpyHostToDevice));
    // tile maxima for the coarse terrain test ahead of the prism walk (terrain_max_under)
    const int mrow = (model->hfield_nrow + HF_TILE - 1) / HF_TILE, mcol = (model->hfield_ncol + HF_TILE - 1) / HF_TILE;
    std::vector<float> mip((size_t)mrow * mcol, 0.f);
    for (int r = 0; r < model->hfield_nrow; r++)
      for (int c = 0; c < model->hfield_ncol; c++) {
        float& m = mip[(size_t)(r / HF_TILE) * mcol + c / HF_TILE];
        const float h = hfield[(size_t)r * model->hfield_ncol + c];
        m = ((r % HF_TILE) == 0 && (c % HF_TILE) == 0) ? h : (h > m ? h : m);
      }
    HIP_TRY(hipMalloc(&e->d_hfield_mip, mip.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(e->d_hfield_mip, mip.data(), mip.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  default_params(e);
  { int rc2 = set_ranges(e, 1); if (rc2) return rc2; }   // (allocates the pacing events of the single-launch path)
  *out = e;
  return COSIM_OK;
}

int cosim_destroy(cosim_engine_t* e) {
  if (!e) return COSIM_OK;
  hipSetDevice(e->device);
  hipFree(e->d_model); hipFree(e->d_obs); hipFree(e->d_state); hipFree(e->d_params); hipFree(e->d_dbg);
  hipFree(e->d_hull_vert); hipFree(e->d_hull_adr); hipFree(e->d_hull_nbr); hipFree(e->d_hfield);
  hipFree(e->d_hull_cell); hipFree(e->d_hull_cand); hipFree(e->d_hfield_mip);
  hipFree(e->d_pairs); hipFree(e->d_gext); hipFree(e->d_ovf); hipFree(e->d_xcon); hipFree(e->d_xcnt); hipFree(e->d_xstate);
  for (hipEvent_t x : e->ev) hipEventDestroy(x);
  for (hipStream_t x : e->rstream) hipStreamDestroy(x);
  for (hipEvent_t x : e->rdone) hipEventDestroy(x);
  if (e->ev_in) hipEventDestroy(e->ev_in);
  delete e;
  return COSIM_OK;
}

int cosim_query(const cosim_engine_t* e, const char* name) {
  if (!e || !name) return fail(COSIM_EINVAL, "cosim_query: null argument");
  std::string n(name);
  if (n == "state_dim") return e->ho.state_dim;
  if (n == "action_dim") return e->model.nu;
  if (n == "command_dim") return e->ho.command_dim;
  if (n == "info_dim") return e->ho.info_dim;
  if (n == "nq") return e->model.nq;
  if (n == "nv") return e->model.nv;
  if (n == "nbody") return e->model.nbody;
  if (n == "ngeom") return e->model.ngeom;
  if (n == "n_envs") return e->n_envs;
  if (n == "state_stride") return e->lay.s_stride;
  if (n == "param_stride") return e->lay.p_stride;
  if (n == "lds_bytes") return e->lds_bytes;
  if (n == "contact_slots") return e->contact_slots;
  if (n == "fixup_contact_slots") return e->launch_fix ? e->fix_contact_slots : 0;   // 0: no large-capacity kernel behind this one
  if (n == "ranges") return e->n_ranges;
  if (n == "rollout") return e->launch_roll && e->epw == 1 ? 1 : 0;   // 1: cosim_rollout is available for this model / terrain
  if (n == "split") return (e->split && e->launch_stepx) ? e->narrow_waves : 0;   // waves per env of the narrowphase kernel; 0: fused kernel
  if (n == "pair_slots") return e->pair_slots;
  if (n == "stacked_dim") return e->ho.stacked_dim;
  if (n == "frame_dim") return e->ho.frame_dim;
  return fail(COSIM_EINVAL, "cosim_query: unknown name " + n);
}

int cosim_set_param(cosim_engine_t* e, const char* name, const float* host, int count) {
  if (!e || !name || !host) return fail(COSIM_EINVAL, "cosim_set_param: null argument");
  std::string n(name);
  const cosim_model_t& m = e->model;
  const Layout& L = e->lay;
  int off, width;
  if (n == "body_mass") { off = L.p_mass; width = m.nbody; }
  else if (n == "body_invweight0") { off = L.p_binvw; width = m.nbody; }
  else if (n == "dof_invweight0") { off = L.p_dinvw; width = m.nv; }
  else if (n == "dof_frictionloss") { off = L.p_floss; width = m.nv; }
  else if (n == "geom_friction") { off = L.p_gmu; width = m.ngeom; }
  else if (n == "kp") { off = L.p_kp; width = m.nu; }
  else if (n == "kd") { off = L.p_kd; width = m.nu; }
  else if (n == "meaninertia") { off = L.p_mean; width = 1; }
  else if (n == "solver_tolerance") { e->tol32 = host[0]; return COSIM_OK; }
  else if (n == "ls_tolerance_scale") { e->ls_scale = host[0]; return COSIM_OK; }
  else if (n == "max_newton") { e->max_newton = (int)host[0]; return COSIM_OK; }
  else if (n == "max_ls") { e->max_ls = (int)host[0]; return COSIM_OK; }
  else if (n == "wave_priority") {   // [usual Newton iterations per substep, lag thresholds of priority 1, 2, 3]; a huge first threshold switches it off
    if (count != 4) return fail(COSIM_EINVAL, "cosim_set_param: wave_priority takes 4 values");
    for (int k = 0; k < 4; k++) e->prio[k] = (int)host[k];
    return COSIM_OK;
  }
  else if (n == "debug_substeps") { e->nsub_override = (int)host[0]; return COSIM_OK; }
  else if (n == "ranges") return set_ranges(e, (int)host[0]);
  else if (n == "deferred_join") {   // 1: cosim_step leaves the join of the range streams to cosim_join (or to the next call that touches the state)
    e->deferred_join = (int)host[0] != 0;
    return COSIM_OK;
  }
  else if (n == "inflight") {   // control steps the host may run ahead of each range stream (0: unbounded)
    const int v = (int)host[0];
    if (v < 0 || v > 1024) return fail(COSIM_EINVAL, "cosim_set_param: inflight must be 0..1024");
    e->inflight = v;
    return set_ranges(e, e->n_ranges);
  }
  else if (n == "split") {   // 0: the fused kernel; 1: narrowphase and solver as kernels of their own, one pair of launches per substep
    if ((int)host[0] != 0 && !e->launch_stepx) return fail(COSIM_EINVAL, "cosim_set_param: no split pipeline for this model / terrain");
    e->split = (int)host[0] != 0;
    return COSIM_OK;
  }
  else if (n == "narrow_waves") {   // waves per env of the narrowphase kernel (wave w takes the geoms g % waves == w)
    const int v = (int)host[0];
    if (v < 1 || v > 24) return fail(COSIM_EINVAL, "cosim_set_param: narrow_waves must be 1..24");
    e->narrow_waves = v;
    return COSIM_OK;
  }
  else if (n == "narrow_occupancy") { e->narrow_occ = (int)host[0]; return COSIM_OK; }   // 2 | 3 | 4 (tuning)
  else if (n == "support_map") {   // 0: mesh support queries scan the whole hull (A/B and tests); 1: through the support maps (default)
    HIP_TRY(hipDeviceSynchronize());
    for (int g = 0; g < 64; g++) e->hm.g_hullmap[g] = (int)host[0] != 0 ? e->hullmap_of_geom[g] : -1;
    HIP_TRY(hipMemcpy(e->d_model, &e->hm, sizeof(DevModel), hipMemcpyHostToDevice));
    return COSIM_OK;
  }
  else if (n == "fixup") {   // 0: no fix-up launches (contacts beyond the fleet kernel's slots are left out and counted, as in round 2)
    if ((int)host[0] == 0) { e->launch_fix = nullptr; if (e->launch_roll_fix) { e->launch_roll = nullptr; e->launch_roll_fix = nullptr; } }
    return COSIM_OK;
  }
  else if (n == "contact_twist") {   // 1: switch a dense-row kernel to its contact-twist variant (more contact slots), where one exists
    if ((int)host[0] != 0) {
      if (!e->launch_ct) return fail(COSIM_EINVAL, "cosim_set_param: no contact-twist variant for this model / terrain");
      e->launch = e->launch_ct; e->launch_prof = e->launch_ct_prof; e->launch2 = nullptr; e->launch_prof2 = nullptr; e->epw = 1;
      e->launch_fix = nullptr; e->launch_roll = nullptr; e->launch_roll_fix = nullptr;
      if (e->model.nv == 14) e->pair_slots = 8;
      e->lds_bytes = e->ct_lds_bytes; e->contact_slots = e->ct_contact_slots;
    }
    return COSIM_OK;
  }
  else if (n == "timing_stride") { e->timing_stride = (int)host[0] >= 1 ? (int)host[0] : 1; return COSIM_OK; }   // time every n-th launch
  else if (n == "coop_walk") { e->coop_walk = (int)host[0] != 0; return COSIM_OK; }   // 1: the round-2 cooperative walk of hulls with few prisms under them (A/B)
  else if (n == "block_cull") { e->block_cull = (int)host[0] != 0; return COSIM_OK; }   // narrowphase kernel's block tests (default 1); 0 for A/B runs and tests
  else if (n == "boxbox_mode") { e->pair_boxbox = (int)host[0] != 0; return COSIM_OK; }   // 1: box-box pairs through mjc_BoxBox (default), 0: through MPR
  else if (n == "pair_mode") { e->pair_coop = (int)host[0] != 0; return COSIM_OK; }   // 1: hull pairs one at a time, wave-cooperative scans
  else if (n == "envs_per_wave") {   // 2: the two-environments-per-wave kernel (flat flamingo_light_v1, even env counts); 1: one per wave
    const int w = (int)host[0];
    if (w != 1 && !(w == 2 && e->launch2 && e->n_envs % 2 == 0)) return fail(COSIM_EINVAL, "cosim_set_param: envs_per_wave not available for this model / env count");
    e->epw = w;
    return e->n_ranges > 1 ? set_ranges(e, e->n_ranges) : COSIM_OK;   // (two envs per wave: even range sizes)
  }
  else return fail(COSIM_EINVAL, "cosim_set_param: unknown parameter " + n);
  if (count != e->n_envs * width) return fail(COSIM_EINVAL, "cosim_set_param: " + n + " expects n_envs*" + std::to_string(width) + " values");
  for (int i = 0; i < e->n_envs; i++)
    memcpy(e->h_params.data() + (size_t)i * L.p_stride + off, host + (size_t)i * width, width * sizeof(float));
  e->params_dirty = true;
  return COSIM_OK;
}

static int drain_events(cosim_engine* e) {
  for (int i = 0; i + 1 < e->ev_used; i += 2) {
    HIP_TRY(hipEventSynchronize(e->ev[i + 1]));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e->ev[i], e->ev[i + 1]));
    e->t_accum_ms += ms;
    e->t_launches++;
  }
  e->ev_used = 0;
  return COSIM_OK;
}

static KArgs base_args(cosim_engine* e) {
  KArgs a;
  memset(&a, 0, sizeof a);
  a.dm = e->d_model; a.ob = e->d_obs; a.lay = e->lay; a.state = e->d_state; a.params = e->d_params;
  a.hull_vert = e->d_hull_vert; a.hull_adr = e->d_hull_adr; a.hull_nbr = e->d_hull_nbr; a.hfield = e->d_hfield;
  a.hull_cell = e->d_hull_cell; a.hull_cand = e->d_hull_cand; a.hfield_mip = e->d_hfield_mip;
  a.pairs = e->d_pairs; a.gext = e->d_gext;
  a.n_envs = e->n_envs; a.seed_lo = (unsigned)e->seed; a.seed_hi = (unsigned)(e->seed >> 32); a.env_id0 = e->env_id0;
  a.tol32 = e->tol32; a.ls_scale = e->ls_scale; a.max_newton = e->max_newton; a.max_ls = e->max_ls; a.nsub_override = e->nsub_override; a.pair_coop = e->pair_coop; a.pair_boxbox = e->pair_boxbox; a.block_cull = e->block_cull; a.coop_walk = e->coop_walk;
  for (int k = 0; k < 4; k++) a.prio[k] = e->prio[k];
  a.ovf = nullptr; a.roll_steps = 1;
  a.xcon = e->d_xcon; a.xcnt = e->d_xcnt; a.xstate = e->d_xstate; a.nw = e->narrow_waves; a.sub_index = 0; a.sub_total = 0;
  return a;
}

int cosim_reset(cosim_engine_t* e, const uint8_t* mask_dev, const float* commands_dev, float* state_out_dev, void* stream) {
  if (!e || !state_out_dev) return fail(COSIM_EINVAL, "cosim_reset: null argument");
  HIP_TRY(hipSetDevice(e->device));
  int rc = upload_params(e);
  if (rc) return rc;
  rc = join_ranges(e, (hipStream_t)stream);
  if (rc) return rc;
  KArgs a = base_args(e);
  a.mode = MODE_RESET; a.mask = mask_dev; a.commands = commands_dev; a.state_out = state_out_dev;
  if (e->split && e->launch_stepx) e->launch_stepx(e, a, e->n_envs, (hipStream_t)stream);
  else (e->epw == 2 ? e->launch2 : e->launch)(e, a, e->n_envs, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return COSIM_OK;
}

int cosim_step(cosim_engine_t* e, const float* actions_dev, const float* commands_dev, float* state_out_dev, uint8_t* terminated_dev,
               uint8_t* truncated_dev, float* info_out_dev, void* stream) {
  if (!e) return fail(COSIM_EINVAL, "cosim_step: null argument");
  if (e->n_ranges <= 1) {   // one launch on the caller's stream, paced like the range launches below
    HIP_TRY(hipSetDevice(e->device));
    hipStreamCaptureStatus cap1 = hipStreamCaptureStatusNone;
    HIP_TRY(hipStreamIsCapturing((hipStream_t)stream, &cap1));
    const bool paced = e->inflight > 0 && cap1 == hipStreamCaptureStatusNone && (int)e->ring.size() >= e->inflight;
    if (paced && e->ring_pos >= e->inflight) HIP_TRY(hipEventSynchronize(e->ring[e->ring_pos % e->inflight]));
    int rc = cosim_step_range(e, 0, e->n_envs, actions_dev, commands_dev, state_out_dev, terminated_dev, truncated_dev, info_out_dev, stream);
    if (rc) return rc;
    if (paced) { HIP_TRY(hipEventRecord(e->ring[e->ring_pos % e->inflight], (hipStream_t)stream)); e->ring_pos++; }
    return COSIM_OK;
  }
  // fork: the range streams wait for whatever the caller's stream has been given so far (the step's inputs), then each steps its
  // range; join: the caller's stream waits for every range -- now, or (deferred_join) at the next cosim_join / state access, which
  // is what lets a range's next control step overlap the tail of the others' current one
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t cs = (hipStream_t)stream;
  // The inputs are ready once everything given to the caller's stream so far has run.  If that stream is idle they are ready now, and
  // the range streams need no wait: a cross-queue wait is a barrier packet ahead of every range launch (measured: 12.6 -> 11.0 M with
  // an idle caller stream).  Not while capturing: a query is illegal there, and the fork edge is what ties the range streams in.
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  HIP_TRY(hipStreamIsCapturing(cs, &cap));
  bool wait_in = true;
  if (cap == hipStreamCaptureStatusNone) {
    const hipError_t q = hipStreamQuery(cs);
    if (q == hipSuccess) wait_in = false;
    else if (q != hipErrorNotReady) return fail(COSIM_EHIP, std::string("hipStreamQuery: ") + hipGetErrorString(q));
    (void)hipGetLastError();   // hipErrorNotReady is sticky in hipGetLastError
  }
  if (wait_in) HIP_TRY(hipEventRecord(e->ev_in, cs));
  for (int i = 0; i < e->n_ranges; i++) {
    if (wait_in) HIP_TRY(hipStreamWaitEvent(e->rstream[i], e->ev_in, 0));
    // (not while capturing: a captured step is replayed, the host does not pace it)
    const bool paced = e->inflight > 0 && cap == hipStreamCaptureStatusNone;
    if (paced && e->ring_pos >= e->inflight)   // the step `inflight` steps back has left this range's stream
      HIP_TRY(hipEventSynchronize(e->ring[(size_t)i * e->inflight + e->ring_pos % e->inflight]));
    int rc = cosim_step_range(e, e->rfirst[i], e->rcount[i], actions_dev, commands_dev, state_out_dev, terminated_dev, truncated_dev, info_out_dev,
                              e->rstream[i]);
    if (rc) return rc;
    if (paced) HIP_TRY(hipEventRecord(e->ring[(size_t)i * e->inflight + e->ring_pos % e->inflight], e->rstream[i]));
  }
  if (e->inflight > 0 && cap == hipStreamCaptureStatusNone) e->ring_pos++;
  e->join_pending = true;
  if (!e->deferred_join) return join_ranges(e, cs);
  return COSIM_OK;
}

// The reference's loop with an action table (core/tester.py:66-97 with policy.get_action replaced by a lookup): `steps` control steps
// in ONE launch per range; row k of the [steps][N][...] buffers is what cosim_step would have been given / would have returned at
// step k.  Returns with the caller's stream waiting for everything (eager join).
int cosim_rollout(cosim_engine_t* e, int steps, const float* actions_dev, const float* commands_dev, float* state_out_dev, uint8_t* terminated_dev,
                  uint8_t* truncated_dev, float* info_out_dev, void* stream) {
  if (!e || !actions_dev || !state_out_dev || !terminated_dev || !truncated_dev || steps < 1) return fail(COSIM_EINVAL, "cosim_rollout: bad argument");
  if (!e->launch_roll || e->epw != 1) return fail(COSIM_EINVAL, "cosim_rollout: no rollout kernel for this model / terrain / kernel variant");
  if (e->ho.command_dim > 0 && !commands_dev) return fail(COSIM_EINVAL, "cosim_rollout: commands_dev is required when command_dim > 0");
  HIP_TRY(hipSetDevice(e->device));
  int rc = upload_params(e);
  if (rc) return rc;
  hipStream_t cs = (hipStream_t)stream;
  rc = join_ranges(e, cs);
  if (rc) return rc;
  KArgs a = base_args(e);
  a.mode = MODE_STEP; a.actions = actions_dev; a.commands = commands_dev; a.state_out = state_out_dev;
  a.terminated = terminated_dev; a.truncated = truncated_dev; a.info = info_out_dev; a.roll_steps = steps;
  a.ovf = e->launch_roll_fix ? e->d_ovf : nullptr;
  const int nr = e->n_ranges > 1 ? e->n_ranges : 1;
  if (nr > 1) HIP_TRY(hipEventRecord(e->ev_in, cs));
  for (int i = 0; i < nr; i++) {
    hipStream_t s = nr > 1 ? e->rstream[i] : cs;
    if (nr > 1) HIP_TRY(hipStreamWaitEvent(s, e->ev_in, 0));
    a.env_first = nr > 1 ? e->rfirst[i] : 0;
    a.env_count = nr > 1 ? e->rcount[i] : e->n_envs;
    int slot = -1;
    if (e->timing && e->ev_used + 2 <= (int)e->ev.size()) { slot = e->ev_used; e->ev_used += 2; HIP_TRY(hipEventRecord(e->ev[slot], s)); }
    e->launch_roll(e, a, a.env_count, s);
    HIP_TRY(hipGetLastError());
    if (slot >= 0) HIP_TRY(hipEventRecord(e->ev[slot + 1], s));
    if (a.ovf) { e->launch_roll_fix(e, a, a.env_count, s); HIP_TRY(hipGetLastError()); }
  }
  if (nr > 1) { e->join_pending = true; return join_ranges(e, cs); }
  return COSIM_OK;
}

// Test hook, host only (no GPU call): the support map of ONE hull (cosim_hullmap.h) against the full scan it replaces, with the
// kernels' fp32 comparisons: for each of `ndir` directions (hull frame) the arg-max vertex through the map -> out_map_idx and by
// scanning all n vertices -> out_scan_idx.  out_stats: [0] candidates in the table, [1] largest cell, [2] cells.
int cosim_hull_support_check(const float* verts, int n, const int* adr, const int* nbr, const float* dirs, int ndir, int* out_map_idx,
                             int* out_scan_idx, int* out_stats) {
  if (!verts || !adr || !nbr || !dirs || !out_map_idx || !out_scan_idx || n < 1 || ndir < 0) return fail(COSIM_EINVAL, "cosim_hull_support_check: bad argument");
  std::vector<float> cells, cand;
  build_support_map(verts, n, adr, nbr, cells, cand);
  auto as_int = [](float f) { union { int i; float f; } u; u.f = f; return u.i; };
  int biggest = 0, total = 0;
  for (int c = 0; c < HM_CELLS; c++) { const int k = as_int(cells[4 * (size_t)HM_REC * c]); biggest = k > biggest ? k : biggest; total += k; }
  if (out_stats) { out_stats[0] = total; out_stats[1] = biggest; out_stats[2] = HM_CELLS; }
  for (int d = 0; d < ndir; d++) {
    const float* l = dirs + 3 * (size_t)d;
    float best = -3.0e38f;
    int bi = 0;
    for (int i = 0; i < n; i++) {
      const float t = l[0] * verts[3 * i] + l[1] * verts[3 * i + 1] + l[2] * verts[3 * i + 2];
      if (t > best) { best = t; bi = i; }
    }
    out_scan_idx[d] = bi;
    const float* rec = &cells[4 * (size_t)HM_REC * support_cell(l)];
    const int count = as_int(rec[0]), ovf = as_int(rec[1]);
    best = -3.0e38f;
    union { int i; float f; } ix;
    ix.f = rec[4 + 3];
    for (int i = 0; i < count; i++) {
      const float* x = i < HM_INLINE ? rec + 4 * (1 + i) : &cand[4 * (size_t)(ovf + i - HM_INLINE)];
      const float t = l[0] * x[0] + l[1] * x[1] + l[2] * x[2];
      if (t > best) { best = t; ix.f = x[3]; }
    }
    out_map_idx[d] = ix.i;
  }
  return COSIM_OK;
}

int cosim_join(cosim_engine_t* e, void* stream) {
  if (!e) return fail(COSIM_EINVAL, "cosim_join: null engine");
  HIP_TRY(hipSetDevice(e->device));
  return join_ranges(e, (hipStream_t)stream);
}

int cosim_range(const cosim_engine_t* e, int i, int* first, int* count, void** stream) {
  if (!e || i < 0 || i >= e->n_ranges) return fail(COSIM_EINVAL, "cosim_range: bad argument");
  if (first) *first = e->n_ranges > 1 ? e->rfirst[i] : 0;
  if (count) *count = e->n_ranges > 1 ? e->rcount[i] : e->n_envs;
  if (stream) *stream = e->n_ranges > 1 ? (void*)e->rstream[i] : nullptr;
  return COSIM_OK;
}

// After something was enqueued on range stream i from outside (a per-range policy, a reporter reduction): re-arm the range's "done"
// event so that a later join also waits for that work.
int cosim_range_mark(cosim_engine_t* e, int i) {
  if (!e || i < 0 || i >= e->n_ranges || e->n_ranges <= 1) return fail(COSIM_EINVAL, "cosim_range_mark: bad argument");
  e->join_pending = true;   // (the "done" events are recorded at join time: everything on the range stream by then is covered)
  return COSIM_OK;
}

int cosim_step_range(cosim_engine_t* e, int first, int count, const float* actions_dev, const float* commands_dev, float* state_out_dev,
                     uint8_t* terminated_dev, uint8_t* truncated_dev, float* info_out_dev, void* stream) {
  if (!e || !actions_dev || !state_out_dev || !terminated_dev || !truncated_dev) return fail(COSIM_EINVAL, "cosim_step: null argument");
  if (e->ho.command_dim > 0 && !commands_dev) return fail(COSIM_EINVAL, "cosim_step: commands_dev is required when command_dim > 0");
  if (first < 0 || count < 1 || first + count > e->n_envs) return fail(COSIM_EINVAL, "cosim_step_range: range outside the fleet");
  if (e->epw == 2 && ((first | count) & 1)) return fail(COSIM_EINVAL, "cosim_step_range: two-environments-per-wave kernel needs even ranges");
  HIP_TRY(hipSetDevice(e->device));
  int rc = upload_params(e);
  if (rc) return rc;
  KArgs a = base_args(e);
  a.mode = MODE_STEP; a.actions = actions_dev; a.commands = commands_dev; a.state_out = state_out_dev;
  a.terminated = terminated_dev; a.truncated = truncated_dev; a.info = info_out_dev;
  a.env_first = first; a.env_count = count;
  if (e->split && e->launch_stepx && e->narrow_occ == 0) a.dbg = e->d_dbg;   // diagnostic narrowphase build accumulates its counters there
  a.ovf = (e->launch_fix && e->epw == 1) ? e->d_ovf : nullptr;
  hipStream_t s = (hipStream_t)stream;
  // kernel timing: one HIP event pair per launch on the launch stream, read back in cosim_kernel_time() (no sync here)
  int slot = -1;
  if (e->timing && (e->launch_seq++ % (unsigned)e->timing_stride) == 0) {
    if (e->ev_used + 2 > (int)e->ev.size()) {
      if (e->ev.size() >= 4096) { int rc2 = drain_events(e); if (rc2) return rc2; }
      else for (int i = 0; i < 2; i++) { hipEvent_t x; HIP_TRY(hipEventCreate(&x)); e->ev.push_back(x); }
    }
    slot = e->ev_used;
    e->ev_used += 2;
    HIP_TRY(hipEventRecord(e->ev[slot], s));
  }
  if (e->split && e->launch_stepx) {
    // one pair of launches per substep: the prism walk (narrow_waves waves per env), then the solver with the contacts it left
    const int fs = e->nsub_override > 0 ? e->nsub_override : e->model.frame_skip;
    for (int sub = 0; sub < fs; sub++) {
      a.sub_index = sub; a.sub_total = fs;
      e->launch_narrow(e, a, count * e->narrow_waves, s);
      e->launch_stepx(e, a, count, s);
    }
  } else (e->epw == 2 ? e->launch2 : e->launch)(e, a, count, s);
  HIP_TRY(hipGetLastError());
  if (slot >= 0) HIP_TRY(hipEventRecord(e->ev[slot + 1], s));
  if (a.ovf) {   // envs the fleet kernel flagged (more contacts than it has slots for) are redone by the large-capacity kernel
    e->launch_fix(e, a, count, s);
    HIP_TRY(hipGetLastError());
  }
  return COSIM_OK;
}

static int locate(cosim_engine* e, const std::string& n, int* off, int* width) {
  if (n == "qpos") { *off = e->lay.s_qpos; *width = e->model.nq; }
  else if (n == "qvel") { *off = e->lay.s_qvel; *width = e->model.nv; }
  else if (n == "qacc_warmstart") { *off = e->lay.s_warm; *width = e->model.nv; }
  else if (n == "meta") { *off = e->lay.s_meta; *width = Layout::NMETA; }
  else return fail(COSIM_EINVAL, "unknown state field " + n);
  return COSIM_OK;
}

int cosim_get(cosim_engine_t* e, const char* name, float* out_dev, void* stream) {
  if (!e || !name || !out_dev) return fail(COSIM_EINVAL, "cosim_get: null argument");
  HIP_TRY(hipSetDevice(e->device));
  int off, width;
  int rc = locate(e, name, &off, &width);
  if (rc) return rc;
  rc = join_ranges(e, (hipStream_t)stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpy2DAsync(out_dev, width * sizeof(float), e->d_state + off, e->lay.s_stride * sizeof(float), width * sizeof(float),
                           e->n_envs, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return COSIM_OK;
}

int cosim_set(cosim_engine_t* e, const char* name, const float* in_dev, void* stream) {
  if (!e || !name || !in_dev) return fail(COSIM_EINVAL, "cosim_set: null argument");
  HIP_TRY(hipSetDevice(e->device));
  int off, width;
  int rc = locate(e, name, &off, &width);
  if (rc) return rc;
  rc = join_ranges(e, (hipStream_t)stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpy2DAsync(e->d_state + off, e->lay.s_stride * sizeof(float), in_dev, width * sizeof(float), width * sizeof(float),
                           e->n_envs, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return COSIM_OK;
}

__global__ void push_kernel(float* state, Layout lay, const float* v, const uint8_t* mask, int n) {
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= n || (mask && !mask[env])) return;
  float* rec = state + (size_t)env * lay.s_stride;
  // qvel[:2] = (R^T v_world)[:2], qvel[2] = v_world[2]   (reference flamingo_light_v1.py:234-243; quaternion used raw)
  float w = rec[lay.s_qpos + 3], x = rec[lay.s_qpos + 4], y = rec[lay.s_qpos + 5], z = rec[lay.s_qpos + 6];
  float R00 = 1 - 2 * y * y - 2 * z * z, R01 = 2 * x * y - 2 * z * w, R10 = 2 * x * y + 2 * z * w, R11 = 1 - 2 * x * x - 2 * z * z,
        R20 = 2 * x * z - 2 * y * w, R21 = 2 * y * z + 2 * x * w;
  const float* vw = v + (size_t)env * 3;
  rec[lay.s_qvel + 0] = R00 * vw[0] + R10 * vw[1] + R20 * vw[2];
  rec[lay.s_qvel + 1] = R01 * vw[0] + R11 * vw[1] + R21 * vw[2];
  rec[lay.s_qvel + 2] = vw[2];
}

int cosim_event_push(cosim_engine_t* e, const float* v_dev, const uint8_t* mask_dev, void* stream) {
  if (!e || !v_dev) return fail(COSIM_EINVAL, "cosim_event_push: null argument");
  HIP_TRY(hipSetDevice(e->device));
  { int rc = join_ranges(e, (hipStream_t)stream); if (rc) return rc; }
  hipLaunchKernelGGL(push_kernel, dim3((e->n_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->d_state, e->lay, v_dev, mask_dev, e->n_envs);
  HIP_TRY(hipGetLastError());
  return COSIM_OK;
}

int cosim_debug_forward(cosim_engine_t* e, int env, const char* name, float* host_out, int capacity) {
  if (!e || !host_out || env < 0 || env >= e->n_envs) return fail(COSIM_EINVAL, "cosim_debug_forward: bad argument");
  (void)name;
  HIP_TRY(hipSetDevice(e->device));
  int rc = upload_params(e);
  if (rc) return rc;
  rc = join_ranges(e, 0);
  if (rc) return rc;
  HIP_TRY(hipMemset(e->d_dbg, 0, 8192 * sizeof(float)));
  KArgs a = base_args(e);
  a.mode = MODE_DEBUG; a.dbg = e->d_dbg; a.dbg_env = env;
  if (e->split && e->launch_stepx) { e->launch_narrow(e, a, e->narrow_waves, 0); e->launch_stepx(e, a, 1, 0); }   // the product's own pair of kernels
  else (e->epw == 2 ? e->launch2 : e->launch)(e, a, 1, 0);   // two-per-wave kernel: both groups replay env `env`, same dump twice
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  int n = capacity < 8192 ? capacity : 8192;
  HIP_TRY(hipMemcpy(host_out, e->d_dbg, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  return COSIM_OK;
}

int cosim_profile_step(cosim_engine_t* e, const float* actions_dev, const float* commands_dev, float* state_out_dev, uint8_t* terminated_dev,
                       uint8_t* truncated_dev, double* cycles_out16) {
  if (!e || !actions_dev || !state_out_dev || !terminated_dev || !truncated_dev || !cycles_out16) return fail(COSIM_EINVAL, "cosim_profile_step: null argument");
  if (!e->launch_prof) return fail(COSIM_EINVAL, "cosim_profile_step: no diagnostic kernel for this model");
  HIP_TRY(hipSetDevice(e->device));
  int rc = upload_params(e);
  if (rc) return rc;
  rc = join_ranges(e, 0);
  if (rc) return rc;
  HIP_TRY(hipMemset(e->d_dbg, 0, 8192 * sizeof(float)));
  KArgs a = base_args(e);
  a.mode = MODE_STEP; a.actions = actions_dev; a.commands = commands_dev; a.state_out = state_out_dev;
  a.terminated = terminated_dev; a.truncated = truncated_dev; a.dbg = e->d_dbg;
  (e->epw == 2 ? e->launch_prof2 : e->launch_prof)(e, a, e->n_envs, 0);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  unsigned long long raw[32];
  HIP_TRY(hipMemcpy(raw, e->d_dbg, sizeof raw, hipMemcpyDeviceToHost));
  for (int i = 0; i < 32; i++) cycles_out16[i] = (double)raw[i] / (double)(e->n_envs / e->epw);   // per wave
  return COSIM_OK;
}

// Test hook: support points of mesh geom `geom` at identity pose for n_dirs directions, from the device's own support routines:
// out [n_dirs][6] = the lane-parallel routine's point, then the wave-cooperative routine's.  use_map 0: full scans of the hull.
int cosim_debug_support(cosim_engine_t* e, int geom, const float* dirs_host, int n_dirs, float* out_host, int use_map) {
  if (!e || !dirs_host || !out_host || n_dirs < 1 || geom < 0 || geom >= e->hm.ngeom) return fail(COSIM_EINVAL, "cosim_debug_support: bad argument");
  if (e->hm.rec[geom].g_type != CS_GEOM_MESH) return fail(COSIM_EINVAL, "cosim_debug_support: not a mesh geom");
  HIP_TRY(hipSetDevice(e->device));
  float *d_dirs = nullptr, *d_out = nullptr;
  auto run = [&]() -> hipError_t {   // (one exit, so that the two scratch buffers are released on every path)
    hipError_t r;
    if ((r = hipMalloc(&d_dirs, (size_t)n_dirs * 3 * sizeof(float))) != hipSuccess) return r;
    if ((r = hipMalloc(&d_out, (size_t)n_dirs * 6 * sizeof(float))) != hipSuccess) return r;
    if ((r = hipMemcpy(d_dirs, dirs_host, (size_t)n_dirs * 3 * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) return r;
    const HullGraph H{e->d_hull_vert, e->d_hull_adr, e->d_hull_nbr, e->d_hull_cell, e->d_hull_cand};
    hipLaunchKernelGGL(support_probe_kernel, dim3((n_dirs + 63) / 64), dim3(64), 0, 0, e->d_model, H, geom, d_dirs, n_dirs, d_out, use_map);
    if ((r = hipGetLastError()) != hipSuccess) return r;
    if ((r = hipDeviceSynchronize()) != hipSuccess) return r;
    return hipMemcpy(out_host, d_out, (size_t)n_dirs * 6 * sizeof(float), hipMemcpyDeviceToHost);
  };
  const hipError_t r = run();
  (void)hipFree(d_dirs); (void)hipFree(d_out);
  if (r != hipSuccess) return fail(COSIM_EHIP, std::string("cosim_debug_support: ") + hipGetErrorString(r));
  return COSIM_OK;
}

int cosim_debug_counters(cosim_engine_t* e, unsigned long long* out32, int clear) {   // the 32 64-bit words diagnostic kernels accumulate into
  if (!e || !out32) return fail(COSIM_EINVAL, "cosim_debug_counters: null argument");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out32, e->d_dbg, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  if (clear) HIP_TRY(hipMemset(e->d_dbg, 0, 8192 * sizeof(float)));
  return COSIM_OK;
}

int cosim_set_timing(cosim_engine_t* e, int enabled) {
  if (!e) return fail(COSIM_EINVAL, "cosim_set_timing: null engine");
  HIP_TRY(hipSetDevice(e->device));
  int rc = drain_events(e);
  if (rc) return rc;
  e->timing = enabled != 0;
  if (e->timing)   // event pool up front: creating events inside a timed loop costs host time per launch
    while (e->ev.size() < 4096) { hipEvent_t x; HIP_TRY(hipEventCreate(&x)); e->ev.push_back(x); }
  e->t_accum_ms = 0.0;
  e->t_launches = 0;
  return COSIM_OK;
}

int cosim_kernel_time(cosim_engine_t* e, float* avg_ms, int* launches) {
  if (!e || !avg_ms || !launches) return fail(COSIM_EINVAL, "cosim_kernel_time: null argument");
  HIP_TRY(hipSetDevice(e->device));
  int rc = drain_events(e);
  if (rc) return rc;
  *launches = e->t_launches;
  *avg_ms = e->t_launches ? (float)(e->t_accum_ms / e->t_launches) : 0.f;
  e->t_accum_ms = 0.0;
  e->t_launches = 0;
  return COSIM_OK;
}

}  // extern "C"
