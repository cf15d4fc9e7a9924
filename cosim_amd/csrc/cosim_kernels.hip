// cosim_kernels.hip — the hot path as hand-written HIP for gfx950 (CDNA4), one environment per wavefront.
//
// One launch = one control step of every environment on this GPU, i.e. everything the reference does inside
//   CommandWrapper.step -> TimeLimitWrapper.step -> StateBuildWrapper.step -> FlamingoLightV1.step
// (reference envs/wrappers.py:258-269,309-320,391-405; envs/flamingo_light_v1/flamingo_light_v1.py:131-164):
//   receive_user_command -> delay_filter -> PD torque -> frame_skip x mj_step -> _get_obs -> _build_state ->
//   _apply_command_inplace, plus _get_info / _is_done / time limit and (engine extension) in-kernel auto-reset.
// mj_step itself is MuJoCo's (mujoco==3.2.7, not in the reference tree); the stages below follow its published
// pipeline (kinematics, comPos, crb, collision, makeConstraint, comVel, rne, fwdActuation, Newton solve with exact
// line search, implicitfast) re-designed for a 64-lane wave: lanes own bodies / dofs / geoms / constraint rows,
// intermediates sit in LDS, the nv x nv Newton Hessian is factorised in registers (row i in lane i) with
// v_readlane broadcasts, and state is read once and written once per control step.
#include "cosim_dev.h"

namespace cosim {

#define WSYNC()                                          \
  do {                                                   \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                     \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)

constexpr float MINVAL = 1e-15f;
constexpr float MINIMP = 0.0001f, MAXIMP = 0.9999f;

enum { RT_NONE = -1, RT_EQ = 0, RT_CONTACT = 1, RT_FRIC = 2, RT_LIMIT = 3 };
enum { MODE_STEP = 0, MODE_RESET = 1, MODE_DEBUG = 2 };

struct KArgs {
  const DevModel* dm;
  const DevObs* ob;
  Layout lay;
  float* state;         // [N, s_stride]
  const float* params;  // [N, p_stride]
  const float* hull_vert;
  const int* hull_adr;
  const int* hull_nbr;
  const float* hfield;
  const float* actions;   // [N, nu]
  const float* commands;  // [N, command_dim]
  float* state_out;       // [N, state_dim]
  uint8_t* terminated;
  uint8_t* truncated;
  float* info;            // [N, info_dim] or null
  const uint8_t* mask;    // reset mask or null
  float* dbg;             // debug dump buffer (MODE_DEBUG)
  int dbg_env;
  int n_envs, mode;
  unsigned seed_lo, seed_hi;
  long long env_id0;
  float tol32;            // fp32 solver tolerance
  int max_newton;
};

// ------------------------------------------------------------------------------------------------ wave helpers
__device__ __forceinline__ float rl(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ float rfl(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ int rfli(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ unsigned long long lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// ------------------------------------------------------------------------------------------------ small math
__device__ __forceinline__ void qmul(float* r, const float* a, const float* b) {
  float w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  float x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  float y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  float z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
__device__ __forceinline__ void qnorm(float* q) {
  float n = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (n < 1e-30f) { q[0] = 1.f; q[1] = q[2] = q[3] = 0.f; return; }
  float s = rsqrtf(n);
  q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
}
__device__ __forceinline__ void q2m(float* m, const float* q) {
  float q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3], q11 = q[1] * q[1], q12 = q[1] * q[2],
        q13 = q[1] * q[3], q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2.f * (q12 - q03); m[2] = 2.f * (q13 + q02); m[3] = 2.f * (q12 + q03);
  m[5] = 2.f * (q23 - q01); m[6] = 2.f * (q13 - q02); m[7] = 2.f * (q23 + q01);
}
__device__ __forceinline__ void qrot(float* r, const float* q, const float* v) {  // r = R(q) v
  float m[9];
  q2m(m, q);
  float x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
        z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ void cross(float* r, const float* a, const float* b) {
  float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void mul_inert(float* res, const float* i, const float* v) {  // mju_mulInertVec
  res[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  res[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  res[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  res[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  res[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  res[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
__device__ __forceinline__ void make_frame(const float* n, float* t1, float* t2) {  // mju_makeFrame with zero y-axis
  t1[0] = t1[1] = t1[2] = 0.f;
  if (n[1] < 0.5f && n[1] > -0.5f) t1[1] = 1.f; else t1[2] = 1.f;
  float d = dot3(n, t1);
  t1[0] -= d * n[0]; t1[1] -= d * n[1]; t1[2] -= d * n[2];
  float s = rsqrtf(dot3(t1, t1));
  t1[0] *= s; t1[1] *= s; t1[2] *= s;
  cross(t2, n, t1);
}

// Philox4x32-10, counter-based: key = (seed, global env id), counter = (step, purpose, index, 0)
__device__ __forceinline__ void philox(unsigned k0, unsigned k1, unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned* out) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float u01(unsigned x) { return (float)(x >> 8) * (1.0f / 16777216.0f) + (0.5f / 16777216.0f); }  // (0,1)

// ------------------------------------------------------------------------------------------------ LDS per env
template <int NV, int NB>
struct EnvLds {
  static constexpr int LDJ = NV | 1;
  static constexpr int TRI = NV * (NV + 1) / 2;
  static constexpr int MC = 14;  // contact slots
  float qpos[CS_MAXQ], qvel[NV], qacc[NV], qact[NV], qsm[NV], qcon[NV], Ma[NV], Mv[NV], sr[NV], dofD[NV], qbias[NV];
  float xpos[NB][3], xquat[NB][4], xanc[NB][3], xax[NB][3];
  float cin[NB][10], cfb[NB][6];
  float cdof[NV][6], cdd[NV][6];
  float M[TRI], H[TRI];
  float J[MAXROW][LDJ];
  float rowf[MAXROW], rowD[MAXROW];
  float cpos[MC][3], cnrm[MC][3], cdist[MC], cmu[MC];
  int cgeom[MC];
  int lim_body[MAXROW];
  float lim_sign[MAXROW], lim_dist[MAXROW];
  float p_mass[NB], p_binvw[NB], p_dinvw[NV], p_floss[NV], p_gmu[MAXG];
  float act[MAXU], cmd[CS_MAXCMD];
  float com[3];
};

// index of (i, k) in the packed lower triangle
__device__ __forceinline__ int tri_idx(int i, int k) { return i >= k ? (i * (i + 1)) / 2 + k : (k * (k + 1)) / 2 + i; }

// ------------------------------------------------------------------------------------------------ Cholesky in registers
// Lane i < NV holds row i of a symmetric positive definite matrix in a[0..NV).  On exit a[k], k <= i, is L[i][k] and
// a[k], k > i, is L[k][i] (row i of L^T); dinv is 1 / L[i][i].  Right-looking, column j broadcast with v_readlane.
template <int NV>
__device__ __forceinline__ void chol_regs(float (&a)[NV], float& dinv, int lane) {
#pragma unroll
  for (int j = 0; j < NV; j++) {
    float ajj = rl(a[j], j);
    float inv = rsqrtf(fmaxf(ajj, 1e-30f));
    if (lane == j) dinv = inv;
    if (lane >= j) a[j] *= inv;  // column j of L (lane j: L[j][j])
#pragma unroll
    for (int k = j + 1; k < NV; k++) {
      float t = a[k] * inv;
      a[k] = (lane == j) ? t : a[k];  // row j of L^T
    }
#pragma unroll
    for (int k = j + 1; k < NV; k++) {
      float lkj = rl(a[k], j);
      float upd = a[k] - a[j] * lkj;
      a[k] = (lane > j) ? upd : a[k];
    }
  }
}
// solve (L L^T) x = b; lane i holds b_i and returns x_i
template <int NV>
__device__ __forceinline__ float chol_solve_regs(const float (&a)[NV], float dinv, float b, int lane) {
#pragma unroll
  for (int j = 0; j < NV; j++) {
    float yj = rl(b, j) * rl(dinv, j);
    float upd = b - a[j] * yj;
    b = (lane == j) ? yj : ((lane > j) ? upd : b);
  }
#pragma unroll
  for (int j = NV - 1; j >= 0; j--) {
    float xj = rl(b, j) * rl(dinv, j);
    float upd = b - a[j] * xj;
    b = (lane == j) ? xj : ((lane < j) ? upd : b);
  }
  return b;
}

// ------------------------------------------------------------------------------------------------ impedance (mj_makeImpedance)
__device__ __forceinline__ float impedance(const float* solimp, float pos, float margin) {
  float d0 = fminf(MAXIMP, fmaxf(MINIMP, solimp[0])), d1 = fminf(MAXIMP, fmaxf(MINIMP, solimp[1]));
  float width = fmaxf(0.f, solimp[2]), mid = fminf(MAXIMP, fmaxf(MINIMP, solimp[3])), power = fmaxf(1.f, solimp[4]);
  if (d0 == d1 || width <= MINVAL) return 0.5f * (d0 + d1);
  float x = fabsf((pos - margin) / width);
  if (x >= 1.f) return d1;
  if (x <= 0.f) return d0;
  float y;
  if (power == 1.f) y = x;
  else if (x <= mid) y = powf(x, power) / powf(mid, power - 1.f);
  else y = 1.f - powf(1.f - x, power) / powf(1.f - mid, power - 1.f);
  return d0 + y * (d1 - d0);
}

// ------------------------------------------------------------------------------------------------ the kernel
template <int NV, int NB>
__global__ __launch_bounds__(64, 4) void env_kernel(KArgs A) {
  using L = EnvLds<NV, NB>;
  constexpr int LDJ = L::LDJ;
  constexpr int TRI = L::TRI;
  constexpr int MC = L::MC;
  constexpr int EPL = (TRI + 63) / 64;
  __shared__ L S;
  const int lane = threadIdx.x;
  const int env = A.mode == MODE_DEBUG ? A.dbg_env : blockIdx.x;
  if (env >= A.n_envs) return;
  const DevModel& dm = *A.dm;
  const DevObs& ob = *A.ob;
  const Layout lay = A.lay;
  float* rec = A.state + (size_t)env * lay.s_stride;
  const float* par = A.params + (size_t)env * lay.p_stride;
  const int nv = NV, nbody = dm.nbody, nu = dm.nu, nq = dm.nq, ngeom = dm.ngeom;
  const float h = dm.timestep;

  // ---- per-env parameters -> LDS
  if (lane < nbody) { S.p_mass[lane] = par[lay.p_mass + lane]; S.p_binvw[lane] = par[lay.p_binvw + lane]; }
  if (lane < nv) { S.p_dinvw[lane] = par[lay.p_dinvw + lane]; S.p_floss[lane] = par[lay.p_floss + lane]; }
  if (lane < ngeom) S.p_gmu[lane] = par[lay.p_gmu + lane];
  const float meaninertia = par[lay.p_mean];
  int* meta = reinterpret_cast<int*>(rec + lay.s_meta);
  int sim_step = meta[0];
  unsigned step_count = (unsigned)meta[1];
  int has_prev = meta[2];
  const unsigned long long gid = (unsigned long long)(A.env_id0 + env);
  const unsigned k0 = A.seed_lo ^ (unsigned)gid, k1 = A.seed_hi ^ (unsigned)(gid >> 32);

  bool do_reset = false;
  if (A.mode == MODE_RESET) {
    do_reset = A.mask == nullptr || A.mask[env] != 0;
    if (!do_reset) return;
  }

  // ---- applied command (CommandWrapper.receive_user_command, wrappers.py:349-375) from the pre-step pose
  float applied_cmd = 0.f;  // lane c < command_dim
  if (lane < ob.command_dim && A.commands != nullptr) {
    float uc = A.commands[(size_t)env * ob.command_dim + lane];
    if (!ob.position_command) applied_cmd = uc * ob.command_scales[lane];
    else {
      float px = rec[lay.s_qpos + 0], py = rec[lay.s_qpos + 1];
      float w = rec[lay.s_qpos + 3], x = rec[lay.s_qpos + 4], y = rec[lay.s_qpos + 5], z = rec[lay.s_qpos + 6];
      float tx = A.commands[(size_t)env * ob.command_dim + 0], ty = A.commands[(size_t)env * ob.command_dim + 1];
      float dx = tx - px, dy = ty - py;
      float yaw = atan2f(2.f * (w * z + x * y), 1.f - 2.f * (y * y + z * z));
      float cy = cosf(-yaw), sy = sinf(-yaw);
      applied_cmd = lane == 0 ? cy * dx - sy * dy : sy * dx + cy * dy;
    }
  }

  // sensor values of the last forward pass (uniform across the wave)
  float s_quat[4] = {1.f, 0.f, 0.f, 0.f}, s_gyro[3] = {0.f, 0.f, 0.f}, s_vel[3] = {0.f, 0.f, 0.f};
  float raw_action = 0.f;   // lane u: self.action of this step
  float prev_action = 0.f;  // lane u: self.prev_action (raw action of the previous step)
  float tq_lane = 0.f;
  int terminated = 0, truncated = 0, bad = 0;

  if (A.mode != MODE_RESET) {
    // ---- state -> LDS
    if (lane < nq) S.qpos[lane] = rec[lay.s_qpos + lane];
    if (lane < nv) { S.qvel[lane] = rec[lay.s_qvel + lane]; S.qacc[lane] = rec[lay.s_warm + lane]; S.qact[lane] = 0.f; }
    WSYNC();

    // ---- control (once per control step): delay filter + PD, zero-order hold over the substeps
    if (A.mode == MODE_STEP) {
      unsigned rnd[4];
      philox(k0, k1, step_count, 0u, 0u, 0u, rnd);
      const bool delayed = (ob.action_delay_prob > u01(rnd[0])) && has_prev;  // control_manager.py:15-23
      if (lane < nu) {
        raw_action = A.actions[(size_t)env * nu + lane];
        prev_action = rec[lay.s_lastact + lane];
        float filt = delayed ? rec[lay.s_delay + lane] : raw_action;
        rec[lay.s_delay + lane] = raw_action;
        float a = filt * dm.ctl_scale[lane], g = dm.ctl_gear[lane];
        float q = S.qpos[dm.ctl_qadr[lane]] * g, qd = S.qvel[dm.ctl_dadr[lane]] * g;
        float kp = par[lay.p_kp + lane], kd = par[lay.p_kd + lane];
        float t = dm.ctl_velmode[lane] ? kd * (a - qd) : kp * (a - q) + kd * (0.f - qd);
        t *= dm.ctl_gamma[lane];
        t = fminf(dm.ctl_maxtq[lane], fmaxf(-dm.ctl_maxtq[lane], t));
        tq_lane = t;
        // mj_fwdActuation: ctrl clamp, gear, then the joint-level actuatorfrcrange clamp (one motor per dof)
        float c = dm.act_ctrllimited[lane] ? fminf(dm.act_ctrlrange[lane][1], fmaxf(dm.act_ctrlrange[lane][0], t)) : t;
        float f = dm.act_gear[lane] * c;
        int d = dm.act_dof[lane];
        if (dm.dof_frclimited[d]) f = fminf(dm.dof_frcrange[d][1], fmaxf(dm.dof_frcrange[d][0], f));
        S.qact[d] = f;
      }
      has_prev = 1;
      sim_step += 1;
      WSYNC();
    }

    const int nsub = A.mode == MODE_DEBUG ? 1 : dm.frame_skip;
    for (int sub = 0; sub < nsub; sub++) {
      // =========================================================== mj_kinematics: level-synchronous over the tree
      const int b_level = lane < nbody ? dm.body_level[lane] : -1;
      for (int lev = 1; lev <= dm.maxdepth; lev++) {
        if (b_level == lev) {
          const int b = lane, jt = dm.body_jtype[b];
          float xp[3], xq[4], anc[3] = {0.f, 0.f, 0.f}, ax[3] = {0.f, 0.f, 1.f};
          if (jt == CS_JNT_FREE) {
            const int qa = dm.body_qadr[b];
            for (int k = 0; k < 3; k++) xp[k] = S.qpos[qa + k];
            for (int k = 0; k < 4; k++) xq[k] = S.qpos[qa + 3 + k];
            qnorm(xq);
            for (int k = 0; k < 3; k++) { anc[k] = xp[k]; ax[k] = dm.jnt_axis[b][k]; }
          } else {
            const int p = dm.body_parent[b];
            float pq[4] = {S.xquat[p][0], S.xquat[p][1], S.xquat[p][2], S.xquat[p][3]};
            float v[3];
            qrot(v, pq, dm.body_pos[b]);
            for (int k = 0; k < 3; k++) xp[k] = S.xpos[p][k] + v[k];
            qmul(xq, pq, dm.body_quat[b]);
            if (jt == CS_JNT_HINGE) {
              qrot(v, xq, dm.jnt_pos[b]);
              for (int k = 0; k < 3; k++) anc[k] = xp[k] + v[k];
              qrot(ax, xq, dm.jnt_axis[b]);
              float ang = S.qpos[dm.body_qadr[b]] - dm.jnt_q0[b];
              float sn, cs;
              sincosf(0.5f * ang, &sn, &cs);
              float ql[4] = {cs, dm.jnt_axis[b][0] * sn, dm.jnt_axis[b][1] * sn, dm.jnt_axis[b][2] * sn};
              qmul(xq, xq, ql);
              qrot(v, xq, dm.jnt_pos[b]);
              for (int k = 0; k < 3; k++) xp[k] = anc[k] - v[k];
            }
            qnorm(xq);
          }
          for (int k = 0; k < 3; k++) { S.xpos[b][k] = xp[k]; S.xanc[b][k] = anc[k]; S.xax[b][k] = ax[k]; }
          for (int k = 0; k < 4; k++) S.xquat[b][k] = xq[k];
        }
        if (lane == 0 && lev == 1) {
          S.xpos[0][0] = S.xpos[0][1] = S.xpos[0][2] = 0.f;
          S.xquat[0][0] = 1.f; S.xquat[0][1] = S.xquat[0][2] = S.xquat[0][3] = 0.f;
        }
        WSYNC();
      }

      // =========================================================== mj_comPos: com, cinert (lane = body)
      float cinert[10];
#pragma unroll
      for (int k = 0; k < 10; k++) cinert[k] = 0.f;
      float xip[3] = {0.f, 0.f, 0.f}, bmass = 0.f, ximat[9];
      if (lane > 0 && lane < nbody) {
        const int b = lane;
        float xq[4] = {S.xquat[b][0], S.xquat[b][1], S.xquat[b][2], S.xquat[b][3]};
        float v[3], qi[4];
        qrot(v, xq, dm.body_ipos[b]);
        for (int k = 0; k < 3; k++) xip[k] = S.xpos[b][k] + v[k];
        qmul(qi, xq, dm.body_iquat[b]);
        q2m(ximat, qi);
        bmass = S.p_mass[b];
      }
      {
        float sx = wave_sum(bmass * xip[0]), sy = wave_sum(bmass * xip[1]), sz = wave_sum(bmass * xip[2]), sm = wave_sum(bmass);
        float inv = 1.f / fmaxf(sm, 1e-20f);
        if (lane == 0) { S.com[0] = sx * inv; S.com[1] = sy * inv; S.com[2] = sz * inv; }
      }
      WSYNC();
      const float com[3] = {S.com[0], S.com[1], S.com[2]};
      if (lane > 0 && lane < nbody) {
        const int b = lane;
        const float* I = dm.body_inertia[b];
        float dif[3] = {xip[0] - com[0], xip[1] - com[1], xip[2] - com[2]};
        float R[9];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c < 3; c++)
            R[3 * r + c] = ximat[3 * r] * I[0] * ximat[3 * c] + ximat[3 * r + 1] * I[1] * ximat[3 * c + 1] + ximat[3 * r + 2] * I[2] * ximat[3 * c + 2];
        float dd = dot3(dif, dif);
        cinert[0] = R[0] + bmass * (dd - dif[0] * dif[0]);
        cinert[1] = R[4] + bmass * (dd - dif[1] * dif[1]);
        cinert[2] = R[8] + bmass * (dd - dif[2] * dif[2]);
        cinert[3] = R[1] - bmass * dif[0] * dif[1];
        cinert[4] = R[2] - bmass * dif[0] * dif[2];
        cinert[5] = R[5] - bmass * dif[1] * dif[2];
        cinert[6] = bmass * dif[0]; cinert[7] = bmass * dif[1]; cinert[8] = bmass * dif[2];
        cinert[9] = bmass;
      }
      if (lane < nbody) {
#pragma unroll
        for (int k = 0; k < 10; k++) S.cin[lane][k] = cinert[k];
      }
      // cdof (lane = dof)
      float cd[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      int d_body = 0, d_parent = -1;
      if (lane < nv) {
        d_body = dm.dof_body[lane];
        d_parent = dm.dof_parent[lane];
        const int b = d_body, jt = dm.body_jtype[b], k = lane - dm.body_dadr[b];
        if (jt == CS_JNT_FREE) {
          if (k < 3) cd[3 + k] = 1.f;
          else {
            float m[9], xq[4] = {S.xquat[b][0], S.xquat[b][1], S.xquat[b][2], S.xquat[b][3]};
            q2m(m, xq);
            float axs[3] = {m[k - 3], m[3 + k - 3], m[6 + k - 3]};
            float off[3] = {com[0] - S.xpos[b][0], com[1] - S.xpos[b][1], com[2] - S.xpos[b][2]};
            cd[0] = axs[0]; cd[1] = axs[1]; cd[2] = axs[2];
            cross(cd + 3, axs, off);
          }
        } else {
          float axs[3] = {S.xax[b][0], S.xax[b][1], S.xax[b][2]};
          float off[3] = {com[0] - S.xanc[b][0], com[1] - S.xanc[b][1], com[2] - S.xanc[b][2]};
          cd[0] = axs[0]; cd[1] = axs[1]; cd[2] = axs[2];
          cross(cd + 3, axs, off);
        }
#pragma unroll
        for (int q = 0; q < 6; q++) S.cdof[lane][q] = cd[q];
      }
      for (int e = lane; e < TRI; e += 64) S.M[e] = 0.f;
      WSYNC();

      // =========================================================== mj_crb: composite inertia of the dof's body, M row
      if (lane < nv) {
        float crb[10];
#pragma unroll
        for (int k = 0; k < 10; k++) crb[k] = 0.f;
        unsigned sub_mask = dm.body_subtree[d_body];
        for (int c = 1; c < nbody; c++)
          if ((sub_mask >> c) & 1u) {
#pragma unroll
            for (int k = 0; k < 10; k++) crb[k] += S.cin[c][k];
          }
        float buf[6];
        mul_inert(buf, crb, cd);
        for (int j = lane; j >= 0; j = dm.dof_parent[j]) {
          float v = 0.f;
#pragma unroll
          for (int q = 0; q < 6; q++) v += S.cdof[j][q] * buf[q];
          if (j == lane) v += dm.dof_armature[lane];
          S.M[(lane * (lane + 1)) / 2 + j] = v;
        }
      }

      // =========================================================== mj_comVel + mj_rne (bias) + passive + smooth force
      float qv = lane < nv ? S.qvel[lane] : 0.f;
      if (lane < nv) {
        // velocity of the parent chain just before this dof (free joint: rotations see the translational part only)
        float cv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int b = d_body, jt = dm.body_jtype[b], k = lane - dm.body_dadr[b];
        float cdd[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (jt == CS_JNT_FREE) {
          if (k >= 3) {
            const int d0 = dm.body_dadr[b];
            cv[3] = S.qvel[d0]; cv[4] = S.qvel[d0 + 1]; cv[5] = S.qvel[d0 + 2];
          }
        } else {
          for (int j = d_parent; j >= 0; j = dm.dof_parent[j]) {
            float qj = S.qvel[j];
#pragma unroll
            for (int q = 0; q < 6; q++) cv[q] += S.cdof[j][q] * qj;
          }
        }
        if (!(jt == CS_JNT_FREE && k < 3)) {  // mju_crossMotion(cv, cdof)
          float a[3], bb[3], c[3];
          cross(a, cv, cd);
          cross(bb, cv, cd + 3);
          cross(c, cv + 3, cd);
          cdd[0] = a[0]; cdd[1] = a[1]; cdd[2] = a[2];
          cdd[3] = bb[0] + c[0]; cdd[4] = bb[1] + c[1]; cdd[5] = bb[2] + c[2];
        }
#pragma unroll
        for (int q = 0; q < 6; q++) S.cdd[lane][q] = cdd[q];
      }
      WSYNC();
      float cvel_b[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (lane > 0 && lane < nbody) {
        float cacc[6] = {0.f, 0.f, 0.f, -dm.gravity[0], -dm.gravity[1], -dm.gravity[2]};
        for (int j = dm.body_lastdof[lane]; j >= 0; j = dm.dof_parent[j]) {
          float qj = S.qvel[j];
#pragma unroll
          for (int q = 0; q < 6; q++) { cvel_b[q] += S.cdof[j][q] * qj; cacc[q] += S.cdd[j][q] * qj; }
        }
        float t1[6], t2[6], t3[6];
        mul_inert(t1, cinert, cacc);
        mul_inert(t2, cinert, cvel_b);
        {  // mju_crossForce(cvel, t2)
          float a[3], bb[3], c[3];
          cross(a, cvel_b, t2);
          cross(bb, cvel_b + 3, t2 + 3);
          cross(c, cvel_b, t2 + 3);
          t3[0] = a[0] + bb[0]; t3[1] = a[1] + bb[1]; t3[2] = a[2] + bb[2];
          t3[3] = c[0]; t3[4] = c[1]; t3[5] = c[2];
        }
#pragma unroll
        for (int q = 0; q < 6; q++) S.cfb[lane][q] = t1[q] + t3[q];
      }
      WSYNC();
      if (lane < nv) {
        float f[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        unsigned sub_mask = dm.body_subtree[d_body];
        for (int c = 1; c < nbody; c++)
          if ((sub_mask >> c) & 1u) {
#pragma unroll
            for (int q = 0; q < 6; q++) f[q] += S.cfb[c][q];
          }
        float bias = 0.f;
#pragma unroll
        for (int q = 0; q < 6; q++) bias += cd[q] * f[q];
        S.qbias[lane] = bias;
        S.qsm[lane] = -dm.dof_damping[lane] * qv - bias + S.qact[lane];
      }

      // sensors of this forward pass (framequat, gyro, velocimeter on the IMU site); uniform code
      {
        const int ib = dm.imu_body;
        float xq[4] = {S.xquat[ib][0], S.xquat[ib][1], S.xquat[ib][2], S.xquat[ib][3]};
        qmul(s_quat, xq, dm.imu_quat);
        qnorm(s_quat);
        float cv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int j = dm.body_lastdof[ib]; j >= 0; j = dm.dof_parent[j]) {
          float qj = S.qvel[j];
#pragma unroll
          for (int q = 0; q < 6; q++) cv[q] += S.cdof[j][q] * qj;
        }
        float sp[3], v[3], dif[3], t[3], lin[3], m[9];
        qrot(v, xq, dm.imu_pos);
        for (int k = 0; k < 3; k++) { sp[k] = S.xpos[ib][k] + v[k]; dif[k] = sp[k] - com[k]; }
        cross(t, dif, cv);
        for (int k = 0; k < 3; k++) lin[k] = cv[3 + k] - t[k];
        q2m(m, s_quat);
        for (int k = 0; k < 3; k++) {
          float g = m[k] * cv[0] + m[3 + k] * cv[1] + m[6 + k] * cv[2];
          float l = m[k] * lin[0] + m[3 + k] * lin[1] + m[6 + k] * lin[2];
          if (dm.gyro_cutoff > 0.f) g = fminf(dm.gyro_cutoff, fmaxf(-dm.gyro_cutoff, g));
          if (dm.vel_cutoff > 0.f) l = fminf(dm.vel_cutoff, fmaxf(-dm.vel_cutoff, l));
          s_gyro[k] = g; s_vel[k] = l;
        }
      }

      // =========================================================== collision: ground plane vs robot geoms (lane = geom)
      int ncon = 0;
      {
        float cp[4][3], cdst[4];
        int cnt = 0;
        bool mesh_near = false;
        const float n[3] = {0.f, 0.f, 1.f};
        if (lane < ngeom && dm.geom_ground[lane]) {
          const int g = lane, b = dm.geom_body[g], gt = dm.geom_type[g];
          float xq[4] = {S.xquat[b][0], S.xquat[b][1], S.xquat[b][2], S.xquat[b][3]};
          const float margin = dm.geom_margin[g];
          if (gt == CS_GEOM_MESH) {
            float v[3];
            qrot(v, xq, dm.geom_rcenter[g]);
            mesh_near = (S.xpos[b][2] + v[2] - dm.ground_pos[2] - dm.geom_rbound[g]) <= margin;
          } else {
            float v[3], pos[3], gq[4], mat[9];
            qrot(v, xq, dm.geom_pos[g]);
            for (int k = 0; k < 3; k++) pos[k] = S.xpos[b][k] + v[k];
            const float dist0 = pos[2] - dm.ground_pos[2];
            if (gt == CS_GEOM_SPHERE) {
              const float r = dm.geom_size[g][0];
              if (dist0 <= margin + r) {
                float dist = dist0 - r;
                cdst[0] = dist;
                cp[0][0] = pos[0]; cp[0][1] = pos[1]; cp[0][2] = pos[2] - (r + 0.5f * dist);
                cnt = 1;
              }
            } else if (gt == CS_GEOM_CYLINDER) {  // mjc_PlaneCylinder
              qmul(gq, xq, dm.geom_quat[g]);
              q2m(mat, gq);
              const float radius = dm.geom_size[g][0], half = dm.geom_size[g][1];
              float axis[3] = {mat[2], mat[5], mat[8]};
              float prjaxis = axis[2];
              if (prjaxis > 0.f) { axis[0] = -axis[0]; axis[1] = -axis[1]; axis[2] = -axis[2]; prjaxis = -prjaxis; }
              float vec[3] = {axis[0] * prjaxis, axis[1] * prjaxis, axis[2] * prjaxis - 1.f};
              float len = sqrtf(dot3(vec, vec));
              if (len < 1e-12f) { vec[0] = mat[0] * radius; vec[1] = mat[3] * radius; vec[2] = mat[6] * radius; }
              else { float s = radius / len; vec[0] *= s; vec[1] *= s; vec[2] *= s; }
              const float prjvec = vec[2];
              axis[0] *= half; axis[1] *= half; axis[2] *= half;
              prjaxis *= half;
              float dist = dist0 + prjaxis + prjvec;
              if (dist <= margin) {
                cdst[0] = dist;
                for (int k = 0; k < 3; k++) cp[0][k] = pos[k] + vec[k] + axis[k] - n[k] * dist * 0.5f;
                cnt = 1;
                dist = dist0 - prjaxis + prjvec;
                if (dist <= margin) {
                  cdst[cnt] = dist;
                  for (int k = 0; k < 3; k++) cp[cnt][k] = pos[k] + vec[k] - axis[k] - n[k] * dist * 0.5f;
                  cnt++;
                }
                float vec1[3];
                cross(vec1, vec, axis);
                float s1 = radius * 0.8660254037844386f * rsqrtf(fmaxf(dot3(vec1, vec1), 1e-30f));
                vec1[0] *= s1; vec1[1] *= s1; vec1[2] *= s1;
                const float prjvec1 = vec1[2];
                dist = dist0 + prjaxis - prjvec * 0.5f + prjvec1;
                if (dist <= margin) {
                  cdst[cnt] = dist;
                  for (int k = 0; k < 3; k++) cp[cnt][k] = pos[k] + vec1[k] + axis[k] - vec[k] * 0.5f - n[k] * dist * 0.5f;
                  cnt++;
                }
                dist = dist0 + prjaxis - prjvec * 0.5f - prjvec1;
                if (dist <= margin) {
                  cdst[cnt] = dist;
                  for (int k = 0; k < 3; k++) cp[cnt][k] = pos[k] - vec1[k] + axis[k] - vec[k] * 0.5f - n[k] * dist * 0.5f;
                  cnt++;
                }
              }
            } else if (gt == CS_GEOM_BOX) {  // mjc_PlaneBox
              qmul(gq, xq, dm.geom_quat[g]);
              q2m(mat, gq);
              for (int i = 0; i < 8 && cnt < 4; i++) {
                float vx = (i & 1) ? dm.geom_size[g][0] : -dm.geom_size[g][0], vy = (i & 2) ? dm.geom_size[g][1] : -dm.geom_size[g][1],
                      vz = (i & 4) ? dm.geom_size[g][2] : -dm.geom_size[g][2];
                float c[3] = {mat[0] * vx + mat[1] * vy + mat[2] * vz, mat[3] * vx + mat[4] * vy + mat[5] * vz, mat[6] * vx + mat[7] * vy + mat[8] * vz};
                float ld = c[2];
                if (dist0 + ld > margin || ld > 0.f) continue;
                float dist = dist0 + ld;
                cdst[cnt] = dist;
                for (int k = 0; k < 3; k++) cp[cnt][k] = pos[k] + c[k] - n[k] * dist * 0.5f;
                cnt++;
              }
            }
          }
        }
        // compaction in (geom, slot) order
        int off = 0, total = 0;
#pragma unroll
        for (int s = 0; s < 4; s++) {
          unsigned long long mk = __ballot(cnt > s);
          off += __popcll(mk & lanemask_lt(lane));
          total += __popcll(mk);
        }
        for (int s = 0; s < cnt; s++) {
          int slot = off + s;
          if (slot < MC) {
            S.cdist[slot] = cdst[s];
            S.cgeom[slot] = lane;
            for (int k = 0; k < 3; k++) { S.cpos[slot][k] = cp[s][k]; S.cnrm[slot][k] = n[k]; }
          }
        }
        ncon = total;
        // convex meshes near the ground: all lanes scan the hull (mjc_PlaneConvex)
        unsigned long long mm = __ballot(mesh_near);
        while (mm) {
          const int g = __builtin_ctzll(mm);
          mm &= mm - 1;
          const int b = dm.geom_body[g], adr = dm.geom_hulladr[g], num = dm.geom_hullnum[g];
          float xq[4] = {S.xquat[b][0], S.xquat[b][1], S.xquat[b][2], S.xquat[b][3]}, m[9];
          q2m(m, xq);
          const float ln[3] = {m[6], m[7], m[8]};  // R^T n for n = +z
          const float offz = S.xpos[b][2] - dm.ground_pos[2];
          const float margin = dm.geom_margin[g];
          float best = 3.0e38f;
          int besti = -1;
          for (int i = lane; i < num; i += 64) {
            const float* v = A.hull_vert + 3 * (adr + i);
            float dist = offz + ln[0] * v[0] + ln[1] * v[1] + ln[2] * v[2];
            if (dist < best) { best = dist; besti = i; }
          }
          float bmin = wave_min(best);
          unsigned long long who = __ballot(best == bmin && besti >= 0);
          if (bmin > margin || who == 0ull) continue;
          int src = __builtin_ctzll(who);
          // lowest vertex index among ties, like a sequential scan
          int bi = besti;
          {
            int cand = (best == bmin && besti >= 0) ? besti : 0x7fffffff;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
            bi = cand;
          }
          (void)src;
          int added = 0;
          for (int pass = 0; pass < 2; pass++) {
            const int lo = pass ? A.hull_adr[adr + bi] : 0, hi = pass ? A.hull_adr[adr + bi + 1] : 1;
            for (int e = lo; e < hi && added < 4; e++) {
              const int i = pass ? A.hull_nbr[e] : bi;
              const float* v = A.hull_vert + 3 * (adr + i);
              float dist = offz + ln[0] * v[0] + ln[1] * v[1] + ln[2] * v[2];
              if (dist > margin) continue;
              if (ncon < MC && lane == 0) {
                float w[3] = {m[0] * v[0] + m[1] * v[1] + m[2] * v[2], m[3] * v[0] + m[4] * v[1] + m[5] * v[2], m[6] * v[0] + m[7] * v[1] + m[8] * v[2]};
                S.cdist[ncon] = dist;
                S.cgeom[ncon] = g;
                for (int k = 0; k < 3; k++) { S.cpos[ncon][k] = S.xpos[b][k] + w[k] - n[k] * dist * 0.5f; S.cnrm[ncon][k] = n[k]; }
              }
              ncon++;
              added++;
            }
          }
        }
      }

      // =========================================================== constraint rows (lane = row)
      const int ne = 3 * dm.neq, nf = dm.nfric;
      // joint limits: compaction over bodies
      int nl = 0;
      {
        bool lo_v = false, hi_v = false;
        float dlo = 0.f, dhi = 0.f;
        if (lane > 0 && lane < nbody && dm.body_jtype[lane] == CS_JNT_HINGE && dm.jnt_limited[lane]) {
          float q = S.qpos[dm.body_qadr[lane]];
          dlo = q - dm.jnt_range[lane][0];
          dhi = dm.jnt_range[lane][1] - q;
          lo_v = dlo < dm.jnt_margin[lane];
          hi_v = dhi < dm.jnt_margin[lane];
        }
        unsigned long long ml = __ballot(lo_v), mh = __ballot(hi_v);
        int rlo = __popcll(ml & lanemask_lt(lane)), rhi = __popcll(ml) + __popcll(mh & lanemask_lt(lane));
        nl = __popcll(ml) + __popcll(mh);
        if (lo_v && rlo < MAXROW) { S.lim_body[rlo] = lane; S.lim_sign[rlo] = 1.f; S.lim_dist[rlo] = dlo; }
        if (hi_v && rhi < MAXROW) { S.lim_body[rhi] = lane; S.lim_sign[rhi] = -1.f; S.lim_dist[rhi] = dhi; }
      }
      {
        int room = (MAXROW - ne - nf - nl) / 4;
        if (room < 0) room = 0;
        if (ncon > room) ncon = room;
        if (ncon > MC) ncon = MC;
      }
      const int ngen = ne + 4 * ncon;       // general rows (dense J): equality + contact
      const int nefc = ngen + nf + nl;      // then unit rows: frictionloss, limits
      WSYNC();

      int rtype = RT_NONE, rdof = 0;
      float rsign = 1.f, rpos = 0.f, rmargin = 0.f, rfloss = 0.f, rdiagA = 0.f, rmu = 0.f;
      float rsolref[2] = {0.02f, 1.f}, rsolimp[5] = {0.9f, 0.95f, 0.001f, 0.5f, 2.f};
      if (lane < ngen) {
        float* Jr = S.J[lane];
        for (int d = 0; d < NV; d++) Jr[d] = 0.f;
        if (lane < ne) {
          rtype = RT_EQ;
          const int e = lane / 3, comp = lane - 3 * e;
          const int b1 = dm.eq_body1[e], b2 = dm.eq_body2[e];
          float dir[3] = {comp == 0 ? 1.f : 0.f, comp == 1 ? 1.f : 0.f, comp == 2 ? 1.f : 0.f};
          float p1[3], p2[3], v[3], q1[4] = {S.xquat[b1][0], S.xquat[b1][1], S.xquat[b1][2], S.xquat[b1][3]},
                                    q2[4] = {S.xquat[b2][0], S.xquat[b2][1], S.xquat[b2][2], S.xquat[b2][3]};
          qrot(v, q1, dm.eq_anchor1[e]);
          for (int k = 0; k < 3; k++) p1[k] = S.xpos[b1][k] + v[k];
          qrot(v, q2, dm.eq_anchor2[e]);
          for (int k = 0; k < 3; k++) p2[k] = S.xpos[b2][k] + v[k];
          rpos = p1[comp] - p2[comp];
          float off[3], od[3];
          for (int k = 0; k < 3; k++) off[k] = p1[k] - com[k];
          cross(od, off, dir);
          for (int j = dm.body_lastdof[b1]; j >= 0; j = dm.dof_parent[j])
            Jr[j] += dir[0] * S.cdof[j][3] + dir[1] * S.cdof[j][4] + dir[2] * S.cdof[j][5] + od[0] * S.cdof[j][0] + od[1] * S.cdof[j][1] + od[2] * S.cdof[j][2];
          for (int k = 0; k < 3; k++) off[k] = p2[k] - com[k];
          cross(od, off, dir);
          for (int j = dm.body_lastdof[b2]; j >= 0; j = dm.dof_parent[j])
            Jr[j] -= dir[0] * S.cdof[j][3] + dir[1] * S.cdof[j][4] + dir[2] * S.cdof[j][5] + od[0] * S.cdof[j][0] + od[1] * S.cdof[j][1] + od[2] * S.cdof[j][2];
          rdiagA = S.p_binvw[b1] + S.p_binvw[b2];
          for (int k = 0; k < 2; k++) rsolref[k] = dm.eq_solref[e][k];
          for (int k = 0; k < 5; k++) rsolimp[k] = dm.eq_solimp[e][k];
        } else {
          rtype = RT_CONTACT;
          const int c = (lane - ne) >> 2, edge = (lane - ne) & 3;
          const int g = S.cgeom[c], b = dm.geom_body[g];
          const float mu = S.p_gmu[g];
          float nrm[3] = {S.cnrm[c][0], S.cnrm[c][1], S.cnrm[c][2]}, t1[3], t2[3];
          make_frame(nrm, t1, t2);
          const float* tk = (edge >> 1) ? t2 : t1;
          const float sg = (edge & 1) ? -mu : mu;
          float dir[3] = {nrm[0] + sg * tk[0], nrm[1] + sg * tk[1], nrm[2] + sg * tk[2]};
          float off[3] = {S.cpos[c][0] - com[0], S.cpos[c][1] - com[1], S.cpos[c][2] - com[2]}, od[3];
          cross(od, off, dir);
          for (int j = dm.body_lastdof[b]; j >= 0; j = dm.dof_parent[j])
            Jr[j] = dir[0] * S.cdof[j][3] + dir[1] * S.cdof[j][4] + dir[2] * S.cdof[j][5] + od[0] * S.cdof[j][0] + od[1] * S.cdof[j][1] + od[2] * S.cdof[j][2];
          rpos = S.cdist[c];
          rmargin = dm.geom_includemargin[g];
          rmu = mu;
          rdiagA = S.p_binvw[b] * (1.f + mu * mu);
          for (int k = 0; k < 2; k++) rsolref[k] = dm.geom_solref[g][k];
          for (int k = 0; k < 5; k++) rsolimp[k] = dm.geom_solimp[g][k];
        }
      }
      // (friction and limit rows are filled below, after the friction dof list is known)
      if (lane >= ngen && lane < nefc) {
        if (lane < ngen + nf) {
          rtype = RT_FRIC;
          rdof = dm.fric_dof[lane - ngen];  // model-level list; a per-env value of zero leaves the row inert
          rfloss = S.p_floss[rdof];
          rdiagA = S.p_dinvw[rdof];
          for (int k = 0; k < 2; k++) rsolref[k] = dm.dof_solref[rdof][k];
          for (int k = 0; k < 5; k++) rsolimp[k] = dm.dof_solimp[rdof][k];
          if (!(rfloss > 0.f)) rtype = RT_NONE;
        } else {
          rtype = RT_LIMIT;
          const int li = lane - ngen - nf;
          const int b = S.lim_body[li];
          rdof = dm.body_dadr[b];
          rsign = S.lim_sign[li];
          rpos = S.lim_dist[li];
          rmargin = dm.jnt_margin[b];
          rdiagA = S.p_dinvw[rdof];
          for (int k = 0; k < 2; k++) rsolref[k] = dm.jnt_solref[b][k];
          for (int k = 0; k < 5; k++) rsolimp[k] = dm.jnt_solimp[b][k];
        }
      }
      // KBIP, R, D, aref
      float rD = 0.f, rR = 1.f, raref = 0.f;
      if (rtype != RT_NONE) {
        float imp = impedance(rsolimp, rpos, rmargin);
        float dmax = fminf(MAXIMP, fmaxf(MINIMP, rsolimp[1]));
        float K, B;
        if (rsolref[0] > 0.f) {
          float tc = fmaxf(rsolref[0], 2.f * h), dr = rsolref[1];
          K = 1.f / fmaxf(MINVAL, dmax * dmax * tc * tc * dr * dr);
          B = 2.f / fmaxf(MINVAL, dmax * tc);
        } else { K = -rsolref[0] / fmaxf(MINVAL, dmax * dmax); B = -rsolref[1] / fmaxf(MINVAL, dmax); }
        if (rtype == RT_FRIC) K = 0.f;
        rR = fmaxf(MINVAL, (1.f - imp) * rdiagA / imp);
        if (rtype == RT_CONTACT) { float mu = rmu * rsqrtf(fmaxf(MINVAL, dm.impratio)); rR = 2.f * mu * mu * rR; }
        rD = 1.f / rR;
        float vel;
        if (rtype == RT_EQ || rtype == RT_CONTACT) {
          vel = 0.f;
          const float* Jr = S.J[lane];
          for (int d = 0; d < NV; d++) vel += Jr[d] * S.qvel[d];
        } else vel = (rtype == RT_LIMIT ? rsign : 1.f) * S.qvel[rdof];
        raref = -B * vel - K * imp * (rpos - rmargin);
      }
      if (rtype == RT_FRIC) rsign = 1.f;
      WSYNC();

      // =========================================================== Newton solver (mj_solNewton), warm-started from qacc
      float Jaref = 0.f, Jv = 0.f, Ma = 0.f;
      float a_row[NV];
      float dinv = 1.f;
      int e_idx[EPL], e_a[EPL], e_b[EPL];
#pragma unroll
      for (int t = 0; t < EPL; t++) {
        int e = lane + 64 * t;
        e_idx[t] = e < TRI ? e : -1;
        e_a[t] = e < TRI ? dm.tri_row[e] : 0;
        e_b[t] = e < TRI ? dm.tri_col[e] : 0;
      }
      auto mulM = [&](const float* v) -> float {  // (M v)[lane]
        float s = 0.f;
        if (lane < NV)
          for (int k = 0; k < NV; k++) s += S.M[tri_idx(lane, k)] * v[k];
        return s;
      };
      auto rowdot = [&](const float* v) -> float {  // J[row] . v for this lane's row
        float s = 0.f;
        if (rtype == RT_EQ || rtype == RT_CONTACT) {
          const float* Jr = S.J[lane];
          for (int d = 0; d < NV; d++) s += Jr[d] * v[d];
        } else if (rtype != RT_NONE) s = rsign * v[rdof];
        return s;
      };
      Jaref = rowdot(S.qacc) - raref;
      if (rtype == RT_NONE) Jaref = 0.f;
      Ma = mulM(S.qacc);
      float qacc_l = lane < NV ? S.qacc[lane] : 0.f;
      const float qsm_l = lane < NV ? S.qsm[lane] : 0.f;
      float cost = 0.f, gauss = 0.f, grad_l = 0.f;
      const float scale = 1.f / (meaninertia * (float)(NV > 1 ? NV : 1));
      int niter = 0;
      const int maxiter = min(dm.iterations, A.max_newton);

      auto update_constraint = [&]() {
        // mj_constraintUpdate: force, active set, cost
        float f = 0.f, c = 0.f, dact = 0.f;
        if (rtype == RT_EQ) { f = -rD * Jaref; c = 0.5f * rD * Jaref * Jaref; dact = rD; }
        else if (rtype == RT_FRIC) {
          float Rf = rR * rfloss;
          if (Jaref <= -Rf) { f = rfloss; c = -0.5f * Rf * rfloss - rfloss * Jaref; }
          else if (Jaref >= Rf) { f = -rfloss; c = -0.5f * Rf * rfloss + rfloss * Jaref; }
          else { f = -rD * Jaref; c = 0.5f * rD * Jaref * Jaref; dact = rD; }
        } else if (rtype == RT_LIMIT || rtype == RT_CONTACT) {
          if (Jaref < 0.f) { f = -rD * Jaref; c = 0.5f * rD * Jaref * Jaref; dact = rD; }
        }
        S.rowf[lane] = f;
        S.rowD[lane] = dact;
        if (lane < NV) { S.dofD[lane] = 0.f; S.qcon[lane] = 0.f; }
        WSYNC();
        if (rtype == RT_FRIC || rtype == RT_LIMIT) {
          atomicAdd(&S.dofD[rdof], dact);
          atomicAdd(&S.qcon[rdof], rsign * f);
        }
        WSYNC();
        float qc = 0.f;
        if (lane < NV) {
          qc = S.qcon[lane];
          for (int r = 0; r < ngen; r++) qc += S.J[r][lane] * S.rowf[r];
          S.qcon[lane] = qc;
        }
        gauss = wave_sum(lane < NV ? (0.5f * Ma - qsm_l) * qacc_l : 0.f);
        cost = wave_sum(c) + gauss;
        grad_l = lane < NV ? Ma - qsm_l - qc : 0.f;
      };

      auto update_search = [&]() {
        // Hessian H = M + J^T diag(D_active) J, entry-parallel over the packed lower triangle
        float hacc[EPL];
#pragma unroll
        for (int t = 0; t < EPL; t++) hacc[t] = e_idx[t] >= 0 ? S.M[e_idx[t]] : 0.f;
        for (int r = 0; r < ngen; r++) {
          float dr = rfl(S.rowD[r]);
          if (dr == 0.f) continue;
          const float* Jr = S.J[r];
#pragma unroll
          for (int t = 0; t < EPL; t++) hacc[t] += dr * Jr[e_a[t]] * Jr[e_b[t]];
        }
#pragma unroll
        for (int t = 0; t < EPL; t++)
          if (e_idx[t] >= 0) S.H[e_idx[t]] = hacc[t];
        WSYNC();
        if (lane < NV) {
#pragma unroll
          for (int k = 0; k < NV; k++) a_row[k] = S.H[tri_idx(lane, k)];
        } else {
#pragma unroll
          for (int k = 0; k < NV; k++) a_row[k] = 0.f;
        }
        // unit rows (frictionloss, limits) only touch the diagonal
        float dd = lane < NV ? S.dofD[lane] : 0.f;
#pragma unroll
        for (int k = 0; k < NV; k++) a_row[k] += (lane == k) ? dd : 0.f;
        chol_regs<NV>(a_row, dinv, lane);
        float mg = chol_solve_regs<NV>(a_row, dinv, grad_l, lane);
        if (lane < NV) S.sr[lane] = -mg;
        WSYNC();
      };

      update_constraint();
      float gradnorm = sqrtf(wave_sum(grad_l * grad_l));
      if (A.mode == MODE_DEBUG && A.dbg != nullptr) {
        // dump position/velocity-stage intermediates before the solve
        float* D = A.dbg;
        if (lane == 0) { D[0] = (float)ncon; D[1] = (float)nefc; D[2] = (float)ne; D[3] = (float)nf; D[4] = (float)nl; D[5] = cost; D[6] = gradnorm; D[7] = (float)ngen; }
        if (lane < nbody) { for (int k = 0; k < 3; k++) D[64 + lane * 3 + k] = S.xpos[lane][k]; for (int k = 0; k < 4; k++) D[192 + lane * 4 + k] = S.xquat[lane][k]; }
        for (int e = lane; e < TRI; e += 64) D[512 + e] = S.M[e];
        if (lane < NV) { D[1100 + lane] = S.qsm[lane]; D[1140 + lane] = S.qbias[lane]; for (int q = 0; q < 6; q++) D[1200 + lane * 6 + q] = S.cdof[lane][q]; }
        D[1400 + lane] = (float)rtype; D[1464 + lane] = rD; D[1528 + lane] = raref; D[1592 + lane] = rpos; D[1656 + lane] = Jaref;
        if (lane < ngen) for (int d = 0; d < NV; d++) D[2048 + lane * NV + d] = S.J[lane][d];
        if (lane < MC) { D[1720 + lane] = lane < ncon ? S.cdist[lane] : 0.f; for (int k = 0; k < 3; k++) D[1740 + lane * 3 + k] = lane < ncon ? S.cpos[lane][k] : 0.f; }
      }
      while (niter < maxiter) {
        if (scale * gradnorm < A.tol32) break;
        update_search();
        // ---- exact line search on the piecewise-quadratic cost (PrimalSearch)
        const float sr_l = lane < NV ? S.sr[lane] : 0.f;
        const float Mv = mulM(S.sr);
        Jv = rowdot(S.sr);
        const float snorm = sqrtf(wave_sum(sr_l * sr_l));
        if (!(snorm >= 1e-20f)) break;
        const float qG1 = wave_sum(sr_l * (Ma - qsm_l)), qG2 = wave_sum(0.5f * sr_l * Mv);
        const float q0 = 0.5f * rD * Jaref * Jaref, q1 = rD * Jaref * Jv, q2 = 0.5f * rD * Jv * Jv;
        const float gtol = A.tol32 * dm.ls_tolerance * snorm / scale;
        struct Pnt { float alpha, cost, d0, d1; };
        auto eval = [&](float alpha) -> Pnt {
          float x = Jaref + alpha * Jv, c0 = 0.f, c1 = 0.f, c2 = 0.f;
          if (rtype == RT_EQ) { c0 = q0; c1 = q1; c2 = q2; }
          else if (rtype == RT_FRIC) {
            float Rf = rR * rfloss;
            if (x > -Rf && x < Rf) { c0 = q0; c1 = q1; c2 = q2; }
            else if (x <= -Rf) { c0 = rfloss * (-0.5f * Rf - Jaref); c1 = -rfloss * Jv; }
            else { c0 = rfloss * (-0.5f * Rf + Jaref); c1 = rfloss * Jv; }
          } else if (rtype == RT_LIMIT || rtype == RT_CONTACT) {
            if (x < 0.f) { c0 = q0; c1 = q1; c2 = q2; }
          }
          float C0 = wave_sum(c0) + gauss, C1 = wave_sum(c1) + qG1, C2 = wave_sum(c2) + qG2;
          Pnt p;
          p.alpha = alpha;
          p.cost = alpha * alpha * C2 + alpha * C1 + C0;
          p.d0 = 2.f * alpha * C2 + C1;
          p.d1 = 2.f * C2;
          if (!(p.d1 > 0.f)) p.d1 = 1e-15f;
          return p;
        };
        float alpha = 0.f;
        {
          int lsit = 0;
          const int maxls = min(dm.ls_iterations, 24);
          Pnt p0 = eval(0.f); lsit++;
          Pnt p1 = eval(p0.alpha - p0.d0 / p0.d1); lsit++;
          if (p0.cost < p1.cost) p1 = p0;
          bool done = false;
          if (fabsf(p1.d0) < gtol) { alpha = p1.alpha; done = true; }
          if (!done) {
            const float dir = p1.d0 < 0.f ? 1.f : -1.f;
            Pnt p2 = p1;
            bool p2update = false;
            while (p1.d0 * dir <= -gtol && lsit < maxls) {
              p2 = p1; p2update = true;
              p1 = eval(p1.alpha - p1.d0 / p1.d1); lsit++;
              if (fabsf(p1.d0) < gtol) { alpha = p1.alpha; done = true; break; }
            }
            if (!done) {
              if (lsit >= maxls || !p2update) { alpha = p1.alpha; done = true; }
            }
            if (!done) {
              Pnt p2next = p1;
              Pnt p1next = eval(p1.alpha - p1.d0 / p1.d1); lsit++;
              while (lsit < maxls) {
                Pnt pmid = eval(0.5f * (p1.alpha + p2.alpha)); lsit++;
                Pnt cand[3] = {p1next, p2next, pmid};
                int best = -1;
                float bestcost = 0.f;
                for (int i = 0; i < 3; i++)
                  if (fabsf(cand[i].d0) < gtol && (best == -1 || cand[i].cost < bestcost)) { best = i; bestcost = cand[i].cost; }
                if (best >= 0) { alpha = cand[best].alpha; done = true; break; }
                int b1 = 0, b2 = 0;
                for (int i = 0; i < 3; i++) {
                  if (p1.d0 < 0.f && cand[i].d0 < 0.f && p1.d0 < cand[i].d0) { p1 = cand[i]; b1 = 1; }
                  else if (p1.d0 > 0.f && cand[i].d0 > 0.f && p1.d0 > cand[i].d0) { p1 = cand[i]; b1 = 2; }
                }
                if (b1) { p1next = eval(p1.alpha - p1.d0 / p1.d1); lsit++; }
                for (int i = 0; i < 3; i++) {
                  if (p2.d0 < 0.f && cand[i].d0 < 0.f && p2.d0 < cand[i].d0) { p2 = cand[i]; b2 = 1; }
                  else if (p2.d0 > 0.f && cand[i].d0 > 0.f && p2.d0 > cand[i].d0) { p2 = cand[i]; b2 = 2; }
                }
                if (b2) { p2next = eval(p2.alpha - p2.d0 / p2.d1); lsit++; }
                if (!b1 && !b2) { alpha = pmid.alpha; done = true; break; }
              }
              if (!done) {
                if (p1.cost <= p2.cost && p1.cost < p0.cost) alpha = p1.alpha;
                else if (p2.cost <= p1.cost && p2.cost < p0.cost) alpha = p2.alpha;
                else alpha = 0.f;
              }
            }
          }
        }
        alpha = rfl(alpha);
        if (alpha == 0.f) break;
        // ---- move
        qacc_l += alpha * sr_l;
        Ma += alpha * Mv;
        Jaref += alpha * Jv;
        if (lane < NV) S.qacc[lane] = qacc_l;
        const float oldcost = cost;
        update_constraint();
        gradnorm = sqrtf(wave_sum(grad_l * grad_l));
        niter++;
        const float improvement = scale * (oldcost - cost);
        if (improvement < A.tol32) break;
      }
      if (A.mode == MODE_DEBUG && A.dbg != nullptr) {
        float* D = A.dbg;
        if (lane == 0) { D[8] = (float)niter; D[9] = cost; D[10] = gradnorm; }
        if (lane < NV) { D[1000 + lane] = qacc_l; D[1040 + lane] = S.qcon[lane]; }
        D[1800 + lane] = S.rowf[lane];
        if (lane < 4) D[16 + lane] = s_quat[lane];
        if (lane < 3) { D[20 + lane] = s_gyro[lane]; D[24 + lane] = s_vel[lane]; }
      }

      // =========================================================== mj_implicit (implicitfast) + mj_advance
      {
        if (lane < NV) {
#pragma unroll
          for (int k = 0; k < NV; k++) a_row[k] = S.M[tri_idx(lane, k)];
        } else {
#pragma unroll
          for (int k = 0; k < NV; k++) a_row[k] = 0.f;
        }
        const float hd = lane < NV ? h * dm.dof_damping[lane] : 0.f;
#pragma unroll
        for (int k = 0; k < NV; k++) a_row[k] += (lane == k) ? hd : 0.f;
        chol_regs<NV>(a_row, dinv, lane);
        const float rhs = lane < NV ? qsm_l + S.qcon[lane] : 0.f;
        const float qa = chol_solve_regs<NV>(a_row, dinv, rhs, lane);
        WSYNC();
        if (A.mode != MODE_DEBUG) {
          if (lane < NV) {
            qv += h * qa;
            S.qvel[lane] = qv;
          }
          WSYNC();
          if (lane > 0 && lane < nbody) {
            const int jt = dm.body_jtype[lane];
            if (jt == CS_JNT_HINGE) S.qpos[dm.body_qadr[lane]] += h * S.qvel[dm.body_dadr[lane]];
            else if (jt == CS_JNT_FREE) {
              const int qa0 = dm.body_qadr[lane], da = dm.body_dadr[lane];
              for (int k = 0; k < 3; k++) S.qpos[qa0 + k] += h * S.qvel[da + k];
              float w[3] = {S.qvel[da + 3], S.qvel[da + 4], S.qvel[da + 5]};
              float wn = sqrtf(dot3(w, w));
              float q[4] = {S.qpos[qa0 + 3], S.qpos[qa0 + 4], S.qpos[qa0 + 5], S.qpos[qa0 + 6]};
              qnorm(q);
              if (wn > 1e-20f) {
                float sn, cs;
                sincosf(0.5f * h * wn, &sn, &cs);
                float ql[4] = {cs, w[0] / wn * sn, w[1] / wn * sn, w[2] / wn * sn};
                qmul(q, q, ql);
              }
              for (int k = 0; k < 4; k++) S.qpos[qa0 + 3 + k] = q[k];
            }
          }
          // qacc (the solver's) stays in S.qacc as next substep's warm start
          WSYNC();
        } else if (A.dbg != nullptr) {
          if (lane < NV) A.dbg[1080 + lane] = qa;
        }
      }
    }  // substeps
    if (A.mode == MODE_DEBUG) return;

    // ---- mj_checkPos/Vel: non-finite state -> this env is reset (MuJoCo resets the data and warns)
    {
      bool nf_ = false;
      if (lane < nq) nf_ = !(fabsf(S.qpos[lane]) < 1e10f);
      if (lane < nv) nf_ = nf_ || !(fabsf(S.qvel[lane]) < 1e10f) || !(fabsf(S.qacc[lane]) < 1e10f);
      bad = __ballot(nf_) != 0ull;
    }
    if (sim_step == ob.max_sim_step) truncated = 1;
    if (bad) terminated = 1;
    // (term_mode 1: cfrc_ext test, flamingo_p_v3 — evaluated by the caller of a later round)
    do_reset = bad || ((terminated || truncated) && ob.auto_reset);
  }

  // =============================================================== reset_model (flamingo_light_v1.py:209-232)
  int nan_resets = meta[4];
  if (bad) nan_resets++;
  if (do_reset) {
    if (lane < nq) S.qpos[lane] = dm.init_qpos[lane];
    if (lane < nv) { S.qvel[lane] = 0.f; S.qacc[lane] = 0.f; }
    WSYNC();
    if (lane < dm.init_noise_nq) {
      unsigned rnd[4];
      philox(k0, k1, step_count, 2u, (unsigned)lane, 0u, rnd);
      S.qpos[dm.init_noise_qadr[lane]] += ob.init_noise * (2.f * u01(rnd[0]) - 1.f);
    }
    WSYNC();
    // sensors of the mj_forward at the reset state: zero velocity, IMU orientation from the base quaternion
    {
      const int ib = dm.imu_body;
      (void)ib;
      float xq[4] = {S.qpos[3], S.qpos[4], S.qpos[5], S.qpos[6]};
      qnorm(xq);
      qmul(s_quat, xq, dm.imu_quat);
      qnorm(s_quat);
      for (int k = 0; k < 3; k++) { s_gyro[k] = 0.f; s_vel[k] = 0.f; }
    }
    raw_action = 0.f;
    prev_action = 0.f;
    tq_lane = 0.f;
    has_prev = 0;
    sim_step = 0;
  }

  // =============================================================== _get_obs + _build_state + _apply_command_inplace
  {
    float m[9];
    q2m(m, s_quat);
    const float pg[3] = {-m[6], -m[7], -m[8]};  // R^T (0,0,-1)
    const bool fill = do_reset;                 // reset fills every stack row with the first frame
    if (lane < nu) S.act[lane] = raw_action;
    if (lane < CS_MAXCMD) S.cmd[lane] = applied_cmd;
    WSYNC();
    float* so = A.state_out + (size_t)env * ob.state_dim;
    const int sd = ob.stacked_dim, S_ = ob.stack_size;
    for (int e = lane; e < ob.frame_dim; e += 64) {
      const int f = ob.el_field[e], idx = ob.el_index[e];
      float val = 0.f;
      switch (f) {
        case CS_OBS_DOF_POS: val = S.qpos[dm.obs_qadr[idx]] * dm.obs_qgear[idx]; break;
        case CS_OBS_DOF_VEL: val = S.qvel[dm.obs_dadr[idx]] * dm.obs_dgear[idx]; break;
        case CS_OBS_ANG_VEL: val = s_gyro[idx]; break;
        case CS_OBS_LIN_VEL: val = s_vel[idx]; break;
        case CS_OBS_PROJ_GRAVITY: val = pg[idx]; break;
        case CS_OBS_LAST_ACTION: val = S.act[idx]; break;
        default: val = 0.f; break;
      }
      if (ob.noise_enabled && f != CS_OBS_LAST_ACTION && f != CS_OBS_COMMAND) {
        // truncated Gaussian by inverse CDF (scipy.stats.truncnorm.rvs, noise_generator_utils.py:22-28)
        unsigned rnd[4];
        philox(k0, k1, step_count, 1u, (unsigned)e, 0u, rnd);
        const float mean = ob.noise_mean[f], sd_ = ob.noise_std[f];
        const float ca = normcdff((ob.noise_lower[f] - mean) / sd_), cb = normcdff((ob.noise_upper[f] - mean) / sd_);
        float z = normcdfinvf(ca + u01(rnd[0]) * (cb - ca));
        float nz = fminf(ob.noise_upper[f], fmaxf(ob.noise_lower[f], mean + sd_ * z));
        val += nz;
      }
      float vs;
      const int interval = ob.el_interval[e];
      float* cache = rec + lay.s_cache + e;
      if (f == CS_OBS_COMMAND) vs = 0.f;
      else if (sim_step == 0 || (sim_step % interval) == 0) { vs = val * ob.el_scale[e]; *cache = vs; }
      else vs = *cache;
      const bool is_cmd = f == CS_OBS_COMMAND;
      if (e < sd) {
        float* st = rec + lay.s_stack;
        for (int k = S_ - 1; k >= 1; k--) {
          float o = fill ? vs : st[(k - 1) * sd + e];
          st[k * sd + e] = o;
          so[k * sd + e] = is_cmd ? S.cmd[idx] : o;
        }
        st[e] = vs;
        so[e] = is_cmd ? S.cmd[idx] : vs;
      } else {
        so[S_ * sd + (e - sd)] = is_cmd ? S.cmd[idx] : vs;
      }
    }
  }

  // =============================================================== info / flags / state write-back
  if (A.mode == MODE_STEP) {
    if (A.info != nullptr) {
      float* inf = A.info + (size_t)env * ob.info_dim;
      float dsq = lane < nu ? (raw_action - prev_action) * (raw_action - prev_action) : 0.f;
      float rmse = sqrtf(wave_sum(dsq) / (float)nu);
      if (lane == 0) { inf[0] = rmse; inf[1] = s_vel[0]; inf[2] = s_vel[1]; inf[3] = s_gyro[2]; }
      if (lane < nu) { inf[4 + lane] = tq_lane; inf[4 + nu + lane] = raw_action * dm.ctl_scale[lane]; }
      if (lane < dm.ninfo_state)
        inf[4 + 2 * nu + lane] = (dm.info_kind[lane] == 0 ? S.qpos[dm.info_adr[lane]] : S.qvel[dm.info_adr[lane]]) * dm.info_gear[lane];
    }
    if (lane == 0) { A.terminated[env] = (uint8_t)terminated; A.truncated[env] = (uint8_t)truncated; }
  }
  if (lane < nq) rec[lay.s_qpos + lane] = S.qpos[lane];
  if (lane < nv) { rec[lay.s_qvel + lane] = S.qvel[lane]; rec[lay.s_warm + lane] = S.qacc[lane]; }
  if (lane < nu) rec[lay.s_lastact + lane] = do_reset ? 0.f : raw_action;
  if (lane == 0) { meta[0] = sim_step; meta[1] = (int)(step_count + 1u); meta[2] = has_prev; meta[4] = nan_resets; }
}

}  // namespace cosim
