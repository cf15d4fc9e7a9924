// cosim_kernels.hip — the hot path as hand-written HIP for gfx950 (CDNA4), one environment per wavefront.
//
// One launch = one control step of every environment on this GPU, i.e. everything the reference does inside
//   CommandWrapper.step -> TimeLimitWrapper.step -> StateBuildWrapper.step -> FlamingoLightV1.step
// (reference envs/wrappers.py:258-269,309-320,391-405; envs/flamingo_light_v1/flamingo_light_v1.py:131-164):
//   receive_user_command -> delay_filter -> PD torque -> frame_skip x mj_step -> _get_obs -> _build_state ->
//   _apply_command_inplace, plus _get_info / _is_done / time limit and (engine extension) in-kernel auto-reset.
// mj_step itself is MuJoCo's (mujoco==3.2.7, not in the reference tree); the stages below follow its published
// pipeline (kinematics, comPos, crb, collision, makeConstraint, comVel, rne, fwdActuation, Newton solve with exact
// line search, implicitfast) re-designed for a 64-lane wave:
//   * lanes own bodies / dofs / geoms / constraint rows; tree recursions become level sweeps (FK) or sums over
//     precomputed ancestor / subtree bit masks (no dependent pointer chasing);
//   * intermediates sit in LDS (<= 10 KB per env -> 16 waves per CU, 4096 envs resident at once on 256 CUs);
//   * the nv x nv Newton Hessian is assembled entry-parallel, then factorised in registers (row i in lane i) with
//     v_readlane broadcasts; reductions use DPP row operations;
//   * per-env state is one contiguous HBM record, read once and written once per control step.
#include "cosim_dev.h"
#include "cosim_hullmap.h"

namespace cosim {

#define WSYNC()                                            \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)
// hide the lane id from loop-invariant code motion: address arithmetic stays local to the phase that uses it
#define LAUNDER(x) asm volatile("" : "+v"(x))

constexpr float MINVAL = 1e-15f;
constexpr float MINMU = 1e-5f;
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
  const float4* hull_cell;   // support maps (cosim_hullmap.h): HM_REC words per cell (header, inline candidates); overflow candidates
  const float4* hull_cand;
  const float* hfield;
  const float* hfield_mip;   // highest vertex of each tile of HF_TILE x HF_TILE vertices (row-major, ceil(nrow / HF_TILE) x ceil(ncol / HF_TILE))
  const unsigned* pairs;  // robot-robot candidate pairs: geom1 | geom2 << 16
  const float4* gext;     // per geom: MPR centre (body frame) xyz, raw sliding friction w
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
  int env_first;          // first env of this launch (cosim_step_range: a launch may cover a sub-range of the fleet)
  unsigned seed_lo, seed_hi;
  long long env_id0;
  float tol32;            // fp32 solver tolerance
  float ls_scale;         // multiplies the line search's gradient tolerance (tol32 * ls_tolerance * |s| / scale)
  int max_newton, max_ls;
  int nsub_override;      // > 0: physics substeps per control step (diagnostics; 0 = the model's frame_skip)
  int pair_coop;          // robot-robot pairs with a hull: 1 = one at a time, wave-cooperative vertex scans; 0 = lane-parallel
  int coop_walk;          // heightfield: 1 = hulls with few prisms under them are walked wave-cooperatively even when they have a support map (A/B)
  int block_cull;         // narrowphase kernel: 1 = blocks of 8 prisms are tested before their prisms; 0 = every block goes on to the per-prism pass
  int pair_boxbox;        // box-box pairs: 1 = mjc_BoxBox (up to eight contacts), 0 = through MPR like the other convex pairs (one contact)
  int prio[4];            // wave priority by solver lag: expected Newton iterations per substep, then the three lag thresholds
  // split pipeline (heightfield narrowphase in a kernel of its own, see env_narrow_kernel): one launch = one substep
  float* xcon;            // [N][XG][XC][8] contacts per (env, geom): dist, pos[3], normal[3], -- ; written by the narrowphase kernel
  int* xcnt;              // [N][XG] contacts found per (env, geom)
  float* xstate;          // [N][XS] what a control step holds across its substep launches: actuation force, raw action, torque
  int sub_index, sub_total;   // this launch is substep sub_index of sub_total (0: the fused kernel, every substep in one launch)
  int nw;                 // narrowphase: waves per env (wave w takes the geoms g with g % nw == w)
  int roll_steps;         // rollout launch (env_rollout_kernel): control steps per launch; actions / state_out / terminated / truncated / info are then [roll_steps][N][...]
  int* ovf;               // [N] per-env flag "this control step needs the large-capacity kernel" (null: no such kernel; contacts that find no slot are left out and counted)
  int env_count;          // envs of this launch (the fix-up kernel scans ovf[env_first .. env_first + env_count))
};

// ------------------------------------------------------------------------------------------------ wave helpers
__device__ __forceinline__ float rl(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ float rfl(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// sum over the 16 lanes of a DPP row, result in every lane of the row (quad_perm xor 1, xor 2, half mirror, mirror)
__device__ __forceinline__ float row_sum(float v) {
  v += dpp<0xB1>(v);
  v += dpp<0x4E>(v);
  v += dpp<0x141>(v);
  v += dpp<0x140>(v);
  return v;
}
// rows 1 and 3 += lane 15 of the row before (row_bcast15, row mask 0xA); rows 2 and 3 += lane 31 (row_bcast31, row mask 0xC):
// lane 63 then holds the wave total -- two DPP adds and one readlane instead of four readlanes and three adds
__device__ __forceinline__ float wave_sum(float v) {
  v = row_sum(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));
  return rl(v, 63);
}
// Three wave sums at once.  A DPP instruction may read a register only two wait states after the VALU instruction that wrote it, and
// left to itself the compiler reduces one value at a time through one register (add_dpp / s_nop 1 / add_dpp / ...: 17 issue slots per
// sum, all dependent).  Interleaved, each step of one sum sits two instructions behind its predecessor: no wait states, 7 slots per
// sum.  row_bcast adds write only the rows of their mask (the other rows keep their value, which is what the sum needs).
__device__ __forceinline__ void wave_sum3(float a, float b, float c, float& sa, float& sb, float& sc) {
#define DPP3(ctl) "v_add_f32_dpp %3, %3, %3 " ctl "\n\tv_add_f32_dpp %4, %4, %4 " ctl "\n\tv_add_f32_dpp %5, %5, %5 " ctl "\n\t"
  asm volatile(
      "s_nop 1\n\t"   // the inputs may have been written by the instruction just before
      DPP3("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
      DPP3("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1")
      DPP3("row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1")
      DPP3("row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1")
      DPP3("row_bcast:15 row_mask:0xa bank_mask:0xf")
      DPP3("row_bcast:31 row_mask:0xc bank_mask:0xf")
      "v_readlane_b32 %0, %3, 63\n\tv_readlane_b32 %1, %4, 63\n\tv_readlane_b32 %2, %5, 63\n\ts_nop 1"
      : "=&s"(sa), "=&s"(sb), "=&s"(sc), "+v"(a), "+v"(b), "+v"(c));
#undef DPP3
}
template <int LW>
__device__ __forceinline__ void grp_sum3(float a, float b, float c, float& sa, float& sb, float& sc);
__device__ __forceinline__ float wave_min(float v) {
  v = fminf(v, dpp<0xB1>(v));
  v = fminf(v, dpp<0x4E>(v));
  v = fminf(v, dpp<0x141>(v));
  v = fminf(v, dpp<0x140>(v));
  return fminf(fminf(rl(v, 0), rl(v, 16)), fminf(rl(v, 32), rl(v, 48)));
}
// k / d for 0 <= k < 2^22, 0 < d < 2^22: one reciprocal and a fix-up instead of the ~35-instruction integer division sequence
__device__ __forceinline__ int small_div(int k, int d) {
  int q = (int)((float)k * __builtin_amdgcn_rcpf((float)d));
  q -= (q * d > k) ? 1 : 0;
  q += ((q + 1) * d <= k) ? 1 : 0;
  return q;
}
__device__ __forceinline__ unsigned long long lanemask_lt(int lane) { return (1ull << lane) - 1ull; }
// The same operations over a group of LW lanes = one environment (LW = 64: the whole wave; LW = 32: two environments per
// wave, lanes 0-31 and 32-63).  hb = first lane of the caller's group.
template <int LW>
__device__ __forceinline__ float grp_sum(float v) {
  if constexpr (LW == 64) return wave_sum(v);
  else {
    // every lane of the group must end with the same bits (there is no final readlane here): keep the producer of v from
    // being contracted into the first add (fma(a, b, neighbour) != fma(a', b', own)); the adds themselves commute
    asm volatile("" : "+v"(v));
    v = row_sum(v);
    return v + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));   // partner row of the 32-lane group (xor 16)
  }
}
template <int LW>
__device__ __forceinline__ void grp_sum3(float a, float b, float c, float& sa, float& sb, float& sc) {
  if constexpr (LW == 64) wave_sum3(a, b, c, sa, sb, sc);
  else { sa = grp_sum<LW>(a); sb = grp_sum<LW>(b); sc = grp_sum<LW>(c); }
}
template <int LW>
__device__ __forceinline__ float grp_min(float v) {
  if constexpr (LW == 64) return wave_min(v);
  else {
    asm volatile("" : "+v"(v));
    v = fminf(v, dpp<0xB1>(v));
    v = fminf(v, dpp<0x4E>(v));
    v = fminf(v, dpp<0x141>(v));
    v = fminf(v, dpp<0x140>(v));
    return fminf(v, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F)));
  }
}
template <int LW>
__device__ __forceinline__ unsigned long long grp_ballot(bool p, int hb) {
  if constexpr (LW == 64) return __ballot(p);
  else return (__ballot(p) >> hb) & 0xffffffffull;
}
template <int LW>
__device__ __forceinline__ float grp_bcast(float v, int j, int hb) {   // value of lane j of the group, j the same in the whole group
  if constexpr (LW == 64) return rl(v, j);
  else return __shfl(v, j + hb, 64);
}

// ------------------------------------------------------------------------------------------------ small math
template <class PA, class PB>
__device__ __forceinline__ void qmul(float* r, PA a, PB b) {
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
template <class PQ>
__device__ __forceinline__ void q2m(float* m, PQ q) {
  float q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3], q11 = q[1] * q[1], q12 = q[1] * q[2],
        q13 = q[1] * q[3], q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2.f * (q12 - q03); m[2] = 2.f * (q13 + q02); m[3] = 2.f * (q12 + q03);
  m[5] = 2.f * (q23 - q01); m[6] = 2.f * (q13 - q02); m[7] = 2.f * (q23 + q01);
}
template <class PQ, class PV>
__device__ __forceinline__ void qrot(float* r, PQ q, PV v) {  // r = R(q) v (templates: model constants live in the constant address space)
  float m[9];
  q2m(m, q);
  float x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
        z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
template <class PA, class PB>
__device__ __forceinline__ void cross(float* r, PA a, PB b) {
  float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
template <class PA, class PB>
__device__ __forceinline__ float dot3(PA a, PB b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
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

// Philox4x32-10, counter-based: key = (seed_lo, seed_hi), counter = (step, purpose | index << 8, gid_lo, gid_hi); first output
// word.  Seed and env id in separate words: two seeds give independent fleets, not permutations of one another.
__device__ __forceinline__ unsigned philox_first(unsigned k0, unsigned k1, unsigned c0, unsigned c1, unsigned c2, unsigned c3) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c0;
}
__device__ __forceinline__ float u01(unsigned x) { return (float)(x >> 8) * (1.0f / 16777216.0f) + (0.5f / 16777216.0f); }  // (0,1)

// ------------------------------------------------------------------------------------------------ LDS per env
// MCT > 0 ("contact-twist" mode, CT): ground contacts do not become dense Jacobian rows.  A ground contact touches one body, so
// each of its four pyramid rows is w_e . twist_b(v) with a 6-vector w_e = [off x d_e; d_e] (d_e = n +- mu t, off = point - com) and
// the body's twist  twist_b(v) = sum_{j in chain(b)} cdof_j v_j:  J v is three dot products per contact, J^T f a per-body wrench
// pushed through the tree, and J^T D J a per-body 6 x 6 matrix pushed through the tree exactly like the composite inertia of
// mj_crb.  The solver's cost per Newton iteration then hardly depends on the number of contact rows, and the contact capacity
// (MCT slots, LW per pass) is a loop bound instead of a register count: humanoid_p_v0 on the 1 cm stairs terrains emits up to 50
// contacts per geom (mjc_ConvexHField), hundreds per env.  Robot-robot contacts (two bodies) keep dense rows (MCP slots).
// CT: robot-robot contacts in slots of their own; per ground contact the regulariser D, the four rows' J a - aref and J s;
// per body: twist of the vector in flight, contact wrench, 6 x 6 contact "inertia" (lower triangle, 21 entries).  Empty otherwise
// (the one-env-per-wave kernel of flamingo_light_v1 must stay within 160 KiB / 16 = 10240 B of LDS).
constexpr int XG = 24, XC = 50, XS = 96;
static_assert(CS_MAXDOF + 2 * (CS_MAXDOF - 6) + CS_MAXCMD <= 96 || true, "xstate: qact[NV], act[NV - 6], tq[NV - 6], cmd[CS_MAXCMD]");   // split pipeline: geoms per env, contacts per geom (mjMAXCONPAIR), floats of per-step state
// SLIM: the narrowphase-only kernel keeps none of the solver's arrays (one element each, never touched)
template <bool CT, bool HF, int NB, int MC, int MCP, bool SLIM = false>
struct CtLds {
  static constexpr int MCA = SLIM ? 1 : MC, MCPA = SLIM ? 1 : MCP, NBA = SLIM ? 1 : NB;
  float ppos[MCPA][3], pnrm[MCPA][3], pdist[MCPA];
  int pgeom[MCPA];
  // the four pyramid rows of a ground contact are base +- x1, base +- x2 (n +- mu t1, n +- mu t2): three numbers per contact
  float cD[MCA], cJar[MCA][3], cJv[MCA][3];
  float tw[NBA][6], bw[NBA][6];
  alignas(16) float bW[NBA][24];   // 21 used; rows padded so that the tree pass reads them as six 16-byte words
  unsigned cbmask;   // bodies that carry a ground contact
  // heightfield narrowphase, per geom: end of its (geom, prism) work items in the flattened list, contacts found so far, sub-grid
  // origin, prisms per strip row, lowest point of the geom
  int hf_end[HF ? 24 : 1], hf_cnt[HF ? 24 : 1], hf_cmin[HF ? 24 : 1], hf_rmin[HF ? 24 : 1], hf_ppr[HF ? 24 : 1];
  int hf_list[HF ? 128 : 1];   // work items the probe pass could not decide, in order, waiting for a full batch
  // work items that passed the height test, in order, waiting for a full probe batch: < 64 left over + one height pass of HFB x 64
  // (three sub-batches where hundreds of contact slots say "fine terrain under a big robot", two elsewhere: 256 more bytes would cost
  // the flamingo_light_v1 heightfield kernel its ninth wave per CU)
  static constexpr int HFB = MC >= 256 ? 3 : 2;
  int hf_zlist[HF ? 64 + 64 * HFB : 1];
  float hf_mg[HF ? 24 : 1];    // per geom: contact margin
  float hf_lo[HF ? 24 : 1];
  // narrowphase kernel: per geom an oriented box around it, grown by the reach of a prism's footprint from its centroid: centre[3],
  // rotation (row-major, local -> world)[9], half extents[3]
  float hf_box[(HF && SLIM) ? 24 : 1][16];
  float hf_org[(HF && SLIM) ? 24 : 1][2];   // local (x, y) of the sub-grid's first vertex (column cmin, row rmin)
  // narrowphase kernel, two-level walk: a block is up to 8 consecutive prisms of one strip row (4 cells); per geom the end of its
  // blocks in the flattened block list; blocks that passed the block tests, in order, waiting for the per-prism pass
  static constexpr int HBB = 2;   // sub-batches of 64 blocks per block pass
  int hf_bend[(HF && SLIM) ? 24 : 1];
  int hf_blist[(HF && SLIM) ? 8 * HFB + 64 * HBB : 1];
};
template <bool HF, int NB, int MC, int MCP, bool SLIM>
struct CtLds<false, HF, NB, MC, MCP, SLIM> {};
// NRM: the contact list stores normals (legacy: heightfield ground or robot-robot pairs; CT: heightfield ground -- the robot-robot
// contacts have slots of their own); MCPT: robot-robot contact slots of a CT kernel (0: the model has no pairs)
template <int NV, int NB, int RPL, bool NRM, int LW = 64, int MCT = 0, int MCPT = 12, bool HFL = NRM, bool SLIM = false>
struct EnvLds : CtLds<(MCT > 0), HFL, NB, (MCT > 0 ? MCT : 1), (MCPT > 0 ? MCPT : 1), SLIM> {
  static constexpr bool CT = MCT > 0;
  static constexpr int LD = NV | 1;   // odd leading dimension: conflict-free column access
  static constexpr int ROWS = LW * RPL;            // constraint rows per env: RPL rows per lane, LW lanes per env
  // rows kept for connect equalities: flamingo_light_v1 (nv 18) is the only cosim robot that has any (two connects, six rows);
  // cosim_create refuses a model with equalities on the other robots' kernels
  static constexpr int NEQR = NV == 18 ? 3 * MAXEQ / 2 : 0;
  // dense rows (one per lane slot) are the connect equalities and the contacts only: friction-loss and limit rows have unit Jacobians
  // (+-e_dof) and live in the lane of their dof ("dof rows", no slot).  64 slots = 6 equality rows + 14 contacts for flamingo_light_v1:
  // every ground contact a fallen robot makes on the plane in the bench (most seen: 14)
  static constexpr int MC = CT ? MCT : ((ROWS - NEQR) / 4 < 16 ? (ROWS - NEQR) / 4 : 16);  // contact slots (4 pyramid rows each); CT: ground contacts only
  static constexpr int MCP = CT ? MCPT : 1;        // CT: robot-robot contact slots (dense rows)
  static constexpr int CPL = (MC + LW - 1) / LW;   // CT: contacts per lane = passes over the contact list
  static constexpr int NGEN = NEQR + 4 * (CT ? MCP : MC);  // dense rows: 2 connect equalities (6 rows) + contacts
  static constexpr int NUMAX = NV - 6;   // actuators: at most one per hinge dof (the free joint's six dofs carry none)
  // (SLIM, the narrowphase-only kernel: kinematics and the prism walk only; the solver's arrays shrink to one element)
  static constexpr int NVA = SLIM ? 1 : NV, NBS = SLIM ? 1 : NB, MCA = SLIM ? 1 : MC, NGENA = SLIM ? 1 : NGEN, ROWSA = SLIM ? 1 : ROWS;
  float qpos[CS_MAXQ], qvel[NV], qacc[NV], qact[NV], qsm[NV], qcon[NV], sr[NV];
  float xpos[NB][3], xquat[NB][4];
  alignas(8) float cdof[NVA][6];
  float M[NVA][LD];
  union {                      // scratch that is dead before the solver starts shares the Hessian's space
    float H[NVA][LD];
    struct { float cdd[NVA][6], cfb[NBS][6]; } v;
    struct { float xanc[NB][3], xax[NB][3]; } k;   // joint anchors and axes: written by the kinematics sweep, dead once cdof is built
  } u;
  union {
    float cin[NBS][10];         // dead after the bias forces
    struct { float rowf[ROWSA], rowD[ROWSA]; } r;
  } w;
  static constexpr int NGS = (int)(sizeof(u) / 64) < 24 ? (int)(sizeof(u) / 64) : 24;   // geom lanes that can stage plane contacts (4 x {dist, pos}) in u's space
  float J[NGENA][LD];
  float cpos[MCA][3], cnrm[NRM ? MCA : 1][3], cdist[MCA];   // contact normals are +z on the plane: not stored
  int cgeom[MCA];   // geom2 | (geom1 + 1) << 8; geom1 + 1 == 0: the ground
  float p_mass[NB], p_binvw[NB], p_dinvw[NV], p_floss[NV], p_gmu[24];   // collision geoms: at most 22 (humanoid_p_v0)
  float act[NUMAX], tq[NUMAX], cmd[CS_MAXCMD + 2];
  float sens[10];   // framequat[4], gyro[3], velocimeter[3] of the last substep's forward pass
  float com[3];
  int ncon_ctr;
};

// contact i of the list a dense row / the debug dump / the termination test walks: in CT mode ground contacts come first, then the
// robot-robot contacts of the p* slots (index i - ncon); otherwise one list
template <bool CT, class LDS> __device__ __forceinline__ int con_geom(LDS& S, bool gnd, int c, int cp) { if constexpr (CT) return gnd ? S.cgeom[gnd ? c : 0] : S.pgeom[gnd ? 0 : cp]; else return S.cgeom[c]; }
template <bool CT, class LDS> __device__ __forceinline__ float con_dist(LDS& S, bool gnd, int c, int cp) { if constexpr (CT) return gnd ? S.cdist[gnd ? c : 0] : S.pdist[gnd ? 0 : cp]; else return S.cdist[c]; }
template <bool CT, class LDS> __device__ __forceinline__ const float* con_pos(LDS& S, bool gnd, int c, int cp) { if constexpr (CT) return gnd ? S.cpos[gnd ? c : 0] : S.ppos[gnd ? 0 : cp]; else return S.cpos[c]; }
template <bool CT, bool NRM, class LDS> __device__ __forceinline__ const float* con_nrm(LDS& S, bool gnd, int c, int cp) {   // NRM false: not stored (+z), do not read
  if constexpr (CT) return gnd ? S.cnrm[(NRM && gnd) ? c : 0] : S.pnrm[gnd ? 0 : cp]; else return S.cnrm[NRM ? c : 0];
}

// ------------------------------------------------------------------------------------------------ Cholesky
// Lane i < NV holds row i of a symmetric positive definite matrix in a[0..NV).  Right-looking factorisation with the
// pivot column broadcast by v_readlane; only the lower triangle is meaningful on exit (a[k], k <= i, is L[i][k]; slots
// k > i hold garbage, which is why no lane masking is needed: 2 VALU per (j,k) pair).  dinv is 1 / L[i][i].
// Trailing update of one pivot stage, a[K0 .. K0 + N) -= aj * (aj of lanes K0 .. K0 + N): the broadcasts go through scalar registers
// (v_readlane), and a VALU instruction may read a scalar register only two wait states after a VALU instruction wrote it.  Left to
// the compiler this became readlane / s_nop 1 / fma per element through ONE scalar register; grouped by four -- four readlanes into
// four scalars, then the four FMAs -- every FMA sits three instructions behind its readlane and no wait state is spent.
// (the leading s_nop 0 is the one wait state between the VALU instruction that produced aj and a v_readlane of it; the compiler's
// hazard recogniser does not look inside inline assembly)
template <int K0, int N, int NV>
__device__ __forceinline__ void chol_update(float (&a)[NV], float aj) {
  if constexpr (N >= 4) {
    float t0, t1, t2, t3;
    asm volatile(
        "s_nop 0\n\tv_readlane_b32 %4, %8, %9\n\tv_readlane_b32 %5, %8, %10\n\tv_readlane_b32 %6, %8, %11\n\tv_readlane_b32 %7, %8, %12\n\t"
        "v_fma_f32 %0, -%8, %4, %0\n\tv_fma_f32 %1, -%8, %5, %1\n\tv_fma_f32 %2, -%8, %6, %2\n\tv_fma_f32 %3, -%8, %7, %3"
        : "+v"(a[K0]), "+v"(a[K0 + 1]), "+v"(a[K0 + 2]), "+v"(a[K0 + 3]), "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3)
        : "v"(aj), "n"(K0), "n"(K0 + 1), "n"(K0 + 2), "n"(K0 + 3));
    chol_update<K0 + 4, N - 4, NV>(a, aj);
  } else if constexpr (N == 3) {
    float t0, t1, t2;
    asm volatile(
        "s_nop 0\n\tv_readlane_b32 %3, %6, %7\n\tv_readlane_b32 %4, %6, %8\n\tv_readlane_b32 %5, %6, %9\n\t"
        "v_fma_f32 %0, -%6, %3, %0\n\tv_fma_f32 %1, -%6, %4, %1\n\tv_fma_f32 %2, -%6, %5, %2"
        : "+v"(a[K0]), "+v"(a[K0 + 1]), "+v"(a[K0 + 2]), "=&s"(t0), "=&s"(t1), "=&s"(t2)
        : "v"(aj), "n"(K0), "n"(K0 + 1), "n"(K0 + 2));
  } else if constexpr (N == 2) {
    float t0, t1;
    asm volatile(
        "s_nop 0\n\tv_readlane_b32 %2, %4, %5\n\tv_readlane_b32 %3, %4, %6\n\ts_nop 0\n\t"
        "v_fma_f32 %0, -%4, %2, %0\n\tv_fma_f32 %1, -%4, %3, %1"
        : "+v"(a[K0]), "+v"(a[K0 + 1]), "=&s"(t0), "=&s"(t1)
        : "v"(aj), "n"(K0), "n"(K0 + 1));
  } else if constexpr (N == 1) {
    float t0;
    asm volatile("s_nop 0\n\tv_readlane_b32 %1, %2, %3\n\ts_nop 1\n\tv_fma_f32 %0, -%2, %1, %0" : "+v"(a[K0]), "=&s"(t0) : "v"(aj), "n"(K0));
  }
}
template <int J, int NV>
__device__ __forceinline__ void chol_stage64(float (&a)[NV], float& dinv, int lane) {
  if constexpr (J < NV) {
    const float ajj = rl(a[J], J);
    const float inv = rsqrtf(fmaxf(ajj, 1e-30f));
    dinv = (lane == J) ? inv : dinv;
    a[J] *= inv;  // column J of L in lanes >= J
    chol_update<J + 1, NV - J - 1, NV>(a, a[J]);
    chol_stage64<J + 1, NV>(a, dinv, lane);
  }
}
template <int NV, int LW>
__device__ __forceinline__ void chol_lower(float (&a)[NV], float& dinv, int lane, int hb) {
  if constexpr (LW == 64) {
    chol_stage64<0, NV>(a, dinv, lane);
  } else {
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const float ajj = grp_bcast<LW>(a[j], j, hb);
      const float inv = rsqrtf(fmaxf(ajj, 1e-30f));
      dinv = (lane == j) ? inv : dinv;
      a[j] *= inv;  // column j of L in lanes >= j
#pragma unroll
      for (int k = j + 1; k < NV; k++) a[k] -= a[j] * grp_bcast<LW>(a[j], k, hb);  // lane k's a[j] = L[k][j]
    }
  }
}
// solve (L L^T) x = b.  L = D L1 with D = diag(L) and L1 unit lower triangular; the LDS matrix holds the strictly lower part
// of L1 (row i scaled by 1 / L[i][i], zeros on and above the diagonal -- see chol_park), so
//   L y = b    <=>  L1 y = D^-1 b         and        L^T x = y   <=>  L1^T (D x) = y,
// and a substitution stage is one broadcast, one LDS read and one FMA with no lane masking (the zeros do it).
// Lane i holds b_i and returns x_i; forward reads row i, backward reads column i (consecutive lanes -> consecutive banks).
template <int NV, int LD, int LW>
__device__ __forceinline__ float chol_solve_lds(const float (*Lm)[LD], float dinv, float b, int lane, int hb) {
  const int li = lane < NV ? lane : 0;
  b *= dinv;
#pragma unroll
  for (int j = 0; j < NV - 1; j++) b -= Lm[li][j] * grp_bcast<LW>(b, j, hb);
#pragma unroll
  for (int j = NV - 1; j > 0; j--) b -= Lm[j][li] * grp_bcast<LW>(b, j, hb);
  return b * dinv;
}
// row `lane` of the factor into LDS in the form chol_solve_lds reads
template <int NV, int LD>
__device__ __forceinline__ void chol_park(float (*Lm)[LD], const float (&a)[NV], float dinv, int lane) {
  if (lane < NV) {
#pragma unroll
    for (int k = 0; k < NV; k++) Lm[lane][k] = k < lane ? a[k] * dinv : 0.f;
  }
}


// ------------------------------------------------------------------------------------------------ geometry helpers
// Contacts of one primitive geom (sphere / cylinder / box) against the plane through P0 with unit normal n:
// mjc_PlaneSphere / mjc_PlaneCylinder / mjc_PlaneBox (engine_collision_primitive.c).  The (at most four) contacts go, in
// MuJoCo's order, to a staging area as {dist, x, y, z}; the count is returned.  Staging in LDS (the caller passes a region that is
// dead during collision) instead of keeping the candidates' building blocks in registers across the slot assignment: a context
// struct with the union of the three types' vectors spilled to scratch on both sides of the ballots.
// GTM: geom types present in the model (bit 0 sphere, 1 cylinder, 2 box, 3 mesh): absent types cost no registers.
constexpr int GT_SPHERE = 1, GT_CYLINDER = 2, GT_BOX = 4, GT_MESH = 8;
template <int GTM, class REC>
__device__ __forceinline__ int prim_plane_contacts(const REC& R, const float* xq, const float* xp, const float* P0, const float* n,
                                                   float margin, float (*stage)[4]) {
  const int gt = R.g_type;
  float v[3], pos[3];
  qrot(v, xq, R.g_pos);
  for (int k = 0; k < 3; k++) pos[k] = xp[k] + v[k];
  const float dist0 = n[0] * (pos[0] - P0[0]) + n[1] * (pos[1] - P0[1]) + n[2] * (pos[2] - P0[2]);
  int cnt = 0;
  auto put = [&](float dist, float px, float py, float pz) {
    stage[cnt][0] = dist; stage[cnt][1] = px; stage[cnt][2] = py; stage[cnt][3] = pz;
    cnt++;
  };
  if ((GTM & GT_SPHERE) && gt == CS_GEOM_SPHERE) {
    const float r = R.g_size[0], dist = dist0 - r;
    if (dist0 <= margin + r) { const float a = r + 0.5f * dist; put(dist, pos[0] - n[0] * a, pos[1] - n[1] * a, pos[2] - n[2] * a); }
  } else if ((GTM & GT_CYLINDER) && gt == CS_GEOM_CYLINDER) {
    float gq[4], mat[9];
    qmul(gq, xq, R.g_quat);
    q2m(mat, gq);
    const float radius = R.g_size[0], half = R.g_size[1];
    float axis[3] = {mat[2], mat[5], mat[8]};
    float prjaxis = dot3(n, axis);
    if (prjaxis > 0.f) { axis[0] = -axis[0]; axis[1] = -axis[1]; axis[2] = -axis[2]; prjaxis = -prjaxis; }
    float vec[3] = {axis[0] * prjaxis - n[0], axis[1] * prjaxis - n[1], axis[2] * prjaxis - n[2]};
    const float len = sqrtf(dot3(vec, vec));
    if (len < 1e-12f) { vec[0] = mat[0] * radius; vec[1] = mat[3] * radius; vec[2] = mat[6] * radius; }
    else { const float s = radius / len; vec[0] *= s; vec[1] *= s; vec[2] *= s; }
    const float prjvec = dot3(vec, n);
    axis[0] *= half; axis[1] *= half; axis[2] *= half;
    prjaxis *= half;
    float vec1[3];
    cross(vec1, vec, axis);
    const float s1 = radius * 0.8660254037844386f * rsqrtf(fmaxf(dot3(vec1, vec1), 1e-30f));
    vec1[0] *= s1; vec1[1] *= s1; vec1[2] *= s1;
    const float prjvec1 = dot3(vec1, n);
    // candidate 0: deepest rim point of the near disk; 1: same rim point of the far disk; 2, 3: +-120 degrees on the near disk
    const float d0 = dist0 + prjaxis + prjvec, d1 = dist0 - prjaxis + prjvec, d2 = dist0 + prjaxis - prjvec * 0.5f + prjvec1,
                d3 = dist0 + prjaxis - prjvec * 0.5f - prjvec1;
    if (d0 <= margin) {   // nothing can touch unless the deepest point does
      put(d0, pos[0] + axis[0] + vec[0] - n[0] * d0 * 0.5f, pos[1] + axis[1] + vec[1] - n[1] * d0 * 0.5f, pos[2] + axis[2] + vec[2] - n[2] * d0 * 0.5f);
      if (d1 <= margin)
        put(d1, pos[0] - axis[0] + vec[0] - n[0] * d1 * 0.5f, pos[1] - axis[1] + vec[1] - n[1] * d1 * 0.5f, pos[2] - axis[2] + vec[2] - n[2] * d1 * 0.5f);
      if (d2 <= margin)
        put(d2, pos[0] + axis[0] - 0.5f * vec[0] + vec1[0] - n[0] * d2 * 0.5f, pos[1] + axis[1] - 0.5f * vec[1] + vec1[1] - n[1] * d2 * 0.5f,
            pos[2] + axis[2] - 0.5f * vec[2] + vec1[2] - n[2] * d2 * 0.5f);
      if (d3 <= margin)
        put(d3, pos[0] + axis[0] - 0.5f * vec[0] - vec1[0] - n[0] * d3 * 0.5f, pos[1] + axis[1] - 0.5f * vec[1] - vec1[1] - n[1] * d3 * 0.5f,
            pos[2] + axis[2] - 0.5f * vec[2] - vec1[2] - n[2] * d3 * 0.5f);
    }
  } else if ((GTM & GT_BOX) && gt == CS_GEOM_BOX) {
    float gq[4], mat[9], a[3], b[3], c[3];
    qmul(gq, xq, R.g_quat);
    q2m(mat, gq);
    // corner i = sum_k (+-size_k) * column k of mat
    for (int k = 0; k < 3; k++) { a[k] = mat[3 * k] * R.g_size[0]; b[k] = mat[3 * k + 1] * R.g_size[1]; c[k] = mat[3 * k + 2] * R.g_size[2]; }
    const float na = dot3(n, a), nb = dot3(n, b), nc = dot3(n, c);
#pragma unroll
    for (int i = 0; i < 8; i++) {   // at most the first four touching corners, in corner-index order (mjc_PlaneBox)
      const float s0 = (i & 1) ? 1.f : -1.f, s1 = (i & 2) ? 1.f : -1.f, s2 = (i & 4) ? 1.f : -1.f;
      const float ld = s0 * na + s1 * nb + s2 * nc;
      if (!(dist0 + ld > margin || ld > 0.f) && cnt < 4) {
        const float dist = dist0 + ld;
        put(dist, pos[0] + s0 * a[0] + s1 * b[0] + s2 * c[0] - n[0] * dist * 0.5f, pos[1] + s0 * a[1] + s1 * b[1] + s2 * c[1] - n[1] * dist * 0.5f,
            pos[2] + s0 * a[2] + s1 * b[2] + s2 * c[2] - n[2] * dist * 0.5f);
      }
    }
  }
  return cnt;
}

constexpr int HF_TILE = 8;
struct Terrain {  // heightfield geometry; (ox, oy) = world position of the local frame origin (the base's x, y)
  const float* data;
  const float* mip;   // per tile of HF_TILE x HF_TILE vertices: the highest one
  int nrow, ncol;
  float sx, sy, sz, gz;
  double ox, oy, dx, dy;   // ox, oy already relative to the hfield centre: local x -> field x = x + ox
};
// highest terrain vertex of the tiles under a bounding sphere's footprint (world z); -inf when the sphere is off the field.
// Conservative pre-test ahead of the prism walk: a geom whose lowest point is above it cannot touch any prism below it.  Reads the
// vertices themselves when the window is at most 4 x 4 (coarse terrain), else the tile maxima (at most 6 x 6 tiles = a footprint of
// 40 cells across, a row of tiles per trip): a 13 x 13 window of vertices read one dependent load at a time cost every wave of the
// narrowphase kernel 64 k cycles per substep.
__device__ __forceinline__ float terrain_max_under(const Terrain& T, const float* ctr, float rb) {
  const double lx = (double)ctr[0] + T.ox, ly = (double)ctr[1] + T.oy;
  if (fabs(lx) - rb > T.sx || fabs(ly) - rb > T.sy) return -3.0e38f;
  int cmin = (int)floor((lx - rb + T.sx) / T.dx), cmax = (int)ceil((lx + rb + T.sx) / T.dx);
  int rmin = (int)floor((ly - rb + T.sy) / T.dy), rmax = (int)ceil((ly + rb + T.sy) / T.dy);
  cmin = max(cmin, 0); rmin = max(rmin, 0); cmax = min(cmax, T.ncol - 1); rmax = min(rmax, T.nrow - 1);
  if (cmax - cmin < 4 && rmax - rmin < 4) {   // coarse terrain (cells of decimetres): the window's own vertices, all loads in flight
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = T.data[min(rmin + (k >> 2), rmax) * T.ncol + min(cmin + (k & 3), cmax)];
    float hmax = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) hmax = fmaxf(hmax, v[k]);
    return T.gz + T.sz * hmax;
  }
  const int mcol = (T.ncol + HF_TILE - 1) / HF_TILE;
  const int c0 = cmin / HF_TILE, c1 = cmax / HF_TILE, r0 = rmin / HF_TILE, r1 = rmax / HF_TILE;
  if (c1 - c0 > 5 || r1 - r0 > 5) return 3.0e38f;   // (the caller's size limit keeps footprints within 6 tiles; wider: no verdict)
  float hmax = 0.f;
  for (int r = r0; r <= r1; r++) {
    float v[6];
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = T.mip[r * mcol + min(c0 + k, c1)];
#pragma unroll
    for (int k = 0; k < 6; k++) hmax = fmaxf(hmax, v[k]);
  }
  return T.gz + T.sz * hmax;
}
// terrain elevation under (x, y) on the ray triangulation (mj_rayHfield: cell split along (r,c)-(r+1,c+1)); inside = on the field
__device__ __forceinline__ float terrain_height(const Terrain& T, float x, float y, bool& inside) {  // (x, y) local
  const double lx = (double)x + T.ox, ly = (double)y + T.oy;
  inside = !(lx < -T.sx || lx > T.sx || ly < -T.sy || ly > T.sy);
  if (!inside) return 0.f;
  const double fx = (lx + T.sx) / T.dx, fy = (ly + T.sy) / T.dy;
  int c = min(max((int)floor(fx), 0), T.ncol - 2), r = min(max((int)floor(fy), 0), T.nrow - 2);
  const float u = (float)(fx - (double)c), v = (float)(fy - (double)r);
  const float h00 = T.data[r * T.ncol + c], h01 = T.data[r * T.ncol + c + 1], h10 = T.data[(r + 1) * T.ncol + c], h11 = T.data[(r + 1) * T.ncol + c + 1];
  const float hh = u >= v ? h00 + u * (h01 - h00) + v * (h11 - h01) : h00 + v * (h10 - h00) + u * (h11 - h10);
  return T.gz + T.sz * hh;
#undef A
#undef ob
#undef lay
#undef KARGS_FENCE
}

}  // namespace cosim
#include "cosim_mpr.h"
#include "cosim_boxbox.h"
namespace cosim {
// ------------------------------------------------------------------------------------------------ impedance (mj_makeImpedance)
template <class PS>
__device__ __forceinline__ float impedance(PS solimp, float pos, float margin) {
  float d0 = fminf(MAXIMP, fmaxf(MINIMP, solimp[0])), d1 = fminf(MAXIMP, fmaxf(MINIMP, solimp[1]));
  float width = fmaxf(0.f, solimp[2]), mid = fminf(MAXIMP, fmaxf(MINIMP, solimp[3])), power = fmaxf(1.f, solimp[4]);
  if (d0 == d1 || width <= MINVAL) return 0.5f * (d0 + d1);
  float x = fabsf((pos - margin) / width);
  if (x >= 1.f) return d1;
  if (x <= 0.f) return d0;
  float y;
  if (power == 1.f) y = x;
  else if (power == 2.f) y = x <= mid ? x * x / mid : 1.f - (1.f - x) * (1.f - x) / (1.f - mid);
  else if (x <= mid) y = powf(x, power) / powf(mid, power - 1.f);
  else y = 1.f - powf(1.f - x, power) / powf(1.f - mid, power - 1.f);
  return d0 + y * (d1 - d0);
}

// LDS layout of a kernel variant (shared with the host side, which reports lds_bytes / contact_slots)
// KM (kernel mode): 0 the fused kernel (a whole control step, collision included); 1 narrowphase only (env_narrow_kernel: kinematics +
// the heightfield prism walk of some of the env's geoms, contacts to global memory); 2 one substep per launch with the ground
// contacts read from global memory (what the narrowphase kernel wrote)
template <int NV, int NB, int RPL, bool HF, bool SC, int EPW, int MCT, int KM = 0>
struct KTraits {
  static constexpr bool CT = MCT > 0;
  static constexpr bool NRM = CT ? HF : (HF || SC);
  static constexpr int MCPT = SC ? (NV >= 22 ? 12 : 8) : 0;
  using L = EnvLds<NV, NB, RPL, NRM, 64 / EPW, MCT, MCPT, HF, KM == 1>;
};

// ------------------------------------------------------------------------------------------------ the kernel
// HF: heightfield ground; SC: robot-robot (self) collision pairs; PROF: diagnostic build with s_memtime phase stamps;
// EPW: environments per wave (1: lane l of 64 plays object l; 2: two groups of 32 lanes, RPL rows per lane of the group)
// The argument block is read from the kernarg segment where it is used, not copied into ~70 scalar registers at entry (that
// copy spilled into vector-register lanes for the whole kernel and cost the solver's loops their scalar temporaries): `A` is the
// block in constant memory, and KARGS_FENCE() at the phase boundaries keeps the loads of a phase inside that phase.
typedef const KArgs __attribute__((address_space(4)))* KArgsP;
#define A (*kargs_p)
// (not in the fix-up kernel: it runs rarely, at one wave per SIMD, and the fence inside its loop over flagged envs does not compile)
#define KARGS_FENCE() do { if constexpr (!FIX) asm volatile("" : "+s"(kargs_p)); } while (0)
// One control step of environment `env` by the calling wave (EPW = 2: by the calling half-wave).  FIX: this is the large-capacity
// kernel redoing a step that the fleet's kernel gave up on (see env_fixup_kernel); otherwise, where the engine has such a kernel
// (A.ovf != null), a step whose contacts do not fit this kernel's slots is abandoned before anything is written and flagged.
// KM: see KTraits.  wsel: (KM == 1) which of the env's A.nw narrowphase waves this is.
template <int NV, int NB, int RPL, bool HF, int GTM, bool SC, bool PROF, int EPW, int MCT, bool FIX, int KM = 0>
__device__ __forceinline__ void env_body(KArgsP kargs_p, const int env, typename KTraits<NV, NB, RPL, HF, SC, EPW, MCT, KM>::L (&SS)[EPW], const int wsel = 0,
                                         const int kstep = 0) {   // kstep: control step of a rollout launch (row of the [K][N][...] I/O buffers)
  static_assert(EPW == 1 || (EPW == 2 && !HF && !SC && NV <= 32 && NB <= 32), "two environments per wave: flat ground, no pairs");
  static_assert(MCT == 0 || EPW == 1, "contact-twist mode: one environment per wave");
  static_assert(KM == 0 || (HF && MCT > 0 && EPW == 1 && !FIX && (!PROF || KM == 1)), "split pipeline: heightfield contact-twist kernels");
  using KT = KTraits<NV, NB, RPL, HF, SC, EPW, MCT, KM>;
  // split pipeline: this launch is substep sub_index of sub_total; the control-step prologue runs in the first, the epilogue in the last
  const bool sub_first = KM == 0 || A.sub_total == 0 || A.sub_index == 0;
  const bool sub_last = KM == 0 || A.sub_total == 0 || A.sub_index == A.sub_total - 1;
  constexpr bool CT = MCT > 0;
  constexpr bool NRM = KT::NRM;            // the (ground) contact list stores normals; otherwise they are +z
  constexpr bool NRMD = CT ? true : NRM;   // normals of the contacts behind dense rows (CT: the robot-robot slots always store them)
  constexpr int LW = 64 / EPW;
  using L = typename KT::L;
  constexpr int CPL = L::CPL;
  constexpr int MAXROWS = L::ROWS;
  constexpr int TRI = NV * (NV + 1) / 2;
  constexpr int MC = L::MC;
  constexpr int NGENMAX = L::NGEN;
  constexpr int EPL = (TRI + 63) / 64;
  // the fix-up kernel only ever steps (a constant there: its loop over flagged envs then has no exit the compiler must treat as divergent)
  const int kmode = FIX ? (int)MODE_STEP : A.mode;
  const int wlane = threadIdx.x;
  const int hb = EPW == 1 ? 0 : (wlane & 32);          // first lane of this lane's group
  const int lane = EPW == 1 ? wlane : (wlane & 31);     // role index inside the group
  L& S = SS[EPW == 1 ? 0 : (wlane >> 5)];
  typedef const DevModel __attribute__((address_space(4)))* DevModelP;   // the model never changes while a kernel runs: constant address space = invariant loads
  const auto& dm = *(DevModelP)(unsigned long long)A.dm;
  typedef const DevObs __attribute__((address_space(4)))* DevObsP;   // wrapper configuration: constant for the engine's lifetime
  const DevObsP ob_p = (DevObsP)(unsigned long long)A.ob;
#define ob (*ob_p)
#define lay (A.lay)
  float* rec = A.state + (size_t)env * lay.s_stride;
  const float* par = A.params + (size_t)env * lay.p_stride;
  const int nbody = dm.nbody, nu = dm.nu, nq = dm.nq, ngeom = dm.ngeom;
  const float h = dm.timestep;
  // diagnostic build only: shader-clock time per phase, summed over the substeps (never executed in the product kernel)
  unsigned long long pt0 = 0, pacc[16], pext[16];   // pext: heightfield narrowphase: cycles in [0] sub-grids [1] probe passes [2] full-MPR batches; counts [3] work items [4] probe batches [5] probes run [6] full batches [7] full MPRs run
  if (PROF) { for (int i = 0; i < 16; i++) pacc[i] = 0; for (int i = 0; i < 16; i++) pext[i] = 0; pt0 = __builtin_amdgcn_s_memtime(); }
  const unsigned long long pt_wave0 = PROF ? pt0 : 0ull;
#define PEXT_T0() unsigned long long pe0_ = 0; if (PROF) { __builtin_amdgcn_s_waitcnt(0); pe0_ = __builtin_amdgcn_s_memtime(); }
#define PEXT_ADD(i) do { if (PROF) { __builtin_amdgcn_s_waitcnt(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); pext[i] += t_ - pe0_; pe0_ = t_; } } while (0)
#define STAMP(i) do { KARGS_FENCE(); if (PROF) { __builtin_amdgcn_s_waitcnt(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[i] += t_ - pt0; pt0 = t_; } } while (0)

  // ---- per-env parameters -> LDS
  if (lane < nbody) { S.p_mass[lane] = par[lay.p_mass + lane]; S.p_binvw[lane] = par[lay.p_binvw + lane]; }
  if (lane < NV) { S.p_dinvw[lane] = par[lay.p_dinvw + lane]; S.p_floss[lane] = par[lay.p_floss + lane]; }
  if (lane < ngeom) S.p_gmu[lane] = par[lay.p_gmu + lane];
  const float meaninertia = par[lay.p_mean];
  int* meta = reinterpret_cast<int*>(rec + lay.s_meta);
  int sim_step = meta[0];
  const unsigned step_count = (unsigned)meta[1];
  int has_prev = meta[2];
  const unsigned long long gid = (unsigned long long)(A.env_id0 + env);
  const unsigned k0 = A.seed_lo, k1 = A.seed_hi, g0 = (unsigned)gid, g1 = (unsigned)(gid >> 32);

  bool do_reset = false;
  if (kmode == MODE_RESET) {
    do_reset = A.mask == nullptr || A.mask[env] != 0;
    if (!do_reset) return;
  }

  // ---- applied command (CommandWrapper.receive_user_command, wrappers.py:349-375) from the pre-step pose
  if (lane < CS_MAXCMD) {
    float applied = 0.f;
    if (lane < ob.command_dim && A.commands != nullptr) {
      const float* uc = A.commands + (size_t)env * ob.command_dim;
      if (!ob.position_command) applied = uc[lane] * ob.command_scales[lane];
      else {
        float px = rec[lay.s_qpos + 0], py = rec[lay.s_qpos + 1];
        float w = rec[lay.s_qpos + 3], x = rec[lay.s_qpos + 4], y = rec[lay.s_qpos + 5], z = rec[lay.s_qpos + 6];
        float dx = uc[0] - px, dy = uc[1] - py;
        float yaw = atan2f(2.f * (w * z + x * y), 1.f - 2.f * (y * y + z * z));
        float cy = cosf(-yaw), sy = sinf(-yaw);
        applied = lane == 0 ? cy * dx - sy * dy : sy * dx + cy * dy;
      }
    }
    S.cmd[lane] = applied;
  }

  // sensor values of the last forward pass (uniform across the wave)
  // (sensor values of the last forward pass, the raw action and the applied torque live in LDS: S.sens, S.act, S.tq)
  if (lane < L::NUMAX) { S.act[lane] = 0.f; S.tq[lane] = 0.f; }
  if (lane < 10) S.sens[lane] = lane == 0 ? 1.f : 0.f;
  int terminated = 0, truncated = 0, bad = 0;
  int st_newton = 0, st_ls = 0, st_build = 0, st_rows = 0;  // solver statistics of this control step
  int st_dropcon = 0, st_droplim = 0, st_maxcon = 0;          // capacity: contacts / limit rows left out, most contacts seen in one substep
  int st_walkcut = 0;                                         // heightfield: geoms whose prism walk was cut short (bounded walk)

  if (kmode != MODE_RESET) {
    // ---- state -> LDS
    if (lane < nq) S.qpos[lane] = rec[lay.s_qpos + lane];
    if (lane < NV) { S.qvel[lane] = rec[lay.s_qvel + lane]; S.qacc[lane] = rec[lay.s_warm + lane]; S.qact[lane] = 0.f; }
    WSYNC();

    // ---- control (once per control step): delay filter + PD, zero-order hold over the substeps
    if constexpr (KM == 2) {
      if (kmode == MODE_STEP && !sub_first) {   // a later substep launch of the control step: what the first one computed
        const float* xs = A.xstate + (size_t)env * XS;
        if (lane < NV) S.qact[lane] = xs[lane];
        if (lane < nu) { S.act[lane] = xs[NV + lane]; S.tq[lane] = xs[NV + L::NUMAX + lane]; }
        if (lane < CS_MAXCMD) S.cmd[lane] = xs[NV + 2 * L::NUMAX + lane];   // the applied command is a function of the PRE-step pose
        WSYNC();
      }
    }
    if (KM != 1 && kmode == MODE_STEP && sub_first) {
      const bool delayed = (ob.action_delay_prob > u01(philox_first(k0, k1, step_count, 0u, g0, g1))) && has_prev;  // control_manager.py:15-23
      if (lane < nu) {
        const auto& R = dm.rec[lane];
        const float raw_action = A.actions[((size_t)kstep * A.n_envs + env) * nu + lane];
        S.act[lane] = raw_action;
        float filt = delayed ? rec[lay.s_delay + lane] : raw_action;   // (the delay line is rewritten with the state, at the end)
        float a = filt * R.a_scale, g = R.a_cgear;
        float q = S.qpos[R.a_qadr] * g, qd = S.qvel[R.a_dadr] * g;
        float kp = par[lay.p_kp + lane], kd = par[lay.p_kd + lane];
        float t = R.a_velmode ? kd * (a - qd) : kp * (a - q) + kd * (0.f - qd);
        t *= R.a_gamma;
        t = fminf(R.a_maxtq, fmaxf(-R.a_maxtq, t));
        S.tq[lane] = t;
        // mj_fwdActuation: ctrl clamp, gear, then the joint-level actuatorfrcrange clamp (one motor per dof)
        float c = R.a_ctrllimited ? fminf(R.a_ctrlrange[1], fmaxf(R.a_ctrlrange[0], t)) : t;
        float f = R.a_gear * c;
        const int d = R.a_dof;
        const auto& RD = dm.rec[d];
        if (RD.d_frclimited) f = fminf(RD.d_frcrange[1], fmaxf(RD.d_frcrange[0], f));
        S.qact[d] = f;
      }
      has_prev = 1;
      sim_step += 1;
      WSYNC();
    }

    const int nsub = (kmode == MODE_DEBUG || KM != 0) ? 1 : (A.nsub_override > 0 ? A.nsub_override : dm.frame_skip);
#pragma nounroll
    for (int sub = 0; sub < nsub; sub++) {
      int ln = lane;
      LAUNDER(ln);
      STAMP(0);   // prologue / previous epilogue
      // =========================================================== mj_kinematics: level-synchronous over the tree
      {
        const auto& R = dm.rec[ln];
        const int b_level = ln < nbody ? R.b_level : -1;
        const int maxdepth = dm.maxdepth;
        // the joint's own rotation depends on this body's qpos only: one sincos per body ahead of the sweep, not one per level
        float ql[4] = {1.f, 0.f, 0.f, 0.f};
        if (b_level > 0 && R.b_jtype == CS_JNT_HINGE) {
          float sn, cs;
          sincosf(0.5f * (S.qpos[R.b_qadr] - R.j_q0), &sn, &cs);
          ql[0] = cs; ql[1] = R.j_axis[0] * sn; ql[2] = R.j_axis[1] * sn; ql[3] = R.j_axis[2] * sn;
        }
        for (int lev = 1; lev <= maxdepth; lev++) {
          if (b_level == lev) {
            const int b = ln, jt = R.b_jtype;
            float xp[3], xq[4], anc[3] = {0.f, 0.f, 0.f}, ax[3] = {0.f, 0.f, 1.f};
            if (jt == CS_JNT_FREE) {
              const int qa = R.b_qadr;
              // the step is invariant under horizontal translation except for terrain lookups: run it in a frame whose
              // origin is under the base (fp32 keeps mm-level contact depths exact even 100 m from the world origin)
              xp[0] = 0.f; xp[1] = 0.f; xp[2] = S.qpos[qa + 2];
              for (int k = 0; k < 4; k++) xq[k] = S.qpos[qa + 3 + k];
              qnorm(xq);
              for (int k = 0; k < 3; k++) { anc[k] = xp[k]; ax[k] = R.j_axis[k]; }
            } else {
              const int p = R.b_parent;
              float pq[4] = {S.xquat[p][0], S.xquat[p][1], S.xquat[p][2], S.xquat[p][3]};
              float v[3];
              qrot(v, pq, R.b_pos);
              for (int k = 0; k < 3; k++) xp[k] = S.xpos[p][k] + v[k];
              qmul(xq, pq, R.b_quat);
              if (jt == CS_JNT_HINGE) {
                qrot(ax, xq, R.j_axis);
                if (dm.any_jpos) {   // anchor away from the body origin: the origin swings around it
                  qrot(v, xq, R.j_pos);
                  for (int k = 0; k < 3; k++) anc[k] = xp[k] + v[k];
                  qmul(xq, xq, ql);
                  qrot(v, xq, R.j_pos);
                  for (int k = 0; k < 3; k++) xp[k] = anc[k] - v[k];
                } else {
                  for (int k = 0; k < 3; k++) anc[k] = xp[k];
                  qmul(xq, xq, ql);
                }
              }
              qnorm(xq);
            }
            for (int k = 0; k < 3; k++) { S.xpos[b][k] = xp[k]; S.u.k.xanc[b][k] = anc[k]; S.u.k.xax[b][k] = ax[k]; }
            for (int k = 0; k < 4; k++) S.xquat[b][k] = xq[k];
          }
          if (ln == 0 && lev == 1) {
            S.xpos[0][0] = S.xpos[0][1] = S.xpos[0][2] = 0.f;
            S.xquat[0][0] = 1.f; S.xquat[0][1] = S.xquat[0][2] = S.xquat[0][3] = 0.f;
          }
          WSYNC();
        }
      }

      STAMP(1);   // kinematics
      float com[3] = {0.f, 0.f, 0.f};   // the tree's centre of mass
      float qv = 0.f;                   // this lane's dof velocity at the start of the substep
      float cinert[10];
      if constexpr (KM != 1) {   // (the narrowphase-only kernel needs the body poses, nothing else)
      // =========================================================== mj_comPos: com, cinert (lane = body)
#pragma unroll
      for (int k = 0; k < 10; k++) cinert[k] = 0.f;
      {
        const auto& R = dm.rec[ln];
        float xip[3] = {0.f, 0.f, 0.f}, bmass = 0.f, ximat[9];
#pragma unroll
        for (int k = 0; k < 9; k++) ximat[k] = 0.f;
        if (ln > 0 && ln < nbody) {
          const int b = ln;
          float xq[4] = {S.xquat[b][0], S.xquat[b][1], S.xquat[b][2], S.xquat[b][3]};
          float v[3], qi[4];
          qrot(v, xq, R.b_ipos);
          for (int k = 0; k < 3; k++) xip[k] = S.xpos[b][k] + v[k];
          qmul(qi, xq, R.b_iquat);
          q2m(ximat, qi);
          bmass = S.p_mass[b];
        }
        float sx, sy, sz;
        grp_sum3<LW>(bmass * xip[0], bmass * xip[1], bmass * xip[2], sx, sy, sz);
        const float sm = grp_sum<LW>(bmass);
        const float inv = 1.f / fmaxf(sm, 1e-20f);
        const float c0 = sx * inv, c1 = sy * inv, c2 = sz * inv;
        if (ln == 0) { S.com[0] = c0; S.com[1] = c1; S.com[2] = c2; }
        if (ln > 0 && ln < nbody) {
          const auto* I = R.b_inertia;
          float dif[3] = {xip[0] - c0, xip[1] - c1, xip[2] - c2};
          float Rm[9];
#pragma unroll
          for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++)
              Rm[3 * r + c] = ximat[3 * r] * I[0] * ximat[3 * c] + ximat[3 * r + 1] * I[1] * ximat[3 * c + 1] + ximat[3 * r + 2] * I[2] * ximat[3 * c + 2];
          float dd = dot3(dif, dif);
          cinert[0] = Rm[0] + bmass * (dd - dif[0] * dif[0]);
          cinert[1] = Rm[4] + bmass * (dd - dif[1] * dif[1]);
          cinert[2] = Rm[8] + bmass * (dd - dif[2] * dif[2]);
          cinert[3] = Rm[1] - bmass * dif[0] * dif[1];
          cinert[4] = Rm[2] - bmass * dif[0] * dif[2];
          cinert[5] = Rm[5] - bmass * dif[1] * dif[2];
          cinert[6] = bmass * dif[0]; cinert[7] = bmass * dif[1]; cinert[8] = bmass * dif[2];
          cinert[9] = bmass;
        }
        if (ln < nbody) {
#pragma unroll
          for (int k = 0; k < 10; k++) S.w.cin[ln][k] = cinert[k];
        }
      }
      WSYNC();
      com[0] = S.com[0]; com[1] = S.com[1]; com[2] = S.com[2];
      // cdof (lane = dof)
      float cd[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      {
        const auto& R = dm.rec[ln];
        if (ln < NV) {
          qv = S.qvel[ln];
          const int b = R.d_body;
          const auto& RB = dm.rec[b];
          const int jt = RB.b_jtype, k = ln - RB.b_dadr;
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
            float axs[3] = {S.u.k.xax[b][0], S.u.k.xax[b][1], S.u.k.xax[b][2]};
            float off[3] = {com[0] - S.u.k.xanc[b][0], com[1] - S.u.k.xanc[b][1], com[2] - S.u.k.xanc[b][2]};
            cd[0] = axs[0]; cd[1] = axs[1]; cd[2] = axs[2];
            cross(cd + 3, axs, off);
          }
#pragma unroll
          for (int q = 0; q < 6; q++) S.cdof[ln][q] = cd[q];
#pragma unroll
          for (int k2 = 0; k2 < NV; k2++) S.M[ln][k2] = 0.f;
        }
      }
      WSYNC();

      STAMP(2);   // comPos + cdof
      // =========================================================== mj_crb: composite inertia of the dof's body, M row + column
      // the whole tree's composite inertia (what the free joint's six dofs need) is a sum over all body lanes: ten group
      // reductions instead of a 13-body loop that every lane of the wave would have to sit through
      const unsigned allbodies = ((nbody >= 32 ? 0u : (1u << nbody)) - 1u) & ~1u;
      float crb_all[10];
#pragma unroll
      for (int k = 0; k < 9; k += 3) {
        const bool inb = ln > 0 && ln < nbody;
        grp_sum3<LW>(inb ? cinert[k] : 0.f, inb ? cinert[k + 1] : 0.f, inb ? cinert[k + 2] : 0.f, crb_all[k], crb_all[k + 1], crb_all[k + 2]);
      }
      crb_all[9] = grp_sum<LW>((ln > 0 && ln < nbody) ? cinert[9] : 0.f);
      if (ln < NV) {
        const auto& R = dm.rec[ln];
        float crb[10];
        const unsigned sub = dm.rec[R.d_body].b_subtree;
#pragma unroll
        for (int k = 0; k < 10; k++) crb[k] = sub == allbodies ? crb_all[k] : 0.f;
        for (unsigned mk = sub == allbodies ? 0u : sub; mk; mk &= mk - 1) {
          const int c = __builtin_ctz(mk);
#pragma unroll
          for (int k = 0; k < 10; k++) crb[k] += S.w.cin[c][k];
        }
        float buf[6];
        mul_inert(buf, crb, cd);
        for (unsigned mk = R.d_ancmask; mk; mk &= mk - 1) {
          const int j = __builtin_ctz(mk);
          float v = 0.f;
#pragma unroll
          for (int q = 0; q < 6; q++) v += S.cdof[j][q] * buf[q];
          if (j == ln) v += R.d_armature;
          S.M[ln][j] = v;
          S.M[j][ln] = v;
        }
      }

      STAMP(3);   // crb
      // =========================================================== mj_comVel + mj_rne (bias) + passive + smooth force
      if (ln < NV) {
        const auto& R = dm.rec[ln];
        // velocity of the parent chain just before this dof (free joint: rotations see the translational part only)
        float cv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const auto& RB = dm.rec[R.d_body];
        const int jt = RB.b_jtype, k = ln - RB.b_dadr;
        float cdd[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (jt == CS_JNT_FREE) {
          if (k >= 3) {
            const int d0 = RB.b_dadr;
            cv[3] = S.qvel[d0]; cv[4] = S.qvel[d0 + 1]; cv[5] = S.qvel[d0 + 2];
          }
        } else {
          for (unsigned mk = R.d_ancmask & ~(1u << ln); mk; mk &= mk - 1) {
            const int j = __builtin_ctz(mk);
            const float qj = S.qvel[j];
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
        for (int q = 0; q < 6; q++) S.u.v.cdd[ln][q] = cdd[q];
      }
      WSYNC();
      float cfb_l[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // this body's inertial + Coriolis wrench (lane = body)
      if (ln > 0 && ln < nbody) {
        const auto& R = dm.rec[ln];
        float cvel_b[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float cacc[6] = {0.f, 0.f, 0.f, -dm.gravity[0], -dm.gravity[1], -dm.gravity[2]};
        for (unsigned mk = R.b_dofmask; mk; mk &= mk - 1) {
          const int j = __builtin_ctz(mk);
          const float qj = S.qvel[j];
#pragma unroll
          for (int q = 0; q < 6; q++) { cvel_b[q] += S.cdof[j][q] * qj; cacc[q] += S.u.v.cdd[j][q] * qj; }
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
        for (int q = 0; q < 6; q++) { cfb_l[q] = t1[q] + t3[q]; S.u.v.cfb[ln][q] = cfb_l[q]; }
      }
      WSYNC();
      float cfb_all[6];   // the whole tree's wrench, for the free joint's dofs: group reductions instead of a 13-body loop
#pragma unroll
      for (int q = 0; q < 6; q += 3) grp_sum3<LW>(cfb_l[q], cfb_l[q + 1], cfb_l[q + 2], cfb_all[q], cfb_all[q + 1], cfb_all[q + 2]);
      if (ln < NV) {
        const auto& R = dm.rec[ln];
        const unsigned sub = dm.rec[R.d_body].b_subtree;
        float f[6];
#pragma unroll
        for (int q = 0; q < 6; q++) f[q] = sub == allbodies ? cfb_all[q] : 0.f;
        for (unsigned mk = sub == allbodies ? 0u : sub; mk; mk &= mk - 1) {
          const int c = __builtin_ctz(mk);
#pragma unroll
          for (int q = 0; q < 6; q++) f[q] += S.u.v.cfb[c][q];
        }
        float bias = 0.f;
#pragma unroll
        for (int q = 0; q < 6; q++) bias += cd[q] * f[q];
        S.qsm[ln] = -R.d_damping * qv - bias + S.qact[ln];
        if (kmode == MODE_DEBUG && A.dbg != nullptr) A.dbg[1140 + ln] = bias;
      }

      // sensors of this forward pass (framequat, gyro, velocimeter on the IMU site); only the last substep's are read
      if (sub == nsub - 1 && sub_last) {
        const int ib = dm.imu_body;
        float xq[4] = {S.xquat[ib][0], S.xquat[ib][1], S.xquat[ib][2], S.xquat[ib][3]};
        float s_quat[4];
        qmul(s_quat, xq, dm.imu_quat);
        qnorm(s_quat);
        if (ln < 4) S.sens[ln] = ln == 0 ? s_quat[0] : (ln == 1 ? s_quat[1] : (ln == 2 ? s_quat[2] : s_quat[3]));
        float cv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (unsigned mk = dm.imu_dofmask; mk; mk &= mk - 1) {
          const int j = __builtin_ctz(mk);
          const float qj = S.qvel[j];
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
          if (ln == 0) { S.sens[4 + k] = g; S.sens[7 + k] = l; }
        }
      }

      }   // KM != 1
      STAMP(4);   // comVel + rne + sensors
      // =========================================================== collision: ground (plane or heightfield) vs robot geoms
      int ncon = 0;
      int npc = 0;   // CT: robot-robot contacts (slots of their own, dense rows); otherwise they follow the ground contacts in ncon
      const HullGraph HG{A.hull_vert, A.hull_adr, A.hull_nbr, A.hull_cell, A.hull_cand};
      const float4* const gext_ = A.gext;   // (locals, so that the lambdas below do not capture the argument-block pointer)
      const float* const hfdata_ = A.hfield;
      float* const xcon_ = KM == 1 ? A.xcon + (size_t)env * (XG * XC * 8) : nullptr;   // narrowphase kernel: this env's contact records
      // world pose of geom g as a convex object (mesh: body frame, vertices in body coordinates; primitive: geom frame)
      auto make_cobj = [&](CObj& o, int g) __attribute__((always_inline)) {
        const auto& G = dm.rec[g];
        const int gb = G.g_body;
        const float bq[4] = {S.xquat[gb][0], S.xquat[gb][1], S.xquat[gb][2], S.xquat[gb][3]};
        const float4 ge = gext_[g];
        const float cl[3] = {ge.x, ge.y, ge.z};
        float v[3];
        qrot(v, bq, cl);
        for (int k = 0; k < 3; k++) o.center[k] = S.xpos[gb][k] + v[k];
        o.kind = G.g_type; o.adr = G.g_hulladr; o.num = G.g_hullnum; o.map = dm.g_hullmap[g];
        for (int k = 0; k < 3; k++) o.size[k] = G.g_size[k];
        if (G.g_type == CS_GEOM_MESH) {
          for (int k = 0; k < 3; k++) o.pos[k] = S.xpos[gb][k];
          for (int k = 0; k < 4; k++) o.q[k] = bq[k];
        } else {
          qrot(v, bq, G.g_pos);
          for (int k = 0; k < 3; k++) o.pos[k] = S.xpos[gb][k] + v[k];
          qmul(o.q, bq, G.g_quat);
        }
      };
      {
        const auto& R = dm.rec[ln];
        constexpr bool is_plane = !HF;
        Terrain T;
        T.data = hfdata_; T.mip = A.hfield_mip; T.nrow = dm.hfield_nrow; T.ncol = dm.hfield_ncol;
        T.sx = dm.hfield_size[0]; T.sy = dm.hfield_size[1]; T.sz = dm.hfield_size[2]; T.gz = dm.ground_pos[2];
        T.ox = (double)S.qpos[0] - (double)dm.ground_pos[0]; T.oy = (double)S.qpos[1] - (double)dm.ground_pos[1];
        T.dx = is_plane ? 1.0 : 2.0 * (double)T.sx / (double)(T.ncol - 1); T.dy = is_plane ? 1.0 : 2.0 * (double)T.sy / (double)(T.nrow - 1);
        bool mesh_near = false;
        const float gpos[3] = {0.f, 0.f, T.gz};
        // (narrowphase kernel: every wave of the env sizes every geom's walk -- lane-parallel, cheap -- and then takes its share of the
        // geoms by work, see `mine` below)
        const bool active = ln < ngeom && R.g_ground;
        const int b = active ? R.g_body : 0, gt = active ? R.g_type : -1;
        float xq[4] = {S.xquat[b][0], S.xquat[b][1], S.xquat[b][2], S.xquat[b][3]};
        float xp[3] = {S.xpos[b][0], S.xpos[b][1], S.xpos[b][2]};
        float ctr[3];
        {
          float v[3];
          qrot(v, xq, R.g_rcenter);
          for (int k = 0; k < 3; k++) ctr[k] = xp[k] + v[k];
        }
        const float margin = R.g_margin, rb = R.g_rbound;
        if (is_plane) {
          const float nz[3] = {0.f, 0.f, 1.f};
          // staging: the Hessian's LDS space, dead until the solver (16 floats per geom lane)
          float (*stage)[4] = reinterpret_cast<float (*)[4]>(&S.u.H[0][0]) + 4 * (ln < L::NGS ? ln : 0);   // (cosim_create checks ngeom <= NGS)
          int cnt = 0;
          if (active && gt == CS_GEOM_MESH) {
            // exact reject: lowest corner of the hull's body-frame box (oriented into the world) above the plane
            float m[9];
            q2m(m, xq);
            const float low = ctr[2] - T.gz - (fabsf(m[6]) * R.g_half[0] + fabsf(m[7]) * R.g_half[1] + fabsf(m[8]) * R.g_half[2]);
            mesh_near = low <= margin;
          } else if (active) {
            if constexpr ((GTM & ~GT_MESH) != 0) {
              if (ln < L::NGS) cnt = prim_plane_contacts<GTM>(R, xq, xp, gpos, nz, margin, stage);
            }
          }
          // compaction in (geom, contact) order: slot = contacts of lower lanes + rank inside this geom
          int off = 0, total = 0;
#pragma unroll
          for (int s2 = 0; s2 < 4; s2++) {
            unsigned long long mk = grp_ballot<LW>(cnt > s2, hb);
            off += __popcll(mk & lanemask_lt(ln));
            total += __popcll(mk);
          }
          if constexpr ((GTM & ~GT_MESH) != 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const int slot = off + j;
              if (j < cnt && slot < MC) {
                S.cdist[slot] = stage[j][0];
                S.cgeom[slot] = ln;
                S.cpos[slot][0] = stage[j][1]; S.cpos[slot][1] = stage[j][2]; S.cpos[slot][2] = stage[j][3];
                if (NRM) { S.cnrm[NRM ? slot : 0][0] = 0.f; S.cnrm[NRM ? slot : 0][1] = 0.f; S.cnrm[NRM ? slot : 0][2] = 1.f; }
              }
            }
          }
          ncon = total;
        } else {
          // heightfield (mjc_ConvexHField).  Every (geom, prism) pair is one work item; the items of all geoms form one list in
          // (geom, strip) order -- MuJoCo's order -- and are dealt to the lanes 64 at a time: each lane runs its own MPR (primitive
          // supports are O(1), hull supports climb the neighbour graph).  A geom keeps its first mjMAXCONPAIR = 50 penetrating
          // prisms.  The walk is a loop over however many prisms lie under the geoms: 1 cm stairs cells cost passes, not slots.
          static_assert(!HF || CT, "heightfield kernels run in contact-twist mode");
          if constexpr (CT && KM == 2) {
            // split pipeline: the narrowphase kernel has left this substep's contacts in global memory, per geom in strip order
            // (MuJoCo's order within a geom); geoms in index order make the list
            const int* xc = A.xcnt + (size_t)env * XG;
            int end = ln < ngeom ? min(max(xc[ln], 0), XC) : 0;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) { const int t = __shfl_up(end, o, 64); if (ln >= o) end += t; }
            if (ln < 24) S.hf_end[ln] = end;
            const int total = __builtin_amdgcn_readlane(end, 23);
            WSYNC();
            const float* xb = A.xcon + (size_t)env * (XG * XC * 8);
            for (int idx = ln; idx < total && idx < MC; idx += 64) {
              int g = 0;
              while (idx >= S.hf_end[g]) g++;
              const float4* r = reinterpret_cast<const float4*>(xb + ((size_t)g * XC + (idx - (g > 0 ? S.hf_end[g > 0 ? g - 1 : 0] : 0))) * 8);
              const float4 a = r[0], bb = r[1];
              S.cdist[idx] = a.x; S.cgeom[idx] = g;
              S.cpos[idx][0] = a.y; S.cpos[idx][1] = a.z; S.cpos[idx][2] = a.w;
              S.cnrm[NRM ? idx : 0][0] = bb.x; S.cnrm[NRM ? idx : 0][1] = bb.y; S.cnrm[NRM ? idx : 0][2] = bb.z;
            }
            ncon = total;
          } else if constexpr (CT) {
            if (ln == 0) S.ncon_ctr = 0;
            PEXT_T0();
            int n_items = 0;
            bool coop_geom = false, walk_cut = false;
            if (active) {
              // phase A (lane = geom): MuJoCo's early outs, then the sub-grid under the geom's axis-aligned box
              CObj og;
              make_cobj(og, ln);
              const float base_ = dm.hfield_size[3];
              const double lx = (double)ctr[0] + T.ox, ly = (double)ctr[1] + T.oy;
              bool out = !(fabsf(ctr[0]) + fabsf(ctr[1]) + fabsf(ctr[2]) < 1e8f) ||   // non-finite pose (the env is reset at the end of the step)
                         (double)T.sx < lx - rb - margin || -(double)T.sx > lx + rb + margin || (double)T.sy < ly - rb - margin ||
                         -(double)T.sy > ly + rb + margin || T.sz < ctr[2] - T.gz - rb - margin || -base_ > ctr[2] - T.gz + rb + margin;
              float lo[3], hi[3];
              {
                // world box around the geom's body-frame box: exact enough to reject, and for hulls also the box the sub-grid is cut
                // from (six support queries would be six scans of the hull by this one lane; a box that contains the hull only adds
                // prisms that cannot be hit, the contacts are the same)
                float m[9];
                q2m(m, xq);
#pragma unroll
                for (int k = 0; k < 3; k++) {
                  const float e = fabsf(m[3 * k]) * R.g_half[0] + fabsf(m[3 * k + 1]) * R.g_half[1] + fabsf(m[3 * k + 2]) * R.g_half[2];
                  lo[k] = ctr[k] - e; hi[k] = ctr[k] + e;
                }
              }
              // coarse terrain: the highest vertex of the 8 x 8-vertex tiles under the bounding sphere's footprint (6 x 6 tiles at
              // most) against the box's lowest point
              if (!out && 2.0 * rb <= 40.0 * T.dx && 2.0 * rb <= 40.0 * T.dy) out = lo[2] - margin > terrain_max_under(T, ctr, rb);
              if (!out) {
                if (gt != CS_GEOM_MESH) {
#pragma unroll
                  for (int k = 0; k < 3; k++) {
                    float d[3] = {0.f, 0.f, 0.f}, p_[3];
                    d[k] = 1.f;
                    cobj_support<GTM, false>(og, HG, d, p_, ln);
                    hi[k] = p_[k];
                    d[k] = -1.f;
                    cobj_support<GTM, false>(og, HG, d, p_, ln);
                    lo[k] = p_[k];
                  }
                }
                const double x0 = (double)lo[0] + T.ox, x1 = (double)hi[0] + T.ox, y0 = (double)lo[1] + T.oy, y1 = (double)hi[1] + T.oy;
                out = x0 - margin > T.sx || x1 + margin < -T.sx || y0 - margin > T.sy || y1 + margin < -T.sy || lo[2] - T.gz - margin > T.sz ||
                      hi[2] - T.gz + margin < -base_;
                if (!out) {
                  int cmin = (int)floor((x0 + T.sx) / T.dx), cmax = (int)ceil((x1 + T.sx) / T.dx);
                  int rmin = (int)floor((y0 + T.sy) / T.dy), rmax = (int)ceil((y1 + T.sy) / T.dy);
                  cmin = max(cmin, 0); rmin = max(rmin, 0); cmax = min(cmax, T.ncol - 1); rmax = min(rmax, T.nrow - 1);
                  const int ppr = 2 * (cmax - cmin + 1) - 2;   // prisms per strip row: consecutive vertex triples of 2 ncols strip vertices
                  if (ppr > 0 && (rmax - rmin) * ppr > 32768) { rmax = rmin + 32768 / ppr; walk_cut = true; }   // bounded walk (a 0.6 m x 0.6 m footprint of 1 cm cells fits); counted below
                  if (rmax > rmin && ppr > 0) {
                    n_items = (rmax - rmin) * ppr;
                    S.hf_cmin[ln] = cmin; S.hf_rmin[ln] = rmin; S.hf_ppr[ln] = ppr; S.hf_lo[ln] = lo[2]; S.hf_mg[ln] = margin;
                    if constexpr (KM == 1) {
                      S.hf_org[ln][0] = (float)((double)cmin * T.dx - (double)T.sx - T.ox);
                      S.hf_org[ln][1] = (float)((double)rmin * T.dy - (double)T.sy - T.oy);
                      // oriented box for the per-prism cull of the height pass: primitives in their own frame (box: its size; cylinder:
                      // (r, r, h); sphere: r), hulls in the body frame (the box of the hull's vertices), each grown by the farthest a
                      // point of a prism's footprint lies from the footprint's centroid
                      const float grow = 0.7f * sqrtf((float)(T.dx * T.dx + T.dy * T.dy));
                      float bm[9], bc[3], bh[3];
                      if (gt == CS_GEOM_MESH) {
                        q2m(bm, xq);
                        for (int k = 0; k < 3; k++) { bc[k] = ctr[k]; bh[k] = R.g_half[k]; }
                      } else {
                        q2m(bm, og.q);
                        for (int k = 0; k < 3; k++) bc[k] = og.pos[k];
                        bh[0] = og.size[0]; bh[1] = gt == CS_GEOM_BOX ? og.size[1] : og.size[0];
                        bh[2] = gt == CS_GEOM_BOX ? og.size[2] : (gt == CS_GEOM_CYLINDER ? og.size[1] : og.size[0]);
                      }
                      for (int k = 0; k < 3; k++) { S.hf_box[ln][k] = bc[k]; S.hf_box[ln][12 + k] = bh[k] + grow; }
                      for (int k = 0; k < 9; k++) S.hf_box[ln][3 + k] = bm[k];
                    }
                    // a hull with only a few prisms under it (coarse terrain) and no support map: its prisms one at a time with all 64
                    // lanes sharing the vertex scans -- 64 lanes each scanning a 700-vertex hull for a handful of items would cost
                    // more.  With a support map a lane's query walks a handful of candidates: the prisms of all such hulls go through
                    // the staged walk together, one MPR per lane, instead of one after the other.
                    if (gt == CS_GEOM_MESH && n_items < 32 && R.g_hullnum > 64 && (dm.g_hullmap[ln] < 0 || A.coop_walk != 0)) { coop_geom = true; n_items = 0; }
                  }
                }
              }
            }
            bool mine = active;
            if constexpr (KM == 1) {
              // Which of the env's A.nw waves walks which geom: geoms ranked by the size of their walk (work items; a cooperative hull
              // walk counts as one batch), dealt to the waves in serpentine order -- the biggest walks land on different waves and the
              // waves' totals stay close.  Every wave of the env computes the same ranking from the same poses.
              const int work = coop_geom ? 64 : n_items;
              int rank = 0;
              for (int g = 0; g < 24; g++) {
                const int wg = __builtin_amdgcn_readlane(work, g);
                rank += (wg > work || (wg == work && g < ln)) ? 1 : 0;
              }
              const int nw_ = A.nw, round_ = rank / nw_, pos_ = rank - round_ * nw_;
              mine = active && ((round_ & 1) ? nw_ - 1 - pos_ : pos_) == wsel;
              if (!mine) { n_items = 0; coop_geom = false; walk_cut = false; }
            }
            st_walkcut += __popcll(__ballot(walk_cut));   // wave-uniform (lane 0 writes the counters back): one per geom whose walk was cut short
            int end = n_items;   // inclusive prefix sum over the geom lanes
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) { const int t = __shfl_up(end, o, 64); if (ln >= o) end += t; }
            if (ln < 24) { S.hf_end[ln] = end; S.hf_cnt[ln] = 0; }
            const int total = __builtin_amdgcn_readlane(end, 23);   // lanes past ngeom add nothing
            int btotal = 0;
            if constexpr (KM == 1) {
              // blocks of up to 8 consecutive prisms of a strip row: rows x ceil(ppr / 8) per geom, same (geom, strip) order
              int bend = 0;
              if (n_items > 0) { const int ppr_ = S.hf_ppr[ln]; bend = small_div(n_items, ppr_) * ((ppr_ + 7) >> 3); }
#pragma unroll
              for (int o = 1; o < 32; o <<= 1) { const int t = __shfl_up(bend, o, 64); if (ln >= o) bend += t; }
              if (ln < 24) S.hf_bend[ln] = bend;
              btotal = __builtin_amdgcn_readlane(bend, 23);
            }
            WSYNC();
            PEXT_ADD(0);

            // work item -> its geom, the geom's contact margin and the prism (strip vertices kk, kk + 1, kk + 2 of strip row r: vertex v
            // sits in column cmin + (v >> 1), row r + 1 for even v and r for odd v)
            auto item_prism = [&](int item, int& g, float& gmargin, PrismObj& P) __attribute__((always_inline)) {
              while (item >= S.hf_end[g]) g++;
              const int k = item - (g > 0 ? S.hf_end[g > 0 ? g - 1 : 0] : 0), ppr = S.hf_ppr[g], rrow = small_div(k, ppr), kk = k - rrow * ppr;
              const int r = S.hf_rmin[g] + rrow, cmin = S.hf_cmin[g];
              gmargin = S.hf_mg[g];
              P.zb = T.gz - dm.hfield_size[3];
#pragma unroll
              for (int i = 0; i < 3; i++) {
                const int v = kk + i, c = cmin + (v >> 1), rr = r + 1 - (v & 1);
                P.x[i] = (float)(c * T.dx - (double)T.sx - T.ox);
                P.y[i] = (float)(rr * T.dy - (double)T.sy - T.oy);
                P.zt[i] = T.data[rr * T.ncol + c] * T.sz + T.gz + gmargin;
              }
            };
            auto prism_centre = [](const PrismObj& P, float* c1) __attribute__((always_inline)) {
              c1[0] = (P.x[0] + P.x[1] + P.x[2]) * (1.f / 3.f); c1[1] = (P.y[0] + P.y[1] + P.y[2]) * (1.f / 3.f);
              c1[2] = (P.zt[0] + P.zt[1] + P.zt[2] + 3.f * P.zb) * (1.f / 6.f);
            };
            // full MPR on up to 64 listed items (one per lane, list order = (geom, strip) order) and ordered append of the hits
            auto run_listed = [&](int cnt) __attribute__((always_inline)) {
              bool hit = false;
              int mpr_iters = 0;
              int g = 0;
              float depth = 0.f, nrm_[3] = {0.f, 0.f, 1.f}, pos_[3] = {0.f, 0.f, 0.f}, gmargin = 0.f;
              unsigned long long tq0 = 0, tq1 = 0;
              if (PROF) { __builtin_amdgcn_s_waitcnt(0); tq0 = __builtin_amdgcn_s_memtime(); }
              if (ln < cnt) {
                PrismObj P;
                item_prism(S.hf_list[ln], g, gmargin, P);
                if (S.hf_cnt[g] < 50) {
                  CObj o;
                  make_cobj(o, g);
                  float c1[3];
                  prism_centre(P, c1);
                  if (PROF) { __builtin_amdgcn_s_waitcnt(0); tq1 = __builtin_amdgcn_s_memtime(); }
                  const MprPrismGeom<GTM, false> sup{P, o, HG, ln};
                  int nit = 0;
                  hit = mpr_penetration(sup, c1, o.center, depth, nrm_, pos_, PROF ? &nit : nullptr) && (nrm_[0] != 0.f || nrm_[1] != 0.f || nrm_[2] != 0.f);
                  if (PROF) mpr_iters = nit;
                }
              }
              if (PROF) {   // [3] is reused in this build: sum over batches of the slowest lane's MPR loop iterations
                __builtin_amdgcn_s_waitcnt(0);
                const unsigned long long tq2 = __builtin_amdgcn_s_memtime();
                const unsigned long long am_ = __ballot(tq1 != 0);   // tq1 was taken inside the divergent branch: read it from a lane that ran
                const unsigned t1 = am_ ? (unsigned)__builtin_amdgcn_readlane((int)(unsigned)tq1, __builtin_ctzll(am_)) : (unsigned)tq0;
                pext[5] += (unsigned)((unsigned)tq2 - t1);            // diagnostic: cycles inside mpr_penetration ...
                pext[7] += (unsigned)(t1 - (unsigned)tq0);            // ... and in the item / geom set-up before it
                int mx = mpr_iters;
                for (int o_ = 32; o_ > 0; o_ >>= 1) mx = max(mx, __shfl_xor(mx, o_, 64));
                pext[3] += mx;
              }
              int rank_g = 0;   // rank of a hit among the hits of its own geom
              for (unsigned long long rest = __ballot(hit); rest;) {
                const int gs = __shfl(g, __builtin_ctzll(rest), 64);
                const unsigned long long same = __ballot(hit && g == gs);
                if (hit && g == gs) rank_g = __popcll(same & lanemask_lt(ln));
                rest &= ~same;
              }
              const bool keep = hit && S.hf_cnt[g] + rank_g < 50;
              const unsigned long long km = __ballot(keep);
              const int slot = S.ncon_ctr + __popcll(km & lanemask_lt(ln));
              if constexpr (KM == 1) {   // to the env's record in global memory: slot (geom, rank among the geom's hits)
                if (keep) {
                  float4* o = reinterpret_cast<float4*>(xcon_ + ((size_t)g * XC + S.hf_cnt[g] + rank_g) * 8);
                  o[0] = make_float4(gmargin - depth, pos_[0], pos_[1], pos_[2]);
                  o[1] = make_float4(nrm_[0], nrm_[1], nrm_[2], 0.f);
                }
              } else if (keep && slot < MC) {
                S.cdist[slot] = gmargin - depth;
                S.cgeom[slot] = g;
                for (int k = 0; k < 3; k++) { S.cpos[slot][k] = pos_[k]; S.cnrm[NRM ? slot : 0][k] = nrm_[k]; }
              }
              WSYNC();   // every lane has read the counters
              if (keep) atomicAdd(&S.hf_cnt[g], 1);
              if (ln == 0) S.ncon_ctr += __popcll(km);
              WSYNC();
            };
            // Three stages, each on compacted batches of up to 64 items, one call site per stage (so that each inlines once):
            //   height pass (all items): a prism whose top lies entirely below the geom's lowest point cannot touch it
            //     (mjc_ConvexHField's own test) -- three loads and three compares per item; survivors -> zlist;
            //   probe pass (zlist): MPR's first two exits decide most survivors; the undecided ones -> list;
            //   full pass (list): MPR, ordered append of the hits.
            // The later stage runs as soon as its list holds a full batch; the tails are drained at the end.
            // The narrowphase kernel (fine terrain under a big robot: ~15 000 prisms under an env's geoms per substep, one in twenty of
            // them near a geom) walks in two levels: a block pass over blocks of 8 prisms -- one height test and one oriented-box test
            // for the block, from its 10 vertices -- and the per-prism height pass only for the prisms of blocks that passed.
            int nlist = 0, nz = 0, base = 0, nb = 0;
            bool walked = KM == 1 ? btotal == 0 : total == 0;
            while (!(walked && nb == 0 && nz == 0 && nlist == 0)) {
              if (nlist >= 64 || (walked && nb == 0 && nz == 0)) {
                const int cnt = min(nlist, 64);
                if (PROF) { pext[6] += 1; }
                run_listed(cnt);
                const int moved = (ln + 64 < nlist) ? S.hf_list[ln + 64] : 0;
                WSYNC();
                if (ln + 64 < nlist) S.hf_list[ln] = moved;
                nlist -= cnt;
                WSYNC();
                PEXT_ADD(2);
              } else if (nz >= 64 || (walked && nb == 0)) {
                const int cnt = min(nz, 64);
                bool maybe = false;
                int item = 0;
                if (ln < cnt) {
                  item = S.hf_zlist[ln];
                  int g = 0;
                  float gmargin;
                  PrismObj P;
                  item_prism(item, g, gmargin, P);
                  if (S.hf_cnt[g] < 50) {
                    CObj o;
                    make_cobj(o, g);
                    float c1[3];
                    prism_centre(P, c1);
                    const MprPrismGeom<GTM, false> sup{P, o, HG, ln};
                    maybe = mpr_probe(sup, c1, o.center);
                  }
                }
                int moved[L::HFB];
#pragma unroll
                for (int j = 0; j < L::HFB; j++) moved[j] = (ln + 64 * (j + 1) < nz) ? S.hf_zlist[ln + 64 * (j + 1)] : 0;
                const unsigned long long mm_ = __ballot(maybe);
                if (maybe) S.hf_list[nlist + __popcll(mm_ & lanemask_lt(ln))] = item;
                nlist += __popcll(mm_);
                WSYNC();
#pragma unroll
                for (int j = 0; j < L::HFB; j++) if (ln + 64 * (j + 1) < nz) S.hf_zlist[ln + 64 * j] = moved[j];
                nz -= cnt;
                WSYNC();
                PEXT_ADD(1);
                if (PROF) { pext[4] += 1; }
              } else if (KM == 1 && (nb >= 8 * L::HFB || walked)) {
                if constexpr (KM == 1) {
                  // per-prism pass over the prisms of up to 8 x HB listed blocks (8 lanes per block): the height test of
                  // mjc_ConvexHField and the oriented-box test, per prism; survivors -> zlist, in (geom, strip) order
                  constexpr int HB = L::HFB;
                  const int nbk = min(nb, 8 * HB);
                  bool alive[HB];
                  float hv[HB][3], lo2[HB], add[HB], cxy[HB][2];
                  int gsel[HB], itm[HB];
#pragma unroll
                  for (int j = 0; j < HB; j++) {
                    const int slot = (ln >> 3) + 8 * j, sub = ln & 7;
                    alive[j] = false;
                    lo2[j] = 0.f; add[j] = 0.f; cxy[j][0] = cxy[j][1] = 0.f; gsel[j] = 0; itm[j] = 0;
                    int idx[3] = {0, 0, 0};
                    if (slot < nbk) {
                      const int bi = S.hf_blist[slot];
                      int g = 0;
                      while (bi >= S.hf_bend[g]) g++;
                      const int kb = bi - (g > 0 ? S.hf_bend[g > 0 ? g - 1 : 0] : 0), ppr = S.hf_ppr[g], bpr = (ppr + 7) >> 3;
                      const int rrow = small_div(kb, bpr), kk = ((kb - rrow * bpr) << 3) + sub;
                      if (kk < ppr) {
                        const int r = S.hf_rmin[g] + rrow, cmin = S.hf_cmin[g];
                        itm[j] = (g > 0 ? S.hf_end[g > 0 ? g - 1 : 0] : 0) + rrow * ppr + kk;
                        lo2[j] = S.hf_lo[g]; add[j] = T.gz + S.hf_mg[g];
                        int csum = 0, rsum = 0;
#pragma unroll
                        for (int i = 0; i < 3; i++) {
                          const int v = kk + i, c = cmin + (v >> 1), rr = r + 1 - (v & 1);
                          idx[i] = rr * T.ncol + c;
                          csum += c; rsum += rr;
                        }
                        // centroid of the footprint, relative to the sub-grid's first vertex (fp32 is ample: the box is grown by 0.7 cell
                        // diagonals where 0.67 would do, and a sub-grid is at most a few metres wide)
                        gsel[j] = g;
                        cxy[j][0] = S.hf_org[g][0] + (float)(csum - 3 * cmin) * (1.f / 3.f) * (float)T.dx;
                        cxy[j][1] = S.hf_org[g][1] + (float)(rsum - 3 * S.hf_rmin[g]) * (1.f / 3.f) * (float)T.dy;
                        alive[j] = S.hf_cnt[g] < 50;
                      }
                    }
                    if (8 * j < nbk) {
#pragma unroll
                      for (int i = 0; i < 3; i++) hv[j][i] = T.data[idx[i]];
                    } else {
#pragma unroll
                      for (int i = 0; i < 3; i++) hv[j][i] = 0.f;
                    }
                  }
                  int moved[2];   // blocks left over: < 8 x HB + 64 x HBB, the first 8 x HB of them just consumed
#pragma unroll
                  for (int q = 0; q < 2; q++) moved[q] = (nbk + ln + 64 * q < nb) ? S.hf_blist[nbk + ln + 64 * q] : 0;
#pragma unroll
                  for (int j = 0; j < HB; j++) {
                    if (!(8 * j < nbk)) break;
                    bool below = true;
#pragma unroll
                    for (int i = 0; i < 3; i++) below = below && (hv[j][i] * T.sz + add[j] < lo2[j]);
                    // A prism that touches the geom holds a point p of the geom's box whose (x, y) lies in the prism's footprint, i.e.
                    // within `grow` of the footprint's centroid c: (c.x, c.y, p.z) then lies in the GROWN box, so the vertical line
                    // through c meets the grown box, and no lower than it enters it can p be.  Line misses the box, or enters it above
                    // the prism's (margin-raised) top: the prism cannot touch.  This is what thins out the walk of a long limb lying
                    // askew (its footprint is a sliver of its axis-aligned sub-grid) or tilted (one end high above the steps).
                    if (alive[j] && !below) {
                      const float* B = S.hf_box[gsel[j]];
                      const float rx = cxy[j][0] - B[0], ry = cxy[j][1] - B[1], rz = -B[2];   // line origin (c.x, c.y, 0) relative to the box centre
                      float t0 = -3.0e38f, t1 = 3.0e38f;
                      bool miss = false;
#pragma unroll
                      for (int k = 0; k < 3; k++) {
                        const float o_ = B[3 + k] * rx + B[6 + k] * ry + B[9 + k] * rz;   // R^T (origin - centre), component k
                        const float d_ = B[9 + k];                                           // R^T e_z, component k
                        const float hk = B[12 + k];
                        if (fabsf(d_) < 1e-6f) miss = miss || fabsf(o_) > hk;
                        else {
                          const float inv = 1.f / d_, ta = (-hk - o_) * inv, tb = (hk - o_) * inv;
                          t0 = fmaxf(t0, fminf(ta, tb)); t1 = fminf(t1, fmaxf(ta, tb));
                        }
                      }
                      const float ztop = fmaxf(hv[j][0], fmaxf(hv[j][1], hv[j][2])) * T.sz + add[j];
                      if (miss || t0 > t1 || t0 > ztop + 1e-5f) below = true;
                    }
                    const bool al = alive[j] && !below;
                    const unsigned long long am = __ballot(al);
                    if (al) S.hf_zlist[nz + __popcll(am & lanemask_lt(ln))] = itm[j];
                    nz += __popcll(am);
                  }
                  WSYNC();
#pragma unroll
                  for (int q = 0; q < 2; q++) if (nbk + ln + 64 * q < nb) S.hf_blist[ln + 64 * q] = moved[q];
                  nb -= nbk;
                  WSYNC();
                  PEXT_ADD(0);
                }
              } else if (KM == 1) {
                if constexpr (KM == 1) {
                  // block pass: HBB sub-batches of 64 blocks, their vertex heights (10 per block: 5 columns of two rows) in flight together
                  constexpr int HBB = L::HBB;
                  int g0 = 0;
                  while (base >= __builtin_amdgcn_readfirstlane(S.hf_bend[g0])) g0++;   // uniform; terminates: base < btotal = hf_bend[23]
                  if (__builtin_amdgcn_readfirstlane(S.hf_cnt[g0]) >= 50) {   // this geom has its 50 contacts: skip the rest of its blocks
                    base = __builtin_amdgcn_readfirstlane(S.hf_bend[g0]);
                  } else {
                    const float cell_reach = 0.7f * sqrtf((float)(T.dx * T.dx + T.dy * T.dy));   // what hf_box is grown by already
                    bool alive[HBB];
                    float hmax[HBB], lo2[HBB], add[HBB], cxy[HBB][2], extra[HBB];
                    int gsel[HBB];
#pragma unroll
                    for (int j = 0; j < HBB; j++) {
                      const int bi = base + ln + 64 * j;
                      alive[j] = false;
                      hmax[j] = 0.f; lo2[j] = 0.f; add[j] = 0.f; cxy[j][0] = cxy[j][1] = 0.f; extra[j] = 0.f; gsel[j] = 0;
                      int idx[10];
#pragma unroll
                      for (int i = 0; i < 10; i++) idx[i] = 0;
                      int nvx = 0;
                      if (bi < btotal) {
                        int g = g0;
                        while (bi >= S.hf_bend[g]) g++;
                        const int kb = bi - (g > 0 ? S.hf_bend[g > 0 ? g - 1 : 0] : 0), ppr = S.hf_ppr[g], bpr = (ppr + 7) >> 3;
                        const int rrow = small_div(kb, bpr), kk0 = (kb - rrow * bpr) << 3, nk = min(8, ppr - kk0);
                        const int r = S.hf_rmin[g] + rrow, cmin = S.hf_cmin[g];
                        nvx = nk + 2;   // strip vertices kk0 .. kk0 + nk + 1 carry the block's prisms
#pragma unroll
                        for (int i = 0; i < 10; i++) {
                          const int v = kk0 + min(i, nvx - 1), c = cmin + (v >> 1), rr = r + 1 - (v & 1);
                          idx[i] = rr * T.ncol + c;
                        }
                        lo2[j] = S.hf_lo[g]; add[j] = T.gz + S.hf_mg[g];
                        gsel[j] = g;
                        // centre of the block's footprint (columns kk0 / 2 .. (kk0 + nk + 1) / 2 of the sub-grid, one row of cells) and how
                        // far a point of the footprint lies from it at most, beyond what the box is grown by already
                        const int ca = kk0 >> 1, cb = (kk0 + nk + 1) >> 1;
                        const float hx = 0.5f * (float)(cb - ca) * (float)T.dx, hy = 0.5f * (float)T.dy;
                        cxy[j][0] = S.hf_org[g][0] + 0.5f * (float)(ca + cb) * (float)T.dx;
                        cxy[j][1] = S.hf_org[g][1] + ((float)rrow + 0.5f) * (float)T.dy;
                        extra[j] = fmaxf(sqrtf(hx * hx + hy * hy) * 1.0001f - cell_reach, 0.f);
                        alive[j] = S.hf_cnt[g] < 50;
                      }
                      if (j == 0 || __builtin_amdgcn_readfirstlane(base) + 64 * j < __builtin_amdgcn_readfirstlane(btotal)) {
                        float m_ = -3.0e38f;
#pragma unroll
                        for (int i = 0; i < 10; i++) m_ = fmaxf(m_, T.data[idx[i]]);
                        hmax[j] = m_;
                      }
                    }
#pragma unroll
                    for (int j = 0; j < HBB; j++) {
                      if (j > 0 && !(__builtin_amdgcn_readfirstlane(base) + 64 * j < __builtin_amdgcn_readfirstlane(btotal))) break;
                      // every prism of the block lies below the geom's lowest point <=> the block's highest vertex does
                      const float ztop = hmax[j] * T.sz + add[j];
                      bool below = ztop < lo2[j] && A.block_cull != 0;
                      // the oriented-box test of the per-prism pass for the whole block: a prism of the block that touches the geom
                      // holds a point of the geom's box over the block's footprint, within `reach` of its centre c; the vertical line
                      // through c then meets the box grown by that reach, no higher than the block's highest (margin-raised) vertex
                      if (alive[j] && !below && A.block_cull != 0) {
                        const float* B = S.hf_box[gsel[j]];
                        const float rx = cxy[j][0] - B[0], ry = cxy[j][1] - B[1], rz = -B[2];
                        float t0 = -3.0e38f, t1 = 3.0e38f;
                        bool miss = false;
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                          const float o_ = B[3 + k] * rx + B[6 + k] * ry + B[9 + k] * rz;
                          const float d_ = B[9 + k];
                          const float hk = B[12 + k] + extra[j];
                          if (fabsf(d_) < 1e-6f) miss = miss || fabsf(o_) > hk;
                          else {
                            const float inv = 1.f / d_, ta = (-hk - o_) * inv, tb = (hk - o_) * inv;
                            t0 = fmaxf(t0, fminf(ta, tb)); t1 = fminf(t1, fmaxf(ta, tb));
                          }
                        }
                        if (miss || t0 > t1 || t0 > ztop + 1e-5f) below = true;
                      }
                      const bool al = alive[j] && !below;
                      const unsigned long long am = __ballot(al);
                      if (al) S.hf_blist[nb + __popcll(am & lanemask_lt(ln))] = base + ln + 64 * j;
                      nb += __popcll(am);
                    }
                    base += 64 * HBB;
                    WSYNC();
                  }
                  walked = base >= btotal;
                  PEXT_ADD(0);
                }
              } else {
                int g0 = 0;
                while (base >= __builtin_amdgcn_readfirstlane(S.hf_end[g0])) g0++;   // uniform; terminates: base < total = hf_end[23]
                if (__builtin_amdgcn_readfirstlane(S.hf_cnt[g0]) >= 50) {   // this geom has its 50 contacts: skip the rest of its prisms
                  base = __builtin_amdgcn_readfirstlane(S.hf_end[g0]);
                } else {
                  // HB sub-batches of 64 items per pass, their height loads in flight together (one dependent trip per 64 items
                  // made this pass a chain of memory latencies: 217 trips per substep for a fallen humanoid on 1 cm cells)
                  constexpr int HB = L::HFB;
                  bool alive[HB];
                  float hv[HB][3], lo2[HB], add[HB];
#pragma unroll
                  for (int j = 0; j < HB; j++) {
                    const int item = base + ln + 64 * j;
                    alive[j] = false;
                    lo2[j] = 0.f; add[j] = 0.f;
                    int idx[3] = {0, 0, 0};
                    if (item < total) {
                      int g = g0;
                      while (item >= S.hf_end[g]) g++;
                      const int k = item - (g > 0 ? S.hf_end[g > 0 ? g - 1 : 0] : 0), ppr = S.hf_ppr[g], rrow = small_div(k, ppr), kk = k - rrow * ppr;
                      const int r = S.hf_rmin[g] + rrow, cmin = S.hf_cmin[g];
                      lo2[j] = S.hf_lo[g]; add[j] = T.gz + S.hf_mg[g];
#pragma unroll
                      for (int i = 0; i < 3; i++) {
                        const int v = kk + i, c = cmin + (v >> 1), rr = r + 1 - (v & 1);
                        idx[i] = rr * T.ncol + c;
                      }
                      alive[j] = S.hf_cnt[g] < 50;
                    }
                    // (a sub-batch past the end of the list is skipped by the whole wave: short walks pay for one)
                    if (j == 0 || __builtin_amdgcn_readfirstlane(base) + 64 * j < __builtin_amdgcn_readfirstlane(total)) {
#pragma unroll
                      for (int i = 0; i < 3; i++) hv[j][i] = T.data[idx[i]];
                    } else {
#pragma unroll
                      for (int i = 0; i < 3; i++) hv[j][i] = 0.f;
                    }
                  }
#pragma unroll
                  for (int j = 0; j < HB; j++) {
                    if (j > 0 && !(__builtin_amdgcn_readfirstlane(base) + 64 * j < __builtin_amdgcn_readfirstlane(total))) break;
                    bool below = true;
#pragma unroll
                    for (int i = 0; i < 3; i++) below = below && (hv[j][i] * T.sz + add[j] < lo2[j]);
                    const bool al = alive[j] && !below;
                    const unsigned long long am = __ballot(al);
                    if (al) S.hf_zlist[nz + __popcll(am & lanemask_lt(ln))] = base + ln + 64 * j;
                    nz += __popcll(am);
                  }
                  base += 64 * HB;
                  WSYNC();
                }
                walked = base >= total;
                PEXT_ADD(0);
              }
            }
            ncon = S.ncon_ctr;
            // hulls with few prisms: wave-cooperative, one (geom, prism) at a time (hfield_geom: the same walk, sequential)
            for (unsigned long long cm_ = __ballot(coop_geom); cm_; cm_ &= cm_ - 1) {
              const int g = __builtin_ctzll(cm_);
              const auto& G = dm.rec[g];
              CObj o;
              make_cobj(o, g);
              float gctr[3];
              {
                float v[3];
                const float gq[4] = {S.xquat[G.g_body][0], S.xquat[G.g_body][1], S.xquat[G.g_body][2], S.xquat[G.g_body][3]};
                qrot(v, gq, G.g_rcenter);
                for (int k = 0; k < 3; k++) gctr[k] = S.xpos[G.g_body][k] + v[k];
              }
              int cg = 0;   // narrowphase kernel: hits of this geom so far
              hfield_geom<GTM, true>(T, o, gctr, G.g_rbound, G.g_margin, dm.hfield_size[3], HG, ln, [&](float dist, const float* pos, const float* n) {
                if constexpr (KM == 1) {
                  if (ln == 0 && cg < XC) {
                    float4* oo = reinterpret_cast<float4*>(xcon_ + ((size_t)g * XC + cg) * 8);
                    oo[0] = make_float4(dist, pos[0], pos[1], pos[2]);
                    oo[1] = make_float4(n[0], n[1], n[2], 0.f);
                  }
                  cg++;
                } else {
                  if (ncon < MC && ln == 0) {
                    S.cdist[ncon] = dist;
                    S.cgeom[ncon] = g;
                    for (int k = 0; k < 3; k++) { S.cpos[ncon][k] = pos[k]; S.cnrm[NRM ? ncon : 0][k] = n[k]; }
                  }
                  ncon++;
                }
              }, PROF ? pext + 8 : nullptr);   // heightfield kernels: [24..29] = cooperative walk: box cycles, MPR runs, MPR cycles, geoms, hits, refinement iterations
              if constexpr (KM == 1) { if (ln == 0) S.hf_cnt[g] = min(cg, XC); }
            }
            if constexpr (KM == 1) {
              // the narrowphase kernel ends here: contacts found per owned geom (every launch overwrites them), counters, done
              WSYNC();
              if (mine) A.xcnt[(size_t)env * XG + ln] = S.hf_cnt[ln];
              if (ln == 0 && st_walkcut > 0) { atomicAdd(&meta[8], st_walkcut); atomicAdd(&meta[13], st_walkcut); }
              if (PROF && A.dbg != nullptr && ln == 0) {   // diagnostic build: wave lifetime (sum, max, count), the walk's phases, items / batches
                unsigned long long* D = reinterpret_cast<unsigned long long*>(A.dbg);
                const unsigned long long life = __builtin_amdgcn_s_memtime() - pt_wave0;
                atomicAdd(D + 0, life); atomicMax(D + 1, life); atomicAdd(D + 2, 1ull);
                atomicAdd(D + 3, (unsigned long long)total);          // work items of this wave's geoms
                for (int i = 0; i < 8; i++) atomicAdd(D + 8 + i, pext[i]);
                if (life > 400000ull) { atomicAdd(D + 4, 1ull); atomicAdd(D + 5, (unsigned long long)total); atomicAdd(D + 6, pext[0]); atomicAdd(D + 7, pext[1] + pext[2]); }
              }
              return;
            }
          }
        }
        // convex meshes near the ground, one at a time, all lanes sharing the scans over the hull's vertices
        unsigned long long mm = grp_ballot<LW>(mesh_near, hb);
        while (mm) {
          const int g = __builtin_ctzll(mm);
          mm &= mm - 1;
          const auto& G = dm.rec[g];
          const int gb = G.g_body, adr = G.g_hulladr, num = G.g_hullnum;
          float gq[4] = {S.xquat[gb][0], S.xquat[gb][1], S.xquat[gb][2], S.xquat[gb][3]}, m[9];
          q2m(m, gq);
          const float gxp[3] = {S.xpos[gb][0], S.xpos[gb][1], S.xpos[gb][2]};
          const float gmargin = G.g_margin, grb = G.g_rbound;
          float gctr[3];
          {
            float v[3];
            qrot(v, gq, G.g_rcenter);
            for (int k = 0; k < 3; k++) gctr[k] = gxp[k] + v[k];
          }
          // plane: mjc_PlaneConvex -- the support vertex, then its hull neighbours within the margin (at most 4 contacts)
          const float P0[3] = {0.f, 0.f, T.gz}, n[3] = {0.f, 0.f, 1.f};
          const float lnv[3] = {m[0] * n[0] + m[3] * n[1] + m[6] * n[2], m[1] * n[0] + m[4] * n[1] + m[7] * n[2], m[2] * n[0] + m[5] * n[1] + m[8] * n[2]};  // R^T n
          const float offn = n[0] * (gxp[0] - P0[0]) + n[1] * (gxp[1] - P0[1]) + n[2] * (gxp[2] - P0[2]);
          float best = 3.0e38f;
          int besti = 0x7fffffff;
          const int hmap = dm.g_hullmap[g];
          if (hmap >= 0) {
            // support map (cosim_hullmap.h): the lowest vertex is the support point along -n; the candidates of that cell, one per lane
            const float dn[3] = {-lnv[0], -lnv[1], -lnv[2]};
            const float4* rec = A.hull_cell + (size_t)HM_REC * (hmap + support_cell(dn));   // (uniform over the env's LW lanes)
            const float4 hd = rec[0], x0 = rec[1 + (ln & (HM_INLINE - 1))];   // one trip: header + inline candidates (lane k < 4: candidate k)
            const int cn = __float_as_int(hd.x);
            {
              const float dist = offn + hull_dot(lnv, x0);
              if (ln < HM_INLINE && ln < cn) { best = dist; besti = __float_as_int(x0.w); }
            }
            if (cn > HM_INLINE) {
              const float4* cp = A.hull_cand + __float_as_int(hd.y);
              const int rest = cn - HM_INLINE;
              for (int i0 = 0; i0 < rest; i0 += LW) {
                const float4 x = cp[min(i0 + ln, rest - 1)];
                const float dist = offn + hull_dot(lnv, x);
                if (i0 + ln < rest && dist < best) { best = dist; besti = __float_as_int(x.w); }
              }
            }
          } else {
            // four vertices per lane in flight per trip (16 bytes per vertex: one load each)
            const float4* hv = reinterpret_cast<const float4*>(A.hull_vert) + adr;
            for (int i0 = 0; i0 < num; i0 += 4 * LW) {
              float4 x[4];
#pragma unroll
              for (int k = 0; k < 4; k++) x[k] = hv[min(i0 + ln + LW * k, num - 1)];
#pragma unroll
              for (int k = 0; k < 4; k++) {
                const int i = i0 + ln + LW * k;
                const float dist = offn + hull_dot(lnv, x[k]);
                if (i < num && dist < best) { best = dist; besti = i; }
              }
            }
          }
          const float bmin = grp_min<LW>(best);
          if (!(bmin <= gmargin)) continue;
          const int bi = (int)grp_min<LW>(best == bmin ? (float)besti : 3.0e38f);  // lowest vertex index among ties, like a sequential scan
          // candidates of mjc_PlaneConvex: the support vertex (candidate 0), then its hull neighbours in list order; the
          // first four within the margin become contacts.  One lane per candidate: one round of dependent loads, not a
          // serial walk that the whole wave would wait on.
          const int lo = A.hull_adr[adr + bi], nnb = A.hull_adr[adr + bi + 1] - lo;
          int added = 0;
          for (int c0 = 0; c0 < nnb + 1 && added < 4; c0 += LW) {
            const int cand = c0 + ln;
            bool okc = false;
            float dist = 0.f, cpw[3] = {0.f, 0.f, 0.f};
            if (cand < nnb + 1) {
              const int i = cand == 0 ? bi : A.hull_nbr[lo + cand - 1];
              const float4 v4 = reinterpret_cast<const float4*>(A.hull_vert)[adr + i];
              const float v[3] = {v4.x, v4.y, v4.z};
              dist = offn + hull_dot(lnv, v4);
              okc = !(dist > gmargin);
              const float w[3] = {m[0] * v[0] + m[1] * v[1] + m[2] * v[2], m[3] * v[0] + m[4] * v[1] + m[5] * v[2], m[6] * v[0] + m[7] * v[1] + m[8] * v[2]};
              for (int k = 0; k < 3; k++) cpw[k] = gxp[k] + w[k] - n[k] * dist * 0.5f;
            }
            const unsigned long long km = grp_ballot<LW>(okc, hb);
            const int rank = added + __popcll(km & lanemask_lt(ln));
            if (okc && rank < 4 && ncon + rank < MC) {
              const int slot = ncon + rank;
              S.cdist[slot] = dist;
              S.cgeom[slot] = g;
              for (int k = 0; k < 3; k++) { S.cpos[slot][k] = cpw[k]; if (NRM) S.cnrm[NRM ? slot : 0][k] = n[k]; }
            }
            added += __popcll(km);
          }
          ncon += min(added, 4);
        }
      }

      unsigned long long pc0_ = 0;
      if (PROF) { __builtin_amdgcn_s_waitcnt(0); pc0_ = __builtin_amdgcn_s_memtime(); }
      if constexpr (SC) {
        // =========================================================== robot-robot pairs (mjc_Convex: one MPR contact per pair)
        // candidates: the compiled pair list (contype/conaffinity, same-body, parent-child and <exclude> filters applied)
        // cut down by bounding spheres; primitive pairs run lane-parallel, pairs with a mesh one at a time wave-wide
        auto pair_put = [&](int slot, float dist, int code, const float* pp, const float* nn) {
          if constexpr (CT) {
            if (slot < L::MCP) {
              S.pdist[slot] = dist; S.pgeom[slot] = code;
              for (int k = 0; k < 3; k++) { S.ppos[slot][k] = pp[k]; S.pnrm[slot][k] = nn[k]; }
            }
          } else if (slot < MC) {
            S.cdist[slot] = dist; S.cgeom[slot] = code;
            for (int k = 0; k < 3; k++) { S.cpos[slot][k] = pp[k]; S.cnrm[NRM ? slot : 0][k] = nn[k]; }
          }
        };
        const int npair = dm.npair;
        for (int p0 = 0; p0 < npair; p0 += 64) {
          const int p = p0 + ln;
          bool cand = false, mesh = false, boxes = false;
          int g1 = 0, g2 = 0;
          if (p < npair) {
            const unsigned pk = A.pairs[p];
            g1 = pk & 0xffffu; g2 = pk >> 16;
            const auto &G1 = dm.rec[g1], &G2 = dm.rec[g2];
            const float q1[4] = {S.xquat[G1.g_body][0], S.xquat[G1.g_body][1], S.xquat[G1.g_body][2], S.xquat[G1.g_body][3]};
            const float q2[4] = {S.xquat[G2.g_body][0], S.xquat[G2.g_body][1], S.xquat[G2.g_body][2], S.xquat[G2.g_body][3]};
            float v1[3], v2[3];
            qrot(v1, q1, G1.g_rcenter);
            qrot(v2, q2, G2.g_rcenter);
            float d2 = 0.f;
            for (int k = 0; k < 3; k++) { const float dk = (S.xpos[G2.g_body][k] + v2[k]) - (S.xpos[G1.g_body][k] + v1[k]); d2 += dk * dk; }
            const float mg = fmaxf(G1.g_margin, G2.g_margin), rs = G1.g_rbound + G2.g_rbound + mg;
            cand = d2 <= rs * rs;
            if (cand) {
              // oriented boxes (body-frame box around each geom): centre line and the six face normals as separating axes
              float m1[9], m2[9], dv[3];
              q2m(m1, q1);
              q2m(m2, q2);
              for (int k = 0; k < 3; k++) dv[k] = (S.xpos[G2.g_body][k] + v2[k]) - (S.xpos[G1.g_body][k] + v1[k]);
              auto radius = [](const float* m, const auto* hf, const float* a) {
                return hf[0] * fabsf(m[0] * a[0] + m[3] * a[1] + m[6] * a[2]) + hf[1] * fabsf(m[1] * a[0] + m[4] * a[1] + m[7] * a[2]) +
                       hf[2] * fabsf(m[2] * a[0] + m[5] * a[1] + m[8] * a[2]);
              };
              const float dn = sqrtf(d2);
              if (dn > 1e-9f) {
                const float a[3] = {dv[0] / dn, dv[1] / dn, dv[2] / dn};
                if (dn > radius(m1, G1.g_half, a) + radius(m2, G2.g_half, a) + mg) cand = false;
              }
#pragma unroll
              for (int k = 0; k < 3; k++) {
                const float a1[3] = {m1[k], m1[3 + k], m1[6 + k]}, a2[3] = {m2[k], m2[3 + k], m2[6 + k]};
                if (fabsf(dot3(dv, a1)) > G1.g_half[k] + radius(m2, G2.g_half, a1) + mg) cand = false;
                if (fabsf(dot3(dv, a2)) > G2.g_half[k] + radius(m1, G1.g_half, a2) + mg) cand = false;
              }
              // ... and the nine edge x edge axes (the full box-box separating-axis test): a pair that survives costs the whole wave an
              // MPR run over up to 700-vertex hulls, a rejected one costs this lane a few dozen FMAs
              if (cand) {
#pragma unroll
                for (int i = 0; i < 3; i++) {
                  const float a1[3] = {m1[i], m1[3 + i], m1[6 + i]};
#pragma unroll
                  for (int j = 0; j < 3; j++) {
                    const float a2[3] = {m2[j], m2[3 + j], m2[6 + j]};
                    float ax[3];
                    cross(ax, a1, a2);
                    const float n2 = dot3(ax, ax);
                    if (n2 > 1e-8f && fabsf(dot3(dv, ax)) > radius(m1, G1.g_half, ax) + radius(m2, G2.g_half, ax) + mg * sqrtf(n2)) cand = false;
                  }
                }
              }
            }
            // pairs with a convex hull: one at a time with all 64 lanes sharing the vertex scans (cosim_set_param "pair_mode" 0 runs them
            // lane-parallel, every lane scanning its own hulls: measured slower on the 700-vertex wheel hulls)
            if constexpr ((GTM & ~GT_MESH) == 0) mesh = cand;   // every geom is a hull (flamingo_p_v3, w4_p_v2): no lane-parallel route is compiled in
            else mesh = cand && A.pair_coop && (G1.g_type == CS_GEOM_MESH || G2.g_type == CS_GEOM_MESH);
            if (PROF && !HF) { const unsigned long long sm = __ballot(d2 <= rs * rs); if (ln == 0) pext[12] += __popcll(sm); }   // pairs past the bounding spheres
            if constexpr ((GTM & GT_BOX) != 0) boxes = cand && A.pair_boxbox && G1.g_type == CS_GEOM_BOX && G2.g_type == CS_GEOM_BOX;
          }
          bool hit = false;
          float depth = 0.f, cn[3] = {0.f, 0.f, 1.f}, cp[3] = {0.f, 0.f, 0.f};
          if constexpr ((GTM & ~GT_MESH) != 0) {
            if (cand && !mesh && !boxes) {
              CObj o1, o2;
              make_cobj(o1, g1);
              make_cobj(o2, g2);
              const MprPair<GTM, false> sup{o1, o2, HG, ln};
              hit = mpr_penetration(sup, o1.center, o2.center, depth, cn, cp) && (cn[0] != 0.f || cn[1] != 0.f || cn[2] != 0.f);
            }
          }
          {
            const unsigned long long hm = __ballot(hit);
            const int slot = (CT ? npc : ncon) + __popcll(hm & lanemask_lt(ln));
            if (hit) pair_put(slot, -depth, g2 | ((g1 + 1) << 8), cp, cn);
            if constexpr (CT) npc += __popcll(hm); else ncon += __popcll(hm);
          }
          if constexpr ((GTM & GT_BOX) != 0) {
            // box-box (mjc_BoxBox): one pair at a time, the lanes share the clipping of the incident face (cosim_boxbox.h)
            unsigned long long bm = __ballot(boxes);
            while (bm) {
              const int src = __builtin_ctzll(bm);
              bm &= bm - 1;
              const int h1 = __shfl(g1, src, 64), h2 = __shfl(g2, src, 64);
              CObj o1, o2;
              make_cobj(o1, h1);
              make_cobj(o2, h2);
              float bp[3], bn[3] = {0.f, 0.f, 1.f}, bd = 0.f;
              const bool okb = box_box_lane(o1.pos, o1.q, o1.size, o2.pos, o2.q, o2.size, fmaxf(dm.rec[h1].g_margin, dm.rec[h2].g_margin), ln, bp, bd, bn);
              const unsigned long long km = __ballot(okb);
              const int rank = __popcll(km & lanemask_lt(ln)), cnt = min((int)__popcll(km), 8);
              if (okb && rank < 8) pair_put((CT ? npc : ncon) + rank, bd, h2 | ((h1 + 1) << 8), bp, bn);
              if constexpr (CT) npc += cnt; else ncon += cnt;
            }
          }
          if constexpr ((GTM & GT_MESH) != 0) {
            unsigned long long mm = __ballot(mesh);
            while (mm) {
              const int src = __builtin_ctzll(mm);
              mm &= mm - 1;
              const int h1 = __shfl(g1, src, 64), h2 = __shfl(g2, src, 64);
              CObj o1, o2;
              make_cobj(o1, h1);
              make_cobj(o2, h2);
              const MprPair<GTM, true> sup{o1, o2, HG, ln};
              float dep2 = 0.f, n2[3] = {0.f, 0.f, 1.f}, c2[3] = {0.f, 0.f, 0.f};
              int nit2 = 0;
              unsigned long long tm0_ = 0;
              if (PROF) tm0_ = __builtin_amdgcn_s_memtime();
              const bool hit2 = mpr_penetration(sup, o1.center, o2.center, dep2, n2, c2, PROF ? &nit2 : nullptr) && (n2[0] != 0.f || n2[1] != 0.f || n2[2] != 0.f);
              if (PROF && !HF) { pext[8] += 1; pext[9] += hit2 ? 1 : 0; pext[10] += nit2; pext[11] += __builtin_amdgcn_s_memtime() - tm0_; }   // hull pairs: run, hit, refinement iterations, cycles
              if (hit2) {
                if (ln == 0) pair_put(CT ? npc : ncon, -dep2, h2 | ((h1 + 1) << 8), c2, n2);
                if constexpr (CT) npc++; else ncon++;
              }
            }
          }
        }
      }

      if (PROF && !HF) { __builtin_amdgcn_s_waitcnt(0); pext[0] += __builtin_amdgcn_s_memtime() - pc0_; }   // flat kernels: [16] = robot-robot pairs (heightfield kernels use pext[0..2] for the prism walk; pairs = collision - those)
      STAMP(5);   // collision
      // =========================================================== constraint rows (lane = row)
      const int ne = 3 * dm.neq, nf = dm.nfric;
      int nl = 0;
      // ---- dof rows (lane = dof): the friction-loss row and the joint-limit row of this lane's dof.  Their Jacobians are +-e_dof, so
      // J a is a lane read of qacc, J^T f a lane add and J^T D J a diagonal add: no dense row, no lane slot, no capacity to run out of
      // (mj_instantiateFriction / mj_instantiateLimit, engine_core_constraint.c).  fD / lD = 0: no such row on this dof.
      float fD = 0.f, fRf = 0.f, fls = 0.f, faref = 0.f;   // friction loss: 1 / R, R f (half-width of the quadratic zone), f, aref
      float lD = 0.f, lsg = 1.f, laref = 0.f;              // limit: 1 / R, sign of the Jacobian entry, aref
      {
        bool lo_v = false, hi_v = false;
        if (ln < NV) {
          const auto& R = dm.rec[ln];
          const float dinvw = S.p_dinvw[ln], vel = S.qvel[ln];
          const float fl = S.p_floss[ln];
          if (fl > 0.f) {   // a per-env value of zero leaves the dof without the row
            const float imp = impedance(R.d_solimp, 0.f, 0.f);
            const float rR = fmaxf(MINVAL, (1.f - imp) * dinvw / imp);
            fD = 1.f / rR; fRf = rR * fl; fls = fl;
            faref = -R.d_solref[1] * vel;   // K = 0 for friction-loss rows
          }
          const auto& B = dm.rec[R.d_body];
          if (B.b_jtype == CS_JNT_HINGE && B.j_limited) {
            const float q = S.qpos[B.b_qadr];
            const float dlo = q - B.j_range[0], dhi = B.j_range[1] - q;
            lo_v = dlo < B.j_margin;
            hi_v = dhi < B.j_margin;
            if (lo_v || hi_v) {
              // (both sides inside the margin at once needs margin > range / 2: the lower one is kept, the other counted as left out)
              const float pos = lo_v ? dlo : dhi;
              lsg = lo_v ? 1.f : -1.f;
              const float imp = impedance(B.j_solimp, pos, B.j_margin);
              const float rR = fmaxf(MINVAL, (1.f - imp) * dinvw / imp);
              lD = 1.f / rR;
              laref = -B.j_solref[1] * (lsg * vel) - B.j_solref[0] * imp * (pos - B.j_margin);
            }
          }
        }
        const unsigned long long ml = grp_ballot<LW>(lo_v, hb), mh = grp_ballot<LW>(hi_v, hb);
        nl = __popcll(ml) + __popcll(mh);
        st_droplim += __popcll(ml & mh);
      }
      {
        // capacity (MuJoCo's arena holds every contact; here the slots are sized per kernel variant): whatever does not fit is
        // left out in detection order AND counted -- meta[8] / solver_stats()["dropped_contacts"] is non-zero whenever an env
        // was stepped with a truncated constraint set
        const int ncon_all = ncon + npc;
        int room = (MAXROWS - ne) / 4;
        if (room < 0) room = 0;
        if constexpr (CT) {   // ground contacts: slots only; robot-robot contacts: slots and dense rows
          if (ncon > MC) ncon = MC;
          if (npc > room) npc = room;
          if (npc > L::MCP) npc = L::MCP;
          if (ne + 4 * npc > NGENMAX) npc = (NGENMAX - ne) / 4;
        } else {
          if (ncon > room) ncon = room;
          if (ncon > MC) ncon = MC;
          if (ne + 4 * ncon > NGENMAX) ncon = (NGENMAX - ne) / 4;
        }
        if constexpr (!FIX && EPW == 1) {
          // more contacts than this kernel has slots for, and the engine has a large-capacity kernel: give the step up (nothing of
          // it has been written) and flag the env; env_fixup_kernel redoes it from the same state right after this launch
          // (readfirstlane: heightfield kernels read the count from LDS, which the compiler must take for a per-lane value)
          if (kmode == MODE_STEP && A.ovf != nullptr && __builtin_amdgcn_readfirstlane(ncon_all - ncon - npc) > 0) {
            if (ln == 0) A.ovf[env] = 1 + kstep;   // (rollout launch: the step the large-capacity kernel takes over at)
            return;
          }
        }
        st_dropcon += ncon_all - ncon - npc;
        st_maxcon = max(st_maxcon, ncon_all);
      }
      const int ndc = CT ? npc : ncon;      // contacts with dense rows
      const int ngen = ne + 4 * ndc;        // dense rows (J row in LDS, one per lane slot): equality + contact
      const int nefc = ngen + nf + nl + (CT ? 4 * ncon : 0);   // + the dof rows, + (CT) the ground contacts' rows in twist space
      WSYNC();

      // per-lane dense row state: row (ln + LW rr), rr < RPL
      bool rdense[RPL];
      // x < rhi: the row's quadratic zone in x = J a - aref (equality: all x; contact: x < 0); beyond it the cost is zero
      float rD[RPL], rhi[RPL], raref[RPL], rpos_dbg = 0.f;
#pragma unroll
      for (int rr = 0; rr < RPL; rr++) {
        const int row = ln + LW * rr;
        rdense[rr] = row < ngen; rD[rr] = 0.f; rhi[rr] = 3.0e38f; raref[rr] = 0.f;
        float rpos = 0.f, rmargin = 0.f, rdiagA = 0.f, rmu = 0.f, rvel = 0.f;
        float rsolref[2] = {0.f, 0.f} /* (K, B) */, rsolimp[5] = {0.9f, 0.95f, 0.001f, 0.5f, 2.f};
        if (row < ngen) {
          float* Jr = S.J[row];
#pragma unroll
          for (int d = 0; d < NV; d++) Jr[d] = 0.f;
          // equality and contact rows share one Jacobian loop: J = +jac(chain A, point A) - jac(chain B, point B) along `dir`
          float dir[3], offA[3], offB[3];
          unsigned maskA, maskB;
          const bool is_eq = row < ne;
          if (is_eq) {
            const int e = row / 3, comp = row - 3 * e;
            const auto& E = dm.rec[e];
            const int b1 = E.e_body1, b2 = E.e_body2;
            dir[0] = comp == 0 ? 1.f : 0.f; dir[1] = comp == 1 ? 1.f : 0.f; dir[2] = comp == 2 ? 1.f : 0.f;
            float p1[3], p2[3], v[3], q1[4] = {S.xquat[b1][0], S.xquat[b1][1], S.xquat[b1][2], S.xquat[b1][3]},
                                      q2[4] = {S.xquat[b2][0], S.xquat[b2][1], S.xquat[b2][2], S.xquat[b2][3]};
            qrot(v, q1, E.e_anchor1);
            for (int k = 0; k < 3; k++) p1[k] = S.xpos[b1][k] + v[k];
            qrot(v, q2, E.e_anchor2);
            for (int k = 0; k < 3; k++) p2[k] = S.xpos[b2][k] + v[k];
            rpos = p1[comp] - p2[comp];
            for (int k = 0; k < 3; k++) { offA[k] = p1[k] - com[k]; offB[k] = p2[k] - com[k]; }
            maskA = dm.rec[b1].b_dofmask; maskB = dm.rec[b2].b_dofmask;
            rdiagA = S.p_binvw[b1] + S.p_binvw[b2];
            for (int k = 0; k < 2; k++) rsolref[k] = E.e_solref[k];
            for (int k = 0; k < 5; k++) rsolimp[k] = E.e_solimp[k];
          } else {
            const int c = (row - ne) >> 2, edge = (row - ne) & 3;
            const int gg = con_geom<CT>(S, !CT, c, c), g = gg & 0xff, g1 = (gg >> 8) - 1;   // g1 < 0: geom1 is the ground (CT: dense rows are robot-robot contacts)
            const auto& G = dm.rec[g];
            const int b = G.g_body, b1 = (SC && g1 >= 0) ? dm.rec[g1].g_body : 0;
            const float mu = (SC && g1 >= 0) ? fmaxf(MINMU, fmaxf(A.gext[g].w, A.gext[g1].w)) : S.p_gmu[g];
            const float* cn_ = con_nrm<CT, NRM>(S, !CT, c, c);
            const float* cp_ = con_pos<CT>(S, !CT, c, c);
            float nrm[3] = {NRMD ? cn_[0] : 0.f, NRMD ? cn_[1] : 0.f, NRMD ? cn_[2] : 1.f}, t1[3], t2[3];
            make_frame(nrm, t1, t2);
            const float* tk = (edge >> 1) ? t2 : t1;
            const float sg = (edge & 1) ? -mu : mu;
            for (int k = 0; k < 3; k++) { dir[k] = nrm[k] + sg * tk[k]; offA[k] = cp_[k] - com[k]; offB[k] = offA[k]; }
            // mj_jacDifPair: jac(body2) - jac(body1) at one point; dofs common to both chains cancel
            const unsigned m2 = dm.rec[b].b_dofmask, m1 = (SC && g1 >= 0) ? dm.rec[b1].b_dofmask : 0u;
            maskA = m2 & ~m1; maskB = m1 & ~m2;
            rpos = con_dist<CT>(S, !CT, c, c);
            rmargin = G.g_incmargin;
            rmu = mu;
            rdiagA = (S.p_binvw[b] + ((SC && g1 >= 0) ? S.p_binvw[b1] : 0.f)) * (1.f + mu * mu);
            for (int k = 0; k < 2; k++) rsolref[k] = G.g_solref[k];
            for (int k = 0; k < 5; k++) rsolimp[k] = G.g_solimp[k];
          }
          {
            float odA[3], odB[3];
            cross(odA, offA, dir);
            cross(odB, offB, dir);
            for (unsigned mk = maskA | maskB; mk; mk &= mk - 1) {
              const int j = __builtin_ctz(mk);
              const float lin = dir[0] * S.cdof[j][3] + dir[1] * S.cdof[j][4] + dir[2] * S.cdof[j][5];
              const float a0 = S.cdof[j][0], a1 = S.cdof[j][1], a2 = S.cdof[j][2];
              float v = 0.f;
              if ((maskA >> j) & 1u) v += lin + odA[0] * a0 + odA[1] * a1 + odA[2] * a2;
              if ((maskB >> j) & 1u) v -= lin + odB[0] * a0 + odB[1] * a1 + odB[2] * a2;
              Jr[j] = v;
              rvel += v * S.qvel[j];   // J qvel over the dofs the row touches
            }
          }
          // KBIP, R, D, aref (mj_makeImpedance; K and B precomputed on the host)
          const float imp = impedance(rsolimp, rpos, rmargin);
          float rR = fmaxf(MINVAL, (1.f - imp) * rdiagA / imp);
          if (!is_eq) { const float mu = rmu * rsqrtf(fmaxf(MINVAL, dm.impratio)); rR = 2.f * mu * mu * rR; rhi[rr] = 0.f; }   // contact: quadratic for x < 0, zero beyond
          rD[rr] = 1.f / rR;
          raref[rr] = -rsolref[1] * rvel - rsolref[0] * imp * (rpos - rmargin);
        }
        if (rr == 0) rpos_dbg = rpos;
      }
      // ---- CT: ground contacts in twist space
      struct CGeo { float n[3], t1[3], t2[3], off[3], mu; int b, g; };
      auto cgeo = [&](int c, CGeo& G) {   // frame, lever arm about the tree's CoM, friction and body of ground contact c
        G.g = S.cgeom[c] & 0xff;
        G.b = dm.rec[G.g].g_body;
        G.mu = S.p_gmu[G.g];
        for (int k = 0; k < 3; k++) { G.n[k] = NRM ? S.cnrm[NRM ? c : 0][k] : (k == 2 ? 1.f : 0.f); G.off[k] = S.cpos[c][k] - S.com[k]; }   // S.com, not the register copy: that copy lives across the whole solver and was reloaded from scratch here, inside the Newton loop
        make_frame(G.n, G.t1, G.t2);
      };
      auto point_proj = [&](const float* tw, const CGeo& G, float& pn, float& p1, float& p2) {   // (n, t1, t2) . (v + w x off)
        float wxo[3];
        cross(wxo, tw, G.off);
        const float vp[3] = {tw[3] + wxo[0], tw[4] + wxo[1], tw[5] + wxo[2]};
        pn = dot3(G.n, vp); p1 = dot3(G.t1, vp); p2 = dot3(G.t2, vp);
      };
      auto body_twists = [&](const float* v1, const float* v2, float (*o1)[6], float (*o2)[6]) {   // o1[b] = sum_j cdof_j v1_j over chain(b); o2[b] likewise for v2 (may be null)
        if (ln > 0 && ln < nbody) {
          float a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, b2[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          for (unsigned mk = dm.rec[ln].b_dofmask; mk; mk &= mk - 1) {
            const int j = __builtin_ctz(mk);
            const float x1 = v1[j], x2 = v2 != nullptr ? v2[j] : 0.f;
#pragma unroll
            for (int q = 0; q < 6; q++) { const float cq = S.cdof[j][q]; a[q] += cq * x1; b2[q] += cq * x2; }
          }
#pragma unroll
          for (int q = 0; q < 6; q++) { o1[ln][q] = a[q]; if (v2 != nullptr) o2[ln][q] = b2[q]; }
        }
        WSYNC();
      };
      if constexpr (CT) {
        if (ln == 0) S.cbmask = 0u;
        body_twists(S.qvel, S.qacc, S.tw, S.bw);   // twists of the velocity and of the warm-start acceleration
#pragma unroll
        for (int cc = 0; cc < CPL; cc++) {
          const int c = ln + LW * cc;
          if (c < ncon) {
            CGeo G;
            cgeo(c, G);
            const auto& GR = dm.rec[G.g];
            float vn, v1, v2, an, a1, a2;
            point_proj(S.tw[G.b], G, vn, v1, v2);
            point_proj(S.bw[G.b], G, an, a1, a2);
            const float rpos = S.cdist[c], rmargin = GR.g_incmargin;
            const float imp = impedance(GR.g_solimp, rpos, rmargin), K = GR.g_solref[0], B = GR.g_solref[1];
            float rR = fmaxf(MINVAL, (1.f - imp) * S.p_binvw[G.b] * (1.f + G.mu * G.mu) / imp);
            const float mur = G.mu * rsqrtf(fmaxf(MINVAL, dm.impratio));
            rR = 2.f * mur * mur * rR;
            S.cD[c] = 1.f / rR;
            // row e = n +- mu t_k:  J a - aref = (a_n +- mu a_k) + B (v_n +- mu v_k) + K imp (r - margin)
            const float base = an + B * vn + K * imp * (rpos - rmargin), x1 = G.mu * (a1 + B * v1), x2 = G.mu * (a2 + B * v2);
            S.cJar[c][0] = base; S.cJar[c][1] = x1; S.cJar[c][2] = x2;   // rows: base +- x1, base +- x2
            atomicOr(&S.cbmask, 1u << G.b);
          }
        }
      }
      WSYNC();

      STAMP(6);   // constraint rows
      // =========================================================== Newton solver (mj_solNewton), warm-started from qacc
      float Jaref[RPL], Jv[RPL], Ma = 0.f;
      float dinv = 1.f;
      auto mulM = [&](const float* v) -> float {  // (M v)[lane]
        float s = 0.f;
        if (ln < NV) {
          const float* Mr = S.M[ln];
#pragma unroll
          for (int k = 0; k < NV; k++) s += Mr[k] * v[k];
        }
        return s;
      };
      auto rowdot = [&](const float* v, int rr) -> float {  // J[row] . v for dense row (ln + LW rr)
        float s = 0.f;
        if (rdense[rr]) {
          const float* Jr = S.J[ln + LW * rr];
#pragma unroll
          for (int d = 0; d < NV; d++) s += Jr[d] * v[d];
        }
        return s;
      };
#pragma unroll
      for (int rr = 0; rr < RPL; rr++) { Jaref[rr] = rdense[rr] ? rowdot(S.qacc, rr) - raref[rr] : 0.f; Jv[rr] = 0.f; }
      Ma = mulM(S.qacc);
      float qacc_l = ln < NV ? S.qacc[ln] : 0.f;
      const float qsm_l = ln < NV ? S.qsm[ln] : 0.f;
      float cost = 0.f, gauss = 0.f, grad_l = 0.f, gradnorm2 = 0.f;
      const float scale = 1.f / (meaninertia * (float)(NV > 1 ? NV : 1));
      int niter = 0;
      const int maxiter = min(dm.iterations, A.max_newton);

      float dact_cur[RPL], dact_fac[RPL];  // this row's active D now / when H was last factorised
#pragma unroll
      for (int rr = 0; rr < RPL; rr++) { dact_cur[rr] = 0.f; dact_fac[rr] = -1.f; }
      int cact_cur[CPL], cact_fac[CPL];    // CT: active edges (4 bits) of this lane's ground contacts now / at the last factorisation
#pragma unroll
      for (int cc = 0; cc < CPL; cc++) { cact_cur[cc] = 0; cact_fac[cc] = -1; }
      int uact_cur = 0, uact_fac = -1;     // dof rows of this lane inside their quadratic zone (bit 0 friction loss, bit 1 limit) now / at the last factorisation
      float dofD_l = 0.f;                  // their active D: this dof's diagonal term of J^T D J
      unsigned cbmask = 0u;
      if constexpr (CT) cbmask = S.cbmask;
      auto update_constraint = [&]() {
        // mj_constraintUpdate: force, active set, cost
        float csum = 0.f;
        if constexpr (CT) {
          if (ln < nbody) {
#pragma unroll
            for (int q = 0; q < 6; q++) S.bw[ln][q] = 0.f;
          }
        }
#pragma unroll
        for (int rr = 0; rr < RPL; rr++) {
          // branch-free (lanes past the last row carry D = 0): quadratic for x < rhi, zero beyond
          const float x = Jaref[rr];
          const bool inq = x < rhi[rr];
          const float dact = inq ? rD[rr] : 0.f;
          dact_cur[rr] = dact;
          csum += 0.5f * dact * x * x;
          S.w.r.rowf[ln + LW * rr] = -dact * x;
          S.w.r.rowD[ln + LW * rr] = dact;
        }
        // dof rows: friction loss (Huber: quadratic for |x| < R f, linear with slope -+f beyond) and limit (one-sided quadratic);
        // x = J a - aref straight from this lane's qacc
        float qcu;   // their force on this dof
        {
          const float xf = qacc_l - faref;
          const bool inf_ = fabsf(xf) < fRf;
          const float sf = xf <= -fRf ? fls : -fls;
          csum += inf_ ? 0.5f * fD * xf * xf : -0.5f * fRf * fls - sf * xf;
          const float xl = lsg * qacc_l - laref;
          const bool inl = xl < 0.f && lD > 0.f;
          csum += inl ? 0.5f * lD * xl * xl : 0.f;
          qcu = (inf_ ? -fD * xf : sf) + (inl ? -lsg * lD * xl : 0.f);
          dofD_l = (inf_ ? fD : 0.f) + (inl ? lD : 0.f);
          uact_cur = (inf_ ? 1 : 0) | (inl ? 2 : 0);
        }
        if constexpr (CT) {
          WSYNC();   // the bodies' wrenches are zero before the contacts add to them
          // ground contacts: one-sided quadratic rows (active while J a - aref < 0); their forces as one wrench per body
#pragma unroll
          for (int cc = 0; cc < CPL; cc++) {
            const int c = ln + LW * cc;
            int bits = 0;
            if (c < ncon) {
              const float D = S.cD[c];
              const float jb = S.cJar[c][0], j1 = S.cJar[c][1], j2 = S.cJar[c][2];
              const float xr[4] = {jb + j1, jb - j1, jb + j2, jb - j2};
              float f[4];
#pragma unroll
              for (int e = 0; e < 4; e++) {
                const float x = xr[e];
                const bool on = x < 0.f;
                f[e] = on ? -D * x : 0.f;
                csum += on ? 0.5f * D * x * x : 0.f;
                bits |= on ? (1 << e) : 0;
              }
              if (bits) {
                CGeo G;
                cgeo(c, G);
                const float fn = f[0] + f[1] + f[2] + f[3], f1 = G.mu * (f[0] - f[1]), f2 = G.mu * (f[2] - f[3]);
                float F[3], T[3];
                for (int k = 0; k < 3; k++) F[k] = G.n[k] * fn + G.t1[k] * f1 + G.t2[k] * f2;
                cross(T, G.off, F);
                for (int k = 0; k < 3; k++) { atomicAdd(&S.bw[G.b][k], T[k]); atomicAdd(&S.bw[G.b][3 + k], F[k]); }
              }
            }
            cact_cur[cc] = bits;
          }
        }
        WSYNC();
        float qc = 0.f;
        if (ln < NV) {
          qc = qcu;
          for (int r = 0; r < ngen; r++) qc += S.J[r][ln] * S.w.r.rowf[r];
          if constexpr (CT) {
            // J^T f of the ground contacts: cdof . (sum of the wrenches on the bodies this dof moves)
            float W[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (unsigned mk = dm.rec[dm.rec[ln].d_body].b_subtree & cbmask; mk; mk &= mk - 1) {
              const int b = __builtin_ctz(mk);
#pragma unroll
              for (int q = 0; q < 6; q++) W[q] += S.bw[b][q];
            }
#pragma unroll
            for (int q = 0; q < 6; q++) qc += S.cdof[ln][q] * W[q];
          }
          S.qcon[ln] = qc;
        }
        grad_l = ln < NV ? Ma - qsm_l - qc : 0.f;
        {   // Gauss term, constraint cost and |grad|^2 in one interleaved reduction
          float sc_, sg_;
          grp_sum3<LW>(ln < NV ? (0.5f * Ma - qsm_l) * qacc_l : 0.f, csum, grad_l * grad_l, gauss, sc_, sg_);
          cost = sc_ + gauss;
          gradnorm2 = sg_;
        }
      };

      auto update_search = [&](bool act) {
        // H = M + J^T diag(D_active) J changes only when the active set does (mj_solNewton rebuilds on state changes):
        // otherwise the factor left in LDS by the previous iteration is reused.  With two environments per wave the
        // rebuild runs for both whenever one of them needs it (the matrix-pipe tile spans the wave; for the other
        // environment it reproduces the factor it already has).
        bool changed = false;
#pragma unroll
        for (int rr = 0; rr < RPL; rr++) changed = changed || (dact_cur[rr] != dact_fac[rr]);
#pragma unroll
        for (int cc = 0; cc < CPL; cc++) changed = changed || (CT && cact_cur[cc] != cact_fac[cc]);
        changed = changed || uact_cur != uact_fac;
        changed = changed && act;
        unsigned long long q0_ = 0;
        if (PROF) { __builtin_amdgcn_s_waitcnt(0); q0_ = __builtin_amdgcn_s_memtime(); }
        if (grp_ballot<LW>(changed, hb) != 0ull) st_build++;
        if (__ballot(changed) != 0ull) {
#pragma unroll
          for (int rr = 0; rr < RPL; rr++) dact_fac[rr] = dact_cur[rr];
#pragma unroll
          for (int cc = 0; cc < CPL; cc++) cact_fac[cc] = cact_cur[cc];
          uact_fac = uact_cur;
          static_assert(NV <= 32, "one 32x32 MFMA tile");
          if constexpr (EPW == 1) {
             // H = M + (D J)^T J on the matrix pipe: v_mfma_f32_32x32x2_f32, two constraint rows per instruction
             // (exact fp32 FMA chain; inactive rows carry D = 0).  Lane l feeds A[i = l & 31][k = l >> 5] = D_k J[k][i]
             // and B[k][j = l & 31] = J[k][j]; C/D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5).
            typedef float f32x16 __attribute__((ext_vector_type(16)));
            const int col = ln & 31, half = ln >> 5;
            f32x16 acc;
#pragma unroll
            for (int v = 0; v < 16; v++) {
              const int row = (v & 3) + 8 * (v >> 2) + 4 * half;
              acc[v] = (row < NV && col < NV) ? S.M[row < NV ? row : 0][col < NV ? col : 0] : 0.f;
            }
            const int colc = col < NV ? col : 0;   // lanes past NV feed tile rows / columns that are never stored
            for (int r0 = 0; r0 < ngen; r0 += 2) {
              const int r = r0 + half;
              const float jv = r < ngen ? S.J[r < ngen ? r : 0][colc] : 0.f;
              const float dv = S.w.r.rowD[r < ngen ? r : 0];
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(jv * dv, jv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < 16; v++) {
              const int row = (v & 3) + 8 * (v >> 2) + 4 * half;
              if (row < NV && col < NV) S.u.H[row][col] = acc[v];
            }
          } else {
            // two environments: v_mfma_f32_32x32x1_2b_f32, block b = environment b.  Lane l feeds block l >> 5 with
            // A[i = l & 31] = D_r J[r][i] and B[j = l & 31] = J[r][j] of its own environment, one constraint row per
            // instruction; C/D registers 16 b .. 16 b + 15 hold block b: col = l & 31, row = (v & 3) + 8 (v >> 2) + 4 (l >> 5)
            // -- every lane carries rows of both environments' tiles, so it reads M and writes H of both.
            typedef float f32x32 __attribute__((ext_vector_type(32)));
            const int col = wlane & 31, rsel = wlane >> 5;
            f32x32 acc;
#pragma unroll
            for (int v = 0; v < 32; v++) {
              const int row = (v & 3) + 8 * ((v & 15) >> 2) + 4 * rsel;
              acc[v] = (row < NV && col < NV) ? SS[v >> 4].M[row < NV ? row : 0][col < NV ? col : 0] : 0.f;
            }
            const int ngen_max = max(__builtin_amdgcn_readlane(ngen, 0), __builtin_amdgcn_readlane(ngen, 32));
            for (int r = 0; r < ngen_max; r++) {
              const bool ok = r < ngen && col < NV;
              const float jv = ok ? S.J[ok ? r : 0][ok ? col : 0] : 0.f;
              const float dv = ok ? S.w.r.rowD[ok ? r : 0] : 0.f;
              acc = __builtin_amdgcn_mfma_f32_32x32x1f32(jv * dv, jv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < 32; v++) {
              const int row = (v & 3) + 8 * ((v & 15) >> 2) + 4 * rsel;
              if (row < NV && col < NV) SS[v >> 4].u.H[row][col] = acc[v];
            }
          }
          WSYNC();
          if (PROF) { __builtin_amdgcn_s_waitcnt(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[10] += t_ - q0_; q0_ = t_; }   // Hessian build
          float a_row[NV];
          {
            // dof rows (frictionloss, limits) only touch the diagonal: added in LDS (one dynamic access) so that the row
            // comes out as plain reads; lanes past NV factor a copy of row 0, which nobody reads
            if (ln < NV) S.u.H[ln][ln] += dofD_l;
            unsigned long long th0_ = 0;
            if constexpr (CT) {
              // J^T D J of the active ground-contact rows: per body the 6 x 6 matrix sum_e D w_e w_e^T (lower triangle), then through
              // the tree like a composite inertia: H[i][j] += cdof_i^T (sum over the subtree of i) cdof_j for j an ancestor of i
              if (PROF) { __builtin_amdgcn_s_waitcnt(0); th0_ = __builtin_amdgcn_s_memtime(); }
              for (int i = ln; i < nbody * 24; i += LW) (&S.bW[0][0])[i] = 0.f;
              WSYNC();
#pragma unroll
              for (int cc = 0; cc < CPL; cc++) {
                const int c = ln + LW * cc, bits = cact_cur[cc];
                if (c < ncon && bits) {
                  CGeo G;
                  cgeo(c, G);
                  const float D = S.cD[c];
                  float acc[21];
#pragma unroll
                  for (int i = 0; i < 21; i++) acc[i] = 0.f;
#pragma unroll
                  for (int e = 0; e < 4; e++) {
                    const float De = ((bits >> e) & 1) ? D : 0.f, sg = (e & 1) ? -G.mu : G.mu;
                    const float* tk = (e >> 1) ? G.t2 : G.t1;
                    float w[6];
                    for (int k = 0; k < 3; k++) w[3 + k] = G.n[k] + sg * tk[k];
                    cross(w, G.off, w + 3);
#pragma unroll
                    for (int i = 0, x = 0; i < 6; i++)
#pragma unroll
                      for (int j = 0; j <= i; j++, x++) acc[x] += De * w[i] * w[j];
                  }
#pragma unroll
                  for (int i = 0; i < 21; i++) atomicAdd(&S.bW[G.b][i], acc[i]);
                }
              }
              WSYNC();
              if (PROF && !HF) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pext[1] += t_ - th0_; th0_ = t_; }   // flat kernels: [17] = contact-matrix accumulation
            }
            const float* Hr = S.u.H[ln < NV ? ln : 0];
#pragma unroll
            for (int k = 0; k < NV; k++) a_row[k] = Hr[k];
            if constexpr (CT) {
              // tree pass, straight into the row this lane factorises (only the lower triangle of H is read by the factorisation, and row
              // i's entries j <= i are exactly i's ancestors): y = (sum of the subtree's matrices) cdof_i, then a_row[j] += cdof_j . y
              if (ln < NV) {
                const auto& R = dm.rec[ln];
                const unsigned sub = dm.rec[R.d_body].b_subtree & cbmask;
                if (sub) {
                  float cdi[6], y[6];
#pragma unroll
                  for (int q = 0; q < 6; q++) { cdi[q] = S.cdof[ln][q]; y[q] = 0.f; }
                  for (unsigned mk = sub; mk; mk &= mk - 1) {
                    const float4* Wb4 = reinterpret_cast<const float4*>(S.bW[__builtin_ctz(mk)]);
                    float Wb[24];
#pragma unroll
                    for (int q = 0; q < 6; q++) { const float4 w4 = Wb4[q]; Wb[4 * q] = w4.x; Wb[4 * q + 1] = w4.y; Wb[4 * q + 2] = w4.z; Wb[4 * q + 3] = w4.w; }
#pragma unroll
                    for (int i = 0; i < 6; i++)
#pragma unroll
                      for (int j = 0; j <= i; j++) {
                        const float wv = Wb[i * (i + 1) / 2 + j];
                        y[i] += wv * cdi[j];
                        if (j != i) y[j] += wv * cdi[i];
                      }
                  }
                  const unsigned anc = R.d_ancmask;
#pragma unroll
                  for (int k = 0; k < NV; k++) {
                    if ((anc >> k) & 1u) {
                      const float2* ck = reinterpret_cast<const float2*>(S.cdof[k]);
                      const float2 c0 = ck[0], c1 = ck[1], c2 = ck[2];
                      a_row[k] += c0.x * y[0] + c0.y * y[1] + c1.x * y[2] + c1.y * y[3] + c2.x * y[4] + c2.y * y[5];
                    }
                  }
                }
              }
              if (PROF && !HF) { __builtin_amdgcn_s_waitcnt(0); pext[2] += __builtin_amdgcn_s_memtime() - th0_; }   // [18] = tree pass
            }
          }
          chol_lower<NV, LW>(a_row, dinv, ln, hb);
          WSYNC();
          chol_park<NV, L::LD>(S.u.H, a_row, dinv, ln);
          WSYNC();
        }
        if (PROF) { __builtin_amdgcn_s_waitcnt(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[11] += t_ - q0_; q0_ = t_; }   // Cholesky + park
        const float mg = chol_solve_lds<NV, L::LD, LW>(S.u.H, dinv, grad_l, ln, hb);
        if (ln < NV) S.sr[ln] = -mg;
        WSYNC();
        if (PROF) { __builtin_amdgcn_s_waitcnt(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[12] += t_ - q0_; }   // triangular solves
      };

      update_constraint();
      float gradnorm = sqrtf(gradnorm2);
      if (kmode == MODE_DEBUG && A.dbg != nullptr) {
        // dump position/velocity-stage intermediates before the solve
        float* D = A.dbg;
        if (ln == 0) { D[0] = (float)(ncon + (CT ? npc : 0)); D[1] = (float)nefc; D[2] = (float)ne; D[3] = (float)nf; D[4] = (float)nl; D[5] = cost; D[6] = gradnorm; D[7] = (float)ngen; }
        if (ln < nbody) { for (int k = 0; k < 3; k++) D[64 + ln * 3 + k] = S.xpos[ln][k]; for (int k = 0; k < 4; k++) D[192 + ln * 4 + k] = S.xquat[ln][k]; }
        for (int e = ln; e < TRI; e += LW) D[512 + e] = S.M[dm.tri_row[e]][dm.tri_col[e]];
        if (ln < NV) { D[1100 + ln] = S.qsm[ln]; for (int q = 0; q < 6; q++) D[1200 + ln * 6 + q] = S.cdof[ln][q]; }
        D[1400 + ln] = rdense[0] ? (ln < ne ? (float)RT_EQ : (float)RT_CONTACT) : (float)RT_NONE; D[1464 + ln] = rD[0]; D[1528 + ln] = raref[0]; D[1592 + ln] = rpos_dbg; D[1656 + ln] = Jaref[0];
        if (ln < ngen) for (int d = 0; d < NV; d++) D[2048 + ln * NV + d] = S.J[ln][d];
        if (ln < MC && ln < 16) { D[1720 + ln] = ln < ncon ? S.cdist[ln] : 0.f; for (int k = 0; k < 3; k++) D[1740 + ln * 3 + k] = ln < ncon ? S.cpos[ln][k] : 0.f;
                       D[1900 + ln] = ln < ncon ? (float)S.cgeom[ln] : -1.f;
                       for (int k = 0; k < 3; k++) D[1920 + ln * 3 + k] = (NRM && ln < ncon) ? S.cnrm[NRM ? ln : 0][k] : (k == 2 ? 1.f : 0.f); }
        // every contact (ground contacts, then -- CT -- the robot-robot ones): 8 floats each from D[4096]: dist, pos, normal, geom code
        if (ln == 0) { D[11] = (float)ncon; D[12] = (float)npc; }
        for (int c = ln; c < ncon + (CT ? npc : 0) && c < 512; c += LW) {
          const bool gnd = !CT || c < ncon;
          const int cp = CT ? c - ncon : 0;
          float* o = D + 4096 + 8 * c;
          o[0] = con_dist<CT>(S, gnd, c, cp);
          for (int k = 0; k < 3; k++) {
            o[1 + k] = con_pos<CT>(S, gnd, c, cp)[k];
            o[4 + k] = (gnd ? NRM : true) ? con_nrm<CT, NRM>(S, gnd, c, cp)[k] : (k == 2 ? 1.f : 0.f);
          }
          o[7] = (float)con_geom<CT>(S, gnd, c, cp);
        }
      }
      // one Newton iteration after the search direction is known: exact line search, move, constraint update; false = stop
      auto newton_iterate = [&]() -> bool {
        unsigned long long q1_ = 0;
        if (PROF) { __builtin_amdgcn_s_waitcnt(0); q1_ = __builtin_amdgcn_s_memtime(); }
        // ---- exact line search on the piecewise-quadratic cost (PrimalSearch)
        const float sr_l = ln < NV ? S.sr[ln] : 0.f;
        const float Mv = mulM(S.sr);
        float q0[RPL], q1[RPL], q2[RPL];
#pragma unroll
        for (int rr = 0; rr < RPL; rr++) {
          Jv[rr] = rowdot(S.sr, rr);
          q0[rr] = 0.5f * rD[rr] * Jaref[rr] * Jaref[rr]; q1[rr] = rD[rr] * Jaref[rr] * Jv[rr]; q2[rr] = 0.5f * rD[rr] * Jv[rr] * Jv[rr];
        }
        const float xf0 = qacc_l - faref, xl0 = lsg * qacc_l - laref, vl = lsg * sr_l;   // dof rows: J a - aref now, J s = +-s_dof
        if constexpr (CT) {
          // J s of the ground contacts from the bodies' twists of the search direction
          body_twists(S.sr, nullptr, S.tw, nullptr);
#pragma unroll
          for (int cc = 0; cc < CPL; cc++) {
            const int c = ln + LW * cc;
            if (c < ncon) {
              CGeo G;
              cgeo(c, G);
              float sn, s1, s2;
              point_proj(S.tw[G.b], G, sn, s1, s2);
              S.cJv[c][0] = sn; S.cJv[c][1] = G.mu * s1; S.cJv[c][2] = G.mu * s2;
            }
          }
        }
        float sn2_, qG1, qG2;
        grp_sum3<LW>(sr_l * sr_l, sr_l * (Ma - qsm_l), 0.5f * sr_l * Mv, sn2_, qG1, qG2);
        const float snorm = sqrtf(sn2_);
        if (!(snorm >= 1e-20f)) return false;
        const float gtol = A.tol32 * A.ls_scale * dm.ls_tolerance * snorm / scale;
        struct Pnt { float alpha, cost, d0, d1, step; };   // step = -d0 / d1 (Newton step of the 1-D search; v_rcp_f32: 1 ulp is ample)
        auto eval = [&](float alpha) -> Pnt {
          float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
          for (int rr = 0; rr < RPL; rr++) {
            const bool inq = Jaref[rr] + alpha * Jv[rr] < rhi[rr];
            c0 += inq ? q0[rr] : 0.f;
            c1 += inq ? q1[rr] : 0.f;
            c2 += inq ? q2[rr] : 0.f;
          }
          {
            const float xf = xf0 + alpha * sr_l;
            const bool inf_ = fabsf(xf) < fRf;
            const float sf = xf <= -fRf ? fls : -fls;
            c0 += inf_ ? 0.5f * fD * xf0 * xf0 : -0.5f * fRf * fls - sf * xf0;
            c1 += inf_ ? fD * xf0 * sr_l : -sf * sr_l;
            c2 += inf_ ? 0.5f * fD * sr_l * sr_l : 0.f;
            const float dl = (xl0 + alpha * vl < 0.f) ? lD : 0.f;
            c0 += 0.5f * dl * xl0 * xl0;
            c1 += dl * xl0 * vl;
            c2 += 0.5f * dl * vl * vl;
          }
          if constexpr (CT) {
#pragma unroll
            for (int cc = 0; cc < CPL; cc++) {
              const int c = ln + LW * cc;
              if (c < ncon) {
                const float D = S.cD[c];
                const float jb = S.cJar[c][0], j1 = S.cJar[c][1], j2 = S.cJar[c][2], vb = S.cJv[c][0], v1 = S.cJv[c][1], v2 = S.cJv[c][2];
                const float ja4[4] = {jb + j1, jb - j1, jb + j2, jb - j2}, jv4[4] = {vb + v1, vb - v1, vb + v2, vb - v2};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                  const float ja = ja4[e], jv = jv4[e];
                  const bool inq = ja + alpha * jv < 0.f;
                  c0 += inq ? 0.5f * D * ja * ja : 0.f;
                  c1 += inq ? D * ja * jv : 0.f;
                  c2 += inq ? 0.5f * D * jv * jv : 0.f;
                }
              }
            }
          }
          float C0, C1, C2;
          grp_sum3<LW>(c0, c1, c2, C0, C1, C2);
          C0 += gauss; C1 += qG1; C2 += qG2;
          Pnt p;
          p.alpha = alpha;
          p.cost = alpha * alpha * C2 + alpha * C1 + C0;
          p.d0 = 2.f * alpha * C2 + C1;
          p.d1 = 2.f * C2;
          if (!(p.d1 > 0.f)) p.d1 = 1e-15f;
          p.step = -p.d0 * __builtin_amdgcn_rcpf(p.d1);
          return p;
        };
        float alpha = 0.f;
        int lsit_total = 0;
        {
          int lsit = 0;
          const int maxls = min(dm.ls_iterations, A.max_ls);
          // PrimalSearch starts with an evaluation at alpha = 0; for the Newton direction its value and derivatives are
          // known in closed form: phi(0) = cost, phi'(0) = g.s, phi''(0) = s'Hs = -g.s (H s = -g on the current active set)
          Pnt p0;
          {
            const float gs = grp_sum<LW>(grad_l * sr_l);
            p0.alpha = 0.f; p0.cost = cost; p0.d0 = gs; p0.d1 = -gs;
            if (!(p0.d1 > 0.f)) p0.d1 = 1e-15f;
            p0.step = -p0.d0 * __builtin_amdgcn_rcpf(p0.d1);
          }
          Pnt p1 = eval(p0.alpha + p0.step); lsit++;
          if (p0.cost < p1.cost) p1 = p0;
          bool done = false;
          if (fabsf(p1.d0) < gtol) { alpha = p1.alpha; done = true; }
          if (!done) {
            const float dir = p1.d0 < 0.f ? 1.f : -1.f;
            Pnt p2 = p1;
            bool p2update = false;
#pragma nounroll
            while (p1.d0 * dir <= -gtol && lsit < maxls) {
              p2 = p1; p2update = true;
              p1 = eval(p1.alpha + p1.step); lsit++;
              if (fabsf(p1.d0) < gtol) { alpha = p1.alpha; done = true; break; }
            }
            if (!done) {
              if (lsit >= maxls || !p2update) { alpha = p1.alpha; done = true; }
            }
            if (!done) {
              Pnt p2next = p1;
              Pnt p1next = eval(p1.alpha + p1.step); lsit++;
#pragma nounroll
              while (lsit < maxls) {
                Pnt pmid = eval(0.5f * (p1.alpha + p2.alpha)); lsit++;
                // candidates: p1next, p2next, pmid (kept as named values: no dynamically indexed array -> no scratch)
                {
                  bool found = false;
                  float bestcost = 0.f, besta = 0.f;
                  if (fabsf(p1next.d0) < gtol) { found = true; bestcost = p1next.cost; besta = p1next.alpha; }
                  if (fabsf(p2next.d0) < gtol && (!found || p2next.cost < bestcost)) { found = true; bestcost = p2next.cost; besta = p2next.alpha; }
                  if (fabsf(pmid.d0) < gtol && (!found || pmid.cost < bestcost)) { found = true; bestcost = pmid.cost; besta = pmid.alpha; }
                  if (found) { alpha = besta; done = true; break; }
                }
                int b1 = 0, b2 = 0;
                const Pnt c0_ = p1next, c1_ = p2next, c2_ = pmid;
#define BRACKET_UPDATE(P, C, FLAG)                                                                   \
  if (P.d0 < 0.f && C.d0 < 0.f && P.d0 < C.d0) { P = C; FLAG = 1; }                                    \
  else if (P.d0 > 0.f && C.d0 > 0.f && P.d0 > C.d0) { P = C; FLAG = 2; }
                BRACKET_UPDATE(p1, c0_, b1) BRACKET_UPDATE(p1, c1_, b1) BRACKET_UPDATE(p1, c2_, b1)
                if (b1) { p1next = eval(p1.alpha + p1.step); lsit++; }
                BRACKET_UPDATE(p2, c0_, b2) BRACKET_UPDATE(p2, c1_, b2) BRACKET_UPDATE(p2, c2_, b2)
                if (b2) { p2next = eval(p2.alpha + p2.step); lsit++; }
#undef BRACKET_UPDATE
                if (!b1 && !b2) { alpha = pmid.alpha; done = true; break; }
              }
              if (!done) {
                if (p1.cost <= p2.cost && p1.cost < p0.cost) alpha = p1.alpha;
                else if (p2.cost <= p1.cost && p2.cost < p0.cost) alpha = p2.alpha;
                else alpha = 0.f;
              }
            }
          }
          lsit_total = lsit;
        }
        if constexpr (EPW == 1) alpha = rfl(alpha);
        if constexpr (EPW == 2) alpha = grp_bcast<LW>(alpha, 0, hb);
        if (PROF) { __builtin_amdgcn_s_waitcnt(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[13] += t_ - q1_; q1_ = t_; }   // line search
        st_ls += lsit_total;
        if (alpha == 0.f) return false;
        // ---- move
        qacc_l += alpha * sr_l;
        Ma += alpha * Mv;
#pragma unroll
        for (int rr = 0; rr < RPL; rr++) Jaref[rr] += alpha * Jv[rr];
        if constexpr (CT) {
#pragma unroll
          for (int cc = 0; cc < CPL; cc++) {
            const int c = ln + LW * cc;
            if (c < ncon) {
#pragma unroll
              for (int e = 0; e < 3; e++) S.cJar[c][e] += alpha * S.cJv[c][e];
            }
          }
        }
        if (ln < NV) S.qacc[ln] = qacc_l;
        const float oldcost = cost;
        update_constraint();
        gradnorm = sqrtf(gradnorm2);
        niter++;
        if (PROF) { __builtin_amdgcn_s_waitcnt(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[14] += t_ - q1_; }   // move + constraint update
        const float improvement = scale * (oldcost - cost);
        return !(improvement < A.tol32);
      };
      // s_setprio by solver lag, refreshed after every Newton iteration (4-waves-per-SIMD kernels; with 2 waves it starves
      // one: measured).  Waves that have needed more iterations than the pack are the ones the launch ends with: letting them
      // issue first trims that tail, and it staggers the four waves of a SIMD so that they are not all inside the same
      // latency-bound phase at once.  lag = iterations so far - usual count + the wave's slot number on its SIMD
      // (HW_REG_HW_ID.WAVE_ID & 3, a tie-break).  Thresholds (cosim_set_param "wave_priority") swept on the 4096-env
      // flamingo_light_v1 bench: off 10.5 M, (3; 1, 3, 6) 10.9 M, (6; -4, -2, 0) 11.3 M, + tie-break 11.5 M, + per-iteration
      // refresh 11.7 M env-steps/s.  Round 2, four range launches: off 13.92 M, (6; -4, -2, 0) 13.88 M, (3; 0, 2, 4) 14.07 M (the default now;
      // every setting with base 2 ... 4 is within 0.3 % of it: with the spills and most exposed waits gone the lever is small).
      auto wave_priority = [&](int iters_so_far) {
        if constexpr (EPW == 1 && ((RPL == 1 && !HF) || NV < 18) && !PROF) {
          const int lag = iters_so_far - A.prio[0] * (sub + 1) + (int)(__builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 3u);
          if (lag >= A.prio[3]) __builtin_amdgcn_s_setprio(3);
          else if (lag >= A.prio[2]) __builtin_amdgcn_s_setprio(2);
          else if (lag >= A.prio[1]) __builtin_amdgcn_s_setprio(1);
          else __builtin_amdgcn_s_setprio(0);
        }
      };
      // Environments of one wave iterate together: `act` marks the ones still running, the wave leaves when none is.
      bool act = true;
#pragma nounroll
      while (true) {
        KARGS_FENCE();
        act = act && niter < maxiter && !(scale * gradnorm < A.tol32);
        const bool any = __ballot(act) != 0ull;
        if (!any) break;
        if constexpr (EPW == 1) act = any;   // one environment: wave-uniform, keeps the body a scalar branch
        update_search(act);
        if (act) act = newton_iterate();
        wave_priority(st_newton + niter);
      }
      st_newton += niter;
      st_rows += nefc;
      if (kmode == MODE_DEBUG && A.dbg != nullptr) {
        float* D = A.dbg;
        if (ln == 0) { D[8] = (float)niter; D[9] = cost; D[10] = gradnorm; }
        if (ln < NV) { D[1000 + ln] = qacc_l; D[1040 + ln] = S.qcon[ln]; }
        D[1800 + ln] = S.w.r.rowf[ln];
        if (ln < 10) D[16 + ln] = S.sens[ln];
      }

      STAMP(7);   // Newton (sub-phases 10..14 inside)
      // =========================================================== _is_done of flamingo_p_v3 (flamingo_p_v3.py:225-233)
      // cfrc_ext of mj_rnePostConstraint for the listed bodies: sum of contact wrenches [torque; force], world aligned,
      // about the tree's CoM; "any signed component > 1.0" terminates.  Uses the last substep's contacts and forces.
      if (dm.term_mode == 1 && sub == nsub - 1 && sub_last) {
        bool hit = false;
        if (ln > 0 && ln < nbody && ((dm.term_bodymask >> ln) & 1u)) {
          float wr[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          for (int c = 0; c < ncon + (CT ? npc : 0); c++) {
            // CT: ground contacts (forces from their rows' J a - aref) first, then the robot-robot contacts (dense rows)
            const bool gnd = !CT || c < ncon;
            const int cp = CT ? c - ncon : 0;
            const int gg = con_geom<CT>(S, gnd, c, cp), g = gg & 0xff, g1 = (gg >> 8) - 1;
            const bool on2 = dm.rec[g].g_body == ln, on1 = SC && g1 >= 0 && dm.rec[g1 >= 0 ? g1 : 0].g_body == ln;
            if (!on2 && !on1) continue;
            const float mu = (SC && g1 >= 0) ? fmaxf(MINMU, fmaxf(A.gext[g].w, A.gext[g1].w)) : S.p_gmu[g];
            float f[4];
            for (int e = 0; e < 4; e++) f[e] = S.w.r.rowf[ne + 4 * (CT ? (gnd ? 0 : cp) : c) + e];
            if constexpr (CT) {
              if (gnd)
                for (int e = 0; e < 4; e++) { const float x = S.cJar[c][0] + ((e & 1) ? -1.f : 1.f) * S.cJar[c][1 + (e >> 1)]; f[e] = x < 0.f ? -S.cD[c] * x : 0.f; }
            }
            const float* cn_ = con_nrm<CT, NRM>(S, gnd, c, cp);
            const float* cp_ = con_pos<CT>(S, gnd, c, cp);
            const bool hasn = gnd ? NRM : true;
            float nrm[3] = {hasn ? cn_[0] : 0.f, hasn ? cn_[1] : 0.f, hasn ? cn_[2] : 1.f}, t1[3], t2[3];
            make_frame(nrm, t1, t2);
            const float fl0 = f[0] + f[1] + f[2] + f[3], fl1 = (f[0] - f[1]) * mu, fl2 = (f[2] - f[3]) * mu;  // mj_contactForce, pyramidal
            float fw[3], dif[3], tq[3];
            for (int k = 0; k < 3; k++) { fw[k] = nrm[k] * fl0 + t1[k] * fl1 + t2[k] * fl2; dif[k] = cp_[k] - S.com[k]; }
            cross(tq, dif, fw);
            const float sgn = on2 ? 1.f : -1.f;   // equal and opposite on geom1's body
            for (int k = 0; k < 3; k++) { wr[k] += sgn * tq[k]; wr[3 + k] += sgn * fw[k]; }
          }
          for (int k = 0; k < 6; k++) hit = hit || (wr[k] > 1.0f);
        }
        if (grp_ballot<LW>(hit, hb) != 0ull) terminated = 1;
      }

      // =========================================================== mj_implicit (implicitfast) + mj_advance
      {
        float a_row[NV];
        {
          // M + h diag(damping): the diagonal term goes into LDS (M is rebuilt next substep), the row is plain reads
          if (ln < NV) S.M[ln][ln] += h * dm.rec[ln].d_damping;
          WSYNC();
          const float* Mr = S.M[ln < NV ? ln : 0];
#pragma unroll
          for (int k = 0; k < NV; k++) a_row[k] = Mr[k];
        }
        chol_lower<NV, LW>(a_row, dinv, ln, hb);
        WSYNC();
        chol_park<NV, L::LD>(S.u.H, a_row, dinv, ln);
        WSYNC();
        const float rhs = ln < NV ? qsm_l + S.qcon[ln] : 0.f;
        const float qa = chol_solve_lds<NV, L::LD, LW>(S.u.H, dinv, rhs, ln, hb);
        WSYNC();
        if (PROF) { __builtin_amdgcn_s_waitcnt(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[15] += t_ - pt0; }   // implicit solve done (advance follows)
        if (kmode != MODE_DEBUG) {
          if (ln < NV) S.qvel[ln] = qv + h * qa;
          WSYNC();
          if (ln > 0 && ln < nbody) {
            const auto& R = dm.rec[ln];
            const int jt = R.b_jtype;
            if (jt == CS_JNT_HINGE) S.qpos[R.b_qadr] += h * S.qvel[R.b_dadr];
            else if (jt == CS_JNT_FREE) {
              const int qa0 = R.b_qadr, da = R.b_dadr;
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
          if (ln < NV) A.dbg[1080 + ln] = qa;
        }
      }
      STAMP(8);   // implicitfast + advance
    }  // substeps
    if (kmode == MODE_DEBUG) return;
    if constexpr (KM == 2) {
      if (kmode == MODE_STEP && !sub_last) {
        // split pipeline, not the control step's last substep: the physics state goes back to the record (the next narrowphase launch
        // reads the new pose), the control step's own values wait in xstate, the counters accumulate; no epilogue yet
        if (lane < nq) rec[lay.s_qpos + lane] = S.qpos[lane];
        if (lane < NV) { rec[lay.s_qvel + lane] = S.qvel[lane]; rec[lay.s_warm + lane] = S.qacc[lane]; }
        if (sub_first) {
          float* xs = A.xstate + (size_t)env * XS;
          if (lane < NV) xs[lane] = S.qact[lane];
          if (lane < nu) { xs[NV + lane] = S.act[lane]; xs[NV + L::NUMAX + lane] = S.tq[lane]; }
          if (lane < CS_MAXCMD) xs[NV + 2 * L::NUMAX + lane] = S.cmd[lane];
        }
        if (lane == 0) {
          meta[0] = sim_step; meta[2] = has_prev;
          meta[5] += st_newton; meta[6] += st_ls; meta[7] += st_build; meta[3] += st_rows;
          meta[8] += st_dropcon + st_walkcut; meta[9] += st_droplim; meta[10] = max(meta[10], st_maxcon); meta[13] += st_walkcut;
        }
        return;
      }
    }

    // ---- mj_checkPos/Vel: non-finite state -> this env is reset (MuJoCo resets the data and warns)
    {
      bool nf_ = false;
      if (lane < nq) nf_ = !(fabsf(S.qpos[lane]) < 1e10f);
      if (lane < NV) nf_ = nf_ || !(fabsf(S.qvel[lane]) < 1e10f) || !(fabsf(S.qacc[lane]) < 1e10f);
      bad = grp_ballot<LW>(nf_, hb) != 0ull;
    }
    if (sim_step == ob.max_sim_step) truncated = 1;
    if (bad) terminated = 1;
    do_reset = bad || ((terminated || truncated) && ob.auto_reset);
  }

  // =============================================================== info / flags (_get_info, flamingo_light_v1.py:166-183)
  // Written from the state the step ended in, BEFORE an auto-reset touches it: on the step that ends an episode the reference's
  // info describes that last step (the returned state vector, as in gym-style auto-reset, is the first one of the next episode).
  if (kmode == MODE_STEP) {
    WSYNC();
    if (A.info != nullptr) {
      float* inf = A.info + ((size_t)kstep * A.n_envs + env) * ob.info_dim;
      const float raw_action = lane < nu ? S.act[lane] : 0.f;
      const float prev_action = lane < nu ? rec[lay.s_lastact + lane] : 0.f;   // still the previous step's action (zeros after a reset)
      float dsq = lane < nu ? (raw_action - prev_action) * (raw_action - prev_action) : 0.f;
      float rmse = sqrtf(grp_sum<LW>(dsq) / (float)nu);
      if (lane == 0) { inf[0] = rmse; inf[1] = S.sens[7]; inf[2] = S.sens[8]; inf[3] = S.sens[6]; }
      if (lane < nu) { inf[4 + lane] = S.tq[lane]; inf[4 + nu + lane] = raw_action * dm.rec[lane].a_scale; }
      if (lane < dm.ninfo_state) {
        const auto& R = dm.rec[lane];
        inf[4 + 2 * nu + lane] = (R.i_kind == 0 ? S.qpos[R.i_adr] : S.qvel[R.i_adr]) * R.i_gear;
      }
    }
    if (lane == 0) { A.terminated[(size_t)kstep * A.n_envs + env] = (uint8_t)terminated; A.truncated[(size_t)kstep * A.n_envs + env] = (uint8_t)truncated; }
  }

  // =============================================================== reset_model (flamingo_light_v1.py:209-232)
  int nan_resets = meta[4];
  if (bad) nan_resets++;
  if (do_reset) {
    if (lane < nq) S.qpos[lane] = dm.rec[lane].init_qpos;
    if (lane < NV) { S.qvel[lane] = 0.f; S.qacc[lane] = 0.f; }
    WSYNC();
    if (lane < dm.init_noise_nq)
      S.qpos[dm.rec[lane].n_qadr] += ob.init_noise * (2.f * u01(philox_first(k0, k1, step_count, 2u | ((unsigned)lane << 8), g0, g1)) - 1.f);
    WSYNC();
    // sensors of the mj_forward at the reset state: zero velocity, IMU orientation from the base quaternion
    {
      float xq[4] = {S.qpos[3], S.qpos[4], S.qpos[5], S.qpos[6]}, sq[4];
      qnorm(xq);
      qmul(sq, xq, dm.imu_quat);
      qnorm(sq);
      if (lane == 0) { for (int k = 0; k < 4; k++) S.sens[k] = sq[k]; for (int k = 4; k < 10; k++) S.sens[k] = 0.f; }
    }
    if (lane < L::NUMAX) { S.act[lane] = 0.f; S.tq[lane] = 0.f; }
    has_prev = 0;
    sim_step = 0;
  }

  // =============================================================== _get_obs + _build_state + _apply_command_inplace
  {
    WSYNC();
    float m[9];
    const float s_quat[4] = {S.sens[0], S.sens[1], S.sens[2], S.sens[3]};
    const float s_gyro[3] = {S.sens[4], S.sens[5], S.sens[6]}, s_vel[3] = {S.sens[7], S.sens[8], S.sens[9]};
    q2m(m, s_quat);
    const float pg[3] = {-m[6], -m[7], -m[8]};  // R^T (0,0,-1)
    const bool fill = do_reset;                 // reset fills every stack row with the first frame
    float* so = A.state_out + ((size_t)kstep * A.n_envs + env) * ob.state_dim;
    const int sd = ob.stacked_dim, S_ = ob.stack_size;
    for (int e = lane; e < ob.frame_dim; e += LW) {
      const int f = ob.el_field[e], idx = ob.el_index[e];
      // the rows of the observation stack that move down and the cached element: their loads (first touch of these lines in this
      // launch: HBM latency) are issued here, ahead of the observation arithmetic, not one dependent trip per stack row after it
      float* const cache = rec + lay.s_cache + e;
      float* const st = rec + lay.s_stack;
      const float cached = *cache;
      float old[4] = {0.f, 0.f, 0.f, 0.f};
      const bool short_stack = S_ <= 5;
      if (e < sd && !fill && short_stack) {
#pragma unroll
        for (int k = 0; k < 4; k++) if (k < S_ - 1) old[k] = st[k * sd + e];
      }
      float val = 0.f;
      switch (f) {
        case CS_OBS_DOF_POS: val = S.qpos[dm.rec[idx].o_qadr] * dm.rec[idx].o_qgear; break;
        case CS_OBS_DOF_VEL: val = S.qvel[dm.rec[idx].o_dadr] * dm.rec[idx].o_dgear; break;
        case CS_OBS_ANG_VEL: val = idx == 0 ? s_gyro[0] : (idx == 1 ? s_gyro[1] : s_gyro[2]); break;
        case CS_OBS_LIN_VEL: val = idx == 0 ? s_vel[0] : (idx == 1 ? s_vel[1] : s_vel[2]); break;
        case CS_OBS_PROJ_GRAVITY: val = idx == 0 ? pg[0] : (idx == 1 ? pg[1] : pg[2]); break;
        case CS_OBS_LAST_ACTION: val = S.act[idx]; break;
        case CS_OBS_HEIGHT_MAP: if (HF) {
          // get_height_map (utils/mujoco_utils.py:98-189): sample (i, j) = idx / res_x, idx % res_x of the window in the base
          // frame rotated by the raw base quaternion; vertical ray from 10 m above; value = robot_z - terrain_z
          const int rx = ob.hm_res_x, ry = ob.hm_res_y, i = idx / rx, j = idx - i * rx;
          const float xr = rx > 1 ? -0.5f * ob.hm_size_x + ob.hm_size_x * (float)j / (float)(rx - 1) : -0.5f * ob.hm_size_x;
          const float yr = ry > 1 ? -0.5f * ob.hm_size_y + ob.hm_size_y * (float)i / (float)(ry - 1) : -0.5f * ob.hm_size_y;
          const float qw = S.qpos[3], qx = S.qpos[4], qy = S.qpos[5], qz = S.qpos[6];
          const float wx = (1.f - 2.f * qy * qy - 2.f * qz * qz) * xr + (2.f * qx * qy - 2.f * qz * qw) * yr;   // relative to the base
          const float wy = (2.f * qx * qy + 2.f * qz * qw) * xr + (1.f - 2.f * qx * qx - 2.f * qz * qz) * yr;
          const float wz = S.qpos[2] + (2.f * qx * qz - 2.f * qy * qw) * xr + (2.f * qy * qz + 2.f * qx * qw) * yr;
          Terrain T;
          T.data = A.hfield; T.mip = A.hfield_mip; T.nrow = dm.hfield_nrow; T.ncol = dm.hfield_ncol;
          T.sx = dm.hfield_size[0]; T.sy = dm.hfield_size[1]; T.sz = dm.hfield_size[2]; T.gz = dm.ground_pos[2];
          T.ox = (double)S.qpos[0] - (double)dm.ground_pos[0]; T.oy = (double)S.qpos[1] - (double)dm.ground_pos[1];
          T.dx = 2.0 * (double)T.sx / (double)(T.ncol - 1); T.dy = 2.0 * (double)T.sy / (double)(T.nrow - 1);
          bool inside = false;
          const float hz = terrain_height(T, wx, wy, inside);
          val = (inside && hz <= wz + 10.f) ? S.qpos[2] - hz : S.qpos[2] + dm.heightmap_miss;
          break;
        }
        default: val = 0.f; break;
      }
      if (ob.noise_enabled && f != CS_OBS_LAST_ACTION && f != CS_OBS_COMMAND) {
        // truncated Gaussian by inverse CDF (scipy.stats.truncnorm.rvs, noise_generator_utils.py:22-28)
        const float mean = ob.noise_mean[f], sd_ = ob.noise_std[f];
        const float ca = ob.noise_ca[f], cb = ob.noise_cb[f];
        float z = normcdfinvf(ca + u01(philox_first(k0, k1, step_count, 1u | ((unsigned)e << 8), g0, g1)) * (cb - ca));
        float nz = fminf(ob.noise_upper[f], fmaxf(ob.noise_lower[f], mean + sd_ * z));
        val += nz;
      }
      float vs;
      const int interval = ob.el_interval[e];
      if (f == CS_OBS_COMMAND) vs = 0.f;
      else if (sim_step == 0 || (sim_step % interval) == 0) { vs = val * ob.el_scale[e]; *cache = vs; }
      else vs = cached;
      const bool is_cmd = f == CS_OBS_COMMAND;
      const float cmdv = is_cmd ? S.cmd[idx] : 0.f;
      if (e < sd) {
        if (short_stack) {
#pragma unroll
          for (int k = 4; k >= 1; k--)
            if (k < S_) {
              const float o = fill ? vs : old[k - 1];
              st[k * sd + e] = o;
              so[k * sd + e] = is_cmd ? cmdv : o;
            }
        } else {
          for (int k = S_ - 1; k >= 1; k--) {
            float o = fill ? vs : st[(k - 1) * sd + e];
            st[k * sd + e] = o;
            so[k * sd + e] = is_cmd ? cmdv : o;
          }
        }
        st[e] = vs;
        so[e] = is_cmd ? cmdv : vs;
      } else {
        so[S_ * sd + (e - sd)] = is_cmd ? cmdv : vs;
      }
    }
  }

  // =============================================================== state write-back
  STAMP(9);   // observation build + info
  if (PROF && A.dbg != nullptr && wlane == 0)
  {
    for (int i = 0; i < 16; i++) atomicAdd(reinterpret_cast<unsigned long long*>(A.dbg) + i, pacc[i]);
    for (int i = 0; i < 16; i++) atomicAdd(reinterpret_cast<unsigned long long*>(A.dbg) + 16 + i, pext[i]);
  }
  if (lane < nq) rec[lay.s_qpos + lane] = S.qpos[lane];
  if (lane < NV) { rec[lay.s_qvel + lane] = S.qvel[lane]; rec[lay.s_warm + lane] = S.qacc[lane]; }
  if (lane < nu) {
    rec[lay.s_lastact + lane] = do_reset ? 0.f : S.act[lane];
    if (kmode == MODE_STEP) rec[lay.s_delay + lane] = S.act[lane];   // control_manager.py:22: the raw action is what the filter keeps (has_prev = 0 after a reset: never read)
  }
  if (lane == 0) {
    meta[0] = sim_step; meta[1] = (int)(step_count + 1u); meta[2] = has_prev; meta[4] = nan_resets;
    meta[5] += st_newton; meta[6] += st_ls; meta[7] += st_build; meta[3] += st_rows;
    meta[8] += st_dropcon + st_walkcut; meta[9] += st_droplim; meta[10] = max(meta[10], st_maxcon); meta[13] += st_walkcut;
    if (kmode == MODE_STEP && (terminated || truncated)) meta[11] += 1;   // episodes ended (device-side count: survives graph replay)
    if constexpr (FIX) { meta[12] += 1; A.ovf[env] = 0; }                  // control steps redone by the large-capacity kernel
  }
#undef ob
#undef lay
}

// waves per CU that LDS admits -> waves per SIMD to ask the register allocator for.  13 waves per CU are 4 + 3 + 3 + 3: asking for
// 4 everywhere (128 registers) made the flamingo_light_v1 coarse-heightfield kernel spill 169 registers for one extra wave on one
// SIMD (4.4 -> 3.7 M env-steps/s on light_rocky); 9 and 10 keep asking for 3 (the fine-cell kernels were tuned there)
constexpr int waves_per_simd(int per_cu) { return per_cu >= 15 ? 4 : per_cu >= 9 ? 3 : per_cu >= 5 ? 2 : 1; }

template <int NV, int NB, int RPL, bool HF, int GTM, bool SC, bool PROF = false, int EPW = 1, int MCT = 0>
// waves per SIMD the register allocator is asked for = what the LDS footprint admits (160 KiB per CU, 4 SIMDs): asking for more makes
// the compiler spill for nothing, asking for less wastes resident waves
__global__ __launch_bounds__(64, (EPW == 2 ? 2 : waves_per_simd(163840 / (int)sizeof(typename KTraits<NV, NB, RPL, HF, SC, EPW, MCT>::L)))) void env_kernel(KArgs kernarg_block) {
  KArgsP kargs_p = (KArgsP)__builtin_amdgcn_kernarg_segment_ptr();
  (void)kernarg_block;
  __shared__ typename KTraits<NV, NB, RPL, HF, SC, EPW, MCT>::L SS[EPW];
  const int env = A.mode == MODE_DEBUG ? A.dbg_env : A.env_first + (int)blockIdx.x * EPW + (EPW == 1 ? 0 : ((int)threadIdx.x >> 5));
  if (env >= A.n_envs) return;
  env_body<NV, NB, RPL, HF, GTM, SC, PROF, EPW, MCT, false>(kargs_p, env, SS);
}

// Split pipeline (heightfield kernels whose narrowphase dwarfs everything else: humanoid_p_v0 on 1 cm stairs cells spends 89 % of a
// fused control step in the prism walk, at one wave per SIMD -- 39 KB of LDS and 278 registers per env -- so every dependent load and
// fp64 portal operation is fully exposed, and a launch lasts as long as its slowest, fallen, env).  Per substep two launches:
//   env_narrow_kernel: kinematics + the prism walk only, A.nw waves per env (wave w takes the geoms g % nw == w: a standing robot's
//     two feet walk side by side, a fallen one's twenty geoms spread over all waves), slim LDS and registers -> several waves per
//     SIMD; contacts go to an L2-resident record per (env, geom), in MuJoCo's strip order;
//   env_step_kernel: the solver kernel, one substep, ground contacts read from that record (robot-robot pairs stay here).
// OCC: waves per SIMD the register allocator is asked for (4: 128 registers, the fp64 portal of MPR spills ~150 of them; 2: 256, none)
template <int NV, int NB, int RPL, bool HF, int GTM, bool SC, int MCT, int OCC, bool PROF = false>
__global__ __launch_bounds__(64, OCC) void env_narrow_kernel(KArgs kernarg_block) {
  KArgsP kargs_p = (KArgsP)__builtin_amdgcn_kernarg_segment_ptr();
  (void)kernarg_block;
  __shared__ typename KTraits<NV, NB, RPL, HF, SC, 1, MCT, 1>::L SS[1];
  const int nw = A.nw;
  const int env = A.mode == MODE_DEBUG ? A.dbg_env : A.env_first + (int)blockIdx.x / nw;
  if (env >= A.n_envs) return;
  env_body<NV, NB, RPL, HF, GTM, SC, PROF, 1, MCT, false, 1>(kargs_p, env, SS, (int)blockIdx.x % nw);
}
template <int NV, int NB, int RPL, bool HF, int GTM, bool SC, int MCT>
__global__ __launch_bounds__(64, waves_per_simd(163840 / (int)sizeof(typename KTraits<NV, NB, RPL, HF, SC, 1, MCT, 2>::L))) void env_step_kernel(KArgs kernarg_block) {
  KArgsP kargs_p = (KArgsP)__builtin_amdgcn_kernarg_segment_ptr();
  (void)kernarg_block;
  __shared__ typename KTraits<NV, NB, RPL, HF, SC, 1, MCT, 2>::L SS[1];
  const int env = A.mode == MODE_DEBUG ? A.dbg_env : A.env_first + (int)blockIdx.x;
  if (env >= A.n_envs) return;
  env_body<NV, NB, RPL, HF, GTM, SC, false, 1, MCT, false, 2>(kargs_p, env, SS);
}

// Rollout launch: the reference's loop (core/tester.py:66-97) with the policy replaced by an action table, K control steps per
// launch.  A wave stays on its env for all K steps, so no env waits for the slowest env of its launch at every step (the tail that
// range launches only partly fill); every step's state vector, flags and info go to row k of [K][N][...] buffers.  Between two steps
// the wave's own stores must be what its loads see: release / acquire at agent scope (write-back + L1 invalidate).  A dense fleet
// kernel that meets more contacts than it has slots at step k flags the env with k + 1 and leaves: env_rollout_fix_kernel takes the
// env from step k to the end with the large-capacity body.
template <int NV, int NB, int RPL, bool HF, int GTM, bool SC, int MCT>
__global__ __launch_bounds__(64, waves_per_simd(163840 / (int)sizeof(typename KTraits<NV, NB, RPL, HF, SC, 1, MCT>::L))) void env_rollout_kernel(KArgs kernarg_block) {
  KArgsP kargs_p = (KArgsP)__builtin_amdgcn_kernarg_segment_ptr();
  (void)kernarg_block;
  __shared__ typename KTraits<NV, NB, RPL, HF, SC, 1, MCT>::L SS[1];
  const int env = A.env_first + (int)blockIdx.x;
  if (env >= A.n_envs) return;
  const int K = A.roll_steps;
#pragma nounroll
  for (int k = 0; k < K; k++) {
    env_body<NV, NB, RPL, HF, GTM, SC, false, 1, MCT, false>(kargs_p, env, SS, 0, k);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (A.ovf != nullptr && __builtin_amdgcn_readfirstlane(A.ovf[env]) != 0) break;   // abandoned at step k: the fix kernel continues
  }
}
template <int NV, int NB, int RPL, bool HF, int GTM, bool SC, int MCT>
__global__ __launch_bounds__(64, 4) void env_rollout_fix_kernel(KArgs kernarg_block) {
  KArgsP kargs_p = (KArgsP)__builtin_amdgcn_kernarg_segment_ptr();
  (void)kernarg_block;
  __shared__ typename KTraits<NV, NB, RPL, HF, SC, 1, MCT>::L SS[1];
  const int base = A.env_first + (int)blockIdx.x * 64, e = base + (int)threadIdx.x;
  const int fl = (e < A.env_first + A.env_count && e < A.n_envs) ? A.ovf[e] : 0;
  unsigned long long m = __ballot(fl != 0);
  const int K = A.roll_steps;
#pragma nounroll
  while (m) {   // wave-uniform
    const int src = __builtin_ctzll(m), env = base + src;
    m &= m - 1;
    const int k0 = __builtin_amdgcn_readlane(fl, src) - 1;
#pragma nounroll
    for (int k = k0; k < K; k++) {
      env_body<NV, NB, RPL, HF, GTM, SC, false, 1, MCT, true>(kargs_p, env, SS, 0, k);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    WSYNC();
  }
}

// The large-capacity kernel behind a fleet kernel whose contact slots can run out (flamingo_light_v1 on the plane: 14 dense
// contacts in the fleet kernel, 40 = four per geom, the most the plane narrowphase can emit, here).  Launched right after the fleet
// kernel on the same stream with one wave per 64 envs of the range: the wave reads its 64 flags (one coalesced load; almost always
// all zero: the launch costs a few microseconds) and redoes the control step of each flagged env from the untouched pre-step state.
// MuJoCo's arena keeps every contact (reference flamingo_light_v1.py:154, do_simulation); this keeps that true at any count.
// Registers and LDS per wave are those of the fleet kernel (launch bounds: four waves per SIMD = 128 registers; 9.7 KB of LDS): a fix-up
// wave then fits the slot a finished fleet wave leaves behind.  With the footprint the compiler would choose freely (247 registers) it
// had to wait until three of the four waves of some SIMD -- other ranges' fleet waves, a whole kernel long -- had gone: 13.9 -> 12.6 M.
template <int NV, int NB, int RPL, bool HF, int GTM, bool SC, int MCT>
__global__ __launch_bounds__(64, 4) void env_fixup_kernel(KArgs kernarg_block) {
  KArgsP kargs_p = (KArgsP)__builtin_amdgcn_kernarg_segment_ptr();
  (void)kernarg_block;
  __shared__ typename KTraits<NV, NB, RPL, HF, SC, 1, MCT>::L SS[1];
  const int base = A.env_first + (int)blockIdx.x * 64, e = base + (int)threadIdx.x;
  const int fl = (e < A.env_first + A.env_count && e < A.n_envs) ? A.ovf[e] : 0;
  unsigned long long m = __ballot(fl != 0);
#pragma nounroll
  while (m) {   // wave-uniform
    const int env = base + __builtin_ctzll(m);
    m &= m - 1;
    env_body<NV, NB, RPL, HF, GTM, SC, false, 1, MCT, true>(kargs_p, env, SS);
    WSYNC();
  }
}
#undef A
#undef KARGS_FENCE


// Diagnostic / test kernel: the support function of mesh geom `geom` (identity pose) for n directions, through every code path that
// answers such queries: [0] lane-parallel (one direction per lane), [1] wave-cooperative (one direction per wave at a time), [2..7] the
// six-direction box routine's hi/lo corners for the pose given by quaternion dirs-as-axis (only for the first direction of each wave).
__global__ void support_probe_kernel(const DevModel* dmp, HullGraph H, int geom, const float* dirs, int n, float* out, int use_map) {
  const int ln = threadIdx.x, i = blockIdx.x * 64 + ln;
  CObj o;
  o.kind = CS_GEOM_MESH;
  for (int k = 0; k < 3; k++) { o.pos[k] = 0.f; o.size[k] = 0.f; o.center[k] = 0.f; }
  o.q[0] = 1.f; o.q[1] = o.q[2] = o.q[3] = 0.f;
  o.adr = dmp->rec[geom].g_hulladr; o.num = dmp->rec[geom].g_hullnum; o.map = use_map ? dmp->g_hullmap[geom] : -1;
  float d[3] = {1.f, 0.f, 0.f};
  if (i < n) for (int k = 0; k < 3; k++) d[k] = dirs[3 * (size_t)i + k];
  float p[3];
  cobj_support<GT_MESH, false>(o, H, d, p, ln);
  if (i < n) for (int k = 0; k < 3; k++) out[6 * (size_t)i + k] = p[k];
  for (int j = 0; j < 64; j++) {
    const int ij = blockIdx.x * 64 + j;
    if (ij >= n) break;
    const float dj[3] = {rl(d[0], j), rl(d[1], j), rl(d[2], j)};
    float pj[3];
    cobj_support<GT_MESH, true>(o, H, dj, pj, ln);
    if (ln == 0) for (int k = 0; k < 3; k++) out[6 * (size_t)ij + 3 + k] = pj[k];
  }
}

}  // namespace cosim
