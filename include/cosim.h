/* cosim.h — C ABI of the MI355X batched rollout engine (libcosim_hip.so).
 *
 * Drop-in boundary.  The reference is pure Python: its hot path is
 *     env.step(action) -> StateBuildWrapper/TimeLimitWrapper/CommandWrapper -> <Robot>.step -> mj_step x frame_skip
 * (reference envs/wrappers.py:258-269,309-320,391-405; envs/flamingo_light_v1/flamingo_light_v1.py:131-164), where
 * the arithmetic is reached through pybind11 into libmujoco.  The entry points below are what a batched FFI for that
 * path binds; each comment names the reference interface it replaces.  Plain pointers and sizes only, no torch types.
 *
 * Conventions: every function returns 0 on success or a negative COSIM_E* code (cosim_last_error() has the text);
 * `*_dev` pointers are device (HBM) pointers owned by the caller; calls are ordered on `stream` (a hipStream_t passed
 * as void*, NULL = default stream) and are not re-entrant per handle.
 */
#ifndef COSIM_H
#define COSIM_H

#include <stdint.h>

#include "cosim_model.h"

#ifdef __cplusplus
extern "C" {
#endif

#define COSIM_OK 0
#define COSIM_EINVAL (-1)  /* bad argument / model the engine cannot run */
#define COSIM_EHIP (-2)    /* HIP runtime error */
#define COSIM_ENOGPU (-3)  /* no usable device */

#define CS_MAXFIELD 16
#define CS_MAXCMD 6

/* observation field ids: keys of obs_to_dim (reference flamingo_light_v1.py:68-77) */
#define CS_OBS_DOF_POS 0
#define CS_OBS_DOF_VEL 1
#define CS_OBS_ANG_VEL 2
#define CS_OBS_LIN_VEL 3
#define CS_OBS_PROJ_GRAVITY 4
#define CS_OBS_LAST_ACTION 5
#define CS_OBS_HEIGHT_MAP 6
#define CS_OBS_COMMAND 7

/* The wrapper-layer configuration: what StateBuildWrapper / TimeLimitWrapper / CommandWrapper and the robot env read
 * from config["observation"], config["env"], config["random"] (reference envs/wrappers.py:89-127,290-301,336-347;
 * flamingo_light_v1.py:36-42,56-77). */
typedef struct cosim_obs_config {
  int stack_size, command_dim;
  int n_stacked, n_non_stacked;                 /* entries used in the field lists below */
  int stacked_field[CS_MAXFIELD], non_stacked_field[CS_MAXFIELD]; /* CS_OBS_* in config order */
  int field_dim[8];                             /* obs_to_dim per CS_OBS_* id */
  int field_interval[8];                        /* max(1, round(control_freq / freq)) (wrappers.py:190) */
  float field_scale[8];                         /* config["observation"][name]["scale"] */
  float noise_mean[8], noise_std[8], noise_lower[8], noise_upper[8]; /* random_table sensor_noise level per field */
  int noise_enabled;                            /* 0: additive noise skipped (parity runs: level "none" == exactly 0) */
  int position_command;                         /* config["env"]["position_command"] */
  float command_scales[CS_MAXCMD];
  int max_sim_step;                             /* int(max_duration * control_freq) (wrappers.py:300) */
  float action_delay_prob, init_noise;          /* config["random"] */
  int auto_reset;                               /* engine extension: reset an env in the step that ends its episode */
  /* height map (config["observation"]["height_map"], reference utils/mujoco_utils.py:98-189) */
  int hm_res_x, hm_res_y;
  float hm_size_x, hm_size_y;
} cosim_obs_config_t;

typedef struct cosim_engine cosim_engine_t;

/* Replaces build_env(config) (reference envs/build.py:8-24): compiled model + wrapper config -> N env instances on
 * one GPU.  env_id0 is the global id of local env 0 (RNG streams are keyed by global env id, so results do not depend
 * on how envs are sharded over GPUs).  hull_* / hfield are host pointers (copied). */
int cosim_create(const cosim_model_t* model, const float* hull_vert, const int* hull_adr, const int* hull_nbr,
                 const float* hfield, const cosim_obs_config_t* obs, int n_envs, int device, uint64_t seed,
                 int64_t env_id0, cosim_engine_t** out);
int cosim_destroy(cosim_engine_t* e);

/* Sizes the caller needs to allocate buffers: "state_dim", "action_dim", "command_dim", "info_dim", "nq", "nv",
 * "n_envs", "state_stride", "param_stride", "lds_bytes", "vgprs" ... ; returns the value or a negative error. */
int cosim_query(const cosim_engine_t* e, const char* name);

/* Per-env parameters (domain randomisation; replaces the per-construction MJCF rewrite of XMLManager.get_model_path,
 * reference manager/xml_manager.py:43-87).  name: "body_mass"[N,nbody] "body_invweight0"[N,nbody] (translational)
 * "dof_invweight0"[N,nv] "meaninertia"[N] "dof_frictionloss"[N,nv] "geom_friction"[N,ngeom] (sliding, already
 * max-combined with the ground; robot-robot pairs use the model's geom friction) "kp"[N,nu] "kd"[N,nu].  `host` points to
 * host memory, float32, row-major.  Engine scalars (count 1): "solver_tolerance" (fp32 Newton tolerance), "max_newton",
 * "max_ls" (iteration caps below the model's), "envs_per_wave" (1 | 2: kernel variant, 2 only for flat flamingo_light_v1 and
 * even env counts), "wave_priority" (count 4: s_setprio by solver lag -- Newton iterations taken as usual per substep, then the
 * lag thresholds of priority 1, 2, 3; a huge first threshold switches it off), "debug_substeps" (diagnostics: physics substeps
 * per control step, 0 = frame_skip), "boxbox_mode" (1, default: box-box geom pairs through the mjc_BoxBox routine, up to eight contacts
 * per pair; 0: through MPR like the other convex pairs, one contact), "pair_mode" (1, default: robot-robot pairs with a hull one at
 * a time with wave-cooperative vertex scans; 0: lane-parallel), "contact_twist" (flat flamingo_light_v1 only; 1: ground contacts in
 * twist space, 32 slots instead of the dense-row kernel's 14, at ~70 % of the speed; before the first step), "ls_tolerance_scale"
 * (multiplies the line-search tolerance; 1 = the model's), "ranges" (1..16: cosim_step issues the fleet as that many launches over
 * contiguous env ranges on engine-owned streams; default 1), "deferred_join" (see cosim_step / cosim_join), "inflight" (control steps cosim_step lets the host run ahead of each
 * range stream before it blocks, default 2, 0 = unbounded: deep queues step slower on this runtime), "split" (heightfield kernels that have the two-kernel pipeline -- humanoid_p_v0:
 * the prism walk in a kernel of its own, "narrow_waves" (default 6) waves per env, and the solver one substep per launch; 0 goes back
 * to the fused kernel), "fixup" (0 switches the
 * large-capacity fix-up launches off: contacts beyond the fleet kernel's slots are then left out and counted), "support_map" (1,
 * default: support queries on mesh geoms with 32 or more hull vertices go through the hull's support map -- the few vertices that can
 * win in the direction's cube-map cell, same arg max as the scan; 0: every query scans the whole hull, for A/B runs and tests),
 * "timing_stride" (default 1: with cosim_set_timing on, an event pair around every launch; n: around every n-th launch -- the
 * events cost ~4 % of a 20-step run at 1, choose n coprime with "ranges" so that every range is sampled),
 * "block_cull" (1, default: the narrowphase kernel of the split pipeline tests blocks of 8 prisms -- height and oriented box -- before
 * their prisms; 0: every block goes on to the per-prism tests; same contacts either way).
 * cosim_query additionally answers "contact_slots" / "pair_slots" (capacity of the selected kernel variant: heightfields with cells
 * of 10 cm or more select the 48-slot variants of flamingo_light_v1 / w4_p_v2), "fixup_contact_slots" (capacity of the kernel that
 * redoes a control step whose contacts did not fit; 0: this model / terrain has none), "ranges" and "lds_bytes". */
int cosim_set_param(cosim_engine_t* e, const char* name, const float* host, int count);

/* Replaces env.reset() (reference envs/wrappers.py:245-256,303-307,385-389; flamingo_light_v1.py:209-232).
 * mask_dev: uint8[N] or NULL (= all).  commands_dev: float[N,command_dim] user commands (see cosim_step).
 * state_out_dev: float[N,state_dim] (rows of envs that are not reset are left untouched). */
int cosim_reset(cosim_engine_t* e, const uint8_t* mask_dev, const float* commands_dev, float* state_out_dev,
                void* stream);

/* Replaces receive_user_command() + env.step(action) (reference core/tester.py:68,90; envs/wrappers.py:349-405).
 * actions_dev float[N,nu]; commands_dev float[N,command_dim] raw user commands (scaling / position-mode transform is
 * applied in the kernel from the pre-step pose, as CommandWrapper.receive_user_command does before the step);
 * state_out_dev float[N,state_dim]; terminated_dev/truncated_dev uint8[N]; info_out_dev float[N,info_dim] or NULL:
 * [action_diff_RMSE, lin_vel_x, lin_vel_y, ang_vel_yaw, torque[nu], set_points[nu], state[k]]. */
int cosim_step(cosim_engine_t* e, const float* actions_dev, const float* commands_dev, float* state_out_dev,
               uint8_t* terminated_dev, uint8_t* truncated_dev, float* info_out_dev, void* stream);
/* With "ranges" > 1, cosim_step forks: the engine's range streams wait for `stream` (the step's inputs), each steps its range, and
 * `stream` then waits for all of them (join) -- unless "deferred_join" is 1: the join is then left to cosim_join, or to the next
 * cosim_reset / cosim_get / cosim_set / cosim_event_push (they join first).  A deferred join is what lets a range's next control
 * step start while the other ranges are still inside the current one (a launch ends with its slowest env): a caller whose next
 * actions do not depend on the whole fleet's last outputs (an action table, or a policy evaluated per range on the range's stream:
 * cosim_range) calls cosim_step back to back and joins when it reads results.  With a deferred join the INPUT buffers of a step (actions_dev,
 * commands_dev) must stay untouched until that step has run: at most "inflight" (default 2) steps are in flight, so rotating three
 * action buffers, or an action table, is enough; the output buffers hold the newest step's results after the join.
 * (The reference steps one env: core/tester.py:90.) */
/* The reference's loop itself (core/tester.py:66-97: command -> policy -> step -> reporter, until done) with the policy replaced by an
 * action table: `steps` control steps in ONE launch per range.  actions_dev is [steps][N][nu]; state_out_dev [steps][N][state_dim],
 * terminated_dev / truncated_dev [steps][N], info_out_dev [steps][N][info_dim] or NULL: row k holds what cosim_step would have been
 * given / would have returned at step k (auto-reset included); commands_dev [N][command_dim] holds for the whole rollout.  A wave
 * stays on its env for all the steps, so no env waits at every step for the slowest env of its launch.  Envs that a dense fleet
 * kernel abandons at step k (more contacts than slots) finish the rollout in the large-capacity kernel.  Available where
 * cosim_query "rollout" is 1; the caller's stream waits for the whole rollout on return. */
int cosim_rollout(cosim_engine_t* e, int steps, const float* actions_dev, const float* commands_dev, float* state_out_dev, uint8_t* terminated_dev,
                  uint8_t* truncated_dev, float* info_out_dev, void* stream);
/* Test hook, host only (no GPU call): the support map the engine builds for a mesh geom's convex hull (csrc/cosim_hullmap.h: per cell
 * of a cube map of directions, the vertices that can be the support point somewhere in the cell) against the full scan over the hull
 * that the reference's support function performs (mjc_support -> arg max of dir . vertex).  verts [n][3]; adr [n + 1] / nbr: CSR
 * neighbour graph of the hull, ids local to the hull; dirs [ndir][3] in the hull's frame.  out_map_idx / out_scan_idx [ndir]: the
 * arg-max vertex through the map / by scanning; out_stats[3] (may be NULL): candidates in the table, largest cell, cells. */
int cosim_hull_support_check(const float* verts, int n, const int* adr, const int* nbr, const float* dirs, int ndir, int* out_map_idx,
                             int* out_scan_idx, int* out_stats);
/* Test hook (GPU): the device's own support routines on mesh geom `geom` at identity pose, for n_dirs directions (host float[n][3]):
 * out_host [n_dirs][6] = support point from the lane-parallel routine, then from the wave-cooperative one.  use_map 0: full scans. */
int cosim_debug_support(cosim_engine_t* e, int geom, const float* dirs_host, int n_dirs, float* out_host, int use_map);
int cosim_join(cosim_engine_t* e, void* stream);
/* Range i of "ranges": its first env, env count and stream (hipStream_t; NULL when ranges == 1).  Work enqueued on that stream from
 * outside (a per-range policy, a reporter reduction) is ordered with the range's steps; cosim_range_mark(i) re-arms the range's
 * "done" event afterwards so that the next join waits for that work too. */
int cosim_range(const cosim_engine_t* e, int i, int* first, int* count, void** stream);
int cosim_range_mark(cosim_engine_t* e, int i);

/* The same control step for envs [first, first + count) only; every pointer still addresses the WHOLE fleet's buffers ([N, ...]).
 * Lets a caller step one fleet as several independent shards on streams of its own (a shard's next control step fills the tail
 * of the others' launches: a launch ends with its slowest env) without one handle per shard.  Envs never interact and every
 * random stream is keyed by the global env id, so results do not depend on the split.  (No reference counterpart: the
 * reference steps one env, core/tester.py:90.) */
int cosim_step_range(cosim_engine_t* e, int first, int count, const float* actions_dev, const float* commands_dev, float* state_out_dev,
                     uint8_t* terminated_dev, uint8_t* truncated_dev, float* info_out_dev, void* stream);

/* Replaces env.get_data() reads (reference flamingo_light_v1.py:247-248; wrappers.py:360-367): copies
 * "qpos"[N,nq] / "qvel"[N,nv] / "qacc_warmstart"[N,nv] / "sim_step"[N] (as float) to a device buffer. */
int cosim_get(cosim_engine_t* e, const char* name, float* out_dev, void* stream);
/* Test / checkpoint hook: overwrite "qpos"/"qvel"/"qacc_warmstart" from a device buffer. */
int cosim_set(cosim_engine_t* e, const char* name, const float* in_dev, void* stream);

/* Replaces env.event("push", v) (reference flamingo_light_v1.py:234-245): v_dev float[N,3] world-frame velocity,
 * mask_dev uint8[N] or NULL. */
int cosim_event_push(cosim_engine_t* e, const float* v_dev, const uint8_t* mask_dev, void* stream);

/* Debug hook for the parity tests: runs ONE mj_forward-equivalent on env `env` in a diagnostic kernel and copies the
 * named intermediate to host doubles-as-float: "xpos" "xquat" "M" "cdof" "contacts" "J" "efc" "qacc" ... */
int cosim_debug_forward(cosim_engine_t* e, int env, const char* name, float* host_out, int capacity);

/* Average duration (ms) of the step kernel since the last call, measured with HIP events on the launch stream, and
 * the number of launches averaged; resets the accumulator. */
/* Diagnostics: the 32 64-bit counters diagnostic kernel builds accumulate (cosim_set_param "narrow_occupancy" 0: the narrowphase
 * kernel of the split pipeline -- [0] sum, [1] max, [2] number of wave lifetimes in shader-clock cycles, [3] work items, [8..15]
 * the walk's phases; see tools/gpu_narrow_prof.py); clear != 0 zeroes them afterwards. */
int cosim_debug_counters(cosim_engine_t* e, unsigned long long* out32, int clear);
int cosim_kernel_time(cosim_engine_t* e, float* avg_ms, int* launches);
int cosim_set_timing(cosim_engine_t* e, int enabled);
/* Diagnostic build of the step kernel with s_memtime stamps at phase boundaries (one variant per bench workload): one control
 * step; cycles_out16 must hold 32 doubles: [i < 16] = mean shader-clock cycles per env spent in phase i, [16..23] = the heightfield
 * narrowphase's split (flat kernels: robot-robot pairs, contact-matrix accumulation, tree pass), [24..28] = hull pairs: MPR runs,
 * hits, refinement iterations, cycles, pairs past the bounding spheres; see tools/gpu_phases.py.  Never timed. */
int cosim_profile_step(cosim_engine_t* e, const float* actions_dev, const float* commands_dev, float* state_out_dev,
                       uint8_t* terminated_dev, uint8_t* truncated_dev, double* cycles_out16);

/* Policy side of the loop (reference core/policy.py:11-21, one state per call on the CPU): the actor MLP of an ONNX policy for
 * all N envs in one launch on the matrix pipe, out = clip(act_L(... act_1(x W_1^T + b_1) ...)).  All pointers are device
 * pointers; dims[n_layers + 1] (each 1..512); w_dev[l] is [dims[l+1], dims[l]] row-major (Gemm with transB = 1), b_dev[l] may
 * be NULL; act[l]: 0 none, 1 relu, 2 tanh, 3 elu, 4 sigmoid, 5 leaky relu (act_alpha[l]); clip > 0 clamps to [-clip, clip]. */
int cosim_mlp_forward(const float* x_dev, int n, int n_layers, const int* dims, const float* const* w_dev, const float* const* b_dev,
                      const int* act, const float* act_alpha, float clip, float* out_dev, void* stream);

/* The recurrent policy's cell (reference core/policy.py:24-47: an ONNX LSTM node fed one step at a time with h_in / c_in kept by the
 * caller): one LSTM step for all N envs in one launch.  Gate order and layouts are ONNX's: w_dev [4H, I], r_dev [4H, H], b_dev [8H]
 * (Wb then Rb) or NULL, gates i, o, f, c; default activations.  h_out_dev / c_out_dev may alias h_dev / c_dev (in-place state). */
int cosim_lstm_cell(const float* x_dev, const float* h_dev, const float* c_dev, int n, int in_dim, int hidden, const float* w_dev,
                    const float* r_dev, const float* b_dev, float* h_out_dev, float* c_out_dev, void* stream);

/* Reporter side (reference core/reporter.py:210-218 write_info, :429-442, :506-508): fleet statistics of one step's info in one
 * launch.  acc_dev is double[3][K], K = 4 + nu + ncmd <= 32: count, sum, sum of squares of info[:, 0:4], |info[:, 4:4+nu]| (torque)
 * and |cmd[:, i] - info[:, 1 + i]| for i < ncmd <= 3 (command tracking); cmd_dev is [N, cmd_stride]. */
int cosim_fleet_stats(const float* info_dev, int n, int info_dim, int nu, const float* cmd_dev, int cmd_stride, int ncmd, double* acc_dev,
                      void* stream);

/* Percentiles for the same columns (reference core/reporter.py:429-442, 506-530 keeps and plots every sample of one env; a fleet
 * keeps a mergeable sketch): adds this step's rows to hist_dev, double[K][nbins], bin b of column c = magnitudes in
 * [b, b + 1) * hi_dev[c] / nbins (the last bin also takes what lies above hi_dev[c]).  Sums over steps and ranks are percentiles'
 * sufficient statistic; cosim_amd/reporter.py turns them into p5 / p50 / p95. */
int cosim_fleet_hist(const float* info_dev, int n, int info_dim, int nu, const float* cmd_dev, int cmd_stride, int ncmd, const float* hi_dev,
                     int nbins, double* hist_dev, void* stream);

const char* cosim_last_error(void);
int cosim_model_sizeof(void);
int cosim_obs_config_sizeof(void);

#ifdef __cplusplus
}
#endif
#endif /* COSIM_H */
