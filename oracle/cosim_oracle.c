/* cosim_oracle.c — CPU fp64 restatement of the reference's physics step.  TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call
 * this file.  The product (cosim_amd/csrc) never links or calls it.
 *
 * What it restates: the arithmetic behind the reference's
 *     self.do_simulation(self.applied_torques, self.frame_skip)
 * (reference envs/flamingo_light_v1/flamingo_light_v1.py:154) and the mj_forward of reset_model
 * (:220), i.e. gymnasium 1.0.0 MujocoEnv.do_simulation -> mujoco.mj_step x frame_skip.  That
 * arithmetic lives in the un-vendored dependency mujoco==3.2.7 (reference requirements.txt:19),
 * which is absent from /root/reference and not installable here, so this file restates MuJoCo's
 * *published* algorithm (MuJoCo documentation, "Computation" chapter, and the open-source tree at
 * tag 3.2.7; the upstream function each block follows is named in its comment) for exactly the
 * feature set the four cosim MJCF files use: free + hinge joints, implicitfast integrator, Newton
 * solver with exact line search, pyramidal cones, connect equalities, dof frictionloss, joint limits,
 * plane/hfield ground against sphere / cylinder / box / convex-mesh geoms, robot-robot pairs (MPR;
 * box-box manifolds), motors, IMU sensors.
 *
 * PARITY UNPINNED at the MuJoCo boundary: the reference ships no test, golden vector or recorded
 * trajectory for this path and MuJoCo cannot be run in this container.  What
 * pins this file instead: analytic known-answer tests (tests/test_oracle_physics.py), closed-form
 * known answers of the published constraint model on hand-built one-body models
 * (tests/test_oracle_kat.py: impedance interpolation a1 = (1 - d) a0 + d aref, solimp curve, (K, B)
 * from solref, Huber friction loss, limit row, cone edges and friction mixing) and the golden
 * vectors captured from the reference's importable pure-Python pieces (tests/golden/).  Still
 * flagged "restated from memory, nothing independent confirms it": the pyramidal regulariser
 * (diagApprox = tran (1 + mu^2), Rpy = 2 mu^2 R; its observable consequence is fixed by
 * test_sphere_rest_penetration_documents_the_pyramidal_regulariser), per-row impedance of connect
 * constraints, the bracketing details of the line search, plane-mesh neighbour contacts, MPR, and
 * the box-box routine (structure of mjc_BoxBox; its clipped manifolds are pinned by closed-form
 * geometry only, test_box_box_manifolds_have_closed_form_vertices).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/cosim_model.h"

#define MINVAL 1e-15
#define MINIMP 0.0001
#define MAXIMP 0.9999
#define MINMU 1e-5
#define MAXCON 1280
#define MAXEFC (6 * CS_MAXEQ / 2 + 3 * CS_MAXDOF + 4 * MAXCON)
#define NV CS_MAXDOF

enum { ST_QUADRATIC = 0, ST_SATISFIED, ST_LINEARNEG, ST_LINEARPOS };
enum { CT_EQUALITY = 0, CT_FRICTION, CT_LIMIT, CT_CONTACT };

typedef struct {
  double dist, pos[3], frame[9], friction[5], solref[2], solimp[5], includemargin, mu;
  int geom, body, dim, efc_address; /* geom2 / its body (the "upper" geom of a ground contact) */
  int geom1, body1;                 /* geom1 = -1, body1 = 0: the static ground */
} contact_t;

typedef struct oracle_data {
  cosim_model_t m; /* private copy: per-env parameters are edited in place */
  const float* hull_vert;
  const int* hull_adr;
  const int* hull_nbr;
  const float* hfield;
  /* state */
  double qpos[CS_MAXQ], qvel[NV], qacc_warmstart[NV], ctrl[CS_MAXU];
  double time;
  /* position-dependent */
  double xpos[CS_MAXBODY][3], xquat[CS_MAXBODY][4], xmat[CS_MAXBODY][9], xipos[CS_MAXBODY][3], ximat[CS_MAXBODY][9];
  double xanchor[CS_MAXJNT][3], xaxis[CS_MAXJNT][3];
  double subtree_com[CS_MAXBODY][3], cinert[CS_MAXBODY][10], crb[CS_MAXBODY][10], cdof[NV][6];
  double M[NV][NV], L[NV][NV]; /* dense mass matrix and its Cholesky factor */
  int ncon, nefc, ne, nf, nl;
  contact_t con[MAXCON];
  double J[MAXEFC][NV], efc_pos[MAXEFC], efc_margin[MAXEFC], efc_frictionloss[MAXEFC], efc_diagApprox[MAXEFC];
  double efc_KBIP[MAXEFC][4], efc_R[MAXEFC], efc_D[MAXEFC], efc_vel[MAXEFC], efc_aref[MAXEFC], efc_force[MAXEFC];
  int efc_type[MAXEFC], efc_id[MAXEFC], efc_state[MAXEFC];
  /* velocity-dependent */
  double cvel[CS_MAXBODY][6], cdof_dot[NV][6];
  double qfrc_passive[NV], qfrc_bias[NV], qfrc_actuator[NV], qfrc_smooth[NV], qacc_smooth[NV];
  double qfrc_constraint[NV], qacc[NV];
  double actuator_force[CS_MAXU];
  /* sensors (IMU site): framequat, gyro, velocimeter */
  double sensor_quat[4], sensor_gyro[3], sensor_vel[3];
  double cfrc_ext[CS_MAXBODY][6];
  /* solver statistics */
  int solver_niter, ls_total, bad;
  double solver_cost;
  int contact_overflow;
  int no_self_collision; /* test switch */
  int boxbox_mpr;        /* test switch: box-box pairs through MPR */
} oracle_data;

/* ------------------------------------------------------------------ small math (engine_util_*.c) */
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static double norm3(const double* a) { return sqrt(dot3(a, a)); }
static double normalize3(double* a) {
  double n = norm3(a);
  if (n < MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; return n; }
  a[0] /= n; a[1] /= n; a[2] /= n;
  return n;
}
static void normalize4(double* q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  for (int i = 0; i < 4; i++) q[i] /= n;
}
static void mul_quat(double* r, const double* a, const double* b) {
  double t[4] = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                 a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
  memcpy(r, t, sizeof t);
}
static void quat2mat(double* m, const double* q) { /* mju_quat2Mat */
  double q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3], q11 = q[1] * q[1], q12 = q[1] * q[2],
         q13 = q[1] * q[3], q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2 * (q12 - q03); m[2] = 2 * (q13 + q02); m[3] = 2 * (q12 + q03);
  m[5] = 2 * (q23 - q01); m[6] = 2 * (q13 - q02); m[7] = 2 * (q23 + q01);
}
static void rot_vec_quat(double* r, const double* v, const double* q) {
  double m[9];
  quat2mat(m, q);
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
         z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void mul_mat_vec3(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
         z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void mul_matT_vec3(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2], y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2],
         z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void axis_angle_quat(double* q, const double* axis, double angle) {
  double s = sin(angle * 0.5);
  q[0] = cos(angle * 0.5); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}

/* ------------------------------------------------------------------ mj_kinematics (engine_core_smooth.c) */
static void kinematics(oracle_data* d) {
  const cosim_model_t* m = &d->m;
  memset(d->xpos[0], 0, sizeof d->xpos[0]);
  d->xquat[0][0] = 1; d->xquat[0][1] = d->xquat[0][2] = d->xquat[0][3] = 0;
  quat2mat(d->xmat[0], d->xquat[0]);
  memcpy(d->ximat[0], d->xmat[0], sizeof d->xmat[0]);
  memset(d->xipos[0], 0, sizeof d->xipos[0]);
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parentid[b], jn = m->body_jntnum[b], ja = m->body_jntadr[b];
    double *xpos = d->xpos[b], *xquat = d->xquat[b];
    if (jn == 1 && m->jnt_type[ja] == CS_JNT_FREE) {
      int qa = m->jnt_qposadr[ja];
      memcpy(xpos, d->qpos + qa, 3 * sizeof(double));
      memcpy(xquat, d->qpos + qa + 3, 4 * sizeof(double));
      normalize4(xquat);
      memcpy(d->xanchor[ja], xpos, 3 * sizeof(double));
      memcpy(d->xaxis[ja], m->jnt_axis[ja], 3 * sizeof(double));
    } else {
      double v[3];
      mul_mat_vec3(v, d->xmat[p], m->body_pos[b]);
      for (int k = 0; k < 3; k++) xpos[k] = d->xpos[p][k] + v[k];
      mul_quat(xquat, d->xquat[p], m->body_quat[b]);
      for (int j = ja; j < ja + jn; j++) {
        double qloc[4];
        rot_vec_quat(v, m->jnt_pos[j], xquat);
        for (int k = 0; k < 3; k++) d->xanchor[j][k] = xpos[k] + v[k];
        rot_vec_quat(d->xaxis[j], m->jnt_axis[j], xquat);
        axis_angle_quat(qloc, m->jnt_axis[j], d->qpos[m->jnt_qposadr[j]] - m->qpos0[m->jnt_qposadr[j]]);
        mul_quat(xquat, xquat, qloc);
        rot_vec_quat(v, m->jnt_pos[j], xquat); /* correct for off-centre rotation */
        for (int k = 0; k < 3; k++) xpos[k] = d->xanchor[j][k] - v[k];
      }
    }
    normalize4(xquat);
    quat2mat(d->xmat[b], xquat);
    double v[3], q[4];
    mul_mat_vec3(v, d->xmat[b], m->body_ipos[b]);
    for (int k = 0; k < 3; k++) d->xipos[b][k] = xpos[k] + v[k];
    mul_quat(q, xquat, m->body_iquat[b]);
    quat2mat(d->ximat[b], q);
  }
}

/* ------------------------------------------------------------------ mj_comPos (engine_core_smooth.c) */
static void com_pos(oracle_data* d) {
  const cosim_model_t* m = &d->m;
  double mass[CS_MAXBODY];
  for (int b = 0; b < m->nbody; b++) {
    mass[b] = m->body_mass[b];
    for (int k = 0; k < 3; k++) d->subtree_com[b][k] = m->body_mass[b] * d->xipos[b][k];
  }
  for (int b = m->nbody - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    mass[p] += mass[b];
    for (int k = 0; k < 3; k++) d->subtree_com[p][k] += d->subtree_com[b][k];
  }
  for (int b = 0; b < m->nbody; b++)
    for (int k = 0; k < 3; k++)
      d->subtree_com[b][k] = mass[b] < MINVAL ? d->xipos[b][k] : d->subtree_com[b][k] / mass[b];
  /* cinert: body inertia about the subtree CoM of its kinematic-tree root, world aligned (mju_inertCom) */
  memset(d->cinert[0], 0, sizeof d->cinert[0]);
  for (int b = 1; b < m->nbody; b++) {
    const double *mat = d->ximat[b], *I = m->body_inertia[b];
    double dif[3], ms = m->body_mass[b], *res = d->cinert[b];
    for (int k = 0; k < 3; k++) dif[k] = d->xipos[b][k] - d->subtree_com[m->body_rootid[b]][k];
    double tmp[9];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) tmp[3 * r + c] = mat[3 * r + c] * I[c]; /* mat * diag(I) */
    double R[9];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) R[3 * r + c] = tmp[3 * r] * mat[3 * c] + tmp[3 * r + 1] * mat[3 * c + 1] + tmp[3 * r + 2] * mat[3 * c + 2];
    double dd = dot3(dif, dif);
    res[0] = R[0] + ms * (dd - dif[0] * dif[0]);
    res[1] = R[4] + ms * (dd - dif[1] * dif[1]);
    res[2] = R[8] + ms * (dd - dif[2] * dif[2]);
    res[3] = R[1] - ms * dif[0] * dif[1];
    res[4] = R[2] - ms * dif[0] * dif[2];
    res[5] = R[5] - ms * dif[1] * dif[2];
    res[6] = ms * dif[0]; res[7] = ms * dif[1]; res[8] = ms * dif[2];
    res[9] = ms;
  }
  /* cdof: motion axes about the same point (mju_dofCom) */
  for (int j = 0; j < m->njnt; j++) {
    int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j];
    double off[3];
    for (int k = 0; k < 3; k++) off[k] = d->subtree_com[m->body_rootid[b]][k] - d->xanchor[j][k];
    if (m->jnt_type[j] == CS_JNT_FREE) {
      for (int k = 0; k < 3; k++) {
        memset(d->cdof[da + k], 0, sizeof d->cdof[0]);
        d->cdof[da + k][3 + k] = 1;
        double ax[3] = {d->xmat[b][k], d->xmat[b][3 + k], d->xmat[b][6 + k]};
        memcpy(d->cdof[da + 3 + k], ax, sizeof ax);
        cross3(d->cdof[da + 3 + k] + 3, ax, off);
      }
    } else {
      memcpy(d->cdof[da], d->xaxis[j], 3 * sizeof(double));
      cross3(d->cdof[da] + 3, d->xaxis[j], off);
    }
  }
}

/* mju_mulInertVec: res = I * v for 10-number inertia and spatial motion vector [ang; lin] */
static void mul_inert_vec(double* res, const double* i, const double* v) {
  res[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  res[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  res[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  res[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  res[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  res[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}

/* ------------------------------------------------------------------ mj_crb + mj_factorM (dense) */
static int chol_factor(double L[NV][NV], int n) {
  for (int j = 0; j < n; j++) {
    double s = L[j][j];
    for (int k = 0; k < j; k++) s -= L[j][k] * L[j][k];
    if (s < MINVAL) s = MINVAL; /* mju_cholFactor(mat, n, mjMINVAL) */
    L[j][j] = sqrt(s);
    for (int i = j + 1; i < n; i++) {
      double t = L[i][j];
      for (int k = 0; k < j; k++) t -= L[i][k] * L[j][k];
      L[i][j] = t / L[j][j];
    }
  }
  return 0;
}
static void chol_solve(const double L[NV][NV], int n, double* x) {
  for (int i = 0; i < n; i++) {
    double s = x[i];
    for (int k = 0; k < i; k++) s -= L[i][k] * x[k];
    x[i] = s / L[i][i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = x[i];
    for (int k = i + 1; k < n; k++) s -= L[k][i] * x[k];
    x[i] = s / L[i][i];
  }
}

static void crb(oracle_data* d) {
  const cosim_model_t* m = &d->m;
  int nv = m->nv;
  memcpy(d->crb, d->cinert, sizeof d->crb);
  for (int b = m->nbody - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    if (p > 0)
      for (int k = 0; k < 10; k++) d->crb[p][k] += d->crb[b][k];
  }
  memset(d->M, 0, sizeof d->M);
  for (int i = 0; i < nv; i++) {
    double buf[6];
    mul_inert_vec(buf, d->crb[m->dof_bodyid[i]], d->cdof[i]);
    for (int j = i; j >= 0; j = m->dof_parentid[j]) {
      double v = 0;
      for (int k = 0; k < 6; k++) v += d->cdof[j][k] * buf[k];
      d->M[i][j] = d->M[j][i] = v;
    }
    d->M[i][i] += m->dof_armature[i];
  }
  memcpy(d->L, d->M, sizeof d->M);
  chol_factor(d->L, nv);
}

/* ------------------------------------------------------------------ mj_jac (translational part) */
static void jac_point(const oracle_data* d, int body, const double* point, double jacp[3][NV]) {
  const cosim_model_t* m = &d->m;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < m->nv; c++) jacp[r][c] = 0;
  if (body <= 0) return;
  double off[3];
  for (int k = 0; k < 3; k++) off[k] = point[k] - d->subtree_com[m->body_rootid[body]][k];
  while (body > 0 && m->body_dofnum[body] == 0) body = m->body_parentid[body];
  if (body <= 0) return;
  int i = m->body_dofadr[body] + m->body_dofnum[body] - 1;
  while (i >= 0) {
    double tmp[3];
    cross3(tmp, d->cdof[i], off);
    for (int r = 0; r < 3; r++) jacp[r][i] = d->cdof[i][3 + r] + tmp[r];
    i = m->dof_parentid[i];
  }
}

/* ------------------------------------------------------------------ collision (ground vs robot geoms) */
static void make_frame(double* frame) { /* mju_makeFrame */
  double *x = frame, *y = frame + 3, *z = frame + 6;
  normalize3(x);
  if (norm3(y) < 0.5) {
    y[0] = y[1] = y[2] = 0;
    if (x[1] < 0.5 && x[1] > -0.5) y[1] = 1; else y[2] = 1;
  }
  double dt = dot3(x, y);
  for (int k = 0; k < 3; k++) y[k] -= dt * x[k];
  normalize3(y);
  cross3(z, x, y);
}

static contact_t* add_contact(oracle_data* d, int geom, double dist, const double* pos, const double* normal) {
  if (d->ncon >= MAXCON) { d->contact_overflow++; return NULL; }
  contact_t* c = &d->con[d->ncon++];
  memset(c, 0, sizeof *c);
  c->geom = geom; c->body = d->m.geom_bodyid[geom]; c->dist = dist;
  c->geom1 = -1; c->body1 = 0;
  memcpy(c->pos, pos, 3 * sizeof(double));
  memcpy(c->frame, normal, 3 * sizeof(double));
  make_frame(c->frame);
  return c;
}

/* geom world pose */
static void geom_pose(const oracle_data* d, int g, double* pos, double* mat) {
  const cosim_model_t* m = &d->m;
  int b = m->geom_bodyid[g];
  double v[3], q[4];
  mul_mat_vec3(v, d->xmat[b], m->geom_pos[g]);
  for (int k = 0; k < 3; k++) pos[k] = d->xpos[b][k] + v[k];
  mul_quat(q, d->xquat[b], m->geom_quat[g]);
  quat2mat(mat, q);
}

/* engine_collision_primitive.c: mjc_PlaneSphere / mjc_PlaneCylinder / mjc_PlaneBox, engine_collision_convex.c: mjc_PlaneConvex */
static void plane_sphere(oracle_data* d, int g, const double* ppos, const double* n, double margin) {
  double pos[3], mat[9], dif[3];
  geom_pose(d, g, pos, mat);
  for (int k = 0; k < 3; k++) dif[k] = pos[k] - ppos[k];
  double r = d->m.geom_size[g][0], cdist = dot3(dif, n);
  if (cdist > margin + r) return;
  double dist = cdist - r, cp[3];
  for (int k = 0; k < 3; k++) cp[k] = pos[k] - n[k] * (r + 0.5 * dist);
  add_contact(d, g, dist, cp, n);
}

static void plane_cylinder(oracle_data* d, int g, const double* ppos, const double* n, double margin) {
  double pos[3], mat[9], dif[3];
  geom_pose(d, g, pos, mat);
  double radius = d->m.geom_size[g][0], half = d->m.geom_size[g][1];
  double axis[3] = {mat[2], mat[5], mat[8]};
  double prjaxis = dot3(n, axis);
  if (prjaxis > 0) { for (int k = 0; k < 3; k++) axis[k] = -axis[k]; prjaxis = -prjaxis; }
  for (int k = 0; k < 3; k++) dif[k] = pos[k] - ppos[k];
  double dist0 = dot3(dif, n);
  double vec[3];
  for (int k = 0; k < 3; k++) vec[k] = axis[k] * prjaxis - n[k];
  double len = norm3(vec);
  if (len < MINVAL) { /* disk parallel to plane: pick x-axis of the cylinder */
    vec[0] = mat[0] * radius; vec[1] = mat[3] * radius; vec[2] = mat[6] * radius;
  } else
    for (int k = 0; k < 3; k++) vec[k] *= radius / len;
  double prjvec = dot3(vec, n);
  for (int k = 0; k < 3; k++) axis[k] *= half;
  prjaxis *= half;
  int cnt = 0;
  double cp[3], dist;
  /* first point: deepest point of the near disk */
  dist = dist0 + prjaxis + prjvec;
  if (dist > margin) return;
  for (int k = 0; k < 3; k++) cp[k] = pos[k] + vec[k] + axis[k] - n[k] * dist * 0.5;
  add_contact(d, g, dist, cp, n); cnt++;
  /* second point: same rim point of the far disk */
  dist = dist0 - prjaxis + prjvec;
  if (dist <= margin) {
    for (int k = 0; k < 3; k++) cp[k] = pos[k] + vec[k] - axis[k] - n[k] * dist * 0.5;
    add_contact(d, g, dist, cp, n); cnt++;
  }
  /* two more points on the near disk, 120 degrees to either side */
  double vec1[3];
  cross3(vec1, vec, axis);
  normalize3(vec1);
  for (int k = 0; k < 3; k++) vec1[k] *= radius * sqrt(3.0) * 0.5;
  double prjvec1 = dot3(vec1, n);
  dist = dist0 + prjaxis - prjvec * 0.5 + prjvec1;
  if (dist <= margin) {
    for (int k = 0; k < 3; k++) cp[k] = pos[k] + vec1[k] + axis[k] - vec[k] * 0.5 - n[k] * dist * 0.5;
    add_contact(d, g, dist, cp, n); cnt++;
  }
  dist = dist0 + prjaxis - prjvec * 0.5 - prjvec1;
  if (dist <= margin) {
    for (int k = 0; k < 3; k++) cp[k] = pos[k] - vec1[k] + axis[k] - vec[k] * 0.5 - n[k] * dist * 0.5;
    add_contact(d, g, dist, cp, n); cnt++;
  }
}

static void plane_box(oracle_data* d, int g, const double* ppos, const double* n, double margin) {
  double pos[3], mat[9], dif[3];
  geom_pose(d, g, pos, mat);
  const double* size = d->m.geom_size[g];
  for (int k = 0; k < 3; k++) dif[k] = pos[k] - ppos[k];
  double dist0 = dot3(dif, n);
  int cnt = 0;
  for (int i = 0; i < 8 && cnt < 4; i++) { /* mjc_PlaneBox: corners in order, at most 4 contacts */
    double vec[3] = {(i & 1 ? size[0] : -size[0]), (i & 2 ? size[1] : -size[1]), (i & 4 ? size[2] : -size[2])};
    double corner[3];
    mul_mat_vec3(corner, mat, vec);
    double ldist = dot3(n, corner);
    if (dist0 + ldist > margin || ldist > 0) continue;
    double dist = dist0 + ldist, cp[3];
    for (int k = 0; k < 3; k++) cp[k] = pos[k] + corner[k] - n[k] * dist * 0.5;
    add_contact(d, g, dist, cp, n); cnt++;
  }
}

static void plane_mesh(oracle_data* d, int g, const double* ppos, const double* n, double margin) {
  const cosim_model_t* m = &d->m;
  int b = m->geom_bodyid[g], adr = m->geom_hulladr[g], num = m->geom_hullnum[g];
  const double *xpos = d->xpos[b], *xmat = d->xmat[b];
  double ln[3]; /* plane normal in the body frame: hull vertices are stored in body coordinates */
  mul_matT_vec3(ln, xmat, n);
  double off = 0;
  for (int k = 0; k < 3; k++) off += (xpos[k] - ppos[k]) * n[k];
  int best = -1;
  double bestd = 0;
  for (int i = 0; i < num; i++) { /* support point in -normal direction (mjc_support over the hull) */
    const float* v = d->hull_vert + 3 * (adr + i);
    double dist = off + ln[0] * v[0] + ln[1] * v[1] + ln[2] * v[2];
    if (best < 0 || dist < bestd) { best = i; bestd = dist; }
  }
  if (best < 0 || bestd > margin) return;
  int cnt = 0;
  for (int pass = 0; pass < 2; pass++) {
    int lo = pass ? d->hull_adr[adr + best] : 0, hi = pass ? d->hull_adr[adr + best + 1] : 1;
    for (int e = lo; e < hi && cnt < 4; e++) { /* then up to 3 hull neighbours of the support vertex within margin */
      int i = pass ? d->hull_nbr[e] : best;
      const float* v = d->hull_vert + 3 * (adr + i);
      double dist = off + ln[0] * v[0] + ln[1] * v[1] + ln[2] * v[2];
      if (dist > margin) continue;
      double lv[3] = {v[0], v[1], v[2]}, w[3], cp[3];
      mul_mat_vec3(w, xmat, lv);
      for (int k = 0; k < 3; k++) cp[k] = xpos[k] + w[k] - n[k] * dist * 0.5;
      add_contact(d, g, dist, cp, n); cnt++;
    }
  }
}


/* ------------------------------------------------------------------ convex-convex narrowphase: MPR
 * MuJoCo 3.2.7 sends every geom pair without an analytic routine (here: mesh, cylinder, box pairs, and hfield prisms)
 * through libccd's ccdMPRPenetration (engine_collision_convex.c mjc_Convex / mjc_ConvexHField -> mjc_MPRIteration;
 * third-party dependency libccd 2.1, src/mpr.c, absent from /root/reference).  Restated from the published algorithm
 * (G. Snethen, "XenoCollide", Game Programming Gems 7) following libccd's function structure: discoverPortal /
 * refinePortal / findPenetr / findPos, its CCD_EPS = DBL_EPSILON predicates, mpr_tolerance = 1e-6 and
 * max_iterations = 50 (MuJoCo's ccd_tolerance / ccd_iterations defaults).  PARITY UNPINNED (no libccd here). */
#define CCD_EPS 2.220446049250313e-16
#define MPR_TOL 1e-6
#define MPR_MAXIT 50
#define MPR_REFINE_CAP 1000 /* libccd's refinePortal loop has no cap; this one is never reached in fp64 */

enum { CO_SPHERE = 0, CO_CYLINDER, CO_BOX, CO_MESH, CO_PRISM };
typedef struct {
  int kind;
  double pos[3], mat[9], size[3]; /* primitive: geom world pose; mesh: *body* pose (hull vertices are in body coordinates) */
  const float* vert; int nvert;
  double prism[6][3];             /* 0..2 bottom, 3..5 top */
  double center[3];
} cobj_t;
typedef struct { double v[3], v1[3], v2[3]; } csup_t;

static double sign0(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : 0.0); } /* mju_sign */
static int ccd_zero(double x) { return fabs(x) < CCD_EPS; }
static int ccd_eq(double a, double b) {
  double ab = fabs(a - b);
  if (ab < CCD_EPS) return 1;
  a = fabs(a); b = fabs(b);
  return b > a ? ab < CCD_EPS * b : ab < CCD_EPS * a;
}
static int vec_eq0(const double* v) { return ccd_eq(v[0], 0) && ccd_eq(v[1], 0) && ccd_eq(v[2], 0); }

/* mjccd_support (engine_collision_convex.c): furthest point of the geom along the unit world direction dir */
static void co_support(const cobj_t* o, const double* dir, double* out) {
  double l[3], r[3] = {0, 0, 0};
  if (o->kind == CO_PRISM) { /* prism support: bottom triangle for dir z < 0, else top */
    int i0 = dir[2] < 0 ? 0 : 3, best = i0;
    double bd = dot3(o->prism[i0], dir);
    for (int i = i0 + 1; i < i0 + 3; i++) { double t = dot3(o->prism[i], dir); if (t > bd) { bd = t; best = i; } }
    memcpy(out, o->prism[best], 3 * sizeof(double));
    return;
  }
  mul_matT_vec3(l, o->mat, dir);
  switch (o->kind) {
    case CO_SPHERE: for (int k = 0; k < 3; k++) r[k] = l[k] * o->size[0]; break;
    case CO_CYLINDER: {
      double t = sqrt(l[0] * l[0] + l[1] * l[1]);
      if (t > MINVAL) { r[0] = l[0] / t * o->size[0]; r[1] = l[1] / t * o->size[0]; }
      r[2] = sign0(l[2]) * o->size[1];
    } break;
    case CO_BOX: for (int k = 0; k < 3; k++) r[k] = sign0(l[k]) * o->size[k]; break;
    default: { /* mesh: hull vertex with the largest projection (first one on ties) */
      int best = 0; double bd = -1e300;
      for (int i = 0; i < o->nvert; i++) {
        const float* v = o->vert + 3 * i;
        double t = l[0] * v[0] + l[1] * v[1] + l[2] * v[2];
        if (t > bd) { bd = t; best = i; }
      }
      for (int k = 0; k < 3; k++) r[k] = o->vert[3 * best + k];
    }
  }
  mul_mat_vec3(out, o->mat, r);
  for (int k = 0; k < 3; k++) out[k] += o->pos[k];
}

long g_mpr_calls = 0, g_mpr_supports = 0, g_mpr_hits = 0; /* diagnostics */
static void mpr_support(const cobj_t* a, const cobj_t* b, const double* dir, csup_t* s) { /* __ccdSupport */
  g_mpr_supports++;
  double nd[3] = {-dir[0], -dir[1], -dir[2]};
  co_support(a, dir, s->v1);
  co_support(b, nd, s->v2);
  for (int k = 0; k < 3; k++) s->v[k] = s->v1[k] - s->v2[k];
}
static void portal_dir(const csup_t* P, double* dir) {
  double a[3], b[3];
  for (int k = 0; k < 3; k++) { a[k] = P[2].v[k] - P[1].v[k]; b[k] = P[3].v[k] - P[1].v[k]; }
  cross3(dir, a, b);
  normalize3(dir);
}
static int portal_reach_tol(const csup_t* P, const csup_t* v4, const double* dir) {
  double dv1 = dot3(P[1].v, dir), dv2 = dot3(P[2].v, dir), dv3 = dot3(P[3].v, dir), dv4 = dot3(v4->v, dir);
  double d1 = dv4 - dv1, d2 = dv4 - dv2, d3 = dv4 - dv3;
  d1 = fmin(d1, fmin(d2, d3));
  return ccd_eq(d1, MPR_TOL) || d1 < MPR_TOL;
}
static void expand_portal(csup_t* P, const csup_t* v4) {
  double v4v0[3];
  cross3(v4v0, v4->v, P[0].v);
  if (dot3(P[1].v, v4v0) > 0) {
    if (dot3(P[2].v, v4v0) > 0) P[1] = *v4; else P[3] = *v4;
  } else {
    if (dot3(P[3].v, v4v0) > 0) P[2] = *v4; else P[1] = *v4;
  }
}
static double point_seg_dist2(const double* x0, const double* b, double* wit) { /* ccdVec3PointSegmentDist2 with P = origin */
  double dd[3], t;
  for (int k = 0; k < 3; k++) dd[k] = b[k] - x0[k];
  t = -dot3(x0, dd) / dot3(dd, dd);
  if (t < 0 || ccd_zero(t)) { memcpy(wit, x0, 24); return dot3(x0, x0); }
  if (t > 1 || ccd_eq(t, 1)) { memcpy(wit, b, 24); return dot3(b, b); }
  for (int k = 0; k < 3; k++) wit[k] = x0[k] + t * dd[k];
  return dot3(wit, wit);
}
static double point_tri_dist2(const double* x0, const double* B, const double* C, double* wit) { /* ccdVec3PointTriDist2, P = origin */
  double d1[3], d2[3];
  for (int k = 0; k < 3; k++) { d1[k] = B[k] - x0[k]; d2[k] = C[k] - x0[k]; }
  double v = dot3(d1, d1), w = dot3(d2, d2), p = dot3(x0, d1), q = dot3(x0, d2), r = dot3(d1, d2);
  double s = (q * r - w * p) / (w * v - r * r), t = (-s * r - q) / w;
  if ((ccd_zero(s) || s > 0) && (ccd_eq(s, 1) || s < 1) && (ccd_zero(t) || t > 0) && (ccd_eq(t, 1) || t < 1) &&
      (ccd_eq(t + s, 1) || t + s < 1)) {
    for (int k = 0; k < 3; k++) wit[k] = x0[k] + s * d1[k] + t * d2[k];
    return dot3(wit, wit);
  }
  double w2[3], dist = point_seg_dist2(x0, B, wit), dist2 = point_seg_dist2(x0, C, w2);
  if (dist2 < dist) { dist = dist2; memcpy(wit, w2, 24); }
  dist2 = point_seg_dist2(B, C, w2);
  if (dist2 < dist) { dist = dist2; memcpy(wit, w2, 24); }
  return dist;
}
static void mpr_find_pos(const csup_t* P, double* pos) {
  double dir[3], vec[3], b[4], sum;
  portal_dir(P, dir);
  cross3(vec, P[1].v, P[2].v); b[0] = dot3(vec, P[3].v);
  cross3(vec, P[3].v, P[2].v); b[1] = dot3(vec, P[0].v);
  cross3(vec, P[0].v, P[1].v); b[2] = dot3(vec, P[3].v);
  cross3(vec, P[2].v, P[1].v); b[3] = dot3(vec, P[0].v);
  sum = b[0] + b[1] + b[2] + b[3];
  if (ccd_zero(sum) || sum < 0) {
    b[0] = 0;
    cross3(vec, P[2].v, P[3].v); b[1] = dot3(vec, dir);
    cross3(vec, P[3].v, P[1].v); b[2] = dot3(vec, dir);
    cross3(vec, P[1].v, P[2].v); b[3] = dot3(vec, dir);
    sum = b[1] + b[2] + b[3];
  }
  double inv = 1.0 / sum;
  for (int k = 0; k < 3; k++) {
    double p1 = 0, p2 = 0;
    for (int i = 0; i < 4; i++) { p1 += b[i] * P[i].v1[k]; p2 += b[i] * P[i].v2[k]; }
    pos[k] = 0.5 * (p1 + p2) * inv;
  }
}

/* ccdMPRPenetration: 0 = penetrating (depth, dir obj1 -> obj2, pos filled), -1 = separated */
static int mpr_penetration(const cobj_t* o1, const cobj_t* o2, double* depth, double* dir_out, double* pos) {
  csup_t P[4], v4;
  g_mpr_calls++;
  double dir[3], va[3], vb[3], dot;
  /* --- discoverPortal */
  for (int k = 0; k < 3; k++) { P[0].v1[k] = o1->center[k]; P[0].v2[k] = o2->center[k]; P[0].v[k] = P[0].v1[k] - P[0].v2[k]; }
  if (vec_eq0(P[0].v)) P[0].v[0] += CCD_EPS * 10;
  for (int k = 0; k < 3; k++) dir[k] = -P[0].v[k];
  normalize3(dir);
  mpr_support(o1, o2, dir, &P[1]);
  dot = dot3(P[1].v, dir);
  if (ccd_zero(dot) || dot < 0) return -1;
  cross3(dir, P[0].v, P[1].v);
  if (ccd_zero(dot3(dir, dir))) {
    if (vec_eq0(P[1].v)) return -1; /* findPenetrTouch: depth 0, dir 0 -> MuJoCo drops the contact (mjc_MPRIteration) */
    /* findPenetrSegment: origin on the v0-v1 segment */
    for (int k = 0; k < 3; k++) { pos[k] = 0.5 * (P[1].v1[k] + P[1].v2[k]); dir_out[k] = P[1].v[k]; }
    *depth = normalize3(dir_out);
    return 0;
  }
  normalize3(dir);
  mpr_support(o1, o2, dir, &P[2]);
  dot = dot3(P[2].v, dir);
  if (ccd_zero(dot) || dot < 0) return -1;
  for (int k = 0; k < 3; k++) { va[k] = P[1].v[k] - P[0].v[k]; vb[k] = P[2].v[k] - P[0].v[k]; }
  cross3(dir, va, vb);
  normalize3(dir);
  if (dot3(dir, P[0].v) > 0) { csup_t t = P[1]; P[1] = P[2]; P[2] = t; for (int k = 0; k < 3; k++) dir[k] = -dir[k]; }
  for (int it = 0;; it++) {
    if (it > MPR_REFINE_CAP) return -1;
    mpr_support(o1, o2, dir, &P[3]);
    dot = dot3(P[3].v, dir);
    if (ccd_zero(dot) || dot < 0) return -1;
    int cont = 0;
    cross3(va, P[1].v, P[3].v);
    dot = dot3(va, P[0].v);
    if (dot < 0 && !ccd_zero(dot)) { P[2] = P[3]; cont = 1; }
    if (!cont) {
      cross3(va, P[3].v, P[2].v);
      dot = dot3(va, P[0].v);
      if (dot < 0 && !ccd_zero(dot)) { P[1] = P[3]; cont = 1; }
    }
    if (!cont) break;
    for (int k = 0; k < 3; k++) { va[k] = P[1].v[k] - P[0].v[k]; vb[k] = P[2].v[k] - P[0].v[k]; }
    cross3(dir, va, vb);
    normalize3(dir);
  }
  /* --- refinePortal */
  for (int it = 0;; it++) {
    if (it > MPR_REFINE_CAP) return -1;
    portal_dir(P, dir);
    dot = dot3(dir, P[1].v);
    if (ccd_zero(dot) || dot > 0) break; /* portalEncapsulesOrigin */
    mpr_support(o1, o2, dir, &v4);
    dot = dot3(v4.v, dir);
    if (!(ccd_zero(dot) || dot > 0) || portal_reach_tol(P, &v4, dir)) return -1; /* !portalCanEncapsuleOrigin || tolerance */
    expand_portal(P, &v4);
  }
  /* --- findPenetr */
  for (int it = 0;; it++) {
    portal_dir(P, dir);
    mpr_support(o1, o2, dir, &v4);
    if (portal_reach_tol(P, &v4, dir) || it > MPR_MAXIT) {
      double pd[3];
      *depth = sqrt(point_tri_dist2(P[1].v, P[2].v, P[3].v, pd));
      if (ccd_zero(pd[0]) && ccd_zero(pd[1]) && ccd_zero(pd[2])) memcpy(pd, dir, 24);
      normalize3(pd);
      memcpy(dir_out, pd, 24);
      mpr_find_pos(P, pos);
      return 0;
    }
    expand_portal(P, &v4);
  }
}

static void make_cobj(const oracle_data* d, int g, cobj_t* o) {
  const cosim_model_t* m = &d->m;
  int b = m->geom_bodyid[g];
  double v[3];
  memset(o, 0, sizeof *o);
  mul_mat_vec3(v, d->xmat[b], m->geom_center[g]);
  for (int k = 0; k < 3; k++) o->center[k] = d->xpos[b][k] + v[k];
  memcpy(o->size, m->geom_size[g], 3 * sizeof(double));
  switch (m->geom_type[g]) {
    case CS_GEOM_SPHERE: o->kind = CO_SPHERE; break;
    case CS_GEOM_CYLINDER: o->kind = CO_CYLINDER; break;
    case CS_GEOM_BOX: o->kind = CO_BOX; break;
    default: o->kind = CO_MESH;
  }
  if (o->kind == CO_MESH) {
    memcpy(o->pos, d->xpos[b], 24); memcpy(o->mat, d->xmat[b], 72);
    o->vert = d->hull_vert + 3 * m->geom_hulladr[g]; o->nvert = m->geom_hullnum[g];
  } else
    geom_pose(d, g, o->pos, o->mat);
}

/* mjc_Convex (engine_collision_convex.c) for one robot-robot pair: a single MPR contact, normal geom1 -> geom2 */
static void convex_convex(oracle_data* d, int g1, int g2) {
  const cosim_model_t* m = &d->m;
  int b1 = m->geom_bodyid[g1], b2 = m->geom_bodyid[g2];
  double c1[3], c2[3], v[3];
  mul_mat_vec3(v, d->xmat[b1], m->geom_rcenter[g1]);
  for (int k = 0; k < 3; k++) c1[k] = d->xpos[b1][k] + v[k];
  mul_mat_vec3(v, d->xmat[b2], m->geom_rcenter[g2]);
  for (int k = 0; k < 3; k++) c2[k] = d->xpos[b2][k] + v[k] - c1[k];
  double margin = fmax(m->geom_margin[g1], m->geom_margin[g2]), rs = m->geom_rbound[g1] + m->geom_rbound[g2] + margin;
  if (dot3(c2, c2) > rs * rs) return; /* bounding-sphere filter (mj_collideGeoms broadphase: conservative, no effect on results) */
  cobj_t o1, o2;
  make_cobj(d, g1, &o1);
  make_cobj(d, g2, &o2);
  double depth, dir[3], pos[3];
  if (mpr_penetration(&o1, &o2, &depth, dir, pos) != 0) return;
  if (vec_eq0(dir)) return;
  contact_t* c = add_contact(d, g2, margin - depth, pos, dir);
  if (c) { c->geom1 = g1; c->body1 = b1; }
}

/* ------------------------------------------------------------------ box-box (engine_collision_box.c mjc_BoxBox)
 * MuJoCo's collision table sends box-box pairs to mjc_BoxBox, not to MPR: a separating-axis search over the 15 axes (6 face
 * normals, 9 edge x edge directions; an edge axis wins only when it beats the best face by more than a 1e-12 relative bias), then
 *   - face case: the face of the other box most opposed to the winning face (the incident face) is clipped against the winning
 *     face's rectangle; every vertex of the clipped polygon (at most 8) within `margin` of the reference face is a contact,
 *     position midway between the incident-face point and the reference face, normal = the face normal (geom1 -> geom2);
 *   - edge case: one contact midway between the closest points of the two edges.
 * RESTATED FROM MEMORY OF THE ALGORITHM'S STRUCTURE (PARITY UNPINNED, like MPR above): the enumeration order of the clipped
 * polygon's vertices -- incident vertices inside the rectangle, rectangle corners inside the incident quad, edge x side
 * crossings -- is this file's (contact order does not change the Newton solution), and MuJoCo's edge case can emit more than one
 * point for parallel edges; here parallel edges never win (their cross product is skipped) and fall to the face case.
 * Humanoid only: humanoid_p_v0.xml:33,40,110,139 (torso / pelvis / forearm boxes), five pairs after the compiled filters. */
typedef struct { double pos[3], dist; } bb_point_t;
static void mat_col(double* c, const double* mat, int k) { c[0] = mat[k]; c[1] = mat[3 + k]; c[2] = mat[6 + k]; }
static int box_box_points(const double* p1, const double* m1, const double* s1, const double* p2, const double* m2, const double* s2,
                          double margin, bb_point_t* out, double* nrm) {
  double d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, A[3][3], B[3][3];
  for (int k = 0; k < 3; k++) { mat_col(A[k], m1, k); mat_col(B[k], m2, k); }
  double best = -1e300, bax[3] = {0, 0, 1};
  int code = -1;
  for (int i = 0; i < 3; i++) { /* faces of box 1, then of box 2: separation s = |d.a| - r1 - r2 (negative: penetration) */
    double r = s1[i];
    for (int k = 0; k < 3; k++) r += s2[k] * fabs(dot3(B[k], A[i]));
    double s = fabs(dot3(d, A[i])) - r;
    if (s > margin) return 0;
    if (s > best) { best = s; code = i; }
  }
  for (int j = 0; j < 3; j++) {
    double r = s2[j];
    for (int k = 0; k < 3; k++) r += s1[k] * fabs(dot3(A[k], B[j]));
    double s = fabs(dot3(d, B[j])) - r;
    if (s > margin) return 0;
    if (s > best) { best = s; code = 3 + j; }
  }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double ax[3];
      cross3(ax, A[i], B[j]);
      double l = sqrt(dot3(ax, ax));
      if (l < 1e-6) continue; /* parallel edges: the face axes decide */
      for (int k = 0; k < 3; k++) ax[k] /= l;
      double r = 0;
      for (int k = 0; k < 3; k++) r += s1[k] * fabs(dot3(A[k], ax)) + s2[k] * fabs(dot3(B[k], ax));
      double s = fabs(dot3(d, ax)) - r;
      if (s > margin) return 0;
      /* penetration_edge < penetration_face (1 - 1e-12), in separations (both negative when penetrating) */
      if (s > best + 1e-12 * fabs(best)) { best = s; code = 6 + 3 * i + j; memcpy(bax, ax, 24); }
    }
  if (code < 0) return 0;
  if (code >= 6) {
    int i = (code - 6) / 3, j = (code - 6) % 3;
    double sg = dot3(bax, d) < 0 ? -1.0 : 1.0, pa[3], pb[3];
    for (int k = 0; k < 3; k++) { nrm[k] = sg * bax[k]; pa[k] = p1[k]; pb[k] = p2[k]; }
    for (int k = 0; k < 3; k++) {
      if (k != i) { double t = dot3(nrm, A[k]) < 0 ? -s1[k] : s1[k]; for (int c = 0; c < 3; c++) pa[c] += t * A[k][c]; }
      if (k != j) { double t = dot3(nrm, B[k]) < 0 ? -s2[k] : s2[k]; for (int c = 0; c < 3; c++) pb[c] -= t * B[k][c]; }
    }
    double pp[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
    double uaub = dot3(A[i], B[j]), q1 = dot3(A[i], pp), q2 = -dot3(B[j], pp), den = 1 - uaub * uaub;
    double al = (q1 + uaub * q2) / den, be = (uaub * q1 + q2) / den;
    for (int k = 0; k < 3; k++) out[0].pos[k] = 0.5 * (pa[k] + al * A[i][k] + pb[k] + be * B[j][k]);
    out[0].dist = best;
    return 1;
  }
  /* face case: reference box R (the face's owner), incident box O */
  int ref2 = code >= 3, a = code % 3, a1 = (a + 1) % 3, a2 = (a + 2) % 3;
  const double *pr = ref2 ? p2 : p1, *po = ref2 ? p1 : p2, *sr = ref2 ? s2 : s1, *so = ref2 ? s1 : s2;
  double(*Rr)[3] = ref2 ? B : A, (*Ro)[3] = ref2 ? A : B;
  double dro[3] = {po[0] - pr[0], po[1] - pr[1], po[2] - pr[2]}, nr[3];
  double sg = dot3(dro, Rr[a]) < 0 ? -1.0 : 1.0;
  for (int k = 0; k < 3; k++) { nr[k] = sg * Rr[a][k]; nrm[k] = ref2 ? -nr[k] : nr[k]; }
  int b = 0;
  double bm = -1;
  for (int k = 0; k < 3; k++) { double v = fabs(dot3(Ro[k], nr)); if (v > bm) { bm = v; b = k; } }
  int k1 = (b + 1) % 3, k2 = (b + 2) % 3;
  double mo[3], fc[3], sb = dot3(Ro[b], nr) < 0 ? 1.0 : -1.0; /* the face of O that looks back at R */
  for (int k = 0; k < 3; k++) { mo[k] = sb * Ro[b][k]; fc[k] = po[k] + sb * so[b] * Ro[b][k] - pr[k]; } /* relative to R's centre */
  static const double SQ[4][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}};
  double v[4][3], vu[4], vw[4];
  for (int q = 0; q < 4; q++) {
    for (int k = 0; k < 3; k++) v[q][k] = fc[k] + SQ[q][0] * so[k1] * Ro[k1][k] + SQ[q][1] * so[k2] * Ro[k2][k];
    vu[q] = dot3(v[q], Rr[a1]); vw[q] = dot3(v[q], Rr[a2]);
  }
  double h1 = sr[a1], h2 = sr[a2], mn = dot3(mo, nr);
  int n = 0;
#define BB_EMIT(X)                                                                                                     \
  do {                                                                                                                 \
    double depth_ = sr[a] - dot3((X), nr);                                                                             \
    if (-depth_ <= margin && n < 8) {                                                                                  \
      for (int k_ = 0; k_ < 3; k_++) out[n].pos[k_] = pr[k_] + (X)[k_] + 0.5 * depth_ * nr[k_];                        \
      out[n++].dist = -depth_;                                                                                         \
    }                                                                                                                  \
  } while (0)
  for (int q = 0; q < 4; q++) /* (A) incident vertices inside the rectangle */
    if (fabs(vu[q]) <= h1 && fabs(vw[q]) <= h2) BB_EMIT(v[q]);
  for (int q = 0; q < 4; q++) { /* (B) rectangle corners strictly inside the projected incident quad, lifted onto the incident face */
    double cu = SQ[q][0] * h1, cw = SQ[q][1] * h2;
    int pos_ = 0, neg_ = 0;
    for (int e = 0; e < 4; e++) {
      int f = (e + 1) & 3;
      double cr = (vu[f] - vu[e]) * (cw - vw[e]) - (vw[f] - vw[e]) * (cu - vu[e]);
      if (cr > 0) pos_++; else if (cr < 0) neg_++; else { pos_ = neg_ = 1; }
    }
    if (pos_ && neg_) continue;
    double x[3], num = 0;
    for (int k = 0; k < 3; k++) { x[k] = cu * Rr[a1][k] + cw * Rr[a2][k]; num += mo[k] * (fc[k] - x[k]); }
    double z = num / mn;
    for (int k = 0; k < 3; k++) x[k] += z * nr[k];
    BB_EMIT(x);
  }
  for (int q = 0; q < 4; q++) { /* (C) incident edge q -> q+1 against the four sides u = +h1, u = -h1, w = +h2, w = -h2 */
    int f = (q + 1) & 3;
    for (int e = 0; e < 4; e++) {
      double c0 = e < 2 ? vu[q] : vw[q], c1 = e < 2 ? vu[f] : vw[f], lim = (e & 1 ? -1.0 : 1.0) * (e < 2 ? h1 : h2);
      double o0 = e < 2 ? vw[q] : vu[q], o1 = e < 2 ? vw[f] : vu[f], ho = e < 2 ? h2 : h1;
      if ((c0 - lim) * (c1 - lim) >= 0) continue; /* strictly across the side's line */
      double t = (lim - c0) / (c1 - c0), oo = o0 + t * (o1 - o0);
      if (fabs(oo) >= ho) continue;                /* corners of the rectangle belong to (B) */
      double x[3];
      for (int k = 0; k < 3; k++) x[k] = v[q][k] + t * (v[f][k] - v[q][k]);
      BB_EMIT(x);
    }
  }
#undef BB_EMIT
  return n;
}

static void box_box(oracle_data* d, int g1, int g2) {
  const cosim_model_t* m = &d->m;
  double p1[3], m1[9], p2[3], m2[9], nrm[3];
  geom_pose(d, g1, p1, m1);
  geom_pose(d, g2, p2, m2);
  double margin = fmax(m->geom_margin[g1], m->geom_margin[g2]);
  bb_point_t pt[8];
  int n = box_box_points(p1, m1, m->geom_size[g1], p2, m2, m->geom_size[g2], margin, pt, nrm);
  for (int i = 0; i < n; i++) {
    contact_t* c = add_contact(d, g2, pt[i].dist, pt[i].pos, nrm);
    if (c) { c->geom1 = g1; c->body1 = m->geom_bodyid[g1]; }
  }
}

/* mj_contactParam (engine_collision_driver.c): equal priority -> solmix weighting, friction = elementwise max */
static void contact_param(const oracle_data* d, contact_t* c) {
  const cosim_model_t* m = &d->m;
  int g = c->geom, h = c->geom1;
  /* geom1 parameters: the ground's when h < 0 */
  double s1 = h < 0 ? m->ground_solmix : m->geom_solmix[h], s2 = m->geom_solmix[g];
  int cd1 = h < 0 ? m->ground_condim : m->geom_condim[h];
  const double* sr1 = h < 0 ? m->ground_solref : m->geom_solref[h];
  const double* si1 = h < 0 ? m->ground_solimp : m->geom_solimp[h];
  const double* fr1 = h < 0 ? m->ground_friction : m->geom_friction[h];
  double mg1 = h < 0 ? m->ground_margin : m->geom_margin[h], gp1 = h < 0 ? m->ground_gap : m->geom_gap[h];
  double mix;
  if (s1 >= MINVAL && s2 >= MINVAL) mix = s1 / (s1 + s2);
  else if (s1 < MINVAL && s2 < MINVAL) mix = 0.5;
  else mix = s1 < MINVAL ? 0.0 : 1.0;
  c->dim = cd1 > m->geom_condim[g] ? cd1 : m->geom_condim[g];
  if (sr1[0] > 0 && m->geom_solref[g][0] > 0)
    for (int k = 0; k < 2; k++) c->solref[k] = mix * sr1[k] + (1 - mix) * m->geom_solref[g][k];
  else
    for (int k = 0; k < 2; k++) c->solref[k] = fmin(sr1[k], m->geom_solref[g][k]);
  for (int k = 0; k < 5; k++) c->solimp[k] = mix * si1[k] + (1 - mix) * m->geom_solimp[g][k];
  double fr[3];
  for (int k = 0; k < 3; k++) fr[k] = fmax(fr1[k], m->geom_friction[g][k]);
  c->friction[0] = c->friction[1] = fmax(MINMU, fr[0]);
  c->friction[2] = fmax(MINMU, fr[1]);
  c->friction[3] = c->friction[4] = fmax(MINMU, fr[2]);
  double margin = fmax(mg1, m->geom_margin[g]), gap = fmax(gp1, m->geom_gap[g]);
  c->includemargin = margin - gap;
}

/* hfield support (mjc_ConvexHField restated for the contact shapes used; see hfield section below) */
static void hfield_collide(oracle_data* d, int g);

static void collision(oracle_data* d) {
  const cosim_model_t* m = &d->m;
  d->ncon = 0;
  double n[3] = {0, 0, 1}; /* ground geom has identity orientation in all four models */
  for (int g = 0; g < m->ngeom; g++) {
    if (!m->geom_ground[g]) continue;
    int first = d->ncon;
    double margin = fmax(m->ground_margin, m->geom_margin[g]);
    if (m->ground_type == CS_GEOM_PLANE) {
      switch (m->geom_type[g]) {
        case CS_GEOM_SPHERE: plane_sphere(d, g, m->ground_pos, n, margin); break;
        case CS_GEOM_CYLINDER: plane_cylinder(d, g, m->ground_pos, n, margin); break;
        case CS_GEOM_BOX: plane_box(d, g, m->ground_pos, n, margin); break;
        case CS_GEOM_MESH: plane_mesh(d, g, m->ground_pos, n, margin); break;
        default: break;
      }
    } else
      hfield_collide(d, g);
    for (int i = first; i < d->ncon; i++) contact_param(d, &d->con[i]);
  }
  /* robot-robot pairs that pass the contype/conaffinity, same-body, parent-child and <exclude> filters (compiled list) */
  for (int p = 0; p < m->npair && !d->no_self_collision; p++) {
    int first = d->ncon;
    int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p];
    if (m->geom_type[g1] == CS_GEOM_BOX && m->geom_type[g2] == CS_GEOM_BOX && !d->boxbox_mpr) box_box(d, g1, g2);
    else convex_convex(d, g1, g2);
    for (int i = first; i < d->ncon; i++) contact_param(d, &d->con[i]);
  }
}

/* ------------------------------------------------------------------ hfield: elevation lookup + ray (engine_ray.c mj_rayHfield) */
static double hfield_height(const oracle_data* d, double x, double y, int* inside) {
  /* piecewise-linear surface over the MuJoCo triangulation: each cell split along the (r,c)-(r+1,c+1) diagonal */
  const cosim_model_t* m = &d->m;
  int nr = m->hfield_nrow, nc = m->hfield_ncol;
  double sx = m->hfield_size[0], sy = m->hfield_size[1], sz = m->hfield_size[2];
  double lx = x - m->ground_pos[0], ly = y - m->ground_pos[1];
  if (lx < -sx || lx > sx || ly < -sy || ly > sy) { *inside = 0; return 0; }
  *inside = 1;
  double fx = (lx + sx) / (2 * sx) * (nc - 1), fy = (ly + sy) / (2 * sy) * (nr - 1);
  int c = (int)floor(fx), r = (int)floor(fy);
  if (c > nc - 2) c = nc - 2;
  if (r > nr - 2) r = nr - 2;
  if (c < 0) c = 0;
  if (r < 0) r = 0;
  double u = fx - c, v = fy - r;
  double h00 = d->hfield[r * nc + c], h01 = d->hfield[r * nc + c + 1], h10 = d->hfield[(r + 1) * nc + c],
         h11 = d->hfield[(r + 1) * nc + c + 1];
  double h;
  if (u >= v) h = h00 + u * (h01 - h00) + v * (h11 - h01); /* triangle (r,c),(r,c+1),(r+1,c+1) */
  else h = h00 + v * (h10 - h00) + u * (h11 - h10);        /* triangle (r,c),(r+1,c+1),(r+1,c) */
  return m->ground_pos[2] + sz * h;
}

/* Heightfield narrowphase: mjc_ConvexHField (engine_collision_convex.c) restated.  The geom's axis-aligned box in the
 * hfield frame (six support queries) selects a sub-grid of cells; the cells are walked row by row as a triangle strip --
 * vertices (r+1, c), (r, c), (r+1, c+1), (r, c+1), ... -- and every three consecutive strip vertices span one prism
 * (top = terrain triangle, bottom at -size[3]).  Prisms whose top lies entirely below the geom's lowest point are
 * skipped; each remaining prism is tested against the geom with MPR and contributes at most one contact (normal prism ->
 * geom); at most mjMAXCONPAIR = 50 contacts per geom.  The ground geom has identity orientation in all four models. */
#define MAXCONPAIR 50
static void hfield_collide(oracle_data* d, int g) {
  const cosim_model_t* m = &d->m;
  int nr = m->hfield_nrow, nc = m->hfield_ncol;
  const double* size = m->hfield_size;
  double margin = fmax(m->ground_margin, m->geom_margin[g]);
  cobj_t o2, pr;
  make_cobj(d, g, &o2);
  for (int k = 0; k < 3; k++) { o2.pos[k] -= m->ground_pos[k]; o2.center[k] -= m->ground_pos[k]; } /* hfield frame */
  /* box-sphere tests (conservative early outs) */
  {
    int b = m->geom_bodyid[g];
    double v[3], ctr[3], rb = m->geom_rbound[g];
    mul_mat_vec3(v, d->xmat[b], m->geom_rcenter[g]);
    for (int k = 0; k < 3; k++) ctr[k] = d->xpos[b][k] + v[k] - m->ground_pos[k];
    for (int k = 0; k < 2; k++)
      if (size[k] < ctr[k] - rb - margin || -size[k] > ctr[k] + rb + margin) return;
    if (size[2] < ctr[2] - rb - margin || -size[3] > ctr[2] + rb + margin) return;
  }
  /* axis-aligned box of the geom through its support function */
  double lo[3], hi[3];
  for (int k = 0; k < 3; k++) {
    double dir[3] = {0, 0, 0}, p[3];
    dir[k] = 1; co_support(&o2, dir, p); hi[k] = p[k];
    dir[k] = -1; co_support(&o2, dir, p); lo[k] = p[k];
  }
  if (lo[0] - margin > size[0] || hi[0] + margin < -size[0] || lo[1] - margin > size[1] || hi[1] + margin < -size[1] ||
      lo[2] - margin > size[2] || hi[2] + margin < -size[3]) return;
  int cmin = (int)floor((lo[0] + size[0]) / (2 * size[0]) * (nc - 1)), cmax = (int)ceil((hi[0] + size[0]) / (2 * size[0]) * (nc - 1));
  int rmin = (int)floor((lo[1] + size[1]) / (2 * size[1]) * (nr - 1)), rmax = (int)ceil((hi[1] + size[1]) / (2 * size[1]) * (nr - 1));
  if (cmin < 0) cmin = 0;
  if (rmin < 0) rmin = 0;
  if (cmax > nc - 1) cmax = nc - 1;
  if (rmax > nr - 1) rmax = nr - 1;
  double dx = 2 * size[0] / (nc - 1), dy = 2 * size[1] / (nr - 1);
  memset(&pr, 0, sizeof pr);
  pr.kind = CO_PRISM;
  const int dr[2] = {1, 0};
  int cnt = 0;
  for (int r = rmin; r < rmax; r++) {
    int nvert = 0;
    for (int c = cmin; c <= cmax; c++)
      for (int i = 0; i < 2; i++) {
        /* addVert: shift the strip window, new vertex into slot 2 (bottom) / 5 (top) */
        for (int k = 0; k < 3; k++) {
          pr.prism[0][k] = pr.prism[1][k]; pr.prism[1][k] = pr.prism[2][k];
          pr.prism[3][k] = pr.prism[4][k]; pr.prism[4][k] = pr.prism[5][k];
        }
        pr.prism[2][0] = pr.prism[5][0] = dx * c - size[0];
        pr.prism[2][1] = pr.prism[5][1] = dy * (r + dr[i]) - size[1];
        pr.prism[2][2] = -size[3];
        pr.prism[5][2] = d->hfield[(r + dr[i]) * nc + c] * size[2] + margin;
        if (++nvert <= 2) continue;
        if (pr.prism[3][2] < lo[2] && pr.prism[4][2] < lo[2] && pr.prism[5][2] < lo[2]) continue;
        for (int k = 0; k < 3; k++) {
          pr.center[k] = 0;
          for (int q = 0; q < 6; q++) pr.center[k] += pr.prism[q][k] / 6.0;
        }
        double depth, dir[3], pos[3];
        if (mpr_penetration(&pr, &o2, &depth, dir, pos) != 0 || vec_eq0(dir)) continue;
        for (int k = 0; k < 3; k++) pos[k] += m->ground_pos[k];
        add_contact(d, g, margin - depth, pos, dir);
        if (++cnt >= MAXCONPAIR) return;
      }
  }
}

/* vertical ray from (x, y, z0) along -z; returns distance or -1 (reference utils/mujoco_utils.py:169 mj_rayHfield) */
double oracle_ray_down(const oracle_data* d, double x, double y, double z0) {
  int inside;
  if (d->m.ground_type == CS_GEOM_PLANE) return z0 - d->m.ground_pos[2];
  double h = hfield_height(d, x, y, &inside);
  if (!inside || h > z0) return -1;
  return z0 - h;
}

/* ------------------------------------------------------------------ constraints (engine_core_constraint.c) */
static void get_impedance(const double* solimp_in, double pos, double margin, double* imp, double* impP) {
  double s[5];
  memcpy(s, solimp_in, sizeof s);
  s[0] = fmin(MAXIMP, fmax(MINIMP, s[0]));
  s[1] = fmin(MAXIMP, fmax(MINIMP, s[1]));
  s[2] = fmax(0, s[2]);
  s[3] = fmin(MAXIMP, fmax(MINIMP, s[3]));
  s[4] = fmax(1, s[4]);
  if (s[0] == s[1] || s[2] <= MINVAL) { *imp = 0.5 * (s[0] + s[1]); *impP = 0; return; }
  double x = (pos - margin) / s[2], sgn = 1;
  if (x < 0) { x = -x; sgn = -1; }
  if (x >= 1 || x <= 0) { *imp = x >= 1 ? s[1] : s[0]; *impP = 0; return; }
  double y, yP;
  if (s[4] == 1) { y = x; yP = 1; }
  else if (x <= s[3]) { double a = 1 / pow(s[3], s[4] - 1); y = a * pow(x, s[4]); yP = s[4] * a * pow(x, s[4] - 1); }
  else { double b = 1 / pow(1 - s[3], s[4] - 1); y = 1 - b * pow(1 - x, s[4]); yP = s[4] * b * pow(1 - x, s[4] - 1); }
  *imp = s[0] + y * (s[1] - s[0]);
  *impP = yP * sgn * (s[1] - s[0]) / s[2];
}

static int add_row(oracle_data* d, int type, int id, double pos, double margin, double frictionloss, double diagApprox) {
  int i = d->nefc++;
  memset(d->J[i], 0, sizeof d->J[i]);
  d->efc_type[i] = type; d->efc_id[i] = id; d->efc_pos[i] = pos; d->efc_margin[i] = margin;
  d->efc_frictionloss[i] = frictionloss; d->efc_diagApprox[i] = diagApprox;
  return i;
}

static void make_constraint(oracle_data* d) {
  const cosim_model_t* m = &d->m;
  int nv = m->nv;
  d->nefc = 0;
  /* --- equality: connect (mj_instantiateEquality) */
  for (int e = 0; e < m->neq; e++) {
    int b1 = m->eq_body1[e], b2 = m->eq_body2[e];
    double p1[3], p2[3], v[3], j1[3][NV], j2[3][NV];
    mul_mat_vec3(v, d->xmat[b1], m->eq_anchor1[e]);
    for (int k = 0; k < 3; k++) p1[k] = d->xpos[b1][k] + v[k];
    mul_mat_vec3(v, d->xmat[b2], m->eq_anchor2[e]);
    for (int k = 0; k < 3; k++) p2[k] = d->xpos[b2][k] + v[k];
    jac_point(d, b1, p1, j1);
    jac_point(d, b2, p2, j2);
    double dA = m->body_invweight0[b1][0] + m->body_invweight0[b2][0];
    for (int k = 0; k < 3; k++) {
      int i = add_row(d, CT_EQUALITY, e, p1[k] - p2[k], 0, 0, dA);
      for (int c = 0; c < nv; c++) d->J[i][c] = j1[k][c] - j2[k][c];
    }
  }
  d->ne = d->nefc;
  /* --- dof friction loss (mj_instantiateFriction) */
  for (int i = 0; i < nv; i++)
    if (m->dof_frictionloss[i] > 0) {
      int r = add_row(d, CT_FRICTION, i, 0, 0, m->dof_frictionloss[i], m->dof_invweight0[i]);
      d->J[r][i] = 1;
    }
  d->nf = d->nefc - d->ne;
  /* --- joint limits (mj_instantiateLimit) */
  for (int j = 0; j < m->njnt; j++)
    if (m->jnt_limited[j] && m->jnt_type[j] == CS_JNT_HINGE) {
      double q = d->qpos[m->jnt_qposadr[j]], margin = m->jnt_margin[j];
      for (int side = -1; side <= 1; side += 2) {
        double dist = side * (m->jnt_range[j][(side + 1) / 2] - q);
        if (dist < margin) {
          int r = add_row(d, CT_LIMIT, j, dist, margin, 0, m->dof_invweight0[m->jnt_dofadr[j]]);
          d->J[r][m->jnt_dofadr[j]] = -side;
        }
      }
    }
  d->nl = d->nefc - d->ne - d->nf;
  /* --- contacts, pyramidal (mj_instantiateContact) */
  for (int ci = 0; ci < d->ncon; ci++) {
    contact_t* c = &d->con[ci];
    c->efc_address = -1;
    if (c->dist >= c->includemargin) continue;
    double jp[3][NV], jf[3][NV];
    jac_point(d, c->body, c->pos, jp); /* mj_jacDifPair: jac(body2) - jac(body1); the ground (world body) contributes 0 */
    if (c->body1 > 0) {
      double j1[3][NV];
      jac_point(d, c->body1, c->pos, j1);
      for (int r = 0; r < 3; r++)
        for (int k = 0; k < nv; k++) jp[r][k] -= j1[r][k];
    }
    for (int r = 0; r < 3; r++)
      for (int k = 0; k < nv; k++) jf[r][k] = c->frame[3 * r] * jp[0][k] + c->frame[3 * r + 1] * jp[1][k] + c->frame[3 * r + 2] * jp[2][k];
    double tran = m->body_invweight0[c->body][0] + m->body_invweight0[c->body1][0]; /* world body contributes 0 */
    c->efc_address = d->nefc;
    if (c->dim == 1) {
      int r = add_row(d, CT_CONTACT, ci, c->dist, c->includemargin, 0, tran);
      memcpy(d->J[r], jf[0], nv * sizeof(double));
    } else {
      for (int k = 1; k < c->dim && k < 3; k++)
        for (int s = 0; s < 2; s++) {
          double fr = c->friction[k - 1];
          int r = add_row(d, CT_CONTACT, ci, c->dist, c->includemargin, 0, tran + fr * fr * tran);
          for (int q = 0; q < nv; q++) d->J[r][q] = jf[0][q] + (s ? -fr : fr) * jf[k][q];
        }
    }
  }
  /* --- mj_makeImpedance */
  for (int i = 0; i < d->nefc; i++) {
    const double *solref, *solimp;
    switch (d->efc_type[i]) {
      case CT_EQUALITY: solref = m->eq_solref[d->efc_id[i]]; solimp = m->eq_solimp[d->efc_id[i]]; break;
      case CT_FRICTION: solref = m->dof_solref[d->efc_id[i]]; solimp = m->dof_solimp[d->efc_id[i]]; break;
      case CT_LIMIT: solref = m->jnt_solref[d->efc_id[i]]; solimp = m->jnt_solimp[d->efc_id[i]]; break;
      default: solref = d->con[d->efc_id[i]].solref; solimp = d->con[d->efc_id[i]].solimp; break;
    }
    double imp, impP, K, B;
    get_impedance(solimp, d->efc_pos[i], d->efc_margin[i], &imp, &impP);
    double dmax = fmin(MAXIMP, fmax(MINIMP, solimp[1]));
    if (solref[0] > 0) {
      double tc = fmax(solref[0], 2 * m->timestep); /* refsafe */
      double dr = solref[1];
      K = 1 / fmax(MINVAL, dmax * dmax * tc * tc * dr * dr);
      B = 2 / fmax(MINVAL, dmax * tc);
    } else { K = -solref[0] / fmax(MINVAL, dmax * dmax); B = -solref[1] / fmax(MINVAL, dmax); }
    if (d->efc_type[i] == CT_FRICTION) K = 0;
    d->efc_KBIP[i][0] = K; d->efc_KBIP[i][1] = B; d->efc_KBIP[i][2] = imp; d->efc_KBIP[i][3] = impP;
    d->efc_R[i] = fmax(MINVAL, (1 - imp) * d->efc_diagApprox[i] / imp);
  }
  /* pyramidal contacts: one regulariser for all edges, Rpy = 2 mu^2 R[first] */
  for (int ci = 0; ci < d->ncon; ci++) {
    contact_t* c = &d->con[ci];
    if (c->efc_address < 0 || c->dim == 1) continue;
    c->mu = c->friction[0] / sqrt(fmax(MINVAL, m->impratio));
    double Rpy = 2 * c->mu * c->mu * d->efc_R[c->efc_address];
    for (int k = 0; k < 2 * (c->dim - 1) && k < 4; k++) d->efc_R[c->efc_address + k] = Rpy;
  }
  for (int i = 0; i < d->nefc; i++) d->efc_D[i] = 1 / d->efc_R[i];
}

/* ------------------------------------------------------------------ velocity stage */
static void cross_motion(double* r, const double* vel, const double* v) { /* mju_crossMotion */
  double a[3], b[3], c[3];
  cross3(a, vel, v);
  cross3(b, vel, v + 3);
  cross3(c, vel + 3, v);
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2];
  r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
static void cross_force(double* r, const double* vel, const double* f) { /* mju_crossForce */
  double a[3], b[3], c[3];
  cross3(a, vel, f);
  cross3(b, vel + 3, f + 3);
  cross3(c, vel, f + 3);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2];
  r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}

static void com_vel(oracle_data* d) { /* mj_comVel */
  const cosim_model_t* m = &d->m;
  memset(d->cvel[0], 0, sizeof d->cvel[0]);
  for (int b = 1; b < m->nbody; b++) {
    double cvel[6];
    memcpy(cvel, d->cvel[m->body_parentid[b]], sizeof cvel);
    for (int j = m->body_jntadr[b]; j < m->body_jntadr[b] + m->body_jntnum[b]; j++) {
      int da = m->jnt_dofadr[j];
      if (m->jnt_type[j] == CS_JNT_FREE) {
        for (int k = 0; k < 3; k++) {
          memset(d->cdof_dot[da + k], 0, sizeof d->cdof_dot[0]);
          for (int q = 0; q < 6; q++) cvel[q] += d->cdof[da + k][q] * d->qvel[da + k];
        }
        for (int k = 3; k < 6; k++) cross_motion(d->cdof_dot[da + k], cvel, d->cdof[da + k]);
        for (int k = 3; k < 6; k++)
          for (int q = 0; q < 6; q++) cvel[q] += d->cdof[da + k][q] * d->qvel[da + k];
      } else {
        cross_motion(d->cdof_dot[da], cvel, d->cdof[da]);
        for (int q = 0; q < 6; q++) cvel[q] += d->cdof[da][q] * d->qvel[da];
      }
    }
    memcpy(d->cvel[b], cvel, sizeof cvel);
  }
}

static void rne_bias(oracle_data* d) { /* mj_rne(flg_acc = 0) */
  const cosim_model_t* m = &d->m;
  double cacc[CS_MAXBODY][6], cfrc[CS_MAXBODY][6];
  memset(cacc[0], 0, sizeof cacc[0]);
  for (int k = 0; k < 3; k++) cacc[0][3 + k] = -m->gravity[k];
  memset(cfrc[0], 0, sizeof cfrc[0]);
  for (int b = 1; b < m->nbody; b++) {
    memcpy(cacc[b], cacc[m->body_parentid[b]], sizeof cacc[b]);
    for (int i = m->body_dofadr[b]; i >= 0 && i < m->body_dofadr[b] + m->body_dofnum[b]; i++)
      for (int q = 0; q < 6; q++) cacc[b][q] += d->cdof_dot[i][q] * d->qvel[i];
    double t1[6], t2[6], t3[6];
    mul_inert_vec(t1, d->cinert[b], cacc[b]);
    mul_inert_vec(t2, d->cinert[b], d->cvel[b]);
    cross_force(t3, d->cvel[b], t2);
    for (int q = 0; q < 6; q++) cfrc[b][q] = t1[q] + t3[q];
  }
  for (int b = m->nbody - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    if (p > 0)
      for (int q = 0; q < 6; q++) cfrc[p][q] += cfrc[b][q];
  }
  for (int i = 0; i < m->nv; i++) {
    double s = 0;
    for (int q = 0; q < 6; q++) s += d->cdof[i][q] * cfrc[m->dof_bodyid[i]][q];
    d->qfrc_bias[i] = s;
  }
}

static void sensors(oracle_data* d) { /* mj_sensorPos framequat + mj_sensorVel gyro, velocimeter (engine_sensor.c) */
  const cosim_model_t* m = &d->m;
  int b = m->imu_bodyid;
  double q[4], smat[9], spos[3], v[3];
  mul_quat(q, d->xquat[b], m->imu_quat);
  normalize4(q);
  memcpy(d->sensor_quat, q, sizeof q);
  quat2mat(smat, q);
  mul_mat_vec3(v, d->xmat[b], m->imu_pos);
  for (int k = 0; k < 3; k++) spos[k] = d->xpos[b][k] + v[k];
  /* mj_objectVelocity(local): transform cvel from the subtree-com point to the site, rotate into the site frame */
  double dif[3], lin[3], tmp[3];
  for (int k = 0; k < 3; k++) dif[k] = spos[k] - d->subtree_com[m->body_rootid[b]][k];
  cross3(tmp, dif, d->cvel[b]);
  for (int k = 0; k < 3; k++) lin[k] = d->cvel[b][3 + k] - tmp[k];
  mul_matT_vec3(d->sensor_gyro, smat, d->cvel[b]);
  mul_matT_vec3(d->sensor_vel, smat, lin);
  for (int k = 0; k < 3; k++) { /* cutoff clamp (apply_cutoff) */
    if (m->gyro_cutoff > 0) d->sensor_gyro[k] = fmin(m->gyro_cutoff, fmax(-m->gyro_cutoff, d->sensor_gyro[k]));
    if (m->velocimeter_cutoff > 0) d->sensor_vel[k] = fmin(m->velocimeter_cutoff, fmax(-m->velocimeter_cutoff, d->sensor_vel[k]));
  }
}

/* ------------------------------------------------------------------ solver (engine_solver.c: mj_solNewton via mj_solPrimal) */
typedef struct {
  oracle_data* d;
  int nv, nefc;
  double Jaref[MAXEFC], Jv[MAXEFC], Ma[NV], Mv[NV], grad[NV], Mgrad[NV], search[NV];
  double quad[MAXEFC][3], quadGauss[3];
  double H[NV][NV];
  double cost, gauss;
  int lsiter;
} primal_ctx;

typedef struct { double alpha, cost, deriv[2]; } primal_pnt;

/* mj_constraintUpdate: forces, states and cost for a given jar = J qacc - aref */
static double constraint_update(oracle_data* d, const double* jar, double* force, int* state) {
  double cost = 0;
  for (int i = 0; i < d->nefc; i++) {
    double D = d->efc_D[i], R = d->efc_R[i], x = jar[i];
    switch (d->efc_type[i]) {
      case CT_EQUALITY:
        force[i] = -D * x; state[i] = ST_QUADRATIC; cost += 0.5 * D * x * x; break;
      case CT_FRICTION: {
        double f = d->efc_frictionloss[i];
        if (x <= -R * f) { force[i] = f; state[i] = ST_LINEARNEG; cost += -0.5 * R * f * f - f * x; }
        else if (x >= R * f) { force[i] = -f; state[i] = ST_LINEARPOS; cost += -0.5 * R * f * f + f * x; }
        else { force[i] = -D * x; state[i] = ST_QUADRATIC; cost += 0.5 * D * x * x; }
        break;
      }
      default:
        if (x >= 0) { force[i] = 0; state[i] = ST_SATISFIED; }
        else { force[i] = -D * x; state[i] = ST_QUADRATIC; cost += 0.5 * D * x * x; }
    }
  }
  return cost;
}

static void mul_M(const oracle_data* d, double* res, const double* v) {
  int nv = d->m.nv;
  for (int i = 0; i < nv; i++) {
    double s = 0;
    for (int j = 0; j < nv; j++) s += d->M[i][j] * v[j];
    res[i] = s;
  }
}

static void primal_update_constraint(primal_ctx* c) {
  oracle_data* d = c->d;
  int nv = c->nv;
  double cost = constraint_update(d, c->Jaref, d->efc_force, d->efc_state);
  for (int j = 0; j < nv; j++) {
    double s = 0;
    for (int i = 0; i < c->nefc; i++) s += d->J[i][j] * d->efc_force[i];
    d->qfrc_constraint[j] = s;
  }
  double g = 0;
  for (int j = 0; j < nv; j++) g += 0.5 * (c->Ma[j] - d->qfrc_smooth[j]) * (d->qacc[j] - d->qacc_smooth[j]);
  c->gauss = g;
  c->cost = cost + g;
}

static void primal_update_gradient(primal_ctx* c) { /* Newton: H = M + J' diag(D_active) J, Mgrad = H^-1 grad */
  oracle_data* d = c->d;
  int nv = c->nv;
  for (int j = 0; j < nv; j++) c->grad[j] = c->Ma[j] - d->qfrc_smooth[j] - d->qfrc_constraint[j];
  for (int a = 0; a < nv; a++)
    for (int b = 0; b < nv; b++) c->H[a][b] = d->M[a][b];
  for (int i = 0; i < c->nefc; i++)
    if (d->efc_state[i] == ST_QUADRATIC) {
      double D = d->efc_D[i];
      for (int a = 0; a < nv; a++) {
        if (d->J[i][a] == 0) continue;
        double t = D * d->J[i][a];
        for (int b = 0; b <= a; b++) c->H[a][b] += t * d->J[i][b];
      }
    }
  for (int a = 0; a < nv; a++)
    for (int b = a + 1; b < nv; b++) c->H[a][b] = c->H[b][a];
  chol_factor(c->H, nv);
  memcpy(c->Mgrad, c->grad, nv * sizeof(double));
  chol_solve(c->H, nv, c->Mgrad);
}

static void primal_eval(primal_ctx* c, primal_pnt* p) { /* PrimalEval */
  oracle_data* d = c->d;
  double a = p->alpha;
  double q0 = c->quadGauss[0], q1 = c->quadGauss[1], q2 = c->quadGauss[2];
  for (int i = 0; i < c->nefc; i++) {
    double x = c->Jaref[i] + a * c->Jv[i];
    switch (d->efc_type[i]) {
      case CT_EQUALITY:
        q0 += c->quad[i][0]; q1 += c->quad[i][1]; q2 += c->quad[i][2]; break;
      case CT_FRICTION: {
        double f = d->efc_frictionloss[i], Rf = d->efc_R[i] * f;
        if (-Rf < x && x < Rf) { q0 += c->quad[i][0]; q1 += c->quad[i][1]; q2 += c->quad[i][2]; }
        else if (x <= -Rf) { q0 += f * (-0.5 * Rf - c->Jaref[i]); q1 += -f * c->Jv[i]; }
        else { q0 += f * (-0.5 * Rf + c->Jaref[i]); q1 += f * c->Jv[i]; }
        break;
      }
      default:
        if (x < 0) { q0 += c->quad[i][0]; q1 += c->quad[i][1]; q2 += c->quad[i][2]; }
    }
  }
  p->cost = a * a * q2 + a * q1 + q0;
  p->deriv[0] = 2 * a * q2 + q1;
  p->deriv[1] = 2 * q2;
  if (p->deriv[1] <= 0) p->deriv[1] = MINVAL;
  c->lsiter++;
}

static int update_bracket(primal_ctx* c, primal_pnt* p, const primal_pnt cand[3], primal_pnt* pnext) {
  int flag = 0;
  for (int i = 0; i < 3; i++) {
    if (p->deriv[0] < 0 && cand[i].deriv[0] < 0 && p->deriv[0] < cand[i].deriv[0]) { *p = cand[i]; flag = 1; }
    else if (p->deriv[0] > 0 && cand[i].deriv[0] > 0 && p->deriv[0] > cand[i].deriv[0]) { *p = cand[i]; flag = 2; }
  }
  if (flag) { pnext->alpha = p->alpha - p->deriv[0] / p->deriv[1]; primal_eval(c, pnext); }
  return flag;
}

long g_ls_stat[6] = {0, 0, 0, 0, 0, 0}; /* diagnostics: searches, done after p1, one-sided Newton evals, bracket entries, bracket evals, total evals */
void oracle_ls_stats(long* out6) { for (int i = 0; i < 6; i++) out6[i] = g_ls_stat[i]; }
static double primal_search(primal_ctx* c) { /* PrimalSearch: exact line search on the piecewise-quadratic cost */
  oracle_data* d = c->d;
  const cosim_model_t* m = &d->m;
  int nv = c->nv, maxiter = m->ls_iterations;
  c->lsiter = 0;
  double snorm = 0;
  for (int j = 0; j < nv; j++) snorm += c->search[j] * c->search[j];
  snorm = sqrt(snorm);
  if (snorm < MINVAL) return 0;
  double scale = 1 / (m->meaninertia * (nv > 1 ? nv : 1));
  double gtol = m->tolerance * m->ls_tolerance * snorm / scale;
  mul_M(d, c->Mv, c->search);
  for (int i = 0; i < c->nefc; i++) {
    double s = 0;
    for (int j = 0; j < nv; j++) s += d->J[i][j] * c->search[j];
    c->Jv[i] = s;
  }
  /* PrimalPrepare */
  c->quadGauss[0] = c->gauss; c->quadGauss[1] = 0; c->quadGauss[2] = 0;
  for (int j = 0; j < nv; j++) {
    c->quadGauss[1] += c->search[j] * (c->Ma[j] - d->qfrc_smooth[j]);
    c->quadGauss[2] += 0.5 * c->search[j] * c->Mv[j];
  }
  for (int i = 0; i < c->nefc; i++) {
    double D = d->efc_D[i];
    c->quad[i][0] = 0.5 * D * c->Jaref[i] * c->Jaref[i];
    c->quad[i][1] = D * c->Jaref[i] * c->Jv[i];
    c->quad[i][2] = 0.5 * D * c->Jv[i] * c->Jv[i];
  }
  primal_pnt p0, p1, p2, pmid, p1next, p2next;
  g_ls_stat[0]++;
  p0.alpha = 0; primal_eval(c, &p0);
  p1.alpha = p0.alpha - p0.deriv[0] / p0.deriv[1]; primal_eval(c, &p1);
  if (p0.cost < p1.cost) p1 = p0;
  if (fabs(p1.deriv[0]) < gtol) { g_ls_stat[1]++; return p1.alpha; }
  int dir = p1.deriv[0] < 0 ? 1 : -1;
  int p2update = 0;
  p2 = p1;
  while (p1.deriv[0] * dir <= -gtol && c->lsiter < maxiter) {
    p2 = p1; p2update = 1;
    p1.alpha -= p1.deriv[0] / p1.deriv[1];
    primal_eval(c, &p1);
    g_ls_stat[2]++;
    if (fabs(p1.deriv[0]) < gtol) return p1.alpha;
  }
  if (c->lsiter >= maxiter) return p1.alpha;
  if (!p2update) return p1.alpha;
  p2next = p1;
  g_ls_stat[3]++;
  { int before = c->lsiter; (void)before; }
  p1next.alpha = p1.alpha - p1.deriv[0] / p1.deriv[1]; primal_eval(c, &p1next);
  while (c->lsiter < maxiter) {
    pmid.alpha = 0.5 * (p1.alpha + p2.alpha); primal_eval(c, &pmid);
    g_ls_stat[4]++;
    primal_pnt cand[3] = {p1next, p2next, pmid};
    int best = -1;
    double bestcost = 0;
    for (int i = 0; i < 3; i++)
      if (fabs(cand[i].deriv[0]) < gtol && (best == -1 || cand[i].cost < bestcost)) { best = i; bestcost = cand[i].cost; }
    if (best >= 0) return cand[best].alpha;
    int b1 = update_bracket(c, &p1, cand, &p1next), b2 = update_bracket(c, &p2, cand, &p2next);
    if (!b1 && !b2) return pmid.alpha;
  }
  if (p1.cost <= p2.cost && p1.cost < p0.cost) return p1.alpha;
  if (p2.cost <= p1.cost && p2.cost < p0.cost) return p2.alpha;
  return 0;
}

static void solve_newton(oracle_data* d) {
  const cosim_model_t* m = &d->m;
  static __thread primal_ctx ctx;
  primal_ctx* c = &ctx;
  int nv = m->nv;
  c->d = d; c->nv = nv; c->nefc = d->nefc;
  for (int i = 0; i < d->nefc; i++) {
    double s = -d->efc_aref[i];
    for (int j = 0; j < nv; j++) s += d->J[i][j] * d->qacc[j];
    c->Jaref[i] = s;
  }
  mul_M(d, c->Ma, d->qacc);
  primal_update_constraint(c);
  primal_update_gradient(c);
  for (int j = 0; j < nv; j++) c->search[j] = -c->Mgrad[j];
  double scale = 1 / (m->meaninertia * (nv > 1 ? nv : 1));
  int iter = 0;
  while (iter < m->iterations) {
    double alpha = primal_search(c);
    d->ls_total += c->lsiter;
    if (alpha == 0) break;
    for (int j = 0; j < nv; j++) { d->qacc[j] += alpha * c->search[j]; c->Ma[j] += alpha * c->Mv[j]; }
    for (int i = 0; i < d->nefc; i++) c->Jaref[i] += alpha * c->Jv[i];
    double oldcost = c->cost;
    primal_update_constraint(c);
    primal_update_gradient(c);
    for (int j = 0; j < nv; j++) c->search[j] = -c->Mgrad[j];
    double improvement = scale * (oldcost - c->cost), gn = 0;
    for (int j = 0; j < nv; j++) gn += c->grad[j] * c->grad[j];
    double gradient = scale * sqrt(gn);
    iter++;
    if (improvement < m->tolerance || gradient < m->tolerance) break;
  }
  d->solver_niter += iter;
  d->solver_cost = c->cost;
}

/* ------------------------------------------------------------------ mj_forward */
static void fwd_position(oracle_data* d) {
  kinematics(d);
  com_pos(d);
  crb(d);
  collision(d);
  make_constraint(d);
}

static void fwd_velocity(oracle_data* d) {
  const cosim_model_t* m = &d->m;
  int nv = m->nv;
  com_vel(d);
  for (int i = 0; i < nv; i++) d->qfrc_passive[i] = -m->dof_damping[i] * d->qvel[i]; /* mj_passive */
  for (int i = 0; i < d->nefc; i++) { /* mj_referenceConstraint */
    double v = 0;
    for (int j = 0; j < nv; j++) v += d->J[i][j] * d->qvel[j];
    d->efc_vel[i] = v;
    d->efc_aref[i] = -d->efc_KBIP[i][1] * v - d->efc_KBIP[i][0] * d->efc_KBIP[i][2] * (d->efc_pos[i] - d->efc_margin[i]);
  }
  rne_bias(d);
}

static void fwd_actuation(oracle_data* d) { /* mj_fwdActuation: motors */
  const cosim_model_t* m = &d->m;
  memset(d->qfrc_actuator, 0, sizeof d->qfrc_actuator);
  for (int u = 0; u < m->nu; u++) {
    double c = d->ctrl[u];
    if (m->act_ctrllimited[u]) c = fmin(m->act_ctrlrange[u][1], fmax(m->act_ctrlrange[u][0], c));
    d->actuator_force[u] = c;
    d->qfrc_actuator[m->act_dofid[u]] += m->act_gear[u] * c;
  }
  for (int j = 0; j < m->njnt; j++) /* joint-level actuatorfrcrange clamp */
    if (m->jnt_actfrclimited[j] && m->jnt_type[j] == CS_JNT_HINGE) {
      int i = m->jnt_dofadr[j];
      d->qfrc_actuator[i] = fmin(m->jnt_actfrcrange[j][1], fmax(m->jnt_actfrcrange[j][0], d->qfrc_actuator[i]));
    }
}

static void fwd_acceleration(oracle_data* d) {
  int nv = d->m.nv;
  for (int i = 0; i < nv; i++) d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
  memcpy(d->qacc_smooth, d->qfrc_smooth, nv * sizeof(double));
  chol_solve(d->L, nv, d->qacc_smooth);
}

long g_warm_total = 0, g_warm_smooth = 0; /* diagnostics: how often qacc_smooth beats the warm start */
int g_warm_always = 0;                    /* diagnostics: 1 = always start from qacc_warmstart (what the HIP engine does) */
void oracle_warm_stats(long* out2, int always) { out2[0] = g_warm_total; out2[1] = g_warm_smooth; g_warm_always = always; }
static void fwd_constraint(oracle_data* d) { /* mj_fwdConstraint incl. warmstart selection */
  const cosim_model_t* m = &d->m;
  int nv = m->nv;
  if (d->nefc == 0) {
    memcpy(d->qacc, d->qacc_smooth, nv * sizeof(double));
    memset(d->qfrc_constraint, 0, sizeof d->qfrc_constraint);
    return;
  }
  double jar[MAXEFC], force[MAXEFC], Ma[NV];
  int state[MAXEFC];
  for (int i = 0; i < d->nefc; i++) {
    double s = -d->efc_aref[i];
    for (int j = 0; j < nv; j++) s += d->J[i][j] * d->qacc_warmstart[j];
    jar[i] = s;
  }
  double cost_warm = constraint_update(d, jar, force, state);
  mul_M(d, Ma, d->qacc_warmstart);
  for (int j = 0; j < nv; j++) cost_warm += 0.5 * (Ma[j] - d->qfrc_smooth[j]) * (d->qacc_warmstart[j] - d->qacc_smooth[j]);
  for (int i = 0; i < d->nefc; i++) {
    double s = -d->efc_aref[i];
    for (int j = 0; j < nv; j++) s += d->J[i][j] * d->qacc_smooth[j];
    jar[i] = s;
  }
  double cost_smooth = constraint_update(d, jar, force, state);
  g_warm_total++;
  if (cost_warm > cost_smooth) g_warm_smooth++;
  memcpy(d->qacc, (cost_warm > cost_smooth && !g_warm_always) ? d->qacc_smooth : d->qacc_warmstart, nv * sizeof(double));
  solve_newton(d);
}

void oracle_forward(oracle_data* d) {
  fwd_position(d);
  fwd_velocity(d);
  sensors(d);
  fwd_actuation(d);
  fwd_acceleration(d);
  fwd_constraint(d);
}

/* ------------------------------------------------------------------ mj_implicit (implicitfast) + mj_advance */
static void integrate(oracle_data* d) {
  const cosim_model_t* m = &d->m;
  int nv = m->nv;
  double h = m->timestep;
  static __thread double MH[NV][NV];
  double qacc[NV];
  /* qDeriv of motors + joint damping = -diag(damping); implicitfast drops the RNE derivative */
  memcpy(MH, d->M, sizeof d->M);
  for (int i = 0; i < nv; i++) MH[i][i] += h * m->dof_damping[i];
  chol_factor(MH, nv);
  for (int i = 0; i < nv; i++) qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
  chol_solve(MH, nv, qacc);
  for (int i = 0; i < nv; i++) d->qvel[i] += h * qacc[i];
  for (int j = 0; j < m->njnt; j++) { /* mj_integratePos */
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == CS_JNT_FREE) {
      for (int k = 0; k < 3; k++) d->qpos[qa + k] += h * d->qvel[da + k];
      double w[3] = {d->qvel[da + 3], d->qvel[da + 4], d->qvel[da + 5]}; /* mju_quatIntegrate */
      double ang = h * normalize3(w), q[4];
      axis_angle_quat(q, w, ang);
      normalize4(d->qpos + qa + 3);
      mul_quat(d->qpos + qa + 3, d->qpos + qa + 3, q);
    } else
      d->qpos[qa] += h * d->qvel[da];
  }
  memcpy(d->qacc_warmstart, d->qacc, nv * sizeof(double)); /* mj_advance: save qacc for next step's warmstart */
  d->time += h;
}

static int bad_state(const oracle_data* d) { /* mj_checkPos / mj_checkVel / mj_checkAcc */
  for (int i = 0; i < d->m.nq; i++) if (!isfinite(d->qpos[i]) || fabs(d->qpos[i]) > 1e10) return 1;
  for (int i = 0; i < d->m.nv; i++) if (!isfinite(d->qvel[i]) || fabs(d->qvel[i]) > 1e10 || !isfinite(d->qacc[i]) || fabs(d->qacc[i]) > 1e10) return 1;
  return 0;
}

void oracle_step(oracle_data* d) { /* mj_step */
  oracle_forward(d);
  if (bad_state(d)) { d->bad = 1; return; }
  integrate(d);
}

/* mj_rnePostConstraint restricted to what the reference reads: cfrc_ext = contact wrenches per body, about the
   subtree CoM of the body's root, world aligned, [torque; force] (reference flamingo_p_v3.py:226) */
void oracle_cfrc_ext(oracle_data* d) {
  const cosim_model_t* m = &d->m;
  memset(d->cfrc_ext, 0, sizeof d->cfrc_ext);
  for (int ci = 0; ci < d->ncon; ci++) {
    const contact_t* c = &d->con[ci];
    if (c->efc_address < 0) continue;
    double fl[3] = {0, 0, 0}; /* mj_contactForce: pyramid edges -> frame coordinates */
    if (c->dim == 1) fl[0] = d->efc_force[c->efc_address];
    else
      for (int k = 1; k < c->dim && k < 3; k++) {
        double f0 = d->efc_force[c->efc_address + 2 * (k - 1)], f1 = d->efc_force[c->efc_address + 2 * (k - 1) + 1];
        fl[0] += f0 + f1;
        fl[k] = (f0 - f1) * c->friction[k - 1];
      }
    double fw[3], tq[3], dif[3];
    for (int k = 0; k < 3; k++) fw[k] = c->frame[k] * fl[0] + c->frame[3 + k] * fl[1] + c->frame[6 + k] * fl[2];
    for (int k = 0; k < 3; k++) dif[k] = c->pos[k] - d->subtree_com[m->body_rootid[c->body]][k];
    cross3(tq, dif, fw);
    for (int k = 0; k < 3; k++) { d->cfrc_ext[c->body][k] += tq[k]; d->cfrc_ext[c->body][3 + k] += fw[k]; }
    if (c->body1 > 0) /* mj_rnePostConstraint: equal and opposite on geom1's body */
      for (int k = 0; k < 3; k++) { d->cfrc_ext[c->body1][k] -= tq[k]; d->cfrc_ext[c->body1][3 + k] -= fw[k]; }
  }
}

/* ------------------------------------------------------------------ C API for the tests (ctypes) */
oracle_data* oracle_new(const cosim_model_t* m, const float* hull_vert, const int* hull_adr, const int* hull_nbr, const float* hfield) {
  if (m->magic != CS_MODEL_MAGIC || m->magic_end != CS_MODEL_MAGIC) return NULL;
  oracle_data* d = (oracle_data*)calloc(1, sizeof(oracle_data));
  d->m = *m;
  d->hull_vert = hull_vert; d->hull_adr = hull_adr; d->hull_nbr = hull_nbr; d->hfield = hfield;
  memcpy(d->qpos, m->qpos0, sizeof(double) * m->nq);
  return d;
}
void oracle_free(oracle_data* d) { free(d); }
int oracle_model_sizeof(void) { return (int)sizeof(cosim_model_t); }
cosim_model_t* oracle_model(oracle_data* d) { return &d->m; }

void oracle_reset(oracle_data* d, const double* qpos, const double* qvel) { /* mj_resetData + state write */
  memset(d->qvel, 0, sizeof d->qvel);
  memset(d->qacc_warmstart, 0, sizeof d->qacc_warmstart);
  memset(d->ctrl, 0, sizeof d->ctrl);
  memset(d->qacc, 0, sizeof d->qacc);
  d->time = 0; d->bad = 0; d->solver_niter = 0; d->ls_total = 0;
  memcpy(d->qpos, qpos ? qpos : d->m.qpos0, sizeof(double) * d->m.nq);
  if (qvel) memcpy(d->qvel, qvel, sizeof(double) * d->m.nv);
}

/* named views for the tests */
double* oracle_ptr(oracle_data* d, const char* name) {
#define F(n) if (!strcmp(name, #n)) return (double*)d->n;
  F(qpos) F(qvel) F(qacc) F(qacc_warmstart) F(ctrl) F(xpos) F(xquat) F(xmat) F(xipos) F(subtree_com) F(cinert) F(cdof)
  F(M) F(qfrc_bias) F(qfrc_passive) F(qfrc_actuator) F(qfrc_smooth) F(qacc_smooth) F(qfrc_constraint) F(efc_force)
  F(efc_pos) F(efc_aref) F(efc_R) F(efc_D) F(efc_vel) F(J) F(sensor_quat) F(sensor_gyro) F(sensor_vel) F(cfrc_ext) F(cvel)
  F(efc_diagApprox) F(efc_KBIP)
#undef F
  if (!strcmp(name, "time")) return &d->time;
  if (!strcmp(name, "solver_cost")) return &d->solver_cost;
  return NULL;
}
int oracle_int(oracle_data* d, const char* name) {
  if (!strcmp(name, "ncon")) return d->ncon;
  if (!strcmp(name, "nefc")) return d->nefc;
  if (!strcmp(name, "ne")) return d->ne;
  if (!strcmp(name, "nf")) return d->nf;
  if (!strcmp(name, "nl")) return d->nl;
  if (!strcmp(name, "solver_niter")) return d->solver_niter;
  if (!strcmp(name, "ls_total")) return d->ls_total;
  if (!strcmp(name, "bad")) return d->bad;
  if (!strcmp(name, "contact_overflow")) return d->contact_overflow;
  if (!strcmp(name, "maxefc")) return MAXEFC;
  if (!strcmp(name, "nvmax")) return NV;
  return -1;
}
/* contact i -> [dist, pos3, normal3, geom, efc_address, geom1] */
void oracle_contact(oracle_data* d, int i, double* out) {
  const contact_t* c = &d->con[i];
  out[0] = c->dist;
  for (int k = 0; k < 3; k++) { out[1 + k] = c->pos[k]; out[4 + k] = c->frame[k]; }
  out[7] = c->geom; out[8] = c->efc_address; out[9] = c->geom1;
}

/* nsteps control steps in one call (actions[nsteps][nu], already delay-filtered): the timing loop of bench.py's CPU leg */
void oracle_control_step(oracle_data* d, const double* filtered_action, double* tq);
int oracle_rollout(oracle_data* d, const double* actions, int nsteps) {
  double tq[CS_MAXU];
  int t;
  for (t = 0; t < nsteps && !d->bad; t++) oracle_control_step(d, actions + (size_t)t * d->m.nu, tq);
  return t;
}

/* The same loop with the robot env's `_is_done` (reference flamingo_p_v3.py:225-233: any signed component of cfrc_ext of the listed
   bodies above 1.0; the other robots never terminate): stops after the control step that terminates.  Returns the steps done. */
int oracle_rollout_env(oracle_data* d, const double* actions, int nsteps, int* terminated) {
  double tq[CS_MAXU];
  const cosim_model_t* m = &d->m;
  int t = 0;
  *terminated = 0;
  while (t < nsteps && !d->bad) {
    oracle_control_step(d, actions + (size_t)t * m->nu, tq);
    t++;
    if (m->term_mode == 1) {
      for (int i = 0; i < m->nterm_body; i++)
        for (int k = 0; k < 6; k++)
          if (d->cfrc_ext[m->term_body[i]][k] > 1.0) *terminated = 1;
      if (*terminated) break;
    }
  }
  return t;
}

/* test hooks: one MPR query between two robot geoms (returns 0 when penetrating); self-collision switch */
int oracle_mpr_pair(oracle_data* d, int g1, int g2, double* out7) {
  cobj_t o1, o2;
  make_cobj(d, g1, &o1);
  make_cobj(d, g2, &o2);
  return mpr_penetration(&o1, &o2, out7, out7 + 1, out7 + 4);
}
void oracle_set_self_collision(oracle_data* d, int on) { d->no_self_collision = !on; }
void oracle_set_boxbox_mpr(oracle_data* d, int on) { d->boxbox_mpr = on; } /* 1: box-box pairs through MPR (one contact), as before mjc_BoxBox was restated */
/* box-box between two boxes given directly (pose = pos3 + row-major mat9); out = up to 8 x {pos3, dist}, nrm3 = normal box1 -> box2 */
int oracle_box_box(const double* pose1, const double* size1, const double* pose2, const double* size2, double margin, double* out32, double* nrm3) {
  bb_point_t pt[8];
  int n = box_box_points(pose1, pose1 + 3, size1, pose2, pose2 + 3, size2, margin, pt, nrm3);
  for (int i = 0; i < n; i++) { memcpy(out32 + 4 * i, pt[i].pos, 24); out32[4 * i + 3] = pt[i].dist; }
  return n;
}
void oracle_mpr_stats(long* out3) { out3[0] = g_mpr_calls; out3[1] = g_mpr_supports; out3[2] = g_mpr_hits; }
/* MPR between two primitives given directly (kind: CO_SPHERE/CO_CYLINDER/CO_BOX; pose = pos3 + row-major mat9; size3) */
int oracle_mpr_prims(int k1, const double* pose1, const double* size1, int k2, const double* pose2, const double* size2, double* out7) {
  cobj_t o1, o2;
  memset(&o1, 0, sizeof o1); memset(&o2, 0, sizeof o2);
  o1.kind = k1; memcpy(o1.pos, pose1, 24); memcpy(o1.mat, pose1 + 3, 72); memcpy(o1.size, size1, 24); memcpy(o1.center, pose1, 24);
  o2.kind = k2; memcpy(o2.pos, pose2, 24); memcpy(o2.mat, pose2 + 3, 72); memcpy(o2.size, size2, 24); memcpy(o2.center, pose2, 24);
  return mpr_penetration(&o1, &o2, out7, out7 + 1, out7 + 4);
}

/* One control step of the robot-env layer (reference flamingo_light_v1.py:131-154): PD torque from the (already
   delay-filtered) action, held over frame_skip substeps.  Returns the applied torques in tq[nu]. */
void oracle_control_step(oracle_data* d, const double* filtered_action, double* tq) {
  const cosim_model_t* m = &d->m;
  for (int u = 0; u < m->nu; u++) {
    double a = filtered_action[u] * m->ctl_scale[u], g = m->ctl_gear[u];
    double q = d->qpos[m->ctl_qadr[u]] * g, qd = d->qvel[m->ctl_dadr[u]] * g;
    double t = m->ctl_velmode[u] ? m->ctl_kd[u] * (a - qd) : m->ctl_kp[u] * (a - q) + m->ctl_kd[u] * (0.0 - qd);
    t *= m->ctl_gamma[u];
    t = fmin(m->ctl_maxtq[u], fmax(-m->ctl_maxtq[u], t));
    tq[u] = t;
    d->ctrl[u] = t;
  }
  for (int s = 0; s < m->frame_skip && !d->bad; s++) oracle_step(d);
  oracle_cfrc_ext(d); /* gymnasium do_simulation calls mj_rnePostConstraint after the substeps */
}
