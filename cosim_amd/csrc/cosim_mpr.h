// cosim_mpr.h — convex-convex narrowphase of the rollout kernel: Minkowski Portal Refinement in fp32.
//
// MuJoCo 3.2.7 sends every geom pair without an analytic routine (for the cosim robots: mesh / cylinder / box pairs)
// through libccd's ccdMPRPenetration (engine_collision_convex.c mjc_Convex -> mjc_MPRIteration; libccd 2.1 src/mpr.c is a
// third-party dependency, absent from the reference tree).  The routine below restates the published algorithm
// (G. Snethen, "XenoCollide", Game Programming Gems 7) with libccd's structure -- discoverPortal, refinePortal,
// findPenetr, findPos -- and its predicates, with FLT_EPSILON in place of DBL_EPSILON, mpr_tolerance 1e-6 and at most
// 50 iterations per loop (MuJoCo's ccd_tolerance / ccd_iterations; libccd's refinePortal is uncapped, here every loop is
// bounded so that every wave reaches the end of the kernel).  oracle/cosim_oracle.c holds the fp64 twin.
//
// Two ways to run it, same code:
//   * lane-parallel: every lane owns one pair of primitives (box / cylinder / sphere supports are O(1));
//   * wave-cooperative (COOP): all 64 lanes run ONE pair with identical values and share the scan over a mesh hull's
//     vertices in the support function (control flow is uniform because the data is).
#pragma once

namespace cosim {

constexpr float MPR_EPS = 1.1920929e-07f;
constexpr float MPR_TOL = 1e-6f;
constexpr int MPR_MAXIT = 50;

struct CObj {            // one convex geom, world pose
  int kind;              // CS_GEOM_*
  float pos[3], q[4];    // primitive: geom frame; mesh: body frame (hull vertices are stored in body coordinates)
  float size[3];
  int adr, num;          // mesh: slice of the hull vertex array
  float center[3];       // mjccd_center
};
struct MprSup { float v[3], v1[3]; };  // Minkowski-difference support point v = s1(dir) - s2(-dir) and its s1 part

__device__ __forceinline__ bool mpr_zero(float x) { return fabsf(x) < MPR_EPS; }
__device__ __forceinline__ bool mpr_eq(float a, float b) {
  const float ab = fabsf(a - b);
  if (ab < MPR_EPS) return true;
  a = fabsf(a); b = fabsf(b);
  return b > a ? ab < MPR_EPS * b : ab < MPR_EPS * a;
}
__device__ __forceinline__ bool mpr_vzero(const float* v) { return mpr_eq(v[0], 0.f) && mpr_eq(v[1], 0.f) && mpr_eq(v[2], 0.f); }
__device__ __forceinline__ float mpr_sign(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }   // mju_sign
__device__ __forceinline__ float mpr_normalize(float* v) {
  const float n = sqrtf(dot3(v, v));
  if (n < 1e-30f) { v[0] = 1.f; v[1] = 0.f; v[2] = 0.f; return 0.f; }
  const float s = 1.f / n;
  v[0] *= s; v[1] *= s; v[2] *= s;
  return n;
}

// mjccd_support: furthest point of the geom along the unit world direction
template <int GTM, bool COOP>
__device__ __forceinline__ void cobj_support(const CObj& o, const float* hull, const float* dir, float* out, int ln) {
  const float qi[4] = {o.q[0], -o.q[1], -o.q[2], -o.q[3]};
  float l[3], r[3] = {0.f, 0.f, 0.f};
  qrot(l, qi, dir);
  if (COOP && (GTM & GT_MESH) && o.kind == CS_GEOM_MESH) {
    float best = -3.0e38f;
    int besti = 0x7fffffff;
    for (int i = ln; i < o.num; i += 64) {
      const float* v = hull + 3 * (o.adr + i);
      const float t = l[0] * v[0] + l[1] * v[1] + l[2] * v[2];
      if (t > best) { best = t; besti = i; }
    }
    const float bmax = -wave_min(-best);
    int bi = (best == bmax) ? besti : 0x7fffffff;   // lowest index among ties, like a sequential scan
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) bi = min(bi, __shfl_xor(bi, s, 64));
    if (bi >= o.num) bi = 0;
    const float* v = hull + 3 * (o.adr + bi);
    r[0] = v[0]; r[1] = v[1]; r[2] = v[2];
  } else if ((GTM & GT_SPHERE) && o.kind == CS_GEOM_SPHERE) {
    r[0] = l[0] * o.size[0]; r[1] = l[1] * o.size[0]; r[2] = l[2] * o.size[0];
  } else if ((GTM & GT_CYLINDER) && o.kind == CS_GEOM_CYLINDER) {
    const float t = sqrtf(l[0] * l[0] + l[1] * l[1]);
    if (t > MINVAL) { r[0] = l[0] / t * o.size[0]; r[1] = l[1] / t * o.size[0]; }
    r[2] = mpr_sign(l[2]) * o.size[1];
  } else if ((GTM & GT_BOX) && o.kind == CS_GEOM_BOX) {
    r[0] = mpr_sign(l[0]) * o.size[0]; r[1] = mpr_sign(l[1]) * o.size[1]; r[2] = mpr_sign(l[2]) * o.size[2];
  }
  float w[3];
  qrot(w, o.q, r);
  out[0] = o.pos[0] + w[0]; out[1] = o.pos[1] + w[1]; out[2] = o.pos[2] + w[2];
}

template <int GTM, bool COOP>
struct MprPair {
  const CObj &a, &b;
  const float* hull;
  int ln;
  __device__ __forceinline__ void operator()(const float* dir, MprSup& s) const {   // __ccdSupport
    const float nd[3] = {-dir[0], -dir[1], -dir[2]};
    float v2[3];
    cobj_support<GTM, COOP>(a, hull, dir, s.v1, ln);
    cobj_support<GTM, COOP>(b, hull, nd, v2, ln);
    s.v[0] = s.v1[0] - v2[0]; s.v[1] = s.v1[1] - v2[1]; s.v[2] = s.v1[2] - v2[2];
  }
};

__device__ __forceinline__ void mpr_portal_dir(const MprSup& p1, const MprSup& p2, const MprSup& p3, float* dir) {
  const float a[3] = {p2.v[0] - p1.v[0], p2.v[1] - p1.v[1], p2.v[2] - p1.v[2]}, b[3] = {p3.v[0] - p1.v[0], p3.v[1] - p1.v[1], p3.v[2] - p1.v[2]};
  cross(dir, a, b);
  mpr_normalize(dir);
}
__device__ __forceinline__ bool mpr_reach_tol(const MprSup& p1, const MprSup& p2, const MprSup& p3, const MprSup& v4, const float* dir) {
  const float dv4 = dot3(v4.v, dir);
  const float d = fminf(dv4 - dot3(p1.v, dir), fminf(dv4 - dot3(p2.v, dir), dv4 - dot3(p3.v, dir)));
  return mpr_eq(d, MPR_TOL) || d < MPR_TOL;
}
__device__ __forceinline__ void mpr_expand(const float* v0, MprSup& p1, MprSup& p2, MprSup& p3, const MprSup& v4) {
  float v4v0[3];
  cross(v4v0, v4.v, v0);
  if (dot3(p1.v, v4v0) > 0.f) {
    if (dot3(p2.v, v4v0) > 0.f) p1 = v4; else p3 = v4;
  } else {
    if (dot3(p3.v, v4v0) > 0.f) p2 = v4; else p1 = v4;
  }
}
__device__ __forceinline__ float mpr_seg_dist2(const float* x0, const float* b, float* wit) {   // ccdVec3PointSegmentDist2, P = origin
  const float d[3] = {b[0] - x0[0], b[1] - x0[1], b[2] - x0[2]};
  const float dd = dot3(d, d);
  const float t = dd > 0.f ? -dot3(x0, d) / dd : 0.f;
  if (t < 0.f || mpr_zero(t)) { wit[0] = x0[0]; wit[1] = x0[1]; wit[2] = x0[2]; return dot3(x0, x0); }
  if (t > 1.f || mpr_eq(t, 1.f)) { wit[0] = b[0]; wit[1] = b[1]; wit[2] = b[2]; return dot3(b, b); }
  wit[0] = x0[0] + t * d[0]; wit[1] = x0[1] + t * d[1]; wit[2] = x0[2] + t * d[2];
  return dot3(wit, wit);
}
__device__ __forceinline__ float mpr_tri_dist2(const float* x0, const float* B, const float* C, float* wit) {   // ccdVec3PointTriDist2, P = origin
  const float d1[3] = {B[0] - x0[0], B[1] - x0[1], B[2] - x0[2]}, d2[3] = {C[0] - x0[0], C[1] - x0[1], C[2] - x0[2]};
  const float v = dot3(d1, d1), w = dot3(d2, d2), p = dot3(x0, d1), q = dot3(x0, d2), r = dot3(d1, d2);
  const float den = w * v - r * r;
  const float s = (q * r - w * p) / den, t = (-s * r - q) / w;   // degenerate triangle: NaN fails every test below
  if ((mpr_zero(s) || s > 0.f) && (mpr_eq(s, 1.f) || s < 1.f) && (mpr_zero(t) || t > 0.f) && (mpr_eq(t, 1.f) || t < 1.f) &&
      (mpr_eq(t + s, 1.f) || t + s < 1.f)) {
    wit[0] = x0[0] + s * d1[0] + t * d2[0]; wit[1] = x0[1] + s * d1[1] + t * d2[1]; wit[2] = x0[2] + s * d1[2] + t * d2[2];
    return dot3(wit, wit);
  }
  float w2[3];
  float dist = mpr_seg_dist2(x0, B, wit);
  float dist2 = mpr_seg_dist2(x0, C, w2);
  if (dist2 < dist) { dist = dist2; wit[0] = w2[0]; wit[1] = w2[1]; wit[2] = w2[2]; }
  dist2 = mpr_seg_dist2(B, C, w2);
  if (dist2 < dist) { dist = dist2; wit[0] = w2[0]; wit[1] = w2[1]; wit[2] = w2[2]; }
  return dist;
}

// ccdMPRPenetration: true when the geoms penetrate; depth, dir (geom1 -> geom2) and pos as libccd returns them.
template <class SUP>
__device__ __forceinline__ bool mpr_penetration(const SUP& sup, const float* c1, const float* c2, float& depth, float* dir_out, float* pos) {
  MprSup p1, p2, p3, v4;
  float v0[3] = {c1[0] - c2[0], c1[1] - c2[1], c1[2] - c2[2]};
  float dir[3], va[3], vb[3], dt;
  // ---- discoverPortal
  if (mpr_vzero(v0)) v0[0] += MPR_EPS * 10.f;
  dir[0] = -v0[0]; dir[1] = -v0[1]; dir[2] = -v0[2];
  mpr_normalize(dir);
  sup(dir, p1);
  dt = dot3(p1.v, dir);
  if (mpr_zero(dt) || dt < 0.f) return false;
  cross(dir, v0, p1.v);
  if (mpr_zero(dot3(dir, dir))) {
    if (mpr_vzero(p1.v)) return false;                    // touching: depth 0, no direction -> MuJoCo drops it
    // origin on the v0-v1 segment (findPenetrSegment): v2 of the support = v1 - v
    pos[0] = p1.v1[0] - 0.5f * p1.v[0]; pos[1] = p1.v1[1] - 0.5f * p1.v[1]; pos[2] = p1.v1[2] - 0.5f * p1.v[2];
    dir_out[0] = p1.v[0]; dir_out[1] = p1.v[1]; dir_out[2] = p1.v[2];
    depth = mpr_normalize(dir_out);
    return depth > 0.f;
  }
  mpr_normalize(dir);
  sup(dir, p2);
  dt = dot3(p2.v, dir);
  if (mpr_zero(dt) || dt < 0.f) return false;
  for (int k = 0; k < 3; k++) { va[k] = p1.v[k] - v0[k]; vb[k] = p2.v[k] - v0[k]; }
  cross(dir, va, vb);
  mpr_normalize(dir);
  if (dot3(dir, v0) > 0.f) {
    const MprSup t = p1; p1 = p2; p2 = t;
    dir[0] = -dir[0]; dir[1] = -dir[1]; dir[2] = -dir[2];
  }
  {
    int it = 0;
    for (;;) {
      if (++it > MPR_MAXIT) return false;
      sup(dir, p3);
      dt = dot3(p3.v, dir);
      if (mpr_zero(dt) || dt < 0.f) return false;
      bool cont = false;
      cross(va, p1.v, p3.v);
      dt = dot3(va, v0);
      if (dt < 0.f && !mpr_zero(dt)) { p2 = p3; cont = true; }
      if (!cont) {
        cross(va, p3.v, p2.v);
        dt = dot3(va, v0);
        if (dt < 0.f && !mpr_zero(dt)) { p1 = p3; cont = true; }
      }
      if (!cont) break;
      for (int k = 0; k < 3; k++) { va[k] = p1.v[k] - v0[k]; vb[k] = p2.v[k] - v0[k]; }
      cross(dir, va, vb);
      mpr_normalize(dir);
    }
  }
  // ---- refinePortal
  {
    int it = 0;
    for (;;) {
      if (++it > MPR_MAXIT) return false;
      mpr_portal_dir(p1, p2, p3, dir);
      dt = dot3(dir, p1.v);
      if (mpr_zero(dt) || dt > 0.f) break;               // portalEncapsulesOrigin
      sup(dir, v4);
      dt = dot3(v4.v, dir);
      if (!(mpr_zero(dt) || dt > 0.f) || mpr_reach_tol(p1, p2, p3, v4, dir)) return false;
      mpr_expand(v0, p1, p2, p3, v4);
    }
  }
  // ---- findPenetr
  {
    int it = 0;
    for (;;) {
      mpr_portal_dir(p1, p2, p3, dir);
      sup(dir, v4);
      if (mpr_reach_tol(p1, p2, p3, v4, dir) || it > MPR_MAXIT) break;
      mpr_expand(v0, p1, p2, p3, v4);
      it++;
    }
  }
  float pd[3];
  depth = sqrtf(mpr_tri_dist2(p1.v, p2.v, p3.v, pd));
  if (mpr_zero(pd[0]) && mpr_zero(pd[1]) && mpr_zero(pd[2])) { pd[0] = dir[0]; pd[1] = dir[1]; pd[2] = dir[2]; }
  mpr_normalize(pd);
  dir_out[0] = pd[0]; dir_out[1] = pd[1]; dir_out[2] = pd[2];
  // ---- findPos: barycentric coordinates of the origin in the portal tetrahedron
  {
    float b0, b1, b2, b3, vec[3];
    mpr_portal_dir(p1, p2, p3, dir);
    cross(vec, p1.v, p2.v); b0 = dot3(vec, p3.v);
    cross(vec, p3.v, p2.v); b1 = dot3(vec, v0);
    cross(vec, v0, p1.v);   b2 = dot3(vec, p3.v);
    cross(vec, p2.v, p1.v); b3 = dot3(vec, v0);
    float sum = b0 + b1 + b2 + b3;
    if (mpr_zero(sum) || sum < 0.f) {
      b0 = 0.f;
      cross(vec, p2.v, p3.v); b1 = dot3(vec, dir);
      cross(vec, p3.v, p1.v); b2 = dot3(vec, dir);
      cross(vec, p1.v, p2.v); b3 = dot3(vec, dir);
      sum = b1 + b2 + b3;
    }
    const float inv = 1.f / sum;
    for (int k = 0; k < 3; k++) {
      // p1' = sum b_i v1_i, p2' = sum b_i v2_i with v2_i = v1_i - v_i;  pos = (p1' + p2') / 2
      const float s1 = b0 * c1[k] + b1 * p1.v1[k] + b2 * p2.v1[k] + b3 * p3.v1[k];
      const float sv = b0 * v0[k] + b1 * p1.v[k] + b2 * p2.v[k] + b3 * p3.v[k];
      pos[k] = (s1 - 0.5f * sv) * inv;
    }
  }
  return isfinite(depth) && isfinite(pos[0]) && isfinite(pos[1]) && isfinite(pos[2]);
}

}  // namespace cosim
