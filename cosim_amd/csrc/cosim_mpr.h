// cosim_mpr.h — convex-convex narrowphase of the rollout kernel: Minkowski Portal Refinement (fp32 supports, fp64 portal).
//
// MuJoCo 3.2.7 sends every geom pair without an analytic routine (for the cosim robots: mesh / cylinder / box pairs)
// through libccd's ccdMPRPenetration (engine_collision_convex.c mjc_Convex -> mjc_MPRIteration; libccd 2.1 src/mpr.c is a
// third-party dependency, absent from the reference tree).  The routine below restates the published algorithm
// (G. Snethen, "XenoCollide", Game Programming Gems 7) with libccd's structure -- discoverPortal, refinePortal,
// findPenetr, findPos -- and its predicates, portal arithmetic in fp64 on fp32 support points, mpr_tolerance 1e-6 and at most
// 50 iterations per loop (MuJoCo's ccd_tolerance / ccd_iterations; libccd's refinePortal is uncapped, here every loop is
// bounded so that every wave reaches the end of the kernel).  oracle/cosim_oracle.c holds the fp64 twin.
//
// Two ways to run it, same code:
//   * lane-parallel: every lane owns one pair of primitives (box / cylinder / sphere supports are O(1));
//   * wave-cooperative (COOP): all 64 lanes run ONE pair with identical values and share the scan over a mesh hull's
//     vertices in the support function (control flow is uniform because the data is).
#pragma once

namespace cosim {

constexpr double MPR_EPS = 2.220446049250313e-16;
constexpr double MPR_TOL = 1e-6;
constexpr int MPR_MAXIT = 50;

struct CObj {            // one convex geom, world pose
  int kind;              // CS_GEOM_*
  float pos[3], q[4];    // primitive: geom frame; mesh: body frame (hull vertices are stored in body coordinates)
  float size[3];
  int adr, num;          // mesh: slice of the hull vertex array
  int map;               // mesh: first cell of the hull's support map (cosim_hullmap.h), or -1: scan the hull
  float center[3];       // mjccd_center
};
// hull arrays (vertices in body coordinates; CSR neighbour graph, used by the plane-hull routine)
struct HullGraph { const float* vert; const int* adr; const int* nbr; const float4* cell; const float4* cand; };
typedef double real;   // the portal arithmetic runs in fp64 (ill-conditioned for edge contacts); supports are fp32
__device__ __forceinline__ real mpr_dot(const real* a, const real* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void mpr_cross(real* r, const real* a, const real* b) {
  r[0] = a[1] * b[2] - a[2] * b[1]; r[1] = a[2] * b[0] - a[0] * b[2]; r[2] = a[0] * b[1] - a[1] * b[0];
}
// dir . vertex of every hull support query, one fixed sequence of roundings: the scans and the support-map walks must agree on
// near-ties, and the compiler contracts a plain a*x + b*y + c*z differently from loop to loop
__device__ __forceinline__ float hull_dot(const float* l, const float4& x) { return __builtin_fmaf(l[2], x.z, __builtin_fmaf(l[1], x.y, l[0] * x.x)); }
struct MprSup { real v[3]; float v1[3]; };  // Minkowski-difference support point v = s1(dir) - s2(-dir) (exact in fp64) and its fp32 s1 part

__device__ __forceinline__ bool mpr_zero(real x) { return fabs(x) < MPR_EPS; }
__device__ __forceinline__ bool mpr_eq(real a, real b) {
  const real ab = fabs(a - b);
  if (ab < MPR_EPS) return true;
  a = fabs(a); b = fabs(b);
  return b > a ? ab < MPR_EPS * b : ab < MPR_EPS * a;
}
__device__ __forceinline__ bool mpr_vzero(const real* v) { return mpr_eq(v[0], 0.) && mpr_eq(v[1], 0.) && mpr_eq(v[2], 0.); }
__device__ __forceinline__ float mpr_sign(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }   // mju_sign
__device__ __forceinline__ real mpr_normalize(real* v) {
  const real n = sqrt(mpr_dot(v, v));
  if (n < 1e-300) { v[0] = 1.; v[1] = 0.; v[2] = 0.; return 0.; }
  const real s = 1. / n;
  v[0] *= s; v[1] *= s; v[2] *= s;
  return n;
}

// mjccd_support: furthest point of the geom along the unit world direction
template <int GTM, bool COOP>
__device__ __forceinline__ void cobj_support(const CObj& o, const HullGraph& H, const float* dir, float* out, int ln) {
  const float* hull = H.vert;
  const float qi[4] = {o.q[0], -o.q[1], -o.q[2], -o.q[3]};
  float l[3], r[3] = {0.f, 0.f, 0.f};
  qrot(l, qi, dir);
  if (!COOP && (GTM & GT_MESH) && o.kind == CS_GEOM_MESH) {
    // every lane scans the whole hull for its own direction; the first maximum wins (lowest index among ties, like a sequential
    // scan and like the COOP path).  The lanes of a batch hold at most a few different hulls: they are served one hull at a time
    // with a wave-uniform base (readfirstlane), so the vertex loads are uniform-address loads, and four vertices are in flight per
    // trip (a dependent load per vertex made this loop cost ~600 cycles per vertex).
    // (Hill climbing on the neighbour graph from a direction-indexed seed was measured 2x slower here: its dependent,
    // lane-divergent loads do not overlap.)
    // A hull with a support map (cosim_hullmap.h) is not scanned: the lane walks the candidates of its direction's cell.
    float best = -3.0e38f;
    int bi = 0;
    bool pending = o.map < 0;
    for (unsigned long long pm = __ballot(pending); pm != 0ull; pm = __ballot(pending)) {
      const int src = __builtin_ctzll(pm);   // a lane that still waits: its hull is served now
      const int adr_u = __builtin_amdgcn_readlane(o.adr, src), num_u = __builtin_amdgcn_readlane(o.num, src);
      const bool mine = pending && o.adr == adr_u;
      const float4* v0 = reinterpret_cast<const float4*>(hull) + adr_u;
      for (int i = 0; i < num_u; i += 4) {
        float4 x[4];
#pragma unroll
        for (int k = 0; k < 4; k++) x[k] = v0[min(i + k, num_u - 1)];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const float t = hull_dot(l, x[k]);
          if (mine && i + k < num_u && t > best) { best = t; bi = i + k; }
        }
      }
      if (mine) pending = false;
    }
    float4 v;
    if (o.map < 0) v = reinterpret_cast<const float4*>(hull)[o.adr + bi];
    else {
      // the cell's record: header and the first HM_INLINE candidates, one trip; the overflow list only where a cell holds more
      const float4* rec = H.cell + (size_t)HM_REC * (o.map + support_cell(l));
      float4 x[HM_INLINE];
      const float4 hd = rec[0];
#pragma unroll
      for (int k = 0; k < HM_INLINE; k++) x[k] = rec[1 + k];
      const int cn = __float_as_int(hd.x);
      v = x[0];
#pragma unroll
      for (int k = 0; k < HM_INLINE; k++) {
        const float t = hull_dot(l, x[k]);
        if (k < cn && t > best) { best = t; v = x[k]; }
      }
      if (cn > HM_INLINE) {
        const float4* cp = H.cand + __float_as_int(hd.y);
        const int rest = cn - HM_INLINE;
        for (int i = 0; i < rest; i += 4) {
#pragma unroll
          for (int k = 0; k < 4; k++) x[k] = cp[min(i + k, rest - 1)];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const float t = hull_dot(l, x[k]);
            if (i + k < rest && t > best) { best = t; v = x[k]; }
          }
        }
      }
    }
    r[0] = v.x; r[1] = v.y; r[2] = v.z;
  } else if (COOP && (GTM & GT_MESH) && o.kind == CS_GEOM_MESH) {
    // all 64 lanes share one scan over the hull (16 bytes per vertex: one load each).  COOP_K vertices per lane are in flight per trip (a dependent trip per vertex made a 696-vertex
    // wheel hull cost eleven memory latencies per support query); the wave-wide maximum and its lowest index come from DPP minima
    // (the index as a float: exact below 2^24), not from six LDS-crossbar shuffles.
    constexpr int COOP_K = 6;
    const int num = __builtin_amdgcn_readfirstlane(o.num);
    const float4* v0 = reinterpret_cast<const float4*>(hull) + __builtin_amdgcn_readfirstlane(o.adr);
    float best = -3.0e38f, bx = 0.f, by = 0.f, bz = 0.f;
    int besti = 0x7fffffff;
    const int map = __builtin_amdgcn_readfirstlane(o.map);
    if (map >= 0) {
      // support map: the candidates of the direction's cell, one per lane (a few; a cell facing a flat side of the hull holds up
      // to ~90); besti is the vertex's index in the hull, so ties fall like in the scan
      const float4* rec = H.cell + (size_t)HM_REC * (map + __builtin_amdgcn_readfirstlane(support_cell(l)));
      const float4 hd = rec[0], x0 = rec[1 + (ln & (HM_INLINE - 1))];   // one trip: the header and the inline candidates (lane k < 4: candidate k)
      const int cn = __builtin_amdgcn_readfirstlane(__float_as_int(hd.x));
      {
        const float t = hull_dot(l, x0);
        if (ln < HM_INLINE && ln < cn) { best = t; besti = __float_as_int(x0.w); bx = x0.x; by = x0.y; bz = x0.z; }
      }
      if (cn > HM_INLINE) {
        const float4* cp = H.cand + __builtin_amdgcn_readfirstlane(__float_as_int(hd.y));
        const int rest = cn - HM_INLINE;
        for (int i0 = 0; i0 < rest; i0 += 64) {
          const float4 x = cp[min(i0 + ln, rest - 1)];
          const float t = hull_dot(l, x);
          if (i0 + ln < rest && t > best) { best = t; besti = __float_as_int(x.w); bx = x.x; by = x.y; bz = x.z; }
        }
      }
    } else
    for (int i0 = 0; i0 < num; i0 += 64 * COOP_K) {
      float4 x[COOP_K];
#pragma unroll
      for (int k = 0; k < COOP_K; k++) x[k] = v0[min(i0 + ln + 64 * k, num - 1)];
#pragma unroll
      for (int k = 0; k < COOP_K; k++) {
        const int i = i0 + ln + 64 * k;
        const float t = hull_dot(l, x[k]);
        if (i < num && t > best) { best = t; besti = i; bx = x[k].x; by = x[k].y; bz = x[k].z; }
      }
    }
    // the winner's coordinates come out of its lane's registers (vertex i lives in lane i & 63), not from a dependent load; one lane
    // at the maximum is the usual case, ties take the lowest index like a sequential scan
    const float bmax = -wave_min(-best);
    const unsigned long long tm = __ballot(best == bmax);
    int src = __builtin_ctzll(tm);
    if (tm & (tm - 1)) {   // (a scanned vertex i lives in lane i & 63; a support-map candidate wherever its list position put it)
      const float imin = wave_min(best == bmax ? (float)besti : 3.0e38f);
      src = __builtin_ctzll(__ballot(best == bmax && (float)besti == imin));
    }
    src = __builtin_amdgcn_readfirstlane(src);
    r[0] = rl(bx, src); r[1] = rl(by, src); r[2] = rl(bz, src);
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

// The six support queries along +-x, +-y, +-z of a hull (its exact axis-aligned box) as ONE cooperative scan: three dot products per
// vertex serve all six arg-maxima, with the same comparisons, tie rule and final rotation as six cobj_support calls (same bits).
__device__ __forceinline__ void cobj_box_coop(const CObj& o, const HullGraph& H, float* lo, float* hi, int ln) {
  const float qi[4] = {o.q[0], -o.q[1], -o.q[2], -o.q[3]};
  float l[3][3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    float d[3] = {0.f, 0.f, 0.f};
    d[k] = 1.f;
    qrot(l[k], qi, d);
  }
  const int num = __builtin_amdgcn_readfirstlane(o.num);
  const float4* v0 = reinterpret_cast<const float4*>(H.vert) + __builtin_amdgcn_readfirstlane(o.adr);
  float best[6], bv[6][3];
  int besti[6];
#pragma unroll
  for (int s = 0; s < 6; s++) { best[s] = -3.0e38f; besti[s] = 0x7fffffff; bv[s][0] = bv[s][1] = bv[s][2] = 0.f; }
  const int map = __builtin_amdgcn_readfirstlane(o.map);
  if (map >= 0) {
    // support map: six cells, their candidates one per lane, all six loads in flight together
    int cb[6], cn[6], nmax = 0;
    {
      float4 hd[6], x[6];   // one trip: six headers and the inline candidates (lane k < 4: candidate k of each cell)
#pragma unroll
      for (int s = 0; s < 6; s++) {
        const float d[3] = {(s & 1) ? -l[s >> 1][0] : l[s >> 1][0], (s & 1) ? -l[s >> 1][1] : l[s >> 1][1], (s & 1) ? -l[s >> 1][2] : l[s >> 1][2]};
        const float4* rec = H.cell + (size_t)HM_REC * (map + __builtin_amdgcn_readfirstlane(support_cell(d)));
        hd[s] = rec[0]; x[s] = rec[1 + (ln & (HM_INLINE - 1))];
      }
#pragma unroll
      for (int s = 0; s < 6; s++) {
        cn[s] = __builtin_amdgcn_readfirstlane(__float_as_int(hd[s].x)); cb[s] = __builtin_amdgcn_readfirstlane(__float_as_int(hd[s].y));
        nmax = max(nmax, cn[s] - HM_INLINE);
        const int a = s >> 1;
        const float t0 = hull_dot(l[a], x[s]);
        const float t = (s & 1) ? -t0 : t0;
        if (ln < HM_INLINE && ln < cn[s]) { best[s] = t; besti[s] = __float_as_int(x[s].w); bv[s][0] = x[s].x; bv[s][1] = x[s].y; bv[s][2] = x[s].z; }
      }
    }
    for (int i0 = 0; i0 < nmax; i0 += 64) {   // overflow lists (cells facing a flat side of the hull)
      float4 x[6];
#pragma unroll
      for (int s = 0; s < 6; s++) x[s] = H.cand[cb[s] + min(i0 + ln, max(cn[s] - HM_INLINE, 1) - 1)];
#pragma unroll
      for (int s = 0; s < 6; s++) {
        const int a = s >> 1;
        const float t0 = hull_dot(l[a], x[s]);
        const float t = (s & 1) ? -t0 : t0;
        if (i0 + ln < cn[s] - HM_INLINE && t > best[s]) { best[s] = t; besti[s] = __float_as_int(x[s].w); bv[s][0] = x[s].x; bv[s][1] = x[s].y; bv[s][2] = x[s].z; }
      }
    }
  } else
  for (int i0 = 0; i0 < num; i0 += 64 * 4) {
    float4 x[4];
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = v0[min(i0 + ln + 64 * k, num - 1)];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int i = i0 + ln + 64 * k;
#pragma unroll
      for (int a = 0; a < 3; a++) {
        // the query along -e_a rotates to exactly -l[a] (qrot is odd in its vector argument), whose dot products are exactly -t
        const float t = hull_dot(l[a], x[k]);
        if (i < num && t > best[2 * a]) { best[2 * a] = t; besti[2 * a] = i; bv[2 * a][0] = x[k].x; bv[2 * a][1] = x[k].y; bv[2 * a][2] = x[k].z; }
        const float u = -t;
        if (i < num && u > best[2 * a + 1]) { best[2 * a + 1] = u; besti[2 * a + 1] = i; bv[2 * a + 1][0] = x[k].x; bv[2 * a + 1][1] = x[k].y; bv[2 * a + 1][2] = x[k].z; }
      }
    }
  }
#pragma unroll
  for (int s = 0; s < 6; s++) {
    const float bmax = -wave_min(-best[s]);
    const unsigned long long tm = __ballot(best[s] == bmax);
    int src = __builtin_ctzll(tm);
    if (tm & (tm - 1)) {
      const float imin = wave_min(best[s] == bmax ? (float)besti[s] : 3.0e38f);
      src = __builtin_ctzll(__ballot(best[s] == bmax && (float)besti[s] == imin));
    }
    src = __builtin_amdgcn_readfirstlane(src);
    const float r[3] = {rl(bv[s][0], src), rl(bv[s][1], src), rl(bv[s][2], src)};
    float w[3];
    qrot(w, o.q, r);
    const int a = s >> 1;
    if (s & 1) lo[a] = o.pos[a] + w[a]; else hi[a] = o.pos[a] + w[a];
  }
}

template <int GTM, bool COOP>
struct MprPair {
  const CObj &a, &b;
  const HullGraph& hull;
  int ln;
  __device__ __forceinline__ void operator()(const real* dir, MprSup& s) const {   // __ccdSupport
    const float fd[3] = {(float)dir[0], (float)dir[1], (float)dir[2]}, nd[3] = {-fd[0], -fd[1], -fd[2]};
    float v1[3], v2[3];
    cobj_support<GTM, COOP>(a, hull, fd, v1, ln);
    cobj_support<GTM, COOP>(b, hull, nd, v2, ln);
    for (int k = 0; k < 3; k++) { s.v1[k] = v1[k]; s.v[k] = (real)v1[k] - (real)v2[k]; }
  }
};

__device__ __forceinline__ void mpr_portal_dir(const MprSup& p1, const MprSup& p2, const MprSup& p3, real* dir) {
  const real a[3] = {p2.v[0] - p1.v[0], p2.v[1] - p1.v[1], p2.v[2] - p1.v[2]}, b[3] = {p3.v[0] - p1.v[0], p3.v[1] - p1.v[1], p3.v[2] - p1.v[2]};
  mpr_cross(dir, a, b);
  mpr_normalize(dir);
}
__device__ __forceinline__ bool mpr_reach_tol(const MprSup& p1, const MprSup& p2, const MprSup& p3, const MprSup& v4, const real* dir) {
  const real dv4 = mpr_dot(v4.v, dir);
  const real d = fmin(dv4 - mpr_dot(p1.v, dir), fmin(dv4 - mpr_dot(p2.v, dir), dv4 - mpr_dot(p3.v, dir)));
  return mpr_eq(d, MPR_TOL) || d < MPR_TOL;
}
// Portal vertices are replaced by predicated field-wise moves, never by `if (c) p1 = v4; else p3 = v4;`: the compiler merges such
// stores into one store through a selected POINTER, which pins all the portal's support points in scratch memory for the whole
// routine (measured: the MPR loops then ran at scratch latency, ~1000 cycles per iteration).
__device__ __forceinline__ void sup_copy_if(MprSup& d, const MprSup& s, bool c) {
#pragma unroll
  for (int k = 0; k < 3; k++) { d.v[k] = c ? s.v[k] : d.v[k]; d.v1[k] = c ? s.v1[k] : d.v1[k]; }
}
__device__ __forceinline__ void sup_swap_if(MprSup& a, MprSup& b, bool c) {
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const real t = c ? b.v[k] : a.v[k];
    b.v[k] = c ? a.v[k] : b.v[k];
    a.v[k] = t;
    const float u = c ? b.v1[k] : a.v1[k];
    b.v1[k] = c ? a.v1[k] : b.v1[k];
    a.v1[k] = u;
  }
}
__device__ __forceinline__ void mpr_expand(const real* v0, MprSup& p1, MprSup& p2, MprSup& p3, const MprSup& v4) {
  real v4v0[3];
  mpr_cross(v4v0, v4.v, v0);
  const bool a = mpr_dot(p1.v, v4v0) > 0., b = mpr_dot(p2.v, v4v0) > 0., c = mpr_dot(p3.v, v4v0) > 0.;
  // a: (b ? p1 : p3) = v4;  !a: (c ? p2 : p1) = v4
  sup_copy_if(p1, v4, a ? b : !c);
  sup_copy_if(p2, v4, !a && c);
  sup_copy_if(p3, v4, a && !b);
}
__device__ __forceinline__ real mpr_seg_dist2(const real* x0, const real* b, real* wit) {   // ccdVec3PointSegmentDist2, P = origin
  const real d[3] = {b[0] - x0[0], b[1] - x0[1], b[2] - x0[2]};
  const real dd = mpr_dot(d, d);
  const real t = dd > 0. ? -mpr_dot(x0, d) / dd : 0.;
  if (t < 0. || mpr_zero(t)) { wit[0] = x0[0]; wit[1] = x0[1]; wit[2] = x0[2]; return mpr_dot(x0, x0); }
  if (t > 1. || mpr_eq(t, 1.)) { wit[0] = b[0]; wit[1] = b[1]; wit[2] = b[2]; return mpr_dot(b, b); }
  wit[0] = x0[0] + t * d[0]; wit[1] = x0[1] + t * d[1]; wit[2] = x0[2] + t * d[2];
  return mpr_dot(wit, wit);
}
__device__ __forceinline__ real mpr_tri_dist2(const real* x0, const real* B, const real* C, real* wit) {   // ccdVec3PointTriDist2, P = origin
  const real d1[3] = {B[0] - x0[0], B[1] - x0[1], B[2] - x0[2]}, d2[3] = {C[0] - x0[0], C[1] - x0[1], C[2] - x0[2]};
  const real v = mpr_dot(d1, d1), w = mpr_dot(d2, d2), p = mpr_dot(x0, d1), q = mpr_dot(x0, d2), r = mpr_dot(d1, d2);
  const real den = w * v - r * r;
  const real s = (q * r - w * p) / den, t = (-s * r - q) / w;   // degenerate triangle: NaN fails every test below
  if ((mpr_zero(s) || s > 0.) && (mpr_eq(s, 1.) || s < 1.) && (mpr_zero(t) || t > 0.) && (mpr_eq(t, 1.) || t < 1.) &&
      (mpr_eq(t + s, 1.) || t + s < 1.)) {
    wit[0] = x0[0] + s * d1[0] + t * d2[0]; wit[1] = x0[1] + s * d1[1] + t * d2[1]; wit[2] = x0[2] + s * d1[2] + t * d2[2];
    return mpr_dot(wit, wit);
  }
  real w2[3];
  real dist = mpr_seg_dist2(x0, B, wit);
  real dist2 = mpr_seg_dist2(x0, C, w2);
  if (dist2 < dist) { dist = dist2; wit[0] = w2[0]; wit[1] = w2[1]; wit[2] = w2[2]; }
  dist2 = mpr_seg_dist2(B, C, w2);
  if (dist2 < dist) { dist = dist2; wit[0] = w2[0]; wit[1] = w2[1]; wit[2] = w2[2]; }
  return dist;
}

// The first two exits of discoverPortal (no support point beyond the origin along -v0, none along the first portal normal): false
// = the full routine would return "no penetration" at exactly these points, true = undecided.  Most prisms under a geom's box are
// decided here; the undecided ones are compacted into dense batches for the full routine.
template <class SUP>
__device__ __forceinline__ bool mpr_probe(const SUP& sup, const float* c1, const float* c2) {
  MprSup p1, p2;
  real v0[3] = {(real)c1[0] - (real)c2[0], (real)c1[1] - (real)c2[1], (real)c1[2] - (real)c2[2]};
  real dir[3], dt;
  if (mpr_vzero(v0)) v0[0] += MPR_EPS * 10.;
  dir[0] = -v0[0]; dir[1] = -v0[1]; dir[2] = -v0[2];
  mpr_normalize(dir);
  sup(dir, p1);
  dt = mpr_dot(p1.v, dir);
  if (mpr_zero(dt) || dt < 0.) return false;
  mpr_cross(dir, v0, p1.v);
  if (mpr_zero(mpr_dot(dir, dir))) return true;
  mpr_normalize(dir);
  sup(dir, p2);
  dt = mpr_dot(p2.v, dir);
  return !(mpr_zero(dt) || dt < 0.);
}

// ccdMPRPenetration: true when the geoms penetrate; depth, dir (geom1 -> geom2) and pos as libccd returns them.
template <class SUP>
__device__ __forceinline__ bool mpr_penetration(const SUP& sup, const float* c1, const float* c2, float& depth_out, float* dir_f, float* pos_f,
                                                int* n_iter = nullptr) {   // n_iter: diagnostic builds count loop iterations
  MprSup p1, p2, p3, v4;
  real depth, dir_out[3], pos[3];
  real v0[3] = {(real)c1[0] - (real)c2[0], (real)c1[1] - (real)c2[1], (real)c1[2] - (real)c2[2]};
  real dir[3], va[3], vb[3], dt;
  // ---- discoverPortal
  if (mpr_vzero(v0)) v0[0] += MPR_EPS * 10.;
  dir[0] = -v0[0]; dir[1] = -v0[1]; dir[2] = -v0[2];
  mpr_normalize(dir);
  sup(dir, p1);
  dt = mpr_dot(p1.v, dir);
  if (mpr_zero(dt) || dt < 0.) return false;
  mpr_cross(dir, v0, p1.v);
  if (mpr_zero(mpr_dot(dir, dir))) {
    if (mpr_vzero(p1.v)) return false;                    // touching: depth 0, no direction -> MuJoCo drops it
    // origin on the v0-v1 segment (findPenetrSegment): v2 of the support = v1 - v
    pos[0] = (real)p1.v1[0] - 0.5 * p1.v[0]; pos[1] = (real)p1.v1[1] - 0.5 * p1.v[1]; pos[2] = (real)p1.v1[2] - 0.5 * p1.v[2];
    dir_out[0] = p1.v[0]; dir_out[1] = p1.v[1]; dir_out[2] = p1.v[2];
    depth = mpr_normalize(dir_out);
    depth_out = (float)depth;
    for (int k = 0; k < 3; k++) { dir_f[k] = (float)dir_out[k]; pos_f[k] = (float)pos[k]; }
    return depth > 0.;
  }
  mpr_normalize(dir);
  sup(dir, p2);
  dt = mpr_dot(p2.v, dir);
  if (mpr_zero(dt) || dt < 0.) return false;
  for (int k = 0; k < 3; k++) { va[k] = p1.v[k] - v0[k]; vb[k] = p2.v[k] - v0[k]; }
  mpr_cross(dir, va, vb);
  mpr_normalize(dir);
  {
    const bool flip = mpr_dot(dir, v0) > 0.;
    sup_swap_if(p1, p2, flip);
    for (int k = 0; k < 3; k++) dir[k] = flip ? -dir[k] : dir[k];
  }
  {
    int it = 0;
    for (;;) {
      if (++it > MPR_MAXIT) return false;
      if (n_iter) ++*n_iter;
      sup(dir, p3);
      dt = mpr_dot(p3.v, dir);
      if (mpr_zero(dt) || dt < 0.) return false;
      mpr_cross(va, p1.v, p3.v);
      dt = mpr_dot(va, v0);
      const bool ca = dt < 0. && !mpr_zero(dt);            // origin outside (v1, v0, v3): p2 = p3
      mpr_cross(va, p3.v, p2.v);
      dt = mpr_dot(va, v0);
      const bool cb = !ca && dt < 0. && !mpr_zero(dt);     // else outside (v3, v0, v2): p1 = p3
      sup_copy_if(p2, p3, ca);
      sup_copy_if(p1, p3, cb);
      if (!(ca || cb)) break;
      for (int k = 0; k < 3; k++) { va[k] = p1.v[k] - v0[k]; vb[k] = p2.v[k] - v0[k]; }
      mpr_cross(dir, va, vb);
      mpr_normalize(dir);
    }
  }
  // ---- refinePortal
  {
    int it = 0;
    for (;;) {
      if (++it > MPR_MAXIT) return false;
      if (n_iter) ++*n_iter;
      mpr_portal_dir(p1, p2, p3, dir);
      dt = mpr_dot(dir, p1.v);
      if (mpr_zero(dt) || dt > 0.) break;               // portalEncapsulesOrigin
      sup(dir, v4);
      dt = mpr_dot(v4.v, dir);
      if (!(mpr_zero(dt) || dt > 0.) || mpr_reach_tol(p1, p2, p3, v4, dir)) return false;
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
      if (n_iter) ++*n_iter;
    }
  }
  real pd[3];
  depth = sqrt(mpr_tri_dist2(p1.v, p2.v, p3.v, pd));
  if (mpr_zero(pd[0]) && mpr_zero(pd[1]) && mpr_zero(pd[2])) { pd[0] = dir[0]; pd[1] = dir[1]; pd[2] = dir[2]; }
  mpr_normalize(pd);
  dir_out[0] = pd[0]; dir_out[1] = pd[1]; dir_out[2] = pd[2];
  // ---- findPos: barycentric coordinates of the origin in the portal tetrahedron
  {
    real b0, b1, b2, b3, vec[3];
    mpr_portal_dir(p1, p2, p3, dir);
    mpr_cross(vec, p1.v, p2.v); b0 = mpr_dot(vec, p3.v);
    mpr_cross(vec, p3.v, p2.v); b1 = mpr_dot(vec, v0);
    mpr_cross(vec, v0, p1.v);   b2 = mpr_dot(vec, p3.v);
    mpr_cross(vec, p2.v, p1.v); b3 = mpr_dot(vec, v0);
    real sum = b0 + b1 + b2 + b3;
    if (mpr_zero(sum) || sum < 0.) {
      b0 = 0.;
      mpr_cross(vec, p2.v, p3.v); b1 = mpr_dot(vec, dir);
      mpr_cross(vec, p3.v, p1.v); b2 = mpr_dot(vec, dir);
      mpr_cross(vec, p1.v, p2.v); b3 = mpr_dot(vec, dir);
      sum = b1 + b2 + b3;
    }
    const real inv = 1. / sum;
    for (int k = 0; k < 3; k++) {
      // p1' = sum b_i v1_i, p2' = sum b_i v2_i with v2_i = v1_i - v_i;  pos = (p1' + p2') / 2
      const real s1 = b0 * (real)c1[k] + b1 * (real)p1.v1[k] + b2 * (real)p2.v1[k] + b3 * (real)p3.v1[k];
      const real sv = b0 * v0[k] + b1 * p1.v[k] + b2 * p2.v[k] + b3 * p3.v[k];
      pos[k] = (s1 - 0.5 * sv) * inv;
    }
  }
  depth_out = (float)depth;
  for (int k = 0; k < 3; k++) { dir_f[k] = (float)dir_out[k]; pos_f[k] = (float)pos[k]; }
  return isfinite(depth) && isfinite(pos[0]) && isfinite(pos[1]) && isfinite(pos[2]);
}

// ------------------------------------------------------------------------------------------------ heightfield prisms
// mjc_ConvexHField (engine_collision_convex.c): one prism per terrain triangle (top = triangle, bottom at -size[3]),
// MPR between the prism (geom1) and the robot geom (geom2), at most one contact per prism.
struct PrismObj { float x[3], y[3], zt[3], zb; };   // strip window: vertex k at (x[k], y[k]), top zt[k], common bottom zb

__device__ __forceinline__ void prism_support(const PrismObj& P, const float* dir, float* out) {   // bottom triangle for dir z < 0, else top
  const bool bot = dir[2] < 0.f;
  const float z0 = bot ? P.zb : P.zt[0], z1 = bot ? P.zb : P.zt[1], z2 = bot ? P.zb : P.zt[2];
  const float t0 = P.x[0] * dir[0] + P.y[0] * dir[1] + z0 * dir[2], t1 = P.x[1] * dir[0] + P.y[1] * dir[1] + z1 * dir[2],
              t2 = P.x[2] * dir[0] + P.y[2] * dir[1] + z2 * dir[2];
  const bool b1 = t1 > t0, b2 = t2 > (b1 ? t1 : t0);
  out[0] = b2 ? P.x[2] : (b1 ? P.x[1] : P.x[0]);
  out[1] = b2 ? P.y[2] : (b1 ? P.y[1] : P.y[0]);
  out[2] = b2 ? z2 : (b1 ? z1 : z0);
}

template <int GTM, bool COOP>
struct MprPrismGeom {
  const PrismObj& P;
  const CObj& g;
  const HullGraph& hull;
  int ln;
  __device__ __forceinline__ void operator()(const real* dir, MprSup& s) const {
    const float fd[3] = {(float)dir[0], (float)dir[1], (float)dir[2]}, nd[3] = {-fd[0], -fd[1], -fd[2]};
    float v1[3], v2[3];
    prism_support(P, fd, v1);
    cobj_support<GTM, COOP>(g, hull, nd, v2, ln);
    for (int k = 0; k < 3; k++) { s.v1[k] = v1[k]; s.v[k] = (real)v1[k] - (real)v2[k]; }
  }
};

// All prisms under one geom.  Coordinates are base-relative (the frame the kernel keeps poses in); T.ox / T.oy carry the
// base's offset on the field in fp64.  emit(dist, pos, normal) is called once per penetrated prism, in strip order.
template <int GTM, bool COOP, class EMIT>
__device__ __forceinline__ void hfield_geom(const Terrain& T, const CObj& o, const float* ctr, float rb, float margin, float base,
                                            const HullGraph& hull, int ln, const EMIT& emit, unsigned long long* prof = nullptr) {
  // box-sphere early outs
  unsigned long long pt_ = 0;
  if (prof) pt_ = __builtin_amdgcn_s_memtime();
  const double lx = (double)ctr[0] + T.ox, ly = (double)ctr[1] + T.oy;
  if ((double)T.sx < lx - rb - margin || -(double)T.sx > lx + rb + margin || (double)T.sy < ly - rb - margin || -(double)T.sy > ly + rb + margin) return;
  if (T.sz < ctr[2] - T.gz - rb - margin || -base > ctr[2] - T.gz + rb + margin) return;
  // axis-aligned box of the geom through its support function
  float lo[3], hi[3];
  if constexpr (COOP) cobj_box_coop(o, hull, lo, hi, ln);
  else {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      float d[3] = {0.f, 0.f, 0.f}, p[3];
      d[k] = 1.f;
      cobj_support<GTM, COOP>(o, hull, d, p, ln);
      hi[k] = p[k];
      d[k] = -1.f;
      cobj_support<GTM, COOP>(o, hull, d, p, ln);
      lo[k] = p[k];
    }
  }
  if (prof) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); prof[0] += t_ - pt_; prof[3] += 1; }
  const double x0 = (double)lo[0] + T.ox, x1 = (double)hi[0] + T.ox, y0 = (double)lo[1] + T.oy, y1 = (double)hi[1] + T.oy;
  if (x0 - margin > T.sx || x1 + margin < -T.sx || y0 - margin > T.sy || y1 + margin < -T.sy || lo[2] - T.gz - margin > T.sz ||
      hi[2] - T.gz + margin < -base) return;
  int cmin = (int)floor((x0 + T.sx) / T.dx), cmax = (int)ceil((x1 + T.sx) / T.dx);
  int rmin = (int)floor((y0 + T.sy) / T.dy), rmax = (int)ceil((y1 + T.sy) / T.dy);
  cmin = max(cmin, 0); rmin = max(rmin, 0); cmax = min(cmax, T.ncol - 1); rmax = min(rmax, T.nrow - 1);
  PrismObj P;
  P.zb = T.gz - base;
  for (int k = 0; k < 3; k++) { P.x[k] = 0.f; P.y[k] = 0.f; P.zt[k] = 0.f; }
  int cnt = 0;
  for (int r = rmin; r < rmax; r++) {
    int nvert = 0;
    for (int c = cmin; c <= cmax; c++) {
      const float xc = (float)(c * T.dx - (double)T.sx - T.ox);
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int rr = r + 1 - i;   // strip order: (r+1, c) then (r, c)
        P.x[0] = P.x[1]; P.x[1] = P.x[2]; P.y[0] = P.y[1]; P.y[1] = P.y[2]; P.zt[0] = P.zt[1]; P.zt[1] = P.zt[2];
        P.x[2] = xc;
        P.y[2] = (float)(rr * T.dy - (double)T.sy - T.oy);
        P.zt[2] = T.data[rr * T.ncol + c] * T.sz + T.gz + margin;
        if (++nvert <= 2) continue;
        if (P.zt[0] < lo[2] && P.zt[1] < lo[2] && P.zt[2] < lo[2]) continue;
        if (cnt >= 50) continue;   // mjMAXCONPAIR
        const float c1[3] = {(P.x[0] + P.x[1] + P.x[2]) * (1.f / 3.f), (P.y[0] + P.y[1] + P.y[2]) * (1.f / 3.f),
                             (P.zt[0] + P.zt[1] + P.zt[2] + 3.f * P.zb) * (1.f / 6.f)};
        const MprPrismGeom<GTM, COOP> sup{P, o, hull, ln};
        float depth = 0.f, n[3] = {0.f, 0.f, 1.f}, pos[3] = {0.f, 0.f, 0.f};
        int nit_ = 0;
        if (prof) pt_ = __builtin_amdgcn_s_memtime();
        const bool hit_ = mpr_penetration(sup, c1, o.center, depth, n, pos, prof ? &nit_ : nullptr) && (n[0] != 0.f || n[1] != 0.f || n[2] != 0.f);
        if (prof) { prof[1] += 1; prof[2] += __builtin_amdgcn_s_memtime() - pt_; prof[4] += hit_ ? 1 : 0; prof[5] += nit_; }
        if (hit_) {
          emit(margin - depth, pos, n);
          cnt++;
        }
      }
    }
  }
}

}  // namespace cosim
