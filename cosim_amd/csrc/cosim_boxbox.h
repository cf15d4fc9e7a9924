// cosim_boxbox.h — box-box narrowphase for the robot-robot pairs (MuJoCo's mjc_BoxBox, engine_collision_box.c): up to eight contacts
// per pair where MPR gives one.  humanoid_p_v0.xml:33,40,110,139 are the only box geoms that meet (five pairs after the filters).
//
// One pair at a time, the wave in step: every lane runs the 15-axis separating-axis search on the same two boxes, then
//   face case: lane q < 4 tests incident vertex q, lane 4 + q rectangle corner q, lane 8 + 4 q + e incident edge q against side e of
//              the reference face; the lanes that hold a vertex of the clipped polygon return true (the caller compacts them in lane
//              order: the same enumeration order as oracle/cosim_oracle.c box_box_points);
//   edge case: lane 0 returns the midpoint of the two edges' closest points.
// The algorithm and its flags ("restated from memory of the routine's structure", parity unpinned) are those of the oracle's
// box_box_points; this is its fp32, lane-parallel form.
#pragma once

namespace cosim {

// element k of three, as a blend with 0 / 1 weights (exact): a chain of selects over array elements is turned into an indexed load by
// the compiler, which sends the arrays to scratch
__device__ __forceinline__ void bb_pick(float* o, const float (*ax)[3], int k) {
  const float w0 = k == 0 ? 1.f : 0.f, w1 = k == 1 ? 1.f : 0.f, w2 = k == 2 ? 1.f : 0.f;
#pragma unroll
  for (int c = 0; c < 3; c++) o[c] = w0 * ax[0][c] + w1 * ax[1][c] + w2 * ax[2][c];
}
__device__ __forceinline__ float bb_pick1(const float* s, int k) {
  return (k == 0 ? 1.f : 0.f) * s[0] + (k == 1 ? 1.f : 0.f) * s[1] + (k == 2 ? 1.f : 0.f) * s[2];
}

// p, q (world quaternion), s (half sizes) of both boxes; returns whether this lane holds a contact: cpos, cdist; nrm (box 1 -> box 2)
// is the same on all lanes
__device__ __forceinline__ bool box_box_lane(const float* p1, const float* q1, const float* s1, const float* p2, const float* q2, const float* s2,
                                             float margin, int ln, float* cpos, float& cdist, float* nrm) {
  float m1[9], m2[9], A[3][3], B[3][3];
  q2m(m1, q1);
  q2m(m2, q2);
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int c = 0; c < 3; c++) { A[k][c] = m1[3 * c + k]; B[k][c] = m2[3 * c + k]; }
  const float d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  float best = -3.0e38f, bax[3] = {0.f, 0.f, 1.f};
  int code = -1;
  bool sep = false;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    float r = s1[i];
#pragma unroll
    for (int k = 0; k < 3; k++) r += s2[k] * fabsf(dot3(B[k], A[i]));
    const float s = fabsf(dot3(d, A[i])) - r;
    sep |= s > margin;
    if (s > best) { best = s; code = i; }
  }
#pragma unroll
  for (int j = 0; j < 3; j++) {
    float r = s2[j];
#pragma unroll
    for (int k = 0; k < 3; k++) r += s1[k] * fabsf(dot3(A[k], B[j]));
    const float s = fabsf(dot3(d, B[j])) - r;
    sep |= s > margin;
    if (s > best) { best = s; code = 3 + j; }
  }
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      float ax[3];
      cross(ax, A[i], B[j]);
      const float l = sqrtf(dot3(ax, ax));
      if (l >= 1e-6f) {
        const float il = 1.f / l;
        for (int k = 0; k < 3; k++) ax[k] *= il;
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < 3; k++) r += s1[k] * fabsf(dot3(A[k], ax)) + s2[k] * fabsf(dot3(B[k], ax));
        const float s = fabsf(dot3(d, ax)) - r;
        sep |= s > margin;
        if (s > best) { best = s; code = 6 + 3 * i + j; bax[0] = ax[0]; bax[1] = ax[1]; bax[2] = ax[2]; }   // the oracle's 1e-12 bias is below fp32 resolution
      }
    }
  if (sep || code < 0) return false;
  if (code >= 6) {
    const int i = (code - 6) / 3, j = (code - 6) - 3 * i;
    const float sg = dot3(bax, d) < 0.f ? -1.f : 1.f;
    float pa[3], pb[3], ua[3], ub[3];
    for (int k = 0; k < 3; k++) { nrm[k] = sg * bax[k]; pa[k] = p1[k]; pb[k] = p2[k]; }
    bb_pick(ua, A, i);
    bb_pick(ub, B, j);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float ta = k == i ? 0.f : (dot3(nrm, A[k]) < 0.f ? -s1[k] : s1[k]);
      const float tb = k == j ? 0.f : (dot3(nrm, B[k]) < 0.f ? -s2[k] : s2[k]);
      for (int c = 0; c < 3; c++) { pa[c] += ta * A[k][c]; pb[c] -= tb * B[k][c]; }
    }
    const float pp[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
    const float uaub = dot3(ua, ub), q1_ = dot3(ua, pp), q2_ = -dot3(ub, pp), den = 1.f - uaub * uaub;
    const float al = (q1_ + uaub * q2_) / den, be = (uaub * q1_ + q2_) / den;
    for (int k = 0; k < 3; k++) cpos[k] = 0.5f * (pa[k] + al * ua[k] + pb[k] + be * ub[k]);
    cdist = best;
    return ln == 0;
  }
  // face case: reference box R owns the face, O is the other one
  const bool ref2 = code >= 3;
  const int a = ref2 ? code - 3 : code, a1 = a == 2 ? 0 : a + 1, a2 = a1 == 2 ? 0 : a1 + 1;
  float pr[3], po[3], sr[3], so[3], Rr[3][3], Ro[3][3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    pr[k] = ref2 ? p2[k] : p1[k]; po[k] = ref2 ? p1[k] : p2[k]; sr[k] = ref2 ? s2[k] : s1[k]; so[k] = ref2 ? s1[k] : s2[k];
#pragma unroll
    for (int c = 0; c < 3; c++) { Rr[k][c] = ref2 ? B[k][c] : A[k][c]; Ro[k][c] = ref2 ? A[k][c] : B[k][c]; }
  }
  float ra[3], t1[3], t2[3], nr[3];
  bb_pick(ra, Rr, a);
  bb_pick(t1, Rr, a1);
  bb_pick(t2, Rr, a2);
  const float dro[3] = {po[0] - pr[0], po[1] - pr[1], po[2] - pr[2]};
  const float sg = dot3(dro, ra) < 0.f ? -1.f : 1.f;
  for (int k = 0; k < 3; k++) { nr[k] = sg * ra[k]; nrm[k] = ref2 ? -nr[k] : nr[k]; }
  int b = 0;
  float bm = -1.f;
#pragma unroll
  for (int k = 0; k < 3; k++) { const float v = fabsf(dot3(Ro[k], nr)); if (v > bm) { bm = v; b = k; } }
  const int k1 = b == 2 ? 0 : b + 1, k2 = k1 == 2 ? 0 : k1 + 1;
  float ob[3], o1[3], o2[3], mo[3], fc[3];
  bb_pick(ob, Ro, b);
  bb_pick(o1, Ro, k1);
  bb_pick(o2, Ro, k2);
  const float sb = dot3(ob, nr) < 0.f ? 1.f : -1.f, sob = bb_pick1(so, b), e1 = bb_pick1(so, k1), e2 = bb_pick1(so, k2);
  for (int k = 0; k < 3; k++) { mo[k] = sb * ob[k]; fc[k] = po[k] + sb * sob * ob[k] - pr[k]; }
  const float h1 = bb_pick1(sr, a1), h2 = bb_pick1(sr, a2), hr = bb_pick1(sr, a), mn = dot3(mo, nr);
  // the four incident vertices in cyclic order (-,-), (+,-), (+,+), (-,+) and their coordinates in the reference face
  float v[4][3], vu[4], vw[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const float x1 = (q == 1 || q == 2) ? e1 : -e1, x2 = (q >= 2) ? e2 : -e2;
    for (int k = 0; k < 3; k++) v[q][k] = fc[k] + x1 * o1[k] + x2 * o2[k];
    vu[q] = dot3(v[q], t1); vw[q] = dot3(v[q], t2);
  }
  float x[3] = {0.f, 0.f, 0.f};
  bool ok = false;
  const int q = ln < 8 ? (ln & 3) : ((ln - 8) >> 2) & 3, f = (q + 1) & 3;
  // this lane's vertex q and its successor f (register selects)
  float vq[3], vf[3];
  const float xq1 = (q == 1 || q == 2) ? e1 : -e1, xq2 = (q >= 2) ? e2 : -e2, xf1 = (f == 1 || f == 2) ? e1 : -e1, xf2 = (f >= 2) ? e2 : -e2;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    vq[k] = fc[k] + xq1 * o1[k] + xq2 * o2[k];
    vf[k] = fc[k] + xf1 * o1[k] + xf2 * o2[k];
  }
  const float uq = dot3(vq, t1), wq = dot3(vq, t2), uf = dot3(vf, t1), wf = dot3(vf, t2);
  if (ln < 4) {
    ok = fabsf(uq) <= h1 && fabsf(wq) <= h2;
    for (int k = 0; k < 3; k++) x[k] = vq[k];
  } else if (ln < 8) {
    const float cu = (q == 1 || q == 2) ? h1 : -h1, cw = (q >= 2) ? h2 : -h2;
    int pos_ = 0, neg_ = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int g = (e + 1) & 3;
      const float cr = (vu[g] - vu[e]) * (cw - vw[e]) - (vw[g] - vw[e]) * (cu - vu[e]);
      if (cr > 0.f) pos_++; else if (cr < 0.f) neg_++; else { pos_ = 1; neg_ = 1; }
    }
    ok = !(pos_ && neg_);
    float num = 0.f;
    for (int k = 0; k < 3; k++) { x[k] = cu * t1[k] + cw * t2[k]; num += mo[k] * (fc[k] - x[k]); }
    const float z = num / mn;
    for (int k = 0; k < 3; k++) x[k] += z * nr[k];
  } else if (ln < 24) {
    const int e = (ln - 8) & 3;
    const float c0 = e < 2 ? uq : wq, c1 = e < 2 ? uf : wf, lim = ((e & 1) ? -1.f : 1.f) * (e < 2 ? h1 : h2);
    const float o0 = e < 2 ? wq : uq, o1_ = e < 2 ? wf : uf, ho = e < 2 ? h2 : h1;
    if ((c0 - lim) * (c1 - lim) < 0.f) {
      const float t = (lim - c0) / (c1 - c0), oo = o0 + t * (o1_ - o0);
      ok = fabsf(oo) < ho;
      for (int k = 0; k < 3; k++) x[k] = vq[k] + t * (vf[k] - vq[k]);
    }
  }
  const float depth = hr - dot3(x, nr);
  ok = ok && -depth <= margin;
  for (int k = 0; k < 3; k++) cpos[k] = pr[k] + x[k] + 0.5f * depth * nr[k];
  cdist = -depth;
  return ok;
}

}  // namespace cosim
