// Support maps of convex hulls: for each cell of a cube map of directions, the hull vertices that can be the support point for SOME
// direction in the cell.  A support query (mjc_support of a mesh geom, engine_collision_convex.c: arg max of dir . vertex over the
// hull's vertices) then scans the handful of candidates of its direction's cell instead of the whole hull (a 696-vertex wheel: 4
// candidates on average) and returns the same vertex: the candidate list is a superset of every vertex that is the arg max anywhere in
// the (slightly grown) cell, in ascending vertex order, so the first maximum of the list is the first maximum of the full scan.
//
// Superset test (exact up to the slack): v is the arg max for direction d iff d . (v - u) >= 0 for every hull neighbour u of v
// (a local maximum of a linear function on a convex polytope's vertex graph is a global one).  Over a cell -- directions p / |p| with
// p on a square of a cube face -- the sign of d . w is the sign of p . w, linear in p, so its maximum over the cell sits at one of the
// four corners.  v is kept when, for every neighbour separately, some corner has p . (v - u) >= -slack: a necessary condition for each
// neighbour, hence a superset of the joint one.  The slack (1e-4 of the hull's diameter) is three orders of magnitude above the fp32
// rounding of the dot products the kernels compare, and the cells are grown by 2e-3 in cube-map coordinates, far above the rounding
// of the cell index arithmetic: a direction that rounds into a neighbouring cell still finds its vertex there.
#pragma once
#include <cmath>
#include <vector>

namespace cosim {
constexpr int HM_R = 16;                       // cells per cube-face edge
constexpr int HM_CELLS = 6 * HM_R * HM_R;      // per hull
constexpr int HM_MIN_VERTS = 32;               // smaller hulls (the 8-vertex boxes) are scanned directly
constexpr int HM_INLINE = 4;                   // candidates stored in the cell's own record; the rest in the overflow list
constexpr int HM_REC = 1 + HM_INLINE;          // float4 words per cell record: header (count, overflow start), then HM_INLINE candidates

// cell of a direction given in the hull's own frame (any length; non-finite input lands in some valid cell)
__host__ __device__ inline int support_cell(const float* l) {
  const float ax = fabsf(l[0]), ay = fabsf(l[1]), az = fabsf(l[2]);
  const int a = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
  const float la = a == 0 ? l[0] : (a == 1 ? l[1] : l[2]);
  const float lb = a == 0 ? l[1] : (a == 1 ? l[2] : l[0]);
  const float lc = a == 0 ? l[2] : (a == 1 ? l[0] : l[1]);
  const float inv = 1.f / fabsf(la);
  const int iu = (int)fminf(fmaxf((lb * inv + 1.f) * (0.5f * HM_R), 0.f), (float)(HM_R - 1));
  const int iv = (int)fminf(fmaxf((lc * inv + 1.f) * (0.5f * HM_R), 0.f), (float)(HM_R - 1));
  return ((2 * a + (la < 0.f ? 1 : 0)) * HM_R + iu) * HM_R + iv;
}

// One hull: `verts` [n][3], CSR neighbour graph (`adr` [n + 1] absolute offsets into `nbr`, neighbour ids local to the hull).
// Appends HM_CELLS records of HM_REC float4 words to `cells`: the header (candidate count, index of the first overflow candidate in
// `cand` / 4, as int bits) and the first HM_INLINE candidates (x, y, z, vertex index as int bits; unused slots repeat the last one);
// candidates beyond HM_INLINE go to `cand`.  A query's first trip to memory -- header and inline candidates, one address computed
// from the cell index alone -- answers most cells (4 candidates on average); only the cells facing a flat side need the second.
inline void build_support_map(const float* verts, int n, const int* adr, const int* nbr, std::vector<float>& cells, std::vector<float>& cand) {
  constexpr double grow = 2e-3, slack = 1e-4;
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) { lo[k] = fmin(lo[k], (double)verts[3 * i + k]); hi[k] = fmax(hi[k], (double)verts[3 * i + k]); }
  const double diam = sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
  for (int face = 0; face < 6; face++) {
    const int a = face >> 1, b = (a + 1) % 3, c = (a + 2) % 3;
    const double s = (face & 1) ? -1.0 : 1.0;
    for (int iu = 0; iu < HM_R; iu++) for (int iv = 0; iv < HM_R; iv++) {
      double P[4][3], pmax = 0.0;
      for (int k = 0; k < 4; k++) {
        P[k][a] = s;
        P[k][b] = -1.0 + 2.0 * (iu + (k >> 1)) / HM_R + ((k >> 1) ? grow : -grow);
        P[k][c] = -1.0 + 2.0 * (iv + (k & 1)) / HM_R + ((k & 1) ? grow : -grow);
        pmax = fmax(pmax, sqrt(P[k][0] * P[k][0] + P[k][1] * P[k][1] + P[k][2] * P[k][2]));
      }
      std::vector<int> list;
      for (int v = 0; v < n; v++) {
        bool keep = true;
        for (int e = adr[v]; e < adr[v + 1] && keep; e++) {
          const int u = nbr[e];
          const double w[3] = {(double)verts[3 * v] - verts[3 * u], (double)verts[3 * v + 1] - verts[3 * u + 1], (double)verts[3 * v + 2] - verts[3 * u + 2]};
          double m = -1e300;
          for (int k = 0; k < 4; k++) m = fmax(m, P[k][0] * w[0] + P[k][1] * w[1] + P[k][2] * w[2]);
          keep = m >= -slack * diam * pmax;
        }
        if (keep) list.push_back(v);
      }
      if (list.empty()) {   // cannot happen for a closed hull (some vertex is the arg max at the cell's centre); keep the table total anyway
        double d[3]; d[a] = s; d[b] = -1.0 + (2.0 * iu + 1.0) / HM_R; d[c] = -1.0 + (2.0 * iv + 1.0) / HM_R;
        int bi = 0; double best = -1e300;
        for (int v = 0; v < n; v++) { const double t = d[0] * verts[3 * v] + d[1] * verts[3 * v + 1] + d[2] * verts[3 * v + 2]; if (t > best) { best = t; bi = v; } }
        list.push_back(bi);
      }
      union { int i; float f; } w0, w1;
      w0.i = (int)list.size(); w1.i = (int)(cand.size() / 4);
      cells.push_back(w0.f); cells.push_back(w1.f); cells.push_back(0.f); cells.push_back(0.f);
      for (int k = 0; k < (int)list.size() || k < HM_INLINE; k++) {
        const int v = list[k < (int)list.size() ? k : (int)list.size() - 1];
        union { int i; float f; } ix;
        ix.i = v;
        std::vector<float>& dst = k < HM_INLINE ? cells : cand;
        dst.push_back(verts[3 * v]); dst.push_back(verts[3 * v + 1]); dst.push_back(verts[3 * v + 2]); dst.push_back(ix.f);
      }
    }
  }
}
}  // namespace cosim
