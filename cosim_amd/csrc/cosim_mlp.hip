// cosim_mlp.hip — fused forward pass of a small actor MLP on the matrix pipe.
//
// The reference evaluates its ONNX policy once per control step for one state (core/policy.py:11-21); for N environments
// the same network is [N, in] -> hidden ... -> [N, action_dim].  Those layers are tiny (52 -> 256 -> 128 -> 4 for the
// usual actor), so a chain of library GEMM + bias + activation launches is launch-bound: here one workgroup takes 32
// environments through all layers, activations stay in LDS, weights stream from L2 and every product runs as
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains, like the CPU evaluation).
//
// Layout of one 32x32 output tile (lane l, register v): A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31];
// C/D: col = l & 31, row = (v & 3) + 8 (v >> 2) + 4 (l >> 5)  (the layout the step kernel's Hessian build uses).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cosim {

constexpr int MLP_MAXL = 6;      // layers
constexpr int MLP_MAXD = 512;    // widest layer: two ping-pong tiles of 32 x (width + 1) floats must fit the 160 KB LDS

struct MlpArgs {
  const float* x;      // [n, dims[0]]
  float* out;          // [n, dims[nl]]
  const float* w[MLP_MAXL];   // [dims[l+1], dims[l]] row-major (ONNX Gemm with transB = 1)
  const float* b[MLP_MAXL];   // [dims[l+1]] or null
  int dims[MLP_MAXL + 1];
  int act[MLP_MAXL];   // 0 none, 1 relu, 2 tanh, 3 elu, 4 sigmoid, 5 leaky relu
  float act_alpha[MLP_MAXL];
  int nl, n, ld;       // ld: leading dimension of the LDS tiles (widest layer + 1: odd, conflict-free column reads)
  float clip;          // > 0: clamp the output to [-clip, clip]
};

__device__ __forceinline__ float mlp_act(float x, int kind, float alpha) {
  switch (kind) {
    case 1: return fmaxf(x, 0.f);
    case 2: return tanhf(x);
    case 3: return x > 0.f ? x : alpha * (expf(x) - 1.f);
    case 4: return 1.f / (1.f + expf(-x));
    case 5: return x > 0.f ? x : alpha * x;
    default: return x;
  }
}

__global__ __launch_bounds__(256) void mlp_forward_kernel(MlpArgs A) {
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  extern __shared__ float mlp_lds[];   // [2][32][ld]
  const int ld = A.ld;
  auto tile = [&](int which, int r, int c) -> float& { return mlp_lds[(which * 32 + r) * ld + c]; };
  const int t = threadIdx.x, l = t & 63, wave = t >> 6;
  const int n0 = blockIdx.x * 32;
  // the input tile, coalesced
  {
    const int I = A.dims[0];
    for (int e = t; e < 32 * I; e += 256) {
      const int r = e / I, c = e - r * I;
      tile(0, r, c) = (n0 + r < A.n) ? A.x[(size_t)(n0 + r) * I + c] : 0.f;
    }
  }
  __syncthreads();
  int cur = 0;
  for (int L = 0; L < A.nl; L++) {
    const int I = A.dims[L], O = A.dims[L + 1];
    const float* W = A.w[L];
    const float* Bv = A.b[L];
    const bool last = L == A.nl - 1;
    const int row_a = l & 31, kk = l >> 5;
    for (int ti = wave; ti * 32 < O; ti += 4) {
      const int o0 = ti * 32, col = o0 + (l & 31);
      const float bias = (Bv != nullptr && col < O) ? Bv[col] : 0.f;
      f32x16 acc;
#pragma unroll
      for (int v = 0; v < 16; v++) acc[v] = bias;
      const float* wrow = W + (size_t)(col < O ? col : 0) * I;
      if ((I & 3) == 0) {   // four weights of the lane's row per load: k0 + kk and k0 + 2 + kk feed two MFMA steps
        for (int k0 = 0; k0 < I; k0 += 4) {
          const float4 w4 = col < O ? *reinterpret_cast<const float4*>(wrow + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
          const float a0 = tile(cur, row_a, k0 + kk), a1 = tile(cur, row_a, k0 + 2 + kk);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, kk ? w4.y : w4.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, kk ? w4.w : w4.z, acc, 0, 0, 0);
        }
      } else {
        for (int k0 = 0; k0 < I; k0 += 2) {
          const int k = k0 + kk;
          const float a = k < I ? tile(cur, row_a, k) : 0.f;
          const float b = (k < I && col < O) ? wrow[k] : 0.f;
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int v = 0; v < 16; v++) {
        const int row = (v & 3) + 8 * (v >> 2) + 4 * (l >> 5);
        float y = mlp_act(acc[v], A.act[L], A.act_alpha[L]);
        if (last) {
          if (A.clip > 0.f) y = fminf(A.clip, fmaxf(-A.clip, y));
          if (col < O && n0 + row < A.n) A.out[(size_t)(n0 + row) * O + col] = y;
        } else if (col < O)
          tile(cur ^ 1, row, col) = y;
      }
    }
    __syncthreads();
    cur ^= 1;
  }
}

// One LSTM cell step for N environments (the recurrent policy of core/policy.py:24-47: the ONNX LSTM node with sequence length 1,
// one direction, default activations; gate order i, o, f, c as ONNX lays W / R / B out).  A workgroup takes 32 environments: [x | h]
// staged in LDS once, each wave owns 32 hidden units at a time and accumulates their four gates (x W^T + h R^T + Wb + Rb) as four
// 32 x 32 tiles on the matrix pipe, then applies c' = sigma(f) c + sigma(i) tanh(g), h' = sigma(o) tanh(c') in registers.
struct LstmArgs {
  const float *x, *h, *c;      // [n, I], [n, H], [n, H]
  const float *W, *R, *B;      // [4H, I], [4H, H], [8H] (Wb then Rb) or null
  float *h_out, *c_out;        // [n, H]; may alias h / c (each element is read before the workgroup that owns its rows writes it)
  int n, I, H, ld;
};

__global__ __launch_bounds__(256) void lstm_cell_kernel(LstmArgs A) {
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  extern __shared__ float lstm_lds[];   // [32][ld]: x then h of 32 environments
  const int t = threadIdx.x, l = t & 63, wave = t >> 6, n0 = blockIdx.x * 32, I = A.I, H = A.H, K = I + H, ld = A.ld;
  for (int e = t; e < 32 * K; e += 256) {
    const int r = e / K, k = e - r * K;
    float v = 0.f;
    if (n0 + r < A.n) v = k < I ? A.x[(size_t)(n0 + r) * I + k] : A.h[(size_t)(n0 + r) * H + (k - I)];
    lstm_lds[r * ld + k] = v;
  }
  __syncthreads();
  const int row_a = l & 31, kk = l >> 5;
  // every c this workgroup needs is read before any of its h / c is written (in-place update across launches is safe; h_out may
  // alias h because h was staged in LDS above and other workgroups own other rows)
  for (int ti = wave; ti * 32 < H; ti += 4) {
    const int col = ti * 32 + (l & 31);
    const bool cok = col < H;
    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const float bias = (A.B != nullptr && cok) ? A.B[q * H + col] + A.B[4 * H + q * H + col] : 0.f;
#pragma unroll
      for (int v = 0; v < 16; v++) acc[q][v] = bias;
    }
    for (int k0 = 0; k0 < K; k0 += 2) {
      const int k = k0 + kk;
      const float a = k < K ? lstm_lds[row_a * ld + k] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        float w = 0.f;
        if (cok && k < K) w = k < I ? A.W[(size_t)(q * H + col) * I + k] : A.R[(size_t)(q * H + col) * H + (k - I)];
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w, acc[q], 0, 0, 0);
      }
    }
#pragma unroll
    for (int v = 0; v < 16; v++) {
      const int row = (v & 3) + 8 * (v >> 2) + 4 * (l >> 5);
      if (cok && n0 + row < A.n) {
        const size_t o = (size_t)(n0 + row) * H + col;
        const float ig = 1.f / (1.f + expf(-acc[0][v])), og = 1.f / (1.f + expf(-acc[1][v])), fg = 1.f / (1.f + expf(-acc[2][v]));
        const float cn = fg * A.c[o] + ig * tanhf(acc[3][v]);
        A.c_out[o] = cn;
        A.h_out[o] = og * tanhf(cn);
      }
    }
  }
}

// Reporter side of the loop (core/reporter.py:210-218, 429-442, 506-508): count / sum / sum of squares per metric column over
// the fleet in one launch.  Columns: info[0..4) as they are, |torque| (info[4 .. 4 + nu)), |command_i - measured_i| for
// i < ncmd (measured = lin_vel_x, lin_vel_y, ang_vel_yaw = info[1 + i]).  acc is [3][K] doubles, K = 4 + nu + ncmd <= 32.
__global__ __launch_bounds__(256) void fleet_stats_kernel(const float* info, int n, int info_dim, int nu, const float* cmd, int cmd_stride,
                                                          int ncmd, double* acc) {
  __shared__ float part[2][8][32];
  const int K = 4 + nu + ncmd, t = threadIdx.x, c = t & 31, g = t >> 5;
  float s = 0.f, q = 0.f;
  if (c < K)
    for (int r = blockIdx.x * 8 + g; r < n; r += gridDim.x * 8) {
      const float* row = info + (size_t)r * info_dim;
      float v;
      if (c < 4) v = row[c];
      else if (c < 4 + nu) v = fabsf(row[c]);
      else { const int i = c - 4 - nu; v = fabsf(cmd[(size_t)r * cmd_stride + i] - row[1 + i]); }
      s += v; q += v * v;
    }
  part[0][g][c] = s; part[1][g][c] = q;
  __syncthreads();
  if (t < 32 && t < K) {
    double ss = 0.0, qq = 0.0;
    for (int k = 0; k < 8; k++) { ss += (double)part[0][k][t]; qq += (double)part[1][k][t]; }
    atomicAdd(&acc[K + t], ss);
    atomicAdd(&acc[2 * K + t], qq);
    if (blockIdx.x == 0) atomicAdd(&acc[t], (double)n);
  }
}

// Percentiles of the same metric columns (N2: core/reporter.py:429-442, 506-530 plots the full series; a fleet keeps a mergeable
// sketch instead): one histogram per column, `nbins` equal bins over [0, hi[c]) (values are magnitudes; column 1..3, the signed base
// velocities, are binned by magnitude too), the last bin also takes everything above the range.  hist is [K][nbins] doubles, summed
// over steps and -- by the caller -- over ranks.
__global__ __launch_bounds__(256) void fleet_hist_kernel(const float* info, int n, int info_dim, int nu, const float* cmd, int cmd_stride,
                                                         int ncmd, const float* hi, int nbins, double* hist) {
  const int K = 4 + nu + ncmd, t = threadIdx.x, c = t & 31, g = t >> 5;
  if (c >= K) return;
  const float scale = (float)nbins / hi[c];
  for (int r = blockIdx.x * 8 + g; r < n; r += gridDim.x * 8) {
    const float* row = info + (size_t)r * info_dim;
    float v;
    if (c < 4 + nu) v = fabsf(row[c]);
    else { const int i = c - 4 - nu; v = fabsf(cmd[(size_t)r * cmd_stride + i] - row[1 + i]); }
    int b = (int)(v * scale);
    b = b < 0 ? 0 : (b >= nbins ? nbins - 1 : b);
    atomicAdd(&hist[(size_t)c * nbins + b], 1.0);
  }
}

}  // namespace cosim
