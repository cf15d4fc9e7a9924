// Hardware check of the lane/register layout assumed for v_mfma_f32_32x32x1_2b_f32 and of the 32-lane group reductions
// used by the two-environments-per-wave kernel.  Build: hipcc --offload-arch=gfx950 -O2 -o mfma_2b_layout mfma_2b_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x32 __attribute__((ext_vector_type(32)));
__global__ void k(float* out, float* red) {
  const int l = threadIdx.x;
  const float a = (float)(64 * (l >> 5) + (l & 31) + 1);   // A_b[i] = 64 b + i + 1
  const float b = (float)((l & 31) + 1) + 0.5f * (l >> 5); // B_b[j] = j + 1 + b / 2
  f32x32 acc;
  for (int v = 0; v < 32; v++) acc[v] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, acc, 0, 0, 0);
  for (int v = 0; v < 32; v++) out[l * 32 + v] = acc[v];
  // group sum over 32 lanes: DPP row sum + ds_swizzle xor 16
  float v = (float)(l + 1);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));
  red[l] = v;
  red[64 + l] = __shfl((float)(l * 10), 5 + (l & 32), 64);
  red[128 + l] = (float)((__ballot((l % 3) == 0) >> (l & 32)) & 0xffffffffull);
}
int main() {
  float *d, *r;
  hipMalloc(&d, 64 * 32 * 4); hipMalloc(&r, 192 * 4);
  k<<<1, 64>>>(d, r);
  float h[64 * 32], hr[192];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(hr, r, sizeof hr, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++)
    for (int v = 0; v < 32; v++) {
      const int blk = v >> 4, row = (v & 3) + 8 * ((v & 15) >> 2) + 4 * (l >> 5), col = l & 31;
      const float exp = (float)(64 * blk + row + 1) * ((float)(col + 1) + 0.5f * blk);
      if (fabsf(h[l * 32 + v] - exp) > 1e-3f) { if (bad < 8) printf("lane %d reg %d: got %g expected %g\n", l, v, h[l * 32 + v], exp); bad++; }
    }
  printf("mfma_32x32x1_2b layout mismatches: %d\n", bad);
  int bad2 = 0;
  for (int l = 0; l < 64; l++) {
    const float exps = l < 32 ? 528.f : 1552.f;   // sum 1..32, 33..64
    if (hr[l] != exps) { if (bad2 < 4) printf("grp_sum lane %d: %g vs %g\n", l, hr[l], exps); bad2++; }
    if (hr[64 + l] != (float)((5 + (l & 32)) * 10)) { if (bad2 < 8) printf("shfl lane %d: %g\n", l, hr[64 + l]); bad2++; }
  }
  printf("group op mismatches: %d; ballot halves: %g %g\n", bad2, hr[128], hr[128 + 32]);
  return bad || bad2;
}
