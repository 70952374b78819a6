// Checks the operand / result lane layout of v_mfma_f32_16x16x1_4b_f32 (4 independent 16x16 outer
// products per instruction) that k_gp_mfma16 relies on:
//   A: lane 16*blk + i holds A_blk[i];  B: lane 16*blk + j holds B_blk[j];
//   D: register 4*blk + r of lane 16*rg + j holds D_blk[4*rg + r][j].
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float16v __attribute__((ext_vector_type(16)));

__global__ void k(float* out) {
  const int lane = threadIdx.x;
  const int blk = lane >> 4, i = lane & 15;
  const float a = float(100 * (blk + 1) + i);        // A_blk[i]
  const float b = float(1000 * (blk + 1) + 7 * i);   // B_blk[j]
  float16v c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
}

int main() {
  float* d; if (hipMalloc(&d, 64 * 16 * sizeof(float)) != hipSuccess) return 1;
  k<<<1, 64>>>(d);
  float h[64 * 16];
  if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int reg = 0; reg < 16; ++reg) {
      const int blk = reg >> 2, r = reg & 3, rg = lane >> 4, j = lane & 15;
      const int i = 4 * rg + r;
      const float want = float(100 * (blk + 1) + i) * float(1000 * (blk + 1) + 7 * j);
      if (h[lane * 16 + reg] != want) { if (bad < 5) printf("lane %d reg %d: got %g want %g\n", lane, reg, h[lane * 16 + reg], want); ++bad; }
    }
  printf(bad ? "LAYOUT MISMATCH (%d)\n" : "layout OK\n", bad);
  return bad != 0;
}
