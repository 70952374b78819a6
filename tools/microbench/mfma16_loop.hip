// Microbenchmark: what a v_mfma_f32_16x16x1_4b_f32 step of k_gp_mfma16 costs with its fillers -- pure MFMA chain, + one
// v_xor_b32_dpp per MFMA (the A operand), + v_pk_mul_f32 / 2 (the B block sign), + the LDS reads of a step (1 b32 + 4 b128
// per 16 MFMAs) -- at 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = seed * (i % 97) * 0.01f - 0.3f;
  __syncthreads();
  float16v acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int lane = threadIdx.x & 63;
  uint32_t a0 = __float_as_uint(seed + lane * 0.001f);
  float2v b2[8];
  for (int i = 0; i < 8; ++i) b2[i] = float2v{0.5f - lane * 0.002f, 0.25f + i};
  uint32_t m[16];
  for (int i = 0; i < 16; ++i) m[i] = uint32_t((lane >> (i & 3)) & 1) << 31;
  for (int it = 0; it < iters; ++it) {
    float2v s2 = float2v{(it & 1) ? -1.f : 1.f, (it & 1) ? -1.f : 1.f};
    if (MODE >= 3) {
      a0 = reinterpret_cast<const uint32_t*>(lds)[((it & 63) << 6) + lane];
      const float4v* bp = reinterpret_cast<const float4v*>(lds + 4096 + (((it + lane) & 63) << 4));
#pragma unroll
      for (int q = 0; q < 4; ++q) { float4v v = bp[(q + (lane >> 2)) & 3]; b2[2 * q] = float2v{v.x, v.y}; b2[2 * q + 1] = float2v{v.z, v.w}; }
    }
    float2v bb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bb[i] = MODE >= 2 ? b2[i] * s2 : b2[i];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      uint32_t av = a0;
      if (MODE >= 1) {
        if ((kk & 3) == 0) av = a0 ^ m[kk];
        if ((kk & 3) == 1) av = uint32_t(__builtin_amdgcn_update_dpp(0, int(a0), 0xB1, 0xf, 0xf, true)) ^ m[kk];
        if ((kk & 3) == 2) av = uint32_t(__builtin_amdgcn_update_dpp(0, int(a0), 0x4E, 0xf, 0xf, true)) ^ m[kk];
        if ((kk & 3) == 3) av = uint32_t(__builtin_amdgcn_update_dpp(0, int(a0), 0x1B, 0xf, 0xf, true)) ^ m[kk];
      }
      const float bv = (kk & 1) ? bb[kk >> 1].y : bb[kk >> 1].x;
      acc = __builtin_amdgcn_mfma_f32_16x16x1f32(__uint_as_float(av), bv, acc, 0, 0, 0);
    }
  }
  float sum = 0;
  for (int r = 0; r < 16; ++r) sum += acc[r];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <typename F> double timeit(F f, int reps) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) f();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e-3;
}

int main() {
  float* out; CK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const double clk = prop.clockRate * 1e3;
  const int iters = 4000;
  for (int bpc : {1, 2, 4}) {
    int blocks = 256 * bpc;
    const double mfma_per_simd = 16.0 * iters * bpc;
    double t0 = timeit([&]{ k<0><<<blocks, 256>>>(out, iters, 1.0f); }, 3);
    double t1 = timeit([&]{ k<1><<<blocks, 256>>>(out, iters, 1.0f); }, 3);
    double t2 = timeit([&]{ k<2><<<blocks, 256>>>(out, iters, 1.0f); }, 3);
    double t3 = timeit([&]{ k<3><<<blocks, 256>>>(out, iters, 1.0f); }, 3);
    printf("waves/SIMD %d: cycles per 16x16x1_4b MFMA: pure chain %.1f | + xor_dpp %.1f | + pk_mul/2 %.1f | + LDS reads of a step %.1f   (32 = peak)\n", bpc,
           t0 * clk / mfma_per_simd, t1 * clk / mfma_per_simd, t2 * clk / mfma_per_simd, t3 * clk / mfma_per_simd);
  }
  return 0;
}
