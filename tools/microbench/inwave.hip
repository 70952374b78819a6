// Microbenchmark: how many vector instructions of the SAME wave hide under its MFMAs?
// One wave per SIMD; loop body = 3 independent v_mfma_f32_32x32x2_f32, each followed by N vector ops.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float16v __attribute__((ext_vector_type(16)));

template <int N, int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  __shared__ float lds[2048];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = seed + i;
  __syncthreads();
  float16v a0, a1, a2;
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; }
  float x = seed + lane * 0.001f, y = 0.5f - lane * 0.002f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = lane + i;
  uint32_t addr = threadIdx.x * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      if (m == 0) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      if (m == 1) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      if (m == 2) a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        if (KIND == 0) asm volatile("v_add_f32 %0, %1, %0" : "+v"(v[j & 7]) : "v"(x));
        if (KIND == 1) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(*(double*)&v[(j & 3) * 2]) : "v"(*(double*)&v[0]));
        if (KIND == 2) asm volatile("ds_read_b32 %0, %1" : "=v"(v[j & 7]) : "v"(addr));
      }
      if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  float sum = 0;
  for (int r = 0; r < 16; ++r) sum += a0[r] + a1[r] + a2[r];
  for (int i = 0; i < 8; ++i) sum += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <typename F> double timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) f();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 3 * 1e-3;
}

template <int N, int KIND> void run(float* out, const char* name) {
  const int iters = 20000;
  double t = timeit([&]{ k<N, KIND><<<256, 256>>>(out, iters, 1.0f); });
  printf("%-12s N = %2d per MFMA: %.1f cycles per MFMA (at 2.4 GHz)\n", name, N, t * 2.4e9 / (3.0 * iters));
}

int main() {
  float* out; if (hipMalloc(&out, 256 * 256 * sizeof(float)) != hipSuccess) return 1;
  run<0, 0>(out, "v_add_f32"); run<2, 0>(out, "v_add_f32"); run<4, 0>(out, "v_add_f32"); run<8, 0>(out, "v_add_f32");
  run<12, 0>(out, "v_add_f32"); run<16, 0>(out, "v_add_f32"); run<24, 0>(out, "v_add_f32");
  run<4, 1>(out, "v_pk_add_f32"); run<8, 1>(out, "v_pk_add_f32"); run<16, 1>(out, "v_pk_add_f32");
  run<2, 2>(out, "ds_read_b32"); run<4, 2>(out, "ds_read_b32"); run<8, 2>(out, "ds_read_b32");
  return 0;
}
