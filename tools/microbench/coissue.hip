// Microbenchmark: do f32 MFMA waves and v_pk_fma_f32 waves on the same SIMD add up?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float2v __attribute__((ext_vector_type(2)));

// mode 0: all waves MFMA; 1: all waves pk_fma; 2: blocks alternate (even block MFMA, odd block pk_fma)
__global__ __launch_bounds__(256) void k(float* out, int iters_m, int iters_v, int mode, float seed) {
  const int lane = threadIdx.x & 63;
  const bool do_mfma = mode == 0 || (mode == 2 && ((blockIdx.x >> 3) & 1) == 0);
  float sum = 0;
  if (do_mfma) {
    float16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = seed + lane * 0.001f, b = 0.5f - lane * 0.002f;
    for (int it = 0; it < iters_m; ++it) {
#pragma unroll
      for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) sum += acc[r];
  } else {
    float2v acc[16];
    float2v av = {seed, seed + 1.f}, bv = {0.5f, -0.5f};
    for (int i = 0; i < 16; ++i) acc[i] = float2v{(float)lane + i, (float)i};
    for (int it = 0; it < iters_v; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(bv));
    }
    for (int i = 0; i < 16; ++i) sum += acc[i].x + acc[i].y;
  }
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
  for (int r = 0; r < 10; ++r) k<<<256 * 4, 256>>>(out, 4000, 16000, 0, 1.0f);
  CK(hipDeviceSynchronize());
  const int blocks = 256 * 8;            // 8 blocks of 4 waves per CU: 8 waves per SIMD
  const int im = 4000;                   // MFMA iterations: 16 x 64 cycles each
  for (int iv : {26000, 52000, 60000}) {  // pk_fma iterations: 16 x ~4.9 cycles each
    double fm = 2.0 * 2048 * 16.0 * im * blocks * 4, fv = 4.0 * 64 * 16.0 * iv * blocks * 4;
    double t0 = timeit([&]{ k<<<blocks, 256>>>(out, im, iv, 0, 1.0f); }, 3);
    double t1 = timeit([&]{ k<<<blocks, 256>>>(out, im, iv, 1, 1.0f); }, 3);
    double t2 = timeit([&]{ k<<<blocks, 256>>>(out, im, iv, 2, 1.0f); }, 3);
    printf("iv %d: all-MFMA %.1f TF (%.2f ms) | all-pk_fma %.1f TF (%.2f ms) | half/half: %.1f TF total (%.2f ms; serial sum of halves would be %.2f ms)\n",
           iv, fm / t0 * 1e-12, t0 * 1e3, fv / t1 * 1e-12, t1 * 1e3, (fm + fv) / 2 / t2 * 1e-12, t2 * 1e3, (t0 + t1) / 2 * 1e3);
  }
  return 0;
}
