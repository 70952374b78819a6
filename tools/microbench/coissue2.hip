// Microbenchmark: which instruction classes of OTHER waves make progress while a wave on the same
// SIMD streams f32 MFMAs?  Half the blocks run MFMAs, the other half one of: v_pk_fma_f32,
// v_pk_add_f32, v_add_f32, v_xor_b32 (integer), ds_read_b32.  Overlap = serial sum / measured.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float2v __attribute__((ext_vector_type(2)));

// who: 0 = every block MFMA, 1 = every block "other", 2 = alternate
template <int kind>
__global__ __launch_bounds__(512) void k(float* out, int iters_m, int iters_v, int who, float seed) {
  __shared__ float lds[4096];
  const int lane = threadIdx.x & 63;
  const bool do_mfma = who == 0 || (who == 2 && (threadIdx.x >> 6) < 4);  // waves w and w+4 share a SIMD
  float sum = 0;
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = seed + i;
  __syncthreads();
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
    uint32_t addr = threadIdx.x * 4;
    for (int it = 0; it < iters_v; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (kind == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(bv));
        else if (kind == 1) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(av));
        else if (kind == 2) asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[i].x) : "v"(av.x));
        else if (kind == 3) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(acc[i].x) : "v"(av.x));
        else asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(acc[i].x) : "v"(addr), "n"(0));
      }
      if (kind == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    for (int i = 0; i < 16; ++i) sum += acc[i].x + acc[i].y;
  }
  out[blockIdx.x * 512 + threadIdx.x] = sum;
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

template <int kind> void run(float* out, const char* name) {
  const int blocks = 256, im = 4000;
  int iv = 40000;
  double tm = timeit([&]{ k<kind><<<blocks, 512>>>(out, im, iv, 0, 1.0f); }, 2);
  double tv = timeit([&]{ k<kind><<<blocks, 512>>>(out, im, iv, 1, 1.0f); }, 2);
  iv = int(iv * tm / tv);
  tv = timeit([&]{ k<kind><<<blocks, 512>>>(out, im, iv, 1, 1.0f); }, 3);
  double t2 = timeit([&]{ k<kind><<<blocks, 512>>>(out, im, iv, 2, 1.0f); }, 3);
  // all-X: 2 waves of X per SIMD; mixed: 1 MFMA wave + 1 other wave per SIMD
  printf("%-13s: 2 MFMA waves/SIMD %.2f ms, 2 other waves/SIMD %.2f ms (%.2f cycles/instr/SIMD at 2.4 GHz), 1+1 mixed %.2f ms; no-overlap %.2f ms, perfect overlap %.2f ms\n",
         name, tm * 1e3, tv * 1e3, tv * 2.4e9 / (16.0 * iv) / 2, t2 * 1e3, (tm + tv) / 2 * 1e3, (tm > tv ? tm : tv) / 2 * 1e3);
}

int main() {
  float* out; if (hipMalloc(&out, 256 * 512 * sizeof(float)) != hipSuccess) return 1;
  for (int r = 0; r < 10; ++r) k<0><<<256, 512>>>(out, 4000, 16000, 0, 1.0f);
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  run<0>(out, "v_pk_fma_f32");
  run<1>(out, "v_pk_add_f32");
  run<2>(out, "v_add_f32");
  run<3>(out, "v_xor_b32");
  run<4>(out, "ds_read_b32");
  return 0;
}
