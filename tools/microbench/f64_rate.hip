// Microbenchmark: issue rate of v_add_f64 / v_mul_f64 / v_fma_f64 (independent chains, 1..8 waves per SIMD) -- what
// bounds the straight-line f64 code of the hiprtc-specialised kernels (config 5: 336 v_mul_f64 + 357 v_add_f64 per item).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
  double a[8], b = seed + threadIdx.x * 1e-3, c = 1.0 + seed * 1e-9;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = seed * (i + 1);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (MODE == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (MODE == 2) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
        if (MODE == 3) { double t; asm volatile("v_mul_f64 %0, %1, %2" : "=v"(t) : "v"(a[(i + 1) & 7]), "v"(c)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(t)); }
        if (MODE == 4) { double t; asm volatile("v_fma_f64 %0, %1, %2, -0" : "=v"(t) : "v"(a[(i + 1) & 7]), "v"(c)); asm volatile("v_fma_f64 %0, %1, 1.0, %0" : "+v"(a[i]) : "v"(t)); }
      }
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
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
  double* out; CK(hipMalloc(&out, 256 * 8 * 256 * sizeof(double)));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const double clk = prop.clockRate * 1e3;
  const int iters = 2000;
  for (int bpc : {1, 2, 4, 8}) {
    int blocks = 256 * bpc;
    const double instr_per_simd = 32.0 * iters * bpc;      // per wave 32 "slots" per iteration, bpc waves per SIMD
    double t[5];
    t[0] = timeit([&]{ k<0><<<blocks, 256>>>(out, iters, 1.0); }, 3);
    t[1] = timeit([&]{ k<1><<<blocks, 256>>>(out, iters, 1.0); }, 3);
    t[2] = timeit([&]{ k<2><<<blocks, 256>>>(out, iters, 1.0); }, 3);
    t[3] = timeit([&]{ k<3><<<blocks, 256>>>(out, iters, 1.0); }, 3);
    t[4] = timeit([&]{ k<4><<<blocks, 256>>>(out, iters, 1.0); }, 3);
    printf("waves/SIMD %d (clock %.2f GHz): cycles per wave-instruction: add %.2f | mul %.2f | fma %.2f | mul+add pair %.2f | fma(-0)+fma(1.0) pair %.2f\n", bpc, clk * 1e-9,
           t[0] * clk / instr_per_simd, t[1] * clk / instr_per_simd, t[2] * clk / instr_per_simd, t[3] * clk / instr_per_simd, t[4] * clk / instr_per_simd);
  }
  return 0;
}
