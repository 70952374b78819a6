// Microbenchmark: which FP32/FP64 vector FMA form reaches the MI355X peak, and at what occupancy.
// Build: hipcc --offload-arch=gfx950 -O3 fma_rate.hip -o fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

typedef float float2v __attribute__((ext_vector_type(2)));

template<int NACC>
__global__ void k_fma32(float* out, int iters, float a, float b) {
  float acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  float s = 0; 
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// SGPR operand form
template<int NACC>
__global__ void k_fma32_s(float* out, int iters, float a, float b) {
  float acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x + i;
  float bb = b + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "s"(a), "v"(bb));
  }
  float s = 0; 
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template<int NACC>
__global__ void k_pkfma32(float* out, int iters, float a, float b) {
  float2v acc[NACC];
  float2v av = {a, a + 1.f}, bv = {b, b - 1.f};
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = float2v{(float)threadIdx.x + i, (float)i};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(bv));
  }
  float s = 0; 
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template<int NACC>
__global__ void k_fma64(float* out, int iters, double a, double b) {
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  double s = 0; 
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}
template<int NACC>
__global__ void k_mul64(float* out, int iters, double a, double b) {
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
  }
  double s = 0; 
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}

template<typename F>
double timeit(F f, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e-3;
}


__global__ void k_clock(unsigned long long* out, int iters) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float acc = threadIdx.x;
  for (int it = 0; it < iters; ++it) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(acc));
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)acc; }
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  float* out; CK(hipMalloc(&out, 256 * 64 * 1024 * sizeof(float)));
  unsigned long long* clk; CK(hipMalloc(&clk, 64));
  const int iters = 40000;   // 16*40000 = 640k instr per wave
  constexpr int NACC = 16;
  // warmup ~1.5 s
  for (int r = 0; r < 30; ++r) k_pkfma32<NACC><<<256*8, 256>>>(out, iters, 1.0001f, 0.5f);
  CK(hipDeviceSynchronize());
  for (int round = 0; round < 2; ++round)
  for (int wpc : {4, 8, 12, 16, 24, 32}) {  // waves per CU
    int threads = 256; int blocks = 256 * wpc * 64 / threads;
    double t;
    t = timeit([&]{ k_fma32<NACC><<<blocks, threads>>>(out, iters, 1.0001f, 0.5f); }, 3);
    printf("r%d wpc %2d v_fma_f32      : %7.1f TFLOP/s  (%.1f ms)\n", round, wpc, 2.0 * NACC * iters * (double)blocks * threads / t * 1e-12, t*1e3);
    t = timeit([&]{ k_fma32_s<NACC><<<blocks, threads>>>(out, iters, 1.0001f, 0.5f); }, 3);
    printf("r%d wpc %2d v_fma_f32 sgpr : %7.1f TFLOP/s\n", round, wpc, 2.0 * NACC * iters * (double)blocks * threads / t * 1e-12);
    t = timeit([&]{ k_pkfma32<NACC><<<blocks, threads>>>(out, iters, 1.0001f, 0.5f); }, 3);
    printf("r%d wpc %2d v_pk_fma_f32   : %7.1f TFLOP/s\n", round, wpc, 4.0 * NACC * iters * (double)blocks * threads / t * 1e-12);
    t = timeit([&]{ k_fma64<NACC><<<blocks, threads>>>(out, iters, 1.0001, 0.5); }, 3);
    printf("r%d wpc %2d v_fma_f64      : %7.1f TFLOP/s\n", round, wpc, 2.0 * NACC * iters * (double)blocks * threads / t * 1e-12);
    t = timeit([&]{ k_mul64<NACC><<<blocks, threads>>>(out, iters, 1.0001, 0.5); }, 3);
    printf("r%d wpc %2d v_mul_f64      : %7.1f Tinstr-lanes/s\n", round, wpc, 1.0 * NACC * iters * (double)blocks * threads / t * 1e-12);
    fflush(stdout);
  }
  // clock under load: run the pk kernel on a second stream while stamping
  hipStream_t s2; CK(hipStreamCreate(&s2));
  for (int r = 0; r < 4; ++r) k_pkfma32<NACC><<<255*8, 256, 0, s2>>>(out, iters, 1.0001f, 0.5f);
  k_clock<<<1, 64>>>(clk, 2000000);
  CK(hipDeviceSynchronize());
  unsigned long long h[3]; CK(hipMemcpy(h, clk, 24, hipMemcpyDeviceToHost));
  printf("clock under load: %.0f MHz (memtime %llu / realtime %llu x100MHz)\n", (double)h[0] / (double)h[1] * 100.0, h[0], h[1]);
  k_clock<<<1, 64>>>(clk, 2000000);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h, clk, 24, hipMemcpyDeviceToHost));
  printf("clock idle-ish: %.0f MHz\n", (double)h[0] / (double)h[1] * 100.0);
  return 0;
}
