// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 (the instruction of k_gp_mfma16x4<double>) at 1 ... 8 waves per
// SIMD, with 1, 2 and 4 independent accumulator chains per wave, and with one ds_read_b64 per operand -- the MEASURED
// FP64 matrix roof beside the 78.6 TFLOP/s datasheet figure bench.py prices the f64 dense kernels against
// (the local guide MI355X_MICROARCH.md has no FP64 MFMA number).
//   flops per instruction: 16 x 16 x 4 x 2 = 2,048
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef double double4v __attribute__((ext_vector_type(4)));

// MODE 0: register operands; MODE 1: both operands come from LDS (one ds_read_b64 each, conflict-free) per instruction
template <int CHAINS, int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
  __shared__ double lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed * (i % 97) * 0.01 - 0.3;
  __syncthreads();
  double4v acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 4; ++r) acc[c][r] = 0.0;
  const int lane = threadIdx.x & 63;
  double a = seed + lane * 0.001, b = 0.5 - lane * 0.002;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        double av = a, bv = b;
        if (MODE == 1) {
          av = lds[(((it + c) & 31) << 6) + (lane ^ s)];
          bv = lds[2048 + (((it + c) & 31) << 6) + (lane ^ (s << 1))];
        }
        acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[c], 0, 0, 0);
      }
    }
  }
  double sum = 0;
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 4; ++r) sum += acc[c][r];
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
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  double* out; CK(hipMalloc(&out, size_t(cus) * 8 * 256 * sizeof(double)));
  const int iters = 2000;
  for (int r = 0; r < 20; ++r) k<1, 0><<<cus * 4, 256>>>(out, iters, 1.0);   // warm the clocks
  CK(hipDeviceSynchronize());
  printf("v_mfma_f64_16x16x4_f64 on %s, %d CUs, clock %d MHz; TFLOP/s (cycles per instruction and SIMD at that clock)\n", prop.gcnArchName, cus, prop.clockRate / 1000);
  for (int bpc : {1, 2, 3, 4, 6, 8}) {   // 256-thread workgroups per CU = waves per SIMD
    const int blocks = cus * bpc;
    auto report = [&](const char* what, int chains, double t) {
      const double insts = double(chains) * 16.0 * iters * blocks * 4;   // 4 waves per workgroup
      const double tf = insts * 2048.0 / t * 1e-12;
      const double cyc = t * (prop.clockRate * 1e3) / (insts / (double(cus) * 4));
      printf("  %-28s %6.1f TF (%5.1f cyc)", what, tf, cyc);
    };
    printf("waves/SIMD %d:", bpc);
    report("1 chain", 1, timeit([&]{ k<1, 0><<<blocks, 256>>>(out, iters, 1.0); }, 3));
    report("2 chains", 2, timeit([&]{ k<2, 0><<<blocks, 256>>>(out, iters, 1.0); }, 3));
    report("4 chains", 4, timeit([&]{ k<4, 0><<<blocks, 256>>>(out, iters, 1.0); }, 3));
    report("4 chains + 2 ds_read_b64", 4, timeit([&]{ k<4, 1><<<blocks, 256>>>(out, iters, 1.0); }, 3));
    printf("\n");
  }
  return 0;
}
