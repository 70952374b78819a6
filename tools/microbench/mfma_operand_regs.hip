// Microbenchmark: does it matter WHERE the LDS reads of the MFMA operands land (VGPR or AccVGPR) and where the
// accumulator lives?  mfma_f64_rate.hip shows v_mfma_f64_16x16x4_f64 dropping from 74 to 60 TFLOP/s when both operands of
// every instruction come from LDS (one ds_read_b64 each), independent of the number of waves: the LDS return data and the
// matrix pipe contend for the register file.  gfx950 has a unified 512-register file per lane: DS instructions can
// return into AccVGPRs and MFMA can source A / B from them.  Variants (operand class, accumulator class):
//   VV: operands in VGPRs, accumulator in VGPRs      VA: operands VGPRs, accumulator AccVGPRs (what the compiler picks)
//   AV: operands in AccVGPRs, accumulator VGPRs      AA: both AccVGPRs
// Reads are software-pipelined one step ahead (explicit s_waitcnt lgkmcnt), so LDS latency is not what is measured.
// f64: 2 ds_read_b64 per v_mfma_f64_16x16x4_f64.  f32: 1 ds_read_b128 + 4 ds_read_b32 per 4 v_mfma_f32_16x16x4_f32
// (the operand pattern of k_gp_mfma16x4<float>).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef double double4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

#define STEP64(OPC, ACCC)                                                                                          \
  asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %3"                                                        \
               : "=" OPC(na), "=" OPC(nb) : "v"(addr_a), "v"(addr_b) : "memory");                                 \
  asm volatile("s_waitcnt lgkmcnt(2)\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+" ACCC(acc) : OPC(ca), OPC(cb)); \
  ca = na; cb = nb;

template <int VARIANT>
__global__ __launch_bounds__(256) void k64(double* out, int iters, double seed) {
  __shared__ double lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed * (i % 97) * 0.01 - 0.3;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint32_t addr_a = uint32_t(lane) * 8u, addr_b = 16384u + uint32_t(lane ^ 5) * 8u;
  double4v acc = {0, 0, 0, 0};
  double ca, cb, na, nb;
  if (VARIANT == 0) {   // VV
    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %3" : "=v"(ca), "=v"(cb) : "v"(addr_a), "v"(addr_b) : "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 64; ++s) { STEP64("v", "v") }
    }
  } else if (VARIANT == 1) {   // VA
    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %3" : "=v"(ca), "=v"(cb) : "v"(addr_a), "v"(addr_b) : "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 64; ++s) { STEP64("v", "a") }
    }
  } else if (VARIANT == 2) {   // AV
    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %3" : "=a"(ca), "=a"(cb) : "v"(addr_a), "v"(addr_b) : "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 64; ++s) { STEP64("a", "v") }
    }
  } else {   // AA
    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %3" : "=a"(ca), "=a"(cb) : "v"(addr_a), "v"(addr_b) : "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 64; ++s) { STEP64("a", "a") }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + ca + cb;
}

// f32: per step one ds_read_b128 (B quad) + four ds_read_b32 (A words), four MFMAs
#define LOAD32(OPC, Q, W0, W1, W2, W3)                                                                             \
  asm volatile("ds_read_b128 %0, %5\n\tds_read_b32 %1, %6\n\tds_read_b32 %2, %6 offset:64\n\t"                    \
               "ds_read_b32 %3, %6 offset:128\n\tds_read_b32 %4, %6 offset:192"                                   \
               : "=" OPC(Q), "=" OPC(W0), "=" OPC(W1), "=" OPC(W2), "=" OPC(W3) : "v"(addr_b), "v"(addr_a) : "memory");
#define STEP32(OPC, ACCC)                                                                                          \
  LOAD32(OPC, nq, n0, n1, n2, n3)                                                                                  \
  asm volatile("s_waitcnt lgkmcnt(5)\n\t"                                                                          \
               "v_mfma_f32_16x16x4_f32 %0, %1, %5, %0\n\tv_mfma_f32_16x16x4_f32 %0, %2, %6, %0\n\t"                \
               "v_mfma_f32_16x16x4_f32 %0, %3, %7, %0\n\tv_mfma_f32_16x16x4_f32 %0, %4, %8, %0"                    \
               : "+" ACCC(acc) : OPC(c0), OPC(c1), OPC(c2), OPC(c3), OPC(cq[0]), OPC(cq[1]), OPC(cq[2]), OPC(cq[3])); \
  cq = nq; c0 = n0; c1 = n1; c2 = n2; c3 = n3;

template <int VARIANT>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float seed) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = seed * (i % 97) * 0.01f - 0.3f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint32_t addr_a = uint32_t(lane & 15) * 4u, addr_b = 16384u + uint32_t(lane) * 16u;
  float4v acc = {0, 0, 0, 0};
  float4v cq, nq;
  float c0, c1, c2, c3, n0, n1, n2, n3;
  if (VARIANT == 0) {
    LOAD32("v", cq, c0, c1, c2, c3)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 16; ++s) { STEP32("v", "v") }
    }
  } else if (VARIANT == 1) {
    LOAD32("v", cq, c0, c1, c2, c3)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 16; ++s) { STEP32("v", "a") }
    }
  } else if (VARIANT == 2) {
    LOAD32("a", cq, c0, c1, c2, c3)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 16; ++s) { STEP32("a", "v") }
    }
  } else {
    LOAD32("a", cq, c0, c1, c2, c3)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 16; ++s) { STEP32("a", "a") }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + c0 + c1 + c2 + c3 + cq[0];
}

// the same f32 MFMA stream with no LDS reads at all (register operands): the ceiling
__global__ __launch_bounds__(256) void k32_pure(float* out, int iters, float seed) {
  const int lane = threadIdx.x & 63;
  float4v acc = {0, 0, 0, 0};
  float a = seed + lane * 0.001f, b = 0.5f - lane * 0.002f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
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
  void* out; CK(hipMalloc(&out, size_t(cus) * 8 * 256 * sizeof(double)));
  double* o64 = (double*)out; float* o32 = (float*)out;
  const int iters = 400;   // x 64 MFMAs per iteration
  for (int r = 0; r < 20; ++r) k64<1><<<cus * 4, 256>>>(o64, iters, 1.0);
  CK(hipDeviceSynchronize());
  printf("f64: 2 ds_read_b64 per v_mfma_f64_16x16x4_f64 (TFLOP/s; operands / accumulator in V = VGPR, A = AccVGPR)\n");
  for (int bpc : {1, 2, 4, 8}) {
    const int blocks = cus * bpc;
    const double fl = 64.0 * iters * blocks * 4 * 2048.0;
    printf("  waves/SIMD %d:  VV %6.1f  VA %6.1f  AV %6.1f  AA %6.1f\n", bpc,
           fl / timeit([&]{ k64<0><<<blocks, 256>>>(o64, iters, 1.0); }, 3) * 1e-12,
           fl / timeit([&]{ k64<1><<<blocks, 256>>>(o64, iters, 1.0); }, 3) * 1e-12,
           fl / timeit([&]{ k64<2><<<blocks, 256>>>(o64, iters, 1.0); }, 3) * 1e-12,
           fl / timeit([&]{ k64<3><<<blocks, 256>>>(o64, iters, 1.0); }, 3) * 1e-12);
  }
  printf("f32: 1 ds_read_b128 + 4 ds_read_b32 per 4 v_mfma_f32_16x16x4_f32\n");
  for (int bpc : {1, 2, 4, 7, 8}) {
    const int blocks = cus * bpc;
    const double fl = 16.0 * 4 * iters * blocks * 4 * 2048.0;
    const double flp = 32.0 * iters * blocks * 4 * 2048.0;
    printf("  waves/SIMD %d:  pure %6.1f | VV %6.1f  VA %6.1f  AV %6.1f  AA %6.1f\n", bpc,
           flp / timeit([&]{ k32_pure<<<blocks, 256>>>(o32, iters, 1.0f); }, 3) * 1e-12,
           fl / timeit([&]{ k32<0><<<blocks, 256>>>(o32, iters, 1.0f); }, 3) * 1e-12,
           fl / timeit([&]{ k32<1><<<blocks, 256>>>(o32, iters, 1.0f); }, 3) * 1e-12,
           fl / timeit([&]{ k32<2><<<blocks, 256>>>(o32, iters, 1.0f); }, 3) * 1e-12,
           fl / timeit([&]{ k32<3><<<blocks, 256>>>(o32, iters, 1.0f); }, 3) * 1e-12);
  }
  return 0;
}
