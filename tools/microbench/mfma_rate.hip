// Microbenchmark: f32 MFMA 32x32x2 issue rate alone, with VALU fillers, and with LDS operand reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef float float16v __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = seed * (i % 97) * 0.01f - 0.3f;
  __syncthreads();
  float16v acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int lane = threadIdx.x & 63;
  float a = seed + lane * 0.001f, b = 0.5f - lane * 0.002f;
  uint32_t m = (lane & 1) << 31;
  int off = lane & 31;
  float4 bq = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float av = a, bv = b;
      if (MODE == 2) {  // one conflict-free ds_read_b32 per MFMA (the A gather)
        av = lds[((it & 127) << 5) + ((off ^ (2 * s)) ^ (lane >> 5))];
      }
      if (MODE == 3) {  // A gather + B from 4 conflict-free ds_read_b128 per 16 MFMAs
        av = lds[((it & 127) << 5) + ((off ^ (2 * s)) ^ (lane >> 5))];
        if ((s & 3) == 0) {
          const float4* bp = reinterpret_cast<const float4*>(lds + 4096 + (((it + lane) & 127) << 5));
          bq = bp[((s >> 2) + (lane >> 3)) & 7];
        }
        bv = (s & 3) == 0 ? bq.x : (s & 3) == 1 ? bq.y : (s & 3) == 2 ? bq.z : bq.w;
      }
      if (MODE == 4) {  // two conflict-free ds_read_b32 per MFMA
        av = lds[((it & 127) << 5) + ((off ^ (2 * s)) ^ (lane >> 5))];
        bv = lds[4096 + ((it & 127) << 5) + ((off ^ s) ^ (lane >> 5))];
      }
      if (MODE >= 1) {  // sign flips
        av = __uint_as_float(__float_as_uint(av) ^ m);
        bv = __uint_as_float(__float_as_uint(bv) ^ (m >> (s & 1)));
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
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
  const int iters = 4000;
  for (int r = 0; r < 20; ++r) k<0><<<256 * 4, 256>>>(out, iters, 1.0f);
  CK(hipDeviceSynchronize());
  for (int bpc : {1, 2, 4, 5, 8}) {
    int blocks = 256 * bpc;
    double fl = 2.0 * 2048 * 16.0 * iters * blocks * 4;  // 4 waves per block
    double t0 = timeit([&]{ k<0><<<blocks, 256>>>(out, iters, 1.0f); }, 3);
    double t1 = timeit([&]{ k<1><<<blocks, 256>>>(out, iters, 1.0f); }, 3);
    double t2 = timeit([&]{ k<2><<<blocks, 256>>>(out, iters, 1.0f); }, 3);
    double t3 = timeit([&]{ k<3><<<blocks, 256>>>(out, iters, 1.0f); }, 3);
    double t4 = timeit([&]{ k<4><<<blocks, 256>>>(out, iters, 1.0f); }, 3);
    printf("waves/SIMD %d: pure %.1f | +xor %.1f | +1 b32/MFMA %.1f | +1 b32 + b128/4 %.1f | +2 b32/MFMA %.1f TF\n", bpc, fl / t0 * 1e-12, fl / t1 * 1e-12, fl / t2 * 1e-12, fl / t3 * 1e-12, fl / t4 * 1e-12);
  }
  return 0;
}
