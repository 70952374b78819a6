// Microbenchmark 2 for k_gp_mfma16's inner loop: per step 1 ds_read_b32 (A word) + 4 ds_read_b128 (B block), 16 x
// (v_xor_b32_dpp for A, v_xor_b32 for B, v_mfma_f32_16x16x1_4b_f32).  Variants:
//   0 as in the kernel | 1 no B sign | 2 LDS reads of step t+1 issued before the MFMAs of step t | 3 two accumulator chains
//   4 no LDS reads (operands evolve in registers) | 5 as 0 with s_setprio 2 | 6 = 2 + no B sign | 7 pure MFMA chain
//   8 A operand read per term from a +A / -A image pair (16 ds_read_b32 with lane-constant addresses, no vector instruction)
//   9 = 8 + B from a +B / -B image pair (address chosen by the block sign)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef float float16v __attribute__((ext_vector_type(16)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));

template <int CTRL> __device__ __forceinline__ uint32_t dpp(uint32_t v) { return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), CTRL, 0xf, 0xf, true)); }

template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k(float* out, int iters, float seed) {
  constexpr int ITEM = MODE == 9 ? 1040 : MODE == 8 ? 784 : 528;   // words per item: +A[, -A], +B[, -B], 16 pad
  constexpr int LDSW = 4 * ITEM + 64;
  __shared__ __attribute__((aligned(16))) float lds[LDSW];
  for (int i = threadIdx.x; i < LDSW; i += 64) lds[i] = seed * (i % 97) * 0.01f - 0.3f;
  __syncthreads();
  float16v acc, acc2;
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
  const int lane = threadIdx.x & 63, blk = lane >> 4, i = lane & 15;
  uint32_t m[16];
  for (int q = 0; q < 16; ++q) m[q] = uint32_t((lane >> (q & 3)) & 1) << 31;
  const uint32_t sign_bits = 0x5a3cu ^ (lane * 0x1111u);
  typedef __attribute__((address_space(3))) const unsigned char lds_bytes;
  lds_bytes* base = (lds_bytes*)(lds) + blk * ITEM * 4;
  uint32_t bq[4];
  for (int q = 0; q < 4; ++q) bq[q] = (uint32_t(i) << 6) | (uint32_t((q ^ (i >> 2)) & 3) << 4);
  uint32_t w0n = 0, bwn[16];
  auto load = [&](int a_hi, uint32_t& w0, uint32_t (&bw)[16]) {
    const uint32_t sx = (uint32_t(a_hi) << 6) | (uint32_t((a_hi >> 2) & 3) << 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const u4v v = *(__attribute__((address_space(3))) const u4v*)(base + 1024 + (bq[q] ^ sx));
      bw[4 * q] = v.x; bw[4 * q + 1] = v.y; bw[4 * q + 2] = v.z; bw[4 * q + 3] = v.w;
    }
    w0 = *(__attribute__((address_space(3))) const uint32_t*)(base + (a_hi << 6) + (i << 2));
  };
  if (MODE == 2 || MODE == 6) load(0, w0n, bwn);
  if (MODE == 4 || MODE == 7) { w0n = __float_as_uint(seed + lane); for (int q = 0; q < 16; ++q) bwn[q] = __float_as_uint(seed * q + lane); }
  if (MODE == 5) __builtin_amdgcn_s_setprio(2);
  if (MODE >= 8) {
    // lane constants: address of the term-k A word inside a block, in the image of its sign
    uint32_t ak[16];
    for (int q = 0; q < 16; ++q) ak[q] = uint32_t(((i ^ q) << 2) + ((m[q] >> 31) ? 1024 : 0));
    lds_bytes* abase = base;                 // +A at 0, -A at 1024, +B at 2048, -B at 3072
    for (int it = 0; it < iters; it += 16) {
#pragma unroll
      for (int a_hi = 0; a_hi < 16; ++a_hi) {
        uint32_t aw[16], bw[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) aw[q] = *(__attribute__((address_space(3))) const uint32_t*)(abase + ak[q] + (a_hi << 6));
        const uint32_t sx = (uint32_t(a_hi) << 6) | (uint32_t((a_hi >> 2) & 3) << 4);
        const uint32_t sb1 = (sign_bits >> a_hi) & 1u;
        const uint32_t sel = MODE == 9 ? (sb1 << 10) : 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const u4v v = *(__attribute__((address_space(3))) const u4v*)(base + 2048 + ((bq[q] ^ sx) | sel));
          bw[4 * q] = v.x; bw[4 * q + 1] = v.y; bw[4 * q + 2] = v.z; bw[4 * q + 3] = v.w;
        }
        const uint32_t sbit = MODE == 9 ? 0u : sb1 << 31;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
          acc = __builtin_amdgcn_mfma_f32_16x16x1f32(__uint_as_float(aw[kk]), __uint_as_float(bw[kk] ^ sbit), acc, 0, 0, 0);
        asm volatile("" ::: "memory");   // keep the steps' LDS reads in their steps
      }
    }
  } else
  for (int it = 0; it < iters; ++it) {
    const int a_hi = it & 15;
    uint32_t w0, bw[16];
    if (MODE == 2 || MODE == 6) {
      w0 = w0n;
#pragma unroll
      for (int q = 0; q < 16; ++q) bw[q] = bwn[q];
      load((a_hi + 1) & 15, w0n, bwn);
    } else if (MODE == 4 || MODE == 7) {
      w0 = w0n;
#pragma unroll
      for (int q = 0; q < 16; ++q) bw[q] = bwn[q];
    } else {
      load(a_hi, w0, bw);
    }
    const uint32_t sbit = (MODE == 1 || MODE == 6 || MODE == 7) ? 0u : ((sign_bits >> a_hi) & 1u) << 31;
    const uint32_t t7 = dpp<0x141>(w0), t15 = dpp<0x140>(w0);
    const uint32_t w4 = dpp<0x1B>(t7), w8 = dpp<0x141>(t15), w12 = dpp<0x1B>(t15);
    const uint32_t wb[4] = {w0, w4, w8, w12};
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const uint32_t b0 = wb[kk >> 2];
      uint32_t av;
      if (MODE == 7) av = b0;
      else if ((kk & 3) == 0) av = b0 ^ m[kk];
      else if ((kk & 3) == 1) av = dpp<0xB1>(b0) ^ m[kk];
      else if ((kk & 3) == 2) av = dpp<0x4E>(b0) ^ m[kk];
      else av = dpp<0x1B>(b0) ^ m[kk];
      const uint32_t bv = bw[kk] ^ sbit;
      if (MODE == 3 && (kk & 1)) acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(__uint_as_float(av), __uint_as_float(bv), acc2, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_16x16x1f32(__uint_as_float(av), __uint_as_float(bv), acc, 0, 0, 0);
    }
    if (MODE == 4) { w0n += 0x10u; }   // keep the operand work inside the loop
  }
  float sum = 0;
  for (int r = 0; r < 16; ++r) sum += acc[r] + acc2[r];
  out[blockIdx.x * 64 + threadIdx.x] = sum;
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
  float* out; CK(hipMalloc(&out, 256 * 16 * 64 * sizeof(float)));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const double clk = prop.clockRate * 1e3;
  const int iters = 4000;
  const char* names[10] = {"kernel's step", "no B sign", "LDS reads one step ahead", "two accumulator chains", "no LDS reads", "s_setprio 2", "ahead + no B sign", "pure chain", "A from +-images", "A and B from +-images"};
  for (int wps : {1, 2, 3, 4}) {
    const int blocks = 256 * 4 * wps;   // one-wave workgroups
    const double mfma_per_simd = 16.0 * iters * wps;
    double t[10];
    t[0] = timeit([&]{ k<0><<<blocks, 64>>>(out, iters, 1.0f); }, 3);
    t[1] = timeit([&]{ k<1><<<blocks, 64>>>(out, iters, 1.0f); }, 3);
    t[2] = timeit([&]{ k<2><<<blocks, 64>>>(out, iters, 1.0f); }, 3);
    t[3] = timeit([&]{ k<3><<<blocks, 64>>>(out, iters, 1.0f); }, 3);
    t[4] = timeit([&]{ k<4><<<blocks, 64>>>(out, iters, 1.0f); }, 3);
    t[5] = timeit([&]{ k<5><<<blocks, 64>>>(out, iters, 1.0f); }, 3);
    t[6] = timeit([&]{ k<6><<<blocks, 64>>>(out, iters, 1.0f); }, 3);
    t[7] = timeit([&]{ k<7><<<blocks, 64>>>(out, iters, 1.0f); }, 3);
    t[8] = timeit([&]{ k<8><<<blocks, 64>>>(out, iters, 1.0f); }, 3);
    t[9] = timeit([&]{ k<9><<<blocks, 64>>>(out, iters, 1.0f); }, 3);
    printf("waves/SIMD %d (cycles per MFMA at the nominal clock, 32 = peak):", wps);
    for (int q = 0; q < 10; ++q) printf("  %s %.1f |", names[q], t[q] * clk / mfma_per_simd);
    printf("\n");
  }
  return 0;
}
