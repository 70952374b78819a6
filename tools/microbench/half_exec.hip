// Microbenchmark (round 4): what does a wave pay for a vector / LDS instruction when only lanes 0-31 are active?
// If the hardware skips the inactive half of a wave64 instruction, a "one row per HALF wave" design (wave-uniform rows:
// scalar table loads, immediate offsets) costs nothing in vector time.  Also: int / f64 issue cost at 1, 2, 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 half_exec.hip -o half_exec
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

template <int KIND>   // 0: v_fma_f64, 1: v_add_u32, 2: ds_read_b64 (conflict-free), 3: v_mul_f64 + v_fma_f64 + 2 v_add_u32 mix
__global__ void k(float* out, int iters, int half) {
    __shared__ double lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    if (half && lane >= 32) { out[blockIdx.x * blockDim.x + threadIdx.x] = 0; return; }   // EXEC = lanes 0-31 for the rest of the kernel
    double acc[8];
    unsigned ia[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] = threadIdx.x + i; ia[i] = threadIdx.x * 8 + i; }
    double a = 1.0000001, b = 0.5;
    unsigned addr = (threadIdx.x & 31) * 8 + (threadIdx.x >> 5 & 1) * 256;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (KIND == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(addr));
            if (KIND == 2) { double v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(i * 512)); acc[i] = v; }
            if (KIND == 3) {
                asm volatile("v_mul_f64 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(addr));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[(i + 1) & 7]) : "v"(addr));
            }
        }
        if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i] + ia[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}

template <int KIND>
int run(const char* name, float* d, int threads, int blocks_per_cu) {
    const int iters = 20000, blocks = 256 * blocks_per_cu;
    for (int half = 0; half < 2; ++half) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 100, half);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, iters, half);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double per_wave_instr_ns = ms * 1e6 / (double(iters) * 8 * (KIND == 3 ? 4 : 1));
        // waves per SIMD = threads / 64 * blocks_per_cu / 4
        printf("%-28s %4d thr x %d blk/CU (%4.1f waves/SIMD) %s: %8.3f ms  %.3f ns per instruction per wave-slot  (~%.2f cyc at 2.4 GHz / waves-per-SIMD)\n", name, threads,
               blocks_per_cu, threads / 64.0 * blocks_per_cu / 4, half ? "lanes 0-31" : "all lanes ", ms, per_wave_instr_ns,
               per_wave_instr_ns * 2.4 / (threads / 64.0 * blocks_per_cu / 4));
    }
    return 0;
}

int main() {
    float* d;
    CK(hipMalloc(&d, sizeof(float) * 256 * 8 * 1024));
    for (int wps : {1, 2, 4}) {
        const int threads = 256, bpc = wps;
        if (run<0>("v_fma_f64", d, threads, bpc)) return 1;
        if (run<1>("v_add_u32", d, threads, bpc)) return 1;
        if (run<2>("ds_read_b64 x8 + wait", d, threads, bpc)) return 1;
        if (run<3>("mul64+fma64+2 add32", d, threads, bpc)) return 1;
    }
    CK(hipFree(d));
    return 0;
}
