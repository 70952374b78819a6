// kernels_fused.hip.hpp -- whole small programs in one launch: the LDS interpreter kernel k_ast_fused
// Included through kernels.hip.hpp.
#pragma once
#include "kernels_common.hip.hpp"

namespace gaast {

// ------------------------------------------------------------------------------------------
// Whole-AST kernel for small (grade-sparse) programs: ONE launch per evaluation.
//
// lane <-> batch item.  Every buffer of the plan (the bound inputs that are read, the cache
// buffers of the product operands, the root result) lives in LDS as one "slab" of S elements
// per item; S is odd so that the 64 lanes of a wave reading the same slab offset touch 64
// different banks.  Inputs are copied HBM -> LDS with coalesced loads (a block's rows are one
// contiguous range), the result goes back the same way: HBM traffic is exactly the inputs
// once plus the root once.  In between each lane runs the plan as a wave-uniform stream of
// 8-word lines fetched through the scalar cache one line ahead, in the reference's order:
// a Product is its comp-mul list grouped by result component, each component's entries in
// the reference's order, so all roundings are those of eval.rs:82.
//
// line = 32 x u32 (128 bytes, two s_load_dwordx16).  word 0 = header, [31:28] kind:
//   LINE_MACS  dst[11:0] begin[12] fresh[13] end[14] count[18:15] (1..10); words 2+3k, 3+3k, 4+3k =
//              slot k: left BYTE offset, right BYTE offset, sign mask (0 or 0x80000000), all
//              pre-computed on the host so that a slot costs no scalar instruction at all:
//                  acc = acc + ((slab[l] * slab[r]) ^ sign)            == + (l*r)*(+-1.0), exact
//              begin: acc = fresh ? 0.0 : slab[dst];   end: slab[dst] = acc
//              The body is straight-line per count; all operands are fetched before the first multiply.
//   LINE_MACS_GEN  same header, count <= 10; word 2+3k = left element | right element << 12,
//              word 3+3k = coefficient id (0: +1, 1: -1, >= 2: table[id-2]); multiplies by the
//              coefficient like eval.rs:82 (general metrics; rare).
//   LINE_MISC  count[20:15] (<= 30); words 2.. = element-wise micro-ops, [31:28] opcode:
//              ADD dst[11:0] src[23:12]   slab[dst] = slab[dst] + slab[src]     (graded.rs:74)
//              COPY dst[11:0] src[23:12]  slab[dst] = 0.0 + slab[src]           (graded.rs:195-201 then :74)
//              NEG dst                    slab[dst] = -slab[dst]                 (graded.rs:63)
//              ZERO dst count[23:12]      slab[dst..dst+count) = 0.0             (graded.rs:195-201)
//              INV / SQRT dst             eval.rs:106-109
// ------------------------------------------------------------------------------------------
enum : uint32_t { LINE_MACS = 0, LINE_MISC = 1, LINE_NOP = 2, LINE_MACS_GEN = 3 };
enum : uint32_t { UOP_ADD = 3, UOP_NEG = 4, UOP_ZERO = 5, UOP_INV = 6, UOP_SQRT = 7, UOP_COPY = 8 };

constexpr int FUSED_MAX_INPUTS = 8;
constexpr int FUSED_ITEMS = 64;    // items per workgroup: lane <-> item
constexpr int FUSED_GROUPS = 8;    // waves per workgroup: the independent result rows of a step are
                                   // dealt to the waves, all working on the same 64 slabs
constexpr int FUSED_THREADS = FUSED_ITEMS * FUSED_GROUPS;

template <typename T>
struct FusedArgs {
    const uint32_t* prog;      // 32-word lines
    const uint32_t* phase_tab; // per (phase, wave): first line, number of lines
    int n_phases;
    T coeff[6];             // general coefficients (c >= 2)
    int slab;               // S: elements per item, odd
    int zero_slot;          // slab offset of an element holding +0.0 (target of unused MAC slots)
    int n_in;
    const T* in_ptr[FUSED_MAX_INPUTS];
    int64_t in_stride[FUSED_MAX_INPUTS];
    int in_len[FUSED_MAX_INPUTS];
    int in_base[FUSED_MAX_INPUTS];
    int in_canon[FUSED_MAX_INPUTS];  // apply 0.0 + x while staging (input only read as a product operand)
    T* out_ptr;
    int64_t out_stride;
    int out_len, out_base;
    int64_t batch;
};

template <typename T>
__device__ __forceinline__ T sqrt_t(T x);
template <>
__device__ __forceinline__ float sqrt_t<float>(float x) { return __builtin_sqrtf(x); }
template <>
__device__ __forceinline__ double sqrt_t<double>(double x) { return __builtin_sqrt(x); }

template <typename T>
__device__ __forceinline__ void fused_misc(uint32_t w, T* __restrict__ my) {
    const uint32_t op = w >> 28;
    T* d = my + (w & 0xfffu);
    if (op == UOP_ADD) {
        *d = *d + my[(w >> 12) & 0xfffu];
    } else if (op == UOP_COPY) {   // the zero fill of a fresh buffer folded into its first add_grades_from: 0.0 + x
        *d = T(0) + my[(w >> 12) & 0xfffu];
    } else if (op == UOP_NEG) {
        *d = -*d;
    } else if (op == UOP_ZERO) {
        const uint32_t cnt = (w >> 12) & 0xfffu;
        for (uint32_t i = 0; i < cnt; ++i) d[i] = T(0);
    } else if (op == UOP_INV) {
        *d = T(1) / *d;
    } else if (op == UOP_SQRT) {
        *d = sqrt_t<T>(*d);
    }
}

__device__ __forceinline__ float xor_sign(float t, uint32_t mask) { return __uint_as_float(__float_as_uint(t) ^ mask); }
__device__ __forceinline__ double xor_sign(double t, uint32_t mask) {
    return __hiloint2double(__double2hiint(t) ^ int(mask), __double2loint(t));
}

template <typename T>
__device__ __forceinline__ T lds_at(const T* my, uint32_t byte_off) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const unsigned char*>(my) + byte_off);
}

struct FusedLine {
    uint4 q[8];  // 32 words, wave-uniform (scalar registers)
    __device__ __forceinline__ uint32_t word(int i) const {
        const uint4& v = q[i >> 2];
        return (i & 3) == 0 ? v.x : (i & 3) == 1 ? v.y : (i & 3) == 2 ? v.z : v.w;
    }
};

// CNT sign-only slots, straight-line: 2*CNT LDS reads in flight, then the sum.
template <typename T, int CNT>
__device__ __forceinline__ void fused_mac_n(uint32_t h, const FusedLine& L, T* __restrict__ my, T& acc) {
    T l[CNT], r[CNT];
#pragma unroll
    for (int k = 0; k < CNT; ++k) {
        l[k] = lds_at<T>(my, L.word(2 + 3 * k));
        r[k] = lds_at<T>(my, L.word(3 + 3 * k));
    }
    T* d = my + (h & 0xfffu);
    if (h & (1u << 12)) acc = (h & (1u << 13)) ? T(0) : *d;
#pragma unroll
    for (int k = 0; k < CNT; ++k) acc = acc + xor_sign(l[k] * r[k], L.word(4 + 3 * k));  // (l*r)*(+-1.0), exact
    if (h & (1u << 14)) *d = acc;
}

template <typename T>
__device__ __forceinline__ void fused_mac_line(uint32_t h, const FusedLine& L, T* __restrict__ my, T& acc) {
    switch ((h >> 15) & 15u) {  // wave-uniform: one jump per line, then straight-line code
    case 10: fused_mac_n<T, 10>(h, L, my, acc); break;
    case 9: fused_mac_n<T, 9>(h, L, my, acc); break;
    case 8: fused_mac_n<T, 8>(h, L, my, acc); break;
    case 7: fused_mac_n<T, 7>(h, L, my, acc); break;
    case 6: fused_mac_n<T, 6>(h, L, my, acc); break;
    case 5: fused_mac_n<T, 5>(h, L, my, acc); break;
    case 4: fused_mac_n<T, 4>(h, L, my, acc); break;
    case 3: fused_mac_n<T, 3>(h, L, my, acc); break;
    case 2: fused_mac_n<T, 2>(h, L, my, acc); break;
    case 1: fused_mac_n<T, 1>(h, L, my, acc); break;
    default: {
        T* d = my + (h & 0xfffu);
        if (h & (1u << 12)) acc = (h & (1u << 13)) ? T(0) : *d;
        if (h & (1u << 14)) *d = acc;
    }
    }
}

// General coefficients: literal eval.rs:82 arithmetic, counted (rare path).
template <typename T>
__device__ __forceinline__ void fused_mac_general(uint32_t h, const FusedLine& L, T* __restrict__ my, T& acc,
                                                  const T* __restrict__ ctab) {
    const uint32_t cnt = (h >> 15) & 15u;
    T* d = my + (h & 0xfffu);
    if (h & (1u << 12)) acc = (h & (1u << 13)) ? T(0) : *d;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        if (uint32_t(k) < cnt) {
            const uint32_t w = L.word(2 + 3 * k);
            const T t = my[w & 0xfffu] * my[(w >> 12) & 0xfffu];
            acc = acc + t * ctab[L.word(3 + 3 * k) & 7u];
        }
    }
    if (h & (1u << 14)) *d = acc;
}

template <typename T>
__global__ __launch_bounds__(FUSED_THREADS) void k_ast_fused(FusedArgs<T> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);
    const int tid = threadIdx.x;
    const int S = p.slab;
    const int64_t item0 = int64_t(blockIdx.x) * FUSED_ITEMS;
    const int nitems = int(p.batch - item0 < FUSED_ITEMS ? p.batch - item0 : FUSED_ITEMS);

    // ---- stage the inputs: element e of the block's contiguous row range -> (item, comp) ----
    for (int s = 0; s < p.n_in; ++s) {
        const int len = p.in_len[s];
        if (len <= 0) continue;
        const T* src = p.in_ptr[s] + item0 * p.in_stride[s];
        const int total = nitems * len;
        int it = tid / len, c = tid - it * len;
        const int dit = FUSED_THREADS / len, dc = FUSED_THREADS - dit * len;
        for (int e = tid; e < total; e += FUSED_THREADS) {
            T v = src[int64_t(it) * p.in_stride[s] + c];
            if (p.in_canon[s]) v = T(0) + v;
            smem[it * S + p.in_base[s] + c] = v;
            it += dit;
            c += dc;
            if (c >= len) {
                c -= len;
                ++it;
            }
        }
    }
    // coefficient table behind the slabs: [+1, -1, general...]
    T* ctab = smem + FUSED_ITEMS * S;
    if (tid < FUSED_ITEMS) smem[tid * S + p.zero_slot] = T(0);  // the item's zero element
    if (tid < 8) ctab[tid] = tid == 0 ? T(1) : tid == 1 ? T(-1) : p.coeff[tid - 2 < 6 ? tid - 2 : 0];
    __syncthreads();

    // ---- run the plan: this wave's share of every step, on this lane's item ----
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    T* my = smem + (tid & 63) * S;
    T acc = T(0);
    // Program lines come through the scalar cache (the stream is wave-uniform), ping-pong
    // buffered one line ahead; only the header is decoded on the scalar unit.
    const uint4* prog4 = reinterpret_cast<const uint4*>(p.prog);
    auto load_line = [&](FusedLine& L, uint32_t line) {
        const uint4* lp = prog4 + 8 * size_t(line);
#pragma unroll
        for (int i = 0; i < 8; ++i) L.q[i] = lp[i];
    };
    auto run_line = [&](const FusedLine& L, uint32_t line) {
        const uint32_t h = L.word(0);
        const uint32_t kind = h >> 28;
        if (kind == LINE_MACS) {
            fused_mac_line<T>(h, L, my, acc);
        } else if (kind == LINE_MACS_GEN) {
            fused_mac_general<T>(h, L, my, acc, ctab);
        } else if (kind == LINE_MISC) {  // element-wise arms: a plain loop over the line in memory
            const uint32_t cnt = (h >> 15) & 63u;
            const uint32_t* ops = p.prog + 32 * size_t(line) + 2;
            for (uint32_t k = 0; k < cnt; ++k) fused_misc<T>(ops[k], my);
        }
    };
    for (int ph = 0; ph < p.n_phases; ++ph) {
        const uint32_t first = p.phase_tab[2 * (ph * FUSED_GROUPS + wave)];
        const uint32_t n_lines = p.phase_tab[2 * (ph * FUSED_GROUPS + wave) + 1];
        if (n_lines > 0) {
            FusedLine A, B;
            load_line(A, first);
            for (uint32_t ln = 0; ln < n_lines; ln += 2) {
                if (ln + 1 < n_lines) load_line(B, first + ln + 1);
                run_line(A, first + ln);
                if (ln + 2 < n_lines) load_line(A, first + ln + 2);
                if (ln + 1 < n_lines) run_line(B, first + ln + 1);
            }
        }
        __syncthreads();  // the next step reads what every wave wrote
    }

    // ---- write the root result rows back, coalesced ----
    if (p.out_len > 0) {
        const int len = p.out_len;
        T* dst = p.out_ptr + item0 * p.out_stride;
        const int total = nitems * len;
        int it = tid / len, c = tid - it * len;
        const int dit = FUSED_THREADS / len, dc = FUSED_THREADS - dit * len;
        for (int e = tid; e < total; e += FUSED_THREADS) {
            dst[int64_t(it) * p.out_stride + c] = smem[it * S + p.out_base + c];
            it += dit;
            c += dc;
            if (c >= len) {
                c -= len;
                ++it;
            }
        }
    }
}

}  // namespace gaast
