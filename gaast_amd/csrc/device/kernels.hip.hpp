// Hand-written gfx950 kernels for every arm of the reference's evaluator (src/eval.rs).
//
//   k_axpy_map      GradedObj arm  eval.rs:45-50 -> graded.rs:67-78   res[k][i] = res[k][i] + in[k][i]
//   k_flip          Negation / Reverse / GradeInvolution  eval.rs:55-60,87-102 -> graded.rs:61-65
//   k_scalar_unary  ScalarUnaryOp  eval.rs:103-110
//   k_product_csr   Product arm    eval.rs:61-86  (any comp-mul list, reference summation order)
//   k_gp_dense      Product arm for dense geometric products, tiled in blade-bitmask space
//
// Data layout ("graded rows"): one row per batch item, grades concatenated ascending, see
// include/gaast_hip.h.  A row stride of 0 broadcasts one row to every item.
//
// The library is compiled with -ffp-contract=off: the reference computes
// `res += (left * right) * coeff` with three roundings (eval.rs:82) and so do the exact
// kernels here.  Only k_gp_dense uses explicit fused multiply-adds.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

namespace gaast {

// ------------------------------------------------------------------------------------------
// element-wise arms (HBM-bound; one thread per (item, mapped component))
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_axpy_map(T* __restrict__ res, int64_t res_stride,
                                                  const T* __restrict__ in, int64_t in_stride,
                                                  const uint32_t* __restrict__ map, int n_map,
                                                  int64_t batch) {
    const int64_t total = batch * n_map;
    for (int64_t idx = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
         idx += int64_t(gridDim.x) * blockDim.x) {
        const int64_t item = idx / n_map;
        const int j = int(idx - item * n_map);
        const uint32_t m = map[j];
        T* r = res + item * res_stride + (m & 0xffffu);
        *r = *r + in[item * in_stride + (m >> 16)];  // graded.rs:74  `*r = *r + i`
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_flip(T* __restrict__ res, int64_t res_stride,
                                              const uint32_t* __restrict__ offs, int n_offs,
                                              int64_t batch) {
    const int64_t total = batch * n_offs;
    for (int64_t idx = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
         idx += int64_t(gridDim.x) * blockDim.x) {
        const int64_t item = idx / n_offs;
        const int j = int(idx - item * n_offs);
        T* r = res + item * res_stride + offs[j];
        *r = -*r;  // graded.rs:63
    }
}

enum : int { SUNARY_INV = 0, SUNARY_SQRT = 1 };

template <typename T>
__global__ __launch_bounds__(256) void k_scalar_unary(T* __restrict__ res, int64_t res_stride,
                                                      int off, int op, int64_t batch) {
    for (int64_t item = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; item < batch;
         item += int64_t(gridDim.x) * blockDim.x) {
        T* r = res + item * res_stride + off;
        const T s = *r;
        // eval.rs:106-109; IEEE division and sqrt are correctly rounded on gfx950 for both types
        *r = op == SUNARY_INV ? T(1) / s : (sizeof(T) == 8 ? T(__builtin_sqrt(double(s)))
                                                           : T(__builtin_sqrtf(float(s))));
    }
}

// ------------------------------------------------------------------------------------------
// Product arm, exact: the comp-mul list grouped by result component (CSR by output), the
// entries of one output kept in the reference's order, so every output component sees the
// very same sequence of roundings as eval.rs:77-83.
//
// Block = `items` batch items; operand rows are staged in LDS with coalesced loads, then
// one thread per (item, output row) walks its entry list.  Entry = left offset | right
// offset << 16 (offsets into the staged rows) with the coefficient in a parallel array.
// ------------------------------------------------------------------------------------------
template <typename T>
struct CsrArgs {
    const T* left;
    const T* right;
    T* out;
    int64_t left_stride, right_stride, out_stride;  // elements; 0 = broadcast row
    int left_len, right_len;                        // row lengths staged in LDS
    int canon_left, canon_right;                    // operand is a raw input: apply 0.0 + x
    const uint32_t* row_start;                      // n_rows + 1
    const uint32_t* row_out;                        // output offset of each row
    const uint32_t* entries;
    const T* coeff;
    int n_rows;
    int beta;                                       // 1: accumulate into out; 0: out is fresh
    int64_t batch;
    int items;                                      // items per block
};

template <typename T>
__global__ __launch_bounds__(256) void k_product_csr(CsrArgs<T> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* ls = reinterpret_cast<T*>(smem_raw);
    T* rs = ls + int64_t(p.items) * p.left_len;
    const int64_t item0 = int64_t(blockIdx.x) * p.items;
    const int nitems = int(p.batch - item0 < p.items ? p.batch - item0 : p.items);
    const int tid = threadIdx.x, nthr = blockDim.x;

    // stage operand rows: consecutive threads read consecutive elements of consecutive rows
    const T zero = T(0);
    for (int i = tid; i < nitems * p.left_len; i += nthr) {
        const int it = i / p.left_len, c = i - it * p.left_len;
        T v = p.left[(item0 + it) * p.left_stride + c];
        ls[i] = p.canon_left ? zero + v : v;  // init_null_mv + add_grades_from: 0.0 + x
    }
    for (int i = tid; i < nitems * p.right_len; i += nthr) {
        const int it = i / p.right_len, c = i - it * p.right_len;
        T v = p.right[(item0 + it) * p.right_stride + c];
        rs[i] = p.canon_right ? zero + v : v;
    }
    __syncthreads();

    for (int w = tid; w < nitems * p.n_rows; w += nthr) {
        const int it = w / p.n_rows, row = w - it * p.n_rows;
        T* o = p.out + (item0 + it) * p.out_stride + p.row_out[row];
        T acc = p.beta ? *o : zero;
        const T* l = ls + int64_t(it) * p.left_len;
        const T* r = rs + int64_t(it) * p.right_len;
        const uint32_t e1 = p.row_start[row + 1];
        for (uint32_t e = p.row_start[row]; e < e1; ++e) {
            const uint32_t lr = p.entries[e];
            acc = acc + (l[lr & 0xffffu] * r[lr >> 16]) * p.coeff[e];  // eval.rs:82
        }
        *o = acc;
    }
}

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
//              NEG dst                    slab[dst] = -slab[dst]                 (graded.rs:63)
//              ZERO dst count[23:12]      slab[dst..dst+count) = 0.0             (graded.rs:195-201)
//              INV / SQRT dst             eval.rs:106-109
// ------------------------------------------------------------------------------------------
enum : uint32_t { LINE_MACS = 0, LINE_MISC = 1, LINE_NOP = 2, LINE_MACS_GEN = 3 };
enum : uint32_t { UOP_ADD = 3, UOP_NEG = 4, UOP_ZERO = 5, UOP_INV = 6, UOP_SQRT = 7 };

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
    int debug_skip;         // diagnostics only (GAAST_DEBUG_FUSED_SKIP): 1 = no compute, 2 = no staging, 4 = no write-back
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
    for (int s = 0; s < ((p.debug_skip & 2) ? 0 : p.n_in); ++s) {
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
    for (int ph = 0; ph < ((p.debug_skip & 1) ? 0 : p.n_phases); ++ph) {
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
    if (p.out_len > 0 && !(p.debug_skip & 4)) {
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

// ------------------------------------------------------------------------------------------
// Product arm, dense geometric product, tiled in blade-bitmask space.
//
// In bitmask space e_a e_b = s(a,b) m(a&b) e_{a^b}  (algebra.rs:73-83), a twisted
// XOR-convolution.  Split a blade into hi = a >> 4 and lo = a & 15:
//     s(a,b) = s_hi(a_hi,b_hi) * s_lo(a_lo,b_lo) * (-1)^(|a_hi| |b_lo|)
// so an aligned 16-block of A times an aligned 16-block of B lands in exactly one 16-block
// of C, with a compile-time sign pattern (two variants, by the parity of |a_hi|) and one
// run-time sign per (a_hi, b_hi).  The lo four basis vectors must square to +1; the
// metric of the others (+1/-1/0) is folded into the per-block sign / zero factor:
//     (-1)^|a_hi & b_hi & NEG|  and  [a_hi & b_hi & ZERO == 0].
//
// Mapping: lane <-> c_hi (one 16-component block of the result in 16 accumulators),
// 2^(n-4) lanes per item.  Both operands of the item sit in LDS in bitmask order (scattered
// there through the index table while loading the graded rows coalesced).  Per step a_hi
// every lane reads the A block a_hi (a broadcast) and the B block a_hi ^ c_hi (a lane
// permutation of the blocks: conflict-free with the quad swizzle below), applies the
// block sign to B and issues 256 FMAs.  With b = a ^ c the block sign is
//     (-1)^( u(a_hi) + parity(c_hi & M(a_hi)) ),  M = sp(a_hi) ^ (a_hi & NEG),
// sp = exclusive suffix parity, u = parity(a_hi & sp) ^ parity(a_hi & NEG): M and u are
// wave-uniform (scalar ALU), the lane pays and + popcount + shift.
// ------------------------------------------------------------------------------------------
template <typename T>
struct DenseArgs {
    const T* left;
    const T* right;
    T* out;
    int64_t left_stride, right_stride, out_stride;
    const uint32_t* left_map;   // per loaded component: row offset | LDS image position << 16 | negate << 31
    const uint32_t* right_map;
    int left_count, right_count;
    int left_full, right_full;  // 1: every blade is loaded, no zero fill needed
    int left_contig, right_contig;  // 1: row offsets are 0,1,2,... and rows are 16-byte aligned: vector loads
    const int32_t* out_map;     // per bitmask: offset in the out row, or -1
    int canon_left, canon_right;
    int n;                      // vector-space dimension, 4 <= n
    uint32_t neg_hi, zero_hi;   // metric signature of basis vectors 4.. (bit i <-> vector 4+i)
    int beta;
    int64_t batch;
    int debug_skip;             // diagnostics only (GAAST_DEBUG_DENSE_SKIP): 2 = no staging, 4 = no write-back
};

// physical position of blade bitmask m inside an operand's LDS image: blocks of 16, the four
// 16-byte quads of block x rotated by (x >> 2) & 3 so that 16 lanes reading the same logical
// quad of 16 different blocks touch 16 different bank quads.
__device__ __forceinline__ int dense_lds_pos(int m) {
    const int x = m >> 4, lo = m & 15;
    return (x << 4) | ((((lo >> 2) ^ (x >> 2)) & 3) << 2) | (lo & 3);
}


// Scatter the operand rows of `nitems` items into their LDS images.  A map word is
//   row offset [15:0] | position in the LDS image [30:16] | negate [31]
// (the position is computed on the host for the kernel that consumes the image).  Loads of a
// trip are all issued before the first use.  When the offsets are simply 0, 1, 2, ... (an
// operand holding every blade) four consecutive components move per 16-byte load.
template <typename T, int THREADS>
__device__ __forceinline__ void stage_operands(const T* __restrict__ src, int64_t stride, const uint32_t* __restrict__ map,
                                               int count, int contig4, int canon, T* __restrict__ images,
                                               int image_stride, int nitems, int tid) {
    constexpr int U = 4;
    if (contig4) {
        const int total4 = (nitems * count) >> 2;  // count % 4 == 0
        const int count4 = count >> 2;
        for (int e0 = tid; e0 < total4; e0 += THREADS * U) {
            uint4 m[U];
            T v[U][4];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int e = e0 + k * THREADS;
                if (e < total4) {
                    const int it = e / count4, j4 = e - it * count4;
                    m[k] = reinterpret_cast<const uint4*>(map)[j4];
                    const T* rp = src + int64_t(it) * stride + (j4 << 2);
                    if (sizeof(T) == 4) {
                        const float4 f = *reinterpret_cast<const float4*>(rp);
                        v[k][0] = T(f.x); v[k][1] = T(f.y); v[k][2] = T(f.z); v[k][3] = T(f.w);
                    } else {
                        const double2 d0 = reinterpret_cast<const double2*>(rp)[0], d1 = reinterpret_cast<const double2*>(rp)[1];
                        v[k][0] = T(d0.x); v[k][1] = T(d0.y); v[k][2] = T(d1.x); v[k][3] = T(d1.y);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int e = e0 + k * THREADS;
                if (e < total4) {
                    const int it = e / count4;
                    T* img = images + it * image_stride;
                    const uint32_t mm[4] = {m[k].x, m[k].y, m[k].z, m[k].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        T x = v[k][c];
                        if (canon) x = T(0) + x;  // the reference's zero-init + add_grades_from copy: 0.0 + x
                        if (mm[c] >> 31) x = -x;  // a folded Negation / Reverse / GradeInvolution of this grade
                        img[(mm[c] >> 16) & 0x7fffu] = x;
                    }
                }
            }
        }
        return;
    }
    const int total = nitems * count;
    for (int e0 = tid; e0 < total; e0 += THREADS * U) {
        uint32_t m[U];
        T v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int e = e0 + k * THREADS;
            m[k] = e < total ? map[e % count] : 0u;
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int e = e0 + k * THREADS;
            v[k] = e < total ? src[int64_t(e / count) * stride + (m[k] & 0xffffu)] : T(0);
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int e = e0 + k * THREADS;
            if (e < total) {
                T x = v[k];
                if (canon) x = T(0) + x;
                if (m[k] >> 31) x = -x;
                images[(e / count) * image_stride + ((m[k] >> 16) & 0x7fffu)] = x;
            }
        }
    }
}

__device__ __forceinline__ constexpr int lo_reorder_parity(int a, int b) {
    // parity of #{(p,q): p in a, q in b, p > q} for 4-bit a, b
    int par = 0;
    for (int p = 1; p < 4; ++p)
        if ((a >> p) & 1)
            for (int q = 0; q < p; ++q) par ^= (b >> q) & 1;
    return par;
}

template <typename T>
__device__ __forceinline__ T fma_t(T a, T b, T c);
template <>
__device__ __forceinline__ float fma_t<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <>
__device__ __forceinline__ double fma_t<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }

// ---- 16x16 block product, generic (used for f64): 256 scalar FMAs with folded signs ----------
template <typename T, bool ODD>
__device__ __forceinline__ void gp_block16(const T (&A)[16], const T (&B)[16], T (&C)[16]) {
#pragma unroll
    for (int al = 0; al < 16; ++al) {
#pragma unroll
        for (int bl = 0; bl < 16; ++bl) {
            const int neg = lo_reorder_parity(al, bl) ^ (ODD ? (__builtin_popcount(bl) & 1) : 0);
            C[al ^ bl] = neg ? fma_t<T>(-A[al], B[bl], C[al ^ bl]) : fma_t<T>(A[al], B[bl], C[al ^ bl]);
        }
    }
}

// ---- 16x16 block product, f32: 128 v_pk_fma_f32, no operand shuffles ---------------------------
// Accumulator pair j holds C[2j], C[2j+1].  For the term with left component a, the low half
// needs B[a ^ 2j] and the high half B[a ^ 2j ^ 1]: the two halves of ONE B pair, swapped when a
// is odd; A[a] is one half of an A pair, broadcast.  op_sel / op_sel_hi pick the halves and
// neg_lo / neg_hi carry the compile-time signs, so every FMA is a single instruction.
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef double double2v __attribute__((ext_vector_type(2)));

template <int X, int NL, int NH>
__device__ __forceinline__ void pk_fma_sel(float2v& c, const float2v& a, const float2v& b) {
    if constexpr (X == 0 && NL == 0 && NH == 0)
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(c) : "v"(a), "v"(b));
    else if constexpr (X == 0 && NL == 1 && NH == 0)
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "+v"(c) : "v"(a), "v"(b));
    else if constexpr (X == 0 && NL == 0 && NH == 1)
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "+v"(c) : "v"(a), "v"(b));
    else if constexpr (X == 0 && NL == 1 && NH == 1)
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(c) : "v"(a), "v"(b));
    else if constexpr (X == 1 && NL == 0 && NH == 0)
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "+v"(c) : "v"(a), "v"(b));
    else if constexpr (X == 1 && NL == 1 && NH == 0)
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(c) : "v"(a), "v"(b));
    else if constexpr (X == 1 && NL == 0 && NH == 1)
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "+v"(c) : "v"(a), "v"(b));
    else
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(c) : "v"(a), "v"(b));
}

template <bool ODD, int IDX>
__device__ __forceinline__ void gp_block16_pk(const float2v (&A2)[8], const float2v (&B2)[8], float2v (&C2)[8]) {
    if constexpr (IDX < 128) {
        constexpr int a = IDX >> 3, j = IDX & 7, c0 = 2 * j;
        constexpr int b_lo = a ^ c0, b_hi = a ^ c0 ^ 1;
        constexpr int nl = lo_reorder_parity(a, b_lo) ^ (ODD ? (__builtin_popcount(b_lo) & 1) : 0);
        constexpr int nh = lo_reorder_parity(a, b_hi) ^ (ODD ? (__builtin_popcount(b_hi) & 1) : 0);
        pk_fma_sel<(a & 1), nl, nh>(C2[j], A2[a >> 1], B2[b_lo >> 1]);
        gp_block16_pk<ODD, IDX + 1>(A2, B2, C2);
    }
}

// one a_hi step for a lane: load the blocks, apply the block sign, 256 multiply-adds
template <typename T>
struct BlockStep;

template <>
struct BlockStep<float> {
    float2v C2[8];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < 8; ++i) C2[i] = float2v{0.f, 0.f};
    }
    template <bool ODD>
    __device__ __forceinline__ void step(const float* As, const float* Bs, int a_hi, int b_hi, float sgn) {
        const int sa = (a_hi >> 2) & 3, sb = (b_hi >> 2) & 3;
        const float4v* ap = reinterpret_cast<const float4v*>(As + (a_hi << 4));
        const float4v* bp = reinterpret_cast<const float4v*>(Bs + (b_hi << 4));
        float2v A2[8], B2[8];
        const float2v s2 = float2v{sgn, sgn};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4v va = ap[q ^ sa];
            const float4v vb = bp[q ^ sb];
            A2[2 * q] = float2v{va.x, va.y};
            A2[2 * q + 1] = float2v{va.z, va.w};
            B2[2 * q] = float2v{vb.x, vb.y} * s2;
            B2[2 * q + 1] = float2v{vb.z, vb.w} * s2;
        }
        gp_block16_pk<ODD, 0>(A2, B2, C2);
    }
    __device__ __forceinline__ float get(int i) const { return (i & 1) ? C2[i >> 1].y : C2[i >> 1].x; }
};

template <>
struct BlockStep<double> {
    double C[16];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < 16; ++i) C[i] = 0.0;
    }
    template <bool ODD>
    __device__ __forceinline__ void step(const double* As, const double* Bs, int a_hi, int b_hi, double sgn) {
        const int sa = (a_hi >> 2) & 3, sb = (b_hi >> 2) & 3;
        // a "quad" is 4 components = two 16-byte halves for f64
        const double2v* ap = reinterpret_cast<const double2v*>(As + (a_hi << 4));
        const double2v* bp = reinterpret_cast<const double2v*>(Bs + (b_hi << 4));
        double A[16], B[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double2v a0 = ap[2 * (q ^ sa)], a1 = ap[2 * (q ^ sa) + 1];
            const double2v b0 = bp[2 * (q ^ sb)], b1 = bp[2 * (q ^ sb) + 1];
            A[4 * q + 0] = a0.x; A[4 * q + 1] = a0.y; A[4 * q + 2] = a1.x; A[4 * q + 3] = a1.y;
            B[4 * q + 0] = b0.x * sgn; B[4 * q + 1] = b0.y * sgn; B[4 * q + 2] = b1.x * sgn; B[4 * q + 3] = b1.y * sgn;
        }
        gp_block16<double, ODD>(A, B, C);
    }
    __device__ __forceinline__ double get(int i) const { return C[i]; }
};

template <typename T, bool DEGENERATE, int THREADS>
__global__ __launch_bounds__(THREADS) void k_gp_dense(DenseArgs<T> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);
    const int n = p.n;
    const int N = 1 << n;
    const int hbits = n - 4;
    const int LPI = 1 << hbits;                   // lanes per item
    const int IPB = THREADS >> hbits;             // items per group (>= 1)
    const int item_stride = 2 * N + (IPB > 1 ? 4 : 0);  // +16 B: de-phase the items' A broadcasts
    const int tid = threadIdx.x;
    const T zero = T(0);
    const int64_t num_groups = (p.batch + IPB - 1) / IPB;

    const int it = tid >> hbits;
    const int c_hi = tid & (LPI - 1);
    int32_t om[16];  // where this lane's 16 results go in the graded row (the same for every group)
#pragma unroll
    for (int i = 0; i < 16; ++i) om[i] = p.out_map[(c_hi << 4) + i];

    // persistent workgroups: each walks the groups blockIdx.x, blockIdx.x + gridDim.x, ...
    for (int64_t grp = blockIdx.x; grp < num_groups; grp += gridDim.x) {
        const int64_t item0 = grp * IPB;
        const int nitems = int(p.batch - item0 < IPB ? p.batch - item0 : IPB);
        // ---- both operands of every item of the group into LDS, in bitmask order ----
        if (!(p.debug_skip & 2)) {
            if (!p.left_full || !p.right_full) {
                for (int i = tid; i < nitems * item_stride; i += THREADS) smem[i] = zero;
                __syncthreads();
            }
            stage_operands<T, THREADS>(p.left + item0 * p.left_stride, p.left_stride, p.left_map, p.left_count, p.left_contig,
                                       p.canon_left, smem, item_stride, nitems, tid);
            stage_operands<T, THREADS>(p.right + item0 * p.right_stride, p.right_stride, p.right_map, p.right_count,
                                       p.right_contig, p.canon_right, smem + N, item_stride, nitems, tid);
        }
        __syncthreads();

        if (it < nitems) {
            const T* As = smem + it * item_stride;
            const T* Bs = As + N;
            BlockStep<T> acc;
            acc.init();

            // Two passes: first the A blocks with |a_hi| even, then those with |a_hi| odd -- each
            // pass has ONE straight-line body (the sign pattern of the 16x16 block depends on that
            // parity), so the accumulators never cross a branch.  a_hi = 2i + (parity(i) ^ pass).
            auto one_step = [&](auto odd_tag, int a_hi) {
                constexpr bool ODD = decltype(odd_tag)::value;
                // wave-uniform part of the block sign
                uint32_t sp = uint32_t(a_hi) >> 1;
                sp ^= sp >> 1;
                sp ^= sp >> 2;
                sp ^= sp >> 4;
                sp ^= sp >> 8;
                const uint32_t M = sp ^ (uint32_t(a_hi) & p.neg_hi);
                const uint32_t u = (__builtin_popcount(uint32_t(a_hi) & sp) ^
                                    __builtin_popcount(uint32_t(a_hi) & p.neg_hi)) & 1u;
                // lane part
                const uint32_t sbit = (u ^ uint32_t(__builtin_popcount(uint32_t(c_hi) & M))) & 1u;
                T sgn = sbit ? T(-1) : T(1);
                if (DEGENERATE) {
                    if (uint32_t(a_hi) & ~uint32_t(c_hi) & p.zero_hi) sgn = zero;
                }
                acc.template step<ODD>(As, Bs, a_hi, a_hi ^ c_hi, sgn);
            };
            const int half = LPI >> 1;
            for (int i = 0; i < half; ++i) one_step(std::false_type{}, (i << 1) | (__builtin_popcount(uint32_t(i)) & 1));
            for (int i = 0; i < half; ++i) one_step(std::true_type{}, (i << 1) | ((__builtin_popcount(uint32_t(i)) & 1) ^ 1));

            // ---- scatter the 16 accumulators to their positions in the graded row ----
            T* orow = p.out + (item0 + it) * p.out_stride;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int32_t off = om[i];
                if (off >= 0) orow[off] = p.beta ? orow[off] + acc.get(i) : acc.get(i);
            }
        }
        __syncthreads();  // the LDS image is rewritten by the next group
    }
}

// ------------------------------------------------------------------------------------------
// Product arm, dense geometric product on the matrix cores (f32, n >= 10).
//
// With lo = 5 bits (blocks of 32) the per-item product is, for every A block a_hi,
//     C[c_lo][c_hi] += sum_k  M[c_lo][k] * Bs[k][c_hi],      k = b_lo,
//     M[i][k]  = s_lo(i^k, k) (-1)^(|a_hi| |k|) A[a_hi][i ^ k]             (32 x 32, built on the fly)
//     Bs[k][j] = s_blk(a_hi, b_hi(j)) B[b_hi(j)][k],   b_hi(j) = a_hi ^ c_hi(j)  (32 x 32 columns)
// i.e. a genuine 32x32x32 GEMM tile per (a_hi, 32 result columns): 16 v_mfma_f32_32x32x2_f32.
// The MFMA A operand of lane l is M[i = l&31][k = 2s + (l>>5)]: one ds_read_b32 of the A block
// at a lane-permuted position (conflict-free: the 32 lanes of a half-wave read a permutation
// of the block's 32 dwords) and one sign flip with a lane-constant mask.  The B operand of
// lane l is Bs[k = 2s + (l>>5)][j = l&31]: the 16 components of its own B block with index
// parity (l>>5), stored de-interleaved so that they are 64 contiguous bytes (4 ds_read_b128,
// quads rotated by (x>>1)&7 per block x: 16 lanes reading the same logical quad of 16 blocks
// hit 16 different bank quads), times the block sign.  f32 MFMA accumulates like a k-ordered
// fmaf chain, at the vector FMA rate, without occupying the vector ALUs.
// Requires the low FIVE basis vectors to square to +1.
// ------------------------------------------------------------------------------------------
typedef float float16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ constexpr int lo5_reorder_parity(int a, int b) {
    int par = 0;
    for (int p = 1; p < 5; ++p)
        if ((a >> p) & 1)
            for (int q = 0; q < p; ++q) par ^= (b >> q) & 1;
    return par;
}

// LDS position of blade m inside the B image: block x = m >> 5; inside it the components are
// de-interleaved by the parity of k = m & 31 (even k first), 8 quads rotated by (x >> 1) & 7.
__device__ __forceinline__ int mfma_b_pos(int m) {
    const int x = m >> 5, k = m & 31;
    const int lq = ((k & 1) << 2) | (k >> 3);   // logical quad: parity * 4 + (k/2)/4
    return (x << 5) | (((lq ^ (x >> 1)) & 7) << 2) | ((k >> 1) & 3);
}

template <bool DEGENERATE, int THREADS>
__global__ __launch_bounds__(THREADS) void k_gp_mfma32(DenseArgs<float> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    const int n = p.n;
    const int N = 1 << n;
    const int hbits = n - 5;
    const int H = 1 << hbits;                     // number of 32-blocks
    const int WPI = H >> 5;                       // waves per item (32 result columns each)
    const int IPB = (THREADS >> 6) / WPI;         // items per block (>= 1)
    const int item_stride = 2 * N;
    const int tid = threadIdx.x;
    const int64_t item0 = int64_t(blockIdx.x) * IPB;
    const int nitems = int(p.batch - item0 < IPB ? p.batch - item0 : IPB);

    // ---- stage both operands in bitmask order (B de-interleaved and quad-rotated) ----
    if (!p.left_full || !p.right_full) {
        for (int i = tid; i < nitems * item_stride; i += THREADS) smem[i] = 0.f;
        __syncthreads();
    }
    if (!(p.debug_skip & 2)) {
        stage_operands<float, THREADS>(p.left + item0 * p.left_stride, p.left_stride, p.left_map, p.left_count, p.left_contig,
                                       p.canon_left, smem, item_stride, nitems, tid);
        stage_operands<float, THREADS>(p.right + item0 * p.right_stride, p.right_stride, p.right_map, p.right_count,
                                       p.right_contig, p.canon_right, smem + N, item_stride, nitems, tid);
    }
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63;
    const int it = wave / WPI, tile = wave - it * WPI;
    if (it < nitems) {
        const float* As = smem + it * item_stride;
        const float* Bs = As + N;
        const int i = lane & 31, h = lane >> 5;
        const int c_hi = (tile << 5) | i;

        // lane constants: sign masks of the A operand for both parities of |a_hi|, and the
        // byte offset of A[i ^ k] inside a block, for k = 2s + h
        uint32_t amask[16];   // for |a_hi| even; flipped in place to the odd-parity pattern between the passes
        uint32_t aoff[16];
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
            const int k = 2 * s2 + h;
            const int a_lo = i ^ k;
            int par = 0;
            for (int pp = 1; pp < 5; ++pp)
                if ((a_lo >> pp) & 1) par ^= __builtin_popcount(k & ((1 << pp) - 1)) & 1;
            amask[s2] = uint32_t(par) << 31;
            aoff[s2] = uint32_t(a_lo) << 2;
        }

        // one accumulator chain per wave: a dependent f32 MFMA issues back to back (measured: a
        // second, independent chain changes nothing; what costs is the LDS -> VGPR operand traffic,
        // about 11 cycles of matrix-pipe time per ds_read_b32 and 24 per ds_read_b128 --
        // tools/microbench/mfma_rate.hip)
        float16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;

        const unsigned char* As_b = reinterpret_cast<const unsigned char*>(As);
        auto one_step = [&](int a_hi) {
            // block sign: wave-uniform part on the scalar unit, lane part = and + popcount
            uint32_t sp = uint32_t(a_hi) >> 1;
            sp ^= sp >> 1;
            sp ^= sp >> 2;
            sp ^= sp >> 4;
            sp ^= sp >> 8;
            const uint32_t M = sp ^ (uint32_t(a_hi) & p.neg_hi);
            const uint32_t u = (__builtin_popcount(uint32_t(a_hi) & sp) ^
                                __builtin_popcount(uint32_t(a_hi) & p.neg_hi)) & 1u;
            const uint32_t bmask = ((u ^ uint32_t(__builtin_popcount(uint32_t(c_hi) & M))) & 1u) << 31;
            float bscale = 1.f;
            if (DEGENERATE) {
                if (uint32_t(a_hi) & ~uint32_t(c_hi) & p.zero_hi) bscale = 0.f;
            }
            const int x = a_hi ^ c_hi;
            const int rot = (x >> 1) & 7;
            const float4v* bp = reinterpret_cast<const float4v*>(Bs + (x << 5));
            float bv[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4v v = bp[((h << 2) | q) ^ rot];
                bv[4 * q + 0] = v.x; bv[4 * q + 1] = v.y; bv[4 * q + 2] = v.z; bv[4 * q + 3] = v.w;
            }
            const uint32_t abase = uint32_t(a_hi) << 7;
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
                float a = *reinterpret_cast<const float*>(As_b + abase + aoff[s2]);
                a = __uint_as_float(__float_as_uint(a) ^ amask[s2]);
                float b = __uint_as_float(__float_as_uint(bv[s2]) ^ bmask);
                if (DEGENERATE) b *= bscale;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
        };
        const int half = H >> 1;
        for (int t2 = 0; t2 < half; ++t2) one_step((t2 << 1) | (__builtin_popcount(uint32_t(t2)) & 1));
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2)  // (-1)^(|a_hi| |k|) for odd |a_hi|: flip where |k| is odd, k = 2 s2 + h
            amask[s2] ^= uint32_t((__builtin_popcount(uint32_t(s2)) + h) & 1) << 31;
        for (int t2 = 0; t2 < half; ++t2) one_step((t2 << 1) | ((__builtin_popcount(uint32_t(t2)) & 1) ^ 1));

        // ---- accumulator (row = c_lo, column = this lane's c_hi) -> graded row ----
        float* orow = p.out + (item0 + it) * p.out_stride;
        const int32_t* om = p.out_map + (c_hi << 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c_lo = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int32_t off = om[c_lo];
            if (off >= 0 && !((p.debug_skip & 4) && acc[r] != 12345.f)) orow[off] = p.beta ? orow[off] + acc[r] : acc[r];
        }
    }
}

// ------------------------------------------------------------------------------------------
// OPT-IN fast path (GAAST_FLAG_SPINOR_GEMM): the geometric product of a non-degenerate algebra
// with n = 12 through its matrix representation -- 16x fewer multiply-adds than the bilinear
// contraction, all of them on the matrix cores.  NOT the reference's algorithm: same result in
// exact arithmetic, different roundings (norm-wise error bound), so it is never
// selected unless the host asks for it.
//
// Cl(p,q), p+q = 12, over the complex numbers is the algebra of 64 x 64 matrices.  With the
// Jordan-Wigner generators gamma_{2j} = Z..Z X_j, gamma_{2j+1} = Z..Z Y_j (times i for the vectors
// that square to -1), a blade e_S is i^k(S) X^x(S) Z^z(S), a Pauli string; S -> (x, z) is a
// bijection onto 6-bit pairs.  (X^x Z^z)[c^x][c] = (-1)^|c & z|, so for the multivector A
//     M_A[c ^ x][c] = sum_z (-1)^|c & z| * i^k(x,z) A_{S(x,z)}       -- a Walsh-Hadamard transform
// over z of row x of the re-indexed components; the product is C = M_A M_B (complex 64^3
// GEMM = 3 real ones = 384 v_mfma_f32_32x32x2_f32 per item instead of 8192); the inverse
// transform of the skewed diagonals of C gives the components back.
//
// Workgroup = 256 threads, persistent over items.  LDS: four 64 x 65 f32 planes (A re/im, B re/im;
// the +1 column makes the row-wise and the XOR-skewed column-wise accesses conflict-free, and
// 4160 words = 65 x 256 B lets one ds_read2st64_b32 fetch re and im together).
//   1. scatter the graded rows into W[x][z] from registers (16-bit table entries, branch-free)
//   2. 256 threads = 256 row transforms (2 operands x re/im x 64 rows), 64 values in registers,
//      written back in place: S_A[x][c] = M_A[c^x][c]; S_B[x][r] = M_B[r][r^x] -- the shift by x
//      of B's transform is a sign (-1)^|x&z| on its input, folded into the right operand's table --
//      so both MFMA operand gathers hit 32 distinct banks
//   3. wave w owns the 32 x 32 complex tile (w>>1, w&1): per k-pair two LDS reads and THREE MFMAs
//      (X = Ar Br, Y = Ai Bi, Z = (Ar+Ai)(Br+Bi); Re = X - Y, Im = Z - X - Y), operands of the next
//      step read while the current MFMAs run
//   4. C tiles go back to LDS skewed (by r, conflict-free); two threads per row fold and transform 32 points each;
//      every transformed value is the component of one blade (real part for even k, imaginary
//      for odd k), gathered in row order so that the stores to HBM are coalesced.
// The operands of the next item are loaded into registers during steps 2-4.
// Measured (profiles/r01_r12s_*): 32 M products/s at B = 65536, 8.7x the contraction kernel.
// ------------------------------------------------------------------------------------------
struct SpinorArgs {
    const float* left;
    const float* right;
    float* out;
    int64_t left_stride, right_stride, out_stride;
    // 4096 16-bit entries each, indexed by ROW OFFSET (two per word).
    //  operands: bit 0 = negate (folded unary signs, i^2, the right operand's shift), bit 1 = plane
    //            (0 real, 1 imaginary: parity of k), bits [14:2] = x*65+z, i.e. entry & 0x7ffc is the byte
    //            offset inside a plane; offsets that hold nothing point at a padding word (x*65+z = 64)
    //  result:   bit 0 = negate, bit 1 = nothing to store, bits [15:2] = plane*4160 + x*65+z
    const uint16_t* left_map;
    const uint16_t* right_map;
    const uint16_t* out_map;
    int left_len, right_len;
    int out_full;
    int left_full, right_full;
    int canon_left, canon_right;
    int beta;
    int64_t batch;
    int debug_skip;
};

template <int N>
__device__ __forceinline__ void wht(float (&v)[N]) {
#pragma unroll
    for (int hlf = 1; hlf < N; hlf <<= 1) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if ((i & hlf) == 0) {
                const float a = v[i], b = v[i | hlf];
                v[i] = a + b;
                v[i | hlf] = a - b;
            }
        }
    }
}

// Persistent workgroups (grid = resident blocks): the three tables live in registers for the
// whole launch and the operands of the NEXT item are fetched while the matrix cores work on the
// current one, so no phase waits on HBM latency.
__global__ __launch_bounds__(256, 2) void k_gp_spinor12(SpinorArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    constexpr int D = 64, LD = 65, P = D * LD;  // plane = 64 rows of 65
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;

    // entry e = tid + 256 u; two 16-bit entries per register (u = 2w, 2w+1)
    uint32_t lm[8], rm[8], om[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        lm[w] = uint32_t(p.left_map[tid + 512 * w]) | (uint32_t(p.left_map[tid + 512 * w + 256]) << 16);
        rm[w] = uint32_t(p.right_map[tid + 512 * w]) | (uint32_t(p.right_map[tid + 512 * w + 256]) << 16);
        om[w] = uint32_t(p.out_map[tid + 512 * w]) | (uint32_t(p.out_map[tid + 512 * w + 256]) << 16);
    }
    auto entry = [](const uint32_t (&m)[8], int u) -> uint32_t { return (u & 1) ? m[u >> 1] >> 16 : m[u >> 1]; };
    float va[16], vb[16];
    const bool rows_full = p.left_len == 4096 && p.right_len == 4096;
    auto fetch = [&](int64_t item) {
        const float* lrow = p.left + item * p.left_stride + tid;
        const float* rrow = p.right + item * p.right_stride + tid;
        if (rows_full) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                va[u] = lrow[256 * u];
                vb[u] = rrow[256 * u];
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int e = tid + 256 * u;
                va[u] = e < p.left_len ? lrow[256 * u] : 0.f;
                vb[u] = e < p.right_len ? rrow[256 * u] : 0.f;
            }
        }
    };
    int64_t item = blockIdx.x;
    if (item < p.batch) fetch(item);

    for (; item < p.batch; item += gridDim.x) {
        // keep the packed tables packed: without this the decoded fields of all 48 entries are hoisted
        // out of the loop and the kernel spills
#pragma unroll
        for (int w = 0; w < 8; ++w) asm volatile("" : "+v"(lm[w]), "+v"(rm[w]), "+v"(om[w]));
        // ---- 1. graded rows -> W[x][z], split into real / imaginary planes by the phase i^k ----
        if (!p.left_full || !p.right_full) {
            for (int i = tid; i < 4 * P; i += 256) smem[i] = 0.f;
            __syncthreads();
        }
        if (!(p.debug_skip & 1)) {
            // branch-free: one write of (re, im) = (a, 0) or (0, a) per component
            auto put = [&](float* planes, uint32_t e, float a, int canon) {
                if (canon) a = 0.f + a;
                a = __uint_as_float(__float_as_uint(a) ^ (e << 31));
                float* q = reinterpret_cast<float*>(reinterpret_cast<char*>(planes) + (e & 0x7ffcu));
                const bool im = (e & 2u) != 0;
                q[0] = im ? 0.f : a;
                q[P] = im ? a : 0.f;
            };
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                put(smem, entry(lm, u), va[u], p.canon_left);
                put(smem + 2 * P, entry(rm, u), vb[u], p.canon_right);
            }
        }
        __syncthreads();
        // operands of the next item: in flight during the transforms and the GEMM
        if (item + gridDim.x < p.batch) fetch(item + gridDim.x);

        // ---- 2. Walsh-Hadamard transform of every row: thread = (operand, plane, x) ----
        if (!(p.debug_skip & 2)) {
            float* row = smem + (tid >> 6) * P + (tid & 63) * LD;
            float v[64];
#pragma unroll
            for (int z = 0; z < 64; ++z) v[z] = row[z];
            wht<64>(v);
            // S_A[x][c] = M_A[c^x][c] = T_x[c].  S_B[x][r] = M_B[r][r^x] = T_x[r^x]: a shift of the
            // transform's index by x is a sign (-1)^|x&z| on its input, which the right operand's table
            // already carries, so both operands are written back in place.
#pragma unroll
            for (int c = 0; c < 64; ++c) row[c] = v[c];
        }
        __syncthreads();

        // ---- 3. complex 64 x 64 x 64 product on the matrix cores ----
        const int i = lane & 31, h = lane >> 5;
        const int r0 = (wave >> 1) << 5, c0 = (wave & 1) << 5;
        // (Ar + i Ai)(Br + i Bi) with three real products: X = Ar Br, Y = Ai Bi, Z = (Ar+Ai)(Br+Bi);
        // Re = X - Y, Im = Z - X - Y
        float16v gx, gy, gz;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            gx[r] = 0.f;
            gy[r] = 0.f;
            gz[r] = 0.f;
        }
        if (!(p.debug_skip & 4)) {
            const float* Are = smem;
            const float* Aim = smem + P;
            const float* Bre = smem + 2 * P;
            const float* Bim = smem + 3 * P;
            (void)Are; (void)Aim; (void)Bre; (void)Bim;
            const uint32_t ra = uint32_t(r0 + i), cb = uint32_t(c0 + i);
            const uint32_t lds0 = uint32_t(uintptr_t(smem));
            const uint32_t row_bytes = LD * 4;
            // M_A[ra][k] = S_A[ra ^ k][k], M_B[k][cb] = S_B[k ^ cb][k].  Hand-scheduled: the operands
            // of step s2+1 are read (re and im planes with one ds_read2st64) while the three MFMAs of
            // step s2 run; byte address = (row * 65 + k) * 4 with a 24-bit multiply-add.
            float2v av[2], bv[2];
            uint32_t k4 = lds0 + 4u * uint32_t(h), kk = uint32_t(h);
            auto issue = [&](float2v& a, float2v& b) {
                uint32_t aa, ab;
                // the running k (and 4k + base) are advanced inside the asm so that the 32 unrolled
                // values are not precomputed outside the item loop (and spilled)
                asm volatile("v_xor_b32 %0, %4, %2\n\tv_xor_b32 %1, %5, %2\n\t"
                             "v_mad_u32_u24 %0, %0, %6, %3\n\tv_mad_u32_u24 %1, %1, %6, %3\n\t"
                             "v_add_u32 %2, 2, %2\n\tv_add_u32 %3, 8, %3"
                             : "=&v"(aa), "=&v"(ab), "+v"(kk), "+v"(k4) : "v"(ra), "v"(cb), "s"(row_bytes));
                asm volatile("ds_read2st64_b32 %0, %1 offset1:65" : "=v"(a) : "v"(aa));
                asm volatile("ds_read2st64_b32 %0, %1 offset0:130 offset1:195" : "=v"(b) : "v"(ab));
            };
            issue(av[0], bv[0]);
#pragma unroll
            for (int s2 = 0; s2 < 32; ++s2) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                float2v& a = av[s2 & 1];
                float2v& b = bv[s2 & 1];
                asm volatile("" : "+v"(a), "+v"(b));   // the reads above have landed: values are live from here
                if (s2 < 31) issue(av[(s2 + 1) & 1], bv[(s2 + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                gx = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, gx, 0, 0, 0);
                gy = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, gy, 0, 0, 0);
                gz = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x + a.y, b.x + b.y, gz, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();  // every wave is done reading A and B

        // ---- 4. C back to LDS, skewed: S_C[r ^ c][r] = C[r][c] (reusing the A planes).  Indexing the
        // diagonal x = r ^ c by r (not c) keeps the 32 lanes of a store on 32 banks; the transform of a
        // row shifted by x is the wanted one times (-1)^|x&z|, a sign the result table carries. ----
        {
            const int c = c0 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = r0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                smem[(rr ^ c) * LD + rr] = gx[r] - gy[r];
                smem[P + (rr ^ c) * LD + rr] = gz[r] - gx[r] - gy[r];
            }
        }
        __syncthreads();
        // inverse transform in place: V[x][z] = 2^-6 sum_c (-1)^|c & z| S_C[x][c]
        // two threads per row (adjacent lanes): thread hb folds the halves with sign (-1)^hb, a 32-point
        // transform then gives the outputs z = j + 32 hb
        if (!(p.debug_skip & 8)) {
            const int hb = tid & 1;
            float* row = smem + (tid >> 7) * P + ((tid >> 1) & 63) * LD;
            const float sg = hb ? -1.0f / 64.0f : 1.0f / 64.0f;
            float v[32];
#pragma unroll
            for (int c = 0; c < 32; ++c) v[c] = row[c] * (1.0f / 64.0f) + row[c + 32] * sg;
            wht<32>(v);
#pragma unroll
            for (int z = 0; z < 32; ++z) row[z + 32 * hb] = v[z];   // same wave as the partner's reads: ordered
        }
        __syncthreads();
        // component of blade S(x,z): Re(V i^-k) = +re, +im, -re, -im for k = 0..3; rows written in order
        if (!(p.debug_skip & 8)) {
            float* orow = p.out + item * p.out_stride + tid;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const uint32_t eo = entry(om, u);
                float val = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(smem) + (eo & 0xfffcu));
                val = __uint_as_float(__float_as_uint(val) ^ (eo << 31));
                if (p.out_full && !p.beta) {
                    orow[256 * u] = val;
                } else if (!(eo & 2u)) {
                    orow[256 * u] = p.beta ? orow[256 * u] + val : val;
                }
            }
        }
        __syncthreads();  // the planes are free for the next item
    }
}

// Same algorithm for the smaller even dimensions: n = 8 (16 x 16 matrices, v_mfma_f32_16x16x4_f32)
// and n = 10 (32 x 32, v_mfma_f32_32x32x2_f32); odd n runs as the subalgebra of n + 1.  One WAVE per
// item (64-thread workgroups, persistent), so the phases need no cross-wave barrier, and the four
// planes of an item are 5 KB / 17 KB of LDS.  Plane stride = a multiple of 64 words (re/im pairs by
// ds_read2st64).  Table formats as in SpinorArgs with 65 -> D + 1 and 4160 -> the plane stride.
template <int M>
__global__ __launch_bounds__(64) void k_gp_spinor_wave(SpinorArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    constexpr int D = 1 << M, LD = D + 1, P = D * LD;
    constexpr int PS = (P + 63) / 64 * 64;
    constexpr int NE = D * D, EPL = NE / 64, RPL = 4 * D / 64;
    using acc_t = typename std::conditional<M == 4, float4v, float16v>::type;
    constexpr int NACC = M == 4 ? 4 : 16;
    const int lane = threadIdx.x;

    uint32_t lm[EPL / 2], rm[EPL / 2], om[EPL / 2];
#pragma unroll
    for (int w = 0; w < EPL / 2; ++w) {
        lm[w] = uint32_t(p.left_map[lane + 128 * w]) | (uint32_t(p.left_map[lane + 128 * w + 64]) << 16);
        rm[w] = uint32_t(p.right_map[lane + 128 * w]) | (uint32_t(p.right_map[lane + 128 * w + 64]) << 16);
        om[w] = uint32_t(p.out_map[lane + 128 * w]) | (uint32_t(p.out_map[lane + 128 * w + 64]) << 16);
    }
    auto entry = [](const uint32_t (&m)[EPL / 2], int u) -> uint32_t { return (u & 1) ? m[u >> 1] >> 16 : m[u >> 1]; };
    float va[EPL], vb[EPL];
    const bool rows_full = p.left_len == NE && p.right_len == NE;
    auto fetch = [&](int64_t item) {
        const float* lrow = p.left + item * p.left_stride + lane;
        const float* rrow = p.right + item * p.right_stride + lane;
        if (rows_full) {
#pragma unroll
            for (int u = 0; u < EPL; ++u) {
                va[u] = lrow[64 * u];
                vb[u] = rrow[64 * u];
            }
        } else {
#pragma unroll
            for (int u = 0; u < EPL; ++u) {
                const int e = lane + 64 * u;
                va[u] = e < p.left_len ? lrow[64 * u] : 0.f;
                vb[u] = e < p.right_len ? rrow[64 * u] : 0.f;
            }
        }
    };
    int64_t item = blockIdx.x;
    if (item < p.batch) fetch(item);

    for (; item < p.batch; item += gridDim.x) {
#pragma unroll
        for (int w = 0; w < EPL / 2; ++w) asm volatile("" : "+v"(lm[w]), "+v"(rm[w]), "+v"(om[w]));
        if (!p.left_full || !p.right_full) {
            for (int i = lane; i < 4 * PS; i += 64) smem[i] = 0.f;
            __syncthreads();
        }
        {
            auto put = [&](float* planes, uint32_t e, float a, int canon) {
                if (canon) a = 0.f + a;
                a = __uint_as_float(__float_as_uint(a) ^ (e << 31));
                float* q = reinterpret_cast<float*>(reinterpret_cast<char*>(planes) + (e & 0x7ffcu));
                const bool im = (e & 2u) != 0;
                q[0] = im ? 0.f : a;
                q[PS] = im ? a : 0.f;
            };
#pragma unroll
            for (int u = 0; u < EPL; ++u) {
                put(smem, entry(lm, u), va[u], p.canon_left);
                put(smem + 2 * PS, entry(rm, u), vb[u], p.canon_right);
            }
        }
        __syncthreads();
        if (item + gridDim.x < p.batch) fetch(item + gridDim.x);

        // row transforms: (operand, plane, x) = 4 D rows over 64 lanes
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
            const int ridx = lane + 64 * j;
            float* row = smem + (ridx >> M) * PS + (ridx & (D - 1)) * LD;
            float v[D];
#pragma unroll
            for (int z = 0; z < D; ++z) v[z] = row[z];
            wht<D>(v);
#pragma unroll
            for (int c = 0; c < D; ++c) row[c] = v[c];
        }
        __syncthreads();

        // complex D x D x D product, three real ones (see k_gp_spinor12)
        acc_t gx, gy, gz;
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
            gx[r] = 0.f;
            gy[r] = 0.f;
            gz[r] = 0.f;
        }
        const int i = lane & (D - 1);
        const int kq = lane >> M;                 // M = 4: 0..3 (k = 4 s + kq);  M = 5: 0..1 (k = 2 s + kq)
        constexpr int KSTEP = 64 / D;             // k values per MFMA
#pragma unroll
        for (int s2 = 0; s2 < D / KSTEP; ++s2) {
            const int k = KSTEP * s2 + kq;
            const int idx = (i ^ k) * LD + k;     // M_A[i][k] = S_A[i^k][k], M_B[k][i] = S_B[k^i][k]
            const float are = smem[idx], aim = smem[PS + idx], bre = smem[2 * PS + idx], bim = smem[3 * PS + idx];
            if constexpr (M == 4) {
                gx = __builtin_amdgcn_mfma_f32_16x16x4f32(are, bre, gx, 0, 0, 0);
                gy = __builtin_amdgcn_mfma_f32_16x16x4f32(aim, bim, gy, 0, 0, 0);
                gz = __builtin_amdgcn_mfma_f32_16x16x4f32(are + aim, bre + bim, gz, 0, 0, 0);
            } else {
                gx = __builtin_amdgcn_mfma_f32_32x32x2f32(are, bre, gx, 0, 0, 0);
                gy = __builtin_amdgcn_mfma_f32_32x32x2f32(aim, bim, gy, 0, 0, 0);
                gz = __builtin_amdgcn_mfma_f32_32x32x2f32(are + aim, bre + bim, gz, 0, 0, 0);
            }
        }
        __syncthreads();
        // C back, diagonals indexed by row: S_C[r ^ c][r] = C[r][c]
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
            const int rr = M == 4 ? 4 * kq + r : (r & 3) + 8 * (r >> 2) + 4 * kq;
            smem[(rr ^ i) * LD + rr] = gx[r] - gy[r];
            smem[PS + (rr ^ i) * LD + rr] = gz[r] - gx[r] - gy[r];
        }
        __syncthreads();
        // inverse transform: (plane, x, half) = 4 D half-rows over 64 lanes
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
            const int hidx = lane + 64 * j;
            const int hb = hidx & 1;
            float* row = smem + (hidx >> (M + 1)) * PS + ((hidx >> 1) & (D - 1)) * LD;
            const float sc = 1.0f / float(D), sg = hb ? -sc : sc;
            float v[D / 2];
#pragma unroll
            for (int c = 0; c < D / 2; ++c) v[c] = row[c] * sc + row[c + D / 2] * sg;
            wht<D / 2>(v);
#pragma unroll
            for (int z = 0; z < D / 2; ++z) row[z + (D / 2) * hb] = v[z];
        }
        __syncthreads();
        {
            float* orow = p.out + item * p.out_stride + lane;
#pragma unroll
            for (int u = 0; u < EPL; ++u) {
                const uint32_t eo = entry(om, u);
                float val = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(smem) + (eo & 0xfffcu));
                val = __uint_as_float(__float_as_uint(val) ^ (eo << 31));
                if (p.out_full && !p.beta) {
                    orow[64 * u] = val;
                } else if (!(eo & 2u)) {
                    orow[64 * u] = p.beta ? orow[64 * u] + val : val;
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace gaast
