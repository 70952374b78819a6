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
    const uint32_t* left_map;   // per loaded component: row offset | bitmask << 16
    const uint32_t* right_map;
    int left_count, right_count;
    int left_full, right_full;  // 1: every blade is loaded, no zero fill needed
    const int32_t* out_map;     // per bitmask: offset in the out row, or -1
    int canon_left, canon_right;
    int n;                      // vector-space dimension, 4 <= n
    uint32_t neg_hi, zero_hi;   // metric signature of basis vectors 4.. (bit i <-> vector 4+i)
    int beta;
    int64_t batch;
};

// physical position of blade bitmask m inside an operand's LDS image: blocks of 16, the four
// 16-byte quads of block x rotated by (x >> 2) & 3 so that 16 lanes reading the same logical
// quad of 16 different blocks touch 16 different bank quads.
__device__ __forceinline__ int dense_lds_pos(int m) {
    const int x = m >> 4, lo = m & 15;
    return (x << 4) | ((((lo >> 2) ^ (x >> 2)) & 3) << 2) | (lo & 3);
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

template <typename T>
struct Vec4 {
    T x, y, z, w;
};

template <typename T, bool DEGENERATE>
__global__ __launch_bounds__(512) void k_gp_dense(DenseArgs<T> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);
    const int n = p.n;
    const int N = 1 << n;
    const int hbits = n - 4;
    const int LPI = 1 << hbits;                   // lanes per item
    const int IPB = blockDim.x >> hbits;          // items per block (>= 1)
    const int item_stride = 2 * N + 4;            // +16 B: de-phase the items' A broadcasts
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int64_t item0 = int64_t(blockIdx.x) * IPB;
    const int nitems = int(p.batch - item0 < IPB ? p.batch - item0 : IPB);
    const T zero = T(0);

    // ---- stage both operands of every item of the block, in bitmask order ----
    if (!p.left_full || !p.right_full) {
        for (int i = tid; i < nitems * item_stride; i += nthr) smem[i] = zero;
        __syncthreads();
    }
    for (int i = tid; i < nitems * p.left_count; i += nthr) {
        const int it = i / p.left_count, j = i - it * p.left_count;
        const uint32_t m = p.left_map[j];
        T v = p.left[(item0 + it) * p.left_stride + (m & 0xffffu)];
        if (p.canon_left) v = zero + v;
        smem[it * item_stride + dense_lds_pos(int(m >> 16))] = v;
    }
    for (int i = tid; i < nitems * p.right_count; i += nthr) {
        const int it = i / p.right_count, j = i - it * p.right_count;
        const uint32_t m = p.right_map[j];
        T v = p.right[(item0 + it) * p.right_stride + (m & 0xffffu)];
        if (p.canon_right) v = zero + v;
        smem[it * item_stride + N + dense_lds_pos(int(m >> 16))] = v;
    }
    __syncthreads();

    const int it = tid >> hbits;
    const int c_hi = tid & (LPI - 1);
    if (it < nitems) {
        const T* As = smem + it * item_stride;
        const T* Bs = As + N;
        T C[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) C[i] = zero;

        for (int a_hi = 0; a_hi < LPI; ++a_hi) {
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
            const int b_hi = a_hi ^ c_hi;
            const int sa = (a_hi >> 2) & 3, sb = (b_hi >> 2) & 3;
            const Vec4<T>* ap = reinterpret_cast<const Vec4<T>*>(As + (a_hi << 4));
            const Vec4<T>* bp = reinterpret_cast<const Vec4<T>*>(Bs + (b_hi << 4));
            T A[16], B[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const Vec4<T> va = ap[q ^ sa];
                const Vec4<T> vb = bp[q ^ sb];
                A[4 * q + 0] = va.x; A[4 * q + 1] = va.y; A[4 * q + 2] = va.z; A[4 * q + 3] = va.w;
                B[4 * q + 0] = vb.x * sgn; B[4 * q + 1] = vb.y * sgn;
                B[4 * q + 2] = vb.z * sgn; B[4 * q + 3] = vb.w * sgn;
            }
            if (__builtin_popcount(uint32_t(a_hi)) & 1)
                gp_block16<T, true>(A, B, C);
            else
                gp_block16<T, false>(A, B, C);
        }

        // ---- scatter the 16 accumulators to their positions in the graded row ----
        T* orow = p.out + (item0 + it) * p.out_stride;
        const int32_t* om = p.out_map + (c_hi << 4);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int32_t off = om[i];
            if (off >= 0) orow[off] = p.beta ? orow[off] + C[i] : C[i];
        }
    }
}

}  // namespace gaast
