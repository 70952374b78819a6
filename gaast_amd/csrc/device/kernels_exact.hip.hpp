// kernels_exact.hip.hpp -- elementwise arms (GradedObj copy, sign flips, scalar unary) and the exact CSR product
// Included through kernels.hip.hpp.
#pragma once
#include "kernels_common.hip.hpp"

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
// EXTENSION (GAAST_FLAG_EXP_LOG; eval.rs:112-113 is todo!() upstream): res += exp(B) / log(a + B) for a k-vector B whose
// square is scalar -- the statements of oracle/gaast_oracle.c: ext_exp_log in its order, one thread per item.  Items whose
// B B has a non-negligible non-scalar part are counted in *dom (gaast_hip_program_domain_errors).
// ------------------------------------------------------------------------------------------
template <typename T>
struct ExpLogArgs {
    T* res;
    const T* arg;
    int64_t res_stride, arg_stride;
    int op;                       // 0 exp, 1 log
    int m, m_res;
    int arg_k, arg_0, res_k, res_0;   // row offsets, -1 = absent
    const T* sq;                  // e_i e_i
    const uint32_t* row_start;    // domain-check rows (n_rows + 1)
    const uint32_t* pairs;        // i | j << 16
    const T* pair_coeff;          // 2 e_i e_j
    int n_rows;
    unsigned long long* dom;
    int64_t batch;
};

__device__ __forceinline__ float sqrt_m(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ double sqrt_m(double x) { return __builtin_sqrt(x); }
__device__ __forceinline__ float sin_m(float x) { return sinf(x); }
__device__ __forceinline__ double sin_m(double x) { return sin(x); }
__device__ __forceinline__ float cos_m(float x) { return cosf(x); }
__device__ __forceinline__ double cos_m(double x) { return cos(x); }
__device__ __forceinline__ float sinh_m(float x) { return sinhf(x); }
__device__ __forceinline__ double sinh_m(double x) { return sinh(x); }
__device__ __forceinline__ float cosh_m(float x) { return coshf(x); }
__device__ __forceinline__ double cosh_m(double x) { return cosh(x); }
__device__ __forceinline__ float atan2_m(float y, float x) { return atan2f(y, x); }
__device__ __forceinline__ double atan2_m(double y, double x) { return atan2(y, x); }
__device__ __forceinline__ float atanh_m(float x) { return atanhf(x); }
__device__ __forceinline__ double atanh_m(double x) { return atanh(x); }

template <typename T>
__global__ __launch_bounds__(256) void k_exp_log(ExpLogArgs<T> p) {
    for (int64_t item = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; item < p.batch; item += int64_t(gridDim.x) * blockDim.x) {
        const T* B = p.arg + item * p.arg_stride + p.arg_k;
        T sq = T(0), nrm = T(0), viol = T(0);
        for (int i = 0; i < p.m; ++i) {
            sq = sq + B[i] * B[i] * p.sq[i];
            nrm = nrm + B[i] * B[i];
        }
        for (int row = 0; row < p.n_rows; ++row) {
            T acc = T(0);
            for (uint32_t e = p.row_start[row]; e < p.row_start[row + 1]; ++e)
                acc = acc + B[p.pairs[e] & 0xffffu] * B[p.pairs[e] >> 16] * p.pair_coeff[e];
            viol = viol + acc * acc;
        }
        if (p.n_rows > 0 && viol > T(9.094947017729282e-13) * (nrm * nrm)) atomicAdd(p.dom, 1ull);
        T c0 = T(0), f;
        if (p.op == 0) {
            if (sq < T(0)) { const T t = sqrt_m(-sq); c0 = cos_m(t); f = sin_m(t) / t; }
            else if (sq > T(0)) { const T t = sqrt_m(sq); c0 = cosh_m(t); f = sinh_m(t) / t; }
            else if (sq == T(0)) { c0 = T(1); f = T(1); }
            else { c0 = sq; f = sq; }
        } else {
            const T a = p.arg_0 >= 0 ? p.arg[item * p.arg_stride + p.arg_0] : T(0);
            if (sq < T(0)) { const T mm = sqrt_m(-sq); f = atan2_m(mm, a) / mm; }
            else if (sq > T(0)) { const T mm = sqrt_m(sq); f = atanh_m(mm / a) / mm; }
            else if (sq == T(0)) { f = T(1) / a; }
            else { f = sq; }
        }
        T* r = p.res + item * p.res_stride;
        if (p.res_0 >= 0) r[p.res_0] = r[p.res_0] + c0;
        if (p.res_k >= 0)
            for (int i = 0; i < p.m_res; ++i) r[p.res_k + i] = r[p.res_k + i] + f * B[i];
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
// k_product_ell: the exact product for lists whose rows (result components) all have the same
// number of entries and whose coefficients are +-1 -- every dense product of a non-degenerate
// algebra.  Same terms in the same (reference) order with the same three roundings per term as
// k_product_csr, but the list is stored [term][row] with the sign in bit 31: a wave's 64 rows read 64
// consecutive words per term instead of 64 separate streams, no coefficient array is read, and one
// pass over the list serves ITEMS batch items (their operand rows stay in LDS).
//     entry = left offset [15:0] | right offset [30:16] | negate [31]   (element or byte offsets)
// ------------------------------------------------------------------------------------------
template <typename T>
struct EllArgs {
    const T* left;
    const T* right;
    T* out;
    int64_t left_stride, right_stride, out_stride;
    int left_len, right_len;
    int canon_left, canon_right;
    const uint32_t* row_out;   // output offset of each row
    const uint32_t* entries;   // [width][n_rows]
    int n_rows, width;
    int beta;
    int64_t batch;
};

__device__ __forceinline__ float ell_flip(float v, uint32_t sign31) { return __uint_as_float(__float_as_uint(v) ^ sign31); }
__device__ __forceinline__ double ell_flip(double v, uint32_t sign31) {
    return __hiloint2double(__double2hiint(v) ^ int(sign31), __double2loint(v));
}

// BYTES: the two offsets of an entry are byte offsets (rows of at most 32 KiB), added to the LDS address of an
// item's row as they are; otherwise element offsets.
template <typename T, int ITEMS, bool BYTES>
__global__ __launch_bounds__(256) void k_product_ell(EllArgs<T> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* ls = reinterpret_cast<T*>(smem_raw);
    T* rs = ls + int64_t(ITEMS) * p.left_len;
    const int64_t item0 = int64_t(blockIdx.x) * ITEMS;
    const int nitems = int(p.batch - item0 < ITEMS ? p.batch - item0 : ITEMS);
    const int tid = threadIdx.x;
    const T zero = T(0);
    // operand rows of ITEMS items; rows beyond the batch are zero (their results are not stored)
    for (int i = tid; i < ITEMS * p.left_len; i += 256) {
        const int it = i / p.left_len, c = i - it * p.left_len;
        T v = it < nitems ? p.left[(item0 + it) * p.left_stride + c] : zero;
        ls[i] = p.canon_left ? zero + v : v;
    }
    for (int i = tid; i < ITEMS * p.right_len; i += 256) {
        const int it = i / p.right_len, c = i - it * p.right_len;
        T v = it < nitems ? p.right[(item0 + it) * p.right_stride + c] : zero;
        rs[i] = p.canon_right ? zero + v : v;
    }
    __syncthreads();
    const char* lsb = reinterpret_cast<const char*>(ls);
    const char* rsb = reinterpret_cast<const char*>(rs);
    const int lbytes = p.left_len * int(sizeof(T)), rbytes = p.right_len * int(sizeof(T));

    for (int row = tid; row < p.n_rows; row += 256) {
        const uint32_t oo = p.row_out[row];
        T acc[ITEMS];
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) acc[it] = (p.beta && it < nitems) ? p.out[(item0 + it) * p.out_stride + oo] : zero;
        const uint32_t* ep = p.entries + row;
        auto term = [&](uint32_t e) {
            const uint32_t sign = e & 0x80000000u;
            const uint32_t lo = BYTES ? (e & 0x7fffu) : (e & 0xffffu) * uint32_t(sizeof(T));
            const uint32_t ro = BYTES ? ((e >> 16) & 0x7fffu) : ((e >> 16) & 0x7fffu) * uint32_t(sizeof(T));
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const T l = *reinterpret_cast<const T*>(lsb + it * lbytes + lo);
                const T r = *reinterpret_cast<const T*>(rsb + it * rbytes + ro);
                const T prod = ell_flip(l * r, sign);      // eval.rs:82: (left * right) * coeff, coeff = +-1
                acc[it] = acc[it] + prod;                   // ... +=
            }
        };
        // 32 list words are loaded before they are used, to have that many loads in flight (the list comes
        // from L2 / MALL: 64 MiB at n = 12); widths of dense products are multiples of 32 from n = 5 on
        int t = 0;
        for (; t + 32 <= p.width; t += 32) {
            uint32_t ev[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) ev[j] = ep[size_t(t + j) * p.n_rows];
#pragma unroll
            for (int j = 0; j < 32; ++j) term(ev[j]);
        }
        for (; t < p.width; ++t) term(ep[size_t(t) * p.n_rows]);
#pragma unroll
        for (int it = 0; it < ITEMS; ++it)
            if (it < nitems) p.out[(item0 + it) * p.out_stride + oo] = acc[it];
    }
}

}  // namespace gaast
