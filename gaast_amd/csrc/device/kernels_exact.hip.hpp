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
                                                  int64_t batch, int beta) {
    const int64_t total = batch * n_map;
    for (int64_t idx = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
         idx += int64_t(gridDim.x) * blockDim.x) {
        const int64_t item = idx / n_map;
        const int j = int(idx - item * n_map);
        const uint32_t m = map[j];
        T* r = res + item * res_stride + (m & 0xffffu);
        // graded.rs:74  `*r = *r + i`; beta = 0: res is the fresh (all +0.0) buffer of init_null_mv and this arm writes every
        // component of it -- the zero fill is folded in: 0.0 + i
        *r = (beta ? *r : T(0)) + in[item * in_stride + (m >> 16)];
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
// k_reduce_scale: out = x (*) f(<l, r>) -- a product whose result is ONE scalar component (a single row of n1 terms: norm_sq =
// (a.rev() * a).g(0)), an optional ScalarUnaryOp on it (eval.rs:103-110), and a product of n2 one-term rows that multiplies another
// row by that scalar: the versor inverse a.rev() * a.norm_sq().sinv() (expr.rs:363-371) and normalisations, at the dimensions where
// the rows no longer fit a fused slab (n >= 9: 2 x 16 KiB per item at n = 12).  As three launches the single long row occupied one
// thread per workgroup and the row travelled twice.  Here: ONE launch, sixteen items per wave, persistent:
//   * the reduction keeps the reference's order and roundings (eval.rs:82: (l * r) * coeff, then +=, term after term): a QUAD of
//     lanes per item fetches and multiplies four consecutive terms, the chain of additions takes them in lane order -- the loads
//     of the next round in flight under it;
//   * 1 / s or sqrt(s), correctly rounded, then 0.0 + s when the scalar is re-read as a product operand (eval.rs:27-31);
//   * the scaling re-reads its row (from L2 / the Infinity Cache: this wave has just streamed it) and stores 0.0 + (x * s) * coeff.
// Bit-identical to the three-launch plan and to the oracle.
// ------------------------------------------------------------------------------------------
template <typename T>
struct ReduceScaleArgs {
    const T* l1;
    const T* r1;
    const T* x;
    T* out;
    int64_t l1_stride, r1_stride, x_stride, out_stride;
    const uint32_t* ent1;   // n1 terms: left offset | right offset << 16 (elements)
    const T* coeff1;        // n1 coefficients
    const uint32_t* ent2;   // n2 rows: x offset | out offset << 16
    const T* coeff2;        // n2 coefficients
    int n1, n2;
    int canon_l1, canon_r1, canon_x, canon_s;
    int s_is_left;          // the scalar is the LEFT operand of the scaling product (values are the same either way)
    int op;                 // 0 none, 1 inversion, 2 square root
    int64_t batch;
};

__device__ __forceinline__ float lane_value(float v, int j) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j)); }
__device__ __forceinline__ double lane_value(double v, int j) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), j), __builtin_amdgcn_readlane(__double2loint(v), j));
}
// the value lane J of this lane's QUAD holds (DPP quad_perm broadcast: no LDS, no scalar round trip)
template <int J>
__device__ __forceinline__ float quad_value(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), J * 0x55, 0xf, 0xf, true));
}
template <int J>
__device__ __forceinline__ double quad_value(double v) {
    return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), J * 0x55, 0xf, 0xf, true),
                            __builtin_amdgcn_mov_dpp(__double2loint(v), J * 0x55, 0xf, 0xf, true));
}

// A QUAD per item in the reduction (16 items per wave): the four lanes of a quad fetch and multiply four consecutive terms of
// their item, and the chain of additions takes them in lane order through DPP quad broadcasts -- three vector instructions per
// four-lane step serve SIXTEEN items (one wave per item spent them on one: 6,000 cycles per item and CU at n = 12; wave-wide
// LDS re-reads of parked products: 8,000).  The scaling then walks the wave's sixteen items with all 64 lanes (coalesced rows).
template <typename T>
__global__ __launch_bounds__(256) void k_reduce_scale(ReduceScaleArgs<T> p) {
    constexpr int CHUNKS = 8;                                  // steps of 4 terms per item and round (their loads in flight one round ahead)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = lane >> 2, ql = lane & 3;                    // this lane's item of the wave's sixteen, its place in the quad
    const int64_t n_waves = int64_t(gridDim.x) * 4;
    const T zero = T(0);
    const int n_steps = (p.n1 + 3) / 4, n_rounds = (n_steps + CHUNKS - 1) / CHUNKS;
    const int64_t n_groups = (p.batch + 15) / 16;
    for (int64_t g = int64_t(blockIdx.x) * 4 + wv; g < n_groups; g += n_waves) {
        const int64_t item = g * 16 + q;
        const bool live_item = item < p.batch;
        const int64_t it = live_item ? item : p.batch - 1;     // (lanes beyond the batch re-read the last item; nothing of theirs is stored)
        const T* l = p.l1 + it * p.l1_stride;
        const T* r = p.r1 + it * p.r1_stride;
        // software pipeline over rounds of CHUNKS steps: a term's operand addresses come from its table word, so the words are
        // fetched two rounds ahead and the operands one round ahead
        uint32_t e_cur[CHUNKS], e_nxt[CHUNKS];
        T lv[CHUNKS], rv[CHUNKS], cf[CHUNKS];
        auto fetch_words = [&](int round, uint32_t (&e)[CHUNKS]) {
#pragma unroll
            for (int k = 0; k < CHUNKS; ++k) {
                const int t = (round * CHUNKS + k) * 4 + ql;
                e[k] = t < p.n1 ? p.ent1[t] : 0u;
            }
        };
        auto fetch_operands = [&](int round, const uint32_t (&e)[CHUNKS]) {
#pragma unroll
            for (int k = 0; k < CHUNKS; ++k) {
                const int t = (round * CHUNKS + k) * 4 + ql;
                const bool live = t < p.n1;
                lv[k] = live ? l[e[k] & 0xffffu] : zero;
                rv[k] = live ? r[e[k] >> 16] : zero;
                cf[k] = live ? p.coeff1[t] : zero;
            }
        };
        fetch_words(0, e_cur);
        fetch_words(1, e_nxt);
        fetch_operands(0, e_cur);
        T acc = zero;                                         // the fresh cache buffer of eval.rs:21-33
        for (int round = 0; round < n_rounds; ++round) {
            T prod[CHUNKS];
#pragma unroll
            for (int k = 0; k < CHUNKS; ++k) {                // this round's products, rounded as eval.rs:82
                T a = lv[k], b = rv[k];
                if (p.canon_l1) a = zero + a;
                if (p.canon_r1) b = zero + b;
                prod[k] = (a * b) * cf[k];
            }
#pragma unroll
            for (int k = 0; k < CHUNKS; ++k) e_cur[k] = e_nxt[k];
            if (round + 1 < n_rounds) fetch_operands(round + 1, e_cur);   // in flight under the chain of additions
            if (round + 2 < n_rounds) fetch_words(round + 2, e_nxt);
#pragma unroll
            for (int k = 0; k < CHUNKS; ++k) {                // the chain, term after term (no padding terms: acc + 0.0 is not a no-op for -0.0)
                const int left = p.n1 - (round * CHUNKS + k) * 4;
                if (left > 0) acc = acc + quad_value<0>(prod[k]);
                if (left > 1) acc = acc + quad_value<1>(prod[k]);
                if (left > 2) acc = acc + quad_value<2>(prod[k]);
                if (left > 3) acc = acc + quad_value<3>(prod[k]);
            }
        }
        T s = acc;
        if (p.op == 1) s = T(1) / s;
        else if (p.op == 2) s = sizeof(T) == 8 ? T(__builtin_sqrt(double(s))) : T(__builtin_sqrtf(float(s)));
        if (p.canon_s) s = zero + s;
        // ---- the scaling: all 64 lanes on 64 consecutive rows; a row's table word and coefficient are fetched ONCE and serve the
        // wave's sixteen items, whose operand loads are all in flight together (item after item, each iteration waiting for its
        // word and then for its operand, this phase alone took 6,000 cycles per item and CU) ----
        T sk[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) sk[k] = lane_value(s, 4 * k);
        const int64_t item0 = g * 16;
        const int n_items = int(p.batch - item0 < 16 ? p.batch - item0 : 16);
        for (int i0 = 0; i0 < p.n2; i0 += 64) {
            const int i = i0 + lane;
            const bool live = i < p.n2;
            const uint32_t e = live ? p.ent2[i] : 0u;
            const T c2 = live ? p.coeff2[i] : zero;
            T xv[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) xv[k] = (live && k < n_items) ? p.x[(item0 + k) * p.x_stride + (e & 0xffffu)] : zero;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                T v = xv[k];
                if (p.canon_x) v = zero + v;
                const T pr = p.s_is_left ? sk[k] * v : v * sk[k];
                if (live && k < n_items) p.out[(item0 + k) * p.out_stride + (e >> 16)] = zero + pr * c2;   // the fresh result buffer: 0.0 + (l * r) * coeff
            }
        }
    }
}

// TOLERANCE MODE of the same step (no GAAST_FLAG_EXACT_ORDER; the dense products' contract, 4 eps sum |terms|) when the reduction is
// a row's signed sum of squares and the scaled row is that same row -- the versor inverse a.rev() * a.norm_sq().sinv() and its
// relatives: ONE WAVE PER ITEM, the row read ONCE (k_reduce_scale reads it for the reduction and again for the scaling: 1.5 x the
// algorithmic traffic at n = 12).  A lane keeps PIECES 16-byte pieces of the row in registers, sums its own terms with fused
// multiply-adds, the 64 partial sums meet through DPP butterflies (quad permutations, half-row and row mirrors) and four lane
// reads -- the north-star's "wavefront shuffles for the partial reductions" --, then every lane scales and stores its pieces.
// signs: [0][lane] bit k = coefficient -1 of the reduction's term on the lane's k-th component, [1][lane] = of the scaling's row.
template <typename T>
__device__ __forceinline__ T flip_bit(T v, uint32_t bit31);
template <>
__device__ __forceinline__ float flip_bit<float>(float v, uint32_t bit31) { return __uint_as_float(__float_as_uint(v) ^ bit31); }
template <>
__device__ __forceinline__ double flip_bit<double>(double v, uint32_t bit31) {
    return __hiloint2double(__double2hiint(v) ^ int(bit31), __double2loint(v));
}
template <int CTRL>
__device__ __forceinline__ float dpp_value(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true)); }
template <int CTRL>
__device__ __forceinline__ double dpp_value(double v) {
    return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true), __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true));
}
template <typename T, int PIECES>
__global__ __launch_bounds__(256) void k_reduce_scale_wave(ReduceScaleArgs<T> p, const uint32_t* __restrict__ signs) {
    constexpr int EPC = 16 / int(sizeof(T));                   // components per 16-byte piece
    static_assert(PIECES * EPC <= 32, "a lane's sign bits fit one word");
    typedef T VT __attribute__((ext_vector_type(EPC)));
    const int lane = threadIdx.x & 63;
    const int64_t wave = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6), n_waves = int64_t(gridDim.x) * 4;
    const uint32_t sg1 = signs[lane], sg2 = signs[64 + lane];
    const T zero = T(0);
    for (int64_t item = wave; item < p.batch; item += n_waves) {
        const VT* row = reinterpret_cast<const VT*>(p.l1 + item * p.l1_stride);
        VT v[PIECES];
#pragma unroll
        for (int m = 0; m < PIECES; ++m) v[m] = __builtin_nontemporal_load(row + m * 64 + lane);
        T acc = zero;
#pragma unroll
        for (int m = 0; m < PIECES; ++m) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                T a = v[m][e];
                if (p.canon_l1 | p.canon_r1 | p.canon_x) a = zero + a;   // (uniform) 0.0 + x of eval.rs:27-31
                v[m][e] = a;
                acc = sizeof(T) == 8 ? T(__builtin_fma(double(a), double(flip_bit<T>(a, (sg1 >> (m * EPC + e)) << 31)), double(acc)))
                                     : T(__builtin_fmaf(float(a), float(flip_bit<T>(a, (sg1 >> (m * EPC + e)) << 31)), float(acc)));
            }
        }
        // 64 partial sums -> every lane: xor 1, xor 2 inside the quads, mirror inside half rows and rows, then the four rows
        acc = acc + dpp_value<0xB1>(acc);     // quad_perm [1,0,3,2]
        acc = acc + dpp_value<0x4E>(acc);     // quad_perm [2,3,0,1]
        acc = acc + dpp_value<0x141>(acc);    // row_half_mirror
        acc = acc + dpp_value<0x140>(acc);    // row_mirror
        T s = ((lane_value(acc, 0) + lane_value(acc, 16)) + lane_value(acc, 32)) + lane_value(acc, 48);
        if (p.op == 1) s = T(1) / s;
        else if (p.op == 2) s = sizeof(T) == 8 ? T(__builtin_sqrt(double(s))) : T(__builtin_sqrtf(float(s)));
        if (p.canon_s) s = zero + s;
        VT* orow = reinterpret_cast<VT*>(p.out + item * p.out_stride);
#pragma unroll
        for (int m = 0; m < PIECES; ++m) {
            VT o;
#pragma unroll
            for (int e = 0; e < EPC; ++e) o[e] = zero + flip_bit<T>(v[m][e] * s, (sg2 >> (m * EPC + e)) << 31);   // 0.0 + (l * r) * (+-1)
            __builtin_nontemporal_store(o, orow + m * 64 + lane);
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_elementwise: a RUN of element-wise arms on one buffer -- GradedObj copies / additions (eval.rs:45-50 -> graded.rs:67-78), the
// sign arms Negation / Reverse / GradeInvolution (eval.rs:55-60, 87-102 -> graded.rs:61-65) -- and, optionally, the product of
// one-term rows that scales the result by a scalar operand (eval.rs:61-86) as ONE pass: every component executes its own
// statements in the program's order (they never look at another component), so the result is bit for bit what one launch per arm
// produces -- k_axpy_map, k_flip, k_flip, ... each a full read-modify-write of the buffer (an expression like
// (-(a.rev()) + b.ginvol()).rev() * s at n = 12: six passes over 268 MB instead of one).
//   ops: [n_ops][n_comp] words, statement k of component c:  [1:0] 0 nothing, 1 v = v + src, 2 v = -v, 3 v = 0.0 + src;
//        [4:2] source slot;  [31:16] source offset.   Epilogue (scale): out[out_off[c]] = 0.0 + ((0.0 + v) * s) * coeff[c].
// ------------------------------------------------------------------------------------------
constexpr int ELEMENTWISE_MAX_SRC = 6;
template <typename T>
struct ElementwiseArgs {
    T* res;
    int64_t res_stride;
    const T* src[ELEMENTWISE_MAX_SRC];
    int64_t src_stride[ELEMENTWISE_MAX_SRC];
    const uint32_t* ops;        // [n_ops][n_comp]
    const uint32_t* comp_off;   // offset of component c in the res row (the run's own buffer; unused when it is folded away)
    int n_ops, n_comp;
    int load_first;             // 1: v starts from the buffer's current value (the run does not begin with a covering copy)
    // epilogue: multiply by a scalar operand and store somewhere else (the run's buffer is then never written)
    T* out;
    int64_t out_stride;
    const uint32_t* out_off;    // per component, or NULL: no epilogue
    const T* coeff;
    const T* scalar;
    int64_t scalar_stride;
    int scalar_off, canon_v, canon_s, s_is_left;
    int64_t batch;
};

constexpr int ELEMENTWISE_MAX_OPS = 8;      // statements per component of one pass (longer runs are cut by the plan builder)
// A thread keeps ONE component and walks the items (blockIdx.y strides over them), FOUR items per step: its statement words,
// offsets and coefficient are decoded once, and per step the operand loads of four items are all in flight before the first
// statement runs (per (item, component) threads re-read the tables for every element: 0.27 of the HBM roof; one item per
// step with its loads behind the previous item's store: 0.20).  Consecutive threads hold consecutive components: coalesced rows.
// NOPS: statements per component this instantiation holds (4: half the registers of 8 -- the loaded operands of U items x NOPS
// statements are live at once --, five waves per SIMD instead of three: the kernel is bound by the loads it keeps in flight)
template <typename T, int NOPS = ELEMENTWISE_MAX_OPS>
__global__ __launch_bounds__(256) void k_elementwise(ElementwiseArgs<T> p) {
    constexpr int U = 4;   // (eight items per step on the four-statement instantiation: -8 %)
    const int c = int(blockIdx.x) * 256 + int(threadIdx.x);
    if (c >= p.n_comp) return;
    const T zero = T(0);
    uint32_t op[NOPS];
    const T* ld_ptr[NOPS];
    int64_t ld_stride[NOPS];
#pragma unroll
    for (int k = 0; k < NOPS; ++k) {
        const uint32_t w = k < p.n_ops ? p.ops[size_t(k) * p.n_comp + c] : 0u;
        op[k] = w & 3u;
        const int sl = int((w >> 2) & 7u);
        ld_ptr[k] = p.src[sl] + (w >> 16);      // (only dereferenced when the statement reads a source)
        ld_stride[k] = p.src_stride[sl];
    }
    const uint32_t own = p.out_off ? 0u : p.comp_off[c];
    const uint32_t oo = p.out_off ? p.out_off[c] : 0u;
    const T cf = p.out_off ? p.coeff[c] : zero;
    for (int64_t item0 = int64_t(blockIdx.y) * U; item0 < p.batch; item0 += int64_t(gridDim.y) * U) {
        T x[U][NOPS], v[U], s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t item = item0 + u < p.batch ? item0 + u : item0;   // (the tail re-reads the step's first item; it is not stored twice)
#pragma unroll
            for (int k = 0; k < NOPS; ++k) x[u][k] = (op[k] & 1u) ? ld_ptr[k][item * ld_stride[k]] : zero;   // ops 1 and 3 read
            v[u] = p.load_first ? p.res[item * p.res_stride + own] : zero;
            s[u] = p.out_off ? p.scalar[item * p.scalar_stride + p.scalar_off] : zero;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (item0 + u >= p.batch) break;
            T val = v[u];
#pragma unroll
            for (int k = 0; k < NOPS; ++k) {
                if (op[k] == 2u) val = -val;                                       // graded.rs:63
                else if (op[k]) val = (op[k] == 3u ? zero : val) + x[u][k];       // graded.rs:74 (op 3: onto the fresh +0.0 of init_null_mv)
            }
            const int64_t item = item0 + u;
            if (p.out_off) {
                T sc = s[u];
                if (p.canon_s) sc = zero + sc;
                if (p.canon_v) val = zero + val;
                const T pr = p.s_is_left ? sc * val : val * sc;
                p.out[item * p.out_stride + oo] = zero + pr * cf;         // the fresh result buffer: 0.0 + (l * r) * coeff
            } else {
                p.res[item * p.res_stride + own] = val;
            }
        }
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

// ------------------------------------------------------------------------------------------
// k_product_ell_chain: TWO list products in one launch -- list 1's result (the "mid" row) is read only by list 2, as one of its
// operands: (R X) ~R projected on a grade, the rotor sandwich applied to a vector (eval.rs:61-86 with its cached operand; README.md:62-67)
// at the dimensions where the program no longer fits a fused small-program kernel and the second product is too sparse for the
// dense kernels (n rows of 2^(n-1) terms).  Both lists keep the reference's order and roundings (eval.rs:82), the mid row is
// canonicalised as the second product's operand staging would have (0.0 + x, eval.rs:27-31): bit for bit the two-launch plan,
// without the mid row's round trip through HBM and with every lane busy in the second list:
//   * a workgroup stages the operand rows of IPB items (a power of two; about 36 KiB of LDS, so that four workgroups share a CU
//     and one's staging overlaps another's lists) in LDS, item i at i * item_stride elements with item_stride = 1 (mod 32):
//     lanes that read the SAME offset of consecutive items touch different banks;
//   * list 1 (many short rows): a thread takes a row and walks the items with the row's words in registers;
//   * list 2 (few long rows) runs over (row, item) pairs, item fastest -- its n rows occupy n x IPB lanes instead of n -- with its
//     words in LDS (copied once per persistent workgroup): no cache round trip inside a row's chain of additions.
// Entries: left offset [14:0] | right offset [30:16] | negate [31], offsets in BYTES (rows of at most 32 KiB).
// ------------------------------------------------------------------------------------------
template <typename T>
struct EllChainArgs {
    const T* l1;
    const T* r1;
    const T* r2;
    T* out;
    int64_t l1_stride, r1_stride, r2_stride, out_stride;
    int l1_len, r1_len, r2_len, mid_len;
    int canon_l1, canon_r1, canon_r2, canon_mid;
    const uint32_t* ent1;   // [width1][rows1]
    const uint32_t* pos1;   // element offset of each row of list 1 in the mid row
    int rows1, width1;
    const uint32_t* ent2;   // [width2][rows2]
    const uint32_t* out2;   // output offset of each row of list 2
    int rows2, width2;
    int mid_is_left;        // list 2 reads the mid row as its left (1) or right (0) operand
    int r2_alias;           // list 2's other operand: 0 = a row of its own (r2), 1 = list 1's left row, 2 = list 1's right row
    int mid_covered;        // every component of the mid row is a row of list 1 (else the rest is zero)
    int beta;
    int ipb, item_stride;   // items per workgroup (any count: rows2 x ipb is chosen just under a multiple of 64), elements per item in LDS
    int ent2_lds_bytes;     // > 0: list 2's words are copied to LDS once per (persistent) workgroup; a multiple of 16
    int64_t batch;
};

template <typename T>
__global__ __launch_bounds__(512) void k_product_ell_chain(EllChainArgs<T> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int ipb = p.ipb;
    const float inv_ipb = 1.0f / float(ipb);
    const T zero = T(0);
    // LDS: [list 2's words, when they fit (ent2_lds)] then per item [list 1 left | list 1 right | mid | list 2's own operand]
    uint32_t* ent2_lds = reinterpret_cast<uint32_t*>(smem_raw);
    T* smem = reinterpret_cast<T*>(smem_raw + p.ent2_lds_bytes);
    // ([row][term] with rows 4 words apart in the banks: a lane's next eight words are two 16-byte reads, lanes of one row read one address)
    const int ent2_row = p.width2 + 4;
    if (p.ent2_lds_bytes)
        for (int e = tid; e < p.rows2 * p.width2; e += nthr) {
            const int t = e / p.rows2, row = e - t * p.rows2;
            ent2_lds[row * ent2_row + t] = p.ent2[e];
        }
    const int off_r1 = p.l1_len, off_mid = off_r1 + p.r1_len, off_r2 = off_mid + p.mid_len;
    const int stride_b = p.item_stride * int(sizeof(T));
    const char* sb = reinterpret_cast<const char*>(smem);
    const int off_other = (p.r2_alias == 1 ? 0 : p.r2_alias == 2 ? off_r1 : off_r2) * int(sizeof(T));
    const int off_l = p.mid_is_left ? off_mid * int(sizeof(T)) : off_other, off_r = p.mid_is_left ? off_other : off_mid * int(sizeof(T));
    auto term = [&](uint32_t e, const char* l, const char* r, T a) -> T {   // eval.rs:82 with coeff = +-1
        return a + ell_flip(*reinterpret_cast<const T*>(l + (e & 0x7fffu)) * *reinterpret_cast<const T*>(r + ((e >> 16) & 0x7fffu)), e & 0x80000000u);
    };
    const int64_t groups = (p.batch + ipb - 1) / ipb;
    for (int64_t g = blockIdx.x; g < groups; g += gridDim.x) {   // persistent workgroups: list 2's words are fetched once
        const int64_t item0 = g * ipb;
        const int nitems = int(p.batch - item0 < ipb ? p.batch - item0 : ipb);
        // operand rows of the group's items: element e = it * len + c of a flattened (item, component) range, four loads in
        // flight per thread (the split by a run-time length is a float reciprocal and one correction step, valid below 2^24)
        auto stage = [&](const T* src, int64_t stride, int len, int canon, int off) {
            const int total = nitems * len;
            const float inv = 1.0f / float(len > 0 ? len : 1);
            for (int e0 = tid; e0 < total; e0 += 4 * nthr) {
                T v[4];
                int dst[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = e0 + k * nthr;
                    int it = int(float(e) * inv), c = e - it * len;
                    if (c < 0) {
                        --it;
                        c += len;
                    }
                    if (c >= len) {
                        ++it;
                        c -= len;
                    }
                    dst[k] = e < total ? it * p.item_stride + off + c : -1;
                    v[k] = e < total ? src[(item0 + it) * stride + c] : zero;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (dst[k] >= 0) smem[dst[k]] = canon ? zero + v[k] : v[k];
            }
        };
        stage(p.l1, p.l1_stride, p.l1_len, p.canon_l1, 0);
        stage(p.r1, p.r1_stride, p.r1_len, p.canon_r1, off_r1);
        if (!p.r2_alias) stage(p.r2, p.r2_stride, p.r2_len, p.canon_r2, off_r2);
        if (!p.mid_covered)
            for (int it = 0; it < nitems; ++it)
                for (int c = tid; c < p.mid_len; c += nthr) smem[it * p.item_stride + off_mid + c] = zero;
        __syncthreads();
        // ---- list 1 -> mid: a thread takes a ROW and walks the group's items with the row's words in registers (rows of up to 16
        // terms -- R X has n; longer rows re-read their words from L1 per item): one fetch of the words per row and group ----
        for (int row = tid; row < p.rows1; row += nthr) {
            const uint32_t* ep = p.ent1 + row;
            // the row's words, decoded once: byte offsets from the item's left row (the right row sits off_r1 further) and the sign
            uint32_t lo[16], ro[16], sg[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t e = j < p.width1 ? ep[size_t(j) * p.rows1] : 0u;
                lo[j] = e & 0x7fffu;
                ro[j] = ((e >> 16) & 0x7fffu) + uint32_t(off_r1) * uint32_t(sizeof(T));
                sg[j] = e & 0x80000000u;
            }
            const int pos = off_mid + int(p.pos1[row]);
            for (int it = 0; it < nitems; ++it) {
                const char* l = sb + it * stride_b;
                T acc = zero;                               // the fresh cache buffer of eval.rs:21-33
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (j < p.width1) acc = acc + ell_flip(*reinterpret_cast<const T*>(l + lo[j]) * *reinterpret_cast<const T*>(l + ro[j]), sg[j]);   // eval.rs:82, coeff = +-1
                for (int t = 16; t < p.width1; ++t) acc = term(ep[size_t(t) * p.rows1], l, l + off_r1 * int(sizeof(T)), acc);
                smem[it * p.item_stride + pos] = p.canon_mid ? zero + acc : acc;
            }
        }
        __syncthreads();
        // ---- list 2: (mid, other operand) -> out, over (row, item) pairs, item fastest: its few rows fill rows2 x IPB lanes ----
        {
            const int pairs = p.rows2 * ipb;
            for (int w = tid; w < pairs; w += nthr) {
                int row = int(float(w) * inv_ipb), it = w - row * ipb;   // w = row * ipb + it (w < 2^24: one correction step)
                if (it < 0) {
                    --row;
                    it += ipb;
                }
                if (it >= ipb) {
                    ++row;
                    it -= ipb;
                }
                if (it >= nitems) continue;
                const char* l = sb + it * stride_b + off_l;
                const char* r = sb + it * stride_b + off_r;
                T* o = p.out + (item0 + it) * p.out_stride + p.out2[row];
                T acc = p.beta ? *o : zero;
                int t = 0;
                if (p.ent2_lds_bytes) {
                    const uint32_t* ep = ent2_lds + row * ent2_row;
                    for (; t + 16 <= p.width2; t += 16) {   // sixteen terms: their words (four 16-byte reads) and 32 operand reads in flight, then the chain
                        uint32_t ev[16];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const uint4 w4 = *reinterpret_cast<const uint4*>(ep + t + 4 * q);
                            ev[4 * q] = w4.x; ev[4 * q + 1] = w4.y; ev[4 * q + 2] = w4.z; ev[4 * q + 3] = w4.w;
                        }
                        T prod[16];
#pragma unroll
                        for (int j = 0; j < 16; ++j)
                            prod[j] = ell_flip(*reinterpret_cast<const T*>(l + (ev[j] & 0x7fffu)) * *reinterpret_cast<const T*>(r + ((ev[j] >> 16) & 0x7fffu)),
                                               ev[j] & 0x80000000u);
#pragma unroll
                        for (int j = 0; j < 16; ++j) acc = acc + prod[j];
                    }
                    for (; t + 8 <= p.width2; t += 8) {
                        uint32_t ev[8];
                        {
                            const uint4 q0 = *reinterpret_cast<const uint4*>(ep + t), q1 = *reinterpret_cast<const uint4*>(ep + t + 4);
                            ev[0] = q0.x; ev[1] = q0.y; ev[2] = q0.z; ev[3] = q0.w;
                            ev[4] = q1.x; ev[5] = q1.y; ev[6] = q1.z; ev[7] = q1.w;
                        }
                        T prod[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            prod[j] = ell_flip(*reinterpret_cast<const T*>(l + (ev[j] & 0x7fffu)) * *reinterpret_cast<const T*>(r + ((ev[j] >> 16) & 0x7fffu)),
                                               ev[j] & 0x80000000u);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc = acc + prod[j];
                    }
                    for (; t < p.width2; ++t) acc = term(ep[t], l, r, acc);
                } else {
                    const uint32_t* ep = p.ent2 + row;
                    for (; t + 8 <= p.width2; t += 8) {
                        uint32_t ev[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) ev[j] = ep[size_t(t + j) * p.rows2];
                        T prod[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            prod[j] = ell_flip(*reinterpret_cast<const T*>(l + (ev[j] & 0x7fffu)) * *reinterpret_cast<const T*>(r + ((ev[j] >> 16) & 0x7fffu)),
                                               ev[j] & 0x80000000u);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc = acc + prod[j];
                    }
                    for (; t < p.width2; ++t) acc = term(ep[size_t(t) * p.rows2], l, r, acc);
                }
                *o = acc;
            }
        }
        __syncthreads();   // the rows are rewritten by the next group
    }
}

}  // namespace gaast
