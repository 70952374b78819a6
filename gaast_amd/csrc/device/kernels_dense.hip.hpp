// kernels_dense.hip.hpp -- dense geometric product in blade-bitmask space: vector-FMA and matrix-core kernels
// Included through kernels.hip.hpp.
#pragma once
#include "kernels_common.hip.hpp"


namespace gaast {

// ------------------------------------------------------------------------------------------
// Product arm, dense geometric product, tiled in blade-bitmask space.
//
// In bitmask space e_a e_b = s(a,b) m(a&b) e_{a^b}  (algebra.rs:73-83), a twisted
// XOR-convolution.  Split a blade into hi = a >> 4 and lo = a & 15:
//     s(a,b) = s_hi(a_hi,b_hi) * s_lo(a_lo,b_lo) * (-1)^(|a_hi| |b_lo|)
// so an aligned 16-block of A times an aligned 16-block of B lands in exactly one 16-block
// of C, with a compile-time sign pattern (two variants, by the parity of |a_hi|) and one
// run-time sign per (a_hi, b_hi).  The metric of the hi basis vectors (+1/-1/0) is folded into the
// per-block sign / zero factor:
//     (-1)^|a_hi & b_hi & NEG|  and  [a_hi & b_hi & ZERO == 0].
// The lo basis vectors must square to +-1 (never 0): their factor (-1)^|a_lo & b_lo & NEG_lo| is a lane
// constant in the matrix-core kernels (any NEG_lo) and a compile-time pattern in the vector-FMA kernel
// (NEG_lo = none or all four).  The host gets any diagonal metric into that shape by PERMUTING the basis
// vectors (plan.cpp: dense_basis_permutation): the kernels work on blades of the permuted basis, the index maps
// carry the blade bijection and the reordering sign of every blade in their negate bits.
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
    const int32_t* out_map;     // per bitmask: offset in the out row [29:0] | negate << 30, or -1
    int canon_left, canon_right;
    int n;                      // vector-space dimension, 4 <= n
    uint32_t neg_hi, zero_hi;   // metric signature of basis vectors 4.. (bit i <-> vector 4+i)
    uint32_t neg_lo;            // lo basis vectors that square to -1
    int beta;
    // general diagonal metric (entries other than +-1 / 0): the kernels work in the rescaled basis f_i = e_i / sqrt|g_i|, whose
    // metric is +-1 / 0.  Per loaded component: w_S = prod_{i in S} sqrt|g_i| (operands); per permuted blade: 1 / w_T
    // (result).  NULL for +-1 / 0 metrics.
    const T* left_scale;
    const T* right_scale;
    const T* out_scale;
    // chained product (plan.cpp: chain_sparse_into_dense): the LEFT operand is the result of a short comp-mul list over two
    // other rows; it is evaluated in LDS while staging (stage_from_list), `left` is unused.  NULL otherwise.
    const T* pre_left;
    const T* pre_right;
    int64_t pre_left_stride, pre_right_stride;
    int pre_left_len, pre_right_len, pre_canon_left, pre_canon_right;
    const uint32_t* pre_row_start;   // pre_rows + 1
    const uint32_t* pre_entries;     // left offset | right offset << 16
    const T* pre_coeff;
    const uint32_t* pre_row_map;     // per row: image position << 16 | negate << 31
    const T* pre_row_scale;          // per row (rescaled basis) or NULL
    int pre_rows;
    int pre_width;                   // > 0: rows of one length, +-1 coefficients: pre_entries is [term][row], sign in bit 31
    int pre_scratch;                 // element offset of the list's operand rows in the kernel's LDS
    int left_signs;             // some left_map word has its negate bit set (folded sign arms, permuted basis)
    int out_signs;              // some out_map word has its sign bit set (permuted basis)
    int64_t batch;
};

// physical position of blade bitmask m inside an operand's LDS image: blocks of 16, the four
// 16-byte quads of block x rotated by (x >> 2) & 3 so that 16 lanes reading the same logical
// quad of 16 different blocks touch 16 different bank quads.
__device__ __forceinline__ int dense_lds_pos(int m) {
    const int x = m >> 4, lo = m & 15;
    return (x << 4) | ((((lo >> 2) ^ (x >> 2)) & 3) << 2) | (lo & 3);
}


// e = it * cnt + j with 0 <= j < cnt, for e < 2^24: a float reciprocal and one correction step instead of the ~30
// instructions of an integer division by a run-time value (every vector instruction of a resident wave costs its SIMD ~4.8
// cycles of matrix-pipe time: the general staging spent most of its time dividing).  nitems == 1 (known at compile time in the
// one-item-per-workgroup kernels): nothing to divide.
__device__ __forceinline__ void split_index(int e, int cnt, float inv_cnt, int nitems, int& it, int& j) {
    if (nitems == 1) {
        it = 0;
        j = e;
        return;
    }
    int q = int(float(e) * inv_cnt);
    int r = e - q * cnt;
    if (r < 0) {
        --q;
        r += cnt;
    }
    if (r >= cnt) {
        ++q;
        r -= cnt;
    }
    it = q;
    j = r;
}

// Scatter the operand rows of `nitems` items into their LDS images.  A map word is
//   row offset [15:0] | position in the LDS image [30:16] | negate [31]
// (the position is computed on the host for the kernel that consumes the image).  Loads of a
// trip are all issued before the first use.  When the offsets are simply 0, 1, 2, ... (an
// operand holding every blade) four consecutive components move per 16-byte load.
template <typename T, int THREADS>
__device__ __forceinline__ void stage_operands(const T* __restrict__ src, int64_t stride, const uint32_t* __restrict__ map,
                                               int count, int contig4, int canon, T* __restrict__ images,
                                               int image_stride, int nitems, int tid, const T* __restrict__ scale = nullptr) {
    constexpr int U = 4;
    const float inv_count = 1.0f / float(count > 0 ? count : 1);
    if (contig4) {
        const int total4 = (nitems * count) >> 2;  // count % 4 == 0
        const int count4 = count >> 2;
        const float inv_count4 = 1.0f / float(count4 > 0 ? count4 : 1);
        for (int e0 = tid; e0 < total4; e0 += THREADS * U) {
            uint4 m[U];
            T v[U][4];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int e = e0 + k * THREADS;
                if (e < total4) {
                    int it, j4;
                    split_index(e, count4, inv_count4, nitems, it, j4);
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
                    int it, j4;
                    split_index(e, count4, inv_count4, nitems, it, j4);
                    T* img = images + it * image_stride;
                    const uint32_t mm[4] = {m[k].x, m[k].y, m[k].z, m[k].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        T x = v[k][c];
                        if (canon) x = T(0) + x;  // the reference's zero-init + add_grades_from copy: 0.0 + x
                        if (scale) x = x * scale[(j4 << 2) + c];   // rescaled basis: w_S A_S
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
        int its[U], js[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int e = e0 + k * THREADS;
            split_index(e < total ? e : 0, count, inv_count, nitems, its[k], js[k]);
            m[k] = e < total ? map[js[k]] : 0u;
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int e = e0 + k * THREADS;
            v[k] = e < total ? src[int64_t(its[k]) * stride + (m[k] & 0xffffu)] : T(0);
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int e = e0 + k * THREADS;
            if (e < total) {
                T x = v[k];
                if (canon) x = T(0) + x;
                if (scale) x = x * scale[js[k]];
                if (m[k] >> 31) x = -x;
                images[its[k] * image_stride + ((m[k] >> 16) & 0x7fffu)] = x;
            }
        }
    }
}

// The left operand of a CHAINED product: rows of a comp-mul list over two other operand rows, evaluated here, in LDS, in the
// reference's order with its roundings (eval.rs:82: `res += (left * right) * coeff`, no contraction) -- bit for bit what
// k_product_csr / k_product_ell would have written to HBM.  scratch: nitems x (pre_left_len + pre_right_len) elements.  Ends
// with the operand image written; the caller's barrier follows.
//   pre_width > 0: every row has pre_width entries with coefficients +-1 (sparse products of non-degenerate algebras: R X has n
//   per row): the list is stored [term][row] with the sign in bit 31, so that a wave's rows read consecutive words, no
//   coefficient is loaded ((l r) (-1.0) == -(l r) exactly), and -- up to 16 terms -- a thread keeps its row's entries in
//   registers for all the items it serves.  Otherwise: CSR with general coefficients.
template <typename T>
__device__ __forceinline__ T list_flip(T v, uint32_t sign31);
template <>
__device__ __forceinline__ float list_flip<float>(float v, uint32_t sign31) { return __uint_as_float(__float_as_uint(v) ^ sign31); }
template <>
__device__ __forceinline__ double list_flip<double>(double v, uint32_t sign31) {
    return __hiloint2double(__double2hiint(v) ^ int(sign31), __double2loint(v));
}

// the list's operand rows of `nitems` items -> scratch (raw, canonicalised like a product operand of eval.rs:27-31)
template <typename T, int THREADS>
__device__ __forceinline__ void list_fill_scratch(const DenseArgs<T>& p, int64_t item0, int nitems, T* __restrict__ scratch, int tid) {
    // per item: the left row, the right row, ONE ZERO (the operand pair of the entries that pad a row to a multiple of 4)
    const int ll = p.pre_left_len, rl = p.pre_right_len, per = ll + rl + 1;
    const T zero = T(0);
    const float inv_per = 1.0f / float(per);
    for (int e = tid; e < nitems * per; e += THREADS) {
        int it, c;
        split_index(e, per, inv_per, nitems, it, c);
        T v;
        if (c == ll + rl) {
            v = zero;
        } else if (c < ll) {
            v = p.pre_left[(item0 + it) * p.pre_left_stride + c];
            if (p.pre_canon_left) v = zero + v;
        } else {
            v = p.pre_right[(item0 + it) * p.pre_right_stride + (c - ll)];
            if (p.pre_canon_right) v = zero + v;
        }
        scratch[e] = v;
    }
}

// the rows of the list over the operand rows in scratch -> the operand image (and, neg_off != 0, its negated copy neg_off
// elements further: the image-pair kernels' register-prefetch path has no separate negation pass)
template <typename T, int THREADS>
__device__ __forceinline__ void list_eval_rows(const DenseArgs<T>& p, int nitems, T* __restrict__ images, int image_stride,
                                               const T* __restrict__ scratch, int tid, int neg_off = 0) {
    const int ll = p.pre_left_len, rl = p.pre_right_len, per = ll + rl + 1;
    const T zero = T(0);
    const int R = p.pre_rows;
    // rows are dealt to threads; when there are more threads than rows, THREADS / R groups of threads serve different items
    const int G = THREADS >= R ? THREADS / R : 1;
    const int g = THREADS >= R ? tid / R : 0;
    const int row_step = THREADS >= R ? R : THREADS;
    if (g >= G) return;
    for (int row = THREADS >= R ? tid - g * R : tid; row < R; row += row_step) {
        const uint32_t w = p.pre_row_map[row];
        const uint32_t pos = (w >> 16) & 0x7fffu, neg = w & 0x80000000u;
        const T sc = p.pre_row_scale ? p.pre_row_scale[row] : T(1);
        if (p.pre_width > 0) {
            // pre_width is a multiple of 4 (the host pads a row with entries over the zero pair: acc + (+0.0) leaves every acc
            // bit for bit, and acc is never -0.0 once it has absorbed a term).  Four terms at a time: their entries (consecutive
            // words for a wave's rows) and eight operand reads are in flight together, then the products join the chain in the
            // list's order (eval.rs:82).  [A switch over compile-time widths 1 ... 16, with a row's entries held in registers for
            // all the items a thread serves, measured 10 % faster on the vector kernel (sand8) and the same elsewhere -- and took
            // the library's build from 2 to 9 minutes.]
            for (int it = g; it < nitems; it += G) {
                const char* l = reinterpret_cast<const char*>(scratch + it * per);
                const char* r = l + size_t(ll) * sizeof(T);
                T acc = zero;                                      // the fresh cache buffer of eval.rs:21-33
                for (int k0 = 0; k0 < p.pre_width; k0 += 4) {
                    uint32_t ev[4];
                    T prod[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) ev[k] = p.pre_entries[(k0 + k) * R + row];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        prod[k] = list_flip<T>(*reinterpret_cast<const T*>(l + (ev[k] & 0x7fffu) * uint32_t(sizeof(T))) *
                                                   *reinterpret_cast<const T*>(r + ((ev[k] >> 16) & 0x7fffu) * uint32_t(sizeof(T))),
                                               ev[k] & 0x80000000u);
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc = acc + prod[k];
                }
                if (p.pre_row_scale) acc = acc * sc;
                {
                    const T val = list_flip<T>(acc, neg);
                    images[it * image_stride + pos] = val;
                    if (neg_off) images[it * image_stride + pos + neg_off] = -val;
                }
            }
        } else {
            const uint32_t e0 = p.pre_row_start[row], e1 = p.pre_row_start[row + 1];
            for (int it = g; it < nitems; it += G) {
                const T* l = scratch + it * per;
                const T* r = l + ll;
                T acc = zero;
                for (uint32_t k = e0; k < e1; ++k) {
                    const uint32_t lr = p.pre_entries[k];
                    acc = acc + (l[lr & 0xffffu] * r[lr >> 16]) * p.pre_coeff[k];   // eval.rs:82
                }
                if (p.pre_row_scale) acc = acc * sc;
                {
                        const T val = list_flip<T>(acc, neg);
                        images[it * image_stride + pos] = val;
                        if (neg_off) images[it * image_stride + pos + neg_off] = -val;
                    }
            }
        }
    }
}

template <typename T, int THREADS>
__device__ __forceinline__ void stage_from_list(const DenseArgs<T>& p, int64_t item0, int nitems, T* __restrict__ images,
                                                int image_stride, T* __restrict__ scratch, int tid) {
    list_fill_scratch<T, THREADS>(p, item0, nitems, scratch, tid);
    if (THREADS > 64) __syncthreads();
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // a single wave: its own writes, in order
    list_eval_rows<T, THREADS>(p, nitems, images, image_stride, scratch, tid);
}

// result component -> graded row: out_map word = offset | negate << 30 (the blade's reordering sign under the basis
// permutation), -1 = not produced
template <typename T>
__device__ __forceinline__ void store_result(T* __restrict__ orow, int32_t w, T v, int beta, const T* __restrict__ out_scale = nullptr,
                                             int blade = 0) {
    if (w < 0) return;
    const int32_t off = w & 0x3fffffff;
    if (out_scale) v = v * out_scale[blade];   // back from the rescaled basis: C_T = C'_T / w_T
    if (w & 0x40000000) v = beta ? -v : T(0) + (-v);   // reordering sign; a zero result stays +0.0 as in the reference
    orow[off] = beta ? orow[off] + v : v;
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
template <typename T, bool ODD, bool NEGLO>
__device__ __forceinline__ void gp_block16(const T (&A)[16], const T (&B)[16], T (&C)[16]) {
#pragma unroll
    for (int al = 0; al < 16; ++al) {
#pragma unroll
        for (int bl = 0; bl < 16; ++bl) {
            const int neg = lo_reorder_parity(al, bl) ^ (ODD ? (__builtin_popcount(bl) & 1) : 0) ^
                            (NEGLO ? (__builtin_popcount(al & bl) & 1) : 0);
            C[al ^ bl] = neg ? fma_t<T>(-A[al], B[bl], C[al ^ bl]) : fma_t<T>(A[al], B[bl], C[al ^ bl]);
        }
    }
}

// ---- 16x16 block product, f32: 128 v_pk_fma_f32, no operand shuffles ---------------------------
// Accumulator pair j holds C[2j], C[2j+1].  For the term with left component a, the low half
// needs B[a ^ 2j] and the high half B[a ^ 2j ^ 1]: the two halves of ONE B pair, swapped when a
// is odd; A[a] is one half of an A pair, broadcast.  op_sel / op_sel_hi pick the halves and
// neg_lo / neg_hi carry the compile-time signs, so every FMA is a single instruction.

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

template <bool ODD, bool NEGLO, int IDX>
__device__ __forceinline__ void gp_block16_pk(const float2v (&A2)[8], const float2v (&B2)[8], float2v (&C2)[8]) {
    if constexpr (IDX < 128) {
        constexpr int a = IDX >> 3, j = IDX & 7, c0 = 2 * j;
        constexpr int b_lo = a ^ c0, b_hi = a ^ c0 ^ 1;
        constexpr int nl = lo_reorder_parity(a, b_lo) ^ (ODD ? (__builtin_popcount(b_lo) & 1) : 0) ^
                           (NEGLO ? (__builtin_popcount(a & b_lo) & 1) : 0);
        constexpr int nh = lo_reorder_parity(a, b_hi) ^ (ODD ? (__builtin_popcount(b_hi) & 1) : 0) ^
                           (NEGLO ? (__builtin_popcount(a & b_hi) & 1) : 0);
        pk_fma_sel<(a & 1), nl, nh>(C2[j], A2[a >> 1], B2[b_lo >> 1]);
        gp_block16_pk<ODD, NEGLO, IDX + 1>(A2, B2, C2);
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
    template <bool ODD, bool NEGLO>
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
        gp_block16_pk<ODD, NEGLO, 0>(A2, B2, C2);
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
    template <bool ODD, bool NEGLO>
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
        gp_block16<double, ODD, NEGLO>(A, B, C);
    }
    __device__ __forceinline__ double get(int i) const { return C[i]; }
};

// NEGLO: all four lo basis vectors square to -1 (else: all four to +1)
template <typename T, bool DEGENERATE, int THREADS, bool NEGLO, bool SCALED = false, bool CHAINED = false>
__global__ __launch_bounds__(THREADS) void k_gp_dense(DenseArgs<T> p) {
    // SCALED (a general diagonal metric, DenseArgs::left_scale ...): a separate instantiation, so that the +-1 / 0 kernels
    // carry none of its code or registers
    const T* const left_scale = SCALED ? p.left_scale : nullptr;
    const T* const right_scale = SCALED ? p.right_scale : nullptr;
    const T* const out_scale = SCALED ? p.out_scale : nullptr;
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
        {
            if (!p.left_full || !p.right_full) {
                for (int i = tid; i < nitems * item_stride; i += THREADS) smem[i] = zero;
                __syncthreads();
            }
            if constexpr (CHAINED)   // (its own instantiation: the plain kernels carry none of this code or its registers)
                stage_from_list<T, THREADS>(p, item0, nitems, smem, item_stride, smem + p.pre_scratch, tid);
            else
                stage_operands<T, THREADS>(p.left + item0 * p.left_stride, p.left_stride, p.left_map, p.left_count, p.left_contig,
                                           p.canon_left, smem, item_stride, nitems, tid, left_scale);
            stage_operands<T, THREADS>(p.right + item0 * p.right_stride, p.right_stride, p.right_map, p.right_count,
                                       p.right_contig, p.canon_right, smem + N, item_stride, nitems, tid, right_scale);
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
                acc.template step<ODD, NEGLO>(As, Bs, a_hi, a_hi ^ c_hi, sgn);
            };
            const int half = LPI >> 1;
            for (int i = 0; i < half; ++i) one_step(std::false_type{}, (i << 1) | (__builtin_popcount(uint32_t(i)) & 1));
            for (int i = 0; i < half; ++i) one_step(std::true_type{}, (i << 1) | ((__builtin_popcount(uint32_t(i)) & 1) ^ 1));

            // ---- scatter the 16 accumulators to their positions in the graded row ----
            T* orow = p.out + (item0 + it) * p.out_stride;
#pragma unroll
            for (int i = 0; i < 16; ++i) store_result<T>(orow, om[i], acc.get(i), p.beta, out_scale, (c_hi << 4) + i);
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
// The low FIVE basis vectors square to +-1 (p.neg_lo; the host permutes the basis to make it so).
// ------------------------------------------------------------------------------------------

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

template <bool DEGENERATE, int THREADS, bool SCALED = false, bool CHAINED = false>
__global__ __launch_bounds__(THREADS) void k_gp_mfma32(DenseArgs<float> p) {
    const float* const left_scale = SCALED ? p.left_scale : nullptr;
    const float* const right_scale = SCALED ? p.right_scale : nullptr;
    const float* const out_scale = SCALED ? p.out_scale : nullptr;
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
    {
        if constexpr (CHAINED)   // (its own instantiation: the plain kernels carry none of this code or its registers)
            stage_from_list<float, THREADS>(p, item0, nitems, smem, item_stride, smem + p.pre_scratch, tid);
        else
            stage_operands<float, THREADS>(p.left + item0 * p.left_stride, p.left_stride, p.left_map, p.left_count, p.left_contig,
                                       p.canon_left, smem, item_stride, nitems, tid, left_scale);
        stage_operands<float, THREADS>(p.right + item0 * p.right_stride, p.right_stride, p.right_map, p.right_count,
                                       p.right_contig, p.canon_right, smem + N, item_stride, nitems, tid, right_scale);
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
            int par = __builtin_popcount(uint32_t(a_lo & k) & p.neg_lo) & 1;   // lo vectors that square to -1
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
        // lane constants of the B side: byte offset of logical quad q of the lane's block, (lane constant) ^ (step constant)
        // as in k_gp_mfma16; c_hi with a spare bit set, so that the uniform part u of the block sign rides in the popcount
        const unsigned char* Bs_b = reinterpret_cast<const unsigned char*>(Bs);
        uint32_t bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = (uint32_t(c_hi) << 7) | (uint32_t((((h << 2) | q) ^ (c_hi >> 1)) & 7) << 4);
        const uint32_t c_hi_u = uint32_t(c_hi) | 0x8000u;
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
            const uint32_t bmask = uint32_t(__builtin_popcount(c_hi_u & (M | (u << 15)))) << 31;
            float bscale = 1.f;
            if (DEGENERATE) {
                if (uint32_t(a_hi) & ~uint32_t(c_hi) & p.zero_hi) bscale = 0.f;
            }
            float bv[16];
            const uint32_t sx = (uint32_t(a_hi) << 7) | (uint32_t((a_hi >> 1) & 7) << 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4v v = *reinterpret_cast<const float4v*>(Bs_b + (bq[q] ^ sx));
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
            store_result<float>(orow, om[c_lo], acc[r], p.beta, out_scale, (c_hi << 5) + c_lo);
        }
    }
}

// Workgroup barrier for LDS traffic only.  __syncthreads() also waits for the wave's outstanding GLOBAL stores and
// loads (vmcnt(0)): between the groups of a persistent workgroup that is the round trip of the result rows to HBM.
// A single-wave workgroup needs no barrier at all: the LDS executes a wave's instructions in order.
template <int THREADS>
__device__ __forceinline__ void lds_barrier() {
    if (THREADS <= 64) {
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef uint32_t uint4v __attribute__((ext_vector_type(4)));

// term t of a step multiplies the words k = mfma16_k(t): even |k| first
__device__ __forceinline__ constexpr int mfma16_k(int t) {
    constexpr int order[16] = {0, 3, 5, 6, 9, 10, 12, 15, 1, 2, 4, 7, 8, 11, 13, 14};
    return order[t];
}

// ------------------------------------------------------------------------------------------
// k_gp_mfma32 in image-pair form (n = 10 ... 13): as k_gp_mfma16 below, NO sign is applied with vector instructions
// (each costs the SIMD ~4.8 cycles of matrix-pipe time, tools/microbench/mfma16_loop2.hip).  Every item keeps +A, -A,
// +B, -B images in LDS (4 x 2^n words) and every sign is an address:
//   * A operand of term t (s2 = mfma16_k(t), k = 2 s2 + h): one ds_read_b32 at (lane constant of t) + 128 a_hi; the
//     constant points into +A or -A by the lane-constant sign of (i, k);
//   * B operand: the lane's 16 words of block a_hi ^ c_hi from +B or -B by the block sign (one address bit per step);
//   * (-1)^(|a_hi| |k|), |a_hi| = |b_hi| + |c_hi| (mod 2): the b_hi part is folded into the B images (host map); for
//     the c_hi part the lane's 16 words are stored even-|s2| first, so that each 16-byte quad holds words of one |k|
//     parity, and lanes with odd |c_hi| read the odd-|k| quads from the image of the other sign.
// Steps run in chunks of eight (immediate offsets for the A reads, the 16 A addresses move once per chunk): per step of
// 16 MFMAs (1,024 matrix-pipe cycles) 16 ds_read_b32 + 4 ds_read_b128 and ~10 vector instructions.
// n = 14 does not fit (4 x 64 KiB) and stays on k_gp_mfma32.
// ------------------------------------------------------------------------------------------
// Persistent workgroups: a workgroup walks the groups of IPB items blockIdx.x, blockIdx.x + gridDim.x, ...; when both
// operands hold every blade in consecutive 16-byte aligned rows (p.left_contig / left_full ...: the host checks), the
// rows of the NEXT group are fetched into registers (8 x 16 B per thread at every n) while the matrix cores work on the
// current one; barriers between the phases wait for LDS traffic only (lds_barrier).
// (The second launch bound, two workgroups per CU, only changes the compiler's scheduling here -- every instantiation
//  stays under the 256 registers of two waves per SIMD anyway: n = 10 / 11 gain 7-9 % with it, n = 12 LOSES 6 %.)
template <bool DEGENERATE, int NDIM, bool SCALED = false, bool CHAINED = false>
__global__ __launch_bounds__(NDIM == 13 ? 512 : 256, NDIM <= 11 ? 2 : 1) void k_gp_mfma32p(DenseArgs<float> p) {
    const float* const left_scale = SCALED ? p.left_scale : nullptr;
    const float* const right_scale = SCALED ? p.right_scale : nullptr;
    const float* const out_scale = SCALED ? p.out_scale : nullptr;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    lds_u8* lds = (lds_u8*)smem_raw;
    constexpr int n = NDIM;
    constexpr int THREADS = NDIM == 13 ? 512 : 256;
    constexpr int N = 1 << n;
    constexpr int H = 1 << (n - 5);               // number of 32-blocks
    constexpr int WPI = H >> 5;                   // waves per item (32 result columns each)
    constexpr int IPB = (THREADS >> 6) / WPI;     // items per group (>= 1)
    constexpr int item_stride = 4 * N;            // words: +A, -A, +B, -B
    constexpr uint32_t NEG = uint32_t(N) << 2;    // bytes from an image to its negated copy (one address bit)
    constexpr int PR = N / 4;                     // 16-byte pieces of a row
    constexpr int MPR = PR / THREADS;             // pieces of one row per thread (1, 2, 4)
    constexpr int ROWS = 2 * IPB;                 // rows of a group: the items' left rows, then their right rows
    static_assert(MPR * ROWS == 8, "eight 16-byte pieces per thread and group");
    const int tid = threadIdx.x;
    const int64_t num_groups = (p.batch + IPB - 1) / IPB;
    const bool fast = !SCALED && !CHAINED && p.left_contig && p.right_contig && p.left_full && p.right_full;   // (a rescaled basis: general staging)

    if (tid < 16) smem[IPB * item_stride + tid] = 0.f;   // the B "block" of a vanishing contribution

    // FAST: thread t moves pieces t + m THREADS (m < MPR) of every row; their map words (position, sign) stay in registers
    uint32_t mw[2][MPR * 4];
    float4 pf[8];
    auto fetch = [&](int64_t g) {
        const int64_t it0 = g * IPB;
        const int cnt = int(p.batch - it0 < IPB ? p.batch - it0 : IPB);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int k = r % IPB;                // item of the group
            if (k < cnt) {
                const float* row = r < IPB ? p.left + (it0 + k) * p.left_stride : p.right + (it0 + k) * p.right_stride;
#pragma unroll
                for (int m = 0; m < MPR; ++m) pf[r * MPR + m] = reinterpret_cast<const float4*>(row)[tid + m * THREADS];
            }
        }
    };
    if (fast) {
#pragma unroll
        for (int m = 0; m < MPR; ++m) {
            const uint4 ml = reinterpret_cast<const uint4*>(p.left_map)[tid + m * THREADS];
            const uint4 mr = reinterpret_cast<const uint4*>(p.right_map)[tid + m * THREADS];
            mw[0][4 * m + 0] = ml.x; mw[0][4 * m + 1] = ml.y; mw[0][4 * m + 2] = ml.z; mw[0][4 * m + 3] = ml.w;
            mw[1][4 * m + 0] = mr.x; mw[1][4 * m + 1] = mr.y; mw[1][4 * m + 2] = mr.z; mw[1][4 * m + 3] = mr.w;
        }
        if (int64_t(blockIdx.x) < num_groups) fetch(blockIdx.x);
    }

    const int wave = tid >> 6, lane = tid & 63;
    const int it = wave / WPI, tile = wave - it * WPI;
    const int i = lane & 31, h = lane >> 5;
    const int c_hi = (tile << 5) | i;
    const uint32_t item_base = uint32_t(it) * uint32_t(item_stride) * 4u;

    // A operand of term t: s2 = mfma16_k(t), k = 2 s2 + h: +-A[a_hi][i ^ k], sign a lane constant: byte address
    // inside the item's +A / -A pair, without the step's 128 a_hi
    // n <= 11: the A addresses are LDS ADDRESSES (the allocation's base included) and are used as such -- otherwise the
    // compiler adds the (zero) base to every one of them in every chunk; at n = 12 that form measured slower (the register
    // allocation it leads to: 16.06 against 15.65 ms), so there the base stays a separate term
    constexpr bool ABS_A = NDIM <= 11;
    const uint32_t lds0 = ABS_A ? uint32_t(size_t(lds)) : 0u;
    uint32_t ak0[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int k = 2 * mfma16_k(t) + h;
        const int a_lo = i ^ k;
        int par = __builtin_popcount(uint32_t(a_lo & k) & p.neg_lo) & 1;   // lo vectors that square to -1
        for (int pp = 1; pp < 5; ++pp)
            if ((a_lo >> pp) & 1) par ^= __builtin_popcount(k & ((1 << pp) - 1)) & 1;
        ak0[t] = lds0 + item_base + (uint32_t(a_lo) << 2) + (par ? NEG : 0u);
    }
    // B side: byte offset of the lane's quad q inside the +B / -B pair = (lane constant) ^ (step constant); quads
    // 0, 1 hold the even-|s2| words (|k| parity h), quads 2, 3 the odd ones: lanes with odd |c_hi| take the odd-|k|
    // quads from the other image.  c_hi with a spare bit set: the uniform part u of the block sign rides in the popcount
    const uint32_t b_base = item_base + 2u * NEG;
    constexpr uint32_t zero_block = uint32_t(IPB) * uint32_t(item_stride) * 4u;
    uint32_t bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bq[q] = (uint32_t(c_hi) << 7) | (uint32_t((((h << 2) | q) ^ (c_hi >> 1)) & 7) << 4);
        if ((__builtin_popcount(uint32_t(c_hi)) & 1) && (((q >> 1) ^ h) & 1)) bq[q] ^= NEG;
    }
    const uint32_t c_hi_u = uint32_t(c_hi) | 0x8000u;
    // block sign of step a_hi for this lane's column: wave-uniform part on the scalar unit, lane part = and + popcount
    auto block_sign = [&](int a_hi) -> uint32_t {
        uint32_t sp = uint32_t(a_hi) >> 1;
        sp ^= sp >> 1;
        sp ^= sp >> 2;
        sp ^= sp >> 4;
        sp ^= sp >> 8;
        const uint32_t M = sp ^ (uint32_t(a_hi) & p.neg_hi);
        const uint32_t u = (__builtin_popcount(uint32_t(a_hi) & sp) ^ __builtin_popcount(uint32_t(a_hi) & p.neg_hi)) & 1u;
        return uint32_t(__builtin_popcount(c_hi_u & (M | (u << 15)))) & 1u;
    };
    // n <= 12: the signs of all H steps as lane-constant bit masks, computed once per launch (a persistent workgroup
    // reuses them for every group); n = 13 (8 words, an 8-fold unrolled loop) keeps computing them per step, and so does
    // the DEGENERATE instantiation at n = 12 (with the masks it needs more than the 256 registers of two waves per SIMD)
    constexpr int SIGN_WORDS = (H <= 64 || (H == 128 && !DEGENERATE)) ? H / 32 : 0;
    uint32_t sgn[SIGN_WORDS > 0 ? SIGN_WORDS : 1] = {0};
#pragma unroll
    for (int w = 0; w < SIGN_WORDS; ++w) {
        uint32_t bits = 0;
#pragma unroll 1
        for (int b = 0; b < 32; ++b) bits |= block_sign(32 * w + b) << b;
        sgn[w] = bits;
    }

    for (int64_t g = blockIdx.x; g < num_groups; g += gridDim.x) {
        const int64_t item0 = g * IPB;
        const int nitems = int(p.batch - item0 < IPB ? p.batch - item0 : IPB);
        // ---- both operands of the group's items into their +/- images ----
        if (fast) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const int k = r % IPB, side = r / IPB;
                if (k < nitems) {
#pragma unroll
                    for (int m = 0; m < MPR; ++m) {
                        const float4 v = pf[r * MPR + m];
                        const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const uint32_t w = mw[side][4 * m + c];
                            float y = x[c];
                            if (side ? p.canon_right : p.canon_left) y = 0.f + y;   // the reference's zero-init + add_grades_from copy
                            const uint32_t yb = __float_as_uint(y) ^ (w & 0x80000000u);
                            const uint32_t at = uint32_t(k) * uint32_t(item_stride * 4) + (side ? 2u * NEG : 0u) + (((w >> 16) & 0x7fffu) << 2);
                            *(__attribute__((address_space(3))) uint32_t*)(lds + at) = yb;
                            *(__attribute__((address_space(3))) uint32_t*)(lds + at + NEG) = yb ^ 0x80000000u;
                        }
                    }
                }
            }
        } else {
            if (!p.left_full || !p.right_full) {
                for (int e = tid; e < nitems * item_stride; e += THREADS) smem[e] = 0.f;
                lds_barrier<THREADS>();
            }
            if constexpr (CHAINED)   // (its own instantiation: the plain kernels carry none of this code or its registers)
                stage_from_list<float, THREADS>(p, item0, nitems, smem, item_stride, smem + p.pre_scratch, tid);
            else
                stage_operands<float, THREADS>(p.left + item0 * p.left_stride, p.left_stride, p.left_map, p.left_count, p.left_contig,
                                           p.canon_left, smem, item_stride, nitems, tid, left_scale);
            stage_operands<float, THREADS>(p.right + item0 * p.right_stride, p.right_stride, p.right_map, p.right_count,
                                           p.right_contig, p.canon_right, smem + 2 * N, item_stride, nitems, tid, right_scale);
            lds_barrier<THREADS>();
            constexpr int quads_per_item = N >> 1;        // 16-byte pieces of +A and +B together
            for (int e = tid; e < nitems * quads_per_item; e += THREADS) {
                const int sit = e / quads_per_item, j = e - sit * quads_per_item;
                const int src = sit * item_stride + (j << 2) + ((j << 2) < N ? 0 : N);
                const float4v v = *reinterpret_cast<const float4v*>(smem + src);
                *reinterpret_cast<float4v*>(smem + src + N) = -v;
            }
        }
        lds_barrier<THREADS>();
        if (fast && g + gridDim.x < num_groups) fetch(g + gridDim.x);   // in flight during the products below

        if (it < nitems) {
            float16v acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            uint32_t ak[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) ak[t] = ak0[t];

            // one step: the lane's B block a_hi ^ c_hi from the image of its block sign (sxs carries the sign as the NEG
            // address bit), the 16 A words of the step at immediate offsets, 16 MFMAs
            auto step = [&](int a_hi, int j, uint32_t sxs) {
                uint32_t bw[16], aw[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t addr = b_base + (bq[q] ^ sxs);
                    if (DEGENERATE) addr = (uint32_t(a_hi) & ~uint32_t(c_hi) & p.zero_hi) ? zero_block : addr;
                    const uint4v v = *(__attribute__((address_space(3))) const uint4v*)(lds + addr);
                    bw[4 * q + 0] = v.x; bw[4 * q + 1] = v.y; bw[4 * q + 2] = v.z; bw[4 * q + 3] = v.w;
                }
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    aw[t] = ABS_A ? *(__attribute__((address_space(3))) const uint32_t*)size_t(ak[t] + uint32_t(j << 7))
                                  : *(__attribute__((address_space(3))) const uint32_t*)(lds + ak[t] + uint32_t(j << 7));
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(aw[t]), __uint_as_float(bw[t]), acc, 0, 0, 0);
                asm volatile("" ::: "memory");   // keep the LDS reads of a step in their step
            };
            if constexpr (SIGN_WORDS > 0 && NDIM <= 11) {
                // block signs from the per-launch bit masks: two vector instructions per step (bit, address bit) instead
                // of five; chunks of 16 steps (the A addresses move once per chunk).
                // The operand reads are SOFTWARE-PIPELINED over the linear sequence of MFMAs m = 16 step + t: the A word of
                // MFMA m + 3 and the B quad after the current one are requested before MFMA m is issued, so that no MFMA
                // waits for a read issued right before it (round 2's loop did: with two waves per SIMD the matrix pipe
                // idled for the part of the LDS latency the other wave's 64-cycle MFMA does not cover -- 89.8 % busy;
                // tools/microbench/mfma_operand_regs.hip: with counted waits every operand can come from LDS at 97 % of
                // the pure matrix rate).  A words rotate through 4 registers, B quads through 2 x 4.
                uint32_t awin[4];
                uint4v qb[2];
                auto b_quad = [&](int a_hi, int q, uint32_t sxs) -> uint4v {
                    uint32_t addr = b_base + (bq[q] ^ sxs);
                    if (DEGENERATE) addr = (uint32_t(a_hi) & ~uint32_t(c_hi) & p.zero_hi) ? zero_block : addr;
                    return *(__attribute__((address_space(3))) const uint4v*)(lds + addr);
                };
                auto a_word = [&](int t, int j) -> uint32_t {   // word t of step j of the current chunk (j = 16: the next chunk's first)
                    return ABS_A ? *(__attribute__((address_space(3))) const uint32_t*)size_t(ak[t] + uint32_t(j << 7))
                                 : *(__attribute__((address_space(3))) const uint32_t*)(lds + ak[t] + uint32_t(j << 7));
                };
                auto sx_of = [&](int a_hi) -> uint32_t {
                    // (wave-uniform: kept on the scalar unit, one v_lshl_or_b32 joins it with the lane's sign bit)
                    uint32_t sx = __builtin_amdgcn_readfirstlane((uint32_t(a_hi) << 7) | (uint32_t((a_hi >> 1) & 7) << 4));
                    asm("" : "+s"(sx));   // whole, in a scalar register: otherwise its parts are OR-ed in one by one on the vector unit
                    return sx;
                };
                // prologue: the first three A words and the first B quad of step 0
                uint32_t sxs = ((sgn[0] & 1u) << (n + 2)) | sx_of(0);   // the current step's; computed once, one step ahead
                qb[0] = b_quad(0, 0, sxs);
#pragma unroll
                for (int t = 0; t < 3; ++t) awin[t] = a_word(t, 0);
#pragma unroll
                for (int w = 0; w < SIGN_WORDS; ++w) {
                    uint32_t sw = sgn[w];
                    const uint32_t next_word_bit0 = w + 1 < SIGN_WORDS ? (sgn[w + 1 < SIGN_WORDS ? w + 1 : w] & 1u) : 0u;
#pragma unroll 1
                    for (int c = 0; c < 2; ++c) {
                        const int a0 = 32 * w + 16 * c;
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const int a_hi = a0 + j;
                            // sign bit of the NEXT step: bit j + 1 of the chunk's bits; past the word: the next word's bit 0
                            const uint32_t nbit = j < 15 ? (sw >> (j + 1)) & 1u : (c == 0 ? (sw >> 16) & 1u : next_word_bit0);
                            const uint32_t nsxs = (nbit << (n + 2)) | sx_of(a_hi + 1);
#pragma unroll
                            for (int t = 0; t < 16; ++t) {
                                const int m = 16 * j + t;                    // position in the chunk's MFMA sequence
                                // requests: A word of MFMA m + 3; at the first MFMA of a quad, the following quad
                                const int t3 = (t + 3) & 15, j3 = j + ((t + 3) >> 4);
                                awin[(m + 3) & 3] = a_word(t3, j3);
                                if ((t & 3) == 0) {
                                    const int q = t >> 2;
                                    qb[(q + 1) & 1] = q < 3 ? b_quad(a_hi, q + 1, sxs) : b_quad(a_hi + 1, 0, nsxs);
                                }
                                __builtin_amdgcn_sched_barrier(0);
                                const uint4v bqv = qb[(t >> 2) & 1];
                                const uint32_t bwv = (t & 3) == 0 ? bqv.x : (t & 3) == 1 ? bqv.y : (t & 3) == 2 ? bqv.z : bqv.w;
                                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(awin[m & 3]), __uint_as_float(bwv), acc, 0, 0, 0);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            sxs = nsxs;
                        }
                        sw >>= 16;
#pragma unroll
                        for (int t = 0; t < 16; ++t) ak[t] += 16u << 7;
                    }
                }
            } else if constexpr (SIGN_WORDS > 0) {
                // n = 12: round 2's loop (every read in its step); the pipelined form measures 1 % SLOWER there (15.65 -> 15.83 ms,
                // same box, tools/ab_r3a.sh) while n = 10 / 11 gain 6.5 / 3.7 % -- at n = 12 the other wave of the SIMD already
                // covers the read latency and the extra bookkeeping of the pipeline costs more than it hides
                // block signs from the per-launch bit masks: two vector instructions per step (bit, address bit) instead
                // of five; chunks of 16 steps (the A addresses move once per chunk)
#pragma unroll
                for (int w = 0; w < SIGN_WORDS; ++w) {
                    uint32_t sw = sgn[w];
#pragma unroll 1
                    for (int c = 0; c < 2; ++c) {
                        const int a0 = 32 * w + 16 * c;
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const int a_hi = a0 + j;
                            // (wave-uniform: kept on the scalar unit, one v_lshl_or_b32 joins it with the lane's sign bit)
                            uint32_t sx = __builtin_amdgcn_readfirstlane((uint32_t(a_hi) << 7) | (uint32_t((a_hi >> 1) & 7) << 4));
                            asm("" : "+s"(sx));   // whole, in a scalar register: otherwise its parts are OR-ed in one by one on the vector unit
                            step(a_hi, j, (((sw >> j) & 1u) << (n + 2)) | sx);
                        }
                        sw >>= 16;
#pragma unroll
                        for (int t = 0; t < 16; ++t) ak[t] += 16u << 7;
                    }
                }
            } else {
                for (int a0 = 0; a0 < H; a0 += 8) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int a_hi = a0 + j;
                        const uint32_t sx = (uint32_t(a_hi) << 7) | (uint32_t((a_hi >> 1) & 7) << 4);
                        step(a_hi, j, (block_sign(a_hi) << (n + 2)) | sx);
                    }
#pragma unroll
                    for (int t = 0; t < 16; ++t) ak[t] += 8u << 7;
                }
            }

            // ---- accumulator (row = c_lo, column = this lane's c_hi) -> graded row ----
            float* orow = p.out + (item0 + it) * p.out_stride;
            const int32_t* om = p.out_map + (c_hi << 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c_lo = (r & 3) + 8 * (r >> 2) + 4 * h;
                store_result<float>(orow, om[c_lo], acc[r], p.beta, out_scale, (c_hi << 5) + c_lo);
            }
        }
        lds_barrier<THREADS>();   // the images are rewritten by the next group
    }
}

// ------------------------------------------------------------------------------------------
// Matrix-core form in f64, the reference's value type (n = 8, 9, 10): v_mfma_f64_16x16x4_f64, image-pair form.
// With lo = 4 bits the contribution of block a_hi to the 16 result columns c_hi of a tile is a 16 x 16 x 16 product,
// four instructions of k = 4 each:
//   lane (i, kq):  A operand of instruction s (k = 4 s + kq) = +-A[a_hi][i ^ k]        (+A / -A image by a lane constant)
//                  B operand                                   = +-B[a_hi ^ c_hi(i)][k]  (+B / -B image by the block sign,
//                                                                 and by the lane constant (-1)^(|c_hi| |k|))
//   accumulator register r = row c_lo = 4 r + kq, column c_hi(i)  (the f64 instruction interleaves the row groups).
// One wave per 16 result columns, one ITEM per workgroup (n = 8: a single wave; 8 KiB of images per item, so many
// workgroups stay resident), persistent.  Every operand is one ds_read_b64; a B block's 16 doubles are stored at
// k ^ (((x >> 1) & 7) << 1) inside block x, so that the 32 lanes of a read -- 16 blocks x two k -- cover 32 different
// bank pairs; the address of a lane's word is (lane constant) ^ (step constant).  Same sums as k_gp_dense<double>: a
// k-ordered fused multiply-add chain per result component.
// ------------------------------------------------------------------------------------------
typedef double double4m __attribute__((ext_vector_type(4)));

// the two 16x16x4 instructions behind one name; their accumulator layouts differ: register r of lane group kq holds row
// 4 kq + r in f32 and row 4 r + kq in f64 (found with one-hot operands)
template <typename T>
struct Mfma16x4;
template <>
struct Mfma16x4<double> {
    typedef double4m acc_t;
    static constexpr int SHIFT = 3;
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int kq, int r) { return 4 * r + kq; }
    static constexpr bool QUAD = false;           // B operands: one ds_read_b64 per instruction
    static __device__ __forceinline__ constexpr int k_of(int kq, int s) { return 4 * s + kq; }
    static __device__ __forceinline__ double flip(double v, uint32_t bit31) { return __hiloint2double(__double2hiint(v) ^ int(bit31), __double2loint(v)); }
};
template <>
struct Mfma16x4<float> {
    typedef float4v acc_t;
    static constexpr int SHIFT = 2;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int kq, int r) { return 4 * kq + r; }
    // f32: the four B words a lane needs in a step are ONE 16-byte quad (a quarter of the LDS read instructions of the B
    // side).  Lane group kq then multiplies the k values k_of(kq, 0..3); each group's four have one |k| parity -- groups
    // 0, 1 even, 2, 3 odd -- because a quad comes from one image and the factor (-1)^(|c_hi| |k|) picks the image.
    static constexpr bool QUAD = true;
    static __device__ __forceinline__ constexpr int k_of(int kq, int s) {
        constexpr int tab[4][4] = {{0, 3, 5, 6}, {9, 10, 12, 15}, {1, 2, 4, 7}, {8, 11, 13, 14}};
        return tab[kq][s];
    }
    static __device__ __forceinline__ float flip(float v, uint32_t bit31) { return __uint_as_float(__float_as_uint(v) ^ bit31); }
};

// k_gp_mfma16x4<T, ...>: T = double is the kernel the documents call k_gp_mfma16d; T = float is the same kernel on
// v_mfma_f32_16x16x4_f32 (one item per workgroup instead of k_gp_mfma16's four items per instruction).
// MODE >= 1 (FAST): both operands hold every blade in consecutive, 16-byte aligned rows (the host checks): a thread moves
// its 16-byte pieces of every row (f64: two per row, f32: one), their map words (image address, sign) stay in registers,
// and the rows of the NEXT item are fetched into registers while the matrix cores work on the current one.  MODE 0: the
// general staging.  MODE 2: in addition every blade is produced and nothing is accumulated (beta = 0, the host checks):
// the result stores are straight-line code -- no per-register branches, (uniform row base) + (lane offset) addressing,
// and, because their number is known, the wait for the prefetched rows no longer waits for the previous item's stores
// (loads and stores share one in-order counter).
// The FAST staging does not apply the `0.0 + x` of the reference's zero-init + add_grades_from copy: it only turns -0.0
// into +0.0, and in these kernels a zero operand of either sign contributes +-0 to a sum that starts from +0.0 and is
// rounded to nearest, which leaves every sum -- also an all-zero one: (+0) + (-0) = +0 -- bit for bit what it would be
// (tests/test_gpu_dense_oracle.py::test_negative_zero_operands_leave_no_trace).
template <typename T, bool DEGENERATE, int NDIM, int MODE, bool SCALED = false, bool CHAINED = false>
__global__ __launch_bounds__(64 << (NDIM - 8)) __attribute__((amdgpu_waves_per_eu(2))) void k_gp_mfma16x4(DenseArgs<T> p) {   // (at least two waves per SIMD: 256 registers)
    constexpr bool FAST = MODE >= 1;
    static_assert(!(SCALED && MODE != 0), "a rescaled basis runs on the general staging and stores");
    const T* const left_scale = SCALED ? p.left_scale : nullptr;
    const T* const right_scale = SCALED ? p.right_scale : nullptr;
    const T* const out_scale = SCALED ? p.out_scale : nullptr;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);
    lds_u8* lds = (lds_u8*)smem_raw;
    typedef __attribute__((address_space(3))) T lds_t;
    typedef Mfma16x4<T> MM;
    constexpr int ES = MM::SHIFT;                 // log2 of the element size
    constexpr int n = NDIM;                       // 8 ... 12
    constexpr int N = 1 << n;
    constexpr int H = 1 << (n - 4);               // number of 16-blocks
    constexpr int THREADS = 64 << (n - 8);        // one wave per 16 result columns
    // LDS layout (elements), n <= 9: +B [0, N), -B [N, 2 N), +A [2 N, 3 N), 16 spare, -A [3 N + 16, 4 N + 16), 16 zeros.
    // The B pair starts at a multiple of 2 N and its halves are N apart, a power of two: the B sign is an address BIT (XOR).
    // -A sits one 16-block further than N from +A: the two 16-lane groups that share a bank cycle of an A read want the
    // same 16 words of a block, with signs of their own -- +A[w] and -A[w] N apart are two addresses in ONE bank (a 2-way
    // conflict on nearly every A read: 128 of the 420 LDS cycles an item cost at n = 8 in round 2), N + 16 apart they are
    // in different halves of the banks.  n >= 10: +A, -A, +B, -B, 16 zeros as before -- measured: the split gains 1 % at
    // n = 8, 9, nothing at n = 10, 11, and with the A pair above 64 KiB (n = 12, f64) the kernel LOSES 6 % (same box:
    // 8.23 -> 8.71 ms per 16,384 items, with or without the extra block), so the B pair keeps the upper half there and
    // rides in the lane constants (as a separate term it does not fit the 16-bit offset field of an LDS instruction).
#ifndef GAAST_MFMA16X4_PAD_MAXN
#define GAAST_MFMA16X4_PAD_MAXN 9    /* the -A image sits one block further up to this dimension (A/B switch) */
#endif
    constexpr int PAD_A = NDIM <= GAAST_MFMA16X4_PAD_MAXN ? 16 : 0;
    constexpr int A_EL = PAD_A ? 2 * N : 0, B_EL = PAD_A ? 0 : 2 * N;   // first element of the +A / +B image
    constexpr int item_stride = 4 * N + 32;
    constexpr uint32_t NEG = uint32_t(N) << ES;   // bytes from +B to -B
    constexpr uint32_t A_BASE = uint32_t(A_EL) << ES, NEG_A = uint32_t(N + PAD_A) << ES;   // bytes: +A, and from +A to -A
    constexpr uint32_t B_BASE = uint32_t(B_EL) << ES;
    constexpr int BS = 4 + ES;                    // log2 of a 16-block's bytes
    const int tid = threadIdx.x;
    if (tid < 16) smem[4 * N + PAD_A + tid] = T(0);  // the B "block" of a vanishing contribution: zero for the whole launch

    const int tile = tid >> 6, lane = tid & 63;
    const int i = lane & 15, kq = lane >> 4;
    const int c_hi = (tile << 4) | i;

    // A operand of instruction s: byte address inside the +A / -A pair, without the step's block offset
    uint32_t ak[4], bk[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int k = MM::k_of(kq, s);
        const int a_lo = i ^ k;
        int par = __builtin_popcount(uint32_t(a_lo & k) & p.neg_lo) & 1;   // lo vectors that square to -1
        for (int pp = 1; pp < 4; ++pp)
            if ((a_lo >> pp) & 1) par ^= __builtin_popcount(k & ((1 << pp) - 1)) & 1;
        ak[s] = A_BASE + (uint32_t(a_lo) << ES) + (par ? NEG_A : 0u);
        // B side: (lane constant) ^ (step constant); lanes with odd |c_hi| take the odd-|k| words from the other image
        if (MM::QUAD) {   // the lane's quad of block x: quad kq ^ (((x >> 2) & 1) << 1) (16 lanes of a b128 group: 16 bank quads)
            bk[s] = (uint32_t(c_hi) << BS) | (uint32_t(kq ^ (((c_hi >> 2) & 1) << 1)) << 4);
            if ((__builtin_popcount(uint32_t(c_hi)) & 1) && kq >= 2) bk[s] ^= NEG;
        } else {
            bk[s] = (uint32_t(c_hi) << BS) | ((uint32_t((c_hi >> 1) & 7) << (1 + ES)) ^ (uint32_t(k) << ES));
            if ((__builtin_popcount(uint32_t(c_hi)) & __builtin_popcount(uint32_t(k))) & 1) bk[s] ^= NEG;
        }
    }
    // block sign of every step, one bit per a_hi, and (DEGENERATE) the steps whose contribution to this column vanishes
    constexpr int SW = (H + 31) / 32, SB = H < 32 ? H : 32;   // words of step bits, steps per word
    uint32_t sign_bits[SW], zero_bits[SW];
#pragma unroll
    for (int w = 0; w < SW; ++w) {
        uint32_t sb = 0, zb = 0;
#pragma unroll 1
        for (int b = 0; b < SB; ++b) {
            const int a_hi = 32 * w + b;
            uint32_t sp = uint32_t(a_hi) >> 1;
            sp ^= sp >> 1;
            sp ^= sp >> 2;
            sp ^= sp >> 4;
            const uint32_t M = sp ^ (uint32_t(a_hi) & p.neg_hi);
            const uint32_t u = (__builtin_popcount(uint32_t(a_hi) & sp) ^ __builtin_popcount(uint32_t(a_hi) & p.neg_hi)) & 1u;
            sb |= ((u ^ uint32_t(__builtin_popcount(uint32_t(c_hi) & M))) & 1u) << b;
            if (DEGENERATE) zb |= ((uint32_t(a_hi) & ~uint32_t(c_hi) & p.zero_hi) ? 1u : 0u) << b;
        }
        sign_bits[w] = sb;
        zero_bits[w] = zb;
    }
    constexpr uint32_t zero_block = uint32_t(4 * N + PAD_A) << ES;
#pragma unroll
    for (int s = 0; s < 4; ++s) bk[s] |= B_BASE;   // a multiple of 2 N elements: above every bit the XOR of a step can touch
    // where this lane's results go: register r = row c_lo = MM::row(kq, r) of column c_hi
    uint32_t ooff[4], osg[4];
    bool ook[4];
    T osc[4];   // general diagonal metric: 1 / w_T of the lane's four result blades (MODE <= 1 only: the host keeps such
                // products on the general staging and the general stores)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int32_t w = p.out_map[(c_hi << 4) + MM::row(kq, r)];
        ook[r] = w >= 0;
        ooff[r] = uint32_t(w & 0x3fffffff) << ES;
        osg[r] = (uint32_t(w) & 0x40000000u) << 1;
        osc[r] = (SCALED && out_scale) ? out_scale[(c_hi << 4) + MM::row(kq, r)] : T(1);
    }

    // FAST: byte address inside the +A / +B image and negate bit of the thread's four components per row
    constexpr int CPP = 16 >> ES;                 // components per 16-byte piece
    constexpr int PPT = 4 / CPP;                  // pieces per thread and row
    static_assert((N / CPP) == PPT * THREADS, "four components of each row per thread");
    uint32_t wa[4] = {0, 0, 0, 0}, wb[4] = {0, 0, 0, 0}, sa[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
    T pf_l[4], pf_r[4];
    // CHAINED with the register-prefetch staging: the left operand is computed from a list over two other rows (stage_from_list).
    // When the list's LEFT row is the very row the dense product reads as its right operand (R in (R X) ~R) the prefetched
    // pieces also fill the list's scratch copy; the list's right row (X: n components) rides in one register per thread.
    const bool same_src = CHAINED && p.pre_left == p.right && p.pre_left_stride == p.right_stride && p.pre_left_len == N &&
                          p.pre_right_len <= THREADS;
    T pf_x = T(0);
    auto fetch = [&](int64_t item) {   // (uniform) row base + the thread's pieces
#pragma unroll
        for (int m = 0; m < PPT; ++m) {
            uint32_t o = uint32_t(tid + m * THREADS) * 16u;   // (uniform row base) + (32-bit lane offset), see the stores
            asm volatile("" : "+v"(o));
            if constexpr (!CHAINED) {
                const uint4 xl = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(p.left + item * p.left_stride) + o);
                __builtin_memcpy(&pf_l[m * CPP], &xl, 16);
            }
            const uint4 xr = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(p.right + item * p.right_stride) + o);
            __builtin_memcpy(&pf_r[m * CPP], &xr, 16);
        }
        if constexpr (CHAINED) {
            if (same_src && tid < p.pre_right_len) pf_x = p.pre_right[item * p.pre_right_stride + tid];
        }
    };
    if (FAST) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = ((e / CPP) * THREADS + tid) * CPP + (e % CPP);   // component of the row
            const uint32_t mr = p.right_map[idx];
            if constexpr (!CHAINED) {
                const uint32_t ml = p.left_map[idx];
                wa[e] = A_BASE + (((ml >> 16) & 0x7fffu) << ES);
                sa[e] = ml & 0x80000000u;
            }
            wb[e] = B_BASE + (((mr >> 16) & 0x7fffu) << ES);
            sb[e] = mr & 0x80000000u;
        }
        if (int64_t(blockIdx.x) < p.batch) fetch(blockIdx.x);
    }

    // CHAINED, register-prefetch path, the list is R X of a sandwich (four rows per thread, one term per basis vector, +-1
    // coefficients, padded to 12 by the host): the thread keeps what its rows need in registers for all its items
    // (list_eval_rows re-reads the words from L1 and decodes them per item: ~9.5 vector instructions per term, and vector
    // instructions do NOT issue beside the matrix ones on this chip -- tools/microbench/coissue*.hip, inwave.hip; round 3's
    // sand9 spent 460 of its 561 vector instructions per item here).  Kept per term: the LDS byte address of both operands;
    // the term's sign is an address too -- X sits in the scratch twice, +X and -X, and l * (-x) has the bits of (l * x) * (-1.0)
    // (eval.rs:82).  So a term is two reads, one multiplication, one addition, in the list's order: the same bits.
    //   NDIM == 8 (sand9): 2 x 36 plain addresses (228 registers with the unrolled matrix loop's);
    //   NDIM == 9 (sand10, two waves per item): both addresses in one register (l | r << 16), two more instructions per term
    // -- 80 registers more would spill.
    constexpr bool LIST_CACHE = CHAINED && FAST && (NDIM == 8 || NDIM == 9);
    constexpr bool LC_PACKED = NDIM != 8;
    constexpr int LW = NDIM + 1;                  // terms per row of R X in n = NDIM + 1 dimensions
    bool list_cached = LIST_CACHE && same_src && p.pre_width == 12 && p.pre_rows == 4 * THREADS && p.pre_row_scale == nullptr;
    uint32_t ce[(LIST_CACHE && LC_PACKED) ? 4 : 1][(LIST_CACHE && LC_PACKED) ? LW : 1];
    // (pointers, not offsets: the LDS base is a link-time constant the compiler adds per use otherwise)
    const lds_t *cl[(LIST_CACHE && !LC_PACKED) ? 4 : 1][(LIST_CACHE && !LC_PACKED) ? LW : 1], *cr[(LIST_CACHE && !LC_PACKED) ? 4 : 1][(LIST_CACHE && !LC_PACKED) ? LW : 1];
    uint32_t crow[4] = {0, 0, 0, 0};
    if constexpr (LIST_CACHE) {
        if (list_cached) {
            const uint32_t ll = uint32_t(p.pre_left_len), rl = uint32_t(p.pre_right_len);
            bool pads_only = true;   // beyond LW terms: the host's padding entries in every row
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                crow[rr] = p.pre_row_map[tid + rr * THREADS];
#pragma unroll
                for (int t = 0; t < 12; ++t) {
                    const uint32_t e = p.pre_entries[t * (4 * THREADS) + tid + rr * THREADS];
                    if (t < LW) {
                        const uint32_t la = (uint32_t(p.pre_scratch) + (e & 0x7fffu)) << ES;
                        const uint32_t ra = (uint32_t(p.pre_scratch) + ll + ((e >> 16) & 0x7fffu) + ((e >> 31) ? rl + 1u : 0u)) << ES;
                        if constexpr (LC_PACKED) {
                            ce[rr][t < LW ? t : 0] = la | (ra << 16);
                        } else {
                            cl[rr][t < LW ? t : 0] = (const lds_t*)(lds + la);
                            cr[rr][t < LW ? t : 0] = (const lds_t*)(lds + ra);
                        }
                    } else {
                        pads_only = pads_only && (e & 0x7fffu) == ll + rl;
                    }
                }
            }
            // (every LDS byte address of this kernel is below 2^16: the packed form loses nothing)
            list_cached = __all(pads_only) != 0;
        }
    }
    auto cached_list = [&]() {
        if constexpr (LIST_CACHE) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                T lv[LW], rv[LW];
#pragma unroll
                for (int t = 0; t < LW; ++t) {
                    if constexpr (LC_PACKED) {
                        uint32_t e = ce[rr][t];
                        asm volatile("" : "+v"(e));   // the word stays packed in its register
                        lv[t] = *(const lds_t*)(lds + (e & 0xffffu));
                        rv[t] = *(const lds_t*)(lds + (e >> 16));
                    } else {
                        lv[t] = *cl[rr][t];
                        rv[t] = *cr[rr][t];
                    }
                }
                T acc = T(0);                                      // the fresh cache buffer of eval.rs:21-33
#pragma unroll
                for (int t = 0; t < LW; ++t) acc = acc + lv[t] * rv[t];
                uint32_t w = crow[rr];
                asm volatile("" : "+v"(w));
                const uint32_t pos = (w >> 16) & 0x7fffu;
                const T val = list_flip<T>(acc, w & 0x80000000u);
                smem[A_EL + pos] = val;
                smem[A_EL + pos + N + PAD_A] = -val;
            }
        }
    };
    for (int64_t item = blockIdx.x; item < p.batch; item += gridDim.x) {
        // ---- both operands into their +/- images ----
        if (FAST) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                T yr = pf_r[e];
                if constexpr (CHAINED) {
                    if (same_src) {   // the list's copy of this row: raw, canonicalised like a product operand (eval.rs:27-31)
                        const int idx = ((e / CPP) * THREADS + tid) * CPP + (e % CPP);
                        smem[p.pre_scratch + idx] = p.pre_canon_left ? T(0) + yr : yr;
                    }
                } else {
                    T yl = pf_l[e];
                    if (p.left_signs) yl = MM::flip(yl, sa[e]);   // (uniform) folded sign arms / a permuted basis; the B side always
                    *(lds_t*)(lds + wa[e]) = yl;                  // carries the b_hi part of (-1)^(|a_hi| |b_lo|)
                    *(lds_t*)(lds + wa[e] + NEG_A) = -yl;
                }
                yr = MM::flip(yr, sb[e]);
                *(lds_t*)(lds + wb[e]) = yr;
                *(lds_t*)(lds + wb[e] + NEG) = -yr;
            }
            if constexpr (CHAINED) {
                if (same_src) {
                    if (tid < p.pre_right_len) {
                        const T x = p.pre_canon_right ? T(0) + pf_x : pf_x;
                        smem[p.pre_scratch + N + tid] = x;
                        smem[p.pre_scratch + N + p.pre_right_len + 1 + tid] = -x;   // the operand of the terms with coefficient -1
                    }
                    if (tid == 0) smem[p.pre_scratch + N + p.pre_right_len] = T(0);   // the zero pair of the padding entries
                } else {
                    list_fill_scratch<T, THREADS>(p, item, 1, smem + p.pre_scratch, tid);
                }
                if (!p.left_full) {   // components no row of the list produces stay zero in both A images
                    for (int e = tid; e < N; e += THREADS) {
                        smem[A_EL + e] = T(0);
                        smem[A_EL + N + PAD_A + e] = T(0);
                    }
                }
                if (THREADS > 64) __syncthreads();
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            } else {
                lds_barrier<THREADS>();
            }
            // in flight during the products below.  Unconditional (the last item re-reads itself): the loop then issues a
            // KNOWN number of loads and stores per item, and the wait at its top is a counted one
            fetch(item + gridDim.x < p.batch ? item + gridDim.x : item);
            if constexpr (CHAINED) {
                if (list_cached) cached_list();
                else list_eval_rows<T, THREADS>(p, 1, smem + A_EL, item_stride, smem + p.pre_scratch, tid, N + PAD_A);   // +A and -A at once
                lds_barrier<THREADS>();
            }
        } else {
            if (!p.left_full || !p.right_full) {
                for (int e = tid; e < 4 * N + PAD_A; e += THREADS) smem[e] = T(0);
                lds_barrier<THREADS>();
            }
            if constexpr (CHAINED)   // (its own instantiation: the plain kernels carry none of this code or its registers)
                stage_from_list<T, THREADS>(p, item, 1, smem + A_EL, item_stride, smem + p.pre_scratch, tid);
            else
                stage_operands<T, THREADS>(p.left + item * p.left_stride, p.left_stride, p.left_map, p.left_count, p.left_contig,
                                           p.canon_left, smem + A_EL, item_stride, 1, tid, left_scale);
            stage_operands<T, THREADS>(p.right + item * p.right_stride, p.right_stride, p.right_map, p.right_count,
                                       p.right_contig, p.canon_right, smem + B_EL, item_stride, 1, tid, right_scale);
            lds_barrier<THREADS>();
            for (int e = tid; e < 2 * N; e += THREADS) {   // the negated images: -B N further, -A N + PAD_A further
                if (e < N) smem[B_EL + N + e] = -smem[B_EL + e];
                else smem[A_EL + PAD_A + e] = -smem[A_EL + e - N];
            }
            lds_barrier<THREADS>();
        }

        typename MM::acc_t acc = {T(0), T(0), T(0), T(0)};
        // One step = the lane's four B words and four A words (one LDS read each; f32: the B words as one quad) and four
        // MFMAs.  The reads are SOFTWARE-PIPELINED: the operands of step a_hi + 1 are requested before the MFMAs of step
        // a_hi are issued, into a second register set, so that an MFMA never waits for a read issued right before it
        // (tools/microbench/mfma_operand_regs.hip: with counted waits both operands of EVERY instruction can come from
        // LDS at 97 % of the pure matrix rate; with a wait for the step's own reads the same stream runs at 77 %).
        auto load_ops = [&](uint32_t sxs, uint32_t zero, const uint32_t (&abase)[4], int joff, T (&av)[4], T (&bv)[4]) {
            if constexpr (MM::QUAD) {
                uint32_t addr = bk[0] ^ sxs;
                if (DEGENERATE) addr = zero ? zero_block : addr;
                const float4v q = *(__attribute__((address_space(3))) const float4v*)(lds + addr);
                bv[0] = q[0]; bv[1] = q[1]; bv[2] = q[2]; bv[3] = q[3];
#pragma unroll
                for (int s = 0; s < 4; ++s) av[s] = *(const lds_t*)(lds + abase[s] + uint32_t(joff << BS));
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    uint32_t addr = bk[s] ^ sxs;
                    if (DEGENERATE) addr = zero ? zero_block : addr;
                    bv[s] = *(const lds_t*)(lds + addr);
                    av[s] = *(const lds_t*)(lds + abase[s] + uint32_t(joff << BS));
                }
            }
        };
        auto mma_ops = [&](const T (&av)[4], const T (&bv)[4]) {
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = MM::mma(av[s], bv[s], acc);
        };
        T av[2][4], bv[2][4];
        if constexpr (SW == 1) {
            // n = 8, 9: all steps unrolled -- every address of a step is a lane constant the compiler keeps across items
            auto sx_of = [&](int a_hi) -> uint32_t {
                const uint32_t sx = (uint32_t(a_hi) << BS) | (MM::QUAD ? uint32_t(((a_hi >> 2) & 1) << 1) << 4 : uint32_t((a_hi >> 1) & 7) << (1 + ES));
                return sx | (((sign_bits[0] >> a_hi) & 1u) ? NEG : 0u);
            };
            load_ops(sx_of(0), zero_bits[0] & 1u, ak, 0, av[0], bv[0]);
#pragma unroll
            for (int a_hi = 0; a_hi < H; ++a_hi) {
                if (a_hi + 1 < H) load_ops(sx_of(a_hi + 1), (zero_bits[0] >> (a_hi + 1)) & 1u, ak, a_hi + 1, av[(a_hi + 1) & 1], bv[(a_hi + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);   // the reads of the next step stay ahead of this step's MFMAs
                mma_ops(av[a_hi & 1], bv[a_hi & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // n >= 10: 32 steps per pass of a rolled outer loop; the step bits of the current pass sit in word 0 and the
            // words rotate once per pass (register indices stay compile-time, the body stays 32 steps long whatever n is).
            // The last step of a pass requests the operands of the next pass's first step (word 1, the next 32 blocks).
            uint32_t sw[SW], zw[SW], aw_base[4];
#pragma unroll
            for (int w = 0; w < SW; ++w) {
                sw[w] = sign_bits[w];
                zw[w] = zero_bits[w];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) aw_base[s] = ak[s];
            auto sx_of = [&](int w, int j, uint32_t sbits) -> uint32_t {
                const int a_hi = 32 * w + j;
                uint32_t sx = __builtin_amdgcn_readfirstlane((uint32_t(a_hi) << BS) | (MM::QUAD ? uint32_t(((a_hi >> 2) & 1) << 1) << 4 : uint32_t((a_hi >> 1) & 7) << (1 + ES)));
                asm("" : "+s"(sx));   // whole, in a scalar register
                return sx | (((sbits >> j) & 1u) << (n + ES));   // NEG = 2^(n + ES) bytes
            };
            load_ops(sx_of(0, 0, sw[0]), zw[0] & 1u, aw_base, 0, av[0], bv[0]);
#pragma unroll 1
            for (int w = 0; w < SW; ++w) {
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    if (j + 1 < 32)
                        load_ops(sx_of(w, j + 1, sw[0]), (zw[0] >> (j + 1)) & 1u, aw_base, j + 1, av[(j + 1) & 1], bv[(j + 1) & 1]);
                    else if (w + 1 < SW)
                        load_ops(sx_of(w + 1, 0, sw[SW > 1 ? 1 : 0]), zw[SW > 1 ? 1 : 0] & 1u, aw_base, 32, av[0], bv[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    mma_ops(av[j & 1], bv[j & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int q = 0; q + 1 < SW; ++q) {
                    sw[q] = sw[q + 1];
                    zw[q] = zw[q + 1];
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) aw_base[s] += 32u << BS;
            }
        }

        // ---- results -> graded row: (uniform) row base + the lane's byte offsets ----
        unsigned char* orow = reinterpret_cast<unsigned char*>(p.out + item * p.out_stride);
        if constexpr (MODE == 2) {
            // every blade produced, nothing accumulated: four unconditional stores.  A reordering sign (permuted basis)
            // is applied as flip + canonicalisation, so that a zero result stays +0.0 as in the reference (F4)
            if (p.out_signs) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = T(0) + MM::flip(acc[r], osg[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // (uniform row base) + (32-bit lane offset): the offset is re-defined here so that its zero-extension is not
                // hoisted out of the item loop as a 64-bit lane constant, which would cost a 64-bit addition per store
                uint32_t o = ooff[r];
                asm volatile("" : "+v"(o));
                *reinterpret_cast<T*>(orow + o) = acc[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (ook[r]) {
                    T* q = reinterpret_cast<T*>(orow + ooff[r]);
                    T v = acc[r];
                    if (SCALED) v = v * osc[r];            // back from the rescaled basis
                    v = MM::flip(v, osg[r]);
                    if (osg[r] && !p.beta) v = T(0) + v;   // a zero result stays +0.0 under a negated reordering sign
                    *q = p.beta ? *q + v : v;
                }
            }
        }
        lds_barrier<THREADS>();   // the images are rewritten by the next item
    }
}

// ------------------------------------------------------------------------------------------
// k_gp_mfma7<T>: the n = 7 product (R^7-sized algebras; the Cl(7) products of parity-pure n = 8 operands: rotor composition
// and the second product of a sandwich in an 8-dimensional algebra) on the 16x16x4 matrix instructions, ONE WAVE PER ITEM,
// 16 instructions per item -- 4^7 multiply-adds, none wasted.  128 = 16 x 8 components do not fill the 16 x 16 result tile
// of an instruction as (c_lo, c_hi) does from n = 8 on; here the TOP basis vector is split over the two sides of the tile:
//     blade = (top, hi3, lo3);   tile row = (u, x): u = top bit of A's blade, x = c_lo;   tile column = (v, y): v = top bit
//     of B's blade, y = c_hi3;   step = a_hi3 (8 steps), instruction s of a step: b_lo = k_of(kq, s) (2 instructions),
// so element (u, x | v, y) of the tile is the partial sum of component c = (u ^ v, y, x) over the terms whose A blade has
// top bit u: every component is the sum of TWO tile elements -- (0, v) and (1, v ^ 1) -- which meet at the end through one
// lane exchange per result.  Signs are addresses as in the other image-pair kernels (+A, -A, +B, -B in LDS):
//     A operand (row, k, step):    R(a_lo, b_lo) + |a_lo & b_lo & NEG_lo| + u (|b_lo| + |a_hi3|)
//     B operand (k, column, step): |y| |b_lo| + R(a_hi3, b_hi3) + |a_hi3 & b_hi3 & NEG_hi3|    (the |b_hi3| |b_lo| part of
//                                  (-1)^(|a_hi3| |b_lo|) is folded into the B image by the host, as for k_gp_mfma16x4)
//     result (row, column):        u |y| + u v [top squares to -1];   u v [top is null]: the element contributes nothing
// (R = parity of the reorderings; a lane's two b_lo have one parity, so that both B words of a step come from one image:
// one ds_read_b64 (f32) / ds_read_b128 (f64) per step).  A null hi3 vector makes a B read a read of the zero words.
// LDS (elements): +B [0, 128), -B [128, 256) in the order [kq][v][b_hi3][s]: the 32 lanes that share an LDS cycle read 32
// consecutive pairs; +A at 256: u * 72 + a_hi3 * 8 + a_lo, -A 144 further: for a step's reads the four (u, sign)
// combinations are four different groups of eight banks, whatever the lanes' a_lo and signs are; 16 zeros at 544.
// MODE as in k_gp_mfma16x4: 0 general staging, 1 register prefetch of full rows (a lane moves two components of each row,
// one per load, dealt by LDS bank on the host), 2 ... and straight-line result stores.
// ------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ constexpr int mfma7_k(int kq, int s) {
    constexpr int tab[4][2] = {{0, 3}, {5, 6}, {1, 2}, {4, 7}};
    return tab[kq][s];
}

#ifndef GAAST_MFMA7_CHAIN_WAVES
#define GAAST_MFMA7_CHAIN_WAVES 4   /* the chained kernels: at most 128 registers, four waves per SIMD (A/B switch; 3: the same time) */
#endif
template <typename T, int MODE, bool SCALED = false, bool CHAINED = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CHAINED ? GAAST_MFMA7_CHAIN_WAVES : 1))) void k_gp_mfma7(DenseArgs<T> p) {
    constexpr bool FAST = MODE >= 1;
    static_assert(!(SCALED && MODE != 0), "a rescaled basis runs on the general staging and stores");
    const T* const left_scale = SCALED ? p.left_scale : nullptr;
    const T* const right_scale = SCALED ? p.right_scale : nullptr;
    const T* const out_scale = SCALED ? p.out_scale : nullptr;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);
    lds_u8* lds = (lds_u8*)smem_raw;
    typedef __attribute__((address_space(3))) T lds_t;
    typedef T vec2 __attribute__((ext_vector_type(2)));
    typedef Mfma16x4<T> MM;
    constexpr int ES = MM::SHIFT;
    constexpr int N = 128, THREADS = 64;
    constexpr int A_EL = 256, NEG_A_EL = 144, ZERO_EL = 544, item_stride = 560;
    constexpr uint32_t NEG = 128u << ES, NEG_A = uint32_t(NEG_A_EL) << ES, A_BASE = uint32_t(A_EL) << ES, ZERO_BASE = uint32_t(ZERO_EL) << ES;
    const int tid = threadIdx.x;
    if (tid < 16) smem[ZERO_EL + tid] = T(0);

    const int i = tid & 15, kq = tid >> 4;
    const int hi = i >> 3, lo3 = i & 7;   // A operand: tile row (u = hi, x = lo3); B operand and results: tile column (v = hi, y = lo3)
    const uint32_t neg_hi3 = p.neg_hi & 7u, zero_hi3 = p.zero_hi & 7u, neg_top = (p.neg_hi >> 3) & 1u, zero_top = (p.zero_hi >> 3) & 1u;

    uint32_t aa[2][2];   // A operand of instruction s, by the parity of |a_hi3|; the step adds a_hi3 * 8 elements
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int k = mfma7_k(kq, s);
        const int al = lo3 ^ k;
        const uint32_t par = uint32_t(lo_reorder_parity(al, k) ^ (__builtin_popcount(uint32_t(al & k) & p.neg_lo) & 1) ^ (hi & __builtin_popcount(k) & 1));
#pragma unroll
        for (int q = 0; q < 2; ++q) aa[s][q] = A_BASE + (uint32_t(hi * 72 + al) << ES) + ((par ^ uint32_t(hi & q)) ? NEG_A : 0u);
    }
    uint32_t ab[8];      // the lane's pair of B words of step a_hi3 (the zero words when the contribution vanishes)
    {
        const uint32_t kpar = uint32_t(__builtin_popcount(mfma7_k(kq, 0)) & 1);
#pragma unroll
        for (int ah = 0; ah < 8; ++ah) {
            const int bh = lo3 ^ ah;
            const uint32_t sg = ((uint32_t(__builtin_popcount(lo3)) & kpar) ^ uint32_t(lo_reorder_parity(ah, bh)) ^
                                 uint32_t(__builtin_popcount(uint32_t(ah & bh) & neg_hi3))) & 1u;
            ab[ah] = (uint32_t(kq * 16 + hi * 8 + bh) << (ES + 1)) + (sg ? NEG : 0u);
            if (uint32_t(ah & bh) & zero_hi3) ab[ah] = ZERO_BASE;   // (a null vector costs nothing per item: no DEGENERATE instantiations)
        }
    }
    // results: each component c = (u ^ v, y, x) is the sum of a tile element this lane keeps and one its partner sends
    //   f32: a lane's four registers are rows x = 4 (kq & 1) + r of ONE u = kq >> 1; partner = (v ^ 1, kq ^ 2): lane ^ 40;
    //        the u = 0 lane finishes r = 0, 1, the u = 1 lane r = 2, 3
    //   f64: register r is row 4 r + kq: u = r >> 1, x = 4 (r & 1) + kq; partner = (v ^ 1): lane ^ 8; every lane finishes
    //        its u = 0 registers
    constexpr bool F32 = sizeof(T) == 4;
    const int u_lane = F32 ? (kq >> 1) : 0;
    const int partner_addr = (tid ^ (F32 ? 40 : 8)) << 2;
    // sign / vanishing of the u = 1 tile elements of this lane (u = 0 elements carry neither)
    const uint32_t u1_sign = ((uint32_t(__builtin_popcount(lo3)) ^ (uint32_t(hi) & neg_top)) & 1u) << 31;
    const uint32_t u1_mask = (uint32_t(hi) & zero_top) ? 0u : ~0u;
    uint32_t ooff[2], osg[2];
    bool ook[2];
    T osc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int x = F32 ? 4 * (kq & 1) + (u_lane ? 2 + j : j) : 4 * j + kq;
        const int c = ((u_lane ^ hi) << 6) | (lo3 << 3) | x;
        const int32_t w = p.out_map[c];
        ook[j] = w >= 0;
        ooff[j] = uint32_t(w & 0x3fffffff) << ES;
        osg[j] = (uint32_t(w) & 0x40000000u) << 1;
        osc[j] = (SCALED && out_scale) ? out_scale[c] : T(1);
    }

    // FAST: every blade of both operands is loaded (the host checks): entry q = load * 64 + lane of a map is this lane's
    // component of load `load` -- row offset, image address, negate bit -- dealt by the host so that the lanes sharing an
    // LDS cycle store to different banks (plan.cpp: build_map); the rows need no alignment
    uint32_t wa[2] = {0, 0}, wb[2] = {0, 0}, sa[2] = {0, 0}, sb[2] = {0, 0}, oa[2] = {0, 0}, ob[2] = {0, 0};
    // TWO items are in flight per wave (register sets 0 / 1, the item loop is unrolled by two): with one, the 32 waves of a CU
    // keep 32 KiB of loads in flight (f32) -- measured 3.6 TB/s; an HBM access under load takes the time of ~2 items
    struct Pf {
        T l[2], r[2], x;
    } pf[2];
    pf[0].x = pf[1].x = T(0);
    const bool same_src = CHAINED && p.pre_left == p.right && p.pre_left_stride == p.right_stride && p.pre_left_len == N &&
                          p.pre_right_len <= THREADS;
    auto fetch = [&](int64_t item, Pf& f) {   // (uniform) row base + the lane's byte offsets
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if constexpr (!CHAINED) {
                uint32_t o = oa[e];
                asm volatile("" : "+v"(o));
                f.l[e] = *reinterpret_cast<const T*>(reinterpret_cast<const unsigned char*>(p.left + item * p.left_stride) + o);
            }
            uint32_t o = ob[e];
            asm volatile("" : "+v"(o));
            f.r[e] = *reinterpret_cast<const T*>(reinterpret_cast<const unsigned char*>(p.right + item * p.right_stride) + o);
        }
        if constexpr (CHAINED) {
            if (same_src && tid < p.pre_right_len) f.x = p.pre_right[item * p.pre_right_stride + tid];
        }
    };
    if (FAST) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int q = e * THREADS + tid;
            const uint32_t mr = p.right_map[q];
            if constexpr (!CHAINED) {
                const uint32_t ml = p.left_map[q];
                oa[e] = (ml & 0xffffu) << ES;
                wa[e] = A_BASE + (((ml >> 16) & 0x7fffu) << ES);
                sa[e] = ml & 0x80000000u;
            }
            ob[e] = (mr & 0xffffu) << ES;
            wb[e] = ((mr >> 16) & 0x7fffu) << ES;
            sb[e] = mr & 0x80000000u;
        }
        if (int64_t(blockIdx.x) < p.batch) {
            fetch(blockIdx.x, pf[0]);
            fetch(int64_t(blockIdx.x) + gridDim.x < p.batch ? int64_t(blockIdx.x) + gridDim.x : int64_t(blockIdx.x), pf[1]);
        }
    }

    // CHAINED, register-prefetch path: when the list is R X at n = 8 -- 128 rows of 8 terms, +-1 coefficients -- a lane keeps the
    // LDS byte addresses of both operands of its 16 terms (and the rows' image words) in registers for all its items; the sign
    // of a term selects +X or -X in the scratch, as in k_gp_mfma16x4: a term is two reads, one multiplication, one addition
    // (list_eval_rows: the words re-read from L1 and decoded per item, ~9.5 vector instructions per term, none of which
    // issues beside a matrix instruction).  Same terms, same order: the same bits.
    const bool list_cached = CHAINED && FAST && same_src && p.pre_width == 8 && p.pre_rows == 2 * THREADS && p.pre_row_scale == nullptr;
    const lds_t *cl[2][8], *cr[2][8];   // (pointers, not offsets: the LDS base is a link-time constant the compiler adds per use otherwise)
    uint32_t crow[2] = {0, 0};
    if constexpr (CHAINED && FAST) {
        if (list_cached) {
            const uint32_t ll = uint32_t(p.pre_left_len), rl = uint32_t(p.pre_right_len);
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                crow[rr] = p.pre_row_map[tid + rr * THREADS];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t e = p.pre_entries[k * (2 * THREADS) + tid + rr * THREADS];
                    cl[rr][k] = (const lds_t*)(lds + ((uint32_t(p.pre_scratch) + (e & 0x7fffu)) << ES));
                    cr[rr][k] = (const lds_t*)(lds + ((uint32_t(p.pre_scratch) + ll + ((e >> 16) & 0x7fffu) + ((e >> 31) ? rl + 1u : 0u)) << ES));
                }
            }
        }
    }
    auto cached_list = [&]() {
        if constexpr (CHAINED && FAST) {
            // (measured, sand8: both rows' reads before either sum, or all eight steps' matrix operands before the first matrix
            // instruction, change nothing or lose; the list costs 18 % of the kernel, removed entirely)
            T acc2[2] = {T(0), T(0)};
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                T lv[8], rv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    lv[k] = *cl[rr][k];
                    rv[k] = *cr[rr][k];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) acc2[rr] = acc2[rr] + lv[k] * rv[k];
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const T acc = acc2[rr];
                uint32_t w = crow[rr];
                asm volatile("" : "+v"(w));
                const uint32_t pos = (w >> 16) & 0x7fffu;
                const T val = list_flip<T>(acc, w & 0x80000000u);
                smem[A_EL + pos] = val;
                smem[A_EL + pos + NEG_A_EL] = -val;
            }
        }
    };
    auto one_item = [&](int64_t item, Pf& f) {
        // ---- both operands into their +/- images ----
        if (FAST) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                T yr = f.r[e];
                if constexpr (CHAINED) {
                    if (same_src) smem[p.pre_scratch + (ob[e] >> ES)] = p.pre_canon_left ? T(0) + yr : yr;   // the list's copy of this row
                } else {
                    T yl = f.l[e];
                    if (p.left_signs) yl = MM::flip(yl, sa[e]);
                    *(lds_t*)(lds + wa[e]) = yl;
                    *(lds_t*)(lds + wa[e] + NEG_A) = -yl;
                }
                yr = MM::flip(yr, sb[e]);
                *(lds_t*)(lds + wb[e]) = yr;
                *(lds_t*)(lds + wb[e] + NEG) = -yr;
            }
            if constexpr (CHAINED) {
                if (same_src) {
                    if (tid < p.pre_right_len) {
                        const T x = p.pre_canon_right ? T(0) + f.x : f.x;
                        smem[p.pre_scratch + N + tid] = x;
                        smem[p.pre_scratch + N + p.pre_right_len + 1 + tid] = -x;   // the operand of the terms with coefficient -1
                    }
                    if (tid == 0) smem[p.pre_scratch + N + p.pre_right_len] = T(0);   // the zero pair of the padding entries
                } else {
                    list_fill_scratch<T, THREADS>(p, item, 1, smem + p.pre_scratch, tid);
                }
                if (!p.left_full) {   // components no row of the list produces stay zero in both A images
                    for (int e = tid; e < 2 * NEG_A_EL; e += THREADS) smem[A_EL + e] = T(0);
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            } else {
                lds_barrier<THREADS>();
            }
            {   // the item after the next one, into the registers just consumed; unconditional (the last items re-read themselves):
                // a known number of loads per item
                const int64_t nn = item + 2 * int64_t(gridDim.x);
                fetch(nn < p.batch ? nn : item, f);
            }
            if constexpr (CHAINED) {
                if (list_cached) cached_list();
                else list_eval_rows<T, THREADS>(p, 1, smem + A_EL, item_stride, smem + p.pre_scratch, tid, NEG_A_EL);   // +A and -A at once
                lds_barrier<THREADS>();
            }
        } else {
            if (!p.left_full || !p.right_full) {
                for (int e = tid; e < ZERO_EL; e += THREADS) smem[e] = T(0);
                lds_barrier<THREADS>();
            }
            if constexpr (CHAINED)
                stage_from_list<T, THREADS>(p, item, 1, smem + A_EL, item_stride, smem + p.pre_scratch, tid);
            else
                stage_operands<T, THREADS>(p.left + item * p.left_stride, p.left_stride, p.left_map, p.left_count, p.left_contig,
                                           p.canon_left, smem + A_EL, item_stride, 1, tid, left_scale);
            stage_operands<T, THREADS>(p.right + item * p.right_stride, p.right_stride, p.right_map, p.right_count,
                                       p.right_contig, p.canon_right, smem, item_stride, 1, tid, right_scale);
            lds_barrier<THREADS>();
            for (int e = tid; e < N + NEG_A_EL; e += THREADS) {   // the negated images
                if (e < N) smem[N + e] = -smem[e];
                else smem[A_EL + NEG_A_EL + (e - N)] = -smem[A_EL + (e - N)];
            }
            lds_barrier<THREADS>();
        }

        typename MM::acc_t acc = {T(0), T(0), T(0), T(0)};
        // eight steps, software-pipelined like k_gp_mfma16x4's: the operands of step a_hi3 + 1 are requested before the
        // two MFMAs of step a_hi3 are issued
        auto load_ops = [&](int ah, T (&av)[2], T (&bv)[2]) {
            const vec2 q = *(const __attribute__((address_space(3))) vec2*)(lds + ab[ah]);
            bv[0] = q[0];
            bv[1] = q[1];
#pragma unroll
            for (int s = 0; s < 2; ++s) av[s] = *(const lds_t*)(lds + aa[s][__builtin_popcount(ah) & 1] + (uint32_t(ah * 8) << ES));
        };
        T av[2][2], bv[2][2];
        load_ops(0, av[0], bv[0]);
#pragma unroll
        for (int ah = 0; ah < 8; ++ah) {
            if (ah + 1 < 8) load_ops(ah + 1, av[(ah + 1) & 1], bv[(ah + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 2; ++s) acc = MM::mma(av[ah & 1][s], bv[ah & 1][s], acc);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- the two halves of every component meet ----
        T keep[2], send[2];
        auto u1_element = [&](T v) -> T {   // sign and vanishing of a u = 1 tile element, as bit operations
            if constexpr (F32) {
                return __uint_as_float((__float_as_uint(float(v)) ^ u1_sign) & u1_mask);
            } else {
                return T(__hiloint2double(int((uint32_t(__double2hiint(double(v))) ^ u1_sign) & u1_mask), int(uint32_t(__double2loint(double(v))) & u1_mask)));
            }
        };
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (F32) {   // u = 1 lanes send r = 0, 1 and finish r = 2, 3
                send[j] = u_lane ? u1_element(acc[j]) : acc[2 + j];
                keep[j] = u_lane ? u1_element(acc[2 + j]) : acc[j];
            } else {
                send[j] = u1_element(acc[2 + j]);
                keep[j] = acc[j];
            }
        }
        T res[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            T got;
            if constexpr (F32) {
                got = __int_as_float(__builtin_amdgcn_ds_bpermute(partner_addr, __float_as_int(float(send[j]))));
            } else {
                const int lo_w = __builtin_amdgcn_ds_bpermute(partner_addr, __double2loint(double(send[j])));
                const int hi_w = __builtin_amdgcn_ds_bpermute(partner_addr, __double2hiint(double(send[j])));
                got = T(__hiloint2double(hi_w, lo_w));
            }
            res[j] = keep[j] + got;
        }

        // ---- results -> graded row ----
        unsigned char* orow = reinterpret_cast<unsigned char*>(p.out + item * p.out_stride);
        if constexpr (MODE == 2) {
            if (p.out_signs) {
#pragma unroll
                for (int j = 0; j < 2; ++j) res[j] = T(0) + MM::flip(res[j], osg[j]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint32_t o = ooff[j];
                asm volatile("" : "+v"(o));
                *reinterpret_cast<T*>(orow + o) = res[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (ook[j]) {
                    T* q = reinterpret_cast<T*>(orow + ooff[j]);
                    T v = res[j];
                    if (SCALED) v = v * osc[j];
                    v = MM::flip(v, osg[j]);
                    if (osg[j] && !p.beta) v = T(0) + v;
                    *q = p.beta ? *q + v : v;
                }
            }
        }
        lds_barrier<THREADS>();   // the images are rewritten by the next item
    };
    for (int64_t item = blockIdx.x; item < p.batch; item += 2 * int64_t(gridDim.x)) {
        one_item(item, pf[0]);
        if (item + gridDim.x < p.batch) one_item(item + gridDim.x, pf[1]);
    }
}

}  // namespace gaast
