// kernels_spinor.hip.hpp -- opt-in: the geometric product through the matrix representation
// Included through kernels.hip.hpp.
#pragma once
#include "kernels_common.hip.hpp"

// build-time A/B switch: compute half of the matrix product and mirror it (real structure of the representation)
#ifndef GAAST_SPINOR_HALF
#define GAAST_SPINOR_HALF 1
#endif

namespace gaast {
#ifndef GAAST_SPINOR_NT
#define GAAST_SPINOR_NT 3   /* bit 0: nontemporal 16-byte result stores, bit 1: nontemporal 16-byte operand loads (A/B switch; both: +1-2 %) */
#endif
__device__ __forceinline__ float4 spinor_load4(const float* row, int idx) {
    if constexpr ((GAAST_SPINOR_NT & 2) != 0) {
        const float4v t = __builtin_nontemporal_load(reinterpret_cast<const float4v*>(row) + idx);
        return make_float4(t[0], t[1], t[2], t[3]);
    } else {
        return reinterpret_cast<const float4*>(row)[idx];
    }
}
__device__ __forceinline__ void spinor_store4(float* row, int idx, float a, float b, float c, float d) {
    if constexpr ((GAAST_SPINOR_NT & 1) != 0) {
        const float4v t = {a, b, c, d};
        __builtin_nontemporal_store(t, reinterpret_cast<float4v*>(row) + idx);
    } else {
        reinterpret_cast<float4*>(row)[idx] = make_float4(a, b, c, d);
    }
}

// ------------------------------------------------------------------------------------------
// OPT-IN fast path (GAAST_FLAG_SPINOR_GEMM): the geometric product of a non-degenerate algebra
// (n = 7..12) through its matrix representation -- at n = 12 21x fewer multiply-adds than the bilinear
// contraction, all of them on the matrix cores.  NOT the reference's algorithm: same result in
// exact arithmetic, different roundings (norm-wise error bound), so it is never
// selected unless the host asks for it.
//
// Cl(p,q), p+q = 2m, over the complex numbers is the algebra of 2^m x 2^m matrices.  With the
// Jordan-Wigner generators gamma_{2j} = Z..Z X_j, gamma_{2j+1} = Z..Z Y_j (times i for the vectors
// that square to -1), a blade e_S is i^k(S) X^x(S) Z^z(S), a Pauli string; S -> (x, z) is a
// bijection onto pairs of m-bit masks.  (X^x Z^z)[c^x][c] = (-1)^|c & z|, so for the multivector A
//     M_A[c ^ x][c] = sum_z (-1)^|c & z| * i^k(x,z) A_{S(x,z)}       -- a Walsh-Hadamard transform
// over z of row x of the re-indexed components; the product is C = M_A M_B (a complex GEMM = 3
// real ones, Gauss: at m = 6, 288 v_mfma_f32_32x32x2_f32 per item instead of 8192); the inverse
// transform of the skewed diagonals of C gives the components back.
//
// Common to the kernels below:
//   * the graded rows are scattered into W[x][z] (rows of 2^m + 1 words: the row-wise and the
//     XOR-skewed column-wise accesses both hit distinct banks) from registers, through 16-bit
//     table entries held in registers for the whole launch; persistent workgroups fetch the next
//     item's rows while the current one is in the transforms and the GEMM;
//   * transformed rows are written back in place, S_A[x][c] = M_A[c^x][c] and S_B[x][r] = M_B[r][r^x]
//     -- the index shift by x of B's transform is a sign (-1)^|x&z| on its input, carried by the
//     right operand's table -- so both MFMA operand gathers are conflict-free;
//   * C goes back to LDS with its diagonals indexed by row (conflict-free; the shift is a sign in
//     the result table), is transformed back by several threads per row (fold, then a short
//     transform), and the result rows are gathered in order so that the stores are coalesced.
// ------------------------------------------------------------------------------------------
struct SpinorArgs {
    const void* left;       // rows of float (f32 kernels) or double (k_gp_spinor12d); strides in elements
    const void* right;
    void* out;
    int64_t left_stride, right_stride, out_stride;
    // 4^m 16-bit entries each, indexed by ROW OFFSET (two per word); D = 2^m, LD = D + 1.
    //  operands: bit 0 = negate (folded unary signs, (-1)^u of the phase, the right operand's shift),
    //            bits [14:2] = x'*LD + z' (indices in the basis of spinor_basis.hpp), i.e. entry & 0x7ffc
    //            is the byte offset inside a plane of floats; offsets that hold nothing point at a
    //            padding word (x'*LD + z' = D)
    //  result:   bit 0 = negate, bit 1 = nothing to store, bits [15:2] = x'*LD + z'
    const uint16_t* left_map;
    const uint16_t* right_map;
    const uint16_t* out_map;
    int left_len, right_len;
    int out_full;
    int left_full, right_full;
    int has_alpha;           // alpha' = the top index bit (else 0), see spinor_basis.hpp
    int canon_left, canon_right;
    int beta;
    int64_t batch;
};

__device__ __forceinline__ float fma_x(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_x(double a, double b, double c) { return __builtin_fma(a, b, c); }

// lambda = 0 (LAMBIT < 0): q = +-p, every entry of the representation matrix is real or purely imaginary, and Y and Z
// are multiples of X = p r: ONE real product instead of three.  From X (all of k) and X1 (the k_top = 0 half), the
// signs rho / gam of the element's row / column (true = -1): what the three-product code stores as (Z - X - Y with the
// second half subtracted, X - Y).
template <typename T>
__device__ __forceinline__ void real_case_planes(T X, T X1, bool has_alpha, bool rho, bool gam, T* re, T* im) {
    if (!has_alpha) {
        *re = T(2) * X;
        *im = T(0);
    } else if (rho == gam) {
        const T d = T(2) * (T(2) * X1 - X);
        *re = rho ? -d : d;
        *im = T(0);
    } else {
        *re = T(0);
        *im = T(2) * X;
    }
}

template <int N, typename T = float>
__device__ __forceinline__ void wht(T (&v)[N]) {
#pragma unroll
    for (int hlf = 1; hlf < N; hlf <<= 1) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if ((i & hlf) == 0) {
                const T a = v[i], b = v[i | hlf];
                v[i] = a + b;
                v[i | hlf] = a - b;
            }
        }
    }
}

// The same transform on N values held as NP = N / 2 register pairs (v[2j], v[2j+1]): the butterflies of the first stage sit
// inside a pair (one v_pk_add_f32 with op_sel picking the halves), every later stage adds / subtracts whole pairs --
// N log2(N) / 2 packed instructions and no register moves; same operations in the same order as wht<N>.
template <int NP>
__device__ __forceinline__ void wht_pairs(float2v (&P)[NP]) {
#pragma unroll
    for (int j = 0; j < NP; ++j)
        asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(P[j]) : "v"(P[j]));
#pragma unroll
    for (int hh = 1; hh < NP; hh <<= 1) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            if ((j & hh) == 0) {
                const float2v a = P[j], b = P[j | hh];
                P[j] = a + b;
                P[j | hh] = a - b;
            }
        }
    }
}
__device__ __forceinline__ float2v pk_fma(float2v a, float2v b, float2v c) { return __builtin_elementwise_fma(a, b, c); }

// ------------------------------------------------------------------------------------------
// n = 11, 12, f32.  ONE real plane per operand (derivation, numpy prototype and the exhaustive check over
// signatures: tools/proto/spinor_single_plane.py; index bookkeeping: spinor_basis.hpp).
//
// The phase of a blade is i^k, k = 2u + f, and f is linear in the index pair.  In the basis chosen by
// the host f = x_5 [HAS_ALPHA] ^ z_LAMBIT, so with W[x][z] = (-1)^u A_S and What = WHT_z(W)
//     M_A[c^x][c] = E(p, sigma q),   p = What[x][c],  q = What[x][c ^ 2^LAMBIT],  sigma = (-1)^(x_5),
//     E(p, q) = ((p + q) + i (p - q)) / 2
// -- one transform per row instead of two, half the staging writes, half the LDS (33 KB), and
//     E(p,q') E(r,s') = (p s' + q' r)/2 + i (p r - q' s')/2
// is again three real products X = p r, Y = q' s', Z = (p+q')(r+s').  sigma_A sigma_B leaves a factor
// (-1)^(k_5) on the real part only: the k loop runs k_5 = 0 first and banks Z-X-Y of that half (real part =
// first half - second half = 2 bank - total; imaginary part = total of X-Y).  Component S(x,z) = (-1)^u Re(i^-f V[x][z]) reads the real or the
// imaginary plane of C per (row, z_LAMBIT): four threads per row fold ONE plane over bits LAMBIT and
// the other of {4,5} and transform the remaining 16 points.
// LAMBIT = -1: lambda = 0 (q = p, every phase of a row is the same).
// ------------------------------------------------------------------------------------------
// FAST: both operands and the result hold every blade in 16-byte aligned rows and nothing is accumulated (the host
// checks): one path through the loop, so that the wait for the prefetched rows can leave the stores in flight.
template <int LAMBIT, bool FAST>
__global__ __launch_bounds__(256, 2) void k_gp_spinor12s(SpinorArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    lds_u8* lds = (lds_u8*)smem_raw;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    constexpr int LD = 65, P0 = 64 * LD;
    // the second plane 16 words further: the inverse transform reads both planes in one instruction (lanes of a row
    // differ in the plane), 16 rows x 2 planes then cover the 32 banks
    constexpr int P = P0 + 16;
    constexpr int LAM = LAMBIT >= 0 ? (1 << LAMBIT) : 0;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;

    // Vector instructions are what this kernel pays for (they take matrix-pipe time, LDS and global accesses do not), so
    // the 16-bit table entries are expanded ONCE per launch into the form the per-item code consumes with two
    // instructions per component: bit 31 = negate, bit 30 = nothing to store (result table), low 16 bits = byte offset.
    // The offsets are LDS ADDRESSES (the planes' base and, for the right operand, its plane included), used as such.
    // Register u of a thread holds component 4 tid + (u & 3) + 1024 (u >> 2) of a row: four 16-byte pieces per row and
    // thread (a quarter of the memory instructions of a dword-per-lane split).
    uint32_t lm[16], rm[16], om[16];
    const uint32_t lds0 = uint32_t(size_t(lds));
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int e = 4 * tid + (u & 3) + 1024 * (u >> 2);
        const uint32_t el = p.left_map[e], er = p.right_map[e], eo = p.out_map[e];
        lm[u] = (el << 31) | (lds0 + (el & 0x7ffcu));
        rm[u] = (er << 31) | (lds0 + uint32_t(P * 4) + (er & 0x7ffcu));
        om[u] = (eo << 31) | ((eo & 2u) << 29) | (lds0 + (eo & 0xfffcu));
    }
    float va[16], vb[16];
    auto aligned16 = [](const void* ptr, int64_t stride) { return ((reinterpret_cast<uintptr_t>(ptr) | uintptr_t(stride * 4)) & 15u) == 0; };
    const bool rows_vec = FAST || (((p.left_len | p.right_len) & 3) == 0 && aligned16(p.left, p.left_stride) && aligned16(p.right, p.right_stride));
    const bool out_vec = FAST || (p.out_full && !p.beta && aligned16(p.out, p.out_stride));
    auto fetch = [&](int64_t item) {   // (uniform) row base + the thread's index: no per-access 64-bit arithmetic
        const float* lrow = static_cast<const float*>(p.left) + item * p.left_stride;
        const float* rrow = static_cast<const float*>(p.right) + item * p.right_stride;
        if (rows_vec) {   // rows shorter than 4096 (odd n: the subalgebra of n + 1) end on a piece boundary
#pragma unroll
            for (int u4 = 0; u4 < 4; ++u4) {
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f), y = x;
                if (FAST || p.left_len == 4096 || 4 * tid + 1024 * u4 < p.left_len) x = spinor_load4(lrow, tid + 256 * u4);
                if (FAST || p.right_len == 4096 || 4 * tid + 1024 * u4 < p.right_len) y = spinor_load4(rrow, tid + 256 * u4);
                va[4 * u4 + 0] = x.x; va[4 * u4 + 1] = x.y; va[4 * u4 + 2] = x.z; va[4 * u4 + 3] = x.w;
                vb[4 * u4 + 0] = y.x; vb[4 * u4 + 1] = y.y; vb[4 * u4 + 2] = y.z; vb[4 * u4 + 3] = y.w;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int e = 4 * tid + (u & 3) + 1024 * (u >> 2);
                va[u] = e < p.left_len ? lrow[e] : 0.f;
                vb[u] = e < p.right_len ? rrow[e] : 0.f;
            }
        }
    };
    // sigma of the tile's rows / columns and the sign of the second k half: wave-uniform
    const uint32_t rho_mask = (p.has_alpha && (wave >> 1)) ? 0x80000000u : 0u;
    const uint32_t gam_mask = (p.has_alpha && (wave & 1)) ? 0x80000000u : 0u;

    // ---- 1. graded rows -> W[x][z], one word per component ----
    // (no `0.0 + x` here: it only turns -0.0 into +0.0, and on this path, whose sums are re-ordered anyway, a zero of
    //  either sign contributes the same to every sum)
    auto stage = [&]() {
        if (!FAST && (!p.left_full || !p.right_full)) {
            for (int i = tid; i < P0 + P; i += 256) smem[i] = 0.f;
            lds_barrier<256>();
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            *(lds_u32*)size_t(lm[u] & 0xffffu) = __float_as_uint(va[u]) ^ (lm[u] & 0x80000000u);
            *(lds_u32*)size_t(rm[u] & 0xffffu) = __float_as_uint(vb[u]) ^ (rm[u] & 0x80000000u);
        }
    };
    // The loop is rotated: the rows of item i + 1 are staged at the END of item i's pass, after its result rows were
    // stored.  Loads and stores share one in-order counter (vmcnt); with the staging at the top of the loop the wait for
    // the prefetched rows sits where the first item's path (loads issued last) and the loop's path (stores issued last)
    // merge and becomes vmcnt(0): the round trip of the previous item's stores on every item.  Here the wait has one
    // path -- loads, then stores -- and leaves the stores in flight.
    int64_t item = blockIdx.x;
    if (item >= p.batch) return;
    fetch(item);
    stage();
    for (;;) {
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("" : "+v"(lm[u]), "+v"(rm[u]), "+v"(om[u]));
        lds_barrier<256>();
        const bool more = item + gridDim.x < p.batch;
        if (more) fetch(item + gridDim.x);

        // ---- 2. one transform per row, two threads per row: fold bit 5 with sign (-1)^hb, 32 points as 16 pairs ----
        {
            const int hb = tid & 1;
            float* row = smem + (tid >> 7) * P + ((tid >> 1) & 63) * LD;
            const float sg = hb ? -1.f : 1.f;
            const float2v sg2 = {sg, sg};
            float2v v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {   // sg = +-1: an exact product, one instruction
                const float2v lo = {row[2 * j], row[2 * j + 1]}, hi = {row[2 * j + 32], row[2 * j + 33]};
                v[j] = pk_fma(hi, sg2, lo);
            }
            wht_pairs<16>(v);
#pragma unroll
            for (int j = 0; j < 16; ++j) {   // the partner (adjacent lane) has read already
                row[2 * j + 32 * hb] = v[j][0];
                row[2 * j + 1 + 32 * hb] = v[j][1];
            }
        }
        lds_barrier<256>();

        if constexpr (LAMBIT >= 0 && GAAST_SPINOR_HALF) {
            // ---- 3'. HALF the product.  With lambda != 0 the representation has a real structure:
            //     M(r ^ lambda, c ^ lambda) = sigma conj(M(r, c)),  sigma = (-1)^(alpha'.(r ^ c))
            // (both entries are built from the same pair (p, q) of row x = r ^ c, swapped), products inherit it, so only
            // the 32 rows of C with r_lambda = 0 are computed -- 3 x (32 x 64 x 64) multiply-adds instead of 3 x 64^3 --
            // and every element also lands, conjugated and signed, at (r ^ lambda, c ^ lambda).  Wave w owns the 32 rows x
            // columns 16 w .. 16 w + 15 as two 16 x 16 tiles of v_mfma_f32_16x16x4_f32; k = 4 s + kq, k_5 = 0 first.
            const int i16 = lane & 15, kq = lane >> 4;
            auto half_row = [](int hr) { return LAMBIT == 5 ? hr : (((hr & 16) << 1) | (hr & 15)); };   // rows with the lambda bit clear
            float4v gx[2], gy[2], gz[2], bank_re[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    gx[t][r] = 0.f;
                    gy[t][r] = 0.f;
                    gz[t][r] = 0.f;
                    bank_re[t][r] = 0.f;
                }
            {
                const float* A = smem;
                const float* B = smem + P;
                const uint32_t cb = uint32_t(16 * wave + i16);
                // sigma of the column (bit 5 of c, wave-uniform): q_B' = gsig q_B.  Applied as a factor of the sum p + q'
                // (one fused multiply-add, exact) and, for Y = sum q_A' q_B', once to the finished accumulator below.
                const float gsig = (p.has_alpha && (wave >> 1)) ? -1.f : 1.f;
#pragma unroll
                for (int s4 = 0; s4 < 16; ++s4) {
                    const uint32_t k = uint32_t(4 * s4 + kq);
                    const bool hi = ((4 * s4) & LAM) != 0;                               // k has the lambda bit: partner below
                    const uint32_t ib = (cb ^ k) * LD + k;
                    const float pb = B[ib];
                    const float qb = hi ? B[ib - LAM] : B[ib + LAM];
                    const float sb = fma_x(qb, gsig, pb);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const uint32_t ra = uint32_t(half_row(16 * t + i16));
                        const uint32_t rho = (p.has_alpha && (ra >> 5)) ? 0x80000000u : 0u;
                        const uint32_t ia = (ra ^ k) * LD + k;
                        const float pa = A[ia];
                        float qa = hi ? A[ia - LAM] : A[ia + LAM];
                        qa = __uint_as_float(__float_as_uint(qa) ^ rho);
                        gx[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, pb, gx[t], 0, 0, 0);
                        gy[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa, qb, gy[t], 0, 0, 0);
                        gz[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa + qa, sb, gz[t], 0, 0, 0);
                    }
                    if (s4 == 7) {   // real part of the k_5 = 0 half; the accumulators keep running
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) bank_re[t][r] = fma_x(-gsig, gy[t][r], gz[t][r] - gx[t][r]);
                    }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) gy[t][r] *= gsig;
            }
            lds_barrier<256>();  // every wave is done reading the operand planes
            // ---- 4'. C and its mirror image -> LDS, diagonals indexed by row: S[r ^ c][r] = C[r][c] ----
            {
                const int c = 16 * wave + i16;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rr = half_row(16 * t + 4 * kq + r);      // accumulator layout: row 4 kq + r, column lane & 15
                        const float re_all = gz[t][r] - gx[t][r] - gy[t][r];
                        const float re = p.has_alpha ? 2.f * bank_re[t][r] - re_all : re_all;   // first - second half
                        const float im = gx[t][r] - gy[t][r];
                        const int x = rr ^ c;
                        const uint32_t sg = (p.has_alpha && (x >> 5)) ? 0x80000000u : 0u;        // sigma
                        float* q = smem + x * LD + rr;                     // element (rr, c); (rr ^ lambda, c ^ lambda) shares the row
                        q[0] = re;
                        q[P] = im;
                        q[LAM] = __uint_as_float(__float_as_uint(re) ^ sg);
                        q[P + LAM] = __uint_as_float(__float_as_uint(im) ^ sg ^ 0x80000000u);
                    }
            }
            lds_barrier<256>();
        } else {
        // ---- 3. the product: X = p r, Y = q' s', Z = (p+q')(r+s'), k_5 = 0 first ----
        const int i = lane & 31, h = lane >> 5;
        const int r0 = (wave >> 1) << 5, c0 = (wave & 1) << 5;
        float16v gx, gy, gz, bank_re;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            gx[r] = 0.f;
            gy[r] = 0.f;
            gz[r] = 0.f;
        }
        {
            // p and q sit in one row, columns k and k ^ 2^LAMBIT (one ds_read2_b32; which of the pair is p
            // is a compile-time property of the step).  The loop is left to the compiler's scheduler:
            // a hand-scheduled, software-pipelined form measured 3 % slower.
            const float* A = smem;
            const float* B = smem + P;
            const uint32_t ra = uint32_t(r0 + i), cb = uint32_t(c0 + i);
#pragma unroll
            for (int s2 = 0; s2 < 32; ++s2) {
                const uint32_t k = uint32_t(2 * s2 + h);
                const uint32_t ia = (ra ^ k) * LD + k, ib = (cb ^ k) * LD + k;
                const bool hi = LAMBIT >= 0 && ((2 * s2) & LAM);          // k has the lambda bit: partner below
                const float pa = A[ia], pb = B[ib];
                float qa = LAMBIT < 0 ? pa : (hi ? A[ia - LAM] : A[ia + LAM]);
                float qb = LAMBIT < 0 ? pb : (hi ? B[ib - LAM] : B[ib + LAM]);
                qa = __uint_as_float(__float_as_uint(qa) ^ rho_mask);
                qb = __uint_as_float(__float_as_uint(qb) ^ gam_mask);
                gx = __builtin_amdgcn_mfma_f32_32x32x2f32(pa, pb, gx, 0, 0, 0);
                if constexpr (LAMBIT >= 0) {
                    gy = __builtin_amdgcn_mfma_f32_32x32x2f32(qa, qb, gy, 0, 0, 0);
                    gz = __builtin_amdgcn_mfma_f32_32x32x2f32(pa + qa, pb + qb, gz, 0, 0, 0);
                }
                if (s2 == 15) {   // real part of the k_5 = 0 half; the accumulators keep running
#pragma unroll
                    for (int r = 0; r < 16; ++r) bank_re[r] = LAMBIT >= 0 ? gz[r] - gx[r] - gy[r] : gx[r];
                }
            }
        }
        lds_barrier<256>();  // every wave is done reading the operand planes

        // ---- 4. C -> LDS, diagonals indexed by row: S[r ^ c][r] = C[r][c]; plane 0 real, plane 1 imaginary ----
        {
            const int c = c0 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = r0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if constexpr (LAMBIT < 0) {
                    float re, im;
                    real_case_planes<float>(gx[r], bank_re[r], p.has_alpha, rho_mask != 0, gam_mask != 0, &re, &im);
                    smem[(rr ^ c) * LD + rr] = re;
                    smem[P + (rr ^ c) * LD + rr] = im;
                } else {
                    const float re_all = gz[r] - gx[r] - gy[r];            // both halves added
                    smem[(rr ^ c) * LD + rr] = p.has_alpha ? 2.f * bank_re[r] - re_all : re_all;   // first - second
                    smem[P + (rr ^ c) * LD + rr] = gx[r] - gy[r];
                }
            }
        }
        lds_barrier<256>();
        }
        // ---- 5. four threads per row: fold the plane that holds this quarter's components ----
        {
            constexpr int B1 = LAMBIT == 4 ? 4 : 5, B2 = LAMBIT == 4 ? 5 : 4;   // first fold on the lambda bit
            const int x = tid >> 2, h1 = (tid >> 1) & 1, h2 = tid & 1;
            const int xi = p.has_alpha ? (x >> 5) & 1 : 0;
            const int f = LAMBIT >= 0 ? (xi ^ h1) : xi;
            const float* q = smem + f * P + x * LD;
            const float sc = 1.0f / 128.0f;                // 2^-6 of the transform, 1/2 of E E
            const float s1 = h1 ? -sc : sc, s2f = h2 ? -1.f : 1.f;
            const float2v sc2 = {sc, sc}, s12 = {s1, s1}, s22 = {s2f, s2f};
            float2v v[8];
#pragma unroll
            for (int jp = 0; jp < 8; ++jp) {
                // sc, s1 = +-2^-7, s2f = +-1: every product is exact, the fused forms round like the unfused ones
                const int j = 2 * jp;
                const float2v q00 = {q[j], q[j + 1]}, q10 = {q[j | (1 << B1)], q[(j | (1 << B1)) + 1]};
                const float2v q01 = {q[j | (1 << B2)], q[(j | (1 << B2)) + 1]};
                const float2v q11 = {q[j | (1 << B1) | (1 << B2)], q[(j | (1 << B1) | (1 << B2)) + 1]};
                const float2v lo = pk_fma(q10, s12, q00 * sc2);
                const float2v hi = pk_fma(q11, s12, q01 * sc2);
                v[jp] = pk_fma(hi, s22, lo);
            }
            wht_pairs<8>(v);
            float* o = smem + x * LD + (h1 << B1) + (h2 << B2);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int jp = 0; jp < 8; ++jp) {               // the four threads of a row are adjacent lanes
                o[2 * jp] = v[jp][0];
                o[2 * jp + 1] = v[jp][1];
            }
        }
        lds_barrier<256>();
        {
            float* orow = static_cast<float*>(p.out) + item * p.out_stride;   // uniform base + the thread's index
            if (out_vec) {
#pragma unroll
                for (int u4 = 0; u4 < 4; ++u4) {
                    uint32_t w[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) w[c] = *(const lds_u32*)size_t(om[4 * u4 + c] & 0x3fffffu) ^ (om[4 * u4 + c] & 0x80000000u);
                    spinor_store4(orow, tid + 256 * u4, __uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3]));
                }
            } else {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const uint32_t eo = om[u];
                    const int e = 4 * tid + (u & 3) + 1024 * (u >> 2);
                    const float val = __uint_as_float(*(const lds_u32*)size_t(eo & 0x3fffffu) ^ (eo & 0x80000000u));
                    if (!(eo & 0x40000000u)) orow[e] = p.beta ? orow[e] + val : val;
                }
            }
        }
        lds_barrier<256>();   // every wave is done with the planes
        if (!more) break;
        item += gridDim.x;
        stage();
    }
}

// ------------------------------------------------------------------------------------------
// The one-plane algorithm in f64 (the reference's value type), n = 11, 12: same tables, same phases,
// planes of doubles (66.5 KB of LDS, two workgroups per CU), v_mfma_f64_16x16x4_f64.  A wave owns a
// 32 x 32 tile of C as 2 x 2 MFMA tiles: per step of 4 k values, 4 ds_read2_b64 (p and q of two row
// blocks of A and two column blocks of B) feed 12 MFMAs (X, Y, Z of the four tiles).  The next item's
// rows are fetched after the accumulators have been written back (their registers are free then).
// Norm-wise error bound as for f32 with eps = 2^-52.
// ------------------------------------------------------------------------------------------
typedef double double4v __attribute__((ext_vector_type(4)));

template <int LAMBIT>
__global__ __launch_bounds__(256, 2) void k_gp_spinor12d(SpinorArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* smem = reinterpret_cast<double*>(smem_raw);
    constexpr int LD = 65, P = 64 * LD;
    constexpr int LAM = LAMBIT >= 0 ? (1 << LAMBIT) : 0;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const double* left = static_cast<const double*>(p.left);
    const double* right = static_cast<const double*>(p.right);
    double* outp = static_cast<double*>(p.out);

    // table entries expanded once per launch (as in k_gp_spinor12s): bit 31 = negate, bit 30 = nothing to store,
    // low bits = LDS address of the double
    typedef __attribute__((address_space(3))) double lds_f64;
    uint32_t lm[16], rm[16], om[16];
    const uint32_t lds0 = uint32_t(size_t((lds_u8*)smem_raw));
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const uint32_t el = p.left_map[tid + 256 * u], er = p.right_map[tid + 256 * u], eo = p.out_map[tid + 256 * u];
        lm[u] = (el << 31) | (lds0 + ((el & 0x7ffcu) << 1));
        rm[u] = (er << 31) | (lds0 + uint32_t(P * 8) + ((er & 0x7ffcu) << 1));
        om[u] = (eo << 31) | ((eo & 2u) << 29) | (lds0 + ((eo & 0xfffcu) << 1));
    }
    auto flip = [](double v, uint32_t sign_bit31) {     // sign flip through the high word
        return __hiloint2double(__double2hiint(v) ^ int(sign_bit31), __double2loint(v));
    };
    double va[16], vb[16];
    auto fetch = [&](int64_t item) {
        const double* lrow = left + item * p.left_stride;    // (uniform) row base + the thread's index
        const double* rrow = right + item * p.right_stride;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = tid + 256 * u;
            va[u] = e < p.left_len ? lrow[e] : 0.0;
            vb[u] = e < p.right_len ? rrow[e] : 0.0;
        }
    };
    int64_t item = blockIdx.x;
    if (item < p.batch) fetch(item);
    const uint32_t rho_mask = (p.has_alpha && (wave >> 1)) ? 0x80000000u : 0u;
    const uint32_t gam_mask = (p.has_alpha && (wave & 1)) ? 0x80000000u : 0u;

    for (; item < p.batch; item += gridDim.x) {
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("" : "+v"(lm[u]), "+v"(rm[u]), "+v"(om[u]));
        // ---- 1. graded rows -> W[x][z] ----  (no `0.0 + x`: a zero of either sign contributes the same to every sum)
        if (!p.left_full || !p.right_full) {
            for (int i = tid; i < 2 * P; i += 256) smem[i] = 0.0;
            lds_barrier<256>();
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            *(lds_f64*)size_t(lm[u] & 0x3fffffu) = flip(va[u], lm[u] & 0x80000000u);
            *(lds_f64*)size_t(rm[u] & 0x3fffffu) = flip(vb[u], rm[u] & 0x80000000u);
        }
        lds_barrier<256>();

        // ---- 2. one transform per row, two threads per row ----
        {
            const int hb = tid & 1;
            double* row = smem + (tid >> 7) * P + ((tid >> 1) & 63) * LD;
            const double sg = hb ? -1.0 : 1.0;
            double v[32];
#pragma unroll
            for (int c = 0; c < 32; ++c) v[c] = fma_x(row[c + 32], sg, row[c]);   // sg = +-1: an exact product, one instruction
#pragma unroll
            for (int hlf = 1; hlf < 32; hlf <<= 1) {
#pragma unroll
                for (int i2 = 0; i2 < 32; ++i2) {
                    if ((i2 & hlf) == 0) {
                        const double a = v[i2], b = v[i2 | hlf];
                        v[i2] = a + b;
                        v[i2 | hlf] = a - b;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 32; ++c) row[c + 32 * hb] = v[c];
        }
        lds_barrier<256>();

        if constexpr (LAMBIT >= 0 && GAAST_SPINOR_HALF) {
            // ---- 3'. half the product (rows with the lambda bit clear), mirrored: see k_gp_spinor12s.  Wave w owns the 32
            // rows x columns 16 w .. 16 w + 15 as two 16 x 16 tiles: 6 instead of 12 MFMAs per step of 4 k values ----
            const int i = lane & 15, kq = lane >> 4;
            auto half_row = [](int hr) { return LAMBIT == 5 ? hr : (((hr & 16) << 1) | (hr & 15)); };
            double4v gx[2], gy[2], gz[2], bank_re[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    gx[t][r] = 0.0;
                    gy[t][r] = 0.0;
                    gz[t][r] = 0.0;
                    bank_re[t][r] = 0.0;
                }
            {
                const double* A = smem;
                const double* B = smem + P;
                const uint32_t cb = uint32_t(16 * wave + i);
                const uint32_t gam = (p.has_alpha && (wave >> 1)) ? 0x80000000u : 0u;
                uint32_t kk = uint32_t(kq);
                asm volatile("" : "+v"(kk));   // addresses are computed per step, not hoisted out of the item loop
#pragma unroll
                for (int s4 = 0; s4 < 16; ++s4) {
                    const uint32_t k = kk;
                    kk += 4u;
                    const bool hi = ((4 * s4) & LAM) != 0;
                    const uint32_t ib = (cb ^ k) * LD + k;
                    const double pb = B[ib];
                    const double qb = flip(hi ? B[ib - LAM] : B[ib + LAM], gam);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const uint32_t ra = uint32_t(half_row(16 * t + i));
                        const uint32_t ia = (ra ^ k) * LD + k;
                        const double pa = A[ia];
                        const double qa = flip(hi ? A[ia - LAM] : A[ia + LAM], (p.has_alpha && (ra >> 5)) ? 0x80000000u : 0u);
                        gx[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, pb, gx[t], 0, 0, 0);
                        gy[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa, qb, gy[t], 0, 0, 0);
                        gz[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa + qa, pb + qb, gz[t], 0, 0, 0);
                    }
                    if (s4 == 7) {   // real part of the k_5 = 0 half
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) bank_re[t][r] = gz[t][r] - gx[t][r] - gy[t][r];
                    }
                }
            }
            lds_barrier<256>();
            // ---- 4'. C and its mirror image -> LDS; accumulator layout: col = lane & 15, row = (lane >> 4) + 4 r ----
            {
                const int c = 16 * wave + i;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rr = half_row(16 * t + kq + 4 * r);
                        const double re_all = gz[t][r] - gx[t][r] - gy[t][r];
                        const double re = p.has_alpha ? 2.0 * bank_re[t][r] - re_all : re_all;
                        const double im = gx[t][r] - gy[t][r];
                        const int x = rr ^ c;
                        const uint32_t sg = (p.has_alpha && (x >> 5)) ? 0x80000000u : 0u;
                        double* q = smem + x * LD + rr;
                        q[0] = re;
                        q[P] = im;
                        q[LAM] = flip(re, sg);
                        q[P + LAM] = flip(im, sg ^ 0x80000000u);
                    }
            }
        } else {
        // ---- 3. the product on v_mfma_f64_16x16x4_f64: tiles (rb, cb), X = p r, Y = q' s', Z = (p+q')(r+s') ----
        const int i = lane & 15, kq = lane >> 4;
        const int r0 = (wave >> 1) << 5, c0 = (wave & 1) << 5;
        double4v gx[4], gy[4], gz[4], bank_re[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                gx[t][r] = 0.0;
                gy[t][r] = 0.0;
                gz[t][r] = 0.0;
            }
        {
            const double* A = smem;
            const double* B = smem + P;
            uint32_t kk = uint32_t(kq);
            asm volatile("" : "+v"(kk));   // addresses are computed per step, not hoisted out of the item loop
#pragma unroll
            for (int s4 = 0; s4 < 16; ++s4) {
                const uint32_t k = kk;
                kk += 4u;
                const bool hi = LAMBIT >= 0 && ((4 * s4) & LAM);
                double pa[2], qa[2], pb[2], qb[2];
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const uint32_t ra = uint32_t(r0 + 16 * blk + i), cb = uint32_t(c0 + 16 * blk + i);
                    const uint32_t ia = (ra ^ k) * LD + k, ib = (cb ^ k) * LD + k;
                    pa[blk] = A[ia];
                    pb[blk] = B[ib];
                    qa[blk] = LAMBIT < 0 ? pa[blk] : (hi ? A[ia - LAM] : A[ia + LAM]);
                    qb[blk] = LAMBIT < 0 ? pb[blk] : (hi ? B[ib - LAM] : B[ib + LAM]);
                    qa[blk] = flip(qa[blk], rho_mask);
                    qb[blk] = flip(qb[blk], gam_mask);
                }
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int cbk = 0; cbk < 2; ++cbk) {
                        const int t = rb * 2 + cbk;
                        gx[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[rb], pb[cbk], gx[t], 0, 0, 0);
                        if constexpr (LAMBIT >= 0) {
                            gy[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[rb], qb[cbk], gy[t], 0, 0, 0);
                            gz[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[rb] + qa[rb], pb[cbk] + qb[cbk], gz[t], 0, 0, 0);
                        }
                    }
                if (s4 == 7) {   // real part of the k_5 = 0 half
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) bank_re[t][r] = LAMBIT >= 0 ? gz[t][r] - gx[t][r] - gy[t][r] : gx[t][r];
                }
            }
        }
        lds_barrier<256>();

        // ---- 4. C -> LDS, diagonals indexed by row; accumulator layout: col = lane & 15, row = (lane >> 4) + 4 r ----
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cbk = 0; cbk < 2; ++cbk) {
                const int t = rb * 2 + cbk;
                const int c = c0 + 16 * cbk + i;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = r0 + 16 * rb + kq + 4 * r;
                    if constexpr (LAMBIT < 0) {
                        double re, im;
                        real_case_planes<double>(gx[t][r], bank_re[t][r], p.has_alpha, rho_mask != 0, gam_mask != 0, &re, &im);
                        smem[(rr ^ c) * LD + rr] = re;
                        smem[P + (rr ^ c) * LD + rr] = im;
                    } else {
                        const double re_all = gz[t][r] - gx[t][r] - gy[t][r];
                        smem[(rr ^ c) * LD + rr] = p.has_alpha ? 2.0 * bank_re[t][r] - re_all : re_all;
                        smem[P + (rr ^ c) * LD + rr] = gx[t][r] - gy[t][r];
                    }
                }
            }
        }
        if (item + gridDim.x < p.batch) fetch(item + gridDim.x);   // the accumulators are dead: room for the rows
        lds_barrier<256>();
        // ---- 5. four threads per row fold the plane that holds this quarter's components ----
        {
            constexpr int B1 = LAMBIT == 4 ? 4 : 5, B2 = LAMBIT == 4 ? 5 : 4;
            const int x = tid >> 2, h1 = (tid >> 1) & 1, h2 = tid & 1;
            const int xi = p.has_alpha ? (x >> 5) & 1 : 0;
            const int f = LAMBIT >= 0 ? (xi ^ h1) : xi;
            const double* q = smem + f * P + x * LD;
            const double sc = 1.0 / 128.0;
            const double s1 = h1 ? -sc : sc, s2f = h2 ? -1.0 : 1.0;
            double v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                // sc, s1 = +-2^-7, s2f = +-1: every product is exact, the fused forms round like the unfused ones
                const double lo = fma_x(q[j | (1 << B1)], s1, q[j] * sc);
                const double hi = fma_x(q[j | (1 << B1) | (1 << B2)], s1, q[j | (1 << B2)] * sc);
                v[j] = fma_x(hi, s2f, lo);
            }
#pragma unroll
            for (int hlf = 1; hlf < 16; hlf <<= 1) {
#pragma unroll
                for (int i2 = 0; i2 < 16; ++i2) {
                    if ((i2 & hlf) == 0) {
                        const double a = v[i2], b = v[i2 | hlf];
                        v[i2] = a + b;
                        v[i2 | hlf] = a - b;
                    }
                }
            }
            double* o = smem + x * LD + (h1 << B1) + (h2 << B2);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 16; ++j) o[j] = v[j];
        }
        lds_barrier<256>();
        {
            double* orow = outp + item * p.out_stride;   // (uniform) row base + the thread's index
            if (p.out_full && !p.beta) {
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    orow[tid + 256 * u] = flip(*(const lds_f64*)size_t(om[u] & 0x3fffffu), om[u] & 0x80000000u);
            } else {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const uint32_t eo = om[u];
                    const double val = flip(*(const lds_f64*)size_t(eo & 0x3fffffu), eo & 0x80000000u);
                    if (!(eo & 0x40000000u)) orow[tid + 256 * u] = p.beta ? orow[tid + 256 * u] + val : val;
                }
            }
        }
        lds_barrier<256>();
    }
}

// The same one-plane algorithm for the smaller even dimensions: n = 8 (16 x 16 matrices,
// v_mfma_f32_16x16x4_f32) and n = 10 (32 x 32, v_mfma_f32_32x32x2_f32); odd n runs as the subalgebra of
// n + 1.  One WAVE per item (64-thread workgroups, persistent), so the phases need no cross-wave
// barrier, and the two planes of an item are 2.2 KB / 8.4 KB of LDS.  LAMBIT is M-1, M-2 or -1
// (spinor_basis.hpp); alpha', when present, is the top bit M-1, so the sign of q / s is a lane
// constant (top bit of the lane's row / column) and the k loop banks the real part after its first
// half.  (A variant with separate real and imaginary planes, two transforms per row, measured 7 %
// slower at n = 8 and is not kept.)
template <int M, int LAMBIT>
__global__ __launch_bounds__(64) void k_gp_spinor_wave1(SpinorArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    constexpr int D = 1 << M, LD = D + 1, P = D * LD;
    constexpr int NE = D * D, EPL = NE / 64;
    constexpr int LAM = LAMBIT >= 0 ? (1 << LAMBIT) : 0;
    constexpr int FWD_PASSES = 4 * D / 64, INV_PASSES = 4 * D / 64;
    using acc_t = typename std::conditional<M == 4, float4v, float16v>::type;
    constexpr int NACC = M == 4 ? 4 : 16;
    constexpr int KSTEP = 64 / D, NSTEPS = D / KSTEP;
    const int lane = threadIdx.x;
    const float* left = static_cast<const float*>(p.left);
    const float* right = static_cast<const float*>(p.right);
    float* outp = static_cast<float*>(p.out);

    // As in k_gp_spinor12s: the 16-bit table entries expanded once per launch (bit 31 = negate, bit 30 = nothing to store,
    // low bits = LDS address), register u of a lane = component 4 lane + (u & 3) + 256 (u >> 2): 16-byte pieces.
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    uint32_t lm[EPL], rm[EPL], om[EPL];
    const uint32_t lds0 = uint32_t(size_t((lds_u8*)smem_raw));
#pragma unroll
    for (int u = 0; u < EPL; ++u) {
        const int e = 4 * lane + (u & 3) + 256 * (u >> 2);
        const uint32_t el = p.left_map[e], er = p.right_map[e], eo = p.out_map[e];
        lm[u] = (el << 31) | (lds0 + (el & 0x7ffcu));
        rm[u] = (er << 31) | (lds0 + uint32_t(P * 4) + (er & 0x7ffcu));
        om[u] = (eo << 31) | ((eo & 2u) << 29) | (lds0 + (eo & 0xfffcu));
    }
    float va[EPL], vb[EPL];
    auto aligned16 = [](const void* ptr, int64_t stride) { return ((reinterpret_cast<uintptr_t>(ptr) | uintptr_t(stride * 4)) & 15u) == 0; };
    const bool rows_vec = ((p.left_len | p.right_len) & 3) == 0 && aligned16(p.left, p.left_stride) && aligned16(p.right, p.right_stride);
    const bool out_vec = p.out_full && !p.beta && aligned16(p.out, p.out_stride);
    auto fetch = [&](int64_t item) {   // (uniform) row base + the lane's index
        const float* lrow = left + item * p.left_stride;
        const float* rrow = right + item * p.right_stride;
        if (rows_vec) {   // shorter rows (odd n: the subalgebra of n + 1) end on a piece boundary
#pragma unroll
            for (int u4 = 0; u4 < EPL / 4; ++u4) {
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f), y = x;
                if (p.left_len == NE || 4 * lane + 256 * u4 < p.left_len) x = spinor_load4(lrow, lane + 64 * u4);
                if (p.right_len == NE || 4 * lane + 256 * u4 < p.right_len) y = spinor_load4(rrow, lane + 64 * u4);
                va[4 * u4 + 0] = x.x; va[4 * u4 + 1] = x.y; va[4 * u4 + 2] = x.z; va[4 * u4 + 3] = x.w;
                vb[4 * u4 + 0] = y.x; vb[4 * u4 + 1] = y.y; vb[4 * u4 + 2] = y.z; vb[4 * u4 + 3] = y.w;
            }
        } else {
#pragma unroll
            for (int u = 0; u < EPL; ++u) {
                const int e = 4 * lane + (u & 3) + 256 * (u >> 2);
                va[u] = e < p.left_len ? lrow[e] : 0.f;
                vb[u] = e < p.right_len ? rrow[e] : 0.f;
            }
        }
    };
    int64_t item = blockIdx.x;
    if (item < p.batch) fetch(item);
    const int i = lane & (D - 1);
    const int kq = lane >> M;
    const uint32_t sig_mask = (p.has_alpha && (i >> (M - 1))) ? 0x80000000u : 0u;   // sigma of row i / column i

    for (; item < p.batch; item += gridDim.x) {
#pragma unroll
        for (int u = 0; u < EPL; ++u) asm volatile("" : "+v"(lm[u]), "+v"(rm[u]), "+v"(om[u]));
        if (!p.left_full || !p.right_full) {
            for (int j = lane; j < 2 * P; j += 64) smem[j] = 0.f;
            lds_barrier<64>();
        }
        // (no `0.0 + x`: on this path a zero of either sign contributes the same to every sum)
#pragma unroll
        for (int u = 0; u < EPL; ++u) {
            *(lds_u32*)size_t(lm[u] & 0xffffu) = __float_as_uint(va[u]) ^ (lm[u] & 0x80000000u);
            *(lds_u32*)size_t(rm[u] & 0xffffu) = __float_as_uint(vb[u]) ^ (rm[u] & 0x80000000u);
        }
        lds_barrier<64>();
        if (item + gridDim.x < p.batch) fetch(item + gridDim.x);

        // one transform per row, two threads per row: (operand, x, half) = 4 D half-rows over 64 lanes
#pragma unroll
        for (int j = 0; j < FWD_PASSES; ++j) {
            const int hidx = lane + 64 * j;
            const int hb = hidx & 1;
            float* row = smem + (hidx >> (M + 1)) * P + ((hidx >> 1) & (D - 1)) * LD;
            const float sg = hb ? -1.f : 1.f;
            const float2v sg2 = {sg, sg};
            float2v v[D / 4];
#pragma unroll
            for (int c = 0; c < D / 4; ++c) {   // sg = +-1: an exact product, one instruction
                const float2v lo = {row[2 * c], row[2 * c + 1]}, up = {row[2 * c + D / 2], row[2 * c + 1 + D / 2]};
                v[c] = pk_fma(up, sg2, lo);
            }
            wht_pairs<D / 4>(v);
#pragma unroll
            for (int c = 0; c < D / 4; ++c) {
                row[2 * c + (D / 2) * hb] = v[c][0];
                row[2 * c + 1 + (D / 2) * hb] = v[c][1];
            }
        }
        lds_barrier<64>();

        if constexpr (M == 5 && LAMBIT >= 0 && GAAST_SPINOR_HALF) {
            // half the product (the 16 rows with the lambda bit clear), mirrored: see k_gp_spinor12s.  Two 16 x 16 column
            // tiles of v_mfma_f32_16x16x4_f32; k = 4 s + kq, the k_top = 0 half first.
            const int i16 = lane & 15, kq4 = lane >> 4;
            auto half_row = [](int hr) { return LAMBIT == M - 1 ? hr : (((hr & 8) << 1) | (hr & 7)); };
            float4v hx[2], hy[2], hz[2], hbank[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    hx[t][r] = 0.f;
                    hy[t][r] = 0.f;
                    hz[t][r] = 0.f;
                    hbank[t][r] = 0.f;
                }
            const uint32_t ra = uint32_t(half_row(i16));
            const uint32_t rho = (p.has_alpha && (ra >> (M - 1))) ? 0x80000000u : 0u;
#pragma unroll
            for (int s4 = 0; s4 < D / 4; ++s4) {
                const uint32_t k = uint32_t(4 * s4 + kq4);
                const bool hi = ((4 * s4) & LAM) != 0;
                const uint32_t ia = (ra ^ k) * LD + k;
                const float pa = smem[ia];
                float qa = hi ? smem[ia - LAM] : smem[ia + LAM];
                qa = __uint_as_float(__float_as_uint(qa) ^ rho);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const uint32_t cb = uint32_t(16 * t + i16);
                    const uint32_t ib = (cb ^ k) * LD + k;
                    const float pb = smem[P + ib];
                    float qb = hi ? smem[P + ib - LAM] : smem[P + ib + LAM];
                    qb = __uint_as_float(__float_as_uint(qb) ^ ((p.has_alpha && t) ? 0x80000000u : 0u));
                    hx[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, pb, hx[t], 0, 0, 0);
                    hy[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa, qb, hy[t], 0, 0, 0);
                    hz[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa + qa, pb + qb, hz[t], 0, 0, 0);
                }
                if (s4 == D / 8 - 1) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) hbank[t][r] = hz[t][r] - hx[t][r] - hy[t][r];
                }
            }
            lds_barrier<64>();
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = half_row(4 * kq4 + r), c = 16 * t + i16;      // accumulator: row 4 kq + r, column lane & 15
                    const float re_all = hz[t][r] - hx[t][r] - hy[t][r];
                    const float re = p.has_alpha ? 2.f * hbank[t][r] - re_all : re_all;
                    const float im = hx[t][r] - hy[t][r];
                    const int x = rr ^ c;
                    const uint32_t sg = (p.has_alpha && (x >> (M - 1))) ? 0x80000000u : 0u;
                    float* q = smem + x * LD + rr;
                    q[0] = re;
                    q[P] = im;
                    q[LAM] = __uint_as_float(__float_as_uint(re) ^ sg);
                    q[P + LAM] = __uint_as_float(__float_as_uint(im) ^ sg ^ 0x80000000u);
                }
            lds_barrier<64>();
        } else {
        acc_t gx, gy, gz, bank_re;
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
            gx[r] = 0.f;
            gy[r] = 0.f;
            gz[r] = 0.f;
        }
#pragma unroll
        for (int s2 = 0; s2 < NSTEPS; ++s2) {
            const int k = KSTEP * s2 + kq;
            const int idx = (i ^ k) * LD + k;
            const bool hi = LAMBIT >= 0 && ((KSTEP * s2) & LAM);
            const float pa = smem[idx], pb = smem[P + idx];
            float qa = LAMBIT < 0 ? pa : (hi ? smem[idx - LAM] : smem[idx + LAM]);
            float qb = LAMBIT < 0 ? pb : (hi ? smem[P + idx - LAM] : smem[P + idx + LAM]);
            qa = __uint_as_float(__float_as_uint(qa) ^ sig_mask);
            qb = __uint_as_float(__float_as_uint(qb) ^ sig_mask);
            if constexpr (M == 4) {
                gx = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, pb, gx, 0, 0, 0);
                if constexpr (LAMBIT >= 0) {
                    gy = __builtin_amdgcn_mfma_f32_16x16x4f32(qa, qb, gy, 0, 0, 0);
                    gz = __builtin_amdgcn_mfma_f32_16x16x4f32(pa + qa, pb + qb, gz, 0, 0, 0);
                }
            } else {
                gx = __builtin_amdgcn_mfma_f32_32x32x2f32(pa, pb, gx, 0, 0, 0);
                if constexpr (LAMBIT >= 0) {
                    gy = __builtin_amdgcn_mfma_f32_32x32x2f32(qa, qb, gy, 0, 0, 0);
                    gz = __builtin_amdgcn_mfma_f32_32x32x2f32(pa + qa, pb + qb, gz, 0, 0, 0);
                }
            }
            if (s2 == NSTEPS / 2 - 1) {   // k_top = 0 half done
#pragma unroll
                for (int r = 0; r < NACC; ++r) bank_re[r] = LAMBIT >= 0 ? gz[r] - gx[r] - gy[r] : gx[r];
            }
        }
        lds_barrier<64>();
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
            const int rr = M == 4 ? 4 * kq + r : (r & 3) + 8 * (r >> 2) + 4 * kq;
            if constexpr (LAMBIT < 0) {
                float re, im;
                real_case_planes<float>(gx[r], bank_re[r], p.has_alpha, (rr >> (M - 1)) != 0, (i >> (M - 1)) != 0, &re, &im);
                smem[(rr ^ i) * LD + rr] = re;
                smem[P + (rr ^ i) * LD + rr] = im;
            } else {
                const float re_all = gz[r] - gx[r] - gy[r];
                smem[(rr ^ i) * LD + rr] = p.has_alpha ? 2.f * bank_re[r] - re_all : re_all;
                smem[P + (rr ^ i) * LD + rr] = gx[r] - gy[r];
            }
        }
        lds_barrier<64>();
        }
        // four threads per row: fold bits B1 (lambda's, or the top one) and B2, transform the rest
        {
            constexpr int B1 = LAMBIT == M - 2 ? M - 2 : M - 1, B2 = LAMBIT == M - 2 ? M - 1 : M - 2;
            constexpr int Q = D / 4;
            float2v v[INV_PASSES][Q / 2];
#pragma unroll
            for (int j = 0; j < INV_PASSES; ++j) {
                const int tix = lane + 64 * j;
                const int x = tix >> 2, h1 = (tix >> 1) & 1, h2 = tix & 1;
                const int xi = p.has_alpha ? (x >> (M - 1)) & 1 : 0;
                const int f = LAMBIT >= 0 ? (xi ^ h1) : xi;
                const float* q = smem + f * P + x * LD;
                const float sc = 0.5f / float(D);
                const float s1 = h1 ? -sc : sc, s2f = h2 ? -1.f : 1.f;
                const float2v sc2 = {sc, sc}, s12 = {s1, s1}, s22 = {s2f, s2f};
#pragma unroll
                for (int cp = 0; cp < Q / 2; ++cp) {
                    // sc, s1 = +-2^-k, s2f = +-1: every product is exact, the fused forms round like the unfused ones
                    const int c = 2 * cp;
                    const float2v q00 = {q[c], q[c + 1]}, q10 = {q[c | (1 << B1)], q[(c | (1 << B1)) + 1]};
                    const float2v q01 = {q[c | (1 << B2)], q[(c | (1 << B2)) + 1]};
                    const float2v q11 = {q[c | (1 << B1) | (1 << B2)], q[(c | (1 << B1) | (1 << B2)) + 1]};
                    const float2v lo = pk_fma(q10, s12, q00 * sc2);
                    const float2v up = pk_fma(q11, s12, q01 * sc2);
                    v[j][cp] = pk_fma(up, s22, lo);
                }
                wht_pairs<Q / 2>(v[j]);
            }
            lds_barrier<64>();   // every pass has read its planes before any result lands in plane 0
#pragma unroll
            for (int j = 0; j < INV_PASSES; ++j) {
                const int tix = lane + 64 * j;
                const int x = tix >> 2, h1 = (tix >> 1) & 1, h2 = tix & 1;
                float* o = smem + x * LD + (h1 << B1) + (h2 << B2);
#pragma unroll
                for (int cp = 0; cp < Q / 2; ++cp) {
                    o[2 * cp] = v[j][cp][0];
                    o[2 * cp + 1] = v[j][cp][1];
                }
            }
        }
        lds_barrier<64>();
        {
            float* orow = outp + item * p.out_stride;   // (uniform) row base + the lane's index
            if (out_vec) {
#pragma unroll
                for (int u4 = 0; u4 < EPL / 4; ++u4) {
                    uint32_t w[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) w[c] = *(const lds_u32*)size_t(om[4 * u4 + c] & 0x3fffffu) ^ (om[4 * u4 + c] & 0x80000000u);
                    spinor_store4(orow, lane + 64 * u4, __uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3]));
                }
            } else {
#pragma unroll
                for (int u = 0; u < EPL; ++u) {
                    const uint32_t eo = om[u];
                    const int e = 4 * lane + (u & 3) + 256 * (u >> 2);
                    const float val = __uint_as_float(*(const lds_u32*)size_t(eo & 0x3fffffu) ^ (eo & 0x80000000u));
                    if (!(eo & 0x40000000u)) orow[e] = p.beta ? orow[e] + val : val;
                }
            }
        }
        lds_barrier<64>();
    }
}

// The wave-per-item kernel in f64 (the reference's value type) for n = 7..10: the same one-plane algorithm on
// v_mfma_f64_16x16x4_f64.  The D x D result (D = 16 at n <= 8, 32 at n = 9, 10) is TB x TB tiles of 16 x 16
// (TB = D / 16); lane (kq, i16) feeds row / column 16 b + i16 of tile row / column b with the k = 4 s + kq slice.
// Planes of doubles: 4.3 KB / 16.9 KB of LDS per item.  Error bound 64 * 2^-52 |A|_2 |B|_2, like the n = 11, 12 kernel.
typedef double double4v __attribute__((ext_vector_type(4)));

template <int M, int LAMBIT>
__global__ __launch_bounds__(64) void k_gp_spinor_wave1d(SpinorArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* smem = reinterpret_cast<double*>(smem_raw);
    constexpr int D = 1 << M, LD = D + 1, P = D * LD;
    constexpr int NE = D * D, EPL = NE / 64;
    constexpr int LAM = LAMBIT >= 0 ? (1 << LAMBIT) : 0;
    constexpr int FWD_PASSES = 4 * D / 64, INV_PASSES = 4 * D / 64;
    constexpr int TB = D / 16;                    // tiles per side
    constexpr int NSTEPS = D / 4;                 // k = 4 s + kq
    const int lane = threadIdx.x;
    const double* left = static_cast<const double*>(p.left);
    const double* right = static_cast<const double*>(p.right);
    double* outp = static_cast<double*>(p.out);

    uint32_t lm[EPL / 2], rm[EPL / 2], om[EPL / 2];
#pragma unroll
    for (int w = 0; w < EPL / 2; ++w) {
        lm[w] = uint32_t(p.left_map[lane + 128 * w]) | (uint32_t(p.left_map[lane + 128 * w + 64]) << 16);
        rm[w] = uint32_t(p.right_map[lane + 128 * w]) | (uint32_t(p.right_map[lane + 128 * w + 64]) << 16);
        om[w] = uint32_t(p.out_map[lane + 128 * w]) | (uint32_t(p.out_map[lane + 128 * w + 64]) << 16);
    }
    auto entry = [](const uint32_t (&m)[EPL / 2], int u) -> uint32_t { return (u & 1) ? m[u >> 1] >> 16 : m[u >> 1]; };
    double va[EPL], vb[EPL];
    const bool rows_full = p.left_len == NE && p.right_len == NE;
    auto fetch = [&](int64_t item) {
        const double* lrow = left + item * p.left_stride + lane;
        const double* rrow = right + item * p.right_stride + lane;
#pragma unroll
        for (int u = 0; u < EPL; ++u) {
            const int e = lane + 64 * u;
            va[u] = (rows_full || e < p.left_len) ? lrow[64 * u] : 0.0;
            vb[u] = (rows_full || e < p.right_len) ? rrow[64 * u] : 0.0;
        }
    };
    int64_t item = blockIdx.x;
    if (item < p.batch) fetch(item);
    const int i16 = lane & 15, kq = lane >> 4;

    for (; item < p.batch; item += gridDim.x) {
#pragma unroll
        for (int w = 0; w < EPL / 2; ++w) asm volatile("" : "+v"(lm[w]), "+v"(rm[w]), "+v"(om[w]));
        if (!p.left_full || !p.right_full) {
            for (int j = lane; j < 2 * P; j += 64) smem[j] = 0.0;
            lds_barrier<64>();
        }
        {
            auto put = [&](double* plane, uint32_t e, double a, int canon) {
                if (canon) a = 0.0 + a;
                if (e & 1u) a = -a;
                *reinterpret_cast<double*>(reinterpret_cast<char*>(plane) + ((e & 0x7ffcu) << 1)) = a;
            };
#pragma unroll
            for (int u = 0; u < EPL; ++u) {
                put(smem, entry(lm, u), va[u], p.canon_left);
                put(smem + P, entry(rm, u), vb[u], p.canon_right);
            }
        }
        lds_barrier<64>();
        if (item + gridDim.x < p.batch) fetch(item + gridDim.x);

        // one transform per row, two threads per row: (operand, x, half) = 4 D half-rows over 64 lanes
#pragma unroll
        for (int j = 0; j < FWD_PASSES; ++j) {
            const int hidx = lane + 64 * j;
            const int hb = hidx & 1;
            double* row = smem + (hidx >> (M + 1)) * P + ((hidx >> 1) & (D - 1)) * LD;
            const double sg = hb ? -1.0 : 1.0;
            double v[D / 2];
#pragma unroll
            for (int c = 0; c < D / 2; ++c) v[c] = fma_x(row[c + D / 2], sg, row[c]);   // sg = +-1: an exact product, one instruction
            wht<D / 2, double>(v);
#pragma unroll
            for (int c = 0; c < D / 2; ++c) row[c + (D / 2) * hb] = v[c];
        }
        lds_barrier<64>();

        constexpr bool HALF = M == 5 && LAMBIT >= 0 && GAAST_SPINOR_HALF;   // rows with the lambda bit clear, mirrored (k_gp_spinor12s)
        constexpr int TR = HALF ? 1 : TB;                                    // row tiles
        auto tile_row = [](int rb, int j) {                                  // row j (0..15) of row tile rb
            if (!HALF) return 16 * rb + j;
            return LAMBIT == M - 1 ? j : (((j & 8) << 1) | (j & 7));
        };
        double4v gx[TB * TB], gy[TB * TB], gz[TB * TB], bank_re[TB * TB];
#pragma unroll
        for (int t = 0; t < TB * TB; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                gx[t][r] = 0.0;
                gy[t][r] = 0.0;
                gz[t][r] = 0.0;
                bank_re[t][r] = 0.0;
            }
#pragma unroll
        for (int s2 = 0; s2 < NSTEPS; ++s2) {
            const int k = 4 * s2 + kq;
            const bool hi = LAMBIT >= 0 && ((4 * s2) & LAM);
            double pa[TB], qa[TB], pb[TB], qb[TB];
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                const int cc = 16 * b + i16;                       // column of B this lane feeds
                const int ib = (cc ^ k) * LD + k;
                pb[b] = smem[P + ib];
                qb[b] = LAMBIT < 0 ? pb[b] : (hi ? smem[P + ib - LAM] : smem[P + ib + LAM]);
                if (p.has_alpha && (cc >> (M - 1))) qb[b] = -qb[b];   // sigma of column cc
                if (b < TR) {
                    const int rc = tile_row(b, i16);               // row of A this lane feeds
                    const int ia = (rc ^ k) * LD + k;
                    pa[b] = smem[ia];
                    qa[b] = LAMBIT < 0 ? pa[b] : (hi ? smem[ia - LAM] : smem[ia + LAM]);
                    if (p.has_alpha && (rc >> (M - 1))) qa[b] = -qa[b];   // sigma of row rc
                }
            }
#pragma unroll
            for (int rb = 0; rb < TR; ++rb)
#pragma unroll
                for (int cb = 0; cb < TB; ++cb) {
                    const int t = rb * TB + cb;
                    gx[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[rb], pb[cb], gx[t], 0, 0, 0);
                    if constexpr (LAMBIT >= 0) {
                        gy[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[rb], qb[cb], gy[t], 0, 0, 0);
                        gz[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[rb] + qa[rb], pb[cb] + qb[cb], gz[t], 0, 0, 0);
                    }
                }
            if (s2 == NSTEPS / 2 - 1) {   // k_top = 0 half done
#pragma unroll
                for (int t = 0; t < TB * TB; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) bank_re[t][r] = LAMBIT >= 0 ? gz[t][r] - gx[t][r] - gy[t][r] : gx[t][r];
            }
        }
        lds_barrier<64>();
#pragma unroll
        for (int rb = 0; rb < TR; ++rb)
#pragma unroll
            for (int cb = 0; cb < TB; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = rb * TB + cb;
                    // accumulator layout of v_mfma_f64_16x16x4_f64: column = lane & 15, row = (lane >> 4) + 4 r
                    const int rr = tile_row(rb, kq + 4 * r), cc = 16 * cb + i16;     // element (row rr, column cc) of the product
                    double re, im;
                    if constexpr (LAMBIT < 0) {
                        real_case_planes<double>(gx[t][r], bank_re[t][r], p.has_alpha, (rr >> (M - 1)) != 0, (cc >> (M - 1)) != 0, &re, &im);
                    } else {
                        const double re_all = gz[t][r] - gx[t][r] - gy[t][r];
                        re = p.has_alpha ? 2.0 * bank_re[t][r] - re_all : re_all;
                        im = gx[t][r] - gy[t][r];
                    }
                    double* q = smem + (rr ^ cc) * LD + rr;
                    q[0] = re;
                    q[P] = im;
                    if (HALF) {   // the mirror image (rr ^ lambda, cc ^ lambda): same row of S, conjugated, times sigma
                        const bool neg = p.has_alpha && (((rr ^ cc) >> (M - 1)) & 1);
                        q[LAM] = neg ? -re : re;
                        q[P + LAM] = neg ? im : -im;
                    }
                }
        lds_barrier<64>();
        // four threads per row: fold bits B1 (lambda's, or the top one) and B2, transform the rest
        {
            constexpr int B1 = LAMBIT == M - 2 ? M - 2 : M - 1, B2 = LAMBIT == M - 2 ? M - 1 : M - 2;
            constexpr int Q = D / 4;
            double v[INV_PASSES][Q];
#pragma unroll
            for (int j = 0; j < INV_PASSES; ++j) {
                const int tix = lane + 64 * j;
                const int x = tix >> 2, h1 = (tix >> 1) & 1, h2 = tix & 1;
                const int xi = p.has_alpha ? (x >> (M - 1)) & 1 : 0;
                const int f = LAMBIT >= 0 ? (xi ^ h1) : xi;
                const double* q = smem + f * P + x * LD;
                const double sc = 0.5 / double(D);
                const double s1 = h1 ? -sc : sc, s2f = h2 ? -1.0 : 1.0;
#pragma unroll
                for (int c = 0; c < Q; ++c) {
                    // sc, s1 = +-2^-k, s2f = +-1: every product is exact, the fused forms round like the unfused ones
                    const double lo = fma_x(q[c | (1 << B1)], s1, q[c] * sc);
                    const double up = fma_x(q[c | (1 << B1) | (1 << B2)], s1, q[c | (1 << B2)] * sc);
                    v[j][c] = fma_x(up, s2f, lo);
                }
                wht<Q, double>(v[j]);
            }
            lds_barrier<64>();   // every pass has read its planes before any result lands in plane 0
#pragma unroll
            for (int j = 0; j < INV_PASSES; ++j) {
                const int tix = lane + 64 * j;
                const int x = tix >> 2, h1 = (tix >> 1) & 1, h2 = tix & 1;
                double* o = smem + x * LD + (h1 << B1) + (h2 << B2);
#pragma unroll
                for (int c = 0; c < Q; ++c) o[c] = v[j][c];
            }
        }
        lds_barrier<64>();
        {
            double* orow = outp + item * p.out_stride + lane;
#pragma unroll
            for (int u = 0; u < EPL; ++u) {
                const uint32_t eo = entry(om, u);
                double val = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(smem) + ((eo & 0xfffcu) << 1));
                if (eo & 1u) val = -val;
                if (p.out_full && !p.beta) {
                    orow[64 * u] = val;
                } else if (!(eo & 2u)) {
                    orow[64 * u] = p.beta ? orow[64 * u] + val : val;
                }
            }
        }
        lds_barrier<64>();
    }
}

}  // namespace gaast
