// kernels_mfma6.hip.hpp -- k_gp_mfma6<T>: the dense geometric product at n = 6 (64 components) on the 16x16x4 matrix
// instructions, both value types.  Included through kernels.hip.hpp.
#pragma once
#include "kernels_dense.hip.hpp"

namespace gaast {

// ------------------------------------------------------------------------------------------
// Product arm (eval.rs:61-86), dense geometric product of a 6-dimensional algebra (R^6, R^{4,2}, R^{3,3}; the Cl(6)
// products of parity-pure n = 7 operands): 4^6 multiply-adds = FOUR 16x16x4 instructions per item, none wasted, ONE WAVE
// PER ITEM, persistent single-wave workgroups.  The path is HBM-bound (768 B and 8,192 flop per item in f32), so everything
// around the four instructions is kept small:
//
//   blade = (top2 | hi2 | lo2);   tile row = (u = a_top, x = c_lo),  tile column = (v = b_top, y = c_hi);
//   K = (a_hi, b_lo) = (instruction s, lane group kq);   T[(u,x),(v,y)] = sum_{s,kq} A[u, s, x ^ kq] B[v, y ^ s, kq] * sign;
//   component (w, y, x) = sum_u T[(u,x),(u ^ w, y)]  --  four tile elements, which sit in the four lanes of a QUAD
//   (v in the low two bits of the column, u = the accumulator register): three DPP quad permutations, no LDS exchange.
//
// The sign (-1)^#{(p,q): p in a, q in b, p > q} * metric(a & b) (algebra.rs:73-83, 199-209) factors into
//   (row, k):      R(a_lo, b_lo) + |a_lo & b_lo & NEG| + |u| (|b_lo| + |a_hi|) + |a_hi| |b_lo|      -> the A image
//   (k, column):   R(a_hi, b_hi) + |a_hi & b_hi & NEG|                                               -> the B image
//   (row, column): |u| |y| + R(u, v) + |u & v & NEG|                                                 -> the accumulators
// (|u| |b_hi| = |u| (|y| + |a_hi|) mod 2 splits over the first and the third); a NULL basis vector zeroes the slots / the
// accumulators whose blades share it -- any +-1 / 0 diagonal metric runs as it is, no basis permutation.  Checked on the
// CPU first: tools/proto/mfma6_tile.py (40 random metrics and operands against the bitmask product).
//
// Staging writes the operands IN INSTRUCTION ORDER: the lane that loaded component A[u, a_hi, a_lo] stores it, signed, into
// slot a_hi of the four lanes (kq, row (u, a_lo ^ kq)) that multiply it, likewise B -- eight 4-byte (8-byte) stores per
// lane and item -- so that a lane's operands of all four instructions are ONE 16-byte read per operand (f64: two).
// Missing components (operands that hold only some grades) and vanishing slots (null vectors) are slots nobody writes: they stay
// zero from the initial clear.
// Which lane moves which component is the host's choice (plan.cpp: build_map deals full operands by LDS bank; the B stores go
// out in the slot order i ^ b_lo): no bank conflict on the eight stores.
// DEPTH items are in flight per wave (two loads each): 32 waves x DEPTH x 512 B per CU hide the HBM latency.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t group_reorder_parity(uint32_t a, uint32_t b) {   // #{(p, q): p in a, q in b, p > q} mod 2, 2-bit groups
    return (((b & 1u) ? __builtin_popcount(a >> 1) : 0) + ((b & 2u) ? __builtin_popcount(a >> 2) : 0)) & 1u;
}

// FAST: both operands hold every blade, every blade is produced and nothing is accumulated (the host checks): no lane is
// ever masked around a memory instruction, so the item loop is straight-line code and the waits for the prefetched rows are
// COUNTED (vmcnt(N): loads and stores share one in-order counter) -- with a conditional store in the loop the compiler waits
// for everything, the loads just issued for the item DEPTH ahead included (first version: 4,400 cycles per item and wave).
#ifndef GAAST_MFMA6_DEPTH
#define GAAST_MFMA6_DEPTH 4   /* (build switch for A/B runs) items in flight per wave: 4 measured best (8: -3 %) */
#endif
#ifndef GAAST_MFMA6_SPLIT
#define GAAST_MFMA6_SPLIT 0   /* (build switch for A/B runs) two accumulator chains of two instructions instead of one of four: measured -4 % */
#endif
#ifndef GAAST_MFMA6_WAVES
#define GAAST_MFMA6_WAVES 1   /* (build switch for A/B runs) independent waves per workgroup */
#endif
template <typename T, bool SCALED, bool FAST>
__global__ __launch_bounds__(64 * GAAST_MFMA6_WAVES) void k_gp_mfma6(DenseArgs<T> p) {
    typedef Mfma16x4<T> MM;
    constexpr bool F32 = sizeof(T) == 4;
    constexpr int ES = MM::SHIFT, DEPTH = GAAST_MFMA6_DEPTH;
    // LDS (bytes): A image, then B image; f32: lane * 16 + slot * 4; f64: (slot >> 1) * 1024 + lane * 16 + (slot & 1) * 8 -- a
    // lane's 16-byte reads of one half are consecutive over the lanes (no bank conflict)
    constexpr uint32_t IMG = F32 ? 1024u : 2048u;
    constexpr uint32_t DUMMY = 2u * IMG, PER_WAVE = 2u * IMG + 64u * uint32_t(sizeof(T));   // one element per lane nobody reads
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // every wave of the workgroup is on its own: its items, its images (no workgroup barrier anywhere)
    constexpr int WAVES = GAAST_MFMA6_WAVES;
    const int wave_id = WAVES == 1 ? 0 : int(threadIdx.x) >> 6;   // (one wave: the item index and every row base stay in scalar registers)
    lds_u8* lds = (lds_u8*)smem_raw + uint32_t(wave_id) * PER_WAVE;
    typedef __attribute__((address_space(3))) T lds_t;
    const int tid = int(threadIdx.x) & 63;
    for (int e = tid; e < int(2 * IMG / 4); e += 64) reinterpret_cast<uint32_t*>(smem_raw + size_t(wave_id) * PER_WAVE)[e] = 0u;
    auto slot_addr = [&](uint32_t lane, uint32_t slot) -> uint32_t {
        return F32 ? lane * 16u + slot * 4u : (slot >> 1) * 1024u + lane * 16u + (slot & 1u) * 8u;
    };
    auto tile_row = [&](uint32_t u, uint32_t x) -> uint32_t { return F32 ? 4u * x + u : 4u * u + x; };   // Mfma16x4<T>::row(kq = x, r = u)
    const uint32_t neg = p.neg_hi & 63u, zero = p.zero_hi & 63u;   // basis vectors squaring to -1 / 0 (blade bits)
    const uint32_t NEG_L = neg & 3u, NEG_H = (neg >> 2) & 3u, NEG_T = neg >> 4, Z_L = zero & 3u, Z_H = (zero >> 2) & 3u, Z_T = zero >> 4;

    // ---- this lane's components: entry `tid` of each operand map (row offset | blade << 16 | negate << 31) ----
    const bool has_a = FAST || tid < p.left_count, has_b = FAST || tid < p.right_count;
    const uint32_t ma = has_a ? p.left_map[tid] : 0u, mb = has_b ? p.right_map[tid] : 0u;
    const uint32_t off_a = (ma & 0xffffu) << ES, off_b = (mb & 0xffffu) << ES;
    // store address and sign bit of the four slots.  A slot that VANISHES (its blades share a null basis vector) is not written at
    // all -- its store goes to the lane's dummy element, the slot keeps the zero of the initial clear: no mask per value and item
    uint32_t wa[4], sa[4], wb[4], sb[4];
    const uint32_t dummy = DUMMY + uint32_t(tid) * uint32_t(sizeof(T));
    {
        const uint32_t a = (ma >> 16) & 63u, u = a >> 4, ah = (a >> 2) & 3u, al = a & 3u;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {   // kq = b_lo
            const uint32_t x = al ^ j;
            const uint32_t par = group_reorder_parity(al, j) ^ (uint32_t(__builtin_popcount(al & j & NEG_L)) & 1u) ^
                                 (uint32_t(__builtin_popcount(u)) & uint32_t(__builtin_popcount(j) + __builtin_popcount(ah)) & 1u) ^
                                 (uint32_t(__builtin_popcount(ah)) & uint32_t(__builtin_popcount(j)) & 1u);
            wa[j] = (al & j & Z_L) ? dummy : slot_addr(j * 16u + tile_row(u, x), ah);
            sa[j] = ((par & 1u) << 31) ^ (ma & 0x80000000u);
        }
        const uint32_t b = (mb >> 16) & 63u, v = b >> 4, bh = (b >> 2) & 3u, bl = b & 3u;
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {   // store i writes slot s = i ^ bl (a_hi): the lanes of one store instruction then spread
            const uint32_t s = i ^ bl;       // over the slots -- with one slot per instruction they share 8 of the 32 banks
            const uint32_t y = bh ^ s;
            const uint32_t par = group_reorder_parity(s, bh) ^ (uint32_t(__builtin_popcount(s & bh & NEG_H)) & 1u);
            wb[i] = (s & bh & Z_H) ? dummy : IMG + slot_addr(bl * 16u + 4u * y + v, s);
            sb[i] = ((par & 1u) << 31) ^ (mb & 0x80000000u);
        }
    }
    const T scale_a = (SCALED && has_a && p.left_scale) ? p.left_scale[tid] : T(1);
    const T scale_b = (SCALED && has_b && p.right_scale) ? p.right_scale[tid] : T(1);
    // ---- this lane's tile elements: x = kq, column = 4 y + v, register r = u ----
    const uint32_t kq = uint32_t(tid) >> 4, col = uint32_t(tid) & 15u, y = col >> 2, v = col & 3u;
    uint32_t su[4], ku[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t par = (uint32_t(__builtin_popcount(u)) & uint32_t(__builtin_popcount(y)) & 1u) ^ group_reorder_parity(u, v) ^
                             (uint32_t(__builtin_popcount(u & v & NEG_T)) & 1u);
        su[u] = (par & 1u) << 31;
        ku[u] = (u & v & Z_T) ? 0u : ~0u;
        asm volatile("" : "+v"(ku[u]));   // opaque: (value ^ sign) & keep is ONE v_bitop3 then (seen as a select, the compiler builds compare +
                                          // v_cndmask pairs per item)
    }
    // the lane ends with component (w = v, y, x = kq)
    const uint32_t comp = (v << 4) | (y << 2) | kq;
    const int32_t ow = p.out_map[comp];
    const bool ook = ow >= 0;
    const uint32_t ooff = uint32_t(ow & 0x3fffffff) << ES, osg = (uint32_t(ow) & 0x40000000u) << 1;
    const T osc = (SCALED && p.out_scale) ? p.out_scale[comp] : T(1);
    const uint32_t ra = uint32_t(tid) * 16u, rb = IMG + uint32_t(tid) * 16u;   // this lane's operand reads

    auto apply = [&](T val, uint32_t sign, uint32_t keep) -> T {   // sign flip and (null vectors) vanishing, as bit operations
        if constexpr (F32) {
            return __uint_as_float((__float_as_uint(float(val)) ^ sign) & keep);
        } else {
            return T(__hiloint2double(int((uint32_t(__double2hiint(double(val))) ^ sign) & keep), int(uint32_t(__double2loint(double(val))) & keep)));
        }
    };
    auto quad_xor = [&](T val, auto ctrl) -> T {   // the value the lane (v ^ j) of this quad holds
        constexpr int CTRL = decltype(ctrl)::value;
        if constexpr (F32) {
            return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(float(val)), CTRL, 0xf, 0xf, true));
        } else {
            const int lo_w = __builtin_amdgcn_mov_dpp(__double2loint(double(val)), CTRL, 0xf, 0xf, true);
            const int hi_w = __builtin_amdgcn_mov_dpp(__double2hiint(double(val)), CTRL, 0xf, 0xf, true);
            return T(__hiloint2double(hi_w, lo_w));
        }
    };

    // ---- DEPTH items in flight per wave (register sets, the item loop unrolled DEPTH times) ----
    T pa[DEPTH], pb[DEPTH];
    const int64_t step = int64_t(gridDim.x) * WAVES, first_item = int64_t(blockIdx.x) * WAVES + wave_id;
    auto fetch = [&](int64_t item, T& fa, T& fb) {   // (uniform) row base + the lane's byte offset; a lane without a component re-reads element 0
        uint32_t oa = off_a, ob = off_b;
        asm volatile("" : "+v"(oa), "+v"(ob));
        fa = __builtin_nontemporal_load(reinterpret_cast<const T*>(reinterpret_cast<const unsigned char*>(p.left + item * p.left_stride) + oa));
        fb = __builtin_nontemporal_load(reinterpret_cast<const T*>(reinterpret_cast<const unsigned char*>(p.right + item * p.right_stride) + ob));
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        const int64_t item = first_item + d * step;
        pa[d] = pb[d] = T(0);
        if (item < p.batch) fetch(item, pa[d], pb[d]);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the clear

    auto one_item = [&](int64_t item, T& fa, T& fb) {
        // (the reference's 0.0 + x on a product operand, eval.rs:27-31, only turns -0.0 into +0.0: a zero contributes the same to a
        //  re-ordered sum whose accumulators start from +0.0 -- not spent here: it cost an addition and a select per value)
        T va = fa, vb = fb;
        if (SCALED) {
            va = va * scale_a;
            vb = vb * scale_b;
        }
        {   // the item DEPTH items ahead, into the registers just consumed (the last items re-read themselves: a known number of loads)
            const int64_t nn = item + DEPTH * step;
            fetch(nn < p.batch ? nn : item, fa, fb);
        }
        if (has_a) {   // (FAST: every lane)
#pragma unroll
            for (int j = 0; j < 4; ++j) *(lds_t*)(lds + wa[j]) = MM::flip(va, sa[j]);
        }
        if (has_b) {
#pragma unroll
            for (int s = 0; s < 4; ++s) *(lds_t*)(lds + wb[s]) = MM::flip(vb, sb[s]);
        }
        lds_barrier<64>();   // one wave: the LDS executes its instructions in order
        T av[4], bv[4];
        if constexpr (F32) {
            const float4v qa = *(const __attribute__((address_space(3))) float4v*)(lds + ra);
            const float4v qb = *(const __attribute__((address_space(3))) float4v*)(lds + rb);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                av[s] = qa[s];
                bv[s] = qb[s];
            }
        } else {
            typedef double double2m __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double2m qa = *(const __attribute__((address_space(3))) double2m*)(lds + ra + h * 1024u);
                const double2m qb = *(const __attribute__((address_space(3))) double2m*)(lds + rb + h * 1024u);
                av[2 * h] = qa[0];
                av[2 * h + 1] = qa[1];
                bv[2 * h] = qb[0];
                bv[2 * h + 1] = qb[1];
            }
        }
        typename MM::acc_t acc = {T(0), T(0), T(0), T(0)};
#if GAAST_MFMA6_SPLIT
        {   // two chains of two dependent instructions (a dependent 16x16x4 waits 40 cycles for its accumulator), summed at the end
            typename MM::acc_t acc2 = {T(0), T(0), T(0), T(0)};
            acc = MM::mma(av[0], bv[0], acc);
            acc2 = MM::mma(av[1], bv[1], acc2);
            acc = MM::mma(av[2], bv[2], acc);
            acc2 = MM::mma(av[3], bv[3], acc2);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = acc[r] + acc2[r];
        }
#else
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = MM::mma(av[s], bv[s], acc);
#endif
        // ---- the four tile elements of every component meet inside a quad ----
        T e1 = apply(acc[1], su[1], ku[1]);
        T e2 = apply(acc[2], su[2], ku[2]);
        T e3 = apply(acc[3], su[3], ku[3]);
        T res = acc[0];                                                              // u = 0: no sign, never vanishes
        res = res + quad_xor(e1, std::integral_constant<int, 0xB1>{});               // from lane v ^ 1
        res = res + quad_xor(e2, std::integral_constant<int, 0x4E>{});               // from lane v ^ 2
        res = res + quad_xor(e3, std::integral_constant<int, 0x1B>{});               // from lane v ^ 3
        // ---- result -> graded row ----
        T val = res;
        if (SCALED) val = val * osc;
        val = MM::flip(val, osg);
        if constexpr (FAST) {
            val = T(0) + val;                       // a zero result under a negated reordering sign stays +0.0 (res itself is never -0.0:
                                                    // the accumulators start from +0.0)
            uint32_t oo = ooff;
            asm volatile("" : "+v"(oo));
            __builtin_nontemporal_store(val, reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(p.out + item * p.out_stride) + oo));
        } else if (ook) {
            T* q = reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(p.out + item * p.out_stride) + ooff);
            if (osg && !p.beta) val = T(0) + val;
            *q = p.beta ? *q + val : val;
        }
        lds_barrier<64>();   // the images are rewritten by the next item
    };
    int64_t item = first_item;
    for (; item + (DEPTH - 1) * step < p.batch; item += DEPTH * step) {   // DEPTH items per round: straight-line code
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) one_item(item + d * step, pa[d], pb[d]);
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (item + d * step < p.batch) one_item(item + d * step, pa[d], pb[d]);
}

}  // namespace gaast
