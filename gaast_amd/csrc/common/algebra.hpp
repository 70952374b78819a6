// Blade/index/sign rules of the reference, in closed form on machine words.
//
// The reference works on heap BitVecs (src/algebra.rs); this is an independent
// implementation of the same rules:
//   T2  component <-> blade: within grade k, index = colex rank of the blade's set of
//       basis-vector positions (algebra.rs:221-246).  Colex rank order == increasing numeric
//       order of the bitmask among masks of equal popcount, so the tables are built by one
//       ascending sweep over 0..2^n (the reference's own TODO, algebra.rs:54-55).
//   T3  reordering sign (algebra.rs:199-209) = (-1)^#{(p,q): p in a, q in b, p > q}
//       = parity(a & X(b)), X(b) = exclusive prefix-xor of b.  Coefficient = sign times the
//       metric diagonal over shared basis vectors, multiplied in ascending bit order
//       (algebra.rs:78-81).
#pragma once
#include <cstdint>
#include <vector>

namespace gaast {

inline uint64_t n_choose_k(uint64_t n, uint64_t k) {  // algebra.rs:252-254
    if (k > n) return 0;
    if (k > n - k) k = n - k;
    uint64_t r = 1;
    for (uint64_t d = 1; d <= k; ++d) r = r * (n - k + d) / d;
    return r;
}

// exclusive prefix xor: bit p of the result = parity of the bits of b below p
inline uint64_t prefix_parity_excl(uint64_t b) {
    uint64_t x = b << 1;
    x ^= x << 1;
    x ^= x << 2;
    x ^= x << 4;
    x ^= x << 8;
    x ^= x << 16;
    x ^= x << 32;
    return x;
}

// 1 when the canonical reordering of blade a times blade b picks up a minus sign
inline int reorder_parity(uint64_t a, uint64_t b) {
    return __builtin_popcountll(a & prefix_parity_excl(b)) & 1;
}

// ortho_basis_blades_gp (algebra.rs:73-83): coefficient of e_a * e_b = coeff * e_{a^b}
inline double blades_gp_coeff(int n, const double* metric_diag, uint64_t a, uint64_t b) {
    double coef = reorder_parity(a, b) ? -1.0 : 1.0;
    uint64_t shared = a & b;
    for (int bit = 0; bit < n; ++bit)
        if ((shared >> bit) & 1ULL) coef *= metric_diag[bit];
    return coef;
}

// Per-dimension tables for the component <-> blade maps.
struct BladeTable {
    int n = 0;
    std::vector<uint32_t> grade_dim;               // C(n,k)
    std::vector<std::vector<uint32_t>> blade_of;   // [grade][index] -> bitmask
    std::vector<uint32_t> index_of;                // [bitmask] -> index within its grade

    explicit BladeTable(int dim) : n(dim) {
        grade_dim.assign(n + 1, 0);
        blade_of.assign(n + 1, {});
        index_of.assign(size_t(1) << n, 0);
        for (uint64_t m = 0; m < (uint64_t(1) << n); ++m) {
            int k = __builtin_popcountll(m);
            index_of[m] = grade_dim[k]++;
            blade_of[k].push_back(uint32_t(m));
        }
    }
};

// Number of components of a row holding the grades of `mask` with slices sized for `dim`.
inline int64_t row_len_of(int dim, uint64_t mask) {
    int64_t len = 0;
    for (int k = 0; k < 64; ++k)
        if ((mask >> k) & 1ULL) len += int64_t(n_choose_k(uint64_t(dim), uint64_t(k)));
    return len;
}

// Offset of grade k inside such a row (-1 when absent).
inline int64_t grade_offset(int dim, uint64_t mask, int k) {
    if (!((mask >> k) & 1ULL)) return -1;
    int64_t off = 0;
    for (int j = 0; j < k; ++j)
        if ((mask >> j) & 1ULL) off += int64_t(n_choose_k(uint64_t(dim), uint64_t(j)));
    return off;
}

}  // namespace gaast
