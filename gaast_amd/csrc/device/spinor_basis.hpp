// spinor_basis.hpp -- host-side index bookkeeping of the one-plane matrix-representation kernel
// (k_gp_spinor12s, kernels_spinor.hip.hpp; derivation and numpy prototype: tools/proto/spinor_single_plane.py).
//
// A blade is e_S = i^k X^x Z^z, k = 2u + f.  The parity f is a linear function of the index pair,
// f = alpha.x ^ lambda.z.  A change of basis of the m-bit index spaces (z' = L z, x' = L^-T x, which keeps
// c.z and therefore the Walsh-Hadamard structure) puts lambda on one coordinate (lam_bit: the top bit
// m-1, or m-2 when alpha' has to take the top one) and alpha on the top coordinate or nowhere.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace gaast {

struct SpinorBasis {
    int m = 0;
    std::vector<uint32_t> L, LinvT;   // rows as bitmasks
    int lam_bit = -1;                 // -1: lambda = 0 (every phase of a row is the same)
    bool has_alpha = false;           // alpha' = e_(m-1), else 0
    static uint32_t par(uint32_t v) { return uint32_t(__builtin_popcount(v)) & 1u; }
    static uint32_t apply(const std::vector<uint32_t>& rows, uint32_t v) {
        uint32_t r = 0;
        for (size_t i = 0; i < rows.size(); ++i) r |= par(rows[i] & v) << i;
        return r;
    }
    uint32_t map_x(uint32_t x) const { return apply(LinvT, x); }
    uint32_t map_z(uint32_t z) const { return apply(L, z); }
};

inline std::vector<uint32_t> gf2_transpose(const std::vector<uint32_t>& rows) {
    const size_t m = rows.size();
    std::vector<uint32_t> t(m, 0);
    for (size_t i = 0; i < m; ++i)
        for (size_t j = 0; j < m; ++j) t[j] |= ((rows[i] >> j) & 1u) << i;
    return t;
}

inline std::vector<uint32_t> gf2_inverse(std::vector<uint32_t> a) {
    const size_t m = a.size();
    std::vector<uint32_t> inv(m);
    for (size_t i = 0; i < m; ++i) inv[i] = 1u << i;
    for (size_t col = 0; col < m; ++col) {
        size_t piv = col;
        while (piv < m && !((a[piv] >> col) & 1u)) ++piv;
        if (piv == m) throw std::runtime_error("spinor basis: singular matrix");
        std::swap(a[col], a[piv]);
        std::swap(inv[col], inv[piv]);
        for (size_t i = 0; i < m; ++i)
            if (i != col && ((a[i] >> col) & 1u)) {
                a[i] ^= a[col];
                inv[i] ^= inv[col];
            }
    }
    return inv;
}

// alpha, lambda: f(x, z) = alpha.x ^ lambda.z
inline SpinorBasis choose_spinor_basis(int m, uint32_t alpha, uint32_t lam) {
    SpinorBasis b;
    b.m = m;
    const int top = m - 1;
    std::vector<uint32_t> rows(size_t(m), 0);
    std::vector<char> fixed(size_t(m), 0);
    int want_one = -1;                                   // the row that must satisfy row.alpha = 1
    if (lam == 0) {
        b.lam_bit = -1;
        if (alpha) want_one = top;
    } else {
        if (alpha && SpinorBasis::par(lam & alpha) == 0) {
            b.lam_bit = m - 2;                           // lambda.alpha = 0: alpha' cannot share lambda's coordinate
            want_one = top;
        } else {
            b.lam_bit = top;                             // (L alpha)_top = lambda.alpha = 1 (or alpha = 0)
        }
        rows[size_t(b.lam_bit)] = lam;
        fixed[size_t(b.lam_bit)] = 1;
    }
    std::vector<uint32_t> span;                          // reduced basis of the rows chosen so far
    auto reduce = [&](uint32_t v) {
        for (uint32_t s : span) v = v < (v ^ s) ? v : (v ^ s);
        return v;
    };
    auto add = [&](uint32_t v) {
        span.push_back(reduce(v));
        for (size_t i = span.size(); i-- > 1;)
            if (span[i] > span[i - 1]) std::swap(span[i], span[i - 1]);
    };
    if (b.lam_bit >= 0) add(lam);
    for (int i = 0; i < m; ++i) {
        if (fixed[size_t(i)]) continue;
        const uint32_t need = i == want_one ? 1u : 0u;
        bool found = false;
        for (uint32_t v = 1; v < (1u << m); ++v) {
            if (SpinorBasis::par(v & alpha) == need && reduce(v) != 0) {
                rows[size_t(i)] = v;
                add(v);
                found = true;
                break;
            }
        }
        if (!found) throw std::runtime_error("spinor basis: cannot complete the basis");
    }
    b.L = rows;
    b.LinvT = gf2_transpose(gf2_inverse(rows));
    const uint32_t a2 = SpinorBasis::apply(b.L, alpha);
    if (a2 != 0 && a2 != (1u << top)) throw std::runtime_error("spinor basis: alpha' is not on the top bit");
    b.has_alpha = a2 != 0;
    if (b.lam_bit >= 0 && SpinorBasis::apply(b.LinvT, lam) != (1u << b.lam_bit))
        throw std::runtime_error("spinor basis: lambda' is not a unit vector");
    return b;
}

}  // namespace gaast
