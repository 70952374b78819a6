// Generation of a Product's component-multiplication list (the slice of the Cayley table it
// needs), in the reference's order T4: (k_left asc, k_right asc, left index asc, right index
// asc), filtered by grade(result) in contribs  (specialize.rs:132-183).
#pragma once
#include <cstdint>
#include <vector>

#include "algebra.hpp"
#include "gaast_hip.h"
#include "grade_set.hpp"

namespace gaast {

// Exact length of the list without generating it: pairs (a,b) with |a|=kl, |b|=kr sharing s
// basis vectors number C(n,kl) C(kl,s) C(n-kl,kr-s) and land in grade kl+kr-2s.
inline uint64_t comp_mul_count(int n, const std::vector<Contrib>& contribs) {
    uint64_t total = 0;
    for (const Contrib& c : contribs) {
        for (int s = 0; s <= c.k_left && s <= c.k_right; ++s) {
            int g = c.k_left + c.k_right - 2 * s;
            if (g < 0 || g > 63 || !((c.contribs >> g) & 1ULL)) continue;
            total += n_choose_k(n, c.k_left) * n_choose_k(c.k_left, s) *
                     n_choose_k(n - c.k_left, c.k_right - s);
        }
    }
    return total;
}

template <class Emit>
inline void for_each_comp_mul(const BladeTable& bt, const double* metric_diag,
                              const std::vector<Contrib>& contribs, Emit&& emit) {
    const int n = bt.n;
    for (const Contrib& c : contribs) {
        if (c.k_left > n || c.k_right > n) continue;  // grade_dim == 0: no blades
        const auto& lb = bt.blade_of[c.k_left];
        const auto& rb = bt.blade_of[c.k_right];
        for (uint32_t li = 0; li < lb.size(); ++li) {
            const uint64_t a = lb[li];
            for (uint32_t ri = 0; ri < rb.size(); ++ri) {
                const uint64_t b = rb[ri];
                const uint64_t r = a ^ b;
                const int g = __builtin_popcountll(r);
                if (!((c.contribs >> g) & 1ULL)) continue;
                gaast_comp_mul m;
                m.left_grade = uint32_t(c.k_left);
                m.left_index = li;
                m.right_grade = uint32_t(c.k_right);
                m.right_index = ri;
                m.result_grade = uint32_t(g);
                m.result_index = bt.index_of[r];
                m.coeff = blades_gp_coeff(n, metric_diag, a, b);
                emit(m);
            }
        }
    }
}

}  // namespace gaast
