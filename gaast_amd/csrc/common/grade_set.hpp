// GradeSet algebra on 64-bit masks (reference: src/grade_set.rs, a heap BitVec there).
// Bit k set <=> grade k present.  Host-only bookkeeping; must agree bit-for-bit with the
// reference's inference, including its over-approximations.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "gaast_expr.h"

namespace gaast {

inline uint64_t gs_single(int64_t k) { return (k < 0 || k > 63) ? 0 : (1ULL << k); }  // :65-71

inline uint64_t gs_range(int x, int y) {  // :74-80
    uint64_t m = 0;
    for (int i = x; i <= y && i < 64; ++i) m |= 1ULL << i;
    return m;
}

// GradeSet::mul (:305-327): grade r is produced by grades (i,j) iff |i-j| <= r <= i+j and
// r = i+j (mod 2).  No dimension cap here (the cap is applied by Builder::add_node).
inline uint64_t gs_mul(uint64_t a, uint64_t b) {
    uint64_t res = 0;
    for (int i = 0; i < 64; ++i) {
        if (!((a >> i) & 1ULL)) continue;
        for (int j = 0; j < 64; ++j) {
            if (!((b >> j) & 1ULL)) continue;
            for (int r = std::abs(i - j); r <= i + j && r < 64; r += 2) res |= 1ULL << r;
        }
    }
    return res;
}

// The five grades_to_produce closures of expr.rs:180-197
inline uint64_t gs_select(int kind, int64_t k1, int64_t k2) {
    switch (kind) {
    case GAAST_PROD_GEOMETRIC: return gs_mul(gs_single(k1), gs_single(k2));
    case GAAST_PROD_OUTER: return gs_single(k1 + k2);
    case GAAST_PROD_INNER: return (k1 == 0 || k2 == 0) ? 0 : gs_single(k1 > k2 ? k1 - k2 : k2 - k1);
    case GAAST_PROD_LCONTRACT: return gs_single(k2 - k1);
    case GAAST_PROD_RCONTRACT: return gs_single(k1 - k2);
    default: return 0;
    }
}

struct Selection {  // KVecsProductGradeSelection, base_types.rs:60-82
    int kind = GAAST_PROD_GEOMETRIC;   // GAAST_PROD_EXPLICIT => custom closure below
    gaast_select_fn fn = nullptr;
    void* user = nullptr;
    uint64_t operator()(int64_t k1, int64_t k2) const {
        return kind >= 0 ? gs_select(kind, k1, k2) : fn(k1, k2, user);
    }
};

struct Contrib {
    int k_left, k_right;
    uint64_t contribs;
};

// iter_contribs_to_product (:221-235) over iter_grade_sets_cp (:268-274)
inline std::vector<Contrib> iter_contribs(uint64_t self, const Selection& sel, uint64_t left,
                                          uint64_t right) {
    std::vector<Contrib> out;
    for (int kl = 0; kl < 64; ++kl) {
        if (!((left >> kl) & 1ULL)) continue;
        for (int kr = 0; kr < 64; ++kr) {
            if (!((right >> kr) & 1ULL)) continue;
            uint64_t c = self & sel(kl, kr);
            if (c) out.push_back({kl, kr, c});
        }
    }
    return out;
}

// parts_contributing_to_product (:239-252)
inline void parts_contributing(uint64_t self, const Selection& sel, uint64_t left, uint64_t right,
                               uint64_t* out_left, uint64_t* out_right) {
    uint64_t fl = 0, fr = 0;
    for (const Contrib& c : iter_contribs(self, sel, left, right)) {
        fl |= 1ULL << c.k_left;
        fr |= 1ULL << c.k_right;
    }
    *out_left = fl;
    *out_right = fr;
}

}  // namespace gaast
