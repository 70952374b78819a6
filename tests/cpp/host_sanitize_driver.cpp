// Host-side logic under AddressSanitizer + UBSan (CPU build only; no GPU, no HIP): programs are built
// through the C host API of include/gaast_expr.h, lowered with gaast::build_plan (the launch plan and
// every table the kernels index), serialized and deserialized.  Any out-of-bounds table write or
// undefined shift in the bitmask arithmetic aborts the run.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gaast_expr.h"
#include "plan.hpp"

static int failures = 0;
#define CHECK(c)                                                        \
    do {                                                                \
        if (!(c)) {                                                     \
            std::printf("CHECK failed: %s (line %d)\n", #c, __LINE__);  \
            ++failures;                                                 \
        }                                                               \
    } while (0)

static uint64_t full_mask(int n) { return (uint64_t(2) << n) - 1; }

// every handle the driver creates is released at the end, so that LeakSanitizer reports the library only
static std::vector<gaast_expr_t> handles;
static gaast_expr_t H(gaast_expr_t e) {
    handles.push_back(e);
    return e;
}
#define gaast_expr_input(...) H(gaast_expr_input(__VA_ARGS__))
#define gaast_expr_product(...) H(gaast_expr_product(__VA_ARGS__))
#define gaast_expr_add(...) H(gaast_expr_add(__VA_ARGS__))
#define gaast_expr_sub(...) H(gaast_expr_sub(__VA_ARGS__))
#define gaast_expr_g(...) H(gaast_expr_g(__VA_ARGS__))
#define gaast_expr_rev(...) H(gaast_expr_rev(__VA_ARGS__))
#define gaast_expr_ginvol(...) H(gaast_expr_ginvol(__VA_ARGS__))
#define gaast_expr_neg(...) H(gaast_expr_neg(__VA_ARGS__))
#define gaast_expr_vinv(...) H(gaast_expr_vinv(__VA_ARGS__))
#define gaast_expr_gselect_mask(...) H(gaast_expr_gselect_mask(__VA_ARGS__))

static void lower(gaast_expr_t e, int n, const double* metric, int dtype, uint32_t flags, const char* what,
                  const char* expect_step) {
    gaast_spec_t spec = gaast_expr_specialize(e, n, metric, 1 << 16);
    CHECK(spec != nullptr);
    if (!spec) return;
    gaast_program_desc desc;
    CHECK(gaast_spec_program_desc(spec, dtype, flags, &desc) == 0);
    gaast::Plan plan;
    gaast::build_plan(desc, plan);
    bool found = expect_step == nullptr;
    for (const gaast::Step& s : plan.steps)
        if (expect_step && s.name.find(expect_step) != std::string::npos) found = true;
    if (!found) std::printf("%s: no step named *%s*\n", what, expect_step);
    CHECK(found);
    // wire format round trip, then lower the decoded image too
    const size_t need = gaast_program_serialize(&desc, nullptr, 0);
    std::vector<unsigned char> buf(need);
    CHECK(gaast_program_serialize(&desc, buf.data(), buf.size()) == need);
    gaast_program_image_t img = gaast_program_deserialize(buf.data(), buf.size());
    CHECK(img != nullptr);
    if (img) {
        gaast::Plan plan2;
        gaast::build_plan(*gaast_program_image_desc(img), plan2);
        CHECK(plan2.steps.size() == plan.steps.size());
        gaast_program_image_free(img);
    }
    for (size_t cut = 0; cut < need; cut += need / 7 + 1)   // truncated images are rejected, not read past
        CHECK(gaast_program_deserialize(buf.data(), cut) == nullptr);
    gaast_spec_free(spec);
    std::printf("ok  %s (%zu steps)\n", what, plan.steps.size());
}

// ---- the tables of a PRODUCT_DENSE step, executed on the CPU --------------------------------------------------
// What the dense kernels compute, restated from the step's tables alone (operand maps with image positions and
// negate bits, out_map with its sign bit, neg_lo / neg_hi / zero_hi of the PERMUTED basis), compared with the
// reference's own comp-mul list for the same product.  Checks the host half of the basis permutation
// (plan.cpp: dense_basis_permutation): blade bijection, reordering signs, permuted metric masks.
static uint32_t vec_pos(uint32_t m) {
    const uint32_t x = m >> 4, lo = m & 15;
    return (x << 4) | ((((lo >> 2) ^ (x >> 2)) & 3) << 2) | (lo & 3);
}
static uint32_t mfma_b_pos(uint32_t m) {
    const uint32_t x = m >> 5, k = m & 31;
    const uint32_t lq = ((k & 1) << 2) | (k >> 3);
    return (x << 5) | (((lq ^ (x >> 1)) & 7) << 2) | ((k >> 1) & 3);
}

// k_gp_mfma32p's B image (plan.cpp: mfma32p_b_pos)
static uint32_t mfma32p_b_pos(uint32_t m) {
    static const int word_of_s[16] = {0, 8, 9, 1, 10, 2, 3, 11, 12, 4, 5, 13, 6, 14, 15, 7};
    const uint32_t x = m >> 5, k = m & 31, w = uint32_t(word_of_s[k >> 1]);
    const uint32_t lq = ((k & 1) << 2) | (w >> 2);
    return (x << 5) | (((lq ^ (x >> 1)) & 7) << 2) | (w & 3);
}

// k_gp_mfma16d's B image (f64; plan.cpp: mfma16d_b_pos)
static uint32_t mfma16d_b_pos(uint32_t m) {
    const uint32_t x = m >> 4, k = m & 15u;
    return (x << 4) | (k ^ (((x >> 1) & 7u) << 1));
}

// k_gp_mfma16x4<float>'s B image (plan.cpp: mfma16q_b_pos)
static uint32_t mfma16q_b_pos(uint32_t m) {
    static const int kq_of[16] = {0, 2, 2, 0, 2, 0, 0, 2, 3, 1, 1, 3, 1, 3, 3, 1};
    static const int s_of[16] = {0, 0, 1, 1, 2, 2, 3, 3, 0, 0, 1, 1, 2, 2, 3, 3};
    const uint32_t x = m >> 4, k = m & 15u;
    return (x << 4) | (uint32_t(kq_of[k] ^ int(((x >> 2) & 1u) << 1)) << 2) | uint32_t(s_of[k]);
}

// k_gp_mfma7's images (plan.cpp: mfma7_a_pos / mfma7_b_pos)
static uint32_t mfma7_a_pos(uint32_t m) { return (m & 63u) + (m >> 6) * 72u; }
static uint32_t mfma7_b_pos(uint32_t m) {
    static const int kq_of[8] = {0, 2, 2, 0, 3, 1, 1, 3};
    static const int s_of[8] = {0, 0, 1, 1, 0, 0, 1, 1};
    const uint32_t v = m >> 6, bh = (m >> 3) & 7u, k = m & 7u;
    return ((uint32_t(kq_of[k]) * 16u + v * 8u + bh) << 1) | uint32_t(s_of[k]);
}

static void dense_tables_agree_with_the_list(int n, const double* metric, int dtype, uint32_t flags, const char* what,
                                             const char* expect_step, uint64_t lmask = 0, uint64_t rmask = 0) {
    if (!lmask) lmask = full_mask(n);
    if (!rmask) rmask = full_mask(n);
    gaast_expr_t a = gaast_expr_input(0, lmask, n), b = gaast_expr_input(1, rmask, n);
    gaast_expr_t e = gaast_expr_product(a, b, GAAST_PROD_GEOMETRIC);
    gaast_spec_t spec = gaast_expr_specialize(e, n, metric, uint64_t(1) << 22);
    CHECK(spec != nullptr);
    if (!spec) return;
    gaast_program_desc desc;
    CHECK(gaast_spec_program_desc(spec, dtype, flags, &desc) == 0);
    gaast::Plan plan;
    gaast::build_plan(desc, plan);
    const gaast::Step* st = nullptr;
    for (const gaast::Step& s : plan.steps)
        if (s.kind == gaast::Step::PRODUCT_DENSE && !s.use_spinor) st = &s;
    CHECK(st != nullptr);
    if (!st) {
        std::printf("%s: no dense step\n", what);
        gaast_spec_free(spec);
        return;
    }
    CHECK(st->name.find(expect_step) != std::string::npos);
    // the kernel's algebra: the program's, or Cl(n - 1) for parity-pure operands (st->dense_n)
    const int n2 = st->dense_n ? st->dense_n : n;
    const uint32_t N = 1u << n2, NROW = 1u << n;
    const int L = st->use_mfma ? 5 : st->use_mfma7 ? 3 : st->use_mfma6 ? 0 : 4;   // (k_gp_mfma6: no lo vectors, the image position is the blade)
    // operands: a fixed pseudo-random row each (exact small integers: every sum below is exact)
    std::vector<double> lrow(NROW), rrow(NROW);
    uint64_t x = 88172645463325252ULL;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return double(int(x % 17) - 8); };
    for (uint32_t i = 0; i < NROW; ++i) lrow[i] = rnd();
    for (uint32_t i = 0; i < NROW; ++i) rrow[i] = rnd();
    // images in permuted-blade order
    std::vector<uint32_t> inv_vec(N), inv_b(N), inv_a7(256, 0);
    for (uint32_t m = 0; m < N; ++m) inv_vec[vec_pos(m)] = m;
    if (st->use_mfma7) for (uint32_t m = 0; m < N; ++m) { inv_b[mfma7_b_pos(m)] = m; inv_a7[mfma7_a_pos(m)] = m; }
    else if (st->mfma32_pairs) for (uint32_t m = 0; m < N; ++m) inv_b[mfma32p_b_pos(m)] = m;
    else if (st->use_mfma) for (uint32_t m = 0; m < N; ++m) inv_b[mfma_b_pos(m)] = m;
    else if (st->mfma16_quads) for (uint32_t m = 0; m < N; ++m) inv_b[mfma16q_b_pos(m)] = m;
    else if (st->use_mfma16d) for (uint32_t m = 0; m < N; ++m) inv_b[mfma16d_b_pos(m)] = m;
    // a general diagonal metric: the step carries w_S per loaded component (coeff / coeff_b) and 1 / w_T per permuted blade
    // (coeff_c); the test metrics have squares of powers of two, so every product below stays exact
    CHECK(st->scaled == (st->coeff_c.empty() ? 0 : 1));
    auto image = [&](const std::vector<uint32_t>& map, const std::vector<double>& row, bool right) {
        std::vector<double> img(N, 0.0);
        const std::vector<double>& scale = right ? st->coeff_b : st->coeff;
        CHECK(!st->scaled || scale.size() == map.size());
        size_t idx = 0;
        for (uint32_t w : map) {
            const uint32_t off = w & 0xffffu, pos = (w >> 16) & 0x7fffu;
            const uint32_t blade = st->use_mfma6 ? pos
                                   : st->use_mfma7 ? (right ? inv_b[pos] : inv_a7[pos])
                                   : (st->use_mfma || st->use_mfma16) ? (right ? inv_b[pos] : pos) : inv_vec[pos];
            uint32_t neg = w >> 31;
            // the image-pair kernels keep the b_hi part of (-1)^(|a_hi| |b_lo|) in the B image (the kernel supplies the
            // c_hi part): taken out again here, the plain formula below applies
            if (right && (st->use_mfma16 || st->mfma32_pairs))
                neg ^= uint32_t(__builtin_popcount(blade >> L) & __builtin_popcount(blade & ((1u << L) - 1u)) & 1);
            if (right && st->use_mfma7)   // the hi3 part only: the top vector's share is in the kernel's A and result signs
                neg ^= uint32_t(__builtin_popcount((blade >> 3) & 7u) & __builtin_popcount(blade & 7u) & 1);
            img[blade] = (neg ? -row[off] : row[off]) * (st->scaled ? scale[idx] : 1.0);
            ++idx;
        }
        return img;
    };
    const std::vector<double> A = image(st->u32_a, lrow, false), B = image(st->u32_b, rrow, true);
    std::vector<double> Cp(N, 0.0);
    const uint32_t lomask = (1u << L) - 1;
    for (uint32_t pa = 0; pa < N; ++pa)
        for (uint32_t pb = 0; pb < N; ++pb) {
            int par = 0;
            for (int p = 1; p < n2; ++p)
                if ((pa >> p) & 1u) par ^= __builtin_popcount(pb & ((1u << p) - 1u)) & 1;
            const uint32_t sh = pa & pb;
            par ^= __builtin_popcount((sh & lomask) & st->neg_lo) & 1;
            par ^= __builtin_popcount((sh >> L) & st->neg_hi) & 1;
            if ((sh >> L) & st->zero_hi) continue;
            Cp[pa ^ pb] += (par ? -1.0 : 1.0) * A[pa] * B[pb];
        }
    std::vector<double> got(NROW, 0.0);
    for (uint32_t m = 0; m < N; ++m) {
        const int32_t w = st->i32_a[m];
        if (w < 0) continue;
        got[size_t(w & 0x3fffffff)] = ((w & 0x40000000) ? -Cp[m] : Cp[m]) * (st->scaled ? st->coeff_c[m] : 1.0);
    }
    // the reference's list (specialize.rs:162-183), on graded rows (each buffer laid out by its own grade set)
    std::vector<double> want(NROW, 0.0);
    const int root = gaast_spec_root(spec);
    gaast_spec_node_info info;
    CHECK(gaast_spec_node(spec, root, &info) == 0);
    auto offsets = [&](uint64_t mask) {
        std::vector<uint64_t> off(size_t(n) + 2, 0);
        for (int k = 0; k <= n; ++k) off[size_t(k) + 1] = off[size_t(k)] + (((mask >> k) & 1ULL) ? gaast_n_choose_k(uint64_t(n), uint64_t(k)) : 0);
        return off;
    };
    const std::vector<uint64_t> loff = offsets(lmask), roff = offsets(rmask), ooff = offsets(info.minimal_grade_mask);
    const gaast_comp_mul* muls = gaast_spec_comp_muls(spec, root);
    CHECK(muls != nullptr && (lmask != full_mask(n) || rmask != full_mask(n) || info.n_comp_muls == (uint64_t(1) << (2 * n))));
    if (muls)
        for (uint64_t i = 0; i < info.n_comp_muls; ++i) {
            const gaast_comp_mul& m = muls[i];
            want[ooff[m.result_grade] + m.result_index] += lrow[loff[m.left_grade] + m.left_index] * rrow[roff[m.right_grade] + m.right_index] * m.coeff;
        }
    size_t bad = 0;
    for (uint32_t i = 0; i < NROW; ++i) bad += got[i] != want[i];
    if (bad) std::printf("%s: %zu of %u components differ\n", what, bad, NROW);
    CHECK(bad == 0);
    gaast_spec_free(spec);
    std::printf("ok  %s (%s)\n", what, st->name.c_str());
}

// The list chain specialised per program (plan.cpp: make_chain_jit): its tables are a TRANSFORMATION of the generic chain's
// (byte offsets from the item's base in a new LDS layout, list 1's signs folded into a negated image).  Both table sets are
// executed here, in plain C++ doubles in the kernels' order, on the same random rows: every result bit must agree.  With a
// directory argument the generated kernel source is written there (tests/test_chain_jit.py compiles it for gfx950).
// (flags = GAAST_FLAG_EXACT_ORDER: the tables of the reference order, compared bit for bit; 0: the tolerance mode's sign-sorted table of
//  list 2 -- plus terms, then minus terms, per (row, slice) -- summed in ITS order and compared within 1e-12)
static void chain_tables_agree(int n, const double* metric, int dtype, const char* dump_dir, uint32_t flags = GAAST_FLAG_EXACT_ORDER) {
    gaast_expr_t r = gaast_expr_input(0, 0x5555555555555555ull & full_mask(n), n), x = gaast_expr_input(1, 0x2, n);
    gaast_expr_t e = gaast_expr_g(gaast_expr_product(gaast_expr_product(r, x, GAAST_PROD_GEOMETRIC), gaast_expr_rev(r), GAAST_PROD_GEOMETRIC), 1);
    gaast_spec_t spec = gaast_expr_specialize(e, n, metric, 1 << 16);
    CHECK(spec != nullptr);
    if (!spec) return;
    gaast_program_desc desc;
    CHECK(gaast_spec_program_desc(spec, dtype, flags, &desc) == 0);
    gaast::Plan plan;
    gaast::build_plan(desc, plan);
    CHECK(plan.steps.size() == 1 && plan.steps[0].list_chain && plan.steps[0].chain_jit == 1);
    if (plan.steps.size() != 1 || plan.steps[0].chain_jit != 1) {
        gaast_spec_free(spec);
        return;
    }
    const gaast::Step& s = plan.steps[0];
    if (dump_dir) {
        char path[512];
        std::snprintf(path, sizeof path, "%s/chain_n%d_%s.hip", dump_dir, n, dtype == GAAST_F32 ? "f32" : "f64");
        if (FILE* f = std::fopen(path, "w")) {
            std::fputs(s.chain_jit_source.c_str(), f);
            std::fclose(f);
        }
    }
    const int esz = dtype == GAAST_F32 ? 4 : 8;
    const int l1 = s.pre_left_len, r1 = s.pre_right_len, mid = s.chain_mid_len;
    const int rows1 = int(s.pre_row_map.size()), w1 = s.pre_width, rows2 = int(s.u32_b.size()), w2 = s.ell_width;
    const int w1p = s.cj_fmt[0], w2p = (w2 + 3) & ~3, wide = s.cj_fmt[1] == 2 ? 2 : 1;
    const bool sorted = s.cj_fmt[1] >= 3;
    // (sign-sorted only without the flag, and only when the rows' signs are balanced: R^{6,3} at n = 9 is, the Euclidean n = 8, 10 are not)
    CHECK(!sorted || !(flags & GAAST_FLAG_EXACT_ORDER));
    if (!(flags & GAAST_FLAG_EXACT_ORDER) && n == 9) CHECK(sorted);
    const int wss = s.cj_sorted[0] + s.cj_sorted[1];
    CHECK(int(s.cj_ent1.size()) == rows1 * w1p && s.chain_alias == 1 && s.list_chain == 1);
    CHECK(int(s.cj_ent2.size()) == (sorted ? rows2 * s.cj_split * wss : rows2 * w2p * wide));
    std::vector<double> L(static_cast<size_t>(l1), 0.0), X(static_cast<size_t>(r1), 0.0);
    unsigned long long seed = 88172645463325252ull + unsigned(n);
    auto rnd = [&]() {
        seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17;
        return double(int64_t(seed >> 11)) / double(1ull << 52) - 1.0;
    };
    for (double& v : L) v = rnd();
    for (double& v : X) v = rnd();
    // (a) the generic tables: [term][row] words, byte offsets from each operand row, sign in bit 31
    std::vector<double> mid_a(size_t(mid), 0.0), out_a(size_t(rows2), 0.0);
    for (int row = 0; row < rows1; ++row) {
        double acc = 0.0;
        for (int t = 0; t < w1; ++t) {
            const uint32_t w = s.pre_entries[size_t(t) * rows1 + row];
            const double p = L[(w & 0x7fffu) / esz] * X[((w >> 16) & 0x7fffu) / esz];
            acc = acc + ((w & 0x80000000u) ? -p : p);
        }
        mid_a[s.pre_row_map[size_t(row)]] = acc;
    }
    for (int row = 0; row < rows2; ++row) {
        double acc = 0.0;
        for (int t = 0; t < w2; ++t) {
            const uint32_t w = s.u32_c[size_t(t) * rows2 + row];
            const double p = mid_a[(w & 0x7fffu) / esz] * L[((w >> 16) & 0x7fffu) / esz];
            acc = acc + ((w & 0x80000000u) ? -p : p);
        }
        out_a[size_t(row)] = acc;
    }
    // (b) the specialised kernel's tables over its item image
    const int* lay = s.cj_layout;
    std::vector<double> img(size_t(lay[5]), 0.0);
    for (int c = 0; c < l1; ++c) img[size_t(lay[0] + c)] = L[size_t(c)];
    for (int c = 0; c < r1; ++c) img[size_t(lay[1] + c)] = X[size_t(c)];
    for (int c = 0; c < (lay[6] ? l1 : r1); ++c) img[size_t(lay[2] + c)] = -(lay[6] ? L[size_t(c)] : X[size_t(c)]);
    auto at = [&](uint32_t byte_off) -> double& {
        CHECK(byte_off % esz == 0 && byte_off / esz < uint32_t(lay[5]));
        return img[byte_off / esz];
    };
    for (int row = 0; row < rows1; ++row) {
        double acc = 0.0;
        if (s.cj_xreg) {   // tolerance mode: term j multiplies the right operand's component j (a register); word = left offset | sign << 31
            CHECK(!(flags & GAAST_FLAG_EXACT_ORDER) && r1 <= w1p);
            int real_terms = 0;
            for (int j = 0; j < r1; ++j) {
                const uint32_t w = s.cj_ent1[size_t(row) * w1p + j];
                const double pr = at(w & 0x7fffffffu) * img[size_t(lay[1] + j)];
                acc = (w & 0x80000000u) ? acc - pr : acc + pr;
                real_terms += (w & 0x7fffffffu) == uint32_t(s.cj_sorted[2]) ? 0 : 1;
            }
            CHECK(real_terms == w1);
            CHECK(std::fabs(acc - mid_a[s.pre_row_map[size_t(row)]]) <= 1e-12 * (std::fabs(acc) + 1.0));
        } else {
            for (int t = 0; t < w1; ++t) {
                const uint32_t w = s.cj_ent1[size_t(row) * w1p + t];
                acc = acc + at(w & 0xffffu) * at(w >> 16);
            }
        }
        at(s.cj_pos1[size_t(row)]) = acc;
    }
    bool same = true;
    for (int row = 0; row < rows2 && sorted; ++row) {   // slices in order, each plus terms then minus terms; every real term exactly once
        double acc = 0.0, mag = 0.0;
        int real_terms = 0;
        for (int sl = 0; sl < s.cj_split; ++sl) {
            double part = 0.0;
            for (int t = 0; t < wss; ++t) {
                const uint32_t w = s.cj_ent2[(size_t(row) * s.cj_split + sl) * wss + t];
                const bool pad = (w & 0xffffu) == uint32_t(s.cj_sorted[2]) && (w >> 16) == uint32_t(s.cj_sorted[2]);
                const double pr = at(w & 0xffffu) * at(w >> 16);
                part = t < s.cj_sorted[0] ? part + pr : part - pr;
                mag += std::fabs(pr);
                real_terms += pad ? 0 : 1;
                CHECK(!pad || pr == 0.0);
            }
            acc = acc + part;
        }
        CHECK(real_terms == w2);
        same = same && std::fabs(acc - out_a[size_t(row)]) <= 1e-12 * (mag + 1.0) && s.cj_out2[size_t(row)] == s.u32_b[size_t(row)];
    }
    for (int row = 0; row < rows2 && !sorted; ++row) {
        double acc = 0.0;
        for (int t = 0; t < w2; ++t) {
            const uint32_t w = s.cj_ent2[(size_t(row) * w2p + t) * wide];
            const uint32_t sign = wide == 2 ? s.cj_ent2[(size_t(row) * w2p + t) * 2 + 1] : (w & 0x80000000u);
            CHECK(sign == 0u || sign == 0x80000000u);
            const double p = at(w & 0xffffu) * at(wide == 2 ? (w >> 16) : ((w >> 16) & 0x7fffu));
            acc = acc + (sign ? -p : p);
        }
        // (list 1 re-ordered by right index leaves the mid row within rounding of the reference's: list 2 then agrees within a tolerance)
        const bool agree = s.cj_xreg ? std::fabs(acc - out_a[size_t(row)]) <= 1e-11 * (std::fabs(acc) + 1.0)
                                     : std::memcmp(&acc, &out_a[size_t(row)], sizeof acc) == 0;
        same = same && agree && s.cj_out2[size_t(row)] == s.u32_b[size_t(row)];
    }
    if (!same) std::printf("chain tables n=%d: the specialised tables compute another result\n", n);
    CHECK(same);
    gaast_spec_free(spec);
}

// ... and a SINGLE list with few long rows on the same kernel (plan.cpp: jit_long_row_lists): d = (a + b * c).g(2), the covering copy of
// a's grade 2 folded into the list's accumulators.  Generic ELL words against the specialised tables, bit for bit.
static void single_list_tables_agree(int n, int dtype, const char* dump_dir, uint32_t flags = GAAST_FLAG_EXACT_ORDER) {
    std::vector<double> metric(size_t(n), 1.0);
    gaast_expr_t a = gaast_expr_input(0, full_mask(n), n), b = gaast_expr_input(1, full_mask(n), n), c = gaast_expr_input(2, full_mask(n), n);
    gaast_expr_t e = gaast_expr_g(gaast_expr_add(a, gaast_expr_product(b, c, GAAST_PROD_GEOMETRIC)), 2);
    gaast_spec_t spec = gaast_expr_specialize(e, n, metric.data(), uint64_t(1) << 22);
    CHECK(spec != nullptr);
    if (!spec) return;
    gaast_program_desc desc;
    CHECK(gaast_spec_program_desc(spec, dtype, flags, &desc) == 0);
    gaast::Plan plan;
    gaast::build_plan(desc, plan);
    CHECK(plan.steps.size() == 2 && plan.steps[0].kind == gaast::Step::AXPY && plan.steps[0].beta == 0 && plan.steps[1].list_jit == 1 &&
          plan.steps[1].fold_prev == 1 && plan.steps[1].chain_jit == 1);
    if (plan.steps.size() != 2 || !plan.steps[1].list_jit) {
        for (const gaast::Step& st : plan.steps) std::printf("    step %s\n", st.name.c_str());
        gaast_spec_free(spec);
        return;
    }
    const gaast::Step& ax = plan.steps[0];
    const gaast::Step& s = plan.steps[1];
    if (dump_dir) {
        char path[512];
        std::snprintf(path, sizeof path, "%s/list_n%d_%s.hip", dump_dir, n, dtype == GAAST_F32 ? "f32" : "f64");
        if (FILE* f = std::fopen(path, "w")) {
            std::fputs(s.chain_jit_source.c_str(), f);
            std::fclose(f);
        }
    }
    const int esz = dtype == GAAST_F32 ? 4 : 8;
    const int N = 1 << n, rows = int(s.u32_b.size()), w2 = s.ell_width, w2p = (w2 + 3) & ~3, wide = s.cj_fmt[1] == 2 ? 2 : 1;
    std::vector<double> A(static_cast<size_t>(N), 0.0), B(static_cast<size_t>(N), 0.0), Cc(static_cast<size_t>(N), 0.0);
    unsigned long long seed = 1234567ull + unsigned(n);
    auto rnd = [&]() {
        seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17;
        return double(int64_t(seed >> 11)) / double(1ull << 52) - 1.0;
    };
    for (double& v : A) v = rnd();
    for (double& v : B) v = rnd();
    for (double& v : Cc) v = rnd();
    // (a) the two-launch plan: copy, then the generic ELL words accumulate onto it
    std::vector<double> out_a(size_t(rows), 0.0);
    for (uint32_t m : ax.u32_a) out_a[m & 0xffffu] = 0.0 + A[m >> 16];
    for (int row = 0; row < rows; ++row) {
        double acc = out_a[s.u32_b[size_t(row)]];
        for (int t = 0; t < w2; ++t) {
            const uint32_t w = s.u32_c[size_t(t) * rows + row];
            const double p = B[(w & 0x7fffu) / esz] * Cc[((w >> 16) & 0x7fffu) / esz];
            acc = acc + ((w & 0x80000000u) ? -p : p);
        }
        out_a[s.u32_b[size_t(row)]] = acc;
    }
    // (b) the specialised tables over the item image: left operand at cj_layout[3] ("mid"), right one at cj_layout[4]
    const int* lay = s.cj_layout;
    std::vector<double> img(size_t(lay[5]), 0.0);
    for (int c2 = 0; c2 < N; ++c2) img[size_t(lay[3] + c2)] = B[size_t(c2)];
    for (int c2 = 0; c2 < N; ++c2) img[size_t(lay[4] + c2)] = Cc[size_t(c2)];
    bool same = int(s.cj_pos1.size()) == rows;
    const bool sorted = s.cj_fmt[1] >= 3;
    CHECK(!sorted || !(flags & GAAST_FLAG_EXACT_ORDER));
    if (!(flags & GAAST_FLAG_EXACT_ORDER) && n == 8 && dtype == GAAST_F64) CHECK(sorted);   // 136 + 136 terms for 256
    if (n == 12) CHECK(!sorted);                                                              // 2 x 32 KiB of operands leave no room for the zero element
    const int wss = s.cj_sorted[0] + s.cj_sorted[1];
    for (int row = 0; row < rows && same && sorted; ++row) {
        double acc = 0.0 + A[s.cj_pos1[size_t(row)]], mag = 1.0;
        int real_terms = 0;
        for (int sl = 0; sl < s.cj_split; ++sl) {
            double part = 0.0;
            for (int t = 0; t < wss; ++t) {
                const uint32_t w = s.cj_ent2[(size_t(row) * s.cj_split + sl) * wss + t];
                const uint32_t mo = w & 0xffffu, oo = w >> 16;
                CHECK(mo % esz == 0 && oo % esz == 0 && mo / esz < uint32_t(lay[5]) && oo / esz < uint32_t(lay[5]));
                const bool pad = mo == uint32_t(s.cj_sorted[2]) && oo == uint32_t(s.cj_sorted[2]);
                const double pr = img[mo / esz] * img[oo / esz];
                part = t < s.cj_sorted[0] ? part + pr : part - pr;
                mag += std::fabs(pr);
                real_terms += pad ? 0 : 1;
                CHECK(!pad || pr == 0.0);
            }
            acc = acc + part;
        }
        CHECK(real_terms == w2);
        same = std::fabs(acc - out_a[s.cj_out2[size_t(row)]]) <= 1e-12 * mag;
    }
    for (int row = 0; row < rows && same && !sorted; ++row) {
        double acc = 0.0 + A[s.cj_pos1[size_t(row)]];
        for (int t = 0; t < w2; ++t) {
            const uint32_t w = s.cj_ent2[(size_t(row) * w2p + t) * wide];
            const uint32_t sign = wide == 2 ? s.cj_ent2[(size_t(row) * w2p + t) * 2 + 1] : (w & 0x80000000u);
            const uint32_t mo = w & 0xffffu, oo = wide == 2 ? (w >> 16) : ((w >> 16) & 0x7fffu);
            CHECK(mo % esz == 0 && oo % esz == 0 && mo / esz < uint32_t(lay[5]) && oo / esz < uint32_t(lay[5]));
            const double p = img[mo / esz] * img[oo / esz];
            acc = acc + (sign ? -p : p);
        }
        same = std::memcmp(&acc, &out_a[s.cj_out2[size_t(row)]], sizeof acc) == 0;
    }
    if (!same) std::printf("single list n=%d: the specialised tables compute another result\n", n);
    CHECK(same);
    gaast_spec_free(spec);
}

int main(int argc, char** argv) {
    const char* dump_dir = argc > 1 ? argv[1] : nullptr;
    single_list_tables_agree(8, GAAST_F64, dump_dir);
    single_list_tables_agree(8, GAAST_F32, dump_dir);
    single_list_tables_agree(9, GAAST_F64, dump_dir);
    single_list_tables_agree(12, GAAST_F64, dump_dir);
    single_list_tables_agree(8, GAAST_F64, dump_dir, 0);    // tolerance mode: the sign-sorted table
    single_list_tables_agree(8, GAAST_F32, dump_dir, 0);
    single_list_tables_agree(12, GAAST_F64, dump_dir, 0);
    {
        const double euclid16[16] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
        const double mixed9[9] = {1, 1, 1, 1, 1, 1, -1, -1, -1};
        chain_tables_agree(8, euclid16, GAAST_F64, dump_dir);
        chain_tables_agree(9, mixed9, GAAST_F64, dump_dir);
        chain_tables_agree(10, euclid16, GAAST_F64, dump_dir);
        chain_tables_agree(9, mixed9, GAAST_F32, dump_dir);
        chain_tables_agree(12, euclid16, GAAST_F64, dump_dir);
        chain_tables_agree(8, euclid16, GAAST_F64, dump_dir, 0);    // tolerance mode: the sign-sorted table of list 2
        chain_tables_agree(9, mixed9, GAAST_F64, dump_dir, 0);
        chain_tables_agree(10, euclid16, GAAST_F64, dump_dir, 0);
        chain_tables_agree(9, mixed9, GAAST_F32, dump_dir, 0);
    }
    const double euclid[16] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
    const double cga[5] = {1, 1, 1, 1, -1};
    const double mixed12[12] = {1, 1, -1, 1, -1, -1, -1, -1, 1, 1, -1, 1};      // lambda on bit 4
    const double alt12[12] = {1, -1, 1, -1, 1, -1, 1, -1, 1, -1, 1, -1};        // lambda = 0
    {   // BASELINE config 1: (a + b*c).g(2) in R^3
        gaast_expr_t a = gaast_expr_input(0, full_mask(3), 3), b = gaast_expr_input(1, full_mask(3), 3),
                     c = gaast_expr_input(2, full_mask(3), 3);
        gaast_expr_t e = gaast_expr_g(gaast_expr_add(a, gaast_expr_product(b, c, GAAST_PROD_GEOMETRIC)), 2);
        lower(e, 3, euclid, GAAST_F64, 0, "cfg1", "ast_");
        lower(e, 3, euclid, GAAST_F64, GAAST_FLAG_NO_FUSION, "cfg1 unfused", "product_csr");
    }
    {   // config 5: R X ~R in R^{4,1}
        gaast_expr_t r = gaast_expr_input(0, 0x15, 5), x = gaast_expr_input(1, 0x2, 5);
        gaast_expr_t e = gaast_expr_product(gaast_expr_product(r, x, GAAST_PROD_GEOMETRIC), gaast_expr_rev(r), GAAST_PROD_GEOMETRIC);
        lower(e, 5, cga, GAAST_F64, 0, "cfg5 sandwich", "ast_");
        lower(e, 5, cga, GAAST_F32, GAAST_FLAG_NO_JIT, "cfg5 sandwich f32", "ast_");
    }
    for (int n : {6, 7, 8, 9, 10, 11, 12, 13, 14}) {   // dense products (n = 14: matrix-core kernel only, f32): vector / matrix-core / matrix-representation tables
        gaast_expr_t a = gaast_expr_input(0, full_mask(n), n), b = gaast_expr_input(1, full_mask(n), n);
        gaast_expr_t e = gaast_expr_product(a, b, GAAST_PROD_GEOMETRIC);
        char what[64];
        std::snprintf(what, sizeof what, "dense gp n=%d f32", n);
        lower(e, n, euclid, GAAST_F32, 0, what, "product_dense");
        if (n <= 12) {
            std::snprintf(what, sizeof what, "dense gp n=%d f64", n);
            lower(e, n, euclid, GAAST_F64, 0, what, "product_dense");
        }
        if (n >= 7 && n <= 12) {
            std::snprintf(what, sizeof what, "matrix representation n=%d f32", n);
            lower(e, n, euclid, GAAST_F32, GAAST_FLAG_SPINOR_GEMM, what, "product_spinor_gemm");
        }
    }
    {
        gaast_expr_t a = gaast_expr_input(0, full_mask(12), 12), b = gaast_expr_input(1, full_mask(12), 12);
        gaast_expr_t e = gaast_expr_product(a, b, GAAST_PROD_GEOMETRIC);
        lower(e, 12, mixed12, GAAST_F32, GAAST_FLAG_SPINOR_GEMM, "matrix representation n=12 lambda on bit 4", "lam=4");
        lower(e, 12, alt12, GAAST_F64, GAAST_FLAG_SPINOR_GEMM, "matrix representation n=12 f64 lambda = 0", "lam=-1");
        // partial operands and result: even * full -> odd grades
        gaast_expr_t ev = gaast_expr_input(0, 0x1555, 12);
        gaast_expr_t e2 = gaast_expr_gselect_mask(gaast_expr_product(gaast_expr_rev(ev), b, GAAST_PROD_GEOMETRIC), 0x0aaa);
        lower(e2, 12, euclid, GAAST_F32, GAAST_FLAG_SPINOR_GEMM, "matrix representation n=12 partial", "product_spinor_gemm");
        lower(e2, 12, euclid, GAAST_F32, 0, "dense n=12 partial", "product_dense");
    }
    for (int n : {7, 8, 9}) {   // reference-order dense products: CSR -> transposed [term][row] list (k_product_ell)
        gaast_expr_t a = gaast_expr_input(0, full_mask(n), n), b = gaast_expr_input(1, full_mask(n), n);
        gaast_expr_t e = gaast_expr_product(a, b, GAAST_PROD_GEOMETRIC);
        char what[64];
        std::snprintf(what, sizeof what, "exact order n=%d f64", n);
        lower(e, n, euclid, GAAST_F64, GAAST_FLAG_EXACT_ORDER, what, "product_ell");
        std::snprintf(what, sizeof what, "exact order n=%d f32", n);
        lower(e, n, euclid, GAAST_F32, GAAST_FLAG_EXACT_ORDER, what, "product_ell");
        // rows of 1 + n + C(n,2) entries: not a multiple of the kernel's chunk
        gaast_expr_t lowg = gaast_expr_input(2, 0x7, n);
        lower(gaast_expr_product(lowg, b, GAAST_PROD_GEOMETRIC), n, euclid, GAAST_F64, GAAST_FLAG_EXACT_ORDER, "exact order, odd width", "product_ell");
    }
    {   // degenerate metric: 0.0 coefficients stay in the CSR list
        const double pga7[7] = {0, 1, 1, 1, 1, 1, 1};
        gaast_expr_t a = gaast_expr_input(0, full_mask(7), 7), b = gaast_expr_input(1, full_mask(7), 7);
        lower(gaast_expr_product(a, b, GAAST_PROD_GEOMETRIC), 7, pga7, GAAST_F64, GAAST_FLAG_EXACT_ORDER, "exact order, degenerate", "product_csr");
    }
    {   // fused for the specialised kernel only (slab too big for the LDS interpreter)
        gaast_expr_t a = gaast_expr_input(0, full_mask(5), 5), b = gaast_expr_input(1, full_mask(5), 5);
        lower(gaast_expr_product(a, b, GAAST_PROD_GEOMETRIC), 5, euclid, GAAST_F64, 0, "r5 f64, JIT-only fusion", "ast_");
        lower(gaast_expr_product(a, b, GAAST_PROD_GEOMETRIC), 5, euclid, GAAST_F64, GAAST_FLAG_NO_JIT, "r5 f64 without JIT", "product_ell");
    }
    {   // the other products and unary arms on R^4
        gaast_expr_t a = gaast_expr_input(0, full_mask(4), 4), b = gaast_expr_input(1, full_mask(4), 4);
        for (int kind : {GAAST_PROD_OUTER, GAAST_PROD_INNER, GAAST_PROD_LCONTRACT, GAAST_PROD_RCONTRACT})
            lower(gaast_expr_product(a, b, kind), 4, euclid, GAAST_F64, 0, "r4 product kinds", nullptr);
        lower(gaast_expr_sub(gaast_expr_ginvol(a), gaast_expr_neg(gaast_expr_rev(b))), 4, euclid, GAAST_F64, 0, "unary chain", nullptr);
        lower(gaast_expr_vinv(gaast_expr_input(2, 0x2, 4)), 4, euclid, GAAST_F64, 0, "vinv", nullptr);
    }
    {   // exp / log extension: the step's tables (blade squares, commuting pairs) for simple and non-simple grades
        gaast_expr_t b = gaast_expr_input(0, 0x4, 5), x = gaast_expr_input(1, 0x2, 5);
        gaast_expr_t r = H(gaast_expr_exp(b));
        gaast_expr_t sw = gaast_expr_product(gaast_expr_product(r, x, GAAST_PROD_GEOMETRIC), gaast_expr_rev(r), GAAST_PROD_GEOMETRIC);
        lower(sw, 5, cga, GAAST_F64, GAAST_FLAG_EXP_LOG, "exp(B) x ~exp(B), fused", "ast_");
        lower(sw, 5, cga, GAAST_F64, GAAST_FLAG_EXP_LOG | GAAST_FLAG_NO_FUSION, "exp(B) x ~exp(B), unfused", "exponential[grade 2, 10 components, 15 domain-check pairs]");
        lower(H(gaast_expr_log(r)), 5, cga, GAAST_F32, GAAST_FLAG_EXP_LOG | GAAST_FLAG_NO_FUSION, "log(exp(B))", "logarithm[grade 2");
        lower(H(gaast_expr_exp(gaast_expr_input(2, 0x2, 4))), 4, euclid, GAAST_F64, GAAST_FLAG_EXP_LOG | GAAST_FLAG_NO_FUSION, "exp(vector)", "0 domain-check pairs");
        lower(sw, 5, cga, GAAST_F64, 0, "exp without the flag: plan records UNIMPLEMENTED", nullptr);
    }
    {   // dense tables under a basis permutation == the reference's list, for every kernel's table format
        const double pga6[6] = {0, 1, 1, 1, 1, -1}, anti6[6] = {-1, 1, -1, -1, -1, 1}, sta6[6] = {-1, 1, 1, 1, 0, 1};
        dense_tables_agree_with_the_list(6, euclid, GAAST_F64, GAAST_FLAG_NO_MFMA, "dense tables n=6 euclid", "product_dense[gp n=6]");
        dense_tables_agree_with_the_list(6, pga6, GAAST_F64, GAAST_FLAG_NO_MFMA, "dense tables n=6 null vector first", "permuted basis");
        dense_tables_agree_with_the_list(6, anti6, GAAST_F64, GAAST_FLAG_NO_MFMA, "dense tables n=6 four negative lo vectors", "permuted basis");
        dense_tables_agree_with_the_list(6, sta6, GAAST_F32, GAAST_FLAG_NO_MFMA, "dense tables n=6 time first", "permuted basis");
        // k_gp_mfma6 (round 4): the basis as it stands, the image position is the blade, all six signature bits in neg_hi / zero_hi
        dense_tables_agree_with_the_list(6, euclid, GAAST_F64, 0, "mfma6 tables n=6 euclid", "product_dense_mfma[gp n=6]");
        dense_tables_agree_with_the_list(6, pga6, GAAST_F64, 0, "mfma6 tables n=6 null vector first", "product_dense_mfma[gp n=6]");
        dense_tables_agree_with_the_list(6, anti6, GAAST_F32, 0, "mfma6 tables n=6 four negative vectors", "product_dense_mfma[gp n=6]");
        dense_tables_agree_with_the_list(6, sta6, GAAST_F32, 0, "mfma6 tables n=6 time first, a null vector", "product_dense_mfma[gp n=6]");
        const double pga8[8] = {0, 1, 1, 1, 1, 1, 1, 1}, neg8[8] = {-1, -1, -1, -1, -1, -1, -1, -1}, mix8[8] = {1, 0, -1, 1, 0, -1, 1, -1};
        dense_tables_agree_with_the_list(8, pga8, GAAST_F32, 0, "mfma16 tables n=8 null vector first", "product_dense_mfma");
        dense_tables_agree_with_the_list(8, neg8, GAAST_F32, 0, "mfma16 tables Cl(0,8)", "product_dense_mfma[gp n=8]");
        dense_tables_agree_with_the_list(8, mix8, GAAST_F32, 0, "mfma16 tables n=8 mixed", "permuted basis");
        dense_tables_agree_with_the_list(8, neg8, GAAST_F64, GAAST_FLAG_NO_MFMA, "vector tables Cl(0,8)", "product_dense[gp n=8]");
        dense_tables_agree_with_the_list(8, neg8, GAAST_F64, 0, "mfma16d tables Cl(0,8)", "product_dense_mfma[gp n=8]");
        dense_tables_agree_with_the_list(8, mix8, GAAST_F64, 0, "mfma16d tables n=8 mixed", "permuted basis");
        dense_tables_agree_with_the_list(9, euclid, GAAST_F64, 0, "mfma16d tables n=9 euclid", "product_dense_mfma[gp n=9]");
        const double mix10[10] = {0, -1, 1, 1, -1, 1, 0, 1, -1, 1};
        dense_tables_agree_with_the_list(10, mix10, GAAST_F32, 0, "mfma32 tables n=10 mixed", "product_dense_mfma[gp n=10 permuted basis]");
        dense_tables_agree_with_the_list(10, euclid, GAAST_F32, 0, "mfma32 tables n=10 euclid", "product_dense_mfma[gp n=10]");
        dense_tables_agree_with_the_list(10, mix10, GAAST_F64, 0, "mfma16d tables n=10 mixed", "product_dense_mfma[gp n=10 permuted basis]");
        // general diagonal metrics (algebra.rs:148-165: any base_vec_dot): the rescaled basis.  Squares of powers of two keep
        // every factor, product and sum exact, so the tables must reproduce the reference's list to the last bit.
        const double gen8[8] = {4, 0.25, -16, 1, 1, -0.0625, 1, 64};
        const double gen8z[8] = {0, 4, -0.25, 1, 16, 0, -1, 0.0625};
        const double gen10[10] = {0.25, -4, 1, 16, -1, 1, 0, 0.0625, -16, 4};
        const double gen7[7] = {4, 4, 0.25, 1, 16, -0.25, 1};
        dense_tables_agree_with_the_list(8, gen8, GAAST_F32, 0, "mfma16x4 f32 tables n=8 general metric", "rescaled basis");
        dense_tables_agree_with_the_list(8, gen8, GAAST_F64, 0, "mfma16x4 f64 tables n=8 general metric", "rescaled basis");
        dense_tables_agree_with_the_list(8, gen8z, GAAST_F64, 0, "mfma16x4 f64 tables n=8 general metric with null vectors", "rescaled basis");
        dense_tables_agree_with_the_list(8, gen8, GAAST_F64, GAAST_FLAG_NO_MFMA, "vector tables n=8 general metric", "rescaled basis");
        dense_tables_agree_with_the_list(7, gen7, GAAST_F32, 0, "mfma7 tables n=7 general metric", "rescaled basis");
        dense_tables_agree_with_the_list(7, gen7, GAAST_F64, GAAST_FLAG_NO_MFMA, "vector tables n=7 general metric", "rescaled basis");
        dense_tables_agree_with_the_list(7, euclid, GAAST_F64, 0, "mfma7 tables n=7 euclid", "product_dense_mfma[gp n=7]");
        dense_tables_agree_with_the_list(7, mix8, GAAST_F32, 0, "mfma7 tables n=7 mixed", "product_dense_mfma[gp n=7");
        dense_tables_agree_with_the_list(7, pga8, GAAST_F64, 0, "mfma7 tables n=7 null vector first", "permuted basis");
        dense_tables_agree_with_the_list(8, mix8, GAAST_F64, 0, "parity-pure n=8 -> mfma7 even x odd", "even x odd in Cl(7)", 0x155, 0x0aa);
        dense_tables_agree_with_the_list(10, gen10, GAAST_F32, 0, "mfma32p tables n=10 general metric", "rescaled basis");
        dense_tables_agree_with_the_list(10, gen10, GAAST_F64, 0, "mfma16x4 f64 tables n=10 general metric", "rescaled basis");
        // parity-pure operands: one product in the even subalgebra Cl(n - 1) (plan.cpp: parity_reduced_frame), all four
        // parity cases, Euclidean / mixed / degenerate / general metrics, the pivot not always the last vector
        const uint64_t EV = 0x5555555555555555ULL, OD = 0xAAAAAAAAAAAAAAAAULL;
        const double mix9[9] = {1, -1, 1, 1, -1, 1, -1, 1, -1};
        const double deg9[9] = {1, 0, -1, 1, 1, 1, -1, 1, 0};          // the last vector is null: the pivot is e_8
        const double gen9[9] = {4, 1, -0.25, 1, 16, 1, -1, 0.25, -4};
        struct { uint64_t l, r; const char* tag; } cases[4] = {{EV, EV, "even x even"}, {EV, OD, "even x odd"}, {OD, EV, "odd x even"}, {OD, OD, "odd x odd"}};
        for (auto& pc : cases) {
            char what[96];
            std::snprintf(what, sizeof what, "parity-pure n=9 euclid f64 %s", pc.tag);
            dense_tables_agree_with_the_list(9, euclid, GAAST_F64, 0, what, pc.tag, pc.l & full_mask(9), pc.r & full_mask(9));
            std::snprintf(what, sizeof what, "parity-pure n=9 mixed f32 %s", pc.tag);
            dense_tables_agree_with_the_list(9, mix9, GAAST_F32, 0, what, pc.tag, pc.l & full_mask(9), pc.r & full_mask(9));
            std::snprintf(what, sizeof what, "parity-pure n=9 degenerate f64 %s", pc.tag);
            dense_tables_agree_with_the_list(9, deg9, GAAST_F64, 0, what, pc.tag, pc.l & full_mask(9), pc.r & full_mask(9));
            std::snprintf(what, sizeof what, "parity-pure n=9 general metric f64 %s", pc.tag);
            dense_tables_agree_with_the_list(9, gen9, GAAST_F64, 0, what, pc.tag, pc.l & full_mask(9), pc.r & full_mask(9));
            std::snprintf(what, sizeof what, "parity-pure n=8 vector kernel f32 %s", pc.tag);
            dense_tables_agree_with_the_list(8, mix9, GAAST_F32, 0, what, pc.tag, pc.l & full_mask(8), pc.r & full_mask(8));
        }
        const double mix11[11] = {1, -1, 1, 1, 0, 1, -1, 1, -1, 1, 1};
        dense_tables_agree_with_the_list(11, mix11, GAAST_F32, 0, "parity-pure n=11 -> mfma32p<10> odd x even", "odd x even", OD & full_mask(11), EV & full_mask(11));
    }
    {   // chained products: R X ~R at n = 8, 9, 10 -- the sparse product R X moves into the dense step's LDS staging
        const double m63[9] = {1, 1, 1, 1, 1, 1, -1, -1, -1};
        for (int n : {8, 9, 10}) {
            uint64_t even = 0;
            for (int k = 0; k <= n; k += 2) even |= uint64_t(1) << k;
            gaast_expr_t r = gaast_expr_input(0, even, n), x = gaast_expr_input(1, 0x2, n);
            gaast_expr_t e = gaast_expr_product(gaast_expr_product(r, x, GAAST_PROD_GEOMETRIC), gaast_expr_rev(r), GAAST_PROD_GEOMETRIC);
            char what[64];
            std::snprintf(what, sizeof what, "sandwich n=%d chained", n);
            lower(e, n, n == 9 ? m63 : euclid, GAAST_F64, 0, what, "<- product_csr");
            std::snprintf(what, sizeof what, "sandwich n=%d two launches", n);
            lower(e, n, n == 9 ? m63 : euclid, GAAST_F64, GAAST_FLAG_DEBUG_NO_CHAIN, what, "product_ell");   // rows of n entries, +-1: the [term][row] form
        }
    }
    {   // list chains: (R X ~R).g(1) -- two lists, one step (k_product_ell_chain); n = 8 would also fit the LDS interpreter
        for (int n : {8, 9, 10, 12}) {
            uint64_t even = 0;
            for (int k = 0; k <= n; k += 2) even |= uint64_t(1) << k;
            gaast_expr_t r = gaast_expr_input(0, even, n), x = gaast_expr_input(1, 0x2, n), b = gaast_expr_input(2, even, n);
            gaast_expr_t e = gaast_expr_g(gaast_expr_product(gaast_expr_product(r, x, GAAST_PROD_GEOMETRIC), gaast_expr_rev(r), GAAST_PROD_GEOMETRIC), 1);
            char what[64];
            std::snprintf(what, sizeof what, "projected sandwich n=%d: list chain", n);
            lower(e, n, euclid, GAAST_F64, 0, what, "product_ell[");
            lower(e, n, euclid, GAAST_F64, 0, what, " <- product_ell[");
            std::snprintf(what, sizeof what, "projected sandwich n=%d: two lists", n);
            lower(e, n, euclid, GAAST_F64, GAAST_FLAG_DEBUG_NO_CHAIN, what, n == 8 ? "ast_" : "product_ell[");
            if (n == 9) {   // the mid row as the RIGHT operand of the second list, whose other operand is a third input
                gaast_expr_t e2 = gaast_expr_g(gaast_expr_product(b, gaast_expr_product(x, gaast_expr_rev(r), GAAST_PROD_GEOMETRIC), GAAST_PROD_GEOMETRIC), 1);
                lower(e2, n, euclid, GAAST_F32, 0, "b (x ~r) projected: mid row on the right", " <- product_ell[");
            }
        }
    }
    for (gaast_expr_t h : handles) gaast_expr_release(h);
    if (failures) {
        std::printf("%d failures\n", failures);
        return 1;
    }
    std::printf("ALL OK\n");
    return 0;
}
