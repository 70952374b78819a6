// Lowering of a flat SpecializedAst program into kernel launches.  See plan.hpp.
//
// The walk below IS the reference's interpreter (src/eval.rs) with every data-touching
// statement replaced by "emit a launch":
//   store_in_cache  eval.rs:21-33     add_to_res  eval.rs:35-115
// Three exact rewrites are applied unless GAAST_FLAG_NO_FUSION is set:
//   (1) a Product operand that is a bound input holding every wanted grade is read in place
//       (the kernel applies the `0.0 + x` of the reference's zero-init + add_grades_from copy,
//       eval.rs:27-31 / graded.rs:74) instead of being copied to a cache buffer;
//   (2) a Product that is the first writer of a fresh buffer starts its sums from the
//       zero-initialised accumulator in registers and the zero-fill launch is dropped;
//   (3) products whose list is a dense slice of the geometric product's table run on the
//       bitmask-tiled kernel (re-ordered sums: tolerance, not bit-exact; GAAST_FLAG_EXACT_ORDER
//       keeps them on the exact kernel).
#include "plan.hpp"
#include "spinor_basis.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>

#ifndef GAAST_JIT_NT
#define GAAST_JIT_NT 3   /* bit 0: nontemporal span stores, bit 1: nontemporal span loads in the specialised kernels (A/B switch; both: config 5 0.70 -> 0.78-0.82 of 8 TB/s) */
#endif
namespace gaast {
namespace {

struct Lowering {
    const gaast_program_desc& d;
    Plan& plan;
    BladeTable bt;
    std::vector<int> cache;               // node -> node_buffers index, -1 = not cached
    std::map<std::pair<int, int>, int> fresh;  // buffer -> index of its still-untouched ZERO step
    std::vector<char> removed;

    Lowering(const gaast_program_desc& desc, Plan& p)
        : d(desc), plan(p), bt(desc.vec_space_dim), cache(size_t(desc.n_nodes), -1) {}

    const gaast_node_desc& node(int i) const { return d.nodes[i]; }

    const Layout& layout(BufRef r) const {
        switch (r.kind) {
        case BufKind::NODE: return plan.node_buffers[size_t(r.idx)];
        case BufKind::INPUT: return plan.input_layouts[size_t(r.idx)];
        default: return plan.out_layout;
        }
    }
    static std::pair<int, int> key(BufRef r) { return {int(r.kind), r.idx}; }
    void touch(BufRef r) { fresh.erase(key(r)); }

    bool fail(int status, const std::string& msg) {
        if (plan.error == GAAST_OK) {
            plan.error = status;
            plan.error_msg = msg;
        }
        return false;
    }
    bool ok() const { return plan.error == GAAST_OK; }

    Step& emit(Step::Kind k, BufRef res, const std::string& name) {
        plan.steps.emplace_back();
        removed.push_back(0);
        Step& s = plan.steps.back();
        s.kind = k;
        s.res = res;
        s.name = name;
        return s;
    }

    void emit_zero(BufRef buf) {
        emit(Step::ZERO, buf, "zero_fill[init_null_mv]");
        fresh[key(buf)] = int(plan.steps.size()) - 1;
    }

    // eval.rs:21-33
    BufRef store_in_cache(int id) {
        if (cache[size_t(id)] >= 0) return BufRef{BufKind::NODE, cache[size_t(id)]};
        plan.node_buffers.push_back(make_layout(node(id).vec_space_dim, node(id).minimal_grade_mask));
        const int b = int(plan.node_buffers.size()) - 1;
        cache[size_t(id)] = b;
        BufRef buf{BufKind::NODE, b};
        emit_zero(buf);
        add_to_res(buf, id);
        return buf;
    }

    // every offset of the grades of `mask` inside `res`; MISSING_GRADE if res lacks one
    bool offsets_of(BufRef res, uint64_t mask, std::vector<uint32_t>& out) {
        const Layout& l = layout(res);
        for (int k = 0; k < 64; ++k) {
            if (!((mask >> k) & 1ULL)) continue;
            if (!((l.mask >> k) & 1ULL))
                return fail(GAAST_ERR_MISSING_GRADE, "grade " + std::to_string(k) + " absent from result buffer");
            for (int64_t i = 0; i < l.grade_len(k); ++i) out.push_back(uint32_t(l.offset(k) + i));
        }
        return true;
    }

    void emit_flip(BufRef res, uint64_t mask, const char* what) {
        std::vector<uint32_t> offs;
        if (!offsets_of(res, mask, offs)) return;
        if (offs.empty()) return;
        Step& s = emit(Step::FLIP, res, std::string("negate_grades[") + what + "]");
        s.u32_a = std::move(offs);
        touch(res);
    }

    // can the operand be read straight from the bound input?
    bool direct_input_ok(int id) const {
        const gaast_node_desc& nd = node(id);
        if (nd.opcode != GAAST_OP_INPUT) return false;
        const Layout& in = plan.input_layouts[size_t(nd.input_slot)];
        const uint64_t want = nd.minimal_grade_mask;
        if ((in.mask & want) != want) return false;  // a wanted grade would stay zero in the copy
        for (int k = 0; k < 64; ++k)
            if (((want >> k) & 1ULL) && in.grade_len(k) != int64_t(n_choose_k(uint64_t(nd.vec_space_dim), uint64_t(k))))
                return false;                         // zip() would truncate (graded.rs:73)
        return true;
    }

    // eval.rs:35-115
    void add_to_res(BufRef res, int id) {
        if (!ok()) return;
        const gaast_node_desc& nd = node(id);
        const uint64_t gs = nd.minimal_grade_mask;
        if (gs == 0) return;  // eval.rs:40-43
        switch (nd.opcode) {
        case GAAST_OP_INPUT: {  // eval.rs:45-50 -> graded.rs:67-78
            const Layout& in = plan.input_layouts[size_t(nd.input_slot)];
            const Layout& r = layout(res);
            std::vector<uint32_t> map;
            for (int k = 0; k < 64; ++k) {
                if (!((gs >> k) & 1ULL) || !((in.mask >> k) & 1ULL)) continue;
                if (!((r.mask >> k) & 1ULL)) {
                    fail(GAAST_ERR_MISSING_GRADE, "grade " + std::to_string(k) + " absent from result buffer");
                    return;
                }
                const int64_t len = std::min(r.grade_len(k), in.grade_len(k));  // zip
                for (int64_t i = 0; i < len; ++i)
                    map.push_back(uint32_t(r.offset(k) + i) | (uint32_t(in.offset(k) + i) << 16));
            }
            if (map.empty()) return;
            // the first writer of a fresh buffer that covers every component of it: the zero fill is folded into the copy
            // (res = 0.0 + in: init_null_mv then `*r = *r + i`, graded.rs:74) -- one launch and one pass over res less
            const auto fr = fresh.find(key(res));
            const bool covers = fr != fresh.end() && !(plan.flags & GAAST_FLAG_NO_FUSION) && int64_t(map.size()) == r.row_len;
            if (covers) removed[size_t(fr->second)] = 1;
            Step& s = emit(Step::AXPY, res, std::string(covers ? "copy_grades_from" : "add_grades_from") + "[input " + std::to_string(nd.input_slot) + "]");
            s.a = BufRef{BufKind::INPUT, nd.input_slot};
            s.u32_a = std::move(map);
            s.beta = covers ? 0 : 1;
            touch(res);
            return;
        }
        case GAAST_OP_ADD:  // eval.rs:51-54
            add_to_res(res, nd.child0);
            add_to_res(res, nd.child1);
            return;
        case GAAST_OP_NEG:  // eval.rs:55-60
            add_to_res(res, nd.child0);
            if (ok()) emit_flip(res, gs, "Negation");
            return;
        case GAAST_OP_REVERSE: {  // eval.rs:87-94
            add_to_res(res, nd.child0);
            if (!ok()) return;
            if ((gs & 1ULL) && (plan.flags & GAAST_FLAG_DEBUG_OVERFLOW)) {
                fail(GAAST_ERR_OVERFLOW, "attempt to subtract with overflow (Reverse over grade 0, debug build)");
                return;
            }
            uint64_t m = 0;
            for (int k = 0; k < 64; ++k)
                if (((gs >> k) & 1ULL) && (k % 4 == 2 || k % 4 == 3)) m |= 1ULL << k;  // (k(k-1)/2) odd
            emit_flip(res, m, "Reverse");
            return;
        }
        case GAAST_OP_GINVOL: {  // eval.rs:95-102
            add_to_res(res, nd.child0);
            if (!ok()) return;
            uint64_t m = 0;
            for (int k = 1; k < 64; k += 2)
                if ((gs >> k) & 1ULL) m |= 1ULL << k;
            emit_flip(res, m, "GradeInvolution");
            return;
        }
        case GAAST_OP_SINV:
        case GAAST_OP_SSQRT: {  // eval.rs:103-110
            add_to_res(res, nd.child0);
            if (!ok()) return;
            const Layout& r = layout(res);
            if (!(r.mask & 1ULL)) {
                fail(GAAST_ERR_MISSING_GRADE, "scalar op on a buffer without grade 0");
                return;
            }
            Step& s = emit(Step::SUNARY, res, nd.opcode == GAAST_OP_SINV ? "scalar_inversion" : "scalar_sqrt");
            s.sunary_op = nd.opcode == GAAST_OP_SINV ? 0 : 1;
            s.sunary_off = int(r.offset(0));
            touch(res);
            return;
        }
        case GAAST_OP_PROJ: add_to_res(res, nd.child0); return;  // eval.rs:111
        case GAAST_OP_EXP:
        case GAAST_OP_LOG:  // eval.rs:112-113
            if (plan.flags & GAAST_FLAG_EXP_LOG) {
                lower_exp_log(res, id);
                return;
            }
            fail(GAAST_ERR_UNIMPLEMENTED, "Exponential / Logarithm evaluation is todo!() in the reference");
            return;
        case GAAST_OP_PRODUCT: lower_product(res, id); return;
        default: throw std::runtime_error("unknown opcode");
        }
    }

    // EXTENSION (GAAST_FLAG_EXP_LOG; eval.rs:112-113 is todo!() upstream, "no reference behaviour"): the semantics the
    // reference's grade rules imply (grade_set.rs:181-197), stated in oracle/gaast_oracle.c: ext_exp_log.  The operand is
    // cached like a product operand (eval.rs:67-68); res += exp / log of it.
    void lower_exp_log(BufRef res, int id) {
        const gaast_node_desc& nd = node(id);
        const bool is_exp = nd.opcode == GAAST_OP_EXP;
        const gaast_node_desc& ch = node(nd.child0);
        BufRef arg = store_in_cache(nd.child0);
        if (!ok()) return;
        if (key(arg) == key(res)) {
            fail(GAAST_ERR_MISSING_GRADE, "exp / log operand aliases its own result buffer");
            return;
        }
        int k = -1, nk = 0;
        for (int g = 0; g < 64; ++g) nk += int((ch.minimal_grade_mask >> g) & 1ULL);
        for (int g = 0; g < 64; ++g)
            if (((ch.minimal_grade_mask >> g) & 1ULL) && (g != 0 || (is_exp && nk == 1))) k = g;
        if (k < 0) {
            fail(GAAST_ERR_INVALID_PROGRAM, "log can only be used on multivectors of the form <A>_0 + <A>_k");
            return;
        }
        const Layout &la = layout(arg), &lr = layout(res);
        if (!((la.mask >> k) & 1ULL)) {
            fail(GAAST_ERR_MISSING_GRADE, "grade absent from exp / log operand");
            return;
        }
        const int dim = ch.vec_space_dim;
        if (dim != d.vec_space_dim) {
            fail(GAAST_ERR_INVALID_PROGRAM, "exp / log operand lives in another vector space than the algebra");
            return;
        }
        const int64_t m = la.grade_len(k);
        Step st;   // filled before emit(): emit invalidates references into plan.steps
        st.explog_op = is_exp ? 0 : 1;
        st.explog_m = int(m);
        st.explog_arg_k = int(la.offset(k));
        st.explog_arg_0 = (!is_exp && (la.mask & 1ULL) && la.grade_len(0) > 0) ? int(la.offset(0)) : -1;
        const uint64_t mine = nd.minimal_grade_mask;
        if (is_exp && (mine & 1ULL)) {
            if (!(lr.mask & 1ULL)) {
                fail(GAAST_ERR_MISSING_GRADE, "grade 0 absent from result buffer");
                return;
            }
            st.explog_res_0 = int(lr.offset(0));
        }
        // (exp of a bare scalar, k = 0: both statements land in grade 0, cosh|a| + (sinh|a| / |a|) a = e^a, as in the oracle)
        if ((mine >> k) & 1ULL) {
            if (!((lr.mask >> k) & 1ULL)) {
                fail(GAAST_ERR_MISSING_GRADE, "grade " + std::to_string(k) + " absent from result buffer");
                return;
            }
            st.explog_res_k = int(lr.offset(k));
            st.explog_mres = int(std::min<int64_t>(lr.grade_len(k), m));
        }
        // blade squares, and the pairs of commuting blades (the non-scalar part of B B) grouped by product blade
        std::vector<uint32_t> blade(static_cast<size_t>(m));
        for (int64_t i = 0; i < m; ++i) blade[size_t(i)] = bt.blade_of[size_t(k)][size_t(i)];
        st.coeff.resize(size_t(m));
        for (int64_t i = 0; i < m; ++i)
            st.coeff[size_t(i)] = blades_gp_coeff(d.vec_space_dim, d.metric_diag, blade[size_t(i)], blade[size_t(i)]);
        bool structurally_scalar = true;   // every pair of distinct grade-k blades anticommutes
        for (int64_t i = 0; i < m && structurally_scalar; ++i)
            for (int64_t j = i + 1; j < m; ++j)
                if (((k - __builtin_popcount(blade[size_t(i)] & blade[size_t(j)])) & 1) == 0) {
                    structurally_scalar = false;
                    break;
                }
        st.u32_a.assign(1, 0u);
        if (!structurally_scalar) {
            if (m > 512) {
                if (plan.unsupported.empty())
                    plan.unsupported = "exp / log of a " + std::to_string(m) + "-component k-vector: the domain check (square is scalar) is built for up to 512 components";
                return;
            }
            std::map<uint64_t, std::vector<std::pair<uint32_t, double>>> rows;   // T -> (i | j << 16, 2 e_i e_j), in (i, j) order
            for (int64_t i = 0; i < m; ++i)
                for (int64_t j = i + 1; j < m; ++j) {
                    const double c1 = blades_gp_coeff(d.vec_space_dim, d.metric_diag, blade[size_t(i)], blade[size_t(j)]);
                    const double c2 = blades_gp_coeff(d.vec_space_dim, d.metric_diag, blade[size_t(j)], blade[size_t(i)]);
                    if (c1 == c2) rows[uint64_t(blade[size_t(i)] ^ blade[size_t(j)])].push_back({uint32_t(i) | (uint32_t(j) << 16), 2.0 * c1});
                }
            for (auto& kv : rows) {
                for (auto& e : kv.second) {
                    st.u32_c.push_back(e.first);
                    st.coeff_b.push_back(e.second);
                }
                st.u32_a.push_back(uint32_t(st.u32_c.size()));
            }
        }
        Step& s = emit(Step::EXPLOG, res, std::string(is_exp ? "exponential" : "logarithm") + "[grade " + std::to_string(k) + ", " +
                                              std::to_string(m) + " components, " + std::to_string(st.u32_c.size()) + " domain-check pairs]");
        const BufRef keep_res = s.res;
        const std::string keep_name = s.name;
        s = std::move(st);
        s.kind = Step::EXPLOG;
        s.res = keep_res;
        s.name = keep_name;
        s.a = arg;
        plan.has_explog = 1;
        touch(res);
    }

    // Which buffer a Product reads for operand `id`.  Exact rewrites (off with NO_FUSION):
    //  - a bound input holding every wanted grade is read in place (canon: 0.0 + x);
    //  - a chain of sign-only arms (Negation / Reverse / GradeInvolution / GradeProjection) over
    //    such an input is not materialised either: the grades it negates are returned in *flip
    //    and folded into the comp-mul coefficients -- ((-l) * r) * c == (l * r) * (-c) bit for bit.
    BufRef operand(int id, int* canon, uint64_t* flip) {
        *canon = 0;
        *flip = 0;
        if (!(plan.flags & GAAST_FLAG_NO_FUSION) && cache[size_t(id)] < 0) {
            uint64_t f = 0;
            int cur = id;
            bool ok_chain = true;
            while (node(cur).opcode != GAAST_OP_INPUT) {
                const gaast_node_desc& nd = node(cur);
                const uint64_t gs = nd.minimal_grade_mask;
                const bool sign_only = nd.opcode == GAAST_OP_NEG || nd.opcode == GAAST_OP_REVERSE ||
                                       nd.opcode == GAAST_OP_GINVOL || nd.opcode == GAAST_OP_PROJ;
                // the child must fill exactly the grades this node's buffer would have (Q2 otherwise)
                if (!sign_only || node(nd.child0).minimal_grade_mask != gs || gs == 0) {
                    ok_chain = false;
                    break;
                }
                if (nd.opcode == GAAST_OP_REVERSE && (gs & 1ULL) && (plan.flags & GAAST_FLAG_DEBUG_OVERFLOW)) {
                    ok_chain = false;  // let the materialising path report the panic
                    break;
                }
                for (int k = 0; k < 64; ++k) {
                    if (!((gs >> k) & 1ULL)) continue;
                    const bool neg = nd.opcode == GAAST_OP_NEG || (nd.opcode == GAAST_OP_REVERSE && (k % 4 == 2 || k % 4 == 3)) ||
                                     (nd.opcode == GAAST_OP_GINVOL && (k & 1));
                    if (neg) f ^= 1ULL << k;
                }
                cur = nd.child0;
            }
            if (ok_chain && direct_input_ok(cur)) {
                *canon = 1;
                *flip = f;
                return BufRef{BufKind::INPUT, node(cur).input_slot};
            }
        }
        return store_in_cache(id);  // eval.rs:67-68
    }

    // A host that cannot name the product (the Rust shim: the grades_to_produce closure is opaque,
    // base_types.rs:60-64) sends GAAST_PROD_EXPLICIT.  A list big enough to matter for the dense
    // kernels is compared entry by entry with the geometric product's list for the same grade
    // sets and metric; only an exact match (indices and coefficient bits) is treated as one.
    bool is_geometric_list(const gaast_node_desc& nd) const {
        if (nd.product_kind == GAAST_PROD_GEOMETRIC) return true;
        if (nd.product_kind != GAAST_PROD_EXPLICIT || !nd.comp_muls) return false;
        const int n = d.vec_space_dim;
        if (n < 6 || double(nd.n_comp_muls) * 32.0 < double(uint64_t(1) << (2 * n))) return false;   // (4^(n-1) / 8: parity-pure products)
        const uint64_t lmin = node(nd.child0).minimal_grade_mask, rmin = node(nd.child1).minimal_grade_mask;
        auto contribs = iter_contribs(nd.minimal_grade_mask, Selection{GAAST_PROD_GEOMETRIC, nullptr, nullptr}, lmin, rmin);
        if (comp_mul_count(n, contribs) != nd.n_comp_muls) return false;
        uint64_t e = 0;
        bool same = true;
        for_each_comp_mul(bt, d.metric_diag, contribs, [&](const gaast_comp_mul& m) {
            if (same && std::memcmp(&m, &nd.comp_muls[e], sizeof(gaast_comp_mul)) != 0) same = false;
            ++e;
        });
        return same;
    }

    // The algebra a dense kernel runs in: the program's (n, metric), or -- for a product of parity-pure operands -- the
    // even subalgebra Cl+(p, q) = Cl(n - 1) built on a pivot vector e_p (parity_reduced_frame below).
    struct DenseFrame {
        int n = 0;
        std::vector<double> metric;
        int pivot = -1;            // >= 0: parity-reduced; the basis vector of the program's algebra the reduction is built on
        int lpar = -1, rpar = -1;  // parity of the left / right operand (0 even, 1 odd) when reduced
    };
    static int parity_of(uint64_t mask) {   // 0: only even grades, 1: only odd grades, -1: mixed or empty
        const uint64_t EVEN = 0x5555555555555555ULL;
        if (!mask) return -1;
        if (!(mask & ~EVEN)) return 0;
        if (!(mask & EVEN)) return 1;
        return -1;
    }

    // Basis permutation that brings a diagonal metric into the shape the dense kernels want: position j of the
    // permuted basis holds original vector perm[j]; positions [0, L) (the "lo" bits of a blade) hold vectors that
    // square to +-1 (+1 first), never 0.  uniform: the four lo vectors must all square to the same sign (vector-FMA
    // kernel: compile-time sign pattern).  The identity is kept whenever it already qualifies.
    static bool dense_basis_permutation(const DenseFrame& f, int L, bool uniform, std::vector<int>& perm) {
        const int n = f.n;
        std::vector<int> plus, minus;   // by the SIGN of the square: a general entry g is rescaled to g / |g| (blade_scale)
        for (int i = 0; i < n; ++i) {
            if (f.metric[size_t(i)] > 0.0) plus.push_back(i);
            if (f.metric[size_t(i)] < 0.0) minus.push_back(i);
        }
        std::vector<int> lo;
        if (uniform) {
            if (int(plus.size()) >= L) lo.assign(plus.begin(), plus.begin() + L);
            else if (int(minus.size()) >= L) lo.assign(minus.begin(), minus.begin() + L);
            else return false;
        } else {
            if (int(plus.size() + minus.size()) < L) return false;
            for (int i : plus) if (int(lo.size()) < L) lo.push_back(i);
            for (int i : minus) if (int(lo.size()) < L) lo.push_back(i);
        }
        std::sort(lo.begin(), lo.end());
        std::vector<char> is_lo(size_t(n), 0);
        for (int i : lo) is_lo[size_t(i)] = 1;
        perm = lo;
        for (int i = 0; i < n; ++i)
            if (!is_lo[size_t(i)]) perm.push_back(i);
        return true;
    }

    // General diagonal metric on the dense kernels.  e_i = r_i f_i with r_i = sqrt|g_i| (1 for a null vector) gives a basis
    // whose metric is sign(g_i) in {+1, -1, 0}; a blade e_S = w_S f_S, w_S = prod_{i in S} r_i.  So
    //     C_T = (1 / w_T) * sum_{S ^ U = T} s'(S, U) (w_S A_S) (w_U B_U),
    // s' the sign-metric coefficient: w_S w_U / w_T = prod_{i in S & U} |g_i| is the reference's coefficient magnitude
    // (algebra.rs:78-81).  The operands are multiplied by w while they are staged, the result by 1 / w_T when it is stored:
    // three more roundings per term than the +-1 / 0 case (the factors themselves are rounded once, from long double).
    // Used only when every w_S and 1 / w_S stays within 2^+-40 (f32) / 2^+-300 (f64) -- no overflow or gradual underflow
    // introduced by the rescaling for operands of ordinary magnitude; otherwise the exact list kernels keep the product.
    static bool metric_is_unit(const DenseFrame& f) {
        for (double g : f.metric)
            if (g != 1.0 && g != -1.0 && g != 0.0) return false;
        return true;
    }
    bool dense_scales_ok(const DenseFrame& f) const {
        long double up = 1.0L, down = 1.0L;
        for (double g : f.metric) {
            if (!(g == g) || g == 1.0 / 0.0 || g == -1.0 / 0.0) return false;
            if (g == 0.0) continue;
            const long double r = sqrtl(fabsl((long double)g));
            if (r > 1.0L) up *= r;
            else down *= r;
        }
        if (f.pivot >= 0) {   // the reduction's own factors are powers of 1 / g_p up to (n - 1) / 2, and g_p itself
            const long double gp = fabsl((long double)d.metric_diag[f.pivot]);
            const long double worst = powl(gp > 1.0L ? gp : 1.0L / gp, (long double)((d.vec_space_dim + 1) / 2));
            up *= worst;
            down /= worst;
        }
        if (metric_is_unit(f) && (f.pivot < 0 || fabs(d.metric_diag[f.pivot]) == 1.0)) return true;
        const long double lim = plan.dtype == GAAST_F32 ? 0x1p40L : 0x1p300L;
        return up <= lim && down >= 1.0L / lim;
    }
    static long double blade_scale(const DenseFrame& f, uint32_t S) {   // w_S in the frame's algebra
        long double w = 1.0L;
        for (int i = 0; i < f.n; ++i)
            if (((S >> i) & 1u) && f.metric[size_t(i)] != 0.0) w *= sqrtl(fabsl((long double)f.metric[size_t(i)]));
        return w;
    }

    // Parity-pure operands (the reference only ever multiplies the entries it needs, specialize.rs:162-183; even x even --
    // rotor composition, the second product of every sandwich -- needs a quarter of the 4^n table).  The even subalgebra
    // Cl+ of Cl(n) is a Clifford algebra of dimension n - 1 on the generators f_i = e_i e_p (i != p, e_p any non-null basis
    // vector): f_i f_j = -f_j f_i, f_i^2 = -g_i g_p.  An even blade e_E is a multiple of the f-blade on E \ {p},
    //     e_E = c(E) f_{E \ p},   c(E) = sigma(E) (-g_p)^(-floor(|E \ p| / 2)),   sigma(E) = (-1)^#{i in E : i > p} if p in E, else 1
    // and an odd multivector is (even) e_p:  A = A~ e_p,  A~_{S ^ p} = A_S tau(S) (p in S ? 1 : 1 / g_p),  tau(S) = (-1)^#{i in S : i > p}.
    // With X^ = e_p X e_p^-1 (even X: the blades containing p change sign):
    //     even x even: A B                      odd x even: (A~ B^) e_p
    //     even x odd : (A B~) e_p               odd x odd : g_p A~ B~^
    // Every case is ONE product of two elements of Cl(n - 1) -- 4^(n-1) multiply-adds -- with per-component factors on the
    // way in and on the way out: exactly what the operand maps (position, negate bit, scale) and the result map carry.
    bool parity_reduced_frame(uint64_t lmask, uint64_t rmask, DenseFrame& f) const {
        const int n = d.vec_space_dim;
        const int lp = parity_of(lmask), rp = parity_of(rmask);
        if (lp < 0 || rp < 0 || n - 1 < 6) return false;
        int p = -1;
        for (int i = n - 1; i >= 0 && p < 0; --i)
            if (d.metric_diag[i] != 0.0) p = i;
        if (p < 0) return false;
        f.n = n - 1;
        f.metric.clear();
        for (int i = 0; i < n; ++i)
            if (i != p) f.metric.push_back(-d.metric_diag[i] * d.metric_diag[p] == 0.0 ? 0.0 : -d.metric_diag[i] * d.metric_diag[p]);
        f.pivot = p;
        f.lpar = lp;
        f.rpar = rp;
        return true;
    }

    // which dense kernel (0 = none, 1 = k_gp_dense, 3 = k_gp_mfma32 / k_gp_mfma32p, 4 = k_gp_mfma16x4<T>, 5 = k_gp_mfma7<T>, 6 = k_gp_mfma6<T>), in which algebra
    // (frame) and in which basis of it (perm)
    int dense_kind_for(const DenseFrame& f, uint64_t n_comp_muls, std::vector<int>& perm) const {
        const int n = f.n;
        if (n < 6 || n > 14) return 0;  // small algebras: the exact kernel is HBM-bound anyway
        // A general diagonal metric (algebra.rs:148-165 multiplies by ANY base_vec_dot, :79-81) runs in the rescaled basis
        // f_i = e_i / sqrt|g_i|; it needs finite, well-scaled factors, else the list kernels keep the product
        if (!dense_scales_ok(f)) return 0;
        if (double(n_comp_muls) * 8.0 < double(uint64_t(1) << (2 * n))) return 0;  // the tiled kernels always do 4^n multiply-adds
        const bool mfma_ok = plan.dtype == GAAST_F32 && !(plan.flags & GAAST_FLAG_NO_MFMA);
        // matrix-core variants: f32, n >= 10 (32 result columns per wave, five lo vectors) / n = 8, 9 (lo = 4 bits)
        if (mfma_ok && n >= 10 && dense_basis_permutation(f, 5, false, perm)) return 3;
        if (n == 14) return 0;          // both operands of an item (128 KiB in f32) fit the LDS of the matrix-core kernel only
        if (mfma_ok && (n == 8 || n == 9) && dense_basis_permutation(f, 4, false, perm)) return 4;   // k_gp_mfma16x4<float>
        // f64 (the reference's value type), n = 8 ... 12: v_mfma_f64_16x16x4_f64, one item per workgroup
        if (plan.dtype == GAAST_F64 && !(plan.flags & GAAST_FLAG_NO_MFMA) && n >= 8 && n <= 12 &&
            dense_basis_permutation(f, 4, false, perm))
            return 4;
        // n = 7, both value types: one wave per item on the 16x16x4 instructions (lo = 3 bits: three non-null vectors)
        if (n == 7 && !(plan.flags & GAAST_FLAG_NO_MFMA) && dense_basis_permutation(f, 3, false, perm)) return 5;
        // n = 6, both value types: four 16x16x4 instructions per item, ANY +-1 / 0 metric in the basis as it stands (signs and
        // vanishing terms are slots of the operand images and bits of the accumulators: no lo vectors, no permutation)
        if (n == 6 && !(plan.flags & GAAST_FLAG_NO_MFMA) && dense_basis_permutation(f, 0, false, perm)) return 6;
        if (dense_basis_permutation(f, 4, true, perm)) return 1;
        return 0;
    }
    int dense_choice(const gaast_node_desc& nd, BufRef res, BufRef l, BufRef r, std::vector<int>& perm, DenseFrame& frame) const {
        if (plan.flags & (GAAST_FLAG_EXACT_ORDER | GAAST_FLAG_NO_FUSION)) return 0;
        const int n = d.vec_space_dim;
        if (n < 6 || n > 15) return 0;
        if (layout(res).dim != n || layout(l).dim != n || layout(r).dim != n) return 0;
        const uint64_t lmask = node(nd.child0).minimal_grade_mask & layout(l).mask, rmask = node(nd.child1).minimal_grade_mask & layout(r).mask;
        int kind = 0;
        bool geometric_known = false, geometric = false;
        auto is_gp = [&]() {
            if (!geometric_known) {
                geometric = is_geometric_list(nd);
                geometric_known = true;
            }
            return geometric;
        };
        DenseFrame reduced;
        if (parity_reduced_frame(lmask, rmask, reduced) && (kind = dense_kind_for(reduced, nd.n_comp_muls, perm)) && is_gp()) {
            frame = reduced;
            return kind;
        }
        DenseFrame full;
        full.n = n;
        full.metric.assign(d.metric_diag, d.metric_diag + n);
        if (n <= 14 && (kind = dense_kind_for(full, nd.n_comp_muls, perm)) && is_gp()) {
            frame = full;
            return kind;
        }
        return 0;
    }

    // opt-in matrix-representation kernels: f32, n = 7..12 (odd n as the subalgebra of n + 1), every
    // vector squaring to +-1
    bool spinor_eligible(const gaast_node_desc& nd, BufRef res, BufRef l, BufRef r) const {
        if (!(plan.flags & GAAST_FLAG_SPINOR_GEMM)) return false;
        if (plan.flags & (GAAST_FLAG_EXACT_ORDER | GAAST_FLAG_NO_FUSION)) return false;
        const int n = d.vec_space_dim;
        if (n < 7 || n > 12) return false;
        if (layout(res).dim != n || layout(l).dim != n || layout(r).dim != n) return false;
        for (int i = 0; i < n; ++i)
            if (d.metric_diag[i] != 1.0 && d.metric_diag[i] != -1.0) return false;
        if (!is_geometric_list(nd)) return false;
        return double(nd.n_comp_muls) * 8.0 >= double(uint64_t(1) << (2 * n));
    }

    // blade -> Pauli string i^k X^x Z^z under the Jordan-Wigner generators (kernels_spinor.hip.hpp);
    // vectors beyond the algebra's dimension (odd n padded to n + 1) square to +1
    void pauli_string(uint32_t blade, uint32_t* x, uint32_t* z, uint32_t* k) const {
        uint32_t px = 0, pz = 0, pk = 0;
        for (int v = 0; v < 32 && (blade >> v); ++v) {
            if (!((blade >> v) & 1u)) continue;
            const int j = v >> 1;
            const uint32_t gx = 1u << j;
            const uint32_t gz = (v & 1) ? (1u << (j + 1)) - 1u : (1u << j) - 1u;
            const bool negative = v < d.vec_space_dim && d.metric_diag[v] < 0.0;
            const uint32_t gk = uint32_t(v & 1) + (negative ? 1u : 0u);
            pk = (pk + gk + 2u * uint32_t(__builtin_popcount(pz & gx))) & 3u;
            px ^= gx;
            pz ^= gz;
        }
        *x = px;
        *z = pz;
        *k = pk;
    }

    void lower_product(BufRef res, int id) {  // eval.rs:61-86
        const gaast_node_desc& nd = node(id);
        int canon_l = 0, canon_r = 0;
        uint64_t flip_l = 0, flip_r = 0;
        BufRef l = operand(nd.child0, &canon_l, &flip_l);
        if (!ok()) return;
        BufRef r = operand(nd.child1, &canon_r, &flip_r);
        if (!ok()) return;
        if (key(l) == key(res) || key(r) == key(res)) {
            fail(GAAST_ERR_MISSING_GRADE, "product operand aliases its own result buffer");
            return;
        }
        const Layout &lr = layout(res), &ll = layout(l), &lrr = layout(r);
        const uint64_t lmin = node(nd.child0).minimal_grade_mask, rmin = node(nd.child1).minimal_grade_mask;
        const uint64_t omin = nd.minimal_grade_mask;
        if (nd.comp_muls == nullptr && nd.product_kind < 0)
            throw std::runtime_error("PRODUCT node has neither a comp-mul list nor a product kind");

        // may the zero-fill of a fresh result buffer be folded into this product?
        auto fr = fresh.find(key(res));
        const bool is_fresh = fr != fresh.end() && !(plan.flags & GAAST_FLAG_NO_FUSION);

        if (spinor_eligible(nd, res, l, r)) {
            if ((omin & lr.mask) != omin) {
                fail(GAAST_ERR_MISSING_GRADE, "product result grade absent from result buffer");
                return;
            }
            const bool beta0 = is_fresh && (lr.mask & ~omin) == 0;
            if (beta0) removed[size_t(fr->second)] = 1;
            const int n = d.vec_space_dim;
            const int m = (n + 1) / 2;                 // 2^m x 2^m complex matrices
            const uint32_t D = 1u << m, LD = D + 1u;
            Step& s = emit(Step::PRODUCT_DENSE, res, "product_spinor_gemm[gp n=" + std::to_string(n) + "]");
            s.a = l;
            s.b = r;
            s.canon_a = canon_l;
            s.canon_b = canon_r;
            s.beta = beta0 ? 0 : 1;
            s.n_entries = nd.n_comp_muls;
            s.use_spinor = m;
            {
                // D*D 16-bit table entries indexed by row offset, two per word (format: SpinorArgs);
                // one real plane per operand: indices in the basis of spinor_basis.hpp
                uint32_t alpha = 0, lam = 0;
                for (uint32_t blade = 0; blade < (1u << (2 * m)); ++blade) {
                    uint32_t px, pz, pk;
                    pauli_string(blade, &px, &pz, &pk);
                    if (pz == 0 && __builtin_popcount(px) == 1 && (pk & 1u)) alpha |= px;
                    if (px == 0 && __builtin_popcount(pz) == 1 && (pk & 1u)) lam |= pz;
                }
                const SpinorBasis sb = choose_spinor_basis(m, alpha, lam);
                s.spinor_lam_bit = sb.lam_bit;
                s.spinor_has_alpha = sb.has_alpha ? 1 : 0;
                auto build1 = [&](const Layout& lay, uint64_t want, uint64_t flip, int role, std::vector<uint32_t>& packed, int* full) {
                    // operands: bit 0 = negate, bits [14:2] = x'*LD + z'; result: bit 0 = negate, bit 1 = nothing
                    // to store, bits [15:2] = x'*LD + z'
                    const uint16_t nothing = role == 2 ? uint16_t(2u) : uint16_t(D << 2);
                    std::vector<uint16_t> map(size_t(D) * D, nothing);
                    size_t count = 0;
                    for (int k = 0; k <= n; ++k) {
                        if (!((want >> k) & 1ULL)) continue;
                        for (uint32_t i = 0; i < bt.grade_dim[size_t(k)]; ++i) {
                            uint32_t px, pz, pk;
                            pauli_string(bt.blade_of[size_t(k)][i], &px, &pz, &pk);
                            const uint32_t x2 = sb.map_x(px), z2 = sb.map_z(pz);
                            const uint32_t f = (sb.has_alpha ? (x2 >> (m - 1)) & 1u : 0u) ^
                                               (sb.lam_bit >= 0 ? (z2 >> sb.lam_bit) & 1u : 0u);
                            if (f != (pk & 1u)) throw std::runtime_error("spinor basis: phase parity mismatch");
                            uint32_t neg = uint32_t((flip >> k) & 1ULL) ^ (pk >> 1);
                            if (role != 0) neg ^= uint32_t(__builtin_popcount(x2 & z2) & 1);
                            map[size_t(lay.offset(k) + i)] = uint16_t((x2 * LD + z2) << 2 | neg);
                            ++count;
                        }
                    }
                    *full = count == size_t(D) * D;
                    packed.resize(map.size() / 2);
                    std::memcpy(packed.data(), map.data(), map.size() * sizeof(uint16_t));
                };
                build1(ll, lmin & ll.mask, flip_l, 0, s.u32_a, &s.left_full);
                build1(lrr, rmin & lrr.mask, flip_r, 1, s.u32_b, &s.right_full);
                build1(lr, omin, 0, 2, s.u32_c, &s.out_full);
                s.name = "product_spinor_gemm[gp n=" + std::to_string(n) + " lam=" + std::to_string(sb.lam_bit) + "]";
                touch(res);
                return;
            }
        }
        std::vector<int> perm;
        DenseFrame frame;
        if (const int dense_kind = dense_choice(nd, res, l, r, perm, frame)) {
            if ((omin & lr.mask) != omin) {
                fail(GAAST_ERR_MISSING_GRADE, "product result grade absent from result buffer");
                return;
            }
            const int n = d.vec_space_dim;   // the program's algebra: graded rows, blade <-> (grade, index)
            const int n2 = frame.n;          // the kernel's algebra: n, or n - 1 for parity-pure operands
            const bool reduced = frame.pivot >= 0;
            // grades this step produces: a parity-pure product fills only the grades of its own parity
            uint64_t prod_mask = omin;
            if (reduced) prod_mask &= ((frame.lpar ^ frame.rpar) ? 0xAAAAAAAAAAAAAAAAULL : 0x5555555555555555ULL);
            const bool beta0 = is_fresh && (lr.mask & ~prod_mask) == 0;
            if (beta0) {
                removed[size_t(fr->second)] = 1;
            }
            Step& s = emit(Step::PRODUCT_DENSE, res, "product_dense[gp n=" + std::to_string(n) + "]");
            s.a = l;
            s.b = r;
            s.canon_a = canon_l;
            s.canon_b = canon_r;
            s.beta = beta0 ? 0 : 1;
            s.n_entries = nd.n_comp_muls;
            s.dense_n = n2;
            s.use_mfma = dense_kind == 3;
            s.use_mfma16 = dense_kind == 4;
            s.use_mfma16d = dense_kind == 4;
            s.use_mfma7 = dense_kind == 5;
            s.use_mfma6 = dense_kind == 6;
            s.mfma16_quads = dense_kind == 4 && plan.dtype == GAAST_F32;   // k_gp_mfma16x4<float>: B words in 16-byte quads
            s.mfma32_pairs = dense_kind == 3 && n2 <= 13;   // k_gp_mfma32p: +A, -A, +B, -B images (n = 14 does not fit)
            // blade R of the frame's basis <-> blade R' of its permuted basis, f_R = sign(R) f'_R' (the parity of the
            // inversions of the new positions of R's vectors taken in ascending original order)
            std::vector<int> inv(size_t(n2), 0);
            bool identity = true;
            for (int j = 0; j < n2; ++j) {
                inv[size_t(perm[size_t(j)])] = j;
                identity = identity && perm[size_t(j)] == j;
            }
            std::vector<uint32_t> new_blade(size_t(1) << n2), blade_sign(size_t(1) << n2);
            for (uint32_t S = 0; S < (1u << n2); ++S) {
                uint32_t S2 = 0, par = 0;
                for (int p = 0; p < n2; ++p) {
                    if (!((S >> p) & 1u)) continue;
                    const int q = inv[size_t(p)];
                    par ^= uint32_t(__builtin_popcount(S2 >> (q + 1))) & 1u;
                    S2 |= 1u << q;
                }
                new_blade[S] = S2;
                blade_sign[S] = par;
            }
            // ---- stage 1: blade of the program's algebra -> blade of the frame's algebra, with a factor ----
            const int pv = frame.pivot;
            const long double gp = reduced ? (long double)d.metric_diag[pv] : 1.0L;
            auto has_p = [&](uint32_t S) { return ((S >> pv) & 1u) != 0; };
            auto tau = [&](uint32_t S) -> long double { return (__builtin_popcount(S >> (pv + 1)) & 1) ? -1.0L : 1.0L; };
            auto compress = [&](uint32_t E) -> uint32_t {   // drop bit p
                E &= ~(1u << pv);
                return (E & ((1u << pv) - 1u)) | ((E >> (pv + 1)) << pv);
            };
            auto c_of = [&](uint32_t E) -> long double {    // e_E = c(E) f_{E \ p} for an even blade E
                const int k = __builtin_popcount(E & ~(1u << pv));
                long double v = powl(-1.0L / gp, (long double)(k / 2));
                return has_p(E) ? v * tau(E) : v;
            };
            auto t_of = [&](uint32_t S) -> long double { return tau(S) * (has_p(S) ? 1.0L : 1.0L / gp); };   // odd S: A~_{S ^ p} = A_S t(S)
            // operand component on blade S -> (frame blade R, factor): image value = A_S * factor
            auto operand_to_frame = [&](uint32_t S, bool right, long double* factor) -> uint32_t {
                if (!reduced) {
                    *factor = 1.0L;
                    return S;
                }
                const int par = right ? frame.rpar : frame.lpar;
                const uint32_t E = par ? S ^ (1u << pv) : S;
                long double f = (par ? t_of(S) : 1.0L) * c_of(E);
                if (right && frame.lpar == 1 && has_p(E)) f = -f;   // the conjugation e_p X e_p^-1 of the right operand
                *factor = f;
                return compress(E);
            };
            // result component on blade T <- (frame blade R, factor): C_T = C'_R * factor
            auto result_from_frame = [&](uint32_t T, long double* factor) -> uint32_t {
                if (!reduced) {
                    *factor = 1.0L;
                    return T;
                }
                if ((frame.lpar ^ frame.rpar) == 0) {
                    *factor = ((frame.lpar == 1) ? gp : 1.0L) / c_of(T);
                    return compress(T);
                }
                const uint32_t E = T ^ (1u << pv);   // C = P e_p: C_{E ^ p} = P_E tau(E) (p in E ? g_p : 1)
                *factor = tau(E) * (has_p(E) ? gp : 1.0L) / c_of(E);
                return compress(E);
            };
            // position of blade m in the LDS image the kernel reads (mirrors kernels.hip.hpp)
            auto vec_pos = [](uint32_t m) {  // dense_lds_pos
                const uint32_t x = m >> 4, lo = m & 15;
                return (x << 4) | ((((lo >> 2) ^ (x >> 2)) & 3) << 2) | (lo & 3);
            };
            // k_gp_mfma32p's B image: the lane's 16 words (k of one parity) even-|k >> 1| first, quads rotated as in mfma_b_pos
            auto mfma32p_b_pos = [](uint32_t m) {
                static const int word_of_s[16] = {0, 8, 9, 1, 10, 2, 3, 11, 12, 4, 5, 13, 6, 14, 15, 7};
                const uint32_t x = m >> 5, k = m & 31, w = uint32_t(word_of_s[k >> 1]);
                const uint32_t lq = ((k & 1) << 2) | (w >> 2);
                return (x << 5) | (((lq ^ (x >> 1)) & 7) << 2) | (w & 3);
            };
            // k_gp_mfma16d's B image (f64): word k of block x at k ^ (((x >> 1) & 7) << 1)
            auto mfma16d_b_pos = [](uint32_t m) {
                const uint32_t x = m >> 4, k = m & 15u;
                return (x << 4) | (k ^ (((x >> 1) & 7u) << 1));
            };
            // k_gp_mfma16x4<float>'s B image: word k of block x in quad kq(k) ^ (((x >> 2) & 1) << 1), slot s(k), with
            // (kq, s) from the kernel's k table (Mfma16x4<float>::k_of)
            auto mfma16q_b_pos = [](uint32_t m) {
                static const int kq_of[16] = {0, 2, 2, 0, 2, 0, 0, 2, 3, 1, 1, 3, 1, 3, 3, 1};
                static const int s_of[16] = {0, 0, 1, 1, 2, 2, 3, 3, 0, 0, 1, 1, 2, 2, 3, 3};
                const uint32_t x = m >> 4, k = m & 15u;
                return (x << 4) | (uint32_t(kq_of[k] ^ int(((x >> 2) & 1u) << 1)) << 2) | uint32_t(s_of[k]);
            };
            // k_gp_mfma7: A image u * 72 + (a_hi3, a_lo); B image [kq][v][b_hi3][s] with (kq, s) from the kernel's k table (mfma7_k)
            auto mfma7_a_pos = [](uint32_t m) { return (m & 63u) + (m >> 6) * 72u; };
            auto mfma7_b_pos = [](uint32_t m) {
                static const int kq_of[8] = {0, 2, 2, 0, 3, 1, 1, 3};
                static const int s_of[8] = {0, 0, 1, 1, 0, 0, 1, 1};
                const uint32_t v = m >> 6, bh = (m >> 3) & 7u, k = m & 7u;
                return ((uint32_t(kq_of[k]) * 16u + v * 8u + bh) << 1) | uint32_t(s_of[k]);
            };
            auto mfma_b_pos = [](uint32_t m) {
                const uint32_t x = m >> 5, k = m & 31;
                const uint32_t lq = ((k & 1) << 2) | (k >> 3);
                return (x << 5) | (((lq ^ (x >> 1)) & 7) << 2) | ((k >> 1) & 3);
            };
            // entry: row offset | image position << 16 | (negate while staging) << 31; scale[entry] = |factor| (when any != 1)
            auto build_map = [&](const Layout& lay, uint64_t want, uint64_t flip, bool right, std::vector<uint32_t>& map,
                                 std::vector<double>& scale, int* full, int* contig) {
                bool seq = true;
                for (int k = 0; k <= n; ++k) {
                    if (!((want >> k) & 1ULL)) continue;
                    for (uint32_t i = 0; i < bt.grade_dim[size_t(k)]; ++i) {
                        const uint32_t orig = bt.blade_of[size_t(k)][i];
                        long double f1;
                        const uint32_t R = operand_to_frame(orig, right, &f1);
                        const uint32_t blade = new_blade[R];
                        const long double factor = f1 * blade_scale(frame, R);
                        uint32_t neg = uint32_t((flip >> k) & 1ULL) ^ blade_sign[R] ^ (factor < 0.0L ? 1u : 0u);
                        // image-pair kernels: the b_hi part of (-1)^(|a_hi| |b_lo|), |a_hi| = |b_hi| + |c_hi| (mod 2), lives in the B image
                        if (s.use_mfma16 && right) neg ^= uint32_t(__builtin_popcount(blade >> 4) & __builtin_popcount(blade & 15u) & 1);
                        if (s.mfma32_pairs && right) neg ^= uint32_t(__builtin_popcount(blade >> 5) & __builtin_popcount(blade & 31u) & 1);
                        if (s.use_mfma7 && right) neg ^= uint32_t(__builtin_popcount((blade >> 3) & 7u) & __builtin_popcount(blade & 7u) & 1);
                        const uint32_t sgn = neg ? 0x80000000u : 0u;
                        const uint32_t pos = s.use_mfma6 ? blade   // k_gp_mfma6 derives its image slots from the blade itself
                                             : s.use_mfma7 ? (right ? mfma7_b_pos(blade) : mfma7_a_pos(blade))
                                             : s.mfma32_pairs ? (right ? mfma32p_b_pos(blade) : blade)
                                             : s.use_mfma ? (right ? mfma_b_pos(blade) : blade)
                                             : s.mfma16_quads ? (right ? mfma16q_b_pos(blade) : blade)
                                             : s.use_mfma16d ? (right ? mfma16d_b_pos(blade) : blade) : vec_pos(blade);
                        const uint32_t off = uint32_t(lay.offset(k) + i);
                        seq = seq && off == map.size();
                        map.push_back(off | (pos << 16) | sgn);
                        scale.push_back(double(fabsl(factor)));
                    }
                }
                *full = map.size() == (size_t(1) << n2);
                *contig = seq && map.size() % 4 == 0 && !map.empty();
                if (s.use_mfma6 && *full) {
                    // k_gp_mfma6: lane q moves the component of entry q into its four image slots.  Entries are dealt to lanes so
                    // that the lanes sharing an LDS cycle of a store hit different banks (row order: 73 % of the LDS cycles were
                    // conflicts, profiles/r04_gp6f32_pmc_first_version.csv).  blade = (top2 | hi2 | lo2) = (u, ah, al) / (v, bh, bl):
                    //   f32 (32 lanes per cycle): A: bank = 16 (al ^ kq)_0 + 4 u + ah    -> group = al_1;       B (slot order
                    //        rotated by bl in the kernel): bank = 16 (bh ^ s)_0 + 4 v + s -> group = bh_1
                    //   f64 (16 lanes per cycle, 8-byte units mod 16): A: (u_0, al ^ kq, ah_0) -> group = (u_1, ah_1);  B: (v, bh_0, bl_0)
                    //        -> group = (bh_1, bl_1)
                    const bool f32 = plan.dtype == GAAST_F32;
                    auto lane_of = [&](uint32_t blade) -> uint32_t {
                        const uint32_t t = blade >> 4, h = (blade >> 2) & 3u, l = blade & 3u;
                        if (!right) return f32 ? ((l >> 1) << 5) | ((l & 1u) << 4) | (t << 2) | h
                                               : ((((t >> 1) << 1) | (h >> 1)) << 4) | ((t & 1u) << 3) | (l << 1) | (h & 1u);
                        return f32 ? ((h >> 1) << 5) | ((h & 1u) << 4) | (t << 2) | l
                                   : ((((h >> 1) << 1) | (l >> 1)) << 4) | (t << 2) | ((h & 1u) << 1) | (l & 1u);
                    };
                    std::vector<uint32_t> m2(map.size());
                    std::vector<double> s2(scale.size());
                    for (size_t e = 0; e < map.size(); ++e) {
                        const uint32_t q = lane_of((map[e] >> 16) & 63u);
                        m2[q] = map[e];
                        s2[q] = scale[e];
                    }
                    map.swap(m2);
                    scale.swap(s2);
                    *contig = 0;
                }
                if (s.use_mfma7 && *full) {
                    // k_gp_mfma7 moves ONE component per lane and load (entry q = load * 64 + lane), so the entries can be dealt
                    // to lanes by LDS bank: the lanes that share an LDS cycle of a store (f32: 32 lanes, bank = position mod 32;
                    // f64: 16 lanes of 8 bytes, position mod 16) get components of different banks -- every residue holds
                    // 128 / G positions of an image, one per run of G entries.  (Row order put 36 conflict cycles into the
                    // ~108 LDS cycles an item cost.)  The row offsets are no longer 0, 1, 2, ...: no 16-byte vector path.
                    const uint32_t G = plan.dtype == GAAST_F32 ? 32u : 16u;
                    std::vector<std::vector<size_t>> bucket(G);
                    for (size_t e = 0; e < map.size(); ++e) bucket[((map[e] >> 16) & 0x7fffu) % G].push_back(e);
                    bool even = true;
                    for (const auto& b : bucket) even = even && b.size() == map.size() / G;
                    if (even) {
                        std::vector<uint32_t> m2;
                        std::vector<double> s2;
                        for (size_t r = 0; r < map.size() / G; ++r)
                            for (uint32_t k = 0; k < G; ++k) {
                                m2.push_back(map[bucket[k][r]]);
                                s2.push_back(scale[bucket[k][r]]);
                            }
                        map.swap(m2);
                        scale.swap(s2);
                        *contig = 0;
                    }
                }
            };
            uint64_t lwant = lmin & ll.mask, rwant = rmin & lrr.mask;
            build_map(ll, lwant, flip_l, false, s.u32_a, s.coeff, &s.left_full, &s.left_contig);
            build_map(lrr, rwant, flip_r, true, s.u32_b, s.coeff_b, &s.right_full, &s.right_contig);
            // out_map: indexed by the blade of the frame's permuted basis; offset | sign << 30, or -1; coeff_c: |factor|
            s.i32_a.assign(size_t(1) << n2, -1);
            s.coeff_c.assign(size_t(1) << n2, 1.0);
            for (uint32_t m = 0; m < (1u << n); ++m) {
                const int g = __builtin_popcount(m);
                if (!((prod_mask >> g) & 1ULL)) continue;
                long double f1;
                const uint32_t R = result_from_frame(m, &f1);
                const long double factor = f1 / blade_scale(frame, R);
                const uint32_t sgn = blade_sign[R] ^ (factor < 0.0L ? 1u : 0u);
                s.i32_a[new_blade[R]] = int32_t(uint32_t(lr.offset(g) + bt.index_of[m]) | (sgn << 30));
                s.coeff_c[new_blade[R]] = double(fabsl(factor));
            }
            // scale tables only when some factor is not 1 (a general metric): +-1 / 0 metrics keep the register-prefetch paths
            s.scaled = 0;
            for (const std::vector<double>* v : {&s.coeff, &s.coeff_b, &s.coeff_c})
                for (double x : *v) s.scaled |= int(x != 1.0);
            if (!s.scaled) {
                s.coeff.clear();
                s.coeff_b.clear();
                s.coeff_c.clear();
            }
            // every blade produced into a row that holds nothing else: whole rows can be written in 16-byte pieces
            s.out_full = lr.row_len == (int64_t(1) << n2);
            for (uint32_t m = 0; m < (1u << n2); ++m) s.out_full = s.out_full && s.i32_a[m] >= 0;
            for (uint32_t w : s.u32_a) s.left_signs |= int(w >> 31);
            for (int32_t w : s.i32_a) s.out_signs |= int(w >= 0 && (uint32_t(w) & 0x40000000u));
            const int lo_bits = s.use_mfma ? 5 : s.use_mfma7 ? 3 : s.use_mfma6 ? 0 : 4;
            for (int j = 0; j < n2; ++j) {
                const double g = frame.metric[size_t(perm[size_t(j)])];   // only its sign matters here: the magnitude is in the scales
                if (j < lo_bits) {
                    if (g < 0.0) s.neg_lo |= 1u << j;
                } else {
                    if (g < 0.0) s.neg_hi |= 1u << (j - lo_bits);
                    if (g == 0.0) s.zero_hi |= 1u << (j - lo_bits);
                }
            }
            s.neg_lo_all = dense_kind == 1 && s.neg_lo == 15u;
            s.degenerate = s.zero_hi != 0;
            static const char* const par_name[2] = {"even", "odd"};
            s.name = std::string(dense_kind == 1 ? "product_dense" : "product_dense_mfma") + "[gp n=" + std::to_string(n) +
                     (reduced ? std::string(" ") + par_name[frame.lpar] + " x " + par_name[frame.rpar] + " in Cl(" + std::to_string(n2) + ")" : std::string()) +
                     (identity ? "" : " permuted basis") + (s.scaled ? " rescaled basis" : "") + "]";
            touch(res);
            return;
        }

        // ---- exact path: CSR by output component, entries in the reference's order ----
        std::vector<gaast_comp_mul> generated;
        const gaast_comp_mul* muls = nd.comp_muls;
        uint64_t n_muls = nd.n_comp_muls;
        {
            // capability limits, checked BEFORE any table is generated: the list kernels stage both operand rows of
            // an item in LDS, and the list itself has to fit the table budget
            const size_t elem = plan.dtype == GAAST_F32 ? 4 : 8;
            const size_t per_item = size_t(ll.row_len + lrr.row_len) * elem;
            const bool fits_fused_slab = ll.row_len + lrr.row_len + lr.row_len < 4096;
            if (per_item > kLdsBytes && !fits_fused_slab && plan.unsupported.empty())
                plan.unsupported = "product operands of " + std::to_string(per_item) + " bytes per item exceed the " +
                                   std::to_string(kLdsBytes) + "-byte LDS the list kernels stage them in";
            if (n_muls > kMaxListEntries && plan.unsupported.empty())
                plan.unsupported = "a comp-mul list of " + std::to_string(n_muls) + " entries exceeds this back end's table budget";
            if (!plan.unsupported.empty()) return;
        }
        if (!muls) {
            Selection sel{nd.product_kind, nullptr, nullptr};
            auto contribs = iter_contribs(omin, sel, lmin, rmin);
            if (comp_mul_count(d.vec_space_dim, contribs) > kMaxListEntries && plan.unsupported.empty())
                plan.unsupported = "a comp-mul list of " + std::to_string(comp_mul_count(d.vec_space_dim, contribs)) +
                                   " entries exceeds this back end's table budget";
            if (!plan.unsupported.empty()) return;
            generated.reserve(size_t(comp_mul_count(d.vec_space_dim, contribs)));
            for_each_comp_mul(bt, d.metric_diag, contribs,
                              [&](const gaast_comp_mul& m) { generated.push_back(m); });
            muls = generated.data();
            n_muls = generated.size();
        }
        if (n_muls == 0) return;  // nothing is added to res
        if (ll.row_len > 65536 || lrr.row_len > 65536) throw std::runtime_error("operand row too long");

        const bool beta0 = is_fresh;
        // rows: with beta0 every component of the result row gets a row (empty rows write 0.0)
        std::vector<int32_t> row_of(size_t(lr.row_len), -1);
        std::vector<uint32_t> row_out, counts;
        if (beta0)
            for (int64_t o = 0; o < lr.row_len; ++o) {
                row_of[size_t(o)] = int32_t(row_out.size());
                row_out.push_back(uint32_t(o));
                counts.push_back(0);
            }
        std::vector<uint32_t> eo(static_cast<size_t>(n_muls), 0u);  // output offset of each entry
        for (uint64_t e = 0; e < n_muls; ++e) {
            const gaast_comp_mul& m = muls[e];
            auto check = [&](const Layout& lay, uint32_t g, uint32_t i, const char* what) -> int64_t {
                if (g >= 64 || !((lay.mask >> g) & 1ULL)) {
                    fail(GAAST_ERR_MISSING_GRADE, std::string("grade absent from product ") + what);
                    return -1;
                }
                if (int64_t(i) >= lay.grade_len(int(g))) throw std::runtime_error("comp-mul index out of range");
                return lay.offset(int(g)) + i;
            };
            const int64_t lo = check(ll, m.left_grade, m.left_index, "left operand");
            const int64_t ro = check(lrr, m.right_grade, m.right_index, "right operand");
            const int64_t oo = check(lr, m.result_grade, m.result_index, "result");
            if (lo < 0 || ro < 0 || oo < 0) return;
            if (row_of[size_t(oo)] < 0) {
                row_of[size_t(oo)] = int32_t(row_out.size());
                row_out.push_back(uint32_t(oo));
                counts.push_back(0);
            }
            counts[size_t(row_of[size_t(oo)])]++;
            eo[size_t(e)] = uint32_t(oo);
        }
        if (beta0) removed[size_t(fr->second)] = 1;
        Step& s = emit(Step::PRODUCT_CSR, res, "product_csr[" + std::to_string(n_muls) + " comp-muls]");
        s.a = l;
        s.b = r;
        s.canon_a = canon_l;
        s.canon_b = canon_r;
        s.beta = beta0 ? 0 : 1;
        s.n_entries = n_muls;
        s.u32_a.assign(row_out.size() + 1, 0);
        for (size_t i = 0; i < counts.size(); ++i) s.u32_a[i + 1] = s.u32_a[i] + counts[i];
        s.u32_b = row_out;
        s.u32_c.resize(size_t(n_muls));
        s.coeff.resize(size_t(n_muls));
        std::vector<uint32_t> cursor(s.u32_a.begin(), s.u32_a.end() - 1);
        for (uint64_t e = 0; e < n_muls; ++e) {  // stable: keeps the reference order per output
            const gaast_comp_mul& m = muls[e];
            const uint32_t pos = cursor[size_t(row_of[eo[size_t(e)]])]++;
            const uint32_t lo = uint32_t(ll.offset(int(m.left_grade)) + m.left_index);
            const uint32_t ro = uint32_t(lrr.offset(int(m.right_grade)) + m.right_index);
            s.u32_c[pos] = lo | (ro << 16);
            const bool neg = (((flip_l >> m.left_grade) ^ (flip_r >> m.right_grade)) & 1ULL) != 0;
            s.coeff[pos] = neg ? -m.coeff : m.coeff;
        }
        touch(res);
    }
};


// ---------------------------------------------------------------------------------------------
// Whole-plan fusion for small programs: every buffer becomes a region of a per-item LDS slab and
// the steps become one micro-op stream (k_ast_fused).  Exact: same operations, same order.
// ---------------------------------------------------------------------------------------------
// slab_probe != nullptr: only report the slab size a fused plan would have (0: cannot be fused) and change nothing
bool try_fuse(Plan& plan, int* slab_probe = nullptr) {
    if (slab_probe) *slab_probe = 0;
    if (plan.flags & GAAST_FLAG_NO_FUSION) return false;
    if (plan.error != GAAST_OK || plan.steps.empty()) return false;
    const size_t elem = plan.dtype == GAAST_F32 ? 4 : 8;
    // which buffers are touched, and how inputs are read
    std::vector<int> in_direct(plan.inputs.size(), 0), in_axpy(plan.inputs.size(), 0);
    for (const Step& s : plan.steps) {
        if (s.kind == Step::PRODUCT_DENSE || s.kind == Step::FUSED) return false;
        if (s.kind == Step::AXPY) in_axpy[size_t(s.a.idx)] = 1;
        if (s.kind == Step::PRODUCT_CSR) {
            if (s.a.kind == BufKind::INPUT) (s.canon_a ? in_direct : in_axpy)[size_t(s.a.idx)] = 1;
            if (s.b.kind == BufKind::INPUT) (s.canon_b ? in_direct : in_axpy)[size_t(s.b.idx)] = 1;
        }
    }
    Step f;
    f.kind = Step::FUSED;
    f.res = BufRef{BufKind::OUT, 0};
    int cursor = 0;
    // an input read both ways gets two images: the raw rows (add_grades_from) and 0.0 + x
    // (the zero-init + copy the reference makes of a product operand)
    std::vector<int> in_base(plan.inputs.size(), -1), in_base_canon(plan.inputs.size(), -1);
    std::vector<int> node_base(plan.node_buffers.size(), -1);
    for (size_t i = 0; i < plan.inputs.size(); ++i) {
        if (plan.input_layouts[i].row_len == 0) continue;
        if (in_axpy[i]) {
            in_base[i] = cursor;
            f.fused_inputs.push_back({int(i), cursor, 0});
            cursor += int(plan.input_layouts[i].row_len);
        }
        if (in_direct[i]) {
            in_base_canon[i] = cursor;
            f.fused_inputs.push_back({int(i), cursor, 1});
            cursor += int(plan.input_layouts[i].row_len);
        }
    }
    if (f.fused_inputs.size() > size_t(uop::MAX_INPUTS)) return false;
    for (size_t i = 0; i < plan.node_buffers.size(); ++i) {
        node_base[i] = cursor;
        cursor += int(plan.node_buffers[i].row_len);
    }
    const int out_base = cursor;
    cursor += int(plan.out_layout.row_len);
    if (plan.out_layout.row_len == 0) return false;
    const int zero_slot = cursor++;  // one element per item holding +0.0: target of unused MAC slots
    int slab = cursor | 1;  // odd: 64 lanes at one slab offset hit 64 different banks
    // the LDS interpreter kernel needs the slabs of 64 items in 48 KiB; the hiprtc-specialised kernel keeps the slab
    // in registers and only needs it to be small enough for that -- plans that fit only the latter are fused
    // "JIT only" (the runtime falls back to an unfused plan if the compilation fails)
    // (plans with exp / log steps have no interpreter micro-ops: the specialised kernel or nothing)
    // (round 3: 144 KiB instead of 48 -- one 512-thread workgroup per CU -- so that programs whose slab is beyond the registers of
    //  the specialised kernel but whose lists are short still run as ONE launch: the projected sandwich (R X ~R).g(1) at n = 7, 8
    //  has two lists of n 2^(n-1) entries over a slab of 2^n + 2 n elements; as two list launches it ran 8 active lanes per item)
    //  -- for plans of SEVERAL steps only: a single big list is better off on k_product_ell (twice the terms per second)
    const size_t interp_budget = plan.steps.size() >= 2 ? kInterpLdsBytes : size_t(48 * 1024);
    const bool interp_ok = !(slab > 4095 || size_t(slab) * elem > 32767 || size_t(slab) * elem * 64 > interp_budget) && !plan.has_explog;
    // One item per thread, the slab in registers: up to 160 (f64) / 200 (f32) elements always; up to 256 / 320 ON TRIAL -- the
    // compiler keeps only the LIVE values in registers, the projection (v & bv) & bv.vinv() at n = 12 (slab 171) compiles to 222
    // registers and runs at 0.75 of the HBM roof against 0.44 with its slabs in LDS, the versor inverse at n = 8 (slab 259: the
    // whole row is live until it is scaled) to 310 with one wave per SIMD and 0.46 against 0.67.  The runtime measures the compiled
    // kernel's occupancy and rebuilds the plan with GAAST_FLAG_INTERNAL_SMALL_REG_SLAB when the trial fails.
    const int jit_slab_small = plan.dtype == GAAST_F32 ? 200 : 160;
    const int jit_slab_limit = (plan.flags & GAAST_FLAG_INTERNAL_SMALL_REG_SLAB) ? jit_slab_small : (plan.dtype == GAAST_F32 ? 320 : 256);
    const bool jit_allowed = !(plan.flags & GAAST_FLAG_NO_JIT) && slab <= jit_slab_limit;
    if (!interp_ok && !jit_allowed) return false;
    if (slab_probe) {
        *slab_probe = slab;
        return false;
    }
    auto base_of = [&](BufRef r, int canon = 0) {
        return r.kind == BufKind::NODE    ? node_base[size_t(r.idx)]
               : r.kind == BufKind::INPUT ? (canon ? in_base_canon : in_base)[size_t(r.idx)]
                                          : out_base;
    };
    auto layout_of = [&](BufRef r) -> const Layout& {
        return r.kind == BufKind::NODE ? plan.node_buffers[size_t(r.idx)]
               : r.kind == BufKind::INPUT ? plan.input_layouts[size_t(r.idx)] : plan.out_layout;
    };
    // One phase per step; inside a phase the independent result rows (and element-wise ops) are
    // dealt to the workgroup's waves, least-loaded first.  Any split is exact: rows of one
    // Product never read what another row of the same Product writes.
    constexpr int G = uop::GROUPS;
    constexpr uint32_t LW = 32;                   // words per line
    std::vector<uint32_t>& prog = f.u32_a;        // 32-word lines, see kernels.hip.hpp
    std::vector<uint32_t>& phase_tab = f.u32_b;   // per (phase, wave): first line, line count
    std::vector<double>& general = f.coeff;
    uint64_t entries = 0;
    const uint32_t esz = uint32_t(elem);
    auto mop = [](uint32_t code, uint32_t lo, uint32_t mid = 0) { return (code << 28) | (mid << 12) | lo; };
    for (const Step& s : plan.steps) {
        std::vector<std::vector<uint32_t>> glines(G);  // lines of each wave, this phase
        std::vector<uint64_t> load(G, 0);
        auto least = [&]() {
            int g = 0;
            for (int i = 1; i < G; ++i)
                if (load[size_t(i)] < load[size_t(g)]) g = i;
            return g;
        };
        auto push_misc = [&](const std::vector<uint32_t>& ops) {  // chunks of <= 30 ops, dealt round
            const size_t per = std::max<size_t>(1, std::min<size_t>(30, (ops.size() + G - 1) / G));
            for (size_t i = 0; i < ops.size(); i += per) {
                const size_t cnt = std::min(per, ops.size() - i);
                const int g = least();
                std::vector<uint32_t>& out = glines[size_t(g)];
                out.push_back((uint32_t(uop::LINE_MISC) << 28) | (uint32_t(cnt) << 15));
                out.push_back(0u);
                for (size_t k = 0; k < 30; ++k) out.push_back(k < cnt ? ops[i + k] : 0u);
                load[size_t(g)] += cnt;
            }
        };
        const uint32_t rb = uint32_t(base_of(s.res));
        std::vector<uint32_t> misc;
        switch (s.kind) {
        case Step::ZERO: {
            const uint32_t len = uint32_t(layout_of(s.res).row_len);
            for (uint32_t o = 0; o < len; o += 8) misc.push_back(mop(uop::ZERO, rb + o, std::min<uint32_t>(8, len - o)));
            push_misc(misc);
            break;
        }
        case Step::AXPY:
            for (uint32_t m : s.u32_a) misc.push_back(mop(s.beta ? uop::ADD : uop::COPY, rb + (m & 0xffffu), uint32_t(base_of(s.a)) + (m >> 16)));   // (COPY: the zero fill folded in)
            push_misc(misc);
            break;
        case Step::FLIP:
            for (uint32_t o : s.u32_a) misc.push_back(mop(uop::NEG, rb + o));
            push_misc(misc);
            break;
        case Step::SUNARY:
            misc.push_back(mop(s.sunary_op == 0 ? uop::INV : uop::SQRT, rb + uint32_t(s.sunary_off)));
            push_misc(misc);
            break;
        case Step::PRODUCT_CSR: {
            const uint32_t lb = uint32_t(base_of(s.a, s.canon_a)), rrb = uint32_t(base_of(s.b, s.canon_b));
            for (size_t row = 0; row + 1 < s.u32_a.size(); ++row) {
                const uint32_t dst = rb + s.u32_b[row];
                const uint32_t e0 = s.u32_a[row], e1 = s.u32_a[row + 1];
                const int g = least();
                std::vector<uint32_t>& out = glines[size_t(g)];
                load[size_t(g)] += (e1 - e0) + 2;
                bool row_general = false;
                for (uint32_t e = e0; e < e1; ++e) row_general |= (s.coeff[e] != 1.0 && s.coeff[e] != -1.0);
                // split long rows evenly over their lines (16 entries -> 8 + 8, not 10 + 6)
                const uint32_t n_l = e1 > e0 ? (e1 - e0 + 9) / 10 : 1;
                const uint32_t per = e1 > e0 ? (e1 - e0 + n_l - 1) / n_l : 0;
                uint32_t e = e0;
                do {  // a row with no entries still stores its (fresh) 0.0
                    const uint32_t cnt = std::min<uint32_t>(per, e1 - e);
                    uint32_t hdr = dst | (cnt << 15);
                    if (e == e0) hdr |= (1u << 12) | (s.beta ? 0u : (1u << 13));
                    if (e + cnt == e1) hdr |= 1u << 14;
                    hdr |= uint32_t(row_general ? uop::LINE_MACS_GEN : uop::LINE_MACS) << 28;
                    const size_t line0 = out.size();
                    out.resize(line0 + LW, 0u);
                    out[line0] = hdr;
                    for (uint32_t k = 0; k < cnt; ++k) {
                        const double c = s.coeff[e + k];
                        const uint32_t lo = lb + (s.u32_c[e + k] & 0xffffu), ro = rrb + (s.u32_c[e + k] >> 16);
                        if (!row_general) {
                            out[line0 + 2 + 3 * k] = lo * esz;
                            out[line0 + 3 + 3 * k] = ro * esz;
                            out[line0 + 4 + 3 * k] = c == -1.0 ? 0x80000000u : 0u;
                        } else {
                            uint32_t ci;
                            if (c == 1.0) {
                                ci = 0;
                            } else if (c == -1.0) {
                                ci = 1;
                            } else {
                                size_t gi = 0;
                                for (; gi < general.size(); ++gi)
                                    if (std::memcmp(&general[gi], &c, sizeof(double)) == 0) break;
                                if (gi == general.size()) {
                                    if (general.size() == size_t(uop::MAX_GENERAL_COEFFS)) return false;
                                    general.push_back(c);
                                }
                                ci = uint32_t(gi) + 2;
                            }
                            out[line0 + 2 + 3 * k] = lo | (ro << 12);
                            out[line0 + 3 + 3 * k] = ci;
                        }
                        ++entries;
                    }
                    e += cnt;
                } while (e < e1);
            }
            break;
        }
        case Step::EXPLOG: break;   // specialised kernel only (interp_ok is false)
        default: return false;
        }
        for (int g = 0; g < G; ++g) {
            phase_tab.push_back(uint32_t(prog.size() / LW));
            phase_tab.push_back(uint32_t(glines[size_t(g)].size() / LW));
            prog.insert(prog.end(), glines[size_t(g)].begin(), glines[size_t(g)].end());
        }
    }
    if (prog.size() > (1u << 20)) return false;
    if (prog.empty()) prog.assign(LW, uint32_t(uop::LINE_NOP) << 28);
    // ---- the same plan as straight-line HIP source, specialised at program_create through
    // hiprtc (the reference's README lists code generation from the specialized AST as roadmap).
    // lane <-> item, every slab element is a local scalar (a register), offsets and signs are
    // constants, the statements are the reference's in the reference's order; the runtime
    // compiles it with -ffp-contract=off so that the roundings stay those of eval.rs:82.
    if (!interp_ok && entries > 8192) return false;
    f.fused_jit_only = interp_ok ? 0 : 1;
    if (jit_allowed && entries <= 8192) {
        std::string src;
        char buf[256];
        const char* ty = plan.dtype == GAAST_F32 ? "float" : "double";
        auto lit = [&](double c) {
            std::snprintf(buf, sizeof(buf), plan.dtype == GAAST_F32 ? "%af" : "%a", plan.dtype == GAAST_F32 ? double(float(c)) : c);
            return std::string(buf);
        };
        auto var = [&](uint32_t i) { return "v" + std::to_string(i); };
        // Row I/O.  lane <-> item, but a lane reading ITS row with 16-byte accesses makes every wave instruction touch 64
        // different 128-byte lines: the CU's L1 then spends a cycle pair per line for 16 useful bytes and bounds the kernel
        // (config 5: 56 % of HBM peak with the vector units 30 % busy).  So a wave (= a workgroup of 64 lanes) moves the
        // rows of its 64 items as ONE contiguous span with fully coalesced 16-byte accesses and transposes through LDS:
        // rows padded to an odd number of 16-byte units, so that both the span-ordered and the row-per-lane accesses are
        // conflict-free.  Used per operand when its rows are contiguous (stride == length) and 16-byte aligned and the wave
        // is full; otherwise (shared rows, strided or unaligned wrapped memory, the last partial wave) the lane reads its row
        // directly.  GAAST_FLAG_NO_COALESCE: always the direct form (A/B measurements).
        const size_t esz = plan.dtype == GAAST_F32 ? 4 : 8;
        const int epc = int(16 / esz);                                   // elements per 16-byte chunk
        auto padded_len = [&](int len) {                                 // row length in LDS, elements
            size_t padb = (size_t(len) * esz + 15) / 16 * 16;
            if ((padb / 16) % 2 == 0) padb += 16;
            return int(padb / esz);
        };
        // which operands go through LDS: largest rows first, within a budget that keeps 16 waves per CU resident
        // (160 KiB / 16 = 10 KiB per wave); the result rows reuse the operands' space.  Operands left out (and rows
        // too long for the budget) are read by their lanes directly.
        const size_t lds_budget = (plan.flags & GAAST_FLAG_DEBUG_LDS_12K) ? 12 * 1024 + 256 : 10 * 1024;
        std::vector<int> lds_off(f.fused_inputs.size(), -1);
        size_t lds_in = 0;
        {
            std::vector<size_t> order(f.fused_inputs.size());
            for (size_t i = 0; i < order.size(); ++i) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) {
                return plan.input_layouts[size_t(f.fused_inputs[x].slot)].row_len > plan.input_layouts[size_t(f.fused_inputs[y].slot)].row_len;
            });
            for (size_t i : order) {
                const size_t need = size_t(64) * size_t(padded_len(int(plan.input_layouts[size_t(f.fused_inputs[i].slot)].row_len))) * esz;
                if (lds_in + need > lds_budget) continue;
                lds_off[i] = int(lds_in);
                lds_in += need;
            }
        }
        const int out_len = int(plan.out_layout.row_len);
        const size_t lds_out_need = size_t(64) * size_t(padded_len(out_len)) * esz;
        const bool out_via_lds = lds_out_need <= lds_budget;
        // Rows too long for the budget go through ONE shared 64 x 144-byte buffer, 128 bytes (a cache line) of every
        // row at a time: 8 lanes move one row's line, a wave instruction 8 whole lines (so the rows need not even be
        // contiguous, only 16-byte aligned with a 16-byte multiple stride).  Lines of a row are consumed one after the
        // other: a few more barriers (single-wave workgroups: cheap), the same coalescing.
        const int line_plen = int(144 / esz);   // padded line in elements: 9 x 16 bytes, odd -> conflict-free both ways
        // (programs with big slabs are register-bound: the transposition's temporaries would spill -- measured on full R^6
        // f32 products, 193 elements: 1.93 -> 1.30 TB/s -- so they keep the row-per-lane form)
        auto line_ok = [&](int len) { return slab <= 128 && size_t(len) * esz >= 128 && (size_t(len) * esz) % 16 == 0; };
        std::vector<char> by_line(f.fused_inputs.size(), 0);
        bool any_line = false;
        for (size_t i = 0; i < f.fused_inputs.size(); ++i)
            if (lds_off[i] < 0 && line_ok(int(plan.input_layouts[size_t(f.fused_inputs[i].slot)].row_len))) by_line[i] = 1, any_line = true;
        const bool out_by_line = !out_via_lds && line_ok(out_len);
        any_line = any_line || out_by_line;
        const size_t line_bytes = any_line ? size_t(64) * 144 : 0;
        if (any_line && lds_in + line_bytes > 12 * 1024 + 1024) {   // keep >= 12 waves per CU resident: drop the line path
            std::fill(by_line.begin(), by_line.end(), 0);
            any_line = false;
        }
        const size_t line_off = lds_in;                      // the shared line buffer sits after the span regions
        const bool out_line = any_line && out_by_line;
        const size_t lds_total = std::max(lds_in + (any_line ? line_bytes : 0), out_via_lds ? lds_out_need : size_t(0));
        const bool coalesce = !(plan.flags & GAAST_FLAG_NO_COALESCE) && lds_total > 0;
        const int threads = coalesce ? 64 : 256;
        src += std::string("typedef ") + ty + " T;\n";
        src += std::string("typedef ") + ty + " VT __attribute__((ext_vector_type(" + std::to_string(epc) + ")));\n";
        src += "extern \"C\" __global__ __launch_bounds__(" + std::to_string(threads) + ") void gaast_jit(";
        for (size_t i = 0; i < f.fused_inputs.size(); ++i)
            src += "const T* __restrict__ in" + std::to_string(i) + ", long long s" + std::to_string(i) + ", ";
        src += std::string("T* __restrict__ out, long long so, long long batch") + (plan.has_explog ? ", unsigned long long* dom" : "") + ") {\n";
        if (coalesce) {
            src += "  __shared__ __attribute__((aligned(16))) unsigned char lds[" + std::to_string(lds_total) + "];\n";
            src += "  const int lane = threadIdx.x;\n  const long long item0 = blockIdx.x * 64LL;\n";
            src += "  const long long item = item0 + lane;\n  const bool live = item < batch;\n  const bool full = item0 + 64 <= batch;\n";
        } else {
            src += "  const long long item = blockIdx.x * 256LL + threadIdx.x;\n  if (item >= batch) return;\n";
        }
        for (int i = 0; i < slab; ++i) src += "  T " + var(uint32_t(i)) + " = 0;\n";
        for (size_t i = 0; i < f.fused_inputs.size(); ++i) {
            const Step::FusedInput& fi = f.fused_inputs[i];
            const int len = int(plan.input_layouts[size_t(fi.slot)].row_len);
            const std::string I = std::to_string(i);
            auto assign = [&](const std::string& from_prefix, const std::string& indent) {
                for (int c = 0; c < len; ++c)
                    src += indent + var(uint32_t(fi.base + c)) + (fi.canon ? " = T(0) + " : " = ") + from_prefix + "[" + std::to_string(c) + "];\n";
            };
            if (!coalesce) {
                src += "  { const T* r = in" + I + " + item * s" + I + ";\n";
                assign("r", "    ");
                src += "  }\n";
                continue;
            }
            if (coalesce && by_line[i]) {   // one cache line of every row at a time through the shared buffer
                const int nblk = int((size_t(len) * esz + 127) / 128);
                src += "  if (full && ((s" + I + " * " + std::to_string(esz) + ") & 15) == 0 && (((unsigned long long)in" + I + ") & 15ull) == 0) {\n";
                src += "    T* buf = (T*)(lds + " + std::to_string(line_off) + ");\n";
                for (int b = 0; b < nblk; ++b) {
                    const int blk_bytes = int(std::min<size_t>(128, size_t(len) * esz - size_t(128) * b));
                    const int cpl = blk_bytes / 16, blk_elems = int(blk_bytes / esz), e0 = int(size_t(128) * b / esz);
                    src += "    { const T* base = in" + I + " + item0 * s" + I + " + " + std::to_string(e0) + " + (lane & 7) * " + std::to_string(epc) + ";\n";
                    for (int j = 0; j < 8; ++j) src += "      VT c" + std::to_string(j) + ";\n";
                    src += std::string("      if ((lane & 7) < ") + std::to_string(cpl) + ") {\n";
                    for (int j = 0; j < 8; ++j)
#if GAAST_JIT_NT & 2
                        src += "        c" + std::to_string(j) + " = __builtin_nontemporal_load((const VT*)(base + (long long)(" + std::to_string(8 * j) + " + (lane >> 3)) * s" + I + "));\n";
#else
                        src += "        c" + std::to_string(j) + " = *(const VT*)(base + (long long)(" + std::to_string(8 * j) + " + (lane >> 3)) * s" + I + ");\n";
#endif
                    for (int j = 0; j < 8; ++j)
                        src += "        *(VT*)(buf + (" + std::to_string(8 * j) + " + (lane >> 3)) * " + std::to_string(line_plen) + " + (lane & 7) * " + std::to_string(epc) + ") = c" + std::to_string(j) + ";\n";
                    src += "      }\n      __syncthreads();\n      const T* r = buf + lane * " + std::to_string(line_plen) + ";\n";
                    for (int e = 0; e < blk_elems; ++e)
                        src += "      " + var(uint32_t(fi.base + e0 + e)) + (fi.canon ? " = T(0) + r[" : " = r[") + std::to_string(e) + "];\n";
                    src += "      __syncthreads();\n    }\n";
                }
                src += "  } else if (live) {\n    const T* r = in" + I + " + item * s" + I + ";\n";
                assign("r", "    ");
                src += "  }\n";
                continue;
            }
            if (lds_off[i] < 0) {   // not staged: the lane reads its own row
                src += "  if (live) { const T* r = in" + I + " + item * s" + I + ";\n";
                assign("r", "    ");
                src += "  }\n";
                continue;
            }
            const int plen = padded_len(len);
            const int nch = 64 * len / epc;                  // 16-byte chunks of the wave's span (64 * len * esz is a multiple of 256)
            const int per_lane = (nch + 63) / 64;
            const bool whole = (size_t(len) * esz) % 16 == 0;   // a chunk never straddles two rows
            src += "  if (full && s" + I + " == " + std::to_string(len) + " && (((unsigned long long)in" + I + ") & 15ull) == 0) {\n";
            src += "    const VT* src" + I + " = (const VT*)(in" + I + " + item0 * " + std::to_string(len) + ");\n";
            src += "    T* img = (T*)(lds + " + std::to_string(lds_off[i]) + ");\n";
            for (int j = 0; j < per_lane; ++j) src += "    VT c" + std::to_string(j) + ";\n";
            for (int j = 0; j < per_lane; ++j) {
                const bool guard = (j + 1) * 64 > nch;
                src += std::string("    ") + (guard ? "if (lane + " + std::to_string(64 * j) + " < " + std::to_string(nch) + ") " : "") + "c" +
#if GAAST_JIT_NT & 2
                       std::to_string(j) + " = __builtin_nontemporal_load(&src" + I + "[lane + " + std::to_string(64 * j) + "]);\n";
#else
                       std::to_string(j) + " = src" + I + "[lane + " + std::to_string(64 * j) + "];\n";
#endif
            }
            for (int j = 0; j < per_lane; ++j) {
                const bool guard = (j + 1) * 64 > nch;
                src += std::string("    ") + (guard ? "if (lane + " + std::to_string(64 * j) + " < " + std::to_string(nch) + ") " : "") + "{ const int el = (lane + " +
                       std::to_string(64 * j) + ") * " + std::to_string(epc) + ";\n";
                if (whole) {
                    src += "      *(VT*)(img + (el / " + std::to_string(len) + ") * " + std::to_string(plen) + " + el % " + std::to_string(len) + ") = c" + std::to_string(j) + ";\n";
                } else {
                    for (int e = 0; e < epc; ++e)
                        src += "      img[((el + " + std::to_string(e) + ") / " + std::to_string(len) + ") * " + std::to_string(plen) + " + (el + " + std::to_string(e) + ") % " +
                               std::to_string(len) + "] = c" + std::to_string(j) + "[" + std::to_string(e) + "];\n";
                }
                src += "    }\n";
            }
            src += "    __syncthreads();\n    const T* r = img + lane * " + std::to_string(plen) + ";\n";
            assign("r", "    ");
            src += "  } else if (live) {\n    const T* r = in" + I + " + item * s" + I + ";\n";
            assign("r", "    ");
            src += "  }\n";
        }
        for (const Step& s : plan.steps) {
            const uint32_t rb = uint32_t(base_of(s.res));
            switch (s.kind) {
            case Step::ZERO:
                for (int64_t o = 0; o < layout_of(s.res).row_len; ++o) src += "  " + var(rb + uint32_t(o)) + " = T(0);\n";
                break;
            case Step::AXPY:
                for (uint32_t m : s.u32_a) {
                    const std::string d = var(rb + (m & 0xffffu));
                    src += "  " + d + " = " + (s.beta ? d : std::string("T(0)")) + " + " + var(uint32_t(base_of(s.a)) + (m >> 16)) + ";\n";
                }
                break;
            case Step::FLIP:
                for (uint32_t o : s.u32_a) src += "  " + var(rb + o) + " = -" + var(rb + o) + ";\n";
                break;
            case Step::SUNARY: {
                const std::string d = var(rb + uint32_t(s.sunary_off));
                if (s.sunary_op == 0)
                    src += "  " + d + " = T(1) / " + d + ";\n";
                else
                    src += "  " + d + (plan.dtype == GAAST_F32 ? " = __builtin_sqrtf(" : " = __builtin_sqrt(") + d + ");\n";
                break;
            }
            case Step::EXPLOG: {   // the statements of oracle/gaast_oracle.c: ext_exp_log, in its order
                const uint32_t ab = uint32_t(base_of(s.a));
                const bool f32 = plan.dtype == GAAST_F32;
                auto fn = [&](const char* name) { return std::string(name) + (f32 ? "f" : ""); };
                auto B = [&](uint32_t i) { return var(ab + uint32_t(s.explog_arg_k) + i); };
                src += "  { T sq = T(0), nrm = T(0), viol = T(0);\n";
                for (int i = 0; i < s.explog_m; ++i) {
                    src += "    sq = sq + " + B(uint32_t(i)) + " * " + B(uint32_t(i)) + " * T(" + lit(s.coeff[size_t(i)]) + ");\n";
                    src += "    nrm = nrm + " + B(uint32_t(i)) + " * " + B(uint32_t(i)) + ";\n";
                }
                for (size_t row = 0; row + 1 < s.u32_a.size(); ++row) {
                    src += "    { T acc = T(0);\n";
                    for (uint32_t e = s.u32_a[row]; e < s.u32_a[row + 1]; ++e)
                        src += "      acc = acc + " + B(s.u32_c[e] & 0xffffu) + " * " + B(s.u32_c[e] >> 16) + " * T(" + lit(s.coeff_b[e]) + ");\n";
                    src += "      viol = viol + acc * acc; }\n";
                }
                if (s.u32_a.size() > 1)
                    src += std::string("    if (viol > T(") + lit(9.094947017729282e-13) + ") * (nrm * nrm)) atomicAdd(dom, 1ull);\n";
                src += "    T c0 = T(0), f;\n";
                if (s.explog_op == 0) {
                    src += "    if (sq < T(0)) { const T t = " + fn("sqrt") + "(-sq); c0 = " + fn("cos") + "(t); f = " + fn("sin") + "(t) / t; }\n";
                    src += "    else if (sq > T(0)) { const T t = " + fn("sqrt") + "(sq); c0 = " + fn("cosh") + "(t); f = " + fn("sinh") + "(t) / t; }\n";
                    src += "    else if (sq == T(0)) { c0 = T(1); f = T(1); }\n    else { c0 = sq; f = sq; }\n";
                } else {
                    const std::string a = s.explog_arg_0 >= 0 ? var(ab + uint32_t(s.explog_arg_0)) : std::string("T(0)");
                    src += "    if (sq < T(0)) { const T mm = " + fn("sqrt") + "(-sq); f = " + fn("atan2") + "(mm, " + a + ") / mm; }\n";
                    src += "    else if (sq > T(0)) { const T mm = " + fn("sqrt") + "(sq); f = " + fn("atanh") + "(mm / " + a + ") / mm; }\n";
                    src += "    else if (sq == T(0)) { f = T(1) / " + a + "; }\n    else { f = sq; }\n";
                }
                if (s.explog_res_0 >= 0) {
                    const std::string d = var(rb + uint32_t(s.explog_res_0));
                    src += "    " + d + " = " + d + " + c0;\n";
                }
                if (s.explog_res_k >= 0)
                    for (int i = 0; i < s.explog_mres; ++i) {
                        const std::string d = var(rb + uint32_t(s.explog_res_k) + uint32_t(i));
                        src += "    " + d + " = " + d + " + f * " + B(uint32_t(i)) + ";\n";
                    }
                src += "  }\n";
                break;
            }
            case Step::PRODUCT_CSR: {
                const uint32_t lb = uint32_t(base_of(s.a, s.canon_a)), rrb = uint32_t(base_of(s.b, s.canon_b));
                for (size_t row = 0; row + 1 < s.u32_a.size(); ++row) {
                    const std::string d = var(rb + s.u32_b[row]);
                    src += "  { T acc = " + (s.beta ? d : std::string("T(0)")) + ";\n";
                    for (uint32_t e = s.u32_a[row]; e < s.u32_a[row + 1]; ++e) {
                        const std::string prod = "(" + var(lb + (s.u32_c[e] & 0xffffu)) + " * " + var(rrb + (s.u32_c[e] >> 16)) + ")";
                        const double c = s.coeff[e];
                        if (c == 1.0)
                            src += "    acc = acc + " + prod + ";\n";
                        else if (c == -1.0)
                            src += "    acc = acc - " + prod + ";\n";
                        else
                            src += "    acc = acc + " + prod + " * T(" + lit(c) + ");\n";
                    }
                    src += "    " + d + " = acc; }\n";
                }
                break;
            }
            default: break;
            }
        }
        if (!coalesce) {
            src += "  T* o = out + item * so;\n";
            for (int64_t c = 0; c < plan.out_layout.row_len; ++c)
                src += "  o[" + std::to_string(c) + "] = " + var(uint32_t(out_base + c)) + ";\n";
        } else {
            const int plen = padded_len(out_len);
            const int nch = 64 * out_len / epc;
            const int per_lane = (nch + 63) / 64;
            const bool whole = (size_t(out_len) * esz) % 16 == 0;
            if (out_line) {
                const int nblk = int((size_t(out_len) * esz + 127) / 128);
                src += "  if (full && ((so * " + std::to_string(esz) + ") & 15) == 0 && (((unsigned long long)out) & 15ull) == 0) {\n";
                src += "    T* buf = (T*)(lds + " + std::to_string(line_off) + ");\n";
                for (int b = 0; b < nblk; ++b) {
                    const int blk_bytes = int(std::min<size_t>(128, size_t(out_len) * esz - size_t(128) * b));
                    const int cpl = blk_bytes / 16, blk_elems = int(blk_bytes / esz), e0 = int(size_t(128) * b / esz);
                    src += "    { __syncthreads();\n      T* r = buf + lane * " + std::to_string(line_plen) + ";\n";
                    for (int e = 0; e < blk_elems; ++e) src += "      r[" + std::to_string(e) + "] = " + var(uint32_t(out_base + e0 + e)) + ";\n";
                    src += "      __syncthreads();\n      T* base = out + item0 * so + " + std::to_string(e0) + " + (lane & 7) * " + std::to_string(epc) + ";\n";
                    src += std::string("      if ((lane & 7) < ") + std::to_string(cpl) + ") {\n";
                    for (int j = 0; j < 8; ++j)
#if GAAST_JIT_NT & 1
                        src += "        __builtin_nontemporal_store(*(const VT*)(buf + (" + std::to_string(8 * j) + " + (lane >> 3)) * " + std::to_string(line_plen) + " + (lane & 7) * " +
                               std::to_string(epc) + "), (VT*)(base + (long long)(" + std::to_string(8 * j) + " + (lane >> 3)) * so));\n";
#else
                        src += "        *(VT*)(base + (long long)(" + std::to_string(8 * j) + " + (lane >> 3)) * so) = *(const VT*)(buf + (" + std::to_string(8 * j) +
                               " + (lane >> 3)) * " + std::to_string(line_plen) + " + (lane & 7) * " + std::to_string(epc) + ");\n";
#endif
                    src += "      }\n    }\n";
                }
                src += "  } else\n";
            }
            src += std::string("  if (") + (out_via_lds ? "full" : "false") + " && so == " + std::to_string(out_len) + " && (((unsigned long long)out) & 15ull) == 0) {\n";
            src += "    __syncthreads();\n    T* img = (T*)lds;\n    { T* r = img + lane * " + std::to_string(plen) + ";\n";
            for (int c = 0; c < out_len; ++c) src += "      r[" + std::to_string(c) + "] = " + var(uint32_t(out_base + c)) + ";\n";
            src += "    }\n    __syncthreads();\n    VT* dst = (VT*)(out + item0 * " + std::to_string(out_len) + ");\n";
            for (int j = 0; j < per_lane; ++j) {
                const bool guard = (j + 1) * 64 > nch;
                src += std::string("    ") + (guard ? "if (lane + " + std::to_string(64 * j) + " < " + std::to_string(nch) + ") " : "") + "{ const int el = (lane + " +
                       std::to_string(64 * j) + ") * " + std::to_string(epc) + ";\n      VT c;\n";
                if (whole) {
                    src += "      c = *(const VT*)(img + (el / " + std::to_string(out_len) + ") * " + std::to_string(plen) + " + el % " + std::to_string(out_len) + ");\n";
                } else {
                    for (int e = 0; e < epc; ++e)
                        src += "      c[" + std::to_string(e) + "] = img[((el + " + std::to_string(e) + ") / " + std::to_string(out_len) + ") * " + std::to_string(plen) +
                               " + (el + " + std::to_string(e) + ") % " + std::to_string(out_len) + "];\n";
                }
#if GAAST_JIT_NT & 1
                src += "      __builtin_nontemporal_store(c, &dst[lane + " + std::to_string(64 * j) + "]);\n    }\n";
#else
                src += "      dst[lane + " + std::to_string(64 * j) + "] = c;\n    }\n";
#endif
            }
            src += "  } else if (live) {\n    T* o = out + item * so;\n";
            for (int64_t c = 0; c < plan.out_layout.row_len; ++c)
                src += "    o[" + std::to_string(c) + "] = " + var(uint32_t(out_base + c)) + ";\n";
            src += "  }\n";
        }
        src += "}\n";
        f.jit_threads = threads;
        f.jit_source = std::move(src);
        f.jit_reg_trial = slab > jit_slab_small;
    } else if (!(plan.flags & GAAST_FLAG_NO_JIT) && interp_ok && !plan.has_explog && entries <= 2048 &&
               size_t(slab | 1) * elem * 64 + 64 <= kLdsBytes) {
        // ---- MEDIUM programs (round 4): the slab is beyond the registers of the specialised kernel above (160 / 200 elements) but the
        // program is short -- the versor inverse a.rev() * a.norm_sq().sinv() at n = 8 (slab 259, 256 comp-muls), the projection KAT at
        // n = 12 (slab 171) -- and used to run on the LDS interpreter (wave-uniform micro-op decode: 0.16 / 0.10 of the HBM roof).  The
        // same plan as straight-line code over slabs that STAY IN LDS: a workgroup of eight waves owns the slabs of 64 items (item i at
        // i * stride elements, stride odd: the lanes of a wave touch 64 different banks at any slab offset), lane <-> item, and the
        // independent rows of every arm are dealt to the waves, least-loaded first -- exactly the interpreter's schedule, with the
        // decode done by hiprtc: offsets are immediates of the LDS instructions, signs are operators.  Same statements, same order.
        std::string src;
        char buf[256];
        const char* ty = plan.dtype == GAAST_F32 ? "float" : "double";
        auto lit = [&](double c) {
            std::snprintf(buf, sizeof(buf), plan.dtype == GAAST_F32 ? "%af" : "%a", plan.dtype == GAAST_F32 ? double(float(c)) : c);
            return std::string(buf);
        };
        const int stride = slab | 1;
        constexpr int W = 8;   // waves per workgroup
        auto at = [&](uint32_t i) { return "my[" + std::to_string(i) + "]"; };
        src += std::string("typedef ") + ty + " T;\n";
        src += "extern \"C\" __global__ __launch_bounds__(512) void gaast_jit(";
        for (size_t i = 0; i < f.fused_inputs.size(); ++i)
            src += "const T* __restrict__ in" + std::to_string(i) + ", long long s" + std::to_string(i) + ", ";
        src += "T* __restrict__ out, long long so, long long batch) {\n";
        src += "  __shared__ T slab[" + std::to_string(64 * stride) + "];\n";
        src += "  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n";
        src += "  T* const my = slab + lane * " + std::to_string(stride) + ";\n";
        // inputs: element e = item * len + c of the flattened range.  Contiguous, 16-byte aligned rows of a full group of 64 items are
        // moved as 16-byte pieces through registers, ALL of them in flight at once, and -- the workgroups are persistent -- the NEXT
        // group's pieces are requested before this group is evaluated (a group is 64 KiB at slab 259: the HBM latency hides under
        // the arithmetic); anything else -- shared rows, strided wrapped memory, the last partial group -- takes a plain loop
        const int epc = int(16 / elem);
        src += std::string("  typedef T VT __attribute__((ext_vector_type(") + std::to_string(epc) + ")));\n";
        std::string fast_cond = "true";
        int total_chunks_per_thread = 0;
        for (size_t i = 0; i < f.fused_inputs.size(); ++i) {
            const int len = int(plan.input_layouts[size_t(f.fused_inputs[i].slot)].row_len);
            const std::string I = std::to_string(i);
            fast_cond += " && s" + I + " == " + std::to_string(len) + " && (((unsigned long long)in" + I + ") & 15ull) == 0";
            if ((64 * len) % epc) fast_cond += " && false";
            total_chunks_per_thread += (64 * len / epc + 511) / 512;
        }
        if (total_chunks_per_thread > 24) fast_cond = "false";
        src += "  const bool fast = " + fast_cond + ";\n";
        src += "  const long long groups = (batch + 63) / 64;\n";
        for (size_t i = 0; i < f.fused_inputs.size(); ++i) {
            const int len = int(plan.input_layouts[size_t(f.fused_inputs[i].slot)].row_len);
            const int nch = 64 * len / epc, cpt = std::max(1, (nch + 511) / 512);
            src += "  VT r" + std::to_string(i) + "[" + std::to_string(cpt) + "];\n";
        }
        src += "  auto issue = [&](long long item0) {\n";
        for (size_t i = 0; i < f.fused_inputs.size(); ++i) {
            const int len = int(plan.input_layouts[size_t(f.fused_inputs[i].slot)].row_len);
            const int nch = 64 * len / epc, cpt = (nch + 511) / 512;
            const std::string I = std::to_string(i);
            src += "    { const VT* src = (const VT*)(in" + I + " + item0 * " + std::to_string(len) + ");\n";
            for (int k = 0; k < cpt; ++k)
                src += "      if (tid + " + std::to_string(512 * k) + " < " + std::to_string(nch) + ") r" + I + "[" + std::to_string(k) +
                       "] = __builtin_nontemporal_load(src + tid + " + std::to_string(512 * k) + ");\n";
            src += "    }\n";
        }
        src += "  };\n";
        src += "  auto commit = [&]() {\n";
        for (size_t i = 0; i < f.fused_inputs.size(); ++i) {
            const Step::FusedInput& fi = f.fused_inputs[i];
            const int len = int(plan.input_layouts[size_t(fi.slot)].row_len);
            const int nch = 64 * len / epc, cpt = (nch + 511) / 512;
            const std::string I = std::to_string(i), L = std::to_string(len);
            for (int k = 0; k < cpt; ++k) {
                src += "    if (tid + " + std::to_string(512 * k) + " < " + std::to_string(nch) + ") {\n";
                for (int j = 0; j < epc; ++j)
                    src += "      { const int e = (tid + " + std::to_string(512 * k) + ") * " + std::to_string(epc) + " + " + std::to_string(j) + ", i2 = e / " + L + ", c = e - i2 * " + L +
                           "; const T v = r" + I + "[" + std::to_string(k) + "][" + std::to_string(j) + "]; slab[i2 * " + std::to_string(stride) + " + " + std::to_string(fi.base) +
                           " + c] = " + (fi.canon ? "T(0) + v" : "v") + "; }\n";
                src += "    }\n";
            }
        }
        src += "  };\n";
        src += "  auto stage = [&](long long item0, int nitems) {\n";
        for (size_t i = 0; i < f.fused_inputs.size(); ++i) {
            const Step::FusedInput& fi = f.fused_inputs[i];
            const int len = int(plan.input_layouts[size_t(fi.slot)].row_len);
            const std::string I = std::to_string(i), L = std::to_string(len);
            src += "#pragma unroll 4\n    for (int e = tid; e < " + std::to_string(64 * len) + "; e += 512) { const int i2 = e / " + L + ", c = e - i2 * " + L +
                   "; T v = i2 < nitems ? in" + I + "[(item0 + i2) * s" + I + " + c] : T(0); slab[i2 * " + std::to_string(stride) + " + " +
                   std::to_string(fi.base) + " + c] = " + (fi.canon ? "T(0) + v" : "v") + "; }\n";
        }
        src += "  };\n";
        src += "  long long g = blockIdx.x;\n  if (g >= groups) return;\n";
        src += "  { const long long item0 = g * 64; const int nitems = int(batch - item0 < 64 ? batch - item0 : 64);\n";
        src += "    if (fast && nitems == 64) { issue(item0); commit(); } else stage(item0, nitems); }\n";
        src += "  for (;;) {\n";
        src += "  const long long item0 = g * 64;\n";
        src += "  const int nitems = int(batch - item0 < 64 ? batch - item0 : 64);\n";
        src += "  const long long gn = g + gridDim.x;\n  const bool more = gn < groups;\n";
        src += "  const int nnext = more ? int(batch - gn * 64 < 64 ? batch - gn * 64 : 64) : 0;\n";
        src += "  const bool pre_next = fast && more && nnext == 64;\n";
        src += "  __syncthreads();\n";
        src += "  if (pre_next) issue(gn * 64);\n";
        for (const Step& s : plan.steps) {
            const uint32_t rb = uint32_t(base_of(s.res));
            // the independent pieces of this arm, each a (cost, statements) pair, dealt to the waves least-loaded first
            std::vector<std::pair<uint64_t, std::string>> pieces;
            auto elementwise = [&](const std::vector<std::string>& stmts) {   // chunks of up to 16 statements
                for (size_t i = 0; i < stmts.size(); i += 16) {
                    std::string blk;
                    for (size_t k = i; k < std::min(stmts.size(), i + 16); ++k) blk += "      " + stmts[k] + "\n";
                    pieces.emplace_back(std::min<size_t>(16, stmts.size() - i), blk);
                }
            };
            std::vector<std::string> st;
            switch (s.kind) {
            case Step::ZERO:
                for (int64_t o = 0; o < layout_of(s.res).row_len; ++o) st.push_back(at(rb + uint32_t(o)) + " = T(0);");
                elementwise(st);
                break;
            case Step::AXPY:
                for (uint32_t m : s.u32_a) {
                    const std::string d = at(rb + (m & 0xffffu));
                    st.push_back(d + " = " + (s.beta ? d : std::string("T(0)")) + " + " + at(uint32_t(base_of(s.a)) + (m >> 16)) + ";");
                }
                elementwise(st);
                break;
            case Step::FLIP:
                for (uint32_t o : s.u32_a) st.push_back(at(rb + o) + " = -" + at(rb + o) + ";");
                elementwise(st);
                break;
            case Step::SUNARY: {
                const std::string d = at(rb + uint32_t(s.sunary_off));
                st.push_back(s.sunary_op == 0 ? d + " = T(1) / " + d + ";"
                                              : d + (plan.dtype == GAAST_F32 ? " = __builtin_sqrtf(" : " = __builtin_sqrt(") + d + ");");
                elementwise(st);
                break;
            }
            case Step::PRODUCT_CSR: {
                const uint32_t lb = uint32_t(base_of(s.a, s.canon_a)), rrb = uint32_t(base_of(s.b, s.canon_b));
                for (size_t row = 0; row + 1 < s.u32_a.size(); ++row) {
                    const std::string d = at(rb + s.u32_b[row]);
                    std::string blk = "      { T acc = " + (s.beta ? d : std::string("T(0)")) + ";\n";
                    for (uint32_t e = s.u32_a[row]; e < s.u32_a[row + 1]; ++e) {
                        const std::string prod = "(" + at(lb + (s.u32_c[e] & 0xffffu)) + " * " + at(rrb + (s.u32_c[e] >> 16)) + ")";
                        const double c = s.coeff[e];
                        blk += c == 1.0 ? "        acc = acc + " + prod + ";\n"
                               : c == -1.0 ? "        acc = acc - " + prod + ";\n"
                                           : "        acc = acc + " + prod + " * T(" + lit(c) + ");\n";
                    }
                    blk += "        " + d + " = acc; }\n";
                    pieces.emplace_back(uint64_t(s.u32_a[row + 1] - s.u32_a[row]) + 2, blk);
                }
                break;
            }
            default: return false;
            }
            std::vector<std::string> per_wave(W);
            std::vector<uint64_t> load(W, 0);
            for (const auto& pc : pieces) {
                int g = 0;
                for (int i = 1; i < W; ++i)
                    if (load[size_t(i)] < load[size_t(g)]) g = i;
                per_wave[size_t(g)] += pc.second;
                load[size_t(g)] += pc.first;
            }
            src += "  switch (wave) {\n";
            for (int g = 0; g < W; ++g)
                if (!per_wave[size_t(g)].empty()) src += "    case " + std::to_string(g) + ": {\n" + per_wave[size_t(g)] + "    } break;\n";
            src += "    default: break;\n  }\n  __syncthreads();\n";
        }
        const int out_len = int(plan.out_layout.row_len);
        {
            const int nch = 64 * out_len / epc, cpt = (nch + 511) / 512;
            const std::string OL = std::to_string(out_len);
            src += "  if (nitems == 64 && so == " + OL + " && (((unsigned long long)out) & 15ull) == 0 && " + ((64 * out_len) % epc == 0 && cpt <= 24 ? "true" : "false") + ") {\n";
            src += "    VT* dst = (VT*)(out + item0 * " + OL + ");\n";
            for (int k = 0; k < cpt; ++k) {
                src += "    if (tid + " + std::to_string(512 * k) + " < " + std::to_string(nch) + ") { VT v;\n";
                for (int j = 0; j < epc; ++j)
                    src += "      { const int e = (tid + " + std::to_string(512 * k) + ") * " + std::to_string(epc) + " + " + std::to_string(j) + ", i2 = e / " + OL + ", c = e - i2 * " + OL +
                           "; v[" + std::to_string(j) + "] = slab[i2 * " + std::to_string(stride) + " + " + std::to_string(out_base) + " + c]; }\n";
                src += "      __builtin_nontemporal_store(v, dst + tid + " + std::to_string(512 * k) + "); }\n";
            }
            src += "  } else {\n";
            src += "#pragma unroll 4\n    for (int e = tid; e < " + std::to_string(64 * out_len) + "; e += 512) { const int i2 = e / " + OL + ", c = e - i2 * " + OL +
                   "; if (i2 < nitems) out[(item0 + i2) * so + c] = slab[i2 * " + std::to_string(stride) + " + " + std::to_string(out_base) + " + c]; }\n";
            src += "  }\n";
        }
        src += "  if (!more) break;\n  g = gn;\n  __syncthreads();\n";   // the slabs are rewritten for the next group
        src += "  if (pre_next) commit(); else stage(gn * 64, nnext);\n";
        src += "  }\n";
        src += "}\n";
        f.jit_threads = 512;
        f.jit_items = 64;
        f.jit_persistent = int(std::max<size_t>(1, kLdsBytes / (size_t(64) * size_t(stride) * elem)));   // workgroups resident per CU (LDS)
        f.jit_source = std::move(src);
    }
    f.fused_slab = slab;
    f.fused_zero_slot = zero_slot;
    f.fused_out_base = out_base;
    f.n_entries = entries;
    f.name = "ast_fused[" + std::to_string(plan.steps.size()) + " arms, " + std::to_string(entries) +
             " comp-muls, slab " + std::to_string(slab) + "]";
    plan.steps.clear();
    plan.steps.push_back(std::move(f));
    plan.node_buffers.clear();  // the cache buffers live in LDS now
    return true;
}

}  // namespace

// Rows of one length with +-1 coefficients (dense products of non-degenerate algebras) that were not fused
// into a small-program launch: the same lists transposed to [term][row] with the sign in bit 31 --
// consecutive threads (rows) then read consecutive words and no coefficient array is streamed
// (k_product_ell).  Same order, same roundings: (l * r) * (-1.0) is -(l * r) exactly.
static void uniform_csr_to_ell(Plan& plan) {
    if (plan.flags & GAAST_FLAG_NO_FUSION) return;
    for (Step& s : plan.steps) {
        if (s.kind != Step::PRODUCT_CSR || s.u32_b.empty()) continue;
        const size_t n_rows = s.u32_b.size();
        const uint32_t width = s.u32_a[1] - s.u32_a[0];
        bool uniform = width >= 4;   // (round 3: from 4 terms per row on -- R X has n per row; 16 before: such lists ran on k_product_csr, an entry and a coefficient load per term)
        for (size_t i = 0; uniform && i < n_rows; ++i) uniform = s.u32_a[i + 1] - s.u32_a[i] == width;
        for (size_t e = 0; uniform && e < s.coeff.size(); ++e) uniform = s.coeff[e] == 1.0 || s.coeff[e] == -1.0;
        for (size_t e = 0; uniform && e < s.u32_c.size(); ++e) uniform = !(s.u32_c[e] & 0x80000000u);   // right offset < 2^15
        if (!uniform) continue;
        // offsets in BYTES when they fit 15 bits (rows of <= 32 KiB: n <= 12 in f64, n <= 13 in f32): the kernel
        // then adds them to an LDS base without scaling
        const uint32_t elem = plan.dtype == GAAST_F32 ? 4u : 8u;
        bool bytes = true;
        for (size_t e = 0; bytes && e < s.u32_c.size(); ++e)
            bytes = (s.u32_c[e] & 0xffffu) * elem < 32768u && (s.u32_c[e] >> 16) * elem < 32768u;
        std::vector<uint32_t> ell(size_t(width) * n_rows);
        for (size_t row = 0; row < n_rows; ++row)
            for (uint32_t t = 0; t < width; ++t) {
                const size_t e = size_t(s.u32_a[row]) + t;
                const uint32_t lo = s.u32_c[e] & 0xffffu, ro = s.u32_c[e] >> 16;
                const uint32_t word = bytes ? (lo * elem) | ((ro * elem) << 16) : s.u32_c[e];
                ell[size_t(t) * n_rows + row] = word | (s.coeff[e] < 0.0 ? 0x80000000u : 0u);
            }
        s.ell_bytes = bytes ? 1 : 0;
        s.u32_c.swap(ell);
        s.coeff.clear();
        s.ell_width = int(width);
        s.name = "product_ell" + s.name.substr(s.name.find('['));
    }
}

// A sparse product whose result is read ONLY as the left operand of a dense product (R X in R X ~R, eval.rs:61-86 with its
// cached operand) is evaluated inside the dense kernel, in LDS, while that kernel stages its operands: one launch instead of
// two, and the intermediate row never goes through HBM (BASELINE configs[4] at the dimensions where the whole program no
// longer fits a fused small-program kernel).  The list keeps the reference's order and roundings, so the dense kernel sees
// the very bits the separate launch would have written.  Off with GAAST_FLAG_DEBUG_NO_CHAIN (A/B, tests).
static void chain_sparse_into_dense(Plan& plan) {
    if (plan.flags & (GAAST_FLAG_NO_FUSION | GAAST_FLAG_DEBUG_NO_CHAIN)) return;
    plan.node_dead.assign(plan.node_buffers.size(), 0);
    const size_t elem = plan.dtype == GAAST_F32 ? 4 : 8;
    auto same = [](BufRef x, BufRef y) { return x.kind == y.kind && x.idx == y.idx; };
    for (size_t j = 0; j < plan.steps.size(); ++j) {
        Step& dn = plan.steps[j];
        if (dn.kind != Step::PRODUCT_DENSE || dn.use_spinor || dn.use_mfma6 || dn.chained || dn.a.kind != BufKind::NODE) continue;   // (k_gp_mfma6 has no chained staging: programs that small are fused whole)
        const BufRef buf = dn.a;
        // exactly one writer (a list product that starts the buffer: beta = 0), no other reader, nothing else touches it
        int writer = -1;
        bool ok = !same(dn.b, buf);
        for (size_t i = 0; i < plan.steps.size() && ok; ++i) {
            const Step& t = plan.steps[i];
            if (i == j) continue;
            if (t.kind == Step::FUSED) ok = false;
            if (same(t.res, buf)) {
                if (writer >= 0 || t.kind != Step::PRODUCT_CSR || t.beta != 0 || i > j) ok = false;
                writer = int(i);
            }
            if ((t.a.idx >= 0 && same(t.a, buf)) || (t.b.idx >= 0 && same(t.b, buf))) ok = false;
            if (t.chained && (same(t.pre_a, buf) || same(t.pre_b, buf))) ok = false;
        }
        if (!ok || writer < 0) continue;
        Step& w = plan.steps[size_t(writer)];
        // the list's operands must still hold at the dense launch: inputs always do; a cache buffer does unless a later step
        // writes it (cache buffers are written by the steps that fill them, all before their first reader)
        for (size_t i = size_t(writer) + 1; i < j && ok; ++i)
            if (same(plan.steps[i].res, w.a) || same(plan.steps[i].res, w.b)) ok = false;
        if (!ok) continue;
        auto row_len = [&](BufRef r) -> int64_t {
            return r.kind == BufKind::NODE ? plan.node_buffers[size_t(r.idx)].row_len : r.kind == BufKind::INPUT ? plan.input_layouts[size_t(r.idx)].row_len
                                                                                                               : plan.out_layout.row_len;
        };
        const int64_t ll = row_len(w.a), rl = row_len(w.b);
        const int n2 = dn.dense_n;
        // items a workgroup of the dense kernel stages at once (runtime.hip: prepare_step)
        int ipb = 1;
        if (dn.use_mfma) ipb = n2 <= 10 ? 4 : n2 == 11 ? 2 : 1;
        else if (!dn.use_mfma16 && !dn.use_mfma7) ipb = std::max(256, 1 << (n2 - 4)) >> (n2 - 4);
        const size_t scratch = size_t(ipb) * size_t(ll + rl + 1) * elem;
        const size_t images = size_t(ipb) * (size_t(dn.use_mfma && !dn.mfma32_pairs ? 2 : (dn.use_mfma || dn.use_mfma16 || dn.use_mfma7) ? 4 : 2) << n2) * elem + 256;
        if (scratch > 48 * 1024 || images + scratch > kLdsBytes - 1024 || w.u32_b.size() > 32768) continue;
        // rows of the list -> components of the dense step's left image
        std::vector<int32_t> map_of(size_t(row_len(buf)), -1);
        for (size_t c = 0; c < dn.u32_a.size(); ++c) map_of[dn.u32_a[c] & 0xffffu] = int32_t(c);
        dn.pre_row_map.clear();
        dn.pre_row_scale.clear();
        bool covered = true;
        std::vector<uint32_t> row_start(1, 0u), entries;
        std::vector<double> coeff;
        std::vector<char> produced(dn.u32_a.size(), 0);
        for (size_t row = 0; row < w.u32_b.size(); ++row) {
            const int32_t c = map_of[w.u32_b[row]];
            if (c < 0) continue;   // a component the dense product does not read (its grade is not wanted there)
            produced[size_t(c)] = 1;
            dn.pre_row_map.push_back(dn.u32_a[size_t(c)] & 0xffff0000u);
            if (dn.scaled) dn.pre_row_scale.push_back(dn.coeff[size_t(c)]);
            for (uint32_t e = w.u32_a[row]; e < w.u32_a[row + 1]; ++e) {
                entries.push_back(w.u32_c[e]);
                coeff.push_back(w.coeff[e]);
            }
            row_start.push_back(uint32_t(entries.size()));
        }
        for (char c : produced) covered = covered && c;
        // rows of one length with +-1 coefficients (R X: n entries per row): [term][row] words, sign in bit 31
        {
            const size_t rows = row_start.size() - 1;
            const uint32_t width = rows ? row_start[1] - row_start[0] : 0;
            bool uniform = rows > 0 && width > 0;
            for (size_t r = 0; uniform && r < rows; ++r) uniform = row_start[r + 1] - row_start[r] == width;
            for (size_t e = 0; uniform && e < coeff.size(); ++e) uniform = coeff[e] == 1.0 || coeff[e] == -1.0;
            for (size_t e = 0; uniform && e < entries.size(); ++e) uniform = (entries[e] & 0xffffu) < 0x8000u && (entries[e] >> 16) < 0x8000u;
            uniform = uniform && ll + rl + 1 < 0x8000;
            if (uniform) {
                // padded to a multiple of 4 terms with entries over the zero pair the kernel keeps behind the two rows
                // (left offset ll + rl from the left row, right offset rl from the right row): acc + (+0.0) changes no acc
                const uint32_t wpad = (width + 3u) & ~3u;
                const uint32_t zero_entry = uint32_t(ll + rl) | (uint32_t(rl) << 16);
                std::vector<uint32_t> ell(size_t(wpad) * rows, zero_entry);
                for (size_t r = 0; r < rows; ++r)
                    for (uint32_t t = 0; t < width; ++t) {
                        const size_t e = size_t(row_start[r]) + t;
                        ell[size_t(t) * rows + r] = entries[e] | (coeff[e] < 0.0 ? 0x80000000u : 0u);
                    }
                entries.swap(ell);
                coeff.clear();
                dn.pre_width = int(wpad);
            }
        }
        dn.chained = 1;
        dn.pre_a = w.a;
        dn.pre_b = w.b;
        dn.pre_canon_a = w.canon_a;
        dn.pre_canon_b = w.canon_b;
        dn.pre_row_start = std::move(row_start);
        dn.pre_entries = std::move(entries);
        dn.pre_coeff = std::move(coeff);
        dn.pre_left_len = int(ll);
        dn.pre_right_len = int(rl);
        // components no row produces stay zero: the kernel zero-fills the image first unless every loaded component is covered
        dn.left_full = dn.left_full && covered;
        dn.left_contig = 0;   // general staging
        dn.name += " <- " + w.name + " in LDS";
        dn.n_entries += w.n_entries;
        dn.a = BufRef{BufKind::NODE, -1};
        plan.node_dead[size_t(buf.idx)] = 1;
        w.kind = Step::ZERO;   // marks the list step for removal below
        w.res = BufRef{BufKind::NODE, -1};
    }
    std::vector<Step> kept;
    for (Step& t : plan.steps)
        if (!(t.kind == Step::ZERO && t.res.kind == BufKind::NODE && t.res.idx < 0)) kept.push_back(std::move(t));
    plan.steps = std::move(kept);
}

// ---------------------------------------------------------------------------------------------------------------------
// The list chain specialised per program (round 4; the generic k_product_ell_chain is the fallback when hiprtc is not
// available).  What the generic kernel paid for: run-time widths and lengths, a row-per-lane first list whose 64 lanes gather
// 64 different offsets of ONE item (bank conflicts: 39 % of the LDS cycles), nine vector instructions per term.  Here:
//   * lane = (row, item) in BOTH lists, the IPB items of the workgroup fastest: the lanes of one LDS access group read the
//     SAME offset of different items, item stride odd (in elements) -> no bank conflict at IPB = 32;
//   * an entry is one word of byte offsets from the ITEM's base: list 1 [15:0] left | [31:16] right, the sign folded into
//     the choice between the smaller operand's image and its NEGATED image (l * (-r) = -(l * r) exactly: eval.rs:82 with
//     coeff = -1); list 2 [15:0] mid | [30:16] other | [31] sign, applied as fma(l * r, +-1.0, acc): the product is rounded
//     first, +-1 is exact, the sum is rounded once -- the reference's three roundings;
//   * entries are read from global memory (the tables are shared by every workgroup and stay in L2 / L1), a quad of entries
//     per 16-byte load, the next quads in flight while the current ones are used; no run-time width, no padding terms
//     (acc + (+0.0) would turn a -0.0 accumulator of a beta = 1 list into +0.0);
//   * reference order and roundings: bit-identical to the two-launch plan and to the oracle.
// ---------------------------------------------------------------------------------------------------------------------
// wp == nullptr: a SINGLE list with few long rows (rows2 x IPB lanes instead of rows2; k_product_ell gives a row to a thread): the
// left operand plays the staged "mid" row (l1 = r1 = 0, mid = its length), the right one is list 2's own operand r2 (alias 0,
// side 1).  init_off (single list only): acc starts from 0.0 + init[row's offset] -- a covering copy_grades_from of an input
// folded into the list that accumulates onto it ((a + b * c).g(2): one launch).
static void make_chain_jit(const Plan& plan, Step& c, const Step* wp, int64_t l1, int64_t r1, int64_t mid, int64_t r2, int alias, int side,
                           bool covered, const std::vector<uint32_t>* init_off = nullptr) {
    if (plan.flags & (GAAST_FLAG_NO_JIT | GAAST_FLAG_DEBUG_JIT_FAILS)) return;
    const bool single = wp == nullptr;
    static const Step no_list;
    const Step& w = single ? no_list : *wp;
    const int64_t esz = plan.dtype == GAAST_F32 ? 4 : 8;
    const int64_t neg_len = single ? 0 : std::min(l1, r1);
    const bool neg_is_left = l1 < r1;
    // layout of an item in LDS (elements): list 2's own operand first (its offsets have 15 bits), then the rest
    int64_t off_l1 = 0, off_r1 = 0, off_neg, off_mid, off_r2 = -1, cur = 0;
    auto place = [&](int64_t len) { const int64_t o = cur; cur += len; return o; };
    if (alias == 0) off_r2 = place(r2);
    if (alias == 2) { off_r1 = place(r1); off_l1 = place(l1); }
    else { off_l1 = place(l1); off_r1 = place(r1); }
    off_neg = place(neg_len);
    off_mid = place(mid);
    // tolerance mode: list 2 sign-sorted (below) -- when the item still fits the 16-bit byte offsets with one more element, and the
    // rows' signs are balanced enough: every (row, slice) pair is padded to the longest plus and the longest minus segment, and a
    // padding term costs what a real one does (Euclidean sandwiches at n = 8, 10 have all-plus rows: +50 % terms -- they keep the
    // sign words; R^{6,3} at n = 9: 64 + 64 of 128, no padding; (a + b c).g(2) at n = 8: 136 + 136 for 256)
    bool sorted = !(plan.flags & GAAST_FLAG_EXACT_ORDER) && (cur + 1) * esz <= 65536;
    if (sorted) {
        const int64_t rows2e = int64_t(c.u32_b.size()), w2e = c.ell_width;
        int64_t sp = 1;   // (the slice count of the tolerance mode, decided again below with the same rule)
        while (sp < 4 && w2e % (2 * sp) == 0 && w2e / (2 * sp) >= 32 && rows2e * 32 * 2 * sp <= 1024 && rows2e * 2 * sp <= mid) sp *= 2;
        const int64_t gran = plan.dtype == GAAST_F32 ? 16 : 8;
        int64_t wp = 0, wm = 0;
        for (int64_t row = 0; row < rows2e; ++row) {
            int64_t np = 0, nm = 0;
            for (int64_t t = 0; t < w2e; ++t) ((c.u32_c[size_t(t * rows2e + row)] & 0x80000000u) ? nm : np) += 1;
            wp = std::max(wp, (np + sp - 1) / sp);
            wm = std::max(wm, (nm + sp - 1) / sp);
        }
        wp = (wp + gran - 1) / gran * gran;
        wm = (wm + gran - 1) / gran * gran;
        sorted = rows2e > 0 && w2e > 0 && (wp + wm) * sp * 100 <= w2e * 107;
    }
    const bool has_zero = !(plan.flags & GAAST_FLAG_EXACT_ORDER) && (cur + 1) * esz <= 65536;
    const int64_t off_zero = has_zero ? place(1) : 0;             // ... padding terms (sorted list 2, list 1 by right index) multiply this element
    int64_t stride = cur | 1;   // odd: lanes reading one offset of consecutive items touch consecutive banks (bank pairs in f64)
    const int64_t off_other = alias == 0 ? off_r2 : alias == 1 ? off_l1 : off_r1;
    const int64_t other_len = alias == 0 ? r2 : alias == 1 ? l1 : r1;
    if ((off_other + other_len) * esz > 32768 || cur * esz > 65536) return;   // (15-bit / 16-bit byte offsets from the item's base)
    const int64_t rows1 = int64_t(w.u32_b.size()), rows2 = int64_t(c.u32_b.size());
    const int64_t w1 = w.ell_width, w2 = c.ell_width;
    if (rows2 <= 0 || w2 <= 0) return;
    if (!single && (rows1 <= 0 || w1 <= 0 || w1 > 32)) return;   // (list 1's operands of a row are all in flight at once)
    // items per workgroup: a power of two up to 32 (the lanes of an LDS access group), as many as the LDS holds
    int64_t ipb = 32;
    const int64_t lds_cap = int64_t(kLdsBytes);
    while (ipb > 1 && ipb * stride * esz > lds_cap - 4096) ipb >>= 1;
    if (ipb < 2) return;
    const int64_t w1p = (w1 + 3) & ~int64_t(3), w2p = (w2 + 3) & ~int64_t(3);
    // The tables ride in LDS when they fit beside the items (broadcast reads; from global memory every lane would get its own
    // copy of the words through the vector cache: 16 cycles of its return path per 16-byte load and wave).
    //   list 2: WIDE entries (8 bytes: [15:0] mid | [31:16] other, then the sign bit alone) cost two vector instructions less per
    //           term than narrow ones ([15:0] mid | [30:16] other | [31] sign) -- taken when they fit;
    //   list 1: rows padded to quads (16-byte reads) when that fits, else unpadded rows (4-byte reads), else global memory.
    const int64_t items_bytes = (ipb * stride * esz + 15) / 16 * 16;
    int64_t used = items_bytes;
    // TOLERANCE MODE, SIGN-SORTED (round 4): the order inside a row is free, so the terms of every (row, slice) pair are stored
    // plus-terms first (padded to WPS), then minus-terms (padded to WMS): the sign is the POSITION -- no sign word, no sign
    // instruction: a term is two SDWA address additions and one fused multiply-add (exact order: 5, narrow entries: 7) -- and the
    // entries are clean 16 + 16-bit words (half the wide table: at n = 9 list 1's table then fits in LDS too).  Padding terms
    // multiply the item's zero element by itself.  The slice count is decided first (it shapes the table).
    int64_t split = 1;
    if (!(plan.flags & GAAST_FLAG_EXACT_ORDER))
        while (split < 4 && w2 % (2 * split) == 0 && w2 / (2 * split) >= 32 && rows2 * ipb * 2 * split <= 1024 && rows2 * 2 * split <= mid) split *= 2;
    int64_t wps = 0, wms = 0;   // plus / minus terms per (row, slice), multiples of 8 (f32: 16): whole register batches
    std::vector<std::vector<uint32_t>> plus_terms, minus_terms;   // per row: mid offset | other offset << 16 (bytes)
    if (sorted) {
        plus_terms.resize(size_t(rows2));
        minus_terms.resize(size_t(rows2));
        const int64_t off_other_b = (alias == 0 ? off_r2 : alias == 1 ? off_l1 : off_r1) * esz;
        for (int64_t row = 0; row < rows2; ++row)
            for (int64_t t = 0; t < w2; ++t) {
                const uint32_t e = c.u32_c[size_t(t * rows2 + row)];
                const int64_t lo = e & 0x7fffu, ro = (e >> 16) & 0x7fffu;
                const int64_t ma = off_mid * esz + (side == 1 ? lo : ro), oa = off_other_b + (side == 1 ? ro : lo);
                ((e & 0x80000000u) ? minus_terms : plus_terms)[size_t(row)].push_back(uint32_t(ma) | (uint32_t(oa) << 16));
            }
        const int64_t gran = plan.dtype == GAAST_F32 ? 16 : 8;
        for (int64_t row = 0; row < rows2; ++row) {
            wps = std::max<int64_t>(wps, (int64_t(plus_terms[size_t(row)].size()) + split - 1) / split);
            wms = std::max<int64_t>(wms, (int64_t(minus_terms[size_t(row)].size()) + split - 1) / split);
        }
        wps = (wps + gran - 1) / gran * gran;
        wms = (wms + gran - 1) / gran * gran;
    }
    const int64_t wss = wps + wms;
    int ent2_mode;   // 0: narrow, global; 1: narrow, LDS; 2: wide, LDS; sorted: 3: LDS, 4: global
    if (sorted) {
        if (used + rows2 * split * wss * 4 <= lds_cap) ent2_mode = 3, used += rows2 * split * wss * 4;
        else ent2_mode = 4;
    } else if (used + rows2 * w2p * 8 <= lds_cap) ent2_mode = 2, used += rows2 * w2p * 8;
    else if (used + rows2 * w2p * 4 <= lds_cap) ent2_mode = 1, used += rows2 * w2p * 4;
    else ent2_mode = 0;
    const int64_t ent2_at = items_bytes;
    int ent1_mode;   // 0: padded rows, global; 1: unpadded rows, LDS; 2: padded rows, LDS
    const int64_t ent1_at = used;
    if (used + rows1 * w1p * 4 <= lds_cap) ent1_mode = 2, used += rows1 * w1p * 4;
    else if (used + rows1 * w1 * 4 <= lds_cap) ent1_mode = 1, used += rows1 * w1 * 4;
    else ent1_mode = 0;
    used = (used + 15) / 16 * 16;
    // TOLERANCE MODE (without GAAST_FLAG_EXACT_ORDER, like the dense products): a long row of list 2 is cut into `split` slices of
    // consecutive terms, each summed by a lane of its own in the reference's order, the partial sums added in slice order at the end --
    // the few long chains that leave most of the workgroup idle become split x as many, half as long.  |error| <= 4 eps sum |terms|
    // per component (the dense path's contract; one extra rounding per slice).  GAAST_FLAG_EXACT_ORDER: split = 1, bit for bit.
    // one workgroup per CU (the usual case from n = 9 on): 512 threads -- list 1 and the staging have work for all of them, list 2
    // (few long rows) for rows2 * split * ipb lanes; two or more workgroups per CU: 256 threads each
    int64_t threads = 2 * used <= lds_cap ? 256 : 512;
    threads = std::max<int64_t>(threads, std::min<int64_t>(1024, (rows2 * split * ipb + 63) / 64 * 64));
    // whole multiples of 256: the waves of a workgroup are dealt round-robin to the CU's four SIMDs, and every phase ends at a
    // barrier -- 9 or 10 waves leave one SIMD with three where the others have two (measured: sand9g1 576 -> 768 threads +6 %,
    // sand10g1 640 -> 768 +2 %; 1,024 where list 2 does not ask for them: -3 % / -23 %)
    threads = std::min<int64_t>(1024, (threads + 255) / 256 * 256);
    const int64_t w1s = ent1_mode == 1 ? w1 : w1p;   // words per row of list 1's table
    // tables
    // TOLERANCE MODE, list 1 with its small right operand IN REGISTERS (the sandwich's X: n components, the same for every row a lane
    // evaluates): the terms of a row are re-ordered by right index -- term j multiplies by x_j, a register --, so a term is ONE LDS
    // read (the left operand), its sign a bit of the table word; rows without a term for some j multiply the item's zero element.
    bool xreg = has_zero && !single && !neg_is_left && r1 <= 12 && w1 <= r1 && ent1_mode != 1 && w1s >= ((r1 + 3) & ~int64_t(3));
    for (int64_t row = 0; row < rows1 && xreg; ++row) {
        uint32_t seen = 0;
        for (int64_t t = 0; t < w1 && xreg; ++t) {
            const int64_t j = int64_t((w.u32_c[size_t(t * rows1 + row)] >> 16) & 0x7fffu) / esz;
            xreg = j < r1 && !(seen & (1u << j));
            seen |= 1u << j;
        }
    }
    c.cj_ent1.assign(size_t(rows1 * w1s), 0u);
    c.cj_pos1.resize(size_t(rows1));
    if (init_off) c.cj_pos1 = *init_off;   // (a single list has no row positions: the slot carries the offsets of the folded copy)
    for (int64_t row = 0; row < rows1; ++row) {
        c.cj_pos1[size_t(row)] = uint32_t((off_mid + int64_t(w.u32_b[size_t(row)])) * esz);
        for (int64_t t = 0; t < w1; ++t) {
            const uint32_t e = w.u32_c[size_t(t * rows1 + row)];
            const bool neg = (e & 0x80000000u) != 0;
            const int64_t lo = e & 0x7fffu, ro = (e >> 16) & 0x7fffu;   // bytes, from the operand rows
            const int64_t la = ((neg && neg_is_left) ? off_neg : off_l1) * esz + lo;
            const int64_t ra = ((neg && !neg_is_left) ? off_neg : off_r1) * esz + ro;
            if (!xreg) c.cj_ent1[size_t(row * w1s + t)] = uint32_t(la) | (uint32_t(ra) << 16);
        }
        if (xreg) {
            for (int64_t j = 0; j < w1s; ++j) c.cj_ent1[size_t(row * w1s + j)] = uint32_t(off_zero * esz);   // no term for x_j: 0 * x_j
            for (int64_t t = 0; t < w1; ++t) {
                const uint32_t e = w.u32_c[size_t(t * rows1 + row)];
                const int64_t lo = e & 0x7fffu, j = int64_t((e >> 16) & 0x7fffu) / esz;
                c.cj_ent1[size_t(row * w1s + j)] = uint32_t(off_l1 * esz + lo) | (e & 0x80000000u);
            }
        }
    }
    const int64_t wpt2 = ent2_mode == 2 ? 2 : 1;   // words per term
    c.cj_out2 = c.u32_b;
    if (sorted) {
        const uint32_t pad = uint32_t(off_zero * esz) | (uint32_t(off_zero * esz) << 16);
        c.cj_ent2.assign(size_t(rows2 * split * wss), pad);
        for (int64_t row = 0; row < rows2; ++row)
            for (int which = 0; which < 2; ++which) {
                const std::vector<uint32_t>& tv = which ? minus_terms[size_t(row)] : plus_terms[size_t(row)];
                const int64_t per = (int64_t(tv.size()) + split - 1) / split;   // consecutive chunks, one per slice
                for (int64_t i = 0; i < int64_t(tv.size()); ++i) {
                    const int64_t sl = per ? i / per : 0, k = per ? i % per : 0;
                    c.cj_ent2[size_t((row * split + sl) * wss + (which ? wps : 0) + k)] = tv[size_t(i)];
                }
            }
    } else
        c.cj_ent2.assign(size_t(rows2 * w2p * wpt2), 0u);
    for (int64_t row = 0; row < rows2 && !sorted; ++row)
        for (int64_t t = 0; t < w2; ++t) {
            const uint32_t e = c.u32_c[size_t(t * rows2 + row)];
            const int64_t lo = e & 0x7fffu, ro = (e >> 16) & 0x7fffu;
            const int64_t ma = off_mid * esz + (side == 1 ? lo : ro), oa = off_other * esz + (side == 1 ? ro : lo);
            if (ent2_mode == 2) {
                c.cj_ent2[size_t((row * w2p + t) * 2)] = uint32_t(ma) | (uint32_t(oa) << 16);
                c.cj_ent2[size_t((row * w2p + t) * 2 + 1)] = e & 0x80000000u;
            } else {
                c.cj_ent2[size_t(row * w2p + t)] = uint32_t(ma) | (uint32_t(oa) << 16) | (e & 0x80000000u);
            }
        }
    // source
    std::string src;
    auto def = [&](const char* name, int64_t v) { src += std::string("#define ") + name + " " + std::to_string(v) + "\n"; };
    src += plan.dtype == GAAST_F32 ? "typedef float T;\n#define F64 0\n" : "typedef double T;\n#define F64 1\n";
    def("ESZ", esz); def("NT", threads); def("IPB", ipb); def("NSUB", threads / ipb); def("STRIDE_B", stride * esz);
    def("L1", l1); def("R1", r1); def("R2", alias ? 0 : r2); def("MID", mid); def("NEGLEN", neg_len); def("NEG_IS_LEFT", neg_is_left ? 1 : 0);
    def("OFF_L1", off_l1 * esz); def("OFF_R1", off_r1 * esz); def("OFF_NEG", off_neg * esz); def("OFF_MID", off_mid * esz);
    def("OFF_R2", (alias ? 0 : off_r2) * esz); def("HAS_R2", alias ? 0 : 1);
    def("ROWS1", rows1); def("W1", w1); def("W1S", w1s); def("ROWS2", rows2); def("W2", w2); def("W2P", w2p);
    def("CANON_L1", w.canon_a); def("CANON_R1", w.canon_b); def("CANON_R2", side == 1 ? c.canon_b : c.canon_a);
    def("CANON_MID", side == 1 ? c.canon_a : c.canon_b); def("COVERED", covered ? 1 : 0); def("BETA", init_off ? 0 : c.beta);
    def("SINGLE", single ? 1 : 0); def("INIT_SRC", init_off ? 1 : 0);
    bool pos1_linear = true;   // row k of list 1 is component k of the mid row (the usual case): no table, no load
    for (int64_t row = 0; row < rows1; ++row) pos1_linear = pos1_linear && int64_t(w.u32_b[size_t(row)]) == row;
    def("POS1_LINEAR", pos1_linear ? 1 : 0);
    // EXACT: every product rounded, then added (eval.rs:82), rows summed whole -- the reference's bits.  Otherwise (the default,
    // the dense products' tolerance contract): rows of list 2 in slices, and l * r + acc as one fused multiply-add
    def("EXACT", (plan.flags & GAAST_FLAG_EXACT_ORDER) ? 1 : 0);
    def("XREG", xreg ? 1 : 0);
    def("SORTED", sorted ? 1 : 0); def("WPS", wps); def("WMS", wms); def("WSS", wss); def("OFF_ZERO", off_zero * esz);
    def("SPLIT", split); def("WS", w2 / split);                    // slices per row of list 2, terms per slice
    def("PASSES2", (rows2 * split + threads / ipb - 1) / (threads / ipb));   // (row, slice) pairs of list 2 per thread
    // terms of list 2 in flight per register set: the registers of two waves per SIMD (512 threads) hold 8 f64 / 16 f32 terms twice; more
    // threads (a single list with many rows x items), fewer registers each
    def("TB", (plan.dtype == GAAST_F32 ? 16 : 8) / (threads > 512 ? 2 : 1));
    def("ENT2_MODE", ent2_mode); def("ENT1_MODE", ent1_mode); def("ENT2_AT", ent2_at); def("ENT1_AT", ent1_at); def("SMEM_BYTES", used);
    src += R"JIT(
typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef T VT __attribute__((ext_vector_type(16 / ESZ)));
#define EPC (16 / ESZ)
#if F64
#define ONE_HI 0x3ff00000u
__device__ __forceinline__ T with_hi(T v, u32 hi) { return __builtin_bit_cast(double, (u64(hi) << 32) | (__builtin_bit_cast(u64, v) & 0xffffffffull)); }
__device__ __forceinline__ u32 hi_of(T v) { return u32(__builtin_bit_cast(u64, v) >> 32); }
__device__ __forceinline__ T pm_one(u32 hi) { return __builtin_bit_cast(double, u64(hi) << 32); }
#define FMA(a, b, c) __builtin_fma(a, b, c)
#else
#define ONE_HI 0x3f800000u
__device__ __forceinline__ T with_hi(T v, u32 hi) { return __builtin_bit_cast(float, hi); }
__device__ __forceinline__ u32 hi_of(T v) { return __builtin_bit_cast(u32, v); }
__device__ __forceinline__ T pm_one(u32 hi) { return __builtin_bit_cast(float, hi); }
#define FMA(a, b, c) __builtin_fmaf(a, b, c)
#endif
#if EXACT
#define MAC(a, b, c) ((c) + (a) * (b))
#else
#define MAC(a, b, c) FMA(a, b, c)
#endif
typedef __attribute__((address_space(3))) unsigned char lds_u8;
#define LDS(addr) (*(const __attribute__((address_space(3))) T*)(smem + (addr)))

// operand rows of the group's items, element e = item * LEN + c of the flattened range; rows beyond the batch are zero
template <int LEN, int CANON, int OFF, int NEGOFF>
__device__ __forceinline__ void put(lds_u8* smem, int e, T v) {
    const int i2 = e / LEN, c = e - i2 * LEN;
    if (CANON) v = T(0) + v;                                      // init_null_mv + add_grades_from: 0.0 + x (eval.rs:27-31)
    *(__attribute__((address_space(3))) T*)(smem + i2 * STRIDE_B + OFF + c * ESZ) = v;
    if (NEGOFF >= 0) *(__attribute__((address_space(3))) T*)(smem + i2 * STRIDE_B + NEGOFF + c * ESZ) = -v;
}
template <int LEN, int CANON, int OFF, int NEGOFF>
__device__ __forceinline__ void stage(lds_u8* smem, const T* __restrict__ src, long long stride, long long item0, int nitems, int tid) {
    constexpr int TOTAL = IPB * LEN;
#pragma unroll 4
    for (int e = tid; e < TOTAL; e += NT) {
        const int i2 = e / LEN, c = e - i2 * LEN;
        put<LEN, CANON, OFF, NEGOFF>(smem, e, i2 < nitems ? src[(item0 + i2) * stride + c] : T(0));
    }
}
// ... the same rows as 16-byte pieces through registers: issued for the NEXT group while the current one is evaluated (contiguous,
// 16-byte aligned rows of a full group; anything else takes stage())
template <int LEN>
struct Pre {
    static constexpr int NCH = IPB * LEN / EPC, CPT = (NCH + NT - 1) / NT;
    VT v[CPT];
    __device__ __forceinline__ void issue(const T* __restrict__ src, long long item0, int tid) {
        const VT* s16 = (const VT*)(src + item0 * LEN);
#pragma unroll
        for (int k = 0; k < CPT; ++k)
            if (tid + k * NT < NCH) v[k] = __builtin_nontemporal_load(s16 + tid + k * NT);
    }
    template <int CANON, int OFF, int NEGOFF>
    __device__ __forceinline__ void commit(lds_u8* smem, int tid) {
#pragma unroll
        for (int k = 0; k < CPT; ++k)
            if (tid + k * NT < NCH) {
#pragma unroll
                for (int j = 0; j < EPC; ++j) put<LEN, CANON, OFF, NEGOFF>(smem, (tid + k * NT) * EPC + j, v[k][j]);
            }
    }
};

extern "C" __global__ __launch_bounds__(NT) void gaast_chain(const T* __restrict__ l1, long long s_l1, const T* __restrict__ r1, long long s_r1,
                                                              const T* __restrict__ r2, long long s_r2, T* __restrict__ out, long long s_out,
                                                              const u32* __restrict__ ent1, const u32* __restrict__ pos1,
                                                              const u32* __restrict__ ent2, const u32* __restrict__ out2, long long batch,
                                                              const T* __restrict__ init, long long s_init) {
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SMEM_BYTES];
    lds_u8* smem = (lds_u8*)smem_raw;
    const int tid = threadIdx.x;
    const int it = tid & (IPB - 1), sub = tid / IPB;
    const u32 base = u32(it) * STRIDE_B;
    u32 one_hi = ONE_HI;
    asm volatile("" : "+v"(one_hi));   // a vector register (the and-or takes one scalar operand)
    // the tables, once per (persistent) workgroup
#if SORTED || XREG
    for (int e = tid; e < IPB; e += NT) *(__attribute__((address_space(3))) T*)(smem + e * STRIDE_B + OFF_ZERO) = T(0);   // the padding terms' operand
#endif
#if ENT2_MODE == 3
    for (int e = tid; e < ROWS2 * SPLIT * WSS / 4; e += NT) ((__attribute__((address_space(3))) u32x4*)(smem + ENT2_AT))[e] = ((const u32x4*)ent2)[e];
#elif ENT2_MODE == 2
    for (int e = tid; e < ROWS2 * W2P / 2; e += NT) ((__attribute__((address_space(3))) u32x4*)(smem + ENT2_AT))[e] = ((const u32x4*)ent2)[e];
#elif ENT2_MODE == 1
    for (int e = tid; e < ROWS2 * W2P / 4; e += NT) ((__attribute__((address_space(3))) u32x4*)(smem + ENT2_AT))[e] = ((const u32x4*)ent2)[e];
#endif
#if ENT1_MODE >= 1 && !SINGLE
    for (int e = tid; e < ROWS1 * W1S; e += NT) ((__attribute__((address_space(3))) u32*)(smem + ENT1_AT))[e] = ent1[e];
#endif
    // entry words: list 1, term t of `row`; list 2, the quad holding terms 4q .. 4q + 3 (narrow) / 2q, 2q + 1 (wide)
#if !SINGLE
    auto ent1_quad = [&](int row, int q) -> u32x4 {
#if ENT1_MODE == 2
        return ((const __attribute__((address_space(3))) u32x4*)(smem + ENT1_AT))[row * (W1S / 4) + q];
#elif ENT1_MODE == 1
        const __attribute__((address_space(3))) u32* pw = (const __attribute__((address_space(3))) u32*)(smem + ENT1_AT) + row * W1S + 4 * q;
        u32x4 r;
        r[0] = pw[0];
        r[1] = 4 * q + 1 < W1 ? pw[1] : 0u;
        r[2] = 4 * q + 2 < W1 ? pw[2] : 0u;
        r[3] = 4 * q + 3 < W1 ? pw[3] : 0u;
        return r;
#else
        return ((const u32x4*)ent1)[row * (W1S / 4) + q];
#endif
    };
#endif
    auto ent2_quad = [&](int row, int q) -> u32x4 {
#if ENT2_MODE == 2
        return ((const __attribute__((address_space(3))) u32x4*)(smem + ENT2_AT))[row * (W2P / 2) + q];
#elif ENT2_MODE == 1
        return ((const __attribute__((address_space(3))) u32x4*)(smem + ENT2_AT))[row * (W2P / 4) + q];
#else
        return ((const u32x4*)ent2)[row * (W2P / 4) + q];
#endif
    };
    u32 oo_reg[PASSES2];   // this thread's rows of list 2: their output offsets, loaded once
#pragma unroll
    for (int k = 0; k < PASSES2; ++k) oo_reg[k] = sub + k * NSUB < ROWS2 * SPLIT ? out2[(sub + k * NSUB) / SPLIT] : 0u;
#if INIT_SRC
    u32 io_reg[PASSES2];   // ... and where their accumulators start from: 0.0 + init[offset] (a covering copy_grades_from folded in)
#pragma unroll
    for (int k = 0; k < PASSES2; ++k) io_reg[k] = sub + k * NSUB < ROWS2 * SPLIT ? pos1[(sub + k * NSUB) / SPLIT] : 0u;
#endif
    const long long groups = (batch + IPB - 1) / IPB;
    // register-prefetch staging needs contiguous, 16-byte aligned rows (wave-uniform test); shared rows (stride 0), wrapped
    // strided memory and the last, partial group take the plain loop
    // (the memory counter is IN ORDER: a wait for any later global load would wait for the prefetched rows first, so the evaluation
    //  issues none where it can: row positions and output offsets in registers, tables in LDS.  A table that has to stay in global
    //  memory (n = 10: list 1's) makes the first row of list 1 wait for the prefetch -- still better than no prefetch: measured
    //  0.214 against 0.164 G items/s at n = 10)
#if SINGLE
    // a single list: the left operand is the staged "mid" row (pointer l1), the right one list 2's own operand (pointer r2)
    const bool fast = s_l1 == MID && s_r2 == R2 && ((u64)l1 & 15ull) == 0 && ((u64)r2 & 15ull) == 0 && (IPB * MID) % EPC == 0 && (IPB * R2) % EPC == 0;
    Pre<MID> pl;
    Pre<R2> pq;
    auto issue = [&](long long item0) {
        pl.issue(l1, item0, tid);
        pq.issue(r2, item0, tid);
    };
    auto commit = [&]() {
        pl.template commit<CANON_MID, OFF_MID, -1>(smem, tid);
        pq.template commit<CANON_R2, OFF_R2, -1>(smem, tid);
    };
    auto stage_all = [&](long long item0, int nitems) {
        stage<MID, CANON_MID, OFF_MID, -1>(smem, l1, s_l1, item0, nitems, tid);
        stage<R2, CANON_R2, OFF_R2, -1>(smem, r2, s_r2, item0, nitems, tid);
    };
#else
    const bool fast = s_l1 == L1 && s_r1 == R1 && ((u64)l1 & 15ull) == 0 && ((u64)r1 & 15ull) == 0 && (IPB * L1) % EPC == 0 && (IPB * R1) % EPC == 0
#if HAS_R2
                      && s_r2 == R2 && ((u64)r2 & 15ull) == 0 && (IPB * R2) % EPC == 0
#endif
        ;
    Pre<L1> pl;
    Pre<R1> pr;
#if HAS_R2
    Pre<R2> pq;
#endif
    auto issue = [&](long long item0) {
        pl.issue(l1, item0, tid);
        pr.issue(r1, item0, tid);
#if HAS_R2
        pq.issue(r2, item0, tid);
#endif
    };
    auto commit = [&]() {
        pl.template commit<CANON_L1, OFF_L1, (NEG_IS_LEFT ? OFF_NEG : -1)>(smem, tid);
        pr.template commit<CANON_R1, OFF_R1, (NEG_IS_LEFT ? -1 : OFF_NEG)>(smem, tid);
#if HAS_R2
        pq.template commit<CANON_R2, OFF_R2, -1>(smem, tid);
#endif
    };
    auto stage_all = [&](long long item0, int nitems) {
        stage<L1, CANON_L1, OFF_L1, (NEG_IS_LEFT ? OFF_NEG : -1)>(smem, l1, s_l1, item0, nitems, tid);
        stage<R1, CANON_R1, OFF_R1, (NEG_IS_LEFT ? -1 : OFF_NEG)>(smem, r1, s_r1, item0, nitems, tid);
#if HAS_R2
        stage<R2, CANON_R2, OFF_R2, -1>(smem, r2, s_r2, item0, nitems, tid);
#endif
    };
#endif
    long long g = blockIdx.x;
    if (g >= groups) return;
    {
        const long long item0 = g * IPB;
        const int nitems = int(batch - item0 < IPB ? batch - item0 : IPB);
        if (fast && nitems == IPB) {
            issue(item0);
            commit();
        } else {
            stage_all(item0, nitems);
        }
    }
    for (;;) {   // persistent workgroups
        const long long item0 = g * IPB;
        const int nitems = int(batch - item0 < IPB ? batch - item0 : IPB);
        const long long gn = g + gridDim.x;
        const bool more = gn < groups;
        const int nnext = more ? int(batch - gn * IPB < IPB ? batch - gn * IPB : IPB) : 0;
        const bool pre_next = fast && more && nnext == IPB;
#if !COVERED && !SINGLE
        for (int e = tid; e < IPB * MID; e += NT) *(__attribute__((address_space(3))) T*)(smem + (e / MID) * STRIDE_B + OFF_MID + (e % MID) * ESZ) = T(0);
#endif
        __syncthreads();
#if BETA
        T acc0[PASSES2];   // list 2 adds into what another arm left in `out`: read BEFORE the prefetch is issued (in-order counter)
#pragma unroll
        for (int k = 0; k < PASSES2; ++k) acc0[k] = (sub + k * NSUB < ROWS2 * SPLIT && it < nitems) ? out[(item0 + it) * s_out + oo_reg[k]] : T(0);
#elif INIT_SRC
        T acc0[PASSES2];   // ... or onto a copy of an input's grades, made here: 0.0 + x (graded.rs:195-201 then :74)
#pragma unroll
        for (int k = 0; k < PASSES2; ++k) acc0[k] = T(0) + ((sub + k * NSUB < ROWS2 * SPLIT && it < nitems) ? init[(item0 + it) * s_init + io_reg[k]] : T(0));
#endif
        if (pre_next) issue(gn * IPB);   // in flight while this group is evaluated
        // ---- list 1 -> mid (eval.rs:77-83 into the fresh cache buffer of eval.rs:21-33): the operands of row k + 1 are in
        // flight while row k's chain is evaluated (two register sets, the loop unrolled by two: no copies) ----
#if !SINGLE
#if XREG
        {   // list 1 with the right operand (R1 components of THIS lane's item) in registers: term j of every row multiplies by xr[j]
            T xr[R1];
#pragma unroll
            for (int j = 0; j < R1; ++j) xr[j] = LDS(base + OFF_R1 + j * ESZ);
            T la[R1], lb[R1];
            u32 ea[R1], eb[R1];
            auto load1 = [&](T (&lv)[R1], u32 (&ev)[R1], int row) {
#pragma unroll
                for (int q = 0; q < (R1 + 3) / 4; ++q) {
                    const u32x4 e4 = ent1_quad(row, q);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (4 * q + j < R1) {
                            lv[4 * q + j] = LDS(base + (e4[j] & 0xffffu));
                            ev[4 * q + j] = e4[j];
                        }
                }
            };
            auto sum1 = [&](const T (&lv)[R1], const u32 (&ev)[R1], int row) {
                T acc = T(0);
#pragma unroll
                for (int t = 0; t < R1; ++t) acc = FMA(with_hi(lv[t], hi_of(lv[t]) ^ (ev[t] & 0x80000000u)), xr[t], acc);
#if POS1_LINEAR
                *(__attribute__((address_space(3))) T*)(smem + base + OFF_MID + row * ESZ) = CANON_MID ? T(0) + acc : acc;
#else
                *(__attribute__((address_space(3))) T*)(smem + base + pos1[row]) = CANON_MID ? T(0) + acc : acc;
#endif
            };
            constexpr int LAST1 = ROWS1 - 1;
            int row = sub;
            if (row < ROWS1) load1(la, ea, row);
#pragma nounroll
            for (; row < ROWS1; row += 2 * NSUB) {
                const int r2_ = row + NSUB, r3_ = row + 2 * NSUB;
                load1(lb, eb, r2_ < ROWS1 ? r2_ : LAST1);
                sum1(la, ea, row);
                load1(la, ea, r3_ < ROWS1 ? r3_ : LAST1);
                if (r2_ < ROWS1) sum1(lb, eb, r2_);
            }
        }
        __syncthreads();
#elif !SINGLE
        {
            T la[W1], ra[W1], lb[W1], rb[W1];
            auto load1 = [&](T (&lv)[W1], T (&rv)[W1], int row) {
#pragma unroll
                for (int q = 0; q < (W1 + 3) / 4; ++q) {
                    const u32x4 e4 = ent1_quad(row, q);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (4 * q + j < W1) {
                            lv[4 * q + j] = LDS(base + (e4[j] & 0xffffu));
                            rv[4 * q + j] = LDS(base + (e4[j] >> 16));
                        }
                }
            };
            auto sum1 = [&](const T (&lv)[W1], const T (&rv)[W1], int row) {
                T acc = T(0);
#pragma unroll
                for (int t = 0; t < W1; ++t) acc = MAC(lv[t], rv[t], acc);   // (l * r) * (+-1) then +=: the sign rides in the image
#if POS1_LINEAR
                *(__attribute__((address_space(3))) T*)(smem + base + OFF_MID + row * ESZ) = CANON_MID ? T(0) + acc : acc;
#else
                *(__attribute__((address_space(3))) T*)(smem + base + pos1[row]) = CANON_MID ? T(0) + acc : acc;
#endif
            };
            constexpr int LAST1 = ROWS1 - 1;
            int row = sub;
            if (row < ROWS1) load1(la, ra, row);
#pragma nounroll
            for (; row < ROWS1; row += 2 * NSUB) {
                const int r2_ = row + NSUB, r3_ = row + 2 * NSUB;
                load1(lb, rb, r2_ < ROWS1 ? r2_ : LAST1);
                sum1(la, ra, row);
                load1(la, ra, r3_ < ROWS1 ? r3_ : LAST1);
                if (r2_ < ROWS1) sum1(lb, rb, r2_);
            }
        }
        __syncthreads();
#endif
#endif
        // ---- list 2: (mid, other operand) -> out; TB terms' operands in flight while the previous TB are summed ----
#if SPLIT > 1
        T part[PASSES2];
#endif
#pragma unroll
        for (int pass = 0; pass < PASSES2; ++pass) {
            const int vrow = sub + pass * NSUB;          // (row, slice)
            if (vrow >= ROWS2 * SPLIT) break;
            const int row = vrow / SPLIT, slice = vrow % SPLIT;
            const u32 oo = oo_reg[pass];
#if (BETA || INIT_SRC) && SPLIT == 1
            T acc = acc0[pass];
#else
            T acc = T(0);                                // (slices start from zero: what the row adds onto joins them at the end)
#endif
#if SORTED
            // tolerance mode, sign-sorted: WPS plus-terms, then WMS minus-terms (both whole batches of TB; the padding multiplies the item's
            // zero element by itself): no sign per term -- two address additions and one fused multiply-add
            {
                T ma[TB], oa[TB], mb[TB], ob[TB];
                auto load2 = [&](T (&mv)[TB], T (&ov)[TB], int q) {      // TB terms from quad q of this (row, slice)
#pragma unroll
                    for (int i = 0; i < TB / 4; ++i) {
#if ENT2_MODE == 3
                        const u32x4 e4 = ((const __attribute__((address_space(3))) u32x4*)(smem + ENT2_AT))[vrow * (WSS / 4) + q + i];
#else
                        const u32x4 e4 = ((const u32x4*)ent2)[vrow * (WSS / 4) + q + i];
#endif
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            mv[4 * i + j] = LDS(base + (e4[j] & 0xffffu));
                            ov[4 * i + j] = LDS(base + (e4[j] >> 16));
                        }
                    }
                };
                constexpr int NBT = WSS / TB, NBP = WPS / TB;            // batches in all, plus-batches first
                auto sum2 = [&](const T (&mv)[TB], const T (&ov)[TB], int b) {
                    if (b < NBP) {
#pragma unroll
                        for (int t = 0; t < TB; ++t) acc = FMA(mv[t], ov[t], acc);
                    } else {
#pragma unroll
                        for (int t = 0; t < TB; ++t) acc = FMA(-mv[t], ov[t], acc);
                    }
                };
                load2(ma, oa, 0);
                int b = 0;
#pragma nounroll
                for (; b + 2 <= NBT; b += 2) {
                    load2(mb, ob, (b + 1) * (TB / 4));
                    sum2(ma, oa, b);
                    load2(ma, oa, (b + 2 < NBT ? b + 2 : NBT - 1) * (TB / 4));
                    sum2(mb, ob, b + 1);
                }
                if (NBT & 1) sum2(ma, oa, NBT - 1);
            }
#else
            constexpr int NB = WS / TB;                  // batches of TB terms
            constexpr int QPT = ENT2_MODE == 2 ? 2 : 4;  // terms per table quad
            const int q0 = slice * (WS / QPT);           // the slice's first quad (WS is a multiple of 4)
#if ENT2_MODE == 2
            T ma[TB], oa[TB], mb[TB], ob[TB];
            u32 sa[TB], sb[TB];
            auto load2 = [&](T (&mv)[TB], T (&ov)[TB], u32 (&sg)[TB], int b) {
#pragma unroll
                for (int q = 0; q < TB / 2; ++q) {
                    const u32x4 e4 = ent2_quad(row, q0 + (TB / 2) * b + q);
                    mv[2 * q] = LDS(base + (e4[0] & 0xffffu));
                    ov[2 * q] = LDS(base + (e4[0] >> 16));
                    sg[2 * q] = e4[1];
                    mv[2 * q + 1] = LDS(base + (e4[2] & 0xffffu));
                    ov[2 * q + 1] = LDS(base + (e4[2] >> 16));
                    sg[2 * q + 1] = e4[3];
                }
            };
            auto sum2 = [&](const T (&mv)[TB], const T (&ov)[TB], const u32 (&sg)[TB]) {
#pragma unroll
                for (int t = 0; t < TB; ++t) {
#if EXACT
                    const T pr_ = mv[t] * ov[t];
                    acc = acc + with_hi(pr_, hi_of(pr_) ^ sg[t]);            // eval.rs:82: (l * r) * (+-1), then +=
#else
                    acc = FMA(with_hi(mv[t], hi_of(mv[t]) ^ sg[t]), ov[t], acc);   // tolerance mode: one rounding per term
#endif
                }
            };
#else
            T ma[TB], oa[TB], sa[TB], mb[TB], ob[TB], sb[TB];
            auto load2 = [&](T (&mv)[TB], T (&ov)[TB], T (&sg)[TB], int b) {
#pragma unroll
                for (int q = 0; q < TB / 4; ++q) {
                    const u32x4 e4 = ent2_quad(row, q0 + (TB / 4) * b + q);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        mv[4 * q + j] = LDS(base + (e4[j] & 0xffffu));
                        ov[4 * q + j] = LDS(base + ((e4[j] >> 16) & 0x7fffu));
                        u32 hi;
                        asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(hi) : "v"(e4[j]), "s"(0x80000000u), "v"(one_hi));   // +-1.0: the high word
                        sg[4 * q + j] = pm_one(hi);
                    }
                }
            };
            auto sum2 = [&](const T (&mv)[TB], const T (&ov)[TB], const T (&sg)[TB]) {
#pragma unroll
                for (int t = 0; t < TB; ++t) acc = FMA(mv[t] * ov[t], sg[t], acc);   // eval.rs:82: the product is rounded, +-1 is exact
            };
#endif
            if (NB > 0) load2(ma, oa, sa, 0);
            int b = 0;
#pragma nounroll
            for (; b + 2 <= NB; b += 2) {
                load2(mb, ob, sb, b + 1);
                sum2(ma, oa, sa);
                load2(ma, oa, sa, b + 2 < NB ? b + 2 : NB - 1);
                sum2(mb, ob, sb);
            }
            if (NB & 1) sum2(ma, oa, sa);
#pragma unroll
            for (int t = NB * TB; t < WS; ++t) {         // the remainder (compile-time count, no padding terms)
#if ENT2_MODE == 2
                const u32x4 e4 = ent2_quad(row, q0 + t / 2);
                const u32 e = e4[2 * (t & 1)], sgb = e4[2 * (t & 1) + 1];
#if EXACT
                const T pr_ = LDS(base + (e & 0xffffu)) * LDS(base + (e >> 16));
                acc = acc + with_hi(pr_, hi_of(pr_) ^ sgb);
#else
                const T m_ = LDS(base + (e & 0xffffu));
                acc = FMA(with_hi(m_, hi_of(m_) ^ sgb), LDS(base + (e >> 16)), acc);
#endif
#else
                const u32 e = ent2_quad(row, q0 + t / 4)[t & 3];
                acc = FMA(LDS(base + (e & 0xffffu)) * LDS(base + ((e >> 16) & 0x7fffu)), pm_one((e & 0x80000000u) | ONE_HI), acc);
#endif
            }
#endif
#if SPLIT > 1
            part[pass] = acc;
#else
            if (it < nitems) out[(item0 + it) * s_out + oo] = acc;
#endif
        }
#if SPLIT > 1
        // the slices of a row meet through LDS (the mid rows are dead by now): partial sums in slice order onto the row's start value
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < PASSES2; ++pass) {
            const int vrow = sub + pass * NSUB;
            if (vrow < ROWS2 * SPLIT) *(__attribute__((address_space(3))) T*)(smem + base + OFF_MID + vrow * ESZ) = part[pass];
        }
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < PASSES2; ++pass) {
            const int vrow = sub + pass * NSUB;
            if (vrow < ROWS2 * SPLIT && vrow % SPLIT == 0 && it < nitems) {
#if BETA || INIT_SRC
                T acc = acc0[pass];
#else
                T acc = T(0);
#endif
#pragma unroll
                for (int sl = 0; sl < SPLIT; ++sl) acc = acc + LDS(base + OFF_MID + (vrow + sl) * ESZ);
                out[(item0 + it) * s_out + oo_reg[pass]] = acc;
            }
        }
#endif
        __syncthreads();   // the rows are rewritten by the next group
        if (!more) break;
        g = gn;
        if (pre_next) commit();
        else stage_all(gn * IPB, nnext);
    }
}
)JIT";
    c.chain_jit = 1;
    c.chain_jit_source = std::move(src);
    const int64_t lay[7] = {off_l1, off_r1, off_neg, off_mid, alias ? -1 : off_r2, stride, neg_is_left ? 1 : 0};
    for (int i = 0; i < 7; ++i) c.cj_layout[i] = int(lay[i]);
    c.cj_fmt[0] = int(w1s);
    c.cj_fmt[1] = ent2_mode;
    c.cj_sorted[0] = int(wps);
    c.cj_sorted[1] = int(wms);
    c.cj_sorted[2] = int(off_zero * esz);
    c.cj_xreg = xreg ? 1 : 0;
    c.cj_split = int(split);
    c.cj_ipb = int(ipb);
    c.cj_threads = int(threads);
    c.cj_lds = size_t(used);
}

// A list product whose result is read ONLY by another list product (as either operand): both run in ONE k_product_ell_chain
// launch with the mid row in LDS -- the rotor sandwich applied to a vector, (R X ~R).g(1), where it no longer fits a fused
// small-program kernel (n >= 9) and its second product (n rows of 2^(n-1) terms) is far too sparse for the dense kernels.  Same
// order, same roundings: bit for bit the two-launch plan.  Off with GAAST_FLAG_DEBUG_NO_CHAIN.
static void chain_list_into_list(Plan& plan) {
    if (plan.flags & (GAAST_FLAG_NO_FUSION | GAAST_FLAG_DEBUG_NO_CHAIN)) return;
    if (plan.node_dead.size() != plan.node_buffers.size()) plan.node_dead.assign(plan.node_buffers.size(), 0);
    const size_t elem = plan.dtype == GAAST_F32 ? 4 : 8;
    auto same = [](BufRef x, BufRef y) { return x.kind == y.kind && x.idx == y.idx; };
    auto row_len = [&](BufRef r) -> int64_t {
        return r.kind == BufKind::NODE ? plan.node_buffers[size_t(r.idx)].row_len : r.kind == BufKind::INPUT ? plan.input_layouts[size_t(r.idx)].row_len
                                                                                                           : plan.out_layout.row_len;
    };
    for (const Step& t : plan.steps)
        if (t.kind == Step::FUSED) return;
    for (size_t j = 0; j < plan.steps.size(); ++j) {
        Step& c = plan.steps[j];
        if (c.kind != Step::PRODUCT_CSR || c.ell_width <= 0 || !c.ell_bytes || c.list_chain) continue;
        for (int side = 1; side <= 2 && !c.list_chain; ++side) {
            const BufRef buf = side == 1 ? c.a : c.b, other = side == 1 ? c.b : c.a;
            if (buf.kind != BufKind::NODE || buf.idx < 0 || same(buf, other)) continue;
            int writer = -1;
            bool ok = true;
            for (size_t i = 0; i < plan.steps.size() && ok; ++i) {
                const Step& t = plan.steps[i];
                if (i == j) continue;
                if (same(t.res, buf)) {
                    if (writer >= 0 || t.kind != Step::PRODUCT_CSR || t.ell_width <= 0 || !t.ell_bytes || t.beta != 0 || t.list_chain || i > j) ok = false;
                    writer = int(i);
                }
                if ((t.a.idx >= 0 && same(t.a, buf)) || (t.b.idx >= 0 && same(t.b, buf))) ok = false;
                if ((t.chained || t.list_chain) && (same(t.pre_a, buf) || same(t.pre_b, buf))) ok = false;
            }
            if (!ok || writer < 0) continue;
            Step& w = plan.steps[size_t(writer)];
            if (same(w.a, buf) || same(w.b, buf) || same(c.res, w.a) || same(c.res, w.b)) continue;
            for (size_t i = size_t(writer) + 1; i < j && ok; ++i)
                if (same(plan.steps[i].res, w.a) || same(plan.steps[i].res, w.b)) ok = false;
            if (!ok) continue;
            const int64_t l1 = row_len(w.a), r1 = row_len(w.b), mid = row_len(buf), r2 = row_len(other);
            const int canon_other = side == 1 ? c.canon_b : c.canon_a;
            int alias = 0;
            if (same(other, w.a) && canon_other == w.canon_a) alias = 1;
            else if (same(other, w.b) && canon_other == w.canon_b) alias = 2;
            const int64_t per_item = l1 + r1 + mid + (alias ? 0 : r2);
            if (per_item * int64_t(elem) >= 32768 * 3) continue;       // byte offsets of the entries stay below 32 KiB per row anyway
            int64_t stride = per_item;
            while (stride % 32 != 1) ++stride;                         // consecutive items: consecutive banks
            // items per workgroup: about 36 KiB of LDS (four workgroups per CU: one's staging overlaps another's lists), at least
            // four items when that still fits the CU
            // this list's words ride in LDS when they take at most 48 KiB (n <= 11 for the sandwich)
            int64_t ent2 = ((int64_t(c.ell_width) + 4) * int64_t(c.u32_b.size()) * 4 + 15) / 16 * 16;   // [row][term], rows 4 words apart; widths are multiples of 4
            if (ent2 > 48 * 1024 || c.ell_width % 4) ent2 = 0;
            // items per workgroup: as many as make this list's (row, item) pairs just fill a wave (9 rows: 7 items = 63 lanes) -- a
            // second, nearly empty wave would issue the whole list again --, fewer when the LDS does not hold them
            const int64_t rows2 = int64_t(c.u32_b.size());
            int64_t ipb = rows2 <= 32 ? 64 / rows2 : 4;
            const int64_t lds_cap = int64_t(kLdsBytes) - 16 * 1024;
            while (ipb > 2 && ent2 + ipb * stride * int64_t(elem) > lds_cap / 2) --ipb;   // two workgroups per CU when that costs at most ...
            if (ipb < 4) {                                                                 // ... down to four items; then one workgroup per CU
                ipb = std::min<int64_t>(rows2 <= 32 ? 64 / rows2 : 4, 4);
                if (ent2 + ipb * stride * int64_t(elem) > lds_cap) ent2 = 0;
                while (ipb > 1 && ipb * stride * int64_t(elem) > lds_cap) --ipb;
            }
            if (ipb < 2 || ent2 + ipb * stride * int64_t(elem) > lds_cap) continue;
            c.chain_ent2_lds = int(ent2);
            // rows of the first list -> their element offsets in the mid row; is every component of the mid row produced?
            std::vector<char> produced(size_t(mid), 0);
            for (uint32_t off : w.u32_b) produced[off] = 1;
            bool covered = true;
            for (char x : produced) covered = covered && x;
            c.list_chain = side;
            c.chain_alias = alias;
            c.chain_mid_len = int(mid);
            c.chain_canon_mid = side == 1 ? c.canon_a : c.canon_b;
            c.chain_covered = covered ? 1 : 0;
            c.chain_ipb = int(ipb);
            c.chain_item_stride = int(stride);
            c.pre_a = w.a;
            c.pre_b = w.b;
            c.pre_canon_a = w.canon_a;
            c.pre_canon_b = w.canon_b;
            c.pre_left_len = int(l1);
            c.pre_right_len = int(r1);
            c.pre_entries = w.u32_c;
            c.pre_row_map = w.u32_b;
            c.pre_width = w.ell_width;
            c.name += " <- " + w.name + " in LDS";
            c.n_entries += w.n_entries;
            make_chain_jit(plan, c, &w, l1, r1, mid, r2, alias, side, covered);
            (side == 1 ? c.a : c.b) = BufRef{BufKind::NODE, -1};
            plan.node_dead[size_t(buf.idx)] = 1;
            w.kind = Step::ZERO;   // marks the first list for removal below
            w.res = BufRef{BufKind::NODE, -1};
        }
    }
    std::vector<Step> kept;
    for (Step& t : plan.steps)
        if (!(t.kind == Step::ZERO && t.res.kind == BufKind::NODE && t.res.idx < 0)) kept.push_back(std::move(t));
    plan.steps = std::move(kept);
}

// A RUN of element-wise arms on one buffer -- add_grades_from copies / additions of bound inputs, Negation / Reverse / GradeInvolution
// sign flips (eval.rs:45-60, 87-102) -- costs one full read-modify-write of the buffer PER ARM when every arm is a launch
// ((-(a.rev()) + b.ginvol()).rev() * s at n = 12: six passes).  The components never look at each other, so the run becomes ONE
// k_elementwise pass in which each component executes its own statements in program order: bit-identical.  When the run's buffer is
// only the operand of a product of one-term rows with a scalar operand, that product rides along as the pass's epilogue and the
// buffer is never written at all.
static void fuse_elementwise_runs(Plan& plan) {
    if (plan.flags & (GAAST_FLAG_NO_FUSION | GAAST_FLAG_DEBUG_NO_CHAIN)) return;
    auto same = [](BufRef x, BufRef y) { return x.kind == y.kind && x.idx == y.idx; };
    auto row_len = [&](BufRef r) -> int64_t {
        return r.kind == BufKind::NODE ? plan.node_buffers[size_t(r.idx)].row_len : r.kind == BufKind::INPUT ? plan.input_layouts[size_t(r.idx)].row_len
                                                                                                           : plan.out_layout.row_len;
    };
    for (size_t i = 0; i < plan.steps.size(); ++i) {
        const Step& first = plan.steps[i];
        if (first.kind != Step::AXPY && first.kind != Step::FLIP) continue;
        const BufRef R = first.res;
        size_t j = i;
        std::vector<BufRef> srcs;
        bool ok = true;
        while (j < plan.steps.size() && ok) {
            const Step& t = plan.steps[j];
            if ((t.kind != Step::AXPY && t.kind != Step::FLIP) || !same(t.res, R)) break;
            if (t.kind == Step::AXPY) {
                if (t.a.kind != BufKind::INPUT) break;
                bool known = false;
                for (const BufRef& b : srcs) known = known || same(b, t.a);
                if (!known) {
                    if (srcs.size() == 6) break;
                    srcs.push_back(t.a);
                }
            }
            ++j;
            if (j - i == 8) break;   // (a pass holds eight statements per component; a longer run continues in a second pass)
        }
        const size_t n_ops = j - i;
        if (n_ops < 2) continue;
        // components the run touches, in row order
        const int64_t rl = row_len(R);
        std::vector<int32_t> comp_of(size_t(rl), -1);
        std::vector<uint32_t> comps;
        for (size_t k = i; k < j; ++k)
            for (uint32_t w : plan.steps[k].u32_a) {
                const uint32_t off = plan.steps[k].kind == Step::AXPY ? (w & 0xffffu) : w;
                if (comp_of[off] < 0) comp_of[off] = 0;
            }
        for (int64_t o = 0; o < rl; ++o)
            if (comp_of[size_t(o)] == 0) {
                comp_of[size_t(o)] = int32_t(comps.size());
                comps.push_back(uint32_t(o));
            }
        const size_t nc = comps.size();
        if (nc == 0) continue;
        std::vector<uint32_t> ops(n_ops * nc, 0u);
        for (size_t k = i; k < j; ++k) {
            const Step& t = plan.steps[k];
            if (t.kind == Step::FLIP) {
                for (uint32_t off : t.u32_a) ops[(k - i) * nc + size_t(comp_of[off])] = 2u;
            } else {
                uint32_t slot = 0;
                for (size_t q = 0; q < srcs.size(); ++q)
                    if (same(srcs[q], t.a)) slot = uint32_t(q);
                for (uint32_t w : t.u32_a) ops[(k - i) * nc + size_t(comp_of[w & 0xffffu])] = (t.beta ? 1u : 3u) | (slot << 2) | ((w >> 16) << 16);
            }
        }
        // Every component executes ITS OWN statements in order, so each list is compacted on its own: empty statements go, and two
        // sign flips in a row cancel exactly (-(-x) has the bits of x) -- (-(a.rev()) + b.ginvol()).rev() on grade-6 rows is
        // copy, flip, flip, add, flip: three statements.  Four or fewer statements run on the kernel instantiation with half the
        // registers (k_elementwise<T, 4>).
        {
            std::vector<std::vector<uint32_t>> lists(nc);
            size_t longest = 1;
            for (size_t c = 0; c < nc; ++c) {
                for (size_t k = 0; k < n_ops; ++k) {
                    const uint32_t w = ops[k * nc + c];
                    if ((w & 3u) == 0u) continue;
                    if ((w & 3u) == 2u && !lists[c].empty() && (lists[c].back() & 3u) == 2u) lists[c].pop_back();
                    else lists[c].push_back(w);
                }
                longest = std::max(longest, lists[c].size());
            }
            ops.assign(longest * nc, 0u);
            for (size_t c = 0; c < nc; ++c)
                for (size_t k = 0; k < lists[c].size(); ++k) ops[k * nc + c] = lists[c][k];
        }
        const size_t n_stmt = ops.size() / nc;
        // does every component start with a copy (the run then never reads its buffer)?
        bool load_first = false;
        for (size_t c = 0; c < nc && !load_first; ++c) {
            size_t k = 0;
            while (k < n_stmt && ops[k * nc + c] == 0u) ++k;
            load_first = k == n_stmt || (ops[k * nc + c] & 3u) != 3u;
        }
        Step f;
        f.kind = Step::ELEMENTWISE;
        f.res = R;
        f.ew_src = srcs;
        f.ew_ops = int(n_stmt);
        f.ew_load_first = load_first ? 1 : 0;
        f.u32_a = std::move(ops);
        f.u32_b = comps;
        f.name = "elementwise[" + std::to_string(n_ops) + " arms:";
        for (size_t k = i; k < j; ++k) f.name += " " + plan.steps[k].name.substr(0, plan.steps[k].name.find('[') == std::string::npos ? 8 : plan.steps[k].name.find('['));
        f.name += "]";
        // the scaling product right after the run, reading the run's buffer and a scalar?
        size_t drop_to = j;
        if (R.kind == BufKind::NODE && j < plan.steps.size() && !load_first && int64_t(nc) == rl) {
            const Step& p2 = plan.steps[j];
            const bool r_left = p2.kind == Step::PRODUCT_CSR && same(p2.a, R), r_right = p2.kind == Step::PRODUCT_CSR && same(p2.b, R);
            if (p2.kind == Step::PRODUCT_CSR && p2.beta == 0 && r_left != r_right && !p2.list_chain && !p2.chained && p2.u32_b.size() == nc) {
                const BufRef S = r_left ? p2.b : p2.a;
                bool fits = row_len(S) == 1 && !same(S, R) && !same(p2.res, R);
                for (size_t r = 0; r + 1 < p2.u32_a.size() && fits; ++r) fits = p2.u32_a[r + 1] - p2.u32_a[r] == 1;
                for (size_t k = 0; k < plan.steps.size() && fits; ++k) {   // nobody else reads the run's buffer
                    if (k >= i && k <= j) continue;
                    const Step& t = plan.steps[k];
                    if (same(t.res, R) || (t.a.idx >= 0 && same(t.a, R)) || (t.b.idx >= 0 && same(t.b, R))) fits = false;
                    if ((t.chained || t.list_chain) && (same(t.pre_a, R) || same(t.pre_b, R))) fits = false;
                }
                std::vector<uint32_t> out_off(nc, 0u);
                std::vector<double> coeff(nc, 0.0);
                std::vector<char> seen(nc, 0);
                for (size_t r = 0; r + 1 < p2.u32_a.size() && fits; ++r) {
                    const uint32_t e = p2.u32_c[p2.u32_a[r]];
                    const uint32_t roff = r_left ? (e & 0xffffu) : (e >> 16);
                    fits = roff < uint32_t(rl) && comp_of[roff] >= 0 && !seen[size_t(comp_of[roff])];
                    if (!fits) break;
                    seen[size_t(comp_of[roff])] = 1;
                    out_off[size_t(comp_of[roff])] = p2.u32_b[r];
                    coeff[size_t(comp_of[roff])] = p2.coeff[p2.u32_a[r]];
                }
                if (fits) {
                    f.ew_scale = 1;
                    f.res = p2.res;
                    f.b = S;
                    f.u32_c = std::move(out_off);
                    f.coeff = std::move(coeff);
                    f.ew_scalar_off = 0;
                    f.ew_canon_v = r_left ? p2.canon_a : p2.canon_b;
                    f.ew_canon_s = r_left ? p2.canon_b : p2.canon_a;
                    f.ew_s_is_left = r_left ? 0 : 1;
                    f.beta = 0;
                    f.n_entries = p2.n_entries;
                    f.name += " * scalar -> " + p2.name;
                    if (plan.node_dead.size() != plan.node_buffers.size()) plan.node_dead.assign(plan.node_buffers.size(), 0);
                    plan.node_dead[size_t(R.idx)] = 1;
                    drop_to = j + 1;
                }
            }
        }
        std::vector<Step> kept;
        for (size_t k = 0; k < plan.steps.size(); ++k) {
            if (k == i) kept.push_back(std::move(f));
            else if (k > i && k < drop_to) continue;
            else kept.push_back(std::move(plan.steps[k]));
        }
        plan.steps = std::move(kept);
    }
}

// x (*) f(<l, r>): a product into ONE scalar component (a single row: norm_sq), an optional ScalarUnaryOp on it, and a product of
// one-term rows that multiplies another row by that scalar -- a.rev() * a.norm_sq().sinv(), the versor inverse of expr.rs:363-371,
// where the rows no longer fit a fused slab -- become ONE k_reduce_scale launch (kernels_exact.hip.hpp): reference order and
// roundings, the scalar never leaves the wave, the row is streamed once from HBM.  Runs on the CSR form, before the ELL pass.
static void fuse_reduce_scale(Plan& plan) {
    if (plan.flags & (GAAST_FLAG_NO_FUSION | GAAST_FLAG_DEBUG_NO_CHAIN)) return;
    auto same = [](BufRef x, BufRef y) { return x.kind == y.kind && x.idx == y.idx; };
    for (size_t i = 0; i + 1 < plan.steps.size(); ++i) {
        Step& p1 = plan.steps[i];
        if (p1.kind != Step::PRODUCT_CSR || p1.u32_b.size() != 1 || p1.beta != 0 || p1.res.kind != BufKind::NODE || p1.list_chain || p1.chained) continue;
        if (plan.node_buffers[size_t(p1.res.idx)].row_len != 1 || p1.u32_c.size() < 64) continue;
        const BufRef S = p1.res;
        size_t j = i + 1;
        int op = 0;
        if (plan.steps[j].kind == Step::SUNARY && same(plan.steps[j].res, S) && plan.steps[j].sunary_off == 0) {
            op = plan.steps[j].sunary_op == 0 ? 1 : 2;
            ++j;
        }
        if (j >= plan.steps.size()) continue;
        Step& p2 = plan.steps[j];
        if (p2.kind != Step::PRODUCT_CSR || p2.beta != 0 || p2.list_chain || p2.chained) continue;
        const bool s_left = same(p2.a, S), s_right = same(p2.b, S);
        if (s_left == s_right) continue;
        const BufRef xop = s_left ? p2.b : p2.a;
        if (same(xop, S) || same(p2.res, p1.a) || same(p2.res, p1.b) || same(p2.res, xop)) continue;
        bool ok = true;
        for (size_t r = 0; r + 1 < p2.u32_a.size() && ok; ++r) ok = p2.u32_a[r + 1] - p2.u32_a[r] == 1;   // one term per row
        for (size_t k = 0; k < plan.steps.size() && ok; ++k) {   // nobody else touches the scalar
            if (k == i || k == j || (op && k == i + 1)) continue;
            const Step& t = plan.steps[k];
            if (same(t.res, S) || (t.a.idx >= 0 && same(t.a, S)) || (t.b.idx >= 0 && same(t.b, S))) ok = false;
            if ((t.chained || t.list_chain) && (same(t.pre_a, S) || same(t.pre_b, S))) ok = false;
        }
        if (!ok) continue;
        Step f;
        f.kind = Step::REDUCE_SCALE;
        f.res = p2.res;
        f.a = p1.a;
        f.b = p1.b;
        f.pre_a = xop;
        f.canon_a = p1.canon_a;
        f.canon_b = p1.canon_b;
        f.pre_canon_a = s_left ? p2.canon_b : p2.canon_a;
        f.rs_canon_s = s_left ? p2.canon_a : p2.canon_b;
        f.rs_op = op;
        f.list_chain = s_left ? 1 : 2;   // (which side of the scaling the scalar is on; run_step reads it)
        f.u32_a = p1.u32_c;
        f.coeff = p1.coeff;
        f.u32_b.resize(p2.u32_c.size());
        f.coeff_b = p2.coeff;
        for (size_t r = 0; r + 1 < p2.u32_a.size(); ++r) {
            const uint32_t e = p2.u32_c[p2.u32_a[r]];
            const uint32_t xoff = s_left ? (e >> 16) : (e & 0xffffu);
            f.u32_b[r] = xoff | (p2.u32_b[r] << 16);
        }
        f.beta = 0;
        f.n_entries = p1.n_entries + p2.n_entries;
        // tolerance mode: can a wave keep the row in registers and read it once (k_reduce_scale_wave)?  The reduction must be one term
        // per component, (i, i), coefficient +-1, in any order; the scaling one row per component in place (x offset = out offset),
        // +-1; the row a whole number of 64 x 16 bytes with at most 32 components per lane.  (Whether the three rows ARE one row is
        // known when they are bound: run_step.)
        if (!(plan.flags & GAAST_FLAG_EXACT_ORDER) && same(p1.a, p1.b) && same(xop, p1.a)) {
            const size_t per_piece = 64 * (plan.dtype == GAAST_F32 ? 4 : 2);   // components a wave moves per 16-byte load
            const size_t R = f.u32_a.size();
            bool okw = R == f.u32_b.size() && R % per_piece == 0 && (R / per_piece) * (per_piece / 64) <= 32 && f.coeff.size() == R && f.coeff_b.size() == R;
            const size_t pieces = okw ? R / per_piece : 0;
            okw = okw && (pieces == 1 || pieces == 2 || pieces == 4 || pieces == 8 || pieces == 16);
            std::vector<uint32_t> sg(128, 0u);
            std::vector<char> seen1(R, 0), seen2(R, 0);
            const size_t ec = per_piece / 64;
            auto place = [&](size_t c, int which) {   // component c of the row -> (lane, bit)
                const size_t piece = c / ec, e = c % ec, m = piece / 64, lane = piece % 64;
                sg[size_t(which) * 64 + lane] |= 1u << (m * ec + e);
            };
            for (size_t t = 0; okw && t < R; ++t) {
                const uint32_t li = f.u32_a[t] & 0xffffu, ri = f.u32_a[t] >> 16;
                okw = li == ri && li < R && !seen1[li] && (f.coeff[t] == 1.0 || f.coeff[t] == -1.0);
                if (okw) {
                    seen1[li] = 1;
                    if (f.coeff[t] < 0) place(li, 0);
                }
            }
            for (size_t r = 0; okw && r < R; ++r) {
                const uint32_t xo = f.u32_b[r] & 0xffffu, oo = f.u32_b[r] >> 16;
                okw = xo == oo && xo < R && !seen2[xo] && (f.coeff_b[r] == 1.0 || f.coeff_b[r] == -1.0);
                if (okw) {
                    seen2[xo] = 1;
                    if (f.coeff_b[r] < 0) place(xo, 1);
                }
            }
            if (okw) {
                f.rs_wave = int(pieces);
                f.u32_c = std::move(sg);
            }
        }
        f.name = "reduce_scale[" + std::to_string(p1.u32_c.size()) + " comp-muls -> scalar" + (op == 1 ? ", 1/s" : op == 2 ? ", sqrt(s)" : "") + ", " +
                 std::to_string(p2.u32_c.size()) + " scaled components]";
        if (plan.node_dead.size() != plan.node_buffers.size()) plan.node_dead.assign(plan.node_buffers.size(), 0);
        plan.node_dead[size_t(S.idx)] = 1;
        // out offsets need 16 bits
        bool fits = true;
        for (uint32_t o : p2.u32_b) fits = fits && o < 65536u;
        if (!fits) continue;
        std::vector<Step> kept;
        for (size_t k = 0; k < plan.steps.size(); ++k) {
            if (k == i) kept.push_back(std::move(f));
            else if (k == j || (op && k == i + 1)) continue;
            else kept.push_back(std::move(plan.steps[k]));
        }
        plan.steps = std::move(kept);
        return fuse_reduce_scale(plan);   // (indices moved: look again for another instance)
    }
}

// A list with FEW LONG rows that stayed a launch of its own -- b * c projected on a low grade: (a + b * c).g(2) at n = 8 is 28 rows of 256
// terms -- gives k_product_ell (a thread per row) 28 busy threads per workgroup.  The specialised chain kernel's second list is exactly
// this shape (lane = (row, item), 32 items per workgroup): the list runs on it alone.  When the step before it is the covering copy of
// an input's grades into the same buffer (a + ...), that copy is folded in as well: the accumulators start from 0.0 + a -- ONE launch.
// Same order, same roundings; k_product_ell (and the copy) stay in charge when hiprtc is not available.
static void jit_long_row_lists(Plan& plan) {
    if (plan.flags & (GAAST_FLAG_NO_FUSION | GAAST_FLAG_NO_JIT | GAAST_FLAG_DEBUG_JIT_FAILS | GAAST_FLAG_DEBUG_NO_CHAIN)) return;
    auto same = [](BufRef x, BufRef y) { return x.kind == y.kind && x.idx == y.idx; };
    auto row_len = [&](BufRef r) -> int64_t {
        return r.kind == BufKind::NODE ? plan.node_buffers[size_t(r.idx)].row_len : r.kind == BufKind::INPUT ? plan.input_layouts[size_t(r.idx)].row_len
                                                                                                           : plan.out_layout.row_len;
    };
    for (size_t j = 0; j < plan.steps.size(); ++j) {
        Step& c = plan.steps[j];
        if (c.kind != Step::PRODUCT_CSR || c.ell_width < 32 || !c.ell_bytes || c.list_chain || c.chain_jit) continue;
        const int64_t rows = int64_t(c.u32_b.size());
        if (rows > 128 || c.a.idx < 0 || c.b.idx < 0) continue;
        const int64_t la = row_len(c.a), lb = row_len(c.b);
        // the covering copy right before it, into the same buffer?
        std::vector<uint32_t> init;
        bool fold = false;
        if (j > 0 && c.beta == 1) {
            const Step& ax = plan.steps[j - 1];
            if (ax.kind == Step::AXPY && ax.beta == 0 && same(ax.res, c.res) && ax.a.kind == BufKind::INPUT && int64_t(ax.u32_a.size()) == rows) {
                std::vector<int64_t> src_of(size_t(row_len(c.res)), -1);
                for (uint32_t m : ax.u32_a) src_of[m & 0xffffu] = int64_t(m >> 16);
                fold = true;
                for (int64_t r = 0; r < rows && fold; ++r) {
                    fold = src_of[c.u32_b[size_t(r)]] >= 0;
                    init.push_back(uint32_t(fold ? src_of[c.u32_b[size_t(r)]] : 0));
                }
            }
        }
        make_chain_jit(plan, c, nullptr, 0, 0, la, lb, 0, 1, true, fold ? &init : nullptr);
        if (!c.chain_jit) continue;
        c.list_jit = 1;
        if (fold) {
            c.fold_prev = 1;
            c.pre_a = plan.steps[j - 1].a;   // the copy's source: the accumulators' starting values
        }
    }
}

void build_plan(const gaast_program_desc& desc, Plan& plan) {
    if (desc.vec_space_dim < 0 || desc.vec_space_dim > GAAST_MAX_DIM) throw std::runtime_error("vec_space_dim out of range");
    if (desc.n_nodes <= 0 || desc.root < 0 || desc.root >= desc.n_nodes) throw std::runtime_error("bad node count / root");
    if (desc.dtype != GAAST_F64 && desc.dtype != GAAST_F32) throw std::runtime_error("bad dtype");
    if (desc.n_inputs < 0 || desc.n_inputs > GAAST_MAX_INPUTS) throw std::runtime_error("too many inputs");
    plan.n = desc.vec_space_dim;
    plan.dtype = desc.dtype;
    plan.flags = desc.flags;
    plan.metric.assign(desc.metric_diag, desc.metric_diag + desc.vec_space_dim);
    plan.inputs.assign(desc.inputs, desc.inputs + desc.n_inputs);
    plan.const_rows.resize(size_t(desc.n_inputs));
    plan.input_layouts.resize(size_t(desc.n_inputs));
    for (int i = 0; i < desc.n_inputs; ++i) {
        const gaast_input_desc& in = desc.inputs[i];
        if (in.storage_dim < 0 || in.storage_dim > GAAST_MAX_DIM) throw std::runtime_error("input storage_dim out of range");
        plan.input_layouts[size_t(i)] = make_layout(in.storage_dim, in.grade_mask);
        if (in.is_const) {
            const int64_t len = plan.input_layouts[size_t(i)].row_len;
            if (len && !in.const_row) throw std::runtime_error("constant input without data");
            plan.const_rows[size_t(i)].assign(in.const_row, in.const_row + len);
            plan.inputs[size_t(i)].const_row = nullptr;  // no host pointer is retained
        }
    }
    for (int i = 0; i < desc.n_nodes; ++i) {
        const gaast_node_desc& nd = desc.nodes[i];
        auto child_ok = [&](int c) { return c >= 0 && c < i; };
        switch (nd.opcode) {
        case GAAST_OP_INPUT:
            if (nd.input_slot < 0 || nd.input_slot >= desc.n_inputs) throw std::runtime_error("input slot out of range");
            break;
        case GAAST_OP_ADD:
        case GAAST_OP_PRODUCT:
            if (!child_ok(nd.child0) || !child_ok(nd.child1)) throw std::runtime_error("nodes are not in post-order");
            break;
        case GAAST_OP_NEG: case GAAST_OP_EXP: case GAAST_OP_LOG: case GAAST_OP_PROJ:
        case GAAST_OP_REVERSE: case GAAST_OP_GINVOL: case GAAST_OP_SINV: case GAAST_OP_SSQRT:
            if (!child_ok(nd.child0)) throw std::runtime_error("nodes are not in post-order");
            break;
        default: throw std::runtime_error("unknown opcode");
        }
        if (nd.vec_space_dim < 0 || nd.vec_space_dim > GAAST_MAX_DIM) throw std::runtime_error("node dim out of range");
    }

    Lowering lw(desc, plan);
    // eval.rs:12-19: the root is cached like any other node; its buffer is the caller's `out`
    const gaast_node_desc& root = desc.nodes[desc.root];
    plan.out_layout = make_layout(root.vec_space_dim, root.minimal_grade_mask);
    BufRef out{BufKind::OUT, 0};
    lw.emit_zero(out);
    lw.add_to_res(out, desc.root);
    // drop the zero-fills that were folded into products
    std::vector<Step> kept;
    for (size_t i = 0; i < plan.steps.size(); ++i)
        if (!lw.removed[i]) kept.push_back(std::move(plan.steps[i]));
    plan.steps = std::move(kept);
    plan.node_dead.assign(plan.node_buffers.size(), 0);
    // A program that would only fit the LDS interpreter (its slab is beyond the registers of the hiprtc-specialised kernel) but is
    // exactly one list chain -- (R X ~R).g(1) at n = 8 -- runs on k_product_ell_chain instead (same box: 1.43 against 1.59 ms per 1 M items)
    int slab = 0;
    try_fuse(plan, &slab);
    if (slab > (plan.dtype == GAAST_F32 ? 200 : 160) && !(plan.flags & (GAAST_FLAG_NO_FUSION | GAAST_FLAG_DEBUG_NO_CHAIN | GAAST_FLAG_NO_JIT))) {
        Plan trial = plan;
        uniform_csr_to_ell(trial);
        chain_list_into_list(trial);
        if (trial.steps.size() == 1 && trial.steps[0].list_chain) {
            plan = std::move(trial);
            return;
        }
    }
    if (!try_fuse(plan)) {
        fuse_elementwise_runs(plan);
        fuse_reduce_scale(plan);
        chain_sparse_into_dense(plan);
        uniform_csr_to_ell(plan);
        chain_list_into_list(plan);
        jit_long_row_lists(plan);
    }
}

}  // namespace gaast
