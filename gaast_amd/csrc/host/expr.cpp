// Host-side mirror of gaast's phases 1-3.  See expr.hpp.
#include "expr.hpp"

#include <algorithm>

#include "../common/comp_mul_table.hpp"

namespace gaast {

// ---------------------------------------------------------------------------------------------
// phase 1
// ---------------------------------------------------------------------------------------------
static ExprPtr node(ExprNode::Kind k, ExprPtr a = nullptr, ExprPtr b = nullptr) {
    auto e = std::make_shared<ExprNode>();
    e->kind = k;
    e->a = std::move(a);
    e->b = std::move(b);
    return e;
}

ExprPtr make_input(int slot, uint64_t mask, int storage_dim) {  // mv(x), expr.rs:162-164
    auto e = node(ExprNode::MV);
    e->mv_slot = slot;
    e->mv_mask = mask;
    e->mv_storage_dim = storage_dim;
    return e;
}

ExprPtr make_const(uint64_t mask, int storage_dim, const double* row, size_t len) {
    auto e = node(ExprNode::MV);
    e->mv_slot = -1;
    e->mv_mask = mask;
    e->mv_storage_dim = storage_dim;
    e->mv_const_row.assign(row, row + len);
    return e;
}

ExprPtr make_from_f64(double x) {  // From<f64>, expr.rs:231-240
    if (x == 0.0) return make_const(0, 0, nullptr, 0);  // init_null_mv(0, empty)
    return make_const(1, 0, &x, 1);                     // {0: [x]} with dim 0
}

ExprPtr make_basis_vector(int dim, int i) {  // basis_vectors, expr.rs:148-157
    std::vector<double> v(size_t(dim), 0.0);
    v[size_t(i)] = 1.0;
    return make_const(2, dim, v.data(), v.size());
}

ExprPtr make_product(ExprPtr l, ExprPtr r, Selection sel) {  // product, expr.rs:123-144
    auto e = node(ExprNode::PRODUCT, std::move(l), std::move(r));
    e->sel = sel;
    return e;
}

ExprPtr make_binary(ExprNode::Kind k, ExprPtr l, ExprPtr r) { return node(k, std::move(l), std::move(r)); }
ExprPtr make_unary(ExprNode::Kind k, ExprPtr e) { return node(k, std::move(e)); }

ExprPtr make_g(ExprPtr e, int64_t k) {  // g, expr.rs:322-324
    auto x = node(ExprNode::GSELECT, std::move(e));
    x->gsel_single = true;
    x->gsel_k = k;
    return x;
}

ExprPtr make_gselect(ExprPtr e, uint64_t mask) {  // gselect, expr.rs:327-335
    auto x = node(ExprNode::GSELECT, std::move(e));
    x->gsel_single = false;
    x->gsel_mask = mask;
    return x;
}

static Selection geometric() { return Selection{GAAST_PROD_GEOMETRIC, nullptr, nullptr}; }

ExprPtr make_sub(ExprPtr l, ExprPtr r) {  // Sub, expr.rs:224-229: self + -rhs
    return make_binary(ExprNode::ADD, std::move(l), make_unary(ExprNode::NEG, std::move(r)));
}

ExprPtr make_div_scalar(ExprPtr e, double s) {  // Div, expr.rs:265-270
    return make_product(std::move(e), make_from_f64(1.0 / s), geometric());
}

ExprPtr make_pow(ExprPtr e, ExprPtr p) {  // pow, expr.rs:300-302: exp(log(self) * p)
    return make_unary(ExprNode::EXP,
                      make_product(make_unary(ExprNode::LOG, std::move(e)), std::move(p), geometric()));
}

ExprPtr make_conj(ExprPtr e) {  // conj, expr.rs:338-340
    return make_unary(ExprNode::GINVOL, make_unary(ExprNode::REV, std::move(e)));
}

ExprPtr make_scal(ExprPtr e, ExprPtr rhs) {  // scal, expr.rs:343-345: (self.rev() * rhs).g(0)
    return make_g(make_product(make_unary(ExprNode::REV, std::move(e)), std::move(rhs), geometric()), 0);
}

ExprPtr make_norm_sq(ExprPtr e) {  // norm_sq, expr.rs:348-350: self.clone().scal(self)
    ExprPtr c = e;
    return make_scal(std::move(c), std::move(e));
}

// ---------------------------------------------------------------------------------------------
// phase 2: reify (expr.rs:13-25, 62-115)
// ---------------------------------------------------------------------------------------------
namespace {

// A grade set together with the length of the reference's BitVec.  The length never changes
// which grades are present; it only matters to GradeSet::includes (grade_set.rs:149-151), whose
// `self.bv | other.bv` keeps self's length and so never sees grades of `other` at positions
// >= self.len -- the assert of specialize.rs:113-117 must pass / fail exactly as upstream.
struct GS {
    uint64_t mask = 0;
    int len = 0;
};
inline int top_len(uint64_t m) { return m ? 64 - __builtin_clzll(m) : 0; }
inline GS gs_union(GS a, GS b) { return GS{a.mask | b.mask, a.len > b.len ? a.len : b.len}; }  // :287-293
inline GS sel_gs(const Selection& sel, int64_t k1, int64_t k2) {
    GS g;
    g.mask = sel(k1, k2);
    switch (sel.kind) {
    case GAAST_PROD_GEOMETRIC: g.len = int(k1 + k2 + 1); break;              // mul: big.len + small.len - 1
    case GAAST_PROD_OUTER: g.len = int(k1 + k2 + 1); break;                   // single(k1 + k2)
    case GAAST_PROD_INNER: g.len = (k1 == 0 || k2 == 0) ? 0 : int((k1 > k2 ? k1 - k2 : k2 - k1) + 1); break;
    case GAAST_PROD_LCONTRACT: g.len = k2 - k1 < 0 ? 0 : int(k2 - k1 + 1); break;
    case GAAST_PROD_RCONTRACT: g.len = k1 - k2 < 0 ? 0 : int(k1 - k2 + 1); break;
    default: g.len = top_len(g.mask); break;  // user closure: as built by add_grade
    }
    return g;
}

struct Builder {
    SpecializedAst& ast;
    std::unordered_map<const ExprNode*, int> index;
    uint64_t full_mask;  // Algebra::full_grade_set, algebra.rs:19-21

    int add_node(const ExprNode* id, GradedNode proto, GS node_gs) {  // expr.rs:13-25
        proto.id = id;
        proto.maximal = node_gs.mask & full_mask;  // node_gs.intersection(full): keeps node_gs's length
        proto.maximal_len = node_gs.len;
        proto.minimal = 0;
        proto.vec_space_dim = ast.n;
        proto.num_uses = 1;
        proto.is_ready = false;
        ast.nodes.push_back(std::move(proto));
        int idx = int(ast.nodes.size()) - 1;
        index[id] = idx;
        return idx;
    }

    // reify_or_reuse, expr.rs:73-84
    int reify_or_reuse(const ExprPtr& e, GS* gs) {
        auto it = index.find(e.get());
        int idx;
        if (it == index.end()) {
            run(e, e.get());
            idx = index.at(e.get());
        } else {
            idx = it->second;
            ast.nodes[size_t(idx)].num_uses += 1;
        }
        *gs = GS{ast.nodes[size_t(idx)].maximal, ast.nodes[size_t(idx)].maximal_len};
        return idx;
    }

    // body of the Expr's `run` closure; this_id differs from e only under `wrap`
    void run(const ExprPtr& e, const ExprNode* this_id) {
        GradedNode p;
        GS gs, lgs, rgs;
        switch (e->kind) {
        case ExprNode::MV:
            p.opcode = GAAST_OP_INPUT;
            p.input = e.get();
            if (e->mv_slot < 0 && !ast.const_slot.count(e.get())) {
                ast.const_nodes.push_back(e.get());
                ast.const_slot[e.get()] = -int(ast.const_nodes.size());  // -(1 + const index)
            }
            add_node(this_id, p, GS{e->mv_mask, top_len(e->mv_mask)});  // fold of add_grade, graded.rs:176-184
            return;
        case ExprNode::ADD: {  // expr.rs:204-209
            p.child0 = reify_or_reuse(e->a, &lgs);
            p.child1 = reify_or_reuse(e->b, &rgs);
            p.opcode = GAAST_OP_ADD;
            add_node(this_id, p, gs_union(lgs, rgs));
            return;
        }
        case ExprNode::PRODUCT: {  // expr.rs:129-143
            p.child0 = reify_or_reuse(e->a, &lgs);
            p.child1 = reify_or_reuse(e->b, &rgs);
            p.opcode = GAAST_OP_PRODUCT;
            p.sel = e->sel;
            for (int kl = 0; kl < 64; ++kl)
                if ((lgs.mask >> kl) & 1ULL)
                    for (int kr = 0; kr < 64; ++kr)
                        if ((rgs.mask >> kr) & 1ULL) gs = gs_union(gs, sel_gs(e->sel, kl, kr));
            add_node(this_id, p, gs);
            return;
        }
        case ExprNode::NEG:
        case ExprNode::REV:
        case ExprNode::GINVOL:
        case ExprNode::SINV: {
            p.child0 = reify_or_reuse(e->a, &gs);
            p.opcode = e->kind == ExprNode::NEG      ? GAAST_OP_NEG
                       : e->kind == ExprNode::REV    ? GAAST_OP_REVERSE
                       : e->kind == ExprNode::GINVOL ? GAAST_OP_GINVOL
                                                     : GAAST_OP_SINV;
            add_node(this_id, p, gs);
            return;
        }
        case ExprNode::EXP: {  // GradeSet::exp, grade_set.rs:181-187
            p.child0 = reify_or_reuse(e->a, &gs);
            if (__builtin_popcountll(gs.mask) != 1)
                throw SpecError{GAAST_ERR_INVALID_PROGRAM,
                                "exp cannot be used on a multivector, only a k-vector"};
            p.opcode = GAAST_OP_EXP;
            add_node(this_id, p, gs_union(GS{1ULL, 1}, gs));
            return;
        }
        case ExprNode::LOG: {  // GradeSet::log, grade_set.rs:190-197
            p.child0 = reify_or_reuse(e->a, &gs);
            uint64_t other = gs.mask & ~1ULL;
            if (__builtin_popcountll(other) != 1)
                throw SpecError{GAAST_ERR_INVALID_PROGRAM,
                                "log can only be used on multivectors of the form <A>_0 + <A>_k"};
            p.opcode = GAAST_OP_LOG;
            add_node(this_id, p, GS{other, gs.len});
            return;
        }
        case ExprNode::GSELECT: {  // expr.rs:327-335
            p.child0 = reify_or_reuse(e->a, &gs);
            GS wanted = e->gsel_single ? GS{gs_single(e->gsel_k), e->gsel_k < 0 ? 0 : int(e->gsel_k + 1)}
                                       : GS{e->gsel_mask, top_len(e->gsel_mask)};
            p.opcode = GAAST_OP_PROJ;
            add_node(this_id, p, GS{wanted.mask & gs.mask, wanted.len});  // wanted.intersection(gs)
            return;
        }
        case ExprNode::WRAP_SQRT:
        case ExprNode::WRAP_VINV: {  // wrap, expr.rs:97-115
            int self_idx = reify_or_reuse(e->a, &gs);
            const bool just_scalar = gs.mask == 1ULL;  // is_just(0)
            if (e->kind == ExprNode::WRAP_SQRT && just_scalar) {  // Wrapper::Node, expr.rs:310-314
                p.child0 = self_idx;
                p.opcode = GAAST_OP_SSQRT;
                add_node(this_id, p, gs);
                return;
            }
            ExprPtr inner;
            if (e->kind == ExprNode::WRAP_SQRT) {
                inner = make_pow(e->a, make_from_f64(0.5));  // expr.rs:316
            } else if (just_scalar) {
                inner = make_unary(ExprNode::SINV, e->a);    // expr.rs:365-366
            } else {                                         // expr.rs:368
                inner = make_product(make_unary(ExprNode::REV, e->a),
                                     make_unary(ExprNode::SINV, make_norm_sq(e->a)), geometric());
            }
            ast.temps.push_back(inner);
            run(inner, this_id);                              // (wrapper_expr.run)(wrapper_id, b)
            ast.nodes[size_t(self_idx)].num_uses -= 1;        // expr.rs:107-110
            return;
        }
        }
    }
};

// ---------------------------------------------------------------------------------------------
// phase 3 (specialize.rs:53-183)
// ---------------------------------------------------------------------------------------------
void rec_update_minimal(SpecializedAst& s, int idx, uint64_t wanted) {  // specialize.rs:53-94
    s.nodes[size_t(idx)].minimal |= wanted;
    const GradedNode& n = s.nodes[size_t(idx)];
    switch (n.opcode) {
    case GAAST_OP_INPUT: return;
    case GAAST_OP_PROJ:
    case GAAST_OP_NEG:
    case GAAST_OP_REVERSE:
    case GAAST_OP_GINVOL:
    case GAAST_OP_SINV:
    case GAAST_OP_SSQRT: rec_update_minimal(s, n.child0, wanted); return;
    case GAAST_OP_ADD: {
        int l = n.child0, r = n.child1;
        rec_update_minimal(s, l, wanted);
        rec_update_minimal(s, r, wanted);
        return;
    }
    case GAAST_OP_PRODUCT: {
        int l = n.child0, r = n.child1;
        uint64_t lw, rw;
        parts_contributing(wanted, n.sel, s.nodes[size_t(l)].maximal, s.nodes[size_t(r)].maximal,
                           &lw, &rw);
        rec_update_minimal(s, l, lw);
        rec_update_minimal(s, r, rw);
        return;
    }
    case GAAST_OP_EXP: {  // wanted.log(), specialize.rs:91
        uint64_t other = wanted & ~1ULL;
        if (__builtin_popcountll(other) != 1)
            throw SpecError{GAAST_ERR_INVALID_PROGRAM,
                            "log can only be used on multivectors of the form <A>_0 + <A>_k"};
        rec_update_minimal(s, n.child0, other);
        return;
    }
    case GAAST_OP_LOG: {  // wanted.exp(), specialize.rs:92
        if (__builtin_popcountll(wanted) != 1)
            throw SpecError{GAAST_ERR_INVALID_PROGRAM,
                            "exp cannot be used on a multivector, only a k-vector"};
        rec_update_minimal(s, n.child0, wanted | 1ULL);
        return;
    }
    }
}

void rec_apply_algebra(SpecializedAst& s, int idx, const BladeTable& bt, uint64_t limit) {
    GradedNode& n = s.nodes[size_t(idx)];  // specialize.rs:96-160
    if (n.is_ready) {
        if (n.num_uses < 2)
            throw SpecError{GAAST_ERR_INVALID_PROGRAM,
                            "Algebra was already applied to a node that is referred to only once"};
        return;
    }
    n.is_ready = true;
    // maximal.includes(minimal), grade_set.rs:149-151, with the BitVec-length blind spot
    const uint64_t seen = n.maximal_len >= 64 ? n.minimal : (n.minimal & ((1ULL << n.maximal_len) - 1ULL));
    if ((n.maximal | seen) != n.maximal)
        throw SpecError{GAAST_ERR_INVALID_PROGRAM,
                        "Inferred minimal grade set contains grades not available in maximal grade set"};
    switch (n.opcode) {
    case GAAST_OP_INPUT: return;
    case GAAST_OP_ADD: {
        int l = n.child0, r = n.child1;
        rec_apply_algebra(s, l, bt, limit);
        rec_apply_algebra(s, r, bt, limit);
        return;
    }
    case GAAST_OP_PRODUCT: {
        int l = n.child0, r = n.child1;
        rec_apply_algebra(s, l, bt, limit);
        rec_apply_algebra(s, r, bt, limit);
        GradedNode& p = s.nodes[size_t(idx)];
        auto contribs = iter_contribs(p.minimal, p.sel, s.nodes[size_t(l)].minimal,
                                      s.nodes[size_t(r)].minimal);
        p.n_comp_muls = comp_mul_count(s.n, contribs);
        if (limit != 0 && p.n_comp_muls > limit && p.sel.kind >= 0) {
            p.compact = true;  // the device regenerates (or never needs) the list
            return;
        }
        p.comp_muls.reserve(size_t(p.n_comp_muls));
        for_each_comp_mul(bt, s.metric.data(), contribs,
                          [&](const gaast_comp_mul& m) { p.comp_muls.push_back(m); });
        return;
    }
    default: rec_apply_algebra(s, n.child0, bt, limit); return;
    }
}

}  // namespace

std::unique_ptr<SpecializedAst> specialize(const ExprPtr& e, int n, const double* metric_diag,
                                           uint64_t materialize_limit) {
    if (n < 0 || n > GAAST_MAX_DIM)
        throw SpecError{GAAST_ERR_INVALID_ARGUMENT, "vector-space dimension out of range"};
    auto s = std::make_unique<SpecializedAst>();
    s->n = n;
    s->metric.assign(metric_diag, metric_diag + n);
    s->root_expr = e;
    Builder b{*s, {}, gs_range(0, n)};
    GS root_gs;
    s->root = b.reify_or_reuse(e, &root_gs);  // Expr::reify, expr.rs:62-69
    rec_update_minimal(*s, s->root, root_gs.mask);
    BladeTable bt(n);
    rec_apply_algebra(*s, s->root, bt, materialize_limit);

    // input table: user slots, then embedded constants
    int max_slot = -1;
    for (const GradedNode& g : s->nodes)
        if (g.opcode == GAAST_OP_INPUT && g.input->mv_slot > max_slot) max_slot = g.input->mv_slot;
    s->n_user_inputs = max_slot + 1;
    s->inputs.assign(size_t(s->n_user_inputs) + s->const_nodes.size(), gaast_input_desc{0, 0, 0, nullptr});
    for (const GradedNode& g : s->nodes) {
        if (g.opcode != GAAST_OP_INPUT) continue;
        const ExprNode* in = g.input;
        int slot = in->mv_slot;
        if (slot < 0) slot = s->n_user_inputs + (-(s->const_slot.at(in)) - 1);
        gaast_input_desc& d = s->inputs[size_t(slot)];
        if (in->mv_slot >= 0 && d.grade_mask != 0 &&
            (d.grade_mask != in->mv_mask || d.storage_dim != in->mv_storage_dim))
            throw SpecError{GAAST_ERR_INVALID_PROGRAM, "input slot declared twice with different grades"};
        d.grade_mask = in->mv_mask;
        d.storage_dim = in->mv_storage_dim;
        d.is_const = in->mv_slot < 0;
        d.const_row = in->mv_slot < 0 ? in->mv_const_row.data() : nullptr;
    }
    return s;
}

void fill_program_desc(SpecializedAst& s, int dtype, uint32_t flags, gaast_program_desc* out) {
    s.flat_nodes.clear();
    for (const GradedNode& g : s.nodes) {
        gaast_node_desc d{};
        d.opcode = g.opcode;
        d.child0 = g.child0;
        d.child1 = g.child1;
        d.minimal_grade_mask = g.minimal;
        d.vec_space_dim = g.vec_space_dim;
        d.input_slot = -1;
        d.product_kind = GAAST_PROD_EXPLICIT;
        if (g.opcode == GAAST_OP_INPUT) {
            int slot = g.input->mv_slot;
            if (slot < 0) slot = s.n_user_inputs + (-(s.const_slot.at(g.input)) - 1);
            d.input_slot = slot;
        }
        if (g.opcode == GAAST_OP_PRODUCT) {
            d.n_comp_muls = g.n_comp_muls;
            d.product_kind = g.sel.kind >= 0 ? g.sel.kind : GAAST_PROD_EXPLICIT;
            d.comp_muls = g.compact ? nullptr : g.comp_muls.data();
        }
        s.flat_nodes.push_back(d);
    }
    out->vec_space_dim = s.n;
    out->metric_diag = s.metric.data();
    out->dtype = dtype;
    out->n_nodes = int(s.flat_nodes.size());
    out->nodes = s.flat_nodes.data();
    out->root = s.root;
    out->n_inputs = int(s.inputs.size());
    out->inputs = s.inputs.data();
    out->flags = flags;
}

}  // namespace gaast
