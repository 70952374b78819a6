// C ABI of the host-side mirror (include/gaast_expr.h).
#include <atomic>
#include <cstring>
#include <memory>
#include <string>

#include "../common/algebra.hpp"
#include "../common/grade_set.hpp"
#include "expr.hpp"

using namespace gaast;

struct gaast_expr_s {
    std::atomic<int> rc{1};
    ExprPtr node;
};

struct gaast_spec_s {
    std::unique_ptr<SpecializedAst> ast;
};

static thread_local std::string g_err;
static gaast_expr_t wrap(ExprPtr p) {
    auto* h = new gaast_expr_s;
    h->node = std::move(p);
    return h;
}
static Selection builtin(int kind) { return Selection{kind, nullptr, nullptr}; }

extern "C" {

const char* gaast_expr_last_error(void) { return g_err.c_str(); }

uint64_t gaast_gs_single(int64_t k) { return gs_single(k); }
uint64_t gaast_gs_range(int x, int y) { return gs_range(x, y); }
uint64_t gaast_gs_mul(uint64_t a, uint64_t b) { return gs_mul(a, b); }
uint64_t gaast_gs_select(int kind, int64_t k1, int64_t k2) { return gs_select(kind, k1, k2); }
void gaast_gs_parts_contributing_to_product(uint64_t self, int kind, uint64_t left, uint64_t right,
                                            uint64_t* out_left, uint64_t* out_right) {
    parts_contributing(self, builtin(kind), left, right, out_left, out_right);
}

uint64_t gaast_n_choose_k(uint64_t n, uint64_t k) { return n_choose_k(n, k); }

uint64_t gaast_component_to_blade(int n, int grade, uint64_t index) {
    // colex unranking without tables: peel the largest position p with C(p, k) <= index
    uint64_t blade = 0;
    int k = grade;
    for (int p = n - 1; p >= 0 && k > 0; --p) {
        uint64_t z = n_choose_k(uint64_t(p), uint64_t(k));
        if (index >= z) {
            blade |= 1ULL << p;
            index -= z;
            --k;
        }
    }
    return blade;
}

uint64_t gaast_blade_to_component(int n, uint64_t blade, int* grade) {
    (void)n;
    int k = 0;
    uint64_t idx = 0;
    for (int p = 0; p < 64; ++p)
        if ((blade >> p) & 1ULL) idx += n_choose_k(uint64_t(p), uint64_t(++k));
    if (grade) *grade = k;
    return idx;
}

double gaast_blades_gp(int n, const double* metric_diag, uint64_t b1, uint64_t b2, uint64_t* res) {
    if (res) *res = b1 ^ b2;
    return blades_gp_coeff(n, metric_diag, b1, b2);
}

gaast_expr_t gaast_expr_retain(gaast_expr_t e) {
    if (e) e->rc.fetch_add(1);
    return e;
}
void gaast_expr_release(gaast_expr_t e) {
    if (e && e->rc.fetch_sub(1) == 1) delete e;
}

gaast_expr_t gaast_expr_input(int slot, uint64_t grade_mask, int storage_dim) {
    if (slot < 0 || slot >= GAAST_MAX_INPUTS) {
        g_err = "input slot out of range";
        return nullptr;
    }
    return wrap(make_input(slot, grade_mask, storage_dim));
}
gaast_expr_t gaast_expr_const(uint64_t grade_mask, int storage_dim, const double* row, size_t row_len) {
    if (int64_t(row_len) != row_len_of(storage_dim, grade_mask)) {
        g_err = "constant row length does not match its grade mask";
        return nullptr;
    }
    return wrap(make_const(grade_mask, storage_dim, row, row_len));
}
gaast_expr_t gaast_expr_from_f64(double x) { return wrap(make_from_f64(x)); }
gaast_expr_t gaast_expr_basis_vector(int dim, int i) {
    if (i < 0 || i >= dim) {
        g_err = "basis vector index out of range";
        return nullptr;
    }
    return wrap(make_basis_vector(dim, i));
}
gaast_expr_t gaast_expr_product(gaast_expr_t l, gaast_expr_t r, int kind) {
    if (kind < 0 || kind > GAAST_PROD_RCONTRACT) {
        g_err = "unknown product kind";
        return nullptr;
    }
    return wrap(make_product(l->node, r->node, builtin(kind)));
}
gaast_expr_t gaast_expr_product_custom(gaast_expr_t l, gaast_expr_t r, gaast_select_fn f, void* user) {
    return wrap(make_product(l->node, r->node, Selection{GAAST_PROD_EXPLICIT, f, user}));
}
gaast_expr_t gaast_expr_add(gaast_expr_t l, gaast_expr_t r) { return wrap(make_binary(ExprNode::ADD, l->node, r->node)); }
gaast_expr_t gaast_expr_neg(gaast_expr_t e) { return wrap(make_unary(ExprNode::NEG, e->node)); }
gaast_expr_t gaast_expr_sub(gaast_expr_t l, gaast_expr_t r) { return wrap(make_sub(l->node, r->node)); }
gaast_expr_t gaast_expr_div_scalar(gaast_expr_t e, double s) { return wrap(make_div_scalar(e->node, s)); }
gaast_expr_t gaast_expr_rev(gaast_expr_t e) { return wrap(make_unary(ExprNode::REV, e->node)); }
gaast_expr_t gaast_expr_ginvol(gaast_expr_t e) { return wrap(make_unary(ExprNode::GINVOL, e->node)); }
gaast_expr_t gaast_expr_exp(gaast_expr_t e) { return wrap(make_unary(ExprNode::EXP, e->node)); }
gaast_expr_t gaast_expr_log(gaast_expr_t e) { return wrap(make_unary(ExprNode::LOG, e->node)); }
gaast_expr_t gaast_expr_pow(gaast_expr_t e, gaast_expr_t p) { return wrap(make_pow(e->node, p->node)); }
gaast_expr_t gaast_expr_sqrt(gaast_expr_t e) { return wrap(make_unary(ExprNode::WRAP_SQRT, e->node)); }
gaast_expr_t gaast_expr_g(gaast_expr_t e, int64_t k) { return wrap(make_g(e->node, k)); }
gaast_expr_t gaast_expr_gselect_mask(gaast_expr_t e, uint64_t wanted) { return wrap(make_gselect(e->node, wanted)); }
gaast_expr_t gaast_expr_conj(gaast_expr_t e) { return wrap(make_conj(e->node)); }
gaast_expr_t gaast_expr_scal(gaast_expr_t e, gaast_expr_t rhs) { return wrap(make_scal(e->node, rhs->node)); }
gaast_expr_t gaast_expr_norm_sq(gaast_expr_t e) { return wrap(make_norm_sq(e->node)); }
gaast_expr_t gaast_expr_sinv(gaast_expr_t e) { return wrap(make_unary(ExprNode::SINV, e->node)); }
gaast_expr_t gaast_expr_vinv(gaast_expr_t e) { return wrap(make_unary(ExprNode::WRAP_VINV, e->node)); }

gaast_spec_t gaast_expr_specialize(gaast_expr_t e, int n, const double* metric_diag,
                                   uint64_t materialize_limit) {
    try {
        auto s = std::make_unique<gaast_spec_s>();   // released if specialize() reports a reference panic by throwing
        s->ast = specialize(e->node, n, metric_diag, materialize_limit);
        return s.release();
    } catch (const SpecError& err) {
        g_err = err.msg;
        return nullptr;
    } catch (const std::exception& ex) {
        g_err = ex.what();
        return nullptr;
    }
}
void gaast_spec_free(gaast_spec_t s) { delete s; }
int gaast_spec_num_nodes(gaast_spec_t s) { return int(s->ast->nodes.size()); }
int gaast_spec_root(gaast_spec_t s) { return s->ast->root; }

int gaast_spec_node(gaast_spec_t s, int idx, gaast_spec_node_info* out) {
    if (idx < 0 || idx >= int(s->ast->nodes.size())) return GAAST_ERR_INVALID_ARGUMENT;
    const GradedNode& g = s->ast->nodes[size_t(idx)];
    out->opcode = g.opcode;
    out->child0 = g.child0;
    out->child1 = g.child1;
    out->maximal_grade_mask = g.maximal;
    out->minimal_grade_mask = g.minimal;
    out->vec_space_dim = g.vec_space_dim;
    out->num_uses = g.num_uses;
    out->input_slot = -1;
    if (g.opcode == GAAST_OP_INPUT)
        out->input_slot = g.input->mv_slot >= 0 ? g.input->mv_slot : s->ast->const_slot.at(g.input);
    out->product_kind = g.opcode == GAAST_OP_PRODUCT ? (g.sel.kind >= 0 ? g.sel.kind : GAAST_PROD_EXPLICIT)
                                                      : GAAST_PROD_EXPLICIT;
    out->n_comp_muls = g.n_comp_muls;
    return GAAST_OK;
}

const gaast_comp_mul* gaast_spec_comp_muls(gaast_spec_t s, int idx) {
    if (idx < 0 || idx >= int(s->ast->nodes.size())) return nullptr;
    const GradedNode& g = s->ast->nodes[size_t(idx)];
    return (g.opcode == GAAST_OP_PRODUCT && !g.compact) ? g.comp_muls.data() : nullptr;
}

int gaast_spec_program_desc(gaast_spec_t s, int dtype, uint32_t flags, gaast_program_desc* out) {
    fill_program_desc(*s->ast, dtype, flags, out);
    return GAAST_OK;
}
int gaast_spec_num_inputs(gaast_spec_t s) { return int(s->ast->inputs.size()); }
int gaast_spec_num_user_inputs(gaast_spec_t s) { return s->ast->n_user_inputs; }

}  // extern "C"
