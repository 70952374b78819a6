// Host-side mirror of gaast's phases 1-3 (reference: src/ast/expr.rs, src/ast/base_types.rs,
// src/ast/specialize.rs).  Host-only bookkeeping: builds the annotated DAG that phase 4
// consumes and flattens it into a gaast_program_desc.  Independent of oracle/.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../common/algebra.hpp"
#include "../common/grade_set.hpp"
#include "gaast_expr.h"
#include "gaast_hip.h"

namespace gaast {

// ---- phase 1: deferred expression DAG (Expr, expr.rs:29-44) ---------------------------------
// The reference stores a closure per Expr and identifies nodes by the closure's Rc pointer
// (expr.rs:74-76); here the node object itself plays that role: identity == address, clones
// (shared_ptr copies) share it.
struct ExprNode {
    enum Kind { MV, ADD, NEG, PRODUCT, REV, GINVOL, EXP, LOG, GSELECT, SINV, WRAP_SQRT, WRAP_VINV };
    Kind kind;
    std::shared_ptr<ExprNode> a, b;
    Selection sel;            // PRODUCT
    bool gsel_single = true;  // GSELECT: g(k) vs fixed mask
    int64_t gsel_k = 0;
    uint64_t gsel_mask = 0;
    // MV: what T::grade_set() reports and how its slices are sized
    uint64_t mv_mask = 0;
    int mv_storage_dim = 0;
    int mv_slot = -1;                 // >= 0: bound at evaluation time
    std::vector<double> mv_const_row; // slot < 0: embedded value
};
using ExprPtr = std::shared_ptr<ExprNode>;

ExprPtr make_input(int slot, uint64_t mask, int storage_dim);
ExprPtr make_const(uint64_t mask, int storage_dim, const double* row, size_t len);
ExprPtr make_from_f64(double x);
ExprPtr make_basis_vector(int dim, int i);
ExprPtr make_product(ExprPtr l, ExprPtr r, Selection sel);
ExprPtr make_binary(ExprNode::Kind k, ExprPtr l, ExprPtr r);
ExprPtr make_unary(ExprNode::Kind k, ExprPtr e);
ExprPtr make_g(ExprPtr e, int64_t k);
ExprPtr make_gselect(ExprPtr e, uint64_t mask);
ExprPtr make_sub(ExprPtr l, ExprPtr r);
ExprPtr make_div_scalar(ExprPtr e, double s);
ExprPtr make_pow(ExprPtr e, ExprPtr p);
ExprPtr make_conj(ExprPtr e);
ExprPtr make_scal(ExprPtr e, ExprPtr rhs);
ExprPtr make_norm_sq(ExprPtr e);

// ---- phases 2-3: GradedNode (base_types.rs:105-121) ---------------------------------------------
struct GradedNode {
    const ExprNode* id = nullptr;  // NodeId
    uint64_t maximal = 0, minimal = 0;
    int maximal_len = 0;                   // length of the reference's BitVec (see GS in expr.cpp)
    int vec_space_dim = 0;
    int opcode = GAAST_OP_INPUT;
    int child0 = -1, child1 = -1;
    Selection sel;                         // Product.grades_to_produce
    std::vector<gaast_comp_mul> comp_muls; // Product.individual_comp_muls (when materialised)
    uint64_t n_comp_muls = 0;              // its length either way
    bool compact = false;                  // list not materialised (dense product)
    int num_uses = 0;
    bool is_ready = false;
    const ExprNode* input = nullptr;       // GradedObj
};

struct SpecError {
    int status;
    std::string msg;
};

struct SpecializedAst {
    int n = 0;
    std::vector<double> metric;
    std::vector<GradedNode> nodes;  // post-order; root last
    int root = -1;
    ExprPtr root_expr;
    std::vector<ExprPtr> temps;     // expressions created inside `wrap`
    // input table: user slots first, embedded constants after
    int n_user_inputs = 0;
    std::vector<gaast_input_desc> inputs;
    std::vector<const ExprNode*> const_nodes;
    std::unordered_map<const ExprNode*, int> const_slot;
    // flattened program storage
    std::vector<gaast_node_desc> flat_nodes;
};

// Expr::specialize (specialize.rs:36-50). Throws SpecError where the reference panics.
std::unique_ptr<SpecializedAst> specialize(const ExprPtr& e, int n, const double* metric_diag,
                                           uint64_t materialize_limit);
void fill_program_desc(SpecializedAst& s, int dtype, uint32_t flags, gaast_program_desc* out);

}  // namespace gaast
