/*
 * gaast_expr.h -- host-side mirror of gaast's phases 1-3 for hosts that are not Rust.
 *
 * A Rust host keeps the reference's own `Expr` / `specialize` (src/ast/expr.rs,
 * src/ast/specialize.rs) and only needs gaast_hip.h.  This header gives C, C++ and Python
 * hosts the same operator surface (same names, argument meaning and failure behaviour) so
 * that programs for gaast_hip_program_create() can be built and tests can read like the
 * reference's own.  It is host-only bookkeeping: no component value is touched here and no
 * GPU is needed.
 *
 * Every function cites the reference item it mirrors.  Failures that are panics upstream
 * return NULL / a non-zero status and leave a message in gaast_expr_last_error().
 */
#ifndef GAAST_EXPR_H
#define GAAST_EXPR_H

#include <stddef.h>
#include <stdint.h>

#include "gaast_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gaast_expr_s *gaast_expr_t;        /* Expr<'a, T>        src/ast/expr.rs:29-44 */
typedef struct gaast_spec_s *gaast_spec_t;        /* SpecializedAst<T>  src/ast/specialize.rs:10-13 */

/* grades_to_produce closure of Expr::product (expr.rs:123-127): returns the grade mask wanted
 * out of the product of a k1-vector and a k2-vector. */
typedef uint64_t (*gaast_select_fn)(int64_t k1, int64_t k2, void *user);

const char *gaast_expr_last_error(void);

/* ---- GradeSet algebra on 64-bit masks (src/grade_set.rs) -------------------------------- */
uint64_t gaast_gs_single(int64_t k);                      /* grade_set.rs:65-71 (negative -> empty) */
uint64_t gaast_gs_range(int x, int y);                    /* grade_set.rs:74-80 */
uint64_t gaast_gs_mul(uint64_t a, uint64_t b);            /* grade_set.rs:305-327 (no dimension cap) */
uint64_t gaast_gs_select(int kind, int64_t k1, int64_t k2); /* the five closures of expr.rs:180-197 */
/* grade_set.rs:239-252 */
void gaast_gs_parts_contributing_to_product(uint64_t self, int kind, uint64_t left, uint64_t right,
                                            uint64_t *out_left, uint64_t *out_right);

/* ---- Algebra (src/algebra.rs) -------------------------------------------------------------- */
uint64_t gaast_n_choose_k(uint64_t n, uint64_t k);                       /* algebra.rs:252-254 */
uint64_t gaast_component_to_blade(int n, int grade, uint64_t index);     /* algebra.rs:31-37 */
uint64_t gaast_blade_to_component(int n, uint64_t blade, int *grade);    /* algebra.rs:41-45 */
/* ortho_basis_blades_gp, algebra.rs:73-83: returns the coefficient, *res = b1 ^ b2 */
double gaast_blades_gp(int n, const double *metric_diag, uint64_t b1, uint64_t b2, uint64_t *res);

/* ---- Expr construction (phase 1) -------------------------------------------------------------- */
gaast_expr_t gaast_expr_retain(gaast_expr_t e);  /* Expr::clone, expr.rs:47-53: same node identity */
void gaast_expr_release(gaast_expr_t e);
/* mv(x), expr.rs:162-164, for an input that is bound at evaluation time: `slot` names it,
 * grade_mask / storage_dim are what T::grade_set() and its slice lengths would report. */
gaast_expr_t gaast_expr_input(int slot, uint64_t grade_mask, int storage_dim);
/* mv(x) for a value fixed at build time (shared by all batch items); row = grades ascending. */
gaast_expr_t gaast_expr_const(uint64_t grade_mask, int storage_dim, const double *row, size_t row_len);
gaast_expr_t gaast_expr_from_f64(double x);                       /* expr.rs:231-240 */
gaast_expr_t gaast_expr_basis_vector(int dim, int i);             /* expr.rs:148-157 */
gaast_expr_t gaast_expr_product(gaast_expr_t l, gaast_expr_t r, int kind); /* expr.rs:166-197 */
gaast_expr_t gaast_expr_product_custom(gaast_expr_t l, gaast_expr_t r, gaast_select_fn f,
                                       void *user);                /* expr.rs:123-144 */
gaast_expr_t gaast_expr_add(gaast_expr_t l, gaast_expr_t r);      /* expr.rs:200-210 */
gaast_expr_t gaast_expr_neg(gaast_expr_t e);                      /* expr.rs:213-221 */
gaast_expr_t gaast_expr_sub(gaast_expr_t l, gaast_expr_t r);      /* expr.rs:224-229 */
gaast_expr_t gaast_expr_div_scalar(gaast_expr_t e, double s);     /* expr.rs:265-270 */
gaast_expr_t gaast_expr_rev(gaast_expr_t e);                      /* expr.rs:292 */
gaast_expr_t gaast_expr_ginvol(gaast_expr_t e);                   /* expr.rs:293 */
gaast_expr_t gaast_expr_exp(gaast_expr_t e);                      /* expr.rs:294 */
gaast_expr_t gaast_expr_log(gaast_expr_t e);                      /* expr.rs:295 */
gaast_expr_t gaast_expr_pow(gaast_expr_t e, gaast_expr_t p);      /* expr.rs:300-302 */
gaast_expr_t gaast_expr_sqrt(gaast_expr_t e);                     /* expr.rs:305-319 */
gaast_expr_t gaast_expr_g(gaast_expr_t e, int64_t k);             /* expr.rs:322-324 */
gaast_expr_t gaast_expr_gselect_mask(gaast_expr_t e, uint64_t wanted); /* expr.rs:327-335 */
gaast_expr_t gaast_expr_conj(gaast_expr_t e);                     /* expr.rs:338-340 */
gaast_expr_t gaast_expr_scal(gaast_expr_t e, gaast_expr_t rhs);   /* expr.rs:343-345 */
gaast_expr_t gaast_expr_norm_sq(gaast_expr_t e);                  /* expr.rs:348-350 */
gaast_expr_t gaast_expr_sinv(gaast_expr_t e);                     /* expr.rs:353-358 */
gaast_expr_t gaast_expr_vinv(gaast_expr_t e);                     /* expr.rs:363-371 */

/* ---- reify + specialize (phases 2-3) ---------------------------------------------------------- */
/* Expr::specialize(&alg), specialize.rs:36-50, with alg = the diagonal metric `metric_diag[n]`
 * ([f64; D], algebra.rs:148-165; all ones = OrthoEuclidN(n), algebra.rs:173-192).
 * materialize_limit: PRODUCT nodes whose comp-mul list would exceed this many entries keep a
 * compact descriptor instead of the explicit list (0 = always explicit). */
gaast_spec_t gaast_expr_specialize(gaast_expr_t e, int n, const double *metric_diag,
                                   uint64_t materialize_limit);
void gaast_spec_free(gaast_spec_t s);

/* public read API of SpecializedAst / GradedNode (specialize.rs:17-24, base_types.rs:124-146);
 * nodes are numbered in post-order, the root is the last one. */
int gaast_spec_num_nodes(gaast_spec_t s);
int gaast_spec_root(gaast_spec_t s);
typedef struct gaast_spec_node_info {
    int32_t opcode;           /* gaast_opcode */
    int32_t child0, child1;
    uint64_t maximal_grade_mask;
    uint64_t minimal_grade_mask;
    int32_t vec_space_dim;
    int32_t num_uses;
    int32_t input_slot;       /* OP_INPUT: slot, or -(1+const index) for embedded constants */
    int32_t product_kind;
    uint64_t n_comp_muls;     /* length the explicit list has / would have */
} gaast_spec_node_info;
int gaast_spec_node(gaast_spec_t s, int idx, gaast_spec_node_info *out);
/* Product.individual_comp_muls (NULL when the node kept a compact descriptor) */
const gaast_comp_mul *gaast_spec_comp_muls(gaast_spec_t s, int idx);

/* The flat program for gaast_hip_program_create(); pointers stay valid until gaast_spec_free.
 * Embedded constants are appended after the caller's input slots. */
int gaast_spec_program_desc(gaast_spec_t s, int dtype, uint32_t flags, gaast_program_desc *out);
/* ---- program wire format ------------------------------------------------------
 * The reference's SpecializedAst cannot be stored (closures and pointer ids, base_types.rs:60-64,
 * 92-96); the flat program can.  Layout (little endian): "GAASTPRG", u32 version = 1, then the
 * fields of gaast_program_desc in order, every array length-prefixed; comp-mul lists as 32-byte
 * gaast_comp_mul records.  gaast_program_serialize returns the number of bytes needed / written
 * (call with buf = NULL to size); gaast_program_deserialize returns a handle that owns the
 * decoded arrays, or NULL on a malformed image. */
typedef struct gaast_program_image_s *gaast_program_image_t;
size_t gaast_program_serialize(const gaast_program_desc *desc, void *buf, size_t cap);
gaast_program_image_t gaast_program_deserialize(const void *buf, size_t len);
const gaast_program_desc *gaast_program_image_desc(gaast_program_image_t img);
void gaast_program_image_free(gaast_program_image_t img);

/* number of input slots the program expects (caller slots + embedded constants) */
int gaast_spec_num_inputs(gaast_spec_t s);
int gaast_spec_num_user_inputs(gaast_spec_t s);

#ifdef __cplusplus
}
#endif
#endif /* GAAST_EXPR_H */
