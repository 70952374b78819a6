/*
 * gaast_oracle.h -- CPU oracle for gaast's phase-4 evaluation (TEST INFRASTRUCTURE ONLY).
 *
 * This is a plain-C, single-threaded, literal restatement of the reference
 * library YPares/gaast (Rust), sufficient to build an expression, specialize
 * it against a diagonal metric (phases 1-3) and evaluate it (phase 4).
 * Every function cites the reference file:line it follows.
 *
 * It is NOT part of the product.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it, and only as the checker / reported
 * baseline.  The product (gaast_amd/, libgaast_hip.so) never links or calls it.
 *
 * Parity status: PINNED by the reference's own known-answer tests
 * (src/eval.rs:134-163, src/algebra.rs:274-300, src/grade_set.rs:338-373,
 * src/graded.rs:230-232), transcribed in tests/golden/ref_kat.json and checked
 * by tests/test_oracle_kat.py.  The Rust reference itself cannot be built in
 * this pipeline (no rustc/cargo, crates not vendored), so there is no
 * oracle/_ref build.  Third-party semantics the restatement assumes (crate
 * bitvec ^1.0.1, unpinned upstream): BitVec::shift_left moves bits toward
 * index 0; `a & b` on unequal lengths keeps a's length and treats b as
 * zero-extended; `a | b` keeps a's length.
 */
#ifndef GAAST_ORACLE_H
#define GAAST_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes: the reference panics; the oracle reports which panic ---- */
enum {
    OG_OK = 0,
    OG_PANIC_MISSING_GRADE = 1, /* graded.rs:188,193 HashMap index / unwrap on absent grade */
    OG_PANIC_TODO = 2,          /* eval.rs:112-113 todo!() for Exponential / Logarithm */
    OG_PANIC_ASSERT = 3,        /* grade_set.rs:182-195, specialize.rs:104-117 asserts */
    OG_PANIC_OVERFLOW = 4,      /* eval.rs:90 `k - 1` on usize with k == 0 (debug builds only) */
    OG_PANIC_DOMAIN = 5,        /* EXTENSION (no reference behaviour): exp / log of a k-vector whose square is not scalar */
    OG_BAD_ARG = 5
};

/* ---- GradeSet (grade_set.rs:24-27): a BitVec; bit k set <=> grade k present ---- */
typedef struct og_gradeset {
    uint64_t bits;
    int len; /* BitVec length; only GradeSet::mul's loop bounds depend on it */
} og_gradeset;

typedef og_gradeset (*og_select_fn)(int64_t k1, int64_t k2, void *user);

og_gradeset og_gs_empty(void);                               /* grade_set.rs:52-55 */
og_gradeset og_gs_single(int64_t k);                         /* grade_set.rs:65-71 */
og_gradeset og_gs_range(int x, int y);                       /* grade_set.rs:74-80 */
og_gradeset og_gs_intersection(og_gradeset a, og_gradeset b);/* grade_set.rs:85-91 */
og_gradeset og_gs_add(og_gradeset a, og_gradeset b);         /* grade_set.rs:287-293 */
og_gradeset og_gs_mul(og_gradeset a, og_gradeset b);         /* grade_set.rs:305-327 */
og_gradeset og_gs_add_grade(og_gradeset a, int k);           /* grade_set.rs:159-165 */
og_gradeset og_gs_rm_grade(og_gradeset a, int k);            /* grade_set.rs:168-173 */
int og_gs_eq(og_gradeset a, og_gradeset b);                  /* grade_set.rs:35-42 */
int og_gs_is_empty(og_gradeset a);                           /* grade_set.rs:124-126 */
int og_gs_is_single(og_gradeset a);                          /* grade_set.rs:129-138 */
int og_gs_contains(og_gradeset a, int k);                    /* grade_set.rs:141-146 */
int og_gs_includes(og_gradeset a, og_gradeset other);        /* grade_set.rs:149-151 */
int og_gs_is_just(og_gradeset a, int k);                     /* grade_set.rs:154-156 */
int og_gs_iter(og_gradeset a, int *out, int cap);            /* grade_set.rs:94-96 (ascending) */
/* grade_set.rs:239-252; sel_kind is one of OG_SEL_* below */
void og_gs_parts_contributing_to_product(og_gradeset self, int sel_kind, og_gradeset left,
                                         og_gradeset right, og_gradeset *out_left,
                                         og_gradeset *out_right);

/* ---- algebra.rs ---- */
uint64_t og_n_choose_k(uint64_t n, uint64_t k);                          /* algebra.rs:252-254 */
uint64_t og_index_to_bitfield_permut(int n, int k, uint64_t i);          /* algebra.rs:221-232 */
uint64_t og_bitfield_permut_to_index(int n, int k, uint64_t v);          /* algebra.rs:236-246 */
double og_canonical_reordering_sign(uint64_t b1, uint64_t b2);           /* algebra.rs:199-209 */

typedef struct og_algebra {
    int dim;          /* vec_space_dim */
    int is_euclid;    /* 1: OrthoEuclidN(dim) (algebra.rs:173-192); 0: [f64; D] (algebra.rs:148-165) */
    double diag[64];  /* squares of the base vectors when !is_euclid */
} og_algebra;

/* algebra.rs:73-83: (b1 ^ b2, sign * prod of metric over shared vectors) */
double og_ortho_basis_blades_gp(const og_algebra *alg, uint64_t b1, uint64_t b2, uint64_t *res);

/* ---- graded.rs:173-202 GradeMapMV: one dense array per grade ---- */
typedef struct og_mv og_mv;
og_mv *og_mv_new(void);
void og_mv_free(og_mv *m);
/* insert/replace grade k with a copy of vals[0..len) */
int og_mv_set_grade(og_mv *m, int k, const double *vals, size_t len);
uint64_t og_mv_grade_mask(const og_mv *m);
size_t og_mv_grade_len(const og_mv *m, int k);
/* pointer to the stored slice (NULL when absent) */
double *og_mv_grade_ptr(og_mv *m, int k);
og_mv *og_mv_init_null(int dim, og_gradeset gs);                          /* graded.rs:195-201 */

/* ---- Expr (ast/expr.rs): reference-counted expression DAG; identity == pointer ---- */
typedef struct og_expr og_expr;
enum { OG_SEL_GEOMETRIC = 0, OG_SEL_OUTER = 1, OG_SEL_INNER = 2, OG_SEL_LCONTRACT = 3,
       OG_SEL_RCONTRACT = 4 };                                            /* expr.rs:180-197 */

og_expr *og_expr_retain(og_expr *e);   /* Expr::clone (expr.rs:47-53) */
void og_expr_release(og_expr *e);
/* mv(x) (expr.rs:162-164). The expression BORROWS x (T = &GradeMapMV): the caller may
 * rewrite x's values between evaluations, but not its grade set. */
og_expr *og_expr_mv(og_mv *x);
og_expr *og_expr_from_f64(double x);                                      /* expr.rs:231-240 */
og_expr *og_expr_basis_vector(int dim, int i);                            /* expr.rs:148-157 */
og_expr *og_expr_product(og_expr *l, og_expr *r, int sel_kind);           /* expr.rs:123-144,166-197 */
og_expr *og_expr_product_custom(og_expr *l, og_expr *r, og_select_fn f, void *user);
og_expr *og_expr_add(og_expr *l, og_expr *r);                             /* expr.rs:200-210 */
og_expr *og_expr_neg(og_expr *e);                                         /* expr.rs:213-221 */
og_expr *og_expr_sub(og_expr *l, og_expr *r);                             /* expr.rs:224-229 */
og_expr *og_expr_div_scalar(og_expr *e, double s);                        /* expr.rs:265-270 */
og_expr *og_expr_rev(og_expr *e);                                         /* expr.rs:292 */
og_expr *og_expr_ginvol(og_expr *e);                                      /* expr.rs:293 */
og_expr *og_expr_exp(og_expr *e);                                         /* expr.rs:294 */
og_expr *og_expr_log(og_expr *e);                                         /* expr.rs:295 */
og_expr *og_expr_pow(og_expr *e, og_expr *p);                             /* expr.rs:300-302 */
og_expr *og_expr_sqrt(og_expr *e);                                        /* expr.rs:305-319 */
og_expr *og_expr_g(og_expr *e, int64_t k);                                /* expr.rs:322-324 */
og_expr *og_expr_gselect_mask(og_expr *e, uint64_t wanted_mask);          /* expr.rs:327-335 */
og_expr *og_expr_conj(og_expr *e);                                        /* expr.rs:338-340 */
og_expr *og_expr_scal(og_expr *e, og_expr *rhs);                          /* expr.rs:343-345 */
og_expr *og_expr_norm_sq(og_expr *e);                                     /* expr.rs:348-350 */
og_expr *og_expr_sinv(og_expr *e);                                        /* expr.rs:353-358 */
og_expr *og_expr_vinv(og_expr *e);                                        /* expr.rs:363-371 */

/* ---- SpecializedAst (ast/specialize.rs) ---- */
typedef struct og_spec og_spec;
enum { OG_N_GRADED_OBJ = 0, OG_N_ADDITION, OG_N_PRODUCT, OG_N_NEGATION, OG_N_EXPONENTIAL,
       OG_N_LOGARITHM, OG_N_GRADE_PROJECTION, OG_N_REVERSE, OG_N_GRADE_INVOLUTION,
       OG_N_SCALAR_INVERSION, OG_N_SCALAR_SQRT };                         /* base_types.rs:8-30,84-88 */

/* base_types.rs:45-55 + algebra.rs:87-91: 6 x usize + f64 = 56 bytes, AoS as in the reference */
typedef struct og_comp_mul {
    size_t left_grade, left_index;
    size_t right_grade, right_index;
    size_t result_grade, result_index;
    double coeff;
} og_comp_mul;

/* Expr::specialize (specialize.rs:36-50). Returns NULL and sets *status on a reference panic. */
og_spec *og_specialize(og_expr *e, const og_algebra *alg, int *status);
void og_spec_free(og_spec *s);
const char *og_last_panic(void);

int og_spec_num_nodes(const og_spec *s);
int og_spec_root(const og_spec *s);  /* index into the node list */
typedef struct og_node_info {
    int kind;            /* OG_N_* */
    int child0, child1;  /* node indices, -1 when absent */
    uint64_t maximal;    /* bits of maximal_grade_set */
    uint64_t minimal;    /* bits of minimal_grade_set */
    int vec_space_dim;
    int num_uses;
    size_t n_comp_muls;  /* Product only */
    og_mv *input;        /* GradedObj only */
} og_node_info;
int og_spec_node(const og_spec *s, int idx, og_node_info *out);
const og_comp_mul *og_spec_comp_muls(const og_spec *s, int idx);

/* ---- SpecializedAst::eval::<GradeMapMV>() (eval.rs:12-115) ---- */
enum { OG_EVAL_RELEASE = 0, /* k*(k-1)/2 wraps for k == 0: grade 0 untouched */
       OG_EVAL_DEBUG = 1,   /* overflow checks on: Reverse over grade 0 panics */
       /* EXTENSION, bit flag, "no reference behaviour, parity unpinned": evaluate Exponential / Logarithm (todo!() in
        * eval.rs:112-113) with the semantics the reference's grade rules imply (grade_set.rs:181-197), see the
        * comment above ext_exp_log() in gaast_oracle.c.  Without the flag these arms panic like the reference. */
       OG_EVAL_EXT_EXPLOG = 2,
       /* F32 MODE, bit flag -- an extension of the BUILD, not of the reference (which is f64-only, graded.rs:46): the same
        * statements in the same order with EVERY operand read and every operation's result rounded to IEEE binary32, so that
        * the f32 exact kernels (an extension too, SURVEY 8b "data-type note") are checked bit for bit like the f64 ones.
        * Values travel in the double slabs; every one of them is exactly a float.  Pinned by the reference's four eval
        * known-answer tests (exactly representable in f32).  exp / log keep their f64 statements (not combined with EXT). */
       OG_EVAL_F32 = 4 };
/* tolerance of the extension's domain check: a k-vector B is accepted when |<B^2>_{not 0}|^2 <= 2^-40 (sum B_i^2)^2 */
#define OG_EXPLOG_DOMAIN_TOL2 9.094947017729282e-13
int og_eval(const og_spec *s, int mode, og_mv **out);

/*
 * Batched driver used by parity tests and bench.py's cpu_baseline: runs og_eval once per
 * item, re-binding the borrowed inputs each time (README.md:80-83 lists re-binding as
 * roadmap; with T = &GradeMapMV it is what a caller would do today).
 *   inputs[j]           : the og_mv bound into the expression for slot j
 *   in_data[j]          : item-major rows, row i holds the grades of inputs[j] concatenated
 *                         in ascending grade order; NULL = leave inputs[j] untouched (shared)
 *   out_data            : item-major rows with the grades of the root's minimal set
 *   out_row_len         : number of doubles per output row (checked)
 */
int og_eval_batch(const og_spec *s, int mode, og_mv **inputs, const double **in_data, int n_inputs,
                  int64_t batch, double *out_data, size_t out_row_len);

/* eval.rs:77-83 inner loop alone over a prebuilt table, on raw arrays: the timed cpu baseline.
 * left/right/res are arrays of per-grade pointers (index = grade). */
void og_product_loop(const og_comp_mul *muls, size_t n, double *const *left, double *const *right,
                     double **res);

/* ---- cpu-packed baselines (`cpu-packed-1t`, `cpu-packed-allcores`) ----
 * The same loop as eval.rs:77-83 (`res[o] += left[l] * right[r] * coeff`, sequential, the reference's order) over 16-byte
 * PACKED entries on flat graded rows instead of the reference's 56-byte {6 x usize, f64}: shows how much of the reference's
 * time is table traffic, and -- with the batch split over threads -- the fair CPU ceiling of batched evaluation.  Bench
 * infrastructure only (never a checker: the literal og_eval above is). */
typedef struct og_packed_mul {
    uint32_t left, right, out; /* offsets in the operands' / result's graded rows (grades ascending, concatenated) */
    float coeff;               /* +-1 / 0 / metric products that are exact in f32; og_pack_root_product REFUSES (returns 0) a list with any other coefficient */
} og_packed_mul;
/* packs the comp-mul list of the ROOT product of `s` (both operands GradedObj leaves); returns the entry count, 0 on
 * failure; *out is malloc'ed (free with og_packed_free) */
size_t og_pack_root_product(const og_spec *s, og_packed_mul **out, size_t *left_len, size_t *right_len, size_t *out_len);
void og_packed_free(og_packed_mul *p);
/* `batch` items (rows item-major), `threads` worker threads over contiguous item ranges; returns wall seconds */
double og_packed_eval_batch(const og_packed_mul *muls, size_t n, const double *left, size_t left_len, const double *right,
                            size_t right_len, double *out, size_t out_len, int64_t batch, int threads);

#ifdef __cplusplus
}
#endif
#endif
