/*
 * gaast_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see gaast_oracle.h).
 *
 * Literal, single-threaded C restatement of YPares/gaast phases 1-4.  The code
 * deliberately keeps the reference's algorithms (O(n) binomial walks, shifted
 * BitVec sign loops, 56-byte AoS comp-mul entries, one heap Vec per grade per
 * cached node and eval call) so that (a) results are bit-identical to what the
 * Rust would compute and (b) it is an honest stand-in when timed as the CPU
 * baseline.  Compile with -ffp-contract=off: Rust never fuses a*b*c + d.
 *
 * BitVec convention: bit p of a uint64_t is BitVec index p (index 0 = e1, the
 * "leftmost" bit of algebra.rs:103-131).  BitVec::shift_left(1) moves every
 * bit toward index 0, i.e. `>>= 1` in this representation.
 */
#include "gaast_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define OG_MAX_GRADE 64

static char g_panic[256] = "";
static void set_panic(const char *msg) {
    strncpy(g_panic, msg, sizeof(g_panic) - 1);
    g_panic[sizeof(g_panic) - 1] = 0;
}
const char *og_last_panic(void) { return g_panic; }

static int popcnt64(uint64_t x) { return __builtin_popcountll(x); }
static uint64_t lowmask(int len) { return len >= 64 ? ~0ULL : ((1ULL << len) - 1ULL); }

/* ===================================================================== */
/* GradeSet -- src/grade_set.rs                                           */
/* ===================================================================== */

og_gradeset og_gs_empty(void) { /* grade_set.rs:52-55 */
    og_gradeset g = {0, 0};
    return g;
}

static og_gradeset gs_from_usize(int k) { /* grade_set.rs:57-61 */
    og_gradeset g;
    g.len = k + 1;
    g.bits = 1ULL << k;
    return g;
}

og_gradeset og_gs_single(int64_t k) { /* grade_set.rs:65-71 */
    if (k < 0) return og_gs_empty();
    return gs_from_usize((int)k);
}

og_gradeset og_gs_range(int x, int y) { /* grade_set.rs:74-80 */
    og_gradeset g;
    g.len = y + 1;
    g.bits = 0;
    for (int i = x; i <= y; ++i) g.bits |= 1ULL << i;
    return g;
}

og_gradeset og_gs_intersection(og_gradeset a, og_gradeset b) { /* grade_set.rs:85-91 */
    /* `self.bv & rhs.bv`: result keeps self's length; rhs is treated as zero-extended
     * (bitvec >= 1.0 clears self's bits past rhs.len()).  See header: unpinned 3rd-party rule. */
    og_gradeset g;
    g.len = a.len;
    g.bits = a.bits & (b.bits & lowmask(b.len)) & lowmask(a.len);
    return g;
}

int og_gs_iter(og_gradeset a, int *out, int cap) { /* grade_set.rs:94-96: iter_ones, ascending */
    int n = 0;
    for (int k = 0; k < a.len && k < OG_MAX_GRADE; ++k)
        if ((a.bits >> k) & 1ULL) {
            if (n < cap) out[n] = k;
            ++n;
        }
    return n;
}

int og_gs_is_empty(og_gradeset a) { return (a.bits & lowmask(a.len)) == 0; } /* :124-126 */

int og_gs_is_single(og_gradeset a) { /* grade_set.rs:129-138 */
    return popcnt64(a.bits & lowmask(a.len)) == 1;
}

int og_gs_contains(og_gradeset a, int k) { /* grade_set.rs:141-146 */
    if (k < 0 || k >= a.len) return 0;
    return (int)((a.bits >> k) & 1ULL);
}

static og_gradeset gs_or_keep_left_len(og_gradeset big, og_gradeset small) {
    /* `big | small` with big.len >= small.len */
    og_gradeset g;
    g.len = big.len;
    g.bits = (big.bits | (small.bits & lowmask(small.len))) & lowmask(big.len);
    return g;
}

int og_gs_eq(og_gradeset a, og_gradeset b) { /* grade_set.rs:35-42: equal up to trailing zeroes */
    return (a.bits & lowmask(a.len)) == (b.bits & lowmask(b.len));
}

int og_gs_includes(og_gradeset a, og_gradeset other) { /* grade_set.rs:149-151 */
    /* (self.bv.clone() | other.bv) == self.bv ; `|` keeps self's length, so grades of `other`
     * at positions >= self.len are not seen -- restated literally. */
    uint64_t o = other.bits & lowmask(other.len) & lowmask(a.len);
    uint64_t s = a.bits & lowmask(a.len);
    return (s | o) == s;
}

int og_gs_is_just(og_gradeset a, int k) { /* grade_set.rs:154-156 */
    return og_gs_contains(a, k) && og_gs_is_single(a);
}

og_gradeset og_gs_add_grade(og_gradeset a, int k) { /* grade_set.rs:159-165 */
    if (k >= a.len) a.len = k + 1;
    a.bits |= 1ULL << k;
    return a;
}

og_gradeset og_gs_rm_grade(og_gradeset a, int k) { /* grade_set.rs:168-173 */
    if (k < a.len) a.bits &= ~(1ULL << k);
    return a;
}

static int gs_exp(og_gradeset a, og_gradeset *out) { /* grade_set.rs:181-187 */
    if (!og_gs_is_single(a)) {
        set_panic("exp cannot be used on a multivector, only a k-vector");
        return OG_PANIC_ASSERT;
    }
    *out = og_gs_add(gs_from_usize(0), a);
    return OG_OK;
}

static int gs_log(og_gradeset a, og_gradeset *out) { /* grade_set.rs:190-197 */
    og_gradeset other = og_gs_rm_grade(a, 0);
    if (!og_gs_is_single(other)) {
        set_panic("log can only be used on multivectors of the form <A>_0 + <A>_k");
        return OG_PANIC_ASSERT;
    }
    *out = other;
    return OG_OK;
}

og_gradeset og_gs_add(og_gradeset a, og_gradeset b) { /* grade_set.rs:287-293 (+ sort_by_len :276-285) */
    if (a.len <= b.len) return gs_or_keep_left_len(b, a);
    return gs_or_keep_left_len(a, b);
}

og_gradeset og_gs_mul(og_gradeset a, og_gradeset b) { /* grade_set.rs:305-327, O(N^3) as written */
    og_gradeset small, big;
    if (a.len <= b.len) {
        small = a;
        big = b;
    } else {
        small = b;
        big = a;
    }
    if (small.len == 0) return small;
    og_gradeset res;
    res.len = big.len + small.len - 1;
    res.bits = 0;
    for (int r = 0; r < res.len && r < OG_MAX_GRADE; ++r) {
        for (int i = 0; i < small.len; ++i) {
            for (int j = 0; j < big.len; ++j) {
                int m = abs(i - j);
                if (i + j >= r && m <= r && m % 2 == r % 2) {
                    int x = (int)((res.bits >> r) & 1ULL);
                    x = x || (((small.bits >> i) & 1ULL) && ((big.bits >> j) & 1ULL));
                    if (x) res.bits |= 1ULL << r;
                }
            }
        }
    }
    return res;
}

/* The five built-in `grades_to_produce` closures, expr.rs:180-197 */
static og_gradeset sel_builtin(int64_t k1, int64_t k2, void *user) {
    int kind = (int)(intptr_t)user;
    switch (kind) {
    case OG_SEL_GEOMETRIC: return og_gs_mul(og_gs_single(k1), og_gs_single(k2));
    case OG_SEL_OUTER: return og_gs_single(k1 + k2);
    case OG_SEL_INNER:
        if (k1 == 0 || k2 == 0) return og_gs_empty();
        return og_gs_single(k1 > k2 ? k1 - k2 : k2 - k1);
    case OG_SEL_LCONTRACT: return og_gs_single(k2 - k1);
    case OG_SEL_RCONTRACT: return og_gs_single(k1 - k2);
    default: return og_gs_empty();
    }
}

typedef struct contrib {
    int k_left, k_right;
    og_gradeset contribs;
} contrib;

/* grade_set.rs:221-235 over iter_grade_sets_cp (:268-274): left grades outer, right grades inner,
 * both ascending; keep the pairs whose selected grades intersect self. */
static int gs_iter_contribs(og_gradeset self, og_select_fn f, void *user, og_gradeset left,
                            og_gradeset right, contrib *out, int cap) {
    int lk[OG_MAX_GRADE], rk[OG_MAX_GRADE];
    int nl = og_gs_iter(left, lk, OG_MAX_GRADE), nr = og_gs_iter(right, rk, OG_MAX_GRADE);
    int n = 0;
    for (int a = 0; a < nl; ++a)
        for (int b = 0; b < nr; ++b) {
            og_gradeset c = og_gs_intersection(self, f(lk[a], rk[b], user));
            if (!og_gs_is_empty(c)) {
                if (n < cap) {
                    out[n].k_left = lk[a];
                    out[n].k_right = rk[b];
                    out[n].contribs = c;
                }
                ++n;
            }
        }
    return n;
}

static void gs_parts_contributing(og_gradeset self, og_select_fn f, void *user, og_gradeset left,
                                  og_gradeset right, og_gradeset *ol, og_gradeset *orr) {
    /* grade_set.rs:239-252 */
    static contrib buf[OG_MAX_GRADE * OG_MAX_GRADE];
    int n = gs_iter_contribs(self, f, user, left, right, buf, OG_MAX_GRADE * OG_MAX_GRADE);
    og_gradeset fl = og_gs_empty(), fr = og_gs_empty();
    for (int i = 0; i < n; ++i) {
        fl = og_gs_add_grade(fl, buf[i].k_left);
        fr = og_gs_add_grade(fr, buf[i].k_right);
    }
    *ol = fl;
    *orr = fr;
}

void og_gs_parts_contributing_to_product(og_gradeset self, int sel_kind, og_gradeset left,
                                         og_gradeset right, og_gradeset *out_left,
                                         og_gradeset *out_right) {
    gs_parts_contributing(self, sel_builtin, (void *)(intptr_t)sel_kind, left, right, out_left,
                          out_right);
}

/* ===================================================================== */
/* Algebra -- src/algebra.rs                                              */
/* ===================================================================== */

static uint64_t binomial_uncached(uint64_t n, uint64_t k) {
    /* num_integer::binomial (0.1.45): 0 when k > n, else the multiplicative formula with the
     * smaller of k, n-k. */
    if (k > n) return 0;
    if (k > n - k) k = n - k;
    uint64_t r = 1;
    for (uint64_t d = 1; d <= k; ++d) {
        r = r * (n - k + d) / d; /* exact at every step: product of d consecutive ints / d! */
    }
    return r;
}

uint64_t og_n_choose_k(uint64_t n, uint64_t k) {
    /* algebra.rs:252-254.  Memoised (the reference carries a commented-out #[memoize] on this
     * very function, algebra.rs:250): same values, fewer cycles in table generation. */
    static uint64_t memo[64][64];
    static unsigned char have[64][64];
    if (n < 64 && k < 64) {
        if (!have[n][k]) {
            memo[n][k] = binomial_uncached(n, k);
            have[n][k] = 1;
        }
        return memo[n][k];
    }
    return binomial_uncached(n, k);
}

uint64_t og_index_to_bitfield_permut(int n, int k, uint64_t i) { /* algebra.rs:221-232 */
    uint64_t res = 0;
    for (int b = 1; b <= n; ++b) {
        uint64_t z = og_n_choose_k((uint64_t)(n - b), (uint64_t)k);
        if (i >= z) {
            res |= 1ULL << (n - b);
            i -= z;
            k -= 1;
        }
    }
    return res;
}

uint64_t og_bitfield_permut_to_index(int n, int k, uint64_t v) { /* algebra.rs:236-246 */
    uint64_t res = 0;
    for (int b = 1; b <= n; ++b) {
        uint64_t z = og_n_choose_k((uint64_t)(n - b), (uint64_t)k);
        if ((v >> (n - b)) & 1ULL) { /* v.get(n-b) ... unwrap_or(false) */
            res += z;
            k -= 1;
        }
    }
    return res;
}

double og_canonical_reordering_sign(uint64_t b1, uint64_t b2) { /* algebra.rs:199-209 */
    int32_t sum = 0;
    for (;;) {
        b1 >>= 1; /* b1.shift_left(1): toward index 0 */
        sum += popcnt64(b1 & b2);
        if (b1 == 0) break;
    }
    return (double)(1 - (sum % 2) * 2);
}

static double base_vec_dot(const og_algebra *alg, int v1, int v2) { /* algebra.rs:156-165,184-192 */
    if (v1 == v2) return alg->is_euclid ? 1.0 : alg->diag[v1];
    return 0.0;
}

double og_ortho_basis_blades_gp(const og_algebra *alg, uint64_t b1, uint64_t b2, uint64_t *res) {
    /* algebra.rs:73-83 */
    double coef = og_canonical_reordering_sign(b1, b2);
    uint64_t shared = b1 & b2;
    for (int bit = 0; bit < alg->dim; ++bit) /* iter_ones: ascending index */
        if ((shared >> bit) & 1ULL) coef *= base_vec_dot(alg, bit, bit);
    *res = b1 ^ b2;
    return coef;
}

static og_gradeset alg_full_grade_set(const og_algebra *alg) { /* algebra.rs:19-21 */
    og_gradeset g = og_gs_empty();
    for (int k = 0; k <= alg->dim; ++k) g = og_gs_add_grade(g, k);
    return g;
}

/* ===================================================================== */
/* GradeMapMV -- src/graded.rs:173-202                                    */
/* ===================================================================== */

struct og_mv {
    uint64_t mask;               /* keys of the HashMap */
    double *slab[OG_MAX_GRADE];  /* Vec<f64> per grade */
    size_t len[OG_MAX_GRADE];
};

og_mv *og_mv_new(void) { return (og_mv *)calloc(1, sizeof(og_mv)); }

void og_mv_free(og_mv *m) {
    if (!m) return;
    for (int k = 0; k < OG_MAX_GRADE; ++k) free(m->slab[k]);
    free(m);
}

int og_mv_set_grade(og_mv *m, int k, const double *vals, size_t len) {
    if (!m || k < 0 || k >= OG_MAX_GRADE) return OG_BAD_ARG;
    free(m->slab[k]);
    m->slab[k] = (double *)malloc((len ? len : 1) * sizeof(double));
    if (len) memcpy(m->slab[k], vals, len * sizeof(double));
    m->len[k] = len;
    m->mask |= 1ULL << k;
    return OG_OK;
}

uint64_t og_mv_grade_mask(const og_mv *m) { return m->mask; }
size_t og_mv_grade_len(const og_mv *m, int k) { return ((m->mask >> k) & 1ULL) ? m->len[k] : 0; }
double *og_mv_grade_ptr(og_mv *m, int k) { return ((m->mask >> k) & 1ULL) ? m->slab[k] : NULL; }

og_mv *og_mv_init_null(int dim, og_gradeset gs) { /* graded.rs:195-201 */
    og_mv *m = og_mv_new();
    int ks[OG_MAX_GRADE];
    int n = og_gs_iter(gs, ks, OG_MAX_GRADE);
    for (int i = 0; i < n; ++i) {
        size_t len = (size_t)og_n_choose_k((uint64_t)dim, (uint64_t)ks[i]);
        m->slab[ks[i]] = (double *)calloc(len ? len : 1, sizeof(double)); /* vec![0.0; C(dim,k)] */
        m->len[ks[i]] = len;
        m->mask |= 1ULL << ks[i];
    }
    return m;
}

static og_gradeset mv_grade_set(const og_mv *m) { /* graded.rs:176-184 */
    og_gradeset g = og_gs_empty();
    for (int k = 0; k < OG_MAX_GRADE; ++k)
        if ((m->mask >> k) & 1ULL) g = og_gs_add_grade(g, k);
    return g;
}

/* ===================================================================== */
/* Expr -- src/ast/expr.rs                                                */
/* ===================================================================== */

typedef enum {
    E_MV,       /* mv(x)                                   expr.rs:162-164 */
    E_ADD,      /* Add                                     expr.rs:200-210 */
    E_NEG,      /* Neg                                     expr.rs:213-221 */
    E_PRODUCT,  /* product                                 expr.rs:123-144 */
    E_REV,      /* rev                                     expr.rs:292 */
    E_GINVOL,   /* ginvol                                  expr.rs:293 */
    E_EXP,      /* exp                                     expr.rs:294 */
    E_LOG,      /* log                                     expr.rs:295 */
    E_GSELECT,  /* gselect / g                             expr.rs:322-335 */
    E_SINV,     /* sinv                                    expr.rs:353-358 */
    E_WRAP_SQRT,/* sqrt via wrap                           expr.rs:305-319 */
    E_WRAP_VINV /* vinv via wrap                           expr.rs:363-371 */
} expr_kind;

struct og_expr {
    int refcount;
    expr_kind kind;
    og_expr *a, *b;
    og_select_fn sel;   /* E_PRODUCT */
    void *sel_user;
    int gsel_is_single; /* E_GSELECT: g(k) -> single(k); else fixed mask */
    int64_t gsel_k;
    uint64_t gsel_mask;
    og_mv *value;       /* E_MV */
    int owns_value;
};

static og_expr *expr_new(expr_kind kind, og_expr *a, og_expr *b) {
    og_expr *e = (og_expr *)calloc(1, sizeof(og_expr));
    e->refcount = 1;
    e->kind = kind;
    e->a = a ? og_expr_retain(a) : NULL;
    e->b = b ? og_expr_retain(b) : NULL;
    return e;
}

og_expr *og_expr_retain(og_expr *e) {
    if (e) e->refcount++;
    return e;
}

void og_expr_release(og_expr *e) {
    if (!e) return;
    if (--e->refcount > 0) return;
    og_expr_release(e->a);
    og_expr_release(e->b);
    if (e->owns_value) og_mv_free(e->value);
    free(e);
}

og_expr *og_expr_mv(og_mv *x) {
    og_expr *e = expr_new(E_MV, NULL, NULL);
    e->value = x;
    return e;
}

og_expr *og_expr_from_f64(double x) { /* expr.rs:231-240 */
    og_expr *e = expr_new(E_MV, NULL, NULL);
    e->owns_value = 1;
    if (x == 0.0) {
        e->value = og_mv_init_null(0, og_gs_empty());
        return e;
    }
    e->value = og_mv_init_null(0, og_gs_single(0));
    e->value->slab[0][0] = x;
    return e;
}

og_expr *og_expr_basis_vector(int dim, int i) { /* expr.rs:148-157 */
    og_expr *e = expr_new(E_MV, NULL, NULL);
    e->owns_value = 1;
    e->value = og_mv_init_null(dim, og_gs_single(1));
    e->value->slab[1][i] = 1.0;
    return e;
}

og_expr *og_expr_product_custom(og_expr *l, og_expr *r, og_select_fn f, void *user) {
    og_expr *e = expr_new(E_PRODUCT, l, r);
    e->sel = f;
    e->sel_user = user;
    return e;
}

og_expr *og_expr_product(og_expr *l, og_expr *r, int sel_kind) {
    return og_expr_product_custom(l, r, sel_builtin, (void *)(intptr_t)sel_kind);
}

og_expr *og_expr_add(og_expr *l, og_expr *r) { return expr_new(E_ADD, l, r); }
og_expr *og_expr_neg(og_expr *e) { return expr_new(E_NEG, e, NULL); }

og_expr *og_expr_sub(og_expr *l, og_expr *r) { /* expr.rs:224-229: self + -rhs */
    og_expr *n = og_expr_neg(r);
    og_expr *s = og_expr_add(l, n);
    og_expr_release(n);
    return s;
}

og_expr *og_expr_div_scalar(og_expr *e, double s) { /* expr.rs:265-270: self * (1.0/(rhs as f64)) */
    og_expr *inv = og_expr_from_f64(1.0 / s);
    og_expr *p = og_expr_product(e, inv, OG_SEL_GEOMETRIC);
    og_expr_release(inv);
    return p;
}

og_expr *og_expr_rev(og_expr *e) { return expr_new(E_REV, e, NULL); }
og_expr *og_expr_ginvol(og_expr *e) { return expr_new(E_GINVOL, e, NULL); }
og_expr *og_expr_exp(og_expr *e) { return expr_new(E_EXP, e, NULL); }
og_expr *og_expr_log(og_expr *e) { return expr_new(E_LOG, e, NULL); }

og_expr *og_expr_pow(og_expr *e, og_expr *p) { /* expr.rs:300-302: exp(log(self) * p) */
    og_expr *l = og_expr_log(e);
    og_expr *m = og_expr_product(l, p, OG_SEL_GEOMETRIC);
    og_expr *x = og_expr_exp(m);
    og_expr_release(l);
    og_expr_release(m);
    return x;
}

og_expr *og_expr_sqrt(og_expr *e) { return expr_new(E_WRAP_SQRT, e, NULL); }

og_expr *og_expr_g(og_expr *e, int64_t k) { /* expr.rs:322-324 */
    og_expr *x = expr_new(E_GSELECT, e, NULL);
    x->gsel_is_single = 1;
    x->gsel_k = k;
    return x;
}

og_expr *og_expr_gselect_mask(og_expr *e, uint64_t wanted_mask) {
    og_expr *x = expr_new(E_GSELECT, e, NULL);
    x->gsel_is_single = 0;
    x->gsel_mask = wanted_mask;
    return x;
}

og_expr *og_expr_conj(og_expr *e) { /* expr.rs:338-340: self.rev().ginvol() */
    og_expr *r = og_expr_rev(e);
    og_expr *g = og_expr_ginvol(r);
    og_expr_release(r);
    return g;
}

og_expr *og_expr_scal(og_expr *e, og_expr *rhs) { /* expr.rs:343-345: (self.rev() * rhs).g(0) */
    og_expr *r = og_expr_rev(e);
    og_expr *p = og_expr_product(r, rhs, OG_SEL_GEOMETRIC);
    og_expr *g = og_expr_g(p, 0);
    og_expr_release(r);
    og_expr_release(p);
    return g;
}

og_expr *og_expr_norm_sq(og_expr *e) { /* expr.rs:348-350: self.clone().scal(self) */
    return og_expr_scal(e, e);
}

og_expr *og_expr_sinv(og_expr *e) { return expr_new(E_SINV, e, NULL); }
og_expr *og_expr_vinv(og_expr *e) { return expr_new(E_WRAP_VINV, e, NULL); }

/* ===================================================================== */
/* Reified / specialized AST -- base_types.rs, expr.rs:13-25,62-115       */
/* ===================================================================== */

typedef struct gnode {
    const og_expr *id;   /* NodeId: the Rc pointer, never dereferenced (base_types.rs:92-96) */
    og_gradeset maximal; /* base_types.rs:107 */
    og_gradeset minimal; /* base_types.rs:110 */
    int vec_space_dim;
    int kind;            /* OG_N_* */
    int child0, child1;  /* indices in the arena */
    og_select_fn sel;    /* Product.grades_to_produce */
    void *sel_user;
    og_comp_mul *muls;   /* Product.individual_comp_muls */
    size_t n_muls, cap_muls;
    og_mv *input;        /* GradedObj(T) with T = &GradeMapMV */
    int num_uses;
    int is_ready;
} gnode;

struct og_spec {
    og_algebra alg; /* the algebra given to specialize (the EXTENSION arms need blade squares at eval time) */
    gnode *nodes; /* NodeArena (a HashMap in the reference; order is irrelevant) */
    int n_nodes, cap_nodes;
    int root;
    og_expr *root_expr;  /* keeps every id alive */
    og_expr **temps;     /* expressions created inside `wrap` closures */
    int n_temps, cap_temps;
};

typedef struct builder {
    const og_algebra *alg;
    og_spec *ast;
    int status;
} builder;

static int arena_find(const og_spec *s, const og_expr *id) {
    for (int i = 0; i < s->n_nodes; ++i)
        if (s->nodes[i].id == id) return i;
    return -1;
}

static void keep_temp(og_spec *s, og_expr *e) {
    if (s->n_temps == s->cap_temps) {
        s->cap_temps = s->cap_temps ? 2 * s->cap_temps : 8;
        s->temps = (og_expr **)realloc(s->temps, (size_t)s->cap_temps * sizeof(og_expr *));
    }
    s->temps[s->n_temps++] = e;
}

/* Builder::add_node, expr.rs:13-25 */
static int add_node(builder *b, const og_expr *id, gnode proto, og_gradeset node_gs) {
    og_spec *s = b->ast;
    if (s->n_nodes == s->cap_nodes) {
        s->cap_nodes = s->cap_nodes ? 2 * s->cap_nodes : 16;
        s->nodes = (gnode *)realloc(s->nodes, (size_t)s->cap_nodes * sizeof(gnode));
    }
    proto.id = id;
    proto.maximal = og_gs_intersection(node_gs, alg_full_grade_set(b->alg));
    proto.minimal = og_gs_empty();
    proto.vec_space_dim = b->alg->dim;
    proto.num_uses = 1;
    proto.is_ready = 0;
    s->nodes[s->n_nodes] = proto;
    return s->n_nodes++;
}

static void run_closure(const og_expr *e, const og_expr *this_id, builder *b);

/* reify_or_reuse, expr.rs:73-84 */
static int reify_or_reuse(const og_expr *e, builder *b, og_gradeset *gs_out) {
    int idx = arena_find(b->ast, e);
    if (idx < 0) {
        run_closure(e, e, b);
        idx = arena_find(b->ast, e);
    } else {
        b->ast->nodes[idx].num_uses += 1;
    }
    if (idx < 0) { /* a panic inside the closure left no node */
        *gs_out = og_gs_empty();
        return -1;
    }
    *gs_out = b->ast->nodes[idx].maximal;
    return idx;
}

static gnode proto_zero(void) {
    gnode p;
    memset(&p, 0, sizeof(p));
    p.child0 = p.child1 = -1;
    return p;
}

/* The body of each Expr's `run` closure; `this_id` is the id the new node is stored under
 * (differs from `e` only when called through `wrap`, expr.rs:97-115). */
static void run_closure(const og_expr *e, const og_expr *this_id, builder *b) {
    if (b->status != OG_OK) return;
    gnode p = proto_zero();
    og_gradeset gs, lgs, rgs;
    switch (e->kind) {
    case E_MV: /* expr.rs:162-164 */
        p.kind = OG_N_GRADED_OBJ;
        p.input = e->value;
        add_node(b, this_id, p, mv_grade_set(e->value));
        return;
    case E_ADD: { /* expr.rs:204-209 */
        int l = reify_or_reuse(e->a, b, &lgs);
        int r = reify_or_reuse(e->b, b, &rgs);
        if (b->status != OG_OK) return;
        p.kind = OG_N_ADDITION;
        p.child0 = l;
        p.child1 = r;
        add_node(b, this_id, p, og_gs_add(lgs, rgs));
        return;
    }
    case E_PRODUCT: { /* expr.rs:129-143 */
        int l = reify_or_reuse(e->a, b, &lgs);
        int r = reify_or_reuse(e->b, b, &rgs);
        if (b->status != OG_OK) return;
        p.kind = OG_N_PRODUCT;
        p.child0 = l;
        p.child1 = r;
        p.sel = e->sel;
        p.sel_user = e->sel_user;
        /* iter_grade_sets_cp(&left_gs, &right_gs).map(grades_to_produce).collect() */
        int lk[OG_MAX_GRADE], rk[OG_MAX_GRADE];
        int nl = og_gs_iter(lgs, lk, OG_MAX_GRADE), nr = og_gs_iter(rgs, rk, OG_MAX_GRADE);
        gs = og_gs_empty();
        for (int i = 0; i < nl; ++i)
            for (int j = 0; j < nr; ++j) gs = og_gs_add(gs, e->sel(lk[i], rk[j], e->sel_user));
        add_node(b, this_id, p, gs);
        return;
    }
    case E_NEG:
    case E_REV:
    case E_GINVOL:
    case E_SINV: { /* expr.rs:216-219, 279-285 (grade_op = id), 354-357 */
        int c = reify_or_reuse(e->a, b, &gs);
        if (b->status != OG_OK) return;
        p.kind = e->kind == E_NEG      ? OG_N_NEGATION
                 : e->kind == E_REV    ? OG_N_REVERSE
                 : e->kind == E_GINVOL ? OG_N_GRADE_INVOLUTION
                                       : OG_N_SCALAR_INVERSION;
        p.child0 = c;
        add_node(b, this_id, p, gs);
        return;
    }
    case E_EXP:
    case E_LOG: { /* expr.rs:294-295 with grade_op = exp / log */
        int c = reify_or_reuse(e->a, b, &gs);
        if (b->status != OG_OK) return;
        og_gradeset out;
        int st = e->kind == E_EXP ? gs_exp(gs, &out) : gs_log(gs, &out);
        if (st != OG_OK) {
            b->status = st;
            return;
        }
        p.kind = e->kind == E_EXP ? OG_N_EXPONENTIAL : OG_N_LOGARITHM;
        p.child0 = c;
        add_node(b, this_id, p, out);
        return;
    }
    case E_GSELECT: { /* expr.rs:327-335: get_wanted_grades(&gs).intersection(gs) */
        int c = reify_or_reuse(e->a, b, &gs);
        if (b->status != OG_OK) return;
        og_gradeset wanted;
        if (e->gsel_is_single) {
            wanted = og_gs_single(e->gsel_k);
        } else {
            wanted.bits = e->gsel_mask;
            wanted.len = 64;
        }
        p.kind = OG_N_GRADE_PROJECTION;
        p.child0 = c;
        add_node(b, this_id, p, og_gs_intersection(wanted, gs));
        return;
    }
    case E_WRAP_SQRT:
    case E_WRAP_VINV: { /* wrap, expr.rs:97-115 */
        int self_idx = reify_or_reuse(e->a, b, &gs);
        if (b->status != OG_OK) return;
        if (e->kind == E_WRAP_SQRT && og_gs_is_just(gs, 0)) {
            /* Wrapper::Node, expr.rs:310-314 */
            p.kind = OG_N_SCALAR_SQRT;
            p.child0 = self_idx;
            add_node(b, this_id, p, gs);
            return;
        }
        og_expr *inner;
        if (e->kind == E_WRAP_SQRT) {
            /* this.pow(0.5), expr.rs:316 */
            og_expr *half = og_expr_from_f64(0.5);
            inner = og_expr_pow(e->a, half);
            og_expr_release(half);
        } else if (og_gs_is_just(gs, 0)) {
            inner = og_expr_sinv(e->a); /* expr.rs:365-366 */
        } else {
            /* this.clone().rev() * this.norm_sq().sinv(), expr.rs:368 */
            og_expr *r = og_expr_rev(e->a);
            og_expr *nsq = og_expr_norm_sq(e->a);
            og_expr *si = og_expr_sinv(nsq);
            inner = og_expr_product(r, si, OG_SEL_GEOMETRIC);
            og_expr_release(r);
            og_expr_release(nsq);
            og_expr_release(si);
        }
        keep_temp(b->ast, inner);
        /* (wrapper_expr.run)(wrapper_id, b) then undo the double count, expr.rs:105-110 */
        run_closure(inner, this_id, b);
        if (b->status != OG_OK) return;
        b->ast->nodes[self_idx].num_uses -= 1;
        return;
    }
    }
}

/* specialize.rs:53-94 */
static void rec_update_minimal(og_spec *s, int idx, og_gradeset wanted, int *status) {
    if (*status != OG_OK) return;
    s->nodes[idx].minimal = og_gs_add(s->nodes[idx].minimal, wanted);
    gnode *n = &s->nodes[idx];
    switch (n->kind) {
    case OG_N_GRADED_OBJ: return;
    case OG_N_GRADE_PROJECTION:
    case OG_N_NEGATION:
    case OG_N_REVERSE:
    case OG_N_GRADE_INVOLUTION:
    case OG_N_SCALAR_INVERSION:
    case OG_N_SCALAR_SQRT: rec_update_minimal(s, n->child0, wanted, status); return;
    case OG_N_ADDITION: {
        int l = n->child0, r = n->child1;
        rec_update_minimal(s, l, wanted, status);
        rec_update_minimal(s, r, wanted, status);
        return;
    }
    case OG_N_PRODUCT: {
        int l = n->child0, r = n->child1;
        og_gradeset lw, rw;
        gs_parts_contributing(wanted, n->sel, n->sel_user, s->nodes[l].maximal,
                              s->nodes[r].maximal, &lw, &rw);
        rec_update_minimal(s, l, lw, status);
        rec_update_minimal(s, r, rw, status);
        return;
    }
    case OG_N_EXPONENTIAL: { /* specialize.rs:91: wanted.log() */
        og_gradeset w;
        int st = gs_log(wanted, &w);
        if (st != OG_OK) {
            *status = st;
            return;
        }
        rec_update_minimal(s, n->child0, w, status);
        return;
    }
    case OG_N_LOGARITHM: { /* specialize.rs:92: wanted.exp() */
        og_gradeset w;
        int st = gs_exp(wanted, &w);
        if (st != OG_OK) {
            *status = st;
            return;
        }
        rec_update_minimal(s, n->child0, w, status);
        return;
    }
    }
}

static void push_mul(gnode *n, og_comp_mul m) {
    if (n->n_muls == n->cap_muls) {
        n->cap_muls = n->cap_muls ? 2 * n->cap_muls : 64;
        n->muls = (og_comp_mul *)realloc(n->muls, n->cap_muls * sizeof(og_comp_mul));
    }
    n->muls[n->n_muls++] = m;
}

/* iter_comp_muls_for_kvectors_prod, specialize.rs:162-183.  iter_basis_blades_of_grade
 * (algebra.rs:50-58) walks index 0..grade_dim(k) and unranks each index; the right-hand list is
 * re-derived for every left blade in the reference -- here it is unranked once per call (same
 * values, same order). */
static void comp_muls_for_kvectors(const og_algebra *alg, gnode *n, int k_left, int k_right,
                                   og_gradeset contribs) {
    int dim = alg->dim;
    uint64_t gl = og_n_choose_k((uint64_t)dim, (uint64_t)k_left);
    uint64_t gr = og_n_choose_k((uint64_t)dim, (uint64_t)k_right);
    uint64_t *rb = (uint64_t *)malloc((gr ? gr : 1) * sizeof(uint64_t));
    for (uint64_t j = 0; j < gr; ++j) rb[j] = og_index_to_bitfield_permut(dim, k_right, j);
    for (uint64_t i = 0; i < gl; ++i) {
        uint64_t bl = og_index_to_bitfield_permut(dim, k_left, i);
        for (uint64_t j = 0; j < gr; ++j) {
            uint64_t br = rb[j], bres;
            double coeff = og_ortho_basis_blades_gp(alg, bl, br, &bres);
            /* basis_blade_to_component, algebra.rs:41-45 */
            int rg = popcnt64(bres);
            if (og_gs_contains(contribs, rg)) {
                og_comp_mul m;
                m.result_grade = (size_t)rg;
                m.result_index = (size_t)og_bitfield_permut_to_index(dim, rg, bres);
                m.left_grade = (size_t)popcnt64(bl);
                m.left_index = (size_t)og_bitfield_permut_to_index(dim, popcnt64(bl), bl);
                m.right_grade = (size_t)popcnt64(br);
                m.right_index = (size_t)og_bitfield_permut_to_index(dim, popcnt64(br), br);
                m.coeff = coeff;
                push_mul(n, m);
            }
        }
    }
    free(rb);
}

/* specialize.rs:96-160 */
static void rec_apply_algebra(og_spec *s, int idx, const og_algebra *alg, int *status) {
    if (*status != OG_OK) return;
    gnode *n = &s->nodes[idx];
    if (n->is_ready) {
        if (!(n->num_uses >= 2)) { /* specialize.rs:104-107 */
            set_panic("Algebra was already applied to a node that is referred to only once");
            *status = OG_PANIC_ASSERT;
        }
        return;
    }
    n->is_ready = 1;
    if (!og_gs_includes(n->maximal, n->minimal)) { /* specialize.rs:113-117 */
        set_panic("Inferred minimal grade set contains grades not available in maximal grade set");
        *status = OG_PANIC_ASSERT;
        return;
    }
    switch (n->kind) {
    case OG_N_GRADED_OBJ: return;
    case OG_N_NEGATION:
    case OG_N_GRADE_PROJECTION:
    case OG_N_REVERSE:
    case OG_N_GRADE_INVOLUTION:
    case OG_N_SCALAR_INVERSION:
    case OG_N_SCALAR_SQRT:
    case OG_N_EXPONENTIAL:
    case OG_N_LOGARITHM: rec_apply_algebra(s, n->child0, alg, status); return;
    case OG_N_ADDITION: {
        int l = n->child0, r = n->child1;
        rec_apply_algebra(s, l, alg, status);
        rec_apply_algebra(s, r, alg, status);
        return;
    }
    case OG_N_PRODUCT: {
        int l = n->child0, r = n->child1;
        rec_apply_algebra(s, l, alg, status);
        rec_apply_algebra(s, r, alg, status);
        if (*status != OG_OK) return;
        n = &s->nodes[idx];
        og_gradeset gs_left = s->nodes[l].minimal, gs_right = s->nodes[r].minimal;
        static contrib buf[OG_MAX_GRADE * OG_MAX_GRADE];
        int nc = gs_iter_contribs(n->minimal, n->sel, n->sel_user, gs_left, gs_right, buf,
                                  OG_MAX_GRADE * OG_MAX_GRADE);
        contrib *mine = (contrib *)malloc((size_t)(nc ? nc : 1) * sizeof(contrib));
        memcpy(mine, buf, (size_t)nc * sizeof(contrib));
        /* Vec::collect grows geometrically in the reference; reserve the upper bound once
         * instead (first-touch of fresh pages is the dominant cost of a 940 MB table). */
        size_t bound = 0;
        for (int c = 0; c < nc; ++c)
            bound += (size_t)og_n_choose_k((uint64_t)alg->dim, (uint64_t)mine[c].k_left) *
                     (size_t)og_n_choose_k((uint64_t)alg->dim, (uint64_t)mine[c].k_right);
        if (bound > n->cap_muls) {
            n->muls = (og_comp_mul *)realloc(n->muls, bound * sizeof(og_comp_mul));
            n->cap_muls = bound;
        }
        for (int c = 0; c < nc; ++c)
            comp_muls_for_kvectors(alg, n, mine[c].k_left, mine[c].k_right, mine[c].contribs);
        free(mine);
        return;
    }
    }
}

og_spec *og_specialize(og_expr *e, const og_algebra *alg, int *status) { /* specialize.rs:36-50 */
    int st = OG_OK;
    og_spec *s = (og_spec *)calloc(1, sizeof(og_spec));
    s->root_expr = og_expr_retain(e);
    s->alg = *alg;
    builder b = {alg, s, OG_OK};
    og_gradeset root_gs;
    /* Expr::reify, expr.rs:62-69 */
    s->root = reify_or_reuse(e, &b, &root_gs);
    st = b.status;
    if (st == OG_OK) rec_update_minimal(s, s->root, root_gs, &st);
    if (st == OG_OK) rec_apply_algebra(s, s->root, alg, &st);
    if (status) *status = st;
    if (st != OG_OK) {
        og_spec_free(s);
        return NULL;
    }
    return s;
}

void og_spec_free(og_spec *s) {
    if (!s) return;
    for (int i = 0; i < s->n_nodes; ++i) free(s->nodes[i].muls);
    free(s->nodes);
    for (int i = 0; i < s->n_temps; ++i) og_expr_release(s->temps[i]);
    free(s->temps);
    og_expr_release(s->root_expr);
    free(s);
}

int og_spec_num_nodes(const og_spec *s) { return s->n_nodes; }
int og_spec_root(const og_spec *s) { return s->root; }

int og_spec_node(const og_spec *s, int idx, og_node_info *out) {
    if (idx < 0 || idx >= s->n_nodes) return OG_BAD_ARG;
    const gnode *n = &s->nodes[idx];
    out->kind = n->kind;
    out->child0 = n->child0;
    out->child1 = n->child1;
    out->maximal = n->maximal.bits & lowmask(n->maximal.len);
    out->minimal = n->minimal.bits & lowmask(n->minimal.len);
    out->vec_space_dim = n->vec_space_dim;
    out->num_uses = n->num_uses;
    out->n_comp_muls = n->n_muls;
    out->input = n->input;
    return OG_OK;
}

const og_comp_mul *og_spec_comp_muls(const og_spec *s, int idx) {
    if (idx < 0 || idx >= s->n_nodes) return NULL;
    return s->nodes[idx].muls;
}

/* ===================================================================== */
/* Evaluation -- src/eval.rs, src/graded.rs:51-79                         */
/* ===================================================================== */

typedef struct cache {
    og_mv **slot; /* HashMap<NodeId, R>: slot[i] != NULL <=> key present */
    int mode;
    int status;
} cache;

static double *grade_slice_mut(og_mv *m, size_t k, int *status) { /* graded.rs:192-194 */
    if (k >= OG_MAX_GRADE || !((m->mask >> k) & 1ULL)) {
        set_panic("called `Option::unwrap()` on a `None` value (grade absent from GradeMapMV)");
        *status = OG_PANIC_MISSING_GRADE;
        return NULL;
    }
    return m->slab[k];
}

static void negate_grade(og_mv *m, int k, int *status) { /* graded.rs:61-65 */
    double *s = grade_slice_mut(m, (size_t)k, status);
    if (!s) return;
    for (size_t i = 0; i < m->len[k]; ++i) s[i] = -s[i];
}

static void add_grades_from(og_mv *self, const og_mv *input, og_gradeset grades, int *status, int f32) {
    /* graded.rs:67-78 */
    og_gradeset igs = mv_grade_set(input);
    int ks[OG_MAX_GRADE];
    int n = og_gs_iter(grades, ks, OG_MAX_GRADE);
    for (int t = 0; t < n; ++t) {
        int k = ks[t];
        if (og_gs_contains(igs, k)) {
            const double *in = input->slab[k];
            double *res = grade_slice_mut(self, (size_t)k, status);
            if (!res) return;
            size_t len = self->len[k] < input->len[k] ? self->len[k] : input->len[k]; /* zip */
            if (f32) { /* OG_EVAL_F32: the same statement on binary32 values (the input is rounded as the upload rounds it) */
                for (size_t i = 0; i < len; ++i) {
                    const float r = (float)res[i], x = (float)in[i];
                    const float sum = r + x;
                    res[i] = (double)sum;
                }
                continue;
            }
            for (size_t i = 0; i < len; ++i) res[i] = res[i] + in[i];
        }
    }
}

void og_product_loop(const og_comp_mul *muls, size_t n, double *const *left, double *const *right,
                     double **res) { /* eval.rs:77-83 */
    for (size_t e = 0; e < n; ++e) {
        const og_comp_mul *mul = &muls[e];
        double val_left = left[mul->left_grade][mul->left_index];
        double val_right = right[mul->right_grade][mul->right_index];
        double *val_result = &res[mul->result_grade][mul->result_index];
        *val_result += val_left * val_right * mul->coeff;
    }
}

static void add_to_res(const og_spec *s, int res_id, int this_id, cache *c);

static void store_in_cache(const og_spec *s, int this_id, cache *c) { /* eval.rs:21-33 */
    if (c->status != OG_OK) return;
    const gnode *this_ = &s->nodes[this_id];
    if (c->slot[this_id] == NULL) {
        c->slot[this_id] = og_mv_init_null(this_->vec_space_dim, this_->minimal);
        add_to_res(s, this_id, this_id, c);
    }
}

/*
 * EXTENSION -- "no reference behaviour, parity unpinned".  eval.rs:112-113 is todo!(); what IS pinned by the reference
 * are the grade rules: exp takes a single-grade k-vector to grades {0, k} (grade_set.rs:181-188), log takes
 * <A>_0 + <A>_k to grade {k} (grade_set.rs:190-197), and minimal sets flow down as wanted.log() / wanted.exp()
 * (specialize.rs:91-92).  Those rules only make sense for a k-vector B whose square is a scalar, s = <B B>_0; then
 *     exp(B) = cos(t) + B sin(t)/t,  t = sqrt(-s)   (s < 0)        log(a + B)|_k = B atan2(m, a) / m,  m = sqrt(-s)
 *            = cosh(t) + B sinh(t)/t, t = sqrt(s)    (s > 0)                      = B atanh(m / a) / m, m = sqrt(s)
 *            = 1 + B                                 (s == 0)                     = B / a
 * (the logarithm's scalar part, ln|A|, has no grade to go to: log is the versor logarithm, exact for unit versors, and
 * pow = exp(log * p), expr.rs:300-303, is the power of the normalised versor).
 * The operand is evaluated into its own buffer like a product operand (eval.rs:67-68) and the result is ADDED to res
 * (like every arm).  s = sum_i (B_i * B_i) * sq_i in component order, sq_i = e_i e_i from ortho_basis_blades_gp
 * (algebra.rs:73-83), each term rounded like eval.rs:82.  Domain: B is refused (OG_PANIC_DOMAIN) when the non-scalar
 * part of B B is not negligible, sum_T (<B B>_T)^2 > 2^-40 (sum_i B_i^2)^2.
 */
static void ext_exp_log(const og_spec *s, int res_id, int this_id, cache *c) {
    const gnode *this_ = &s->nodes[this_id];
    const int is_exp = this_->kind == OG_N_EXPONENTIAL;
    store_in_cache(s, this_->child0, c);
    if (c->status != OG_OK) return;
    og_mv *res = c->slot[res_id];
    og_mv *arg = c->slot[this_->child0];
    if (res == arg) {
        set_panic("exp / log operand aliases its own result buffer");
        c->status = OG_PANIC_MISSING_GRADE;
        return;
    }
    /* the k-vector part: the one non-zero grade of the operand's minimal set (exp: the only grade) */
    const gnode *child = &s->nodes[this_->child0];
    int ks[OG_MAX_GRADE];
    int nk = og_gs_iter(child->minimal, ks, OG_MAX_GRADE);
    int k = -1;
    for (int t = 0; t < nk; ++t)
        if (ks[t] != 0 || (is_exp && nk == 1)) k = ks[t];
    if (k < 0) { /* log of a bare scalar: grade_set.rs:192-195 already refuses it while the Expr is built */
        set_panic("log can only be used on multivectors of the form <A>_0 + <A>_k");
        c->status = OG_PANIC_ASSERT;
        return;
    }
    if (!((arg->mask >> k) & 1ULL)) {
        set_panic("grade absent from exp / log operand");
        c->status = OG_PANIC_MISSING_GRADE;
        return;
    }
    const double *B = arg->slab[k];
    const size_t m = arg->len[k];
    const int dim = child->vec_space_dim;
    double sq = 0.0, nrm = 0.0;
    uint64_t *blade = (uint64_t *)malloc((m ? m : 1) * sizeof(uint64_t));
    for (size_t i = 0; i < m; ++i) {
        uint64_t r;
        blade[i] = og_index_to_bitfield_permut(dim, k, i);
        const double sqi = og_ortho_basis_blades_gp(&s->alg, blade[i], blade[i], &r);
        sq += B[i] * B[i] * sqi;
        nrm += B[i] * B[i];
    }
    /* non-scalar part of B B: pairs of distinct blades that commute contribute 2 B_i B_j e_i e_j */
    double viol = 0.0;
    {
        const size_t nblades = (size_t)1 << dim;
        double *acc = (double *)calloc(nblades, sizeof(double));
        for (size_t i = 0; i < m; ++i)
            for (size_t j = i + 1; j < m; ++j) {
                uint64_t r1, r2;
                const double c1 = og_ortho_basis_blades_gp(&s->alg, blade[i], blade[j], &r1);
                const double c2 = og_ortho_basis_blades_gp(&s->alg, blade[j], blade[i], &r2);
                if (c1 == c2) acc[r1] += B[i] * B[j] * (2.0 * c1); /* they commute (else c1 == -c2 and the pair cancels) */
            }
        for (size_t t = 0; t < nblades; ++t) viol += acc[t] * acc[t];
        free(acc);
    }
    free(blade);
    if (viol > OG_EXPLOG_DOMAIN_TOL2 * (nrm * nrm)) {
        set_panic("exp / log of a k-vector whose square is not a scalar");
        c->status = OG_PANIC_DOMAIN;
        return;
    }
    double c0 = 0.0, f;
    if (is_exp) {
        if (sq < 0.0) {
            const double t = sqrt(-sq);
            c0 = cos(t);
            f = sin(t) / t;
        } else if (sq > 0.0) {
            const double t = sqrt(sq);
            c0 = cosh(t);
            f = sinh(t) / t;
        } else if (sq == 0.0) {
            c0 = 1.0;
            f = 1.0;
        } else {
            c0 = sq; /* NaN propagates */
            f = sq;
        }
    } else {
        double a = 0.0;
        if ((arg->mask & 1ULL) && arg->len[0] > 0) a = arg->slab[0][0];
        if (sq < 0.0) {
            const double mm = sqrt(-sq);
            f = atan2(mm, a) / mm;
        } else if (sq > 0.0) {
            const double mm = sqrt(sq);
            f = atanh(mm / a) / mm;
        } else if (sq == 0.0) {
            f = 1.0 / a;
        } else {
            f = sq;
        }
    }
    og_gradeset mine = this_->minimal;
    if (is_exp && og_gs_contains(mine, 0)) {
        double *r0 = grade_slice_mut(res, 0, &c->status);
        if (!r0) return;
        r0[0] = r0[0] + c0;
    }
    /* exp of a bare scalar (k = 0; grade_set.rs:181 allows it): the two statements land in the same component,
     * r0 = (r0 + cosh|a|) + (sinh|a| / |a|) a = e^a */
    if (og_gs_contains(mine, k)) {
        double *rk = grade_slice_mut(res, (size_t)k, &c->status);
        if (!rk) return;
        const size_t len = res->len[k] < m ? res->len[k] : m;
        for (size_t i = 0; i < len; ++i) rk[i] = rk[i] + f * B[i];
    }
}

static void add_to_res(const og_spec *s, int res_id, int this_id, cache *c) { /* eval.rs:35-115 */
    if (c->status != OG_OK) return;
    const gnode *this_ = &s->nodes[this_id];
    if (og_gs_is_empty(this_->minimal)) return; /* eval.rs:40-43 */
    int ks[OG_MAX_GRADE];
    int nk;
    switch (this_->kind) {
    case OG_N_GRADED_OBJ: /* eval.rs:45-50 */
        add_grades_from(c->slot[res_id], this_->input, this_->minimal, &c->status, c->mode & OG_EVAL_F32);
        return;
    case OG_N_ADDITION: /* eval.rs:51-54 */
        add_to_res(s, res_id, this_->child0, c);
        add_to_res(s, res_id, this_->child1, c);
        return;
    case OG_N_NEGATION: /* eval.rs:55-60 */
        add_to_res(s, res_id, this_->child0, c);
        if (c->status != OG_OK) return;
        nk = og_gs_iter(this_->minimal, ks, OG_MAX_GRADE);
        for (int t = 0; t < nk; ++t) negate_grade(c->slot[res_id], ks[t], &c->status);
        return;
    case OG_N_PRODUCT: { /* eval.rs:61-86 */
        store_in_cache(s, this_->child0, c);
        store_in_cache(s, this_->child1, c);
        if (c->status != OG_OK) return;
        og_mv *res = c->slot[res_id];
        og_mv *left = c->slot[this_->child0];
        og_mv *right = c->slot[this_->child1];
        if (res == left || res == right) {
            /* eval.rs:70-76: res is moved OUT of the cache before left/right are borrowed; if an
             * operand were the result node itself the Rust would read the empty placeholder and
             * panic on its missing grades.  Cannot happen for a well-formed DAG. */
            set_panic("product operand aliases its own result buffer");
            c->status = OG_PANIC_MISSING_GRADE;
            return;
        }
        for (size_t e = 0; e < this_->n_muls; ++e) {
            const og_comp_mul *mul = &this_->muls[e];
            /* grade_slice -> &self.0[&k]: panics when the grade is absent (graded.rs:186-190) */
            if (!((left->mask >> mul->left_grade) & 1ULL) ||
                !((right->mask >> mul->right_grade) & 1ULL)) {
                set_panic("grade absent from product operand");
                c->status = OG_PANIC_MISSING_GRADE;
                return;
            }
            double val_left = left->slab[mul->left_grade][mul->left_index];
            double val_right = right->slab[mul->right_grade][mul->right_index];
            double *rs = grade_slice_mut(res, mul->result_grade, &c->status);
            if (!rs) return;
            double *val_result = &rs[mul->result_index];
            if (c->mode & OG_EVAL_F32) { /* eval.rs:82 on binary32: (l * r) rounded, * coeff rounded, += rounded */
                const float l = (float)val_left, r = (float)val_right, cf = (float)mul->coeff;
                float prod = l * r;
                prod = prod * cf;
                const float acc = (float)*val_result;
                const float sum = acc + prod;
                *val_result = (double)sum;
                continue;
            }
            *val_result += val_left * val_right * mul->coeff;
        }
        return;
    }
    case OG_N_REVERSE: /* eval.rs:87-94 */
        add_to_res(s, res_id, this_->child0, c);
        if (c->status != OG_OK) return;
        nk = og_gs_iter(this_->minimal, ks, OG_MAX_GRADE);
        for (int t = 0; t < nk; ++t) {
            size_t k = (size_t)ks[t];
            if (k == 0 && (c->mode & OG_EVAL_DEBUG)) { /* `k - 1` on usize, overflow checks on */
                set_panic("attempt to subtract with overflow");
                c->status = OG_PANIC_OVERFLOW;
                return;
            }
            /* wrapping arithmetic of a release build: 0 * usize::MAX / 2 % 2 == 0 */
            size_t v = k * (k - 1) / 2;
            if (v % 2 == 1) negate_grade(c->slot[res_id], (int)k, &c->status);
        }
        return;
    case OG_N_GRADE_INVOLUTION: /* eval.rs:95-102 */
        add_to_res(s, res_id, this_->child0, c);
        if (c->status != OG_OK) return;
        nk = og_gs_iter(this_->minimal, ks, OG_MAX_GRADE);
        for (int t = 0; t < nk; ++t)
            if (ks[t] % 2 == 1) negate_grade(c->slot[res_id], ks[t], &c->status);
        return;
    case OG_N_SCALAR_INVERSION:
    case OG_N_SCALAR_SQRT: { /* eval.rs:103-110 */
        add_to_res(s, res_id, this_->child0, c);
        if (c->status != OG_OK) return;
        double *sl = grade_slice_mut(c->slot[res_id], 0, &c->status);
        if (!sl) return;
        if (c->mode & OG_EVAL_F32) { /* correctly rounded binary32 division / square root */
            const float x = (float)sl[0];
            const float y = this_->kind == OG_N_SCALAR_INVERSION ? 1.0f / x : sqrtf(x);
            sl[0] = (double)y;
            return;
        }
        sl[0] = this_->kind == OG_N_SCALAR_INVERSION ? 1.0 / sl[0] : sqrt(sl[0]);
        return;
    }
    case OG_N_GRADE_PROJECTION: /* eval.rs:111 */
        add_to_res(s, res_id, this_->child0, c);
        return;
    case OG_N_EXPONENTIAL:
    case OG_N_LOGARITHM: /* eval.rs:112-113 */
        if (c->mode & OG_EVAL_EXT_EXPLOG) {
            ext_exp_log(s, res_id, this_id, c);
            return;
        }
        set_panic("not yet implemented");
        c->status = OG_PANIC_TODO;
        return;
    }
}

int og_eval(const og_spec *s, int mode, og_mv **out) { /* eval.rs:12-19 */
    cache c;
    c.slot = (og_mv **)calloc((size_t)s->n_nodes, sizeof(og_mv *));
    c.mode = mode;
    c.status = OG_OK;
    store_in_cache(s, s->root, &c);
    og_mv *root = c.slot[s->root];
    c.slot[s->root] = NULL; /* cache.remove(&root_id) */
    for (int i = 0; i < s->n_nodes; ++i) og_mv_free(c.slot[i]);
    free(c.slot);
    if (c.status != OG_OK) {
        og_mv_free(root);
        if (out) *out = NULL;
        return c.status;
    }
    if (out)
        *out = root;
    else
        og_mv_free(root);
    return OG_OK;
}

int og_eval_batch(const og_spec *s, int mode, og_mv **inputs, const double **in_data, int n_inputs,
                  int64_t batch, double *out_data, size_t out_row_len) {
    size_t in_row[64];
    if (n_inputs > 64) return OG_BAD_ARG;
    for (int j = 0; j < n_inputs; ++j) {
        in_row[j] = 0;
        for (int k = 0; k < OG_MAX_GRADE; ++k) in_row[j] += og_mv_grade_len(inputs[j], k);
    }
    for (int64_t i = 0; i < batch; ++i) {
        for (int j = 0; j < n_inputs; ++j) {
            if (!in_data[j]) continue;
            const double *row = in_data[j] + (size_t)i * in_row[j];
            for (int k = 0; k < OG_MAX_GRADE; ++k) {
                size_t len = og_mv_grade_len(inputs[j], k);
                if (!((inputs[j]->mask >> k) & 1ULL)) continue;
                memcpy(inputs[j]->slab[k], row, len * sizeof(double));
                row += len;
            }
        }
        og_mv *out = NULL;
        int st = og_eval(s, mode, &out);
        if (st != OG_OK) return st;
        size_t pos = 0;
        double *orow = out_data + (size_t)i * out_row_len;
        for (int k = 0; k < OG_MAX_GRADE; ++k) {
            if (!((out->mask >> k) & 1ULL)) continue;
            if (pos + out->len[k] > out_row_len) {
                og_mv_free(out);
                return OG_BAD_ARG;
            }
            memcpy(orow + pos, out->slab[k], out->len[k] * sizeof(double));
            pos += out->len[k];
        }
        og_mv_free(out);
        if (pos != out_row_len) return OG_BAD_ARG;
    }
    return OG_OK;
}


/* ---- cpu-packed baselines ------------------------------------------------------------------- */
#include <pthread.h>
#include <time.h>

static size_t row_offset_of(og_gradeset gs, int dim, size_t grade) {
    size_t off = 0;
    for (size_t k = 0; k < grade; ++k)
        if (og_gs_contains(gs, (int)k)) off += (size_t)og_n_choose_k((uint64_t)dim, (uint64_t)k);
    return off;
}
static size_t row_len_of(og_gradeset gs, int dim) {
    size_t len = 0;
    for (size_t k = 0; k <= (size_t)dim; ++k)
        if (og_gs_contains(gs, (int)k)) len += (size_t)og_n_choose_k((uint64_t)dim, (uint64_t)k);
    return len;
}

size_t og_pack_root_product(const og_spec *s, og_packed_mul **out, size_t *left_len, size_t *right_len, size_t *out_len) {
    const gnode *root = &s->nodes[s->root];
    if (root->kind != OG_N_PRODUCT) return 0;
    const gnode *l = &s->nodes[root->child0], *r = &s->nodes[root->child1];
    if (l->kind != OG_N_GRADED_OBJ || r->kind != OG_N_GRADED_OBJ) return 0;
    const int dim = root->vec_space_dim;
    og_packed_mul *p = (og_packed_mul *)malloc(sizeof(og_packed_mul) * (root->n_muls ? root->n_muls : 1));
    if (!p) return 0;
    /* operand rows are laid out by the grades the operands' minimal sets hold (what eval.rs copies into its cache buffers) */
    for (size_t e = 0; e < root->n_muls; ++e) {
        const og_comp_mul *m = &root->muls[e];
        p[e].left = (uint32_t)(row_offset_of(l->minimal, dim, m->left_grade) + m->left_index);
        p[e].right = (uint32_t)(row_offset_of(r->minimal, dim, m->right_grade) + m->right_index);
        p[e].out = (uint32_t)(row_offset_of(root->minimal, dim, m->result_grade) + m->result_index);
        p[e].coeff = (float)m->coeff;
        if ((double)p[e].coeff != m->coeff) { /* a general diagonal metric: the 16-byte entry would compute another product */
            free(p);
            return 0;
        }
    }
    *out = p;
    *left_len = row_len_of(l->minimal, dim);
    *right_len = row_len_of(r->minimal, dim);
    *out_len = row_len_of(root->minimal, dim);
    return root->n_muls;
}

void og_packed_free(og_packed_mul *p) { free(p); }

typedef struct {
    const og_packed_mul *muls;
    size_t n, ll, rl, ol;
    const double *left, *right;
    double *out;
    int64_t first, last;
} packed_job;

static void *packed_worker(void *arg) {
    const packed_job *j = (const packed_job *)arg;
    for (int64_t it = j->first; it < j->last; ++it) {
        const double *l = j->left + (size_t)it * j->ll, *r = j->right + (size_t)it * j->rl;
        double *o = j->out + (size_t)it * j->ol;
        for (size_t c = 0; c < j->ol; ++c) o[c] = 0.0;            /* init_null_mv */
        for (size_t e = 0; e < j->n; ++e) {                       /* eval.rs:77-83 */
            const og_packed_mul *m = &j->muls[e];
            o[m->out] += l[m->left] * r[m->right] * (double)m->coeff;
        }
    }
    return NULL;
}

double og_packed_eval_batch(const og_packed_mul *muls, size_t n, const double *left, size_t left_len, const double *right,
                            size_t right_len, double *out, size_t out_len, int64_t batch, int threads) {
    if (threads < 1) threads = 1;
    if ((int64_t)threads > batch) threads = (int)(batch > 0 ? batch : 1);
    packed_job *jobs = (packed_job *)malloc(sizeof(packed_job) * (size_t)threads);
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    char *started = (char *)calloc((size_t)threads, 1);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const int64_t per = (batch + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        jobs[t] = (packed_job){muls, n, left_len, right_len, out_len, left, right, out, t * per, (t + 1) * per < batch ? (t + 1) * per : batch};
        /* a thread that cannot be started does not silently skip its items (the throughput would read too high): inline */
        if (threads > 1 && pthread_create(&tid[t], NULL, packed_worker, &jobs[t]) == 0) started[t] = 1;
        else packed_worker(&jobs[t]);
    }
    for (int t = 0; t < threads; ++t)
        if (started[t]) pthread_join(tid[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(jobs);
    free(tid);
    free(started);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
