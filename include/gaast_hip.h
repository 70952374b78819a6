/*
 * gaast_hip.h -- C ABI of the MI355X (gfx950) back end for gaast's phase-4 evaluation.
 *
 * This is the drop-in boundary: it replaces the body of the reference's
 *     impl<T: GradedData + Debug> SpecializedAst<T> { pub fn eval<R: GradedDataMut>(&self) -> R }
 * (reference src/eval.rs:10-19) and the storage traits it evaluates over
 * (src/graded.rs:43-79).  Phases 1-3 (Expr operators, reify, specialize) stay on the host;
 * what crosses this boundary is exactly what the reference's public read API exposes after
 * phase 3 (src/ast/specialize.rs:17-24, src/ast/base_types.rs:8-55,124-146), flattened into
 * plain structs, plus per-grade component slabs.
 *
 * Conventions
 *  - every entry point returns a gaast_status (0 = OK); nothing aborts or throws across
 *    the ABI (the reference panics instead: see the GAAST_ERR_* comments);
 *  - one host thread drives the library (SpecializedAst is !Send + !Sync in the reference);
 *    calls are asynchronous on the library stream unless stated, gaast_hip_synchronize()
 *    or a download makes results visible;
 *  - the caller owns handles, the library owns the device memory behind them, no host
 *    pointer is retained after a call returns;
 *  - one process drives one GPU (one rank per GPU; multi-GPU batches shard by item: the
 *    "multi-GPU" section below is the only exchange between ranks);
 *  - the library's state (device, stream, communicator, scratch buffers) is per process and
 *    NOT synchronised: calls must come from one thread at a time.  Every entry point that
 *    touches the GPU makes the library's device current for the calling thread first, so the
 *    driving thread may change between calls (HIP's current device is per thread).
 *
 * Device storage of a batched multivector ("graded rows"): one row per batch item, a row
 * holds the dense per-grade component arrays of that item concatenated in ascending grade
 * order; within grade k the C(dim,k) components are in the reference's index order
 * (src/algebra.rs:221-246: colex rank of the blade's basis-vector set).  So the
 * reference's `grade_slice(k)` of item i is the contiguous run
 *     row(i)[ offset(k) .. offset(k) + C(dim,k) ).
 */
#ifndef GAAST_HIP_H
#define GAAST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GAAST_MAX_DIM 16   /* 2^16 components per full multivector; flat offsets fit 16 bits */
#define GAAST_MAX_INPUTS 64

typedef enum gaast_status {
    GAAST_OK = 0,
    GAAST_ERR_INVALID_PROGRAM = 1,  /* malformed flat program (reference: specialize.rs:104-117 asserts) */
    GAAST_ERR_MISSING_GRADE = 2,    /* reference panic: graded.rs:188,193 (grade absent from a buffer) */
    GAAST_ERR_UNIMPLEMENTED = 3,    /* reference panic: eval.rs:112-113 todo!() for exp / log; also returned by
                                     * gaast_hip_program_create for a program that is valid in the reference but
                                     * beyond this back end (e.g. a product whose operands exceed the LDS budget of
                                     * every kernel): such a program is refused whole, never half-evaluated */
    GAAST_ERR_HIP = 4,              /* a HIP runtime call failed; see gaast_hip_last_error() */
    GAAST_ERR_RCCL = 5,             /* an RCCL call failed / librccl could not be loaded / no communicator */
    GAAST_ERR_INVALID_ARGUMENT = 6,
    GAAST_ERR_NO_DEVICE = 7,        /* no gfx950 device / library not initialised */
    GAAST_ERR_OVERFLOW = 8          /* reference debug-build panic: eval.rs:90 `k - 1` with k == 0 */
} gaast_status;

typedef enum gaast_dtype {
    GAAST_F64 = 0, /* the reference's only value type (graded.rs:46) */
    GAAST_F32 = 1  /* extension used by the dense large-dimension configurations */
} gaast_dtype;

/* AstNode variants, reference src/ast/base_types.rs:8-30 (+ ScalarUnaryOp :84-88) */
typedef enum gaast_opcode {
    GAAST_OP_INPUT = 0,  /* GradedObj(T) */
    GAAST_OP_ADD = 1,    /* Addition */
    GAAST_OP_PRODUCT = 2,/* Product */
    GAAST_OP_NEG = 3,    /* Negation */
    GAAST_OP_EXP = 4,    /* Exponential (evaluation unimplemented upstream) */
    GAAST_OP_LOG = 5,    /* Logarithm   (evaluation unimplemented upstream) */
    GAAST_OP_PROJ = 6,   /* GradeProjection */
    GAAST_OP_REVERSE = 7,/* Reverse */
    GAAST_OP_GINVOL = 8, /* GradeInvolution */
    GAAST_OP_SINV = 9,   /* ScalarUnaryOp(Inversion) */
    GAAST_OP_SSQRT = 10  /* ScalarUnaryOp(SquareRoot) */
} gaast_opcode;

/* the five products of src/ast/expr.rs:180-197, for compact PRODUCT descriptors */
typedef enum gaast_product_kind {
    GAAST_PROD_EXPLICIT = -1, /* use the comp_muls list */
    GAAST_PROD_GEOMETRIC = 0,
    GAAST_PROD_OUTER = 1,
    GAAST_PROD_INNER = 2,
    GAAST_PROD_LCONTRACT = 3,
    GAAST_PROD_RCONTRACT = 4
} gaast_product_kind;

/* IndividualCompMul (base_types.rs:45-55) with Component{grade,index} (algebra.rs:87-91) */
typedef struct gaast_comp_mul {
    uint32_t left_grade, left_index;
    uint32_t right_grade, right_index;
    uint32_t result_grade, result_index;
    double coeff;
} gaast_comp_mul;

/* One GradedNode (base_types.rs:105-146): what eval.rs reads of it. */
typedef struct gaast_node_desc {
    int32_t opcode;              /* gaast_opcode */
    int32_t child0, child1;      /* indices of earlier nodes; -1 when unused */
    uint64_t minimal_grade_mask; /* GradedNode::grade_set(): bit k <=> grade k */
    int32_t vec_space_dim;       /* GradedNode::vec_space_dim() */
    int32_t input_slot;          /* OP_INPUT: which bound input */
    /* OP_PRODUCT, one of:
     *  - explicit: product_kind = GAAST_PROD_EXPLICIT and comp_muls[0..n_comp_muls) is
     *    Product.individual_comp_muls in the reference's order (specialize.rs:162-183);
     *  - compact: product_kind >= 0, comp_muls = NULL; the library regenerates the very same
     *    list from (kind, children's minimal sets, this node's minimal set, metric) by the
     *    rules of specialize.rs:132-183 / algebra.rs:73-83,199-246 -- or, for dense
     *    products, never materialises it. */
    int32_t product_kind;
    uint64_t n_comp_muls;
    const gaast_comp_mul *comp_muls;
} gaast_node_desc;

/* What eval.rs knows about a GradedObj(T): T::grade_set() and the slice lengths. */
typedef struct gaast_input_desc {
    uint64_t grade_mask;    /* Graded::grade_set() of the bound value */
    int32_t storage_dim;    /* slices have C(storage_dim,k) components; scalar literals use 0 (expr.rs:231-240) */
    int32_t is_const;       /* 1: value embedded below and shared by every batch item */
    const double *const_row;/* is_const: the row (grades ascending, concatenated) */
} gaast_input_desc;

#define GAAST_FLAG_DEBUG_OVERFLOW 0x1u /* reproduce the debug-build panic of eval.rs:90 (default: release) */
#define GAAST_FLAG_NO_FUSION 0x2u      /* one kernel per eval.rs arm, every operand materialised (A/B testing) */
/* Every sum in the reference's order with its three roundings per term (eval.rs:82): the result is the reference's, bit for bit.
 * WITHOUT the flag (the default) four things may differ, each within 4 eps sum |terms| per component (8: general metric) of the
 * reference's result OR closer to the exactly rounded sum than the reference's own sequential sum is:
 * dense products run on the re-ordered matrix-core kernels; long rows of a list product are summed in slices with fused
 * multiply-adds; a specialised small program whose arithmetic outweighs its bytes because an operand is SHARED by all items
 * (batch-1 input) contracts l * r + acc into one fused multiply-add; and the norm of a versor inverse / normalisation too big for
 * a fused program (n >= 9) is summed by the 64 lanes of a wave in parallel (the row is then read once).  Small programs over
 * batched operands and element-wise arms keep the reference's bits either way. */
#define GAAST_FLAG_EXACT_ORDER 0x4u
#define GAAST_FLAG_NO_MFMA 0x8u        /* dense products stay on the vector-FMA kernel (A/B testing) */
#define GAAST_FLAG_NO_JIT 0x10u        /* small programs run on the LDS interpreter kernel, not on hiprtc-specialised code */
/* OPT-IN, not the reference's algorithm: dense geometric products of a non-degenerate algebra (f32:
 * dimension 7..12, f32 and f64) go through the 2^m x 2^m complex matrix representation (at n = 12: 21x fewer
 * multiply-adds, all on the matrix cores).  Equal to eval.rs:61-86 in exact arithmetic; the error is bounded norm-wise,
 * |err_S| <= 64 eps(dtype) |A|_2 |B|_2, not per component.  Never selected without this flag. */
#define GAAST_FLAG_SPINOR_GEMM 0x20u
/* Debug hooks (explicit flags, no environment variables): */
#define GAAST_FLAG_DEBUG_JIT_FAILS 0x40u      /* treat every run-time (hiprtc) compilation as failed: exercises the fallbacks */
#define GAAST_FLAG_DEBUG_KEEP_JIT_SOURCE 0x80u /* keep the generated kernel source: gaast_hip_program_jit_source() */
/* OPT-IN extension with no reference behaviour (eval.rs:112-113 is todo!()): evaluate Exponential / Logarithm with the
 * semantics the grade rules imply (grade_set.rs:181-197), see DESIGN.md.  Without the flag such programs report
 * GAAST_ERR_UNIMPLEMENTED exactly where the reference panics. */
#define GAAST_FLAG_EXP_LOG 0x100u
#define GAAST_FLAG_DEBUG_LDS_12K 0x400u /* hiprtc-specialised kernels: 12 KiB instead of 10 KiB of LDS per wave for the row transposition (A/B testing) */
#define GAAST_FLAG_DEBUG_NO_CHAIN 0x800u /* a sparse product that only feeds a dense product stays a launch of its own (default: evaluated in the dense kernel's LDS staging; A/B testing) */
/* (bits 30 and 31 are reserved for the library's own use: a plan rebuilt after a trial compilation carries one) */
#define GAAST_FLAG_DEBUG_FAIL_EVAL 0x1000u /* every evaluation of this program fails with GAAST_ERR_HIP before its first launch: exercises the failure path of gaast_hip_eval_gather on ONE rank */
#define GAAST_FLAG_NO_COALESCE 0x200u  /* hiprtc-specialised kernels: every lane reads / writes its own row (no LDS-transposed coalesced row I/O; A/B testing) */

typedef struct gaast_program_desc {
    int32_t vec_space_dim;      /* n */
    const double *metric_diag;  /* n squares of the base vectors (MetricAlgebra::base_vec_dot(i,i)) */
    int32_t dtype;              /* gaast_dtype of every buffer of this program */
    int32_t n_nodes;
    const gaast_node_desc *nodes; /* post-order: children before parents */
    int32_t root;               /* SpecializedAst::root_id() */
    int32_t n_inputs;
    const gaast_input_desc *inputs;
    uint32_t flags;
} gaast_program_desc;

typedef struct gaast_hip_program_s *gaast_hip_program_t;
typedef struct gaast_hip_mv_s *gaast_hip_mv_t;

/* ---- runtime ------------------------------------------------------------------------- */
/* Selects the GPU this process drives (device_ids[0]; n_dev must be 1: one rank per GPU -- a multi-GPU job is one
 * process per GPU joined by gaast_hip_comm_init below). */
int gaast_hip_init(const int *device_ids, int n_dev);
int gaast_hip_shutdown(void);
/* Launch on an existing HIP stream (e.g. torch's current stream); NULL = default stream. */
int gaast_hip_set_stream(void *hip_stream);
int gaast_hip_synchronize(void);
const char *gaast_hip_last_error(void);
const char *gaast_hip_version(void);

/* ---- SpecializedAst on the device ------------------------------------------------------ */
int gaast_hip_program_create(const gaast_program_desc *desc, gaast_hip_program_t *out);
int gaast_hip_program_destroy(gaast_hip_program_t prog);
/* grade mask / row length of the root result (root.minimal_grade_set) */
int gaast_hip_program_output_info(gaast_hip_program_t prog, uint64_t *grade_mask, int64_t *row_len);
/* number of kernel launches one eval issues, and a one-line description of launch `i` */
int gaast_hip_program_num_launches(gaast_hip_program_t prog);
const char *gaast_hip_program_launch_name(gaast_hip_program_t prog, int i);
/* GAAST_FLAG_EXP_LOG extension: how many items, since the last call, had an exp / log operand outside the domain (a
 * k-vector whose square is not scalar: |<B B>_{not 0}|^2 > 2^-40 (sum B_i^2)^2); their results are the closed form applied
 * to <B B>_0 regardless.  Synchronises the library stream; resets the counter.  0 for programs without exp / log. */
int gaast_hip_program_domain_errors(gaast_hip_program_t prog, int64_t *count);
/* GAAST_FLAG_DEBUG_KEEP_JIT_SOURCE: the HIP source generated for this program ("" if none). */
const char *gaast_hip_program_jit_source(gaast_hip_program_t prog);

/* ---- GradedDataMut on the device (graded.rs:51-79) ------------------------------------- */
/* init_null_mv(dim, gs) for `batch` items: zero-filled rows (graded.rs:195-201). */
int gaast_hip_mv_alloc(int dim, uint64_t grade_mask, int64_t batch, int dtype, gaast_hip_mv_t *out);
/* Same layout over caller-owned device memory (e.g. a torch tensor); row_stride in elements. */
int gaast_hip_mv_wrap(void *device_ptr, int dim, uint64_t grade_mask, int64_t batch, int dtype,
                      int64_t row_stride, gaast_hip_mv_t *out);
int gaast_hip_mv_free(gaast_hip_mv_t mv);
int gaast_hip_mv_info(gaast_hip_mv_t mv, int *dim, uint64_t *grade_mask, int64_t *batch, int *dtype,
                      int64_t *row_len, int64_t *row_stride, void **device_ptr);
/* grade_slice_mut(k) of every item <- host[batch][C(dim,k)] (values in the mv's dtype). Synchronous. */
int gaast_hip_mv_upload(gaast_hip_mv_t mv, int grade, const void *host, int64_t count);
/* host[batch][C(dim,k)] <- grade_slice(k) of every item. Synchronous. */
int gaast_hip_mv_download(gaast_hip_mv_t mv, int grade, void *host, int64_t count);
/* whole rows at once: host[batch][row_len] */
int gaast_hip_mv_upload_rows(gaast_hip_mv_t mv, const void *host, int64_t count);
int gaast_hip_mv_download_rows(gaast_hip_mv_t mv, void *host, int64_t count);
int gaast_hip_mv_zero(gaast_hip_mv_t mv);

/* ---- SpecializedAst::eval (eval.rs:12-19), batched --------------------------------------- */
/*
 * Evaluates `prog` once per batch item.  inputs[slot] binds the GradedObj of that slot
 * (const slots may be NULL); an input whose batch is 1 is shared by all items.  `out` must
 * have the root's grade mask (gaast_hip_program_output_info) and `batch` items; it is
 * overwritten (the reference returns a fresh R).
 */
int gaast_hip_eval(gaast_hip_program_t prog, const gaast_hip_mv_t *inputs, int n_inputs,
                   int64_t batch, gaast_hip_mv_t out);

/* ---- multi-GPU: one rank per GPU, batch sharded by item ------------------------------------ */
/*
 * The reference evaluates one input set per eval() and keeps no cross-item state (eval.rs:16), so a batch
 * shards into contiguous item ranges with no data-path collective; the ONLY exchange is the gather of the
 * result rows to one rank.  The library does that itself over RCCL (xGMI), so that a non-Python host (the
 * Rust shim) needs nothing else: rank 0 calls gaast_hip_comm_unique_id and ships the 128 bytes to the other
 * ranks by any channel (file, socket, MPI, torch.distributed); every rank then calls gaast_hip_comm_init
 * (collective).  librccl is loaded on first use (dlopen), not at link time.
 */
#define GAAST_COMM_ID_BYTES 128
/* Which shared object provides the nccl* entry points the gather uses (ncclGetUniqueId, ncclCommInitRank,
 * ncclCommDestroy, ncclSend, ncclRecv, ncclAllReduce, ncclGroupStart, ncclGroupEnd, ncclGetErrorString): NULL = the
 * system's librccl (default: soname lookup, then /opt/rocm/lib).  For hosts that ship their own RCCL build and for the
 * test transport that lets several ranks share one GPU (tests/cpp/rccl_stub.c).  Call before the first
 * gaast_hip_comm_* use; GAAST_ERR_RCCL once another library has been loaded. */
int gaast_hip_comm_set_library(const char *path);
int gaast_hip_comm_unique_id(void *id_out);
int gaast_hip_comm_init(const void *id, int rank, int world);
int gaast_hip_comm_destroy(void);
/* rank / world of the communicator (GAAST_ERR_RCCL if there is none) */
int gaast_hip_comm_info(int *rank, int *world);
/* all-reduce(sum) of one 1 per rank over the communicator: how many ranks RCCL actually joined (synchronous) */
int gaast_hip_comm_count_ranks(int *n_ranks);
/*
 * Gather result rows to `root`: rank r contributes the first counts[r] rows of `local` (contiguous rows:
 * row_stride == row_len); on root, `gathered` receives them in rank order (item order of the global batch),
 * rank r's rows starting at row counts[0] + ... + counts[r-1].  `gathered` is ignored on the other ranks (may
 * be NULL).  The root receives from all peers concurrently (one direct peer-to-peer transfer per xGMI link,
 * not a ring).  Asynchronous on the library streams like gaast_hip_eval.
 */
int gaast_hip_gather_rows(gaast_hip_mv_t local, gaast_hip_mv_t gathered, const int64_t *counts, int root);
/*
 * gaast_hip_eval + gather, overlapped: the local batch (counts[rank] items) is evaluated in `n_chunks`
 * contiguous chunks and chunk k travels to `root` on a second stream while chunk k+1 is being computed.
 * Equivalent to gaast_hip_eval(prog, inputs, n_inputs, counts[rank], out) followed by
 * gaast_hip_gather_rows(out, gathered, counts, root).  `out` may be the rows of `gathered` that belong to the
 * root itself (same device memory): the local copy is then skipped.
 * FAILURE IS COLLECTIVE: a rank whose evaluation fails after the argument checks (a launch error) still posts
 * its transfers, so that no peer is left waiting in a receive, and before returning every rank all-reduces an
 * error flag over the communicator: if ANY rank failed, EVERY rank returns non-zero (the failing rank its own
 * status, the others GAAST_ERR_RCCL "another rank failed") and the contents of `gathered` are unspecified.  The
 * flag exchange makes this entry point synchronous with the communicator's stream: when it returns GAAST_OK
 * the gathered rows have arrived.  (Argument errors are detected before any transfer and are the caller's to
 * keep consistent across ranks: the same counts / root / n_chunks everywhere.)
 */
int gaast_hip_eval_gather(gaast_hip_program_t prog, const gaast_hip_mv_t *inputs, int n_inputs,
                          gaast_hip_mv_t out, gaast_hip_mv_t gathered, const int64_t *counts, int root,
                          int n_chunks);

#ifdef __cplusplus
}
#endif
#endif /* GAAST_HIP_H */
