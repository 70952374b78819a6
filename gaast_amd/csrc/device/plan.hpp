// Lowering of a flat SpecializedAst program into a launch plan that reproduces the control
// flow of the reference's interpreter (src/eval.rs:12-115) arm by arm.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../common/algebra.hpp"
#include "../common/comp_mul_table.hpp"
#include "gaast_hip.h"

// internal (not in include/gaast_hip.h): the one-item-per-thread specialised kernel is limited to slabs of 160 / 200 elements -- set by
// program_create_impl when a bigger slab's trial compilation needs more than half of a SIMD's registers
#define GAAST_FLAG_INTERNAL_SMALL_REG_SLAB 0x40000000u

namespace gaast {

enum class BufKind : int { NODE = 0, INPUT = 1, OUT = 2 };

struct BufRef {
    BufKind kind = BufKind::NODE;
    int idx = -1;  // NODE: cache-buffer id; INPUT: slot
};

struct Layout {  // how a graded row is laid out
    int dim = 0;
    uint64_t mask = 0;
    int64_t row_len = 0;
    int64_t offset(int k) const { return grade_offset(dim, mask, k); }
    int64_t grade_len(int k) const { return int64_t(n_choose_k(uint64_t(dim), uint64_t(k))); }
};

inline Layout make_layout(int dim, uint64_t mask) {
    Layout l;
    l.dim = dim;
    l.mask = mask;
    l.row_len = row_len_of(dim, mask);
    return l;
}

struct Step {
    enum Kind { ZERO, AXPY, FLIP, SUNARY, PRODUCT_CSR, PRODUCT_DENSE, FUSED, EXPLOG, REDUCE_SCALE, ELEMENTWISE } kind = ZERO;
    BufRef res, a, b;
    std::string name;
    std::string hip_kernel;        // the HIP kernel (template and arguments) prepare_step picked: appended to the launch label
    // host images of the tables (uploaded once at program_create)
    std::vector<uint32_t> u32_a;   // AXPY map | FLIP offsets | CSR row_start | DENSE left_map
    std::vector<uint32_t> u32_b;   // CSR row_out | DENSE right_map
    std::vector<uint32_t> u32_c;   // CSR entries
    std::vector<double> coeff;     // CSR coefficients (converted to the program dtype on upload)
    std::vector<double> coeff_host; // FUSED: the general coefficients, passed by value at launch
    std::vector<int32_t> i32_a;    // DENSE out_map
    int sunary_op = 0, sunary_off = 0;
    // EXPLOG (GAAST_FLAG_EXP_LOG extension): res += exp(arg) / log(arg), arg = buffer `a` holding a k-vector (log: + grade 0).
    //   coeff[i] = e_i e_i (blade squares, component order); pairs of distinct blades that commute, grouped by the blade T of
    //   their product in ascending T: u32_a = row starts, u32_c = i | j << 16, coeff_b = 2 e_i e_j -- the domain check.
    int explog_op = 0;                       // 0 exp, 1 log
    int explog_m = 0, explog_mres = 0;       // components of the k-vector / of them that res holds (zip)
    int explog_arg_k = 0, explog_arg_0 = -1; // offsets in the operand row (grade k; grade 0 for log, -1 = absent)
    int explog_res_k = -1, explog_res_0 = -1;// offsets in the result row, -1 = not produced
    std::vector<double> coeff_b;
    void* d_coeff_b = nullptr;
    // DENSE with a general diagonal metric (entries other than +-1 / 0): the kernels run in the rescaled basis
    // f_i = e_i / sqrt|g_i|: coeff = w_S per loaded left component, coeff_b = per loaded right component, coeff_c = 1 / w_T per
    // blade of the permuted basis (the index of i32_a)
    std::vector<double> coeff_c;
    void* d_coeff_c = nullptr;
    int scaled = 0;
    void* d_domain = nullptr;                // the program's domain-error counter (device, not owned by the step)
    int ell_bytes = 0;             // ... and the offsets of its entries are byte offsets
    int ell_width = 0;             // PRODUCT_CSR with rows of one length and +-1 coefficients: u32_c is [term][row], sign in bit 31
    int canon_a = 0, canon_b = 0;
    int beta = 1;
    int left_full = 0, right_full = 0, out_full = 0;
    int left_contig = 0, right_contig = 0;
    int left_signs = 0, out_signs = 0;   // DENSE: some left_map word negates / some out_map word carries a reordering sign
    uint32_t neg_hi = 0, zero_hi = 0;
    uint32_t neg_lo = 0;   // lo basis vectors (of the permuted basis) that square to -1
    int neg_lo_all = 0;    // vector-FMA kernel: the NEGLO instantiation (all four lo vectors square to -1)
    int degenerate = 0;
    int dense_n = 0;     // DENSE: dimension of the algebra the kernel runs in (the program's n, or n - 1: parity-pure operands)
    // DENSE, chained (plan.cpp: chain_sparse_into_dense): the LEFT operand is the result of a short comp-mul list over two other
    // rows (eval.rs:61-86 with a cached operand that nothing else reads: R X in R X ~R).  The list is evaluated in LDS while
    // the dense kernel stages its operands -- reference order, same roundings -- and the intermediate never goes through HBM.
    int chained = 0;
    BufRef pre_a, pre_b;                 // the list's operand rows
    int pre_canon_a = 0, pre_canon_b = 0;
    std::vector<uint32_t> pre_row_start; // rows + 1
    std::vector<uint32_t> pre_entries;   // left offset | right offset << 16
    std::vector<double> pre_coeff;
    std::vector<uint32_t> pre_row_map;   // per row: image position << 16 | negate << 31 (the left_map word of the component the row produces)
    std::vector<double> pre_row_scale;   // per row (rescaled basis), else empty
    int pre_left_len = 0, pre_right_len = 0;
    int pre_width = 0;                   // > 0: rows of one length with +-1 coefficients: pre_entries is [term][row], sign in bit 31
    size_t pre_scratch_off = 0;          // bytes: where the list's operand rows sit in the kernel's LDS (after its images)
    void* d_pre_row_start = nullptr;
    void* d_pre_entries = nullptr;
    void* d_pre_coeff = nullptr;
    void* d_pre_row_map = nullptr;
    void* d_pre_row_scale = nullptr;
    // PRODUCT_CSR (ELL form), list chain (plan.cpp: chain_list_into_list): one operand of this list is the result of ANOTHER list that
    // nothing else reads -- (R X) ~R projected on a grade.  Both lists run in one k_product_ell_chain launch, the mid row stays in
    // LDS.  Reuses pre_a / pre_b / pre_canon_* / pre_left_len / pre_right_len / pre_entries ([term][row] words of the first list) /
    // pre_row_map (element offset of each of its rows in the mid row) / pre_width.
    int list_chain = 0;                  // 1: the mid row is this list's left operand, 2: its right operand
    int chain_alias = 0;                 // this list's other operand: 0 = a row of its own, 1 = the first list's left row, 2 = its right row
    int chain_mid_len = 0, chain_canon_mid = 0, chain_covered = 0;
    int chain_ipb = 0, chain_item_stride = 0, chain_ent2_lds = 0;   // (bytes of this list's words kept in LDS, or 0)
    // ... specialised per program through hiprtc (plan.cpp: chain_jit_source; round 4): lane = (row, item) in BOTH lists with the items
    // of a workgroup fastest, so that the 32 lanes of an LDS access read one row's operand of 32 different items (odd item stride:
    // no bank conflict); entries carry byte offsets from the item's base (list 1: the sign is folded into a negated image of the
    // smaller operand); compile-time widths, lengths and strides.  The generic k_product_ell_chain stays as the fallback.
    int chain_jit = 0;                   // 1: chain_jit_source is to be compiled (runtime.hip), 2: compiled and in charge
    std::string chain_jit_source;
    std::vector<uint32_t> cj_ent1, cj_pos1, cj_ent2, cj_out2;
    void* d_cj_ent1 = nullptr;
    void* d_cj_pos1 = nullptr;
    void* d_cj_ent2 = nullptr;
    void* d_cj_out2 = nullptr;
    int list_jit = 0;                    // the specialised kernel runs a SINGLE list (few long rows): plan.cpp: jit_long_row_lists
    int fold_prev = 0;                   // ... and, once compiled, also the covering copy_grades_from step right before it (pre_a = its source)
    int cj_ipb = 0, cj_threads = 0;
    size_t cj_lds = 0;
    int cj_split = 1;                            // slices per row of list 2 (> 1: re-ordered sums, tolerance mode; 1 with GAAST_FLAG_EXACT_ORDER)
    int cj_xreg = 0;                             // tolerance mode: list 1's right operand in registers, its table re-ordered by right index (word: left offset | sign << 31)
    int cj_sorted[3] = {0, 0, 0};               // tolerance mode, sign-sorted list 2: plus / minus terms per (row, slice), byte offset of the item's zero element
    int cj_fmt[2] = {0, 0};                     // words per row of list 1's table; list 2's entries: 2 = wide (8 bytes: offsets, then the sign bit), 3 / 4 = sign-sorted clean words (LDS / global), else narrow
    int cj_layout[7] = {0, 0, 0, 0, 0, 0, 0};   // an item in LDS, elements: offsets of l1, r1, the negated image, mid, r2 (-1: aliased); item stride; negated image is of the left operand
    int use_mfma = 0;
    int use_mfma16 = 0;  // k_gp_mfma16x4<T> (lo = 4 bits, one item per workgroup): f64 n = 8 ... 12, f32 n = 8, 9
    int use_mfma16d = 0; // (same; kept apart from use_mfma16 since round 2's four-items-per-instruction kernel shared the first)
    // REDUCE_SCALE (plan.cpp: fuse_reduce_scale): a product whose result is ONE scalar component (a single long row: norm_sq), an
    // optional ScalarUnaryOp on it, and a product of one-term rows that multiplies another row by that scalar -- the versor inverse
    // a.rev() * a.norm_sq().sinv() and normalisations, where the rows no longer fit a fused slab (n >= 9) -- in ONE launch, one wave per
    // item: u32_a / coeff = the reduction's terms in the reference's order (left | right << 16, coefficient), a / b = its operands;
    // u32_b / coeff_b = the scaling's rows (operand offset | result offset << 16, coefficient), pre_a = its row operand.
    // ELEMENTWISE (plan.cpp: fuse_elementwise_runs): a run of AXPY / FLIP steps on one buffer (and, when that buffer is only the operand
    // of a product of one-term rows with a scalar, that product too) in one pass: u32_a = [n_ops][n_comp] statement words, u32_b = the
    // components' offsets in the run's buffer, ew_src = the source buffers; with the scaling epilogue: u32_c = result offsets, coeff =
    // coefficients, b = the scalar operand (1-component row), res = the product's result.
    std::vector<BufRef> ew_src;
    int ew_ops = 0, ew_load_first = 0, ew_scale = 0, ew_scalar_off = 0, ew_canon_v = 0, ew_canon_s = 0, ew_s_is_left = 0;
    int rs_op = 0;                       // 0: none, 1: 1 / s, 2: sqrt(s)  (eval.rs:103-110)
    int rs_canon_s = 0;                  // the scalar is re-read as a product operand: 0.0 + s
    int rs_wave = 0;                     // > 0: tolerance mode, the reduction is a signed sum of squares of the row that is scaled: 16-byte
                                         // pieces of the row per lane for k_reduce_scale_wave (u32_c = the lanes' sign words)
    int use_mfma6 = 0;   // k_gp_mfma6<T> (n = 6: four 16x16x4 instructions per item, the two top vectors split over the tile's rows and columns)
    int use_mfma7 = 0;   // k_gp_mfma7<T> (n = 7: lo = 3 bits, the top vector split over the two sides of the 16 x 16 tile)
    int mfma16_quads = 0; // ... in f32: the B image in the 16-byte-quad layout
    int mfma32_pairs = 0;  // k_gp_mfma32p (image-pair form, f32, n = 10 ... 13) instead of k_gp_mfma32
    int spinor_lam_bit = -1, spinor_has_alpha = 0;  // index basis of the matrix-representation kernels: spinor_basis.hpp
    int use_spinor = 0;  // opt-in matrix-representation kernel (GAAST_FLAG_SPINOR_GEMM): log2 of the matrix size, or 0
    uint64_t n_entries = 0;  // comp-mul count this step stands for
    // FUSED: the whole plan as one micro-op stream over per-item LDS slabs (u32_a = the stream)
    struct FusedInput {
        int slot, base, canon;
    };
    std::vector<FusedInput> fused_inputs;
    int fused_slab = 0, fused_out_base = 0, fused_zero_slot = 0;
    int jit_persistent = 0;   // > 0: the specialised kernel loops over groups (persistent workgroups): this many are resident per CU
    int jit_items = 0;        // items per workgroup of the specialised kernel when it is not one per thread (the slab-in-LDS form: 64)
    int jit_threads = 256;    // workgroup size of the specialised kernel (64: one wave per workgroup, coalesced row I/O through LDS)
    int fused_jit_only = 0;   // the slab is too big for the LDS interpreter: runs only as the hiprtc-specialised kernel
    int jit_reg_trial = 0;    // one item per thread with a slab beyond 160 / 200 elements: kept only if the compiled kernel leaves two
                              // waves per SIMD (runtime.hip: program_create_impl); else the plan is rebuilt with the slabs in LDS
    std::string jit_source;   // FUSED: the plan as straight-line HIP (compiled with hiprtc at program_create)
    void* jit_module = nullptr;
    void* jit_function = nullptr;
    std::vector<char> jit_code;   // the hiprtc code object jit_module was loaded from: kept until the module is unloaded
    // the same source compiled with floating-point contraction (l * r + acc as ONE fused multiply-add: fewer roundings than the
    // reference, so within the tolerance contract but not its bits): built only without GAAST_FLAG_EXACT_ORDER, launched only when
    // an item's arithmetic outweighs its bytes (runtime.hip: run_jit -- in practice: operands shared by all items)
    void* jit_module_fma = nullptr;
    void* jit_function_fma = nullptr;
    std::vector<char> jit_code_fma;
    // launch configuration, fixed once at gaast_hip_program_create (runtime.hip: prepare_step): kernel, block
    // size, dynamic LDS, persistent-grid size.  ELL products pick kern[log2(items per pass)] by batch.
    const void* kern[4] = {nullptr, nullptr, nullptr, nullptr};
    int threads = 0;
    size_t lds = 0;            // bytes per launch (ELL / CSR: per staged item)
    int max_items = 0;         // ELL / CSR: items per workgroup when the batch allows
    int items_per_block = 0;   // dense kernels
    int blocks_per_cu = 0;     // persistent kernels: resident workgroups per CU
    // device copies
    void* d_a = nullptr;
    void* d_b = nullptr;
    void* d_c = nullptr;
    void* d_coeff = nullptr;
    void* d_i32 = nullptr;
};

// limits of this back end (gfx950): a product whose staged operands exceed the LDS of a CU, or whose comp-mul list
// exceeds the table budget, is valid in the reference but refused by gaast_hip_program_create (UNIMPLEMENTED)
constexpr size_t kLdsBytes = 160 * 1024;
#ifndef GAAST_INTERP_BUDGET_KB
#define GAAST_INTERP_BUDGET_KB 144   /* (build switch for A/B runs) */
#endif
constexpr size_t kInterpLdsBytes = size_t(GAAST_INTERP_BUDGET_KB) * 1024;   // slabs of the 64 items of a k_ast_fused workgroup
constexpr uint64_t kMaxListEntries = uint64_t(1) << 27;

struct Plan {
    int n = 0;
    int dtype = GAAST_F64;
    uint32_t flags = 0;
    std::vector<double> metric;
    std::vector<gaast_input_desc> inputs;
    std::vector<std::vector<double>> const_rows;
    std::vector<Layout> input_layouts;
    std::vector<Layout> node_buffers;  // cache buffers other than the root's
    std::vector<char> node_dead;       // ... that a chained product made unnecessary (never allocated)
    Layout out_layout;
    std::vector<Step> steps;
    int error = GAAST_OK;              // what the reference would have panicked with, at eval
    std::string error_msg;
    int has_explog = 0;                // some step evaluates exp / log: the program owns a domain-error counter
    std::string unsupported;           // non-empty: valid in the reference, beyond this back end (program_create fails)
    std::vector<char> slot_used;       // input slots some launch reads (the others may stay unbound)
    std::string jit_source_kept;       // GAAST_FLAG_DEBUG_KEEP_JIT_SOURCE
};

// Throws std::runtime_error (-> GAAST_ERR_INVALID_PROGRAM) on malformed input.
void build_plan(const gaast_program_desc& desc, Plan& plan);

// Micro-op encoding shared by the plan builder and k_ast_fused (see kernels.hip.hpp).
namespace uop {
enum : uint32_t { LINE_MACS = 0, LINE_MISC = 1, LINE_NOP = 2, LINE_MACS_GEN = 3, LINE_MACS_CNT = 4, ADD = 3, NEG = 4, ZERO = 5, INV = 6, SQRT = 7, COPY = 8 };
constexpr int MAX_GENERAL_COEFFS = 6;
constexpr int MAX_INPUTS = 8;
constexpr int GROUPS = 8;  // waves per workgroup of k_ast_fused (FUSED_GROUPS)
}  // namespace uop

}  // namespace gaast
