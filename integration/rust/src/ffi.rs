//! Raw bindings to `libgaast_hip.so` (C ABI of `include/gaast_hip.h`).  NOT COMPILED in this repository.
#![allow(dead_code)]
use std::os::raw::{c_char, c_int, c_void};

pub const GAAST_OK: c_int = 0;
pub const GAAST_F64: c_int = 0;
pub const GAAST_F32: c_int = 1;
pub const GAAST_PROD_EXPLICIT: i32 = -1;
pub const GAAST_PROD_GEOMETRIC: i32 = 0;
pub const GAAST_COMM_ID_BYTES: usize = 128;

// gaast_opcode, in the order of AstNode's variants (base_types.rs:8-30) + ScalarUnaryOp (:84-88)
pub const OP_INPUT: i32 = 0;
pub const OP_ADD: i32 = 1;
pub const OP_PRODUCT: i32 = 2;
pub const OP_NEG: i32 = 3;
pub const OP_EXP: i32 = 4;
pub const OP_LOG: i32 = 5;
pub const OP_PROJ: i32 = 6;
pub const OP_REVERSE: i32 = 7;
pub const OP_GINVOL: i32 = 8;
pub const OP_SINV: i32 = 9;
pub const OP_SSQRT: i32 = 10;

pub const FLAG_EXACT_ORDER: u32 = 0x4; // bit-exact f64 sums even for dense products
pub const FLAG_SPINOR_GEMM: u32 = 0x20; // opt-in matrix-representation products (norm-wise error bound)
pub const FLAG_EXP_LOG: u32 = 0x100; // opt-in extension: evaluate Exponential / Logarithm (todo!() upstream, eval.rs:112-113)

#[repr(C)]
pub struct GaastCompMul {
    pub left_grade: u32,
    pub left_index: u32,
    pub right_grade: u32,
    pub right_index: u32,
    pub result_grade: u32,
    pub result_index: u32,
    pub coeff: f64,
}

#[repr(C)]
pub struct GaastNodeDesc {
    pub opcode: i32,
    pub child0: i32,
    pub child1: i32,
    pub minimal_grade_mask: u64,
    pub vec_space_dim: i32,
    pub input_slot: i32,
    pub product_kind: i32,
    pub n_comp_muls: u64,
    pub comp_muls: *const GaastCompMul,
}

#[repr(C)]
pub struct GaastInputDesc {
    pub grade_mask: u64,
    pub storage_dim: i32,
    pub is_const: i32,
    pub const_row: *const f64,
}

#[repr(C)]
pub struct GaastProgramDesc {
    pub vec_space_dim: i32,
    pub metric_diag: *const f64,
    pub dtype: i32,
    pub n_nodes: i32,
    pub nodes: *const GaastNodeDesc,
    pub root: i32,
    pub n_inputs: i32,
    pub inputs: *const GaastInputDesc,
    pub flags: u32,
}

// Layout of the structs above as the C compiler lays them out (tests/cpp/abi_layout.c -> abi_layout.json, checked
// against the header and the Python mirrors by tests/test_abi_layout.py).  A drift fails the Rust build here.
const _: () = {
    use std::mem::{align_of, size_of};
    assert!(size_of::<GaastCompMul>() == 32 && align_of::<GaastCompMul>() == 8);
    assert!(size_of::<GaastNodeDesc>() == 56);
    assert!(size_of::<GaastInputDesc>() == 24);
    assert!(size_of::<GaastProgramDesc>() == 56);
};
pub const LAYOUT_COMP_MUL_COEFF_OFFSET: usize = 24;
pub const LAYOUT_NODE_DESC_MINIMAL_GRADE_MASK_OFFSET: usize = 16;
pub const LAYOUT_NODE_DESC_N_COMP_MULS_OFFSET: usize = 40;
pub const LAYOUT_NODE_DESC_COMP_MULS_OFFSET: usize = 48;
pub const LAYOUT_INPUT_DESC_CONST_ROW_OFFSET: usize = 16;
pub const LAYOUT_PROGRAM_DESC_NODES_OFFSET: usize = 24;
pub const LAYOUT_PROGRAM_DESC_INPUTS_OFFSET: usize = 40;
pub const LAYOUT_PROGRAM_DESC_FLAGS_OFFSET: usize = 48;

pub type Program = *mut c_void;
pub type Mv = *mut c_void;

#[link(name = "gaast_hip")]
extern "C" {
    pub fn gaast_hip_init(device_ids: *const c_int, n_dev: c_int) -> c_int;
    pub fn gaast_hip_shutdown() -> c_int;
    pub fn gaast_hip_synchronize() -> c_int;
    pub fn gaast_hip_last_error() -> *const c_char;
    pub fn gaast_hip_program_create(desc: *const GaastProgramDesc, out: *mut Program) -> c_int;
    pub fn gaast_hip_program_destroy(p: Program) -> c_int;
    pub fn gaast_hip_program_output_info(p: Program, mask: *mut u64, row_len: *mut i64) -> c_int;
    pub fn gaast_hip_program_domain_errors(p: Program, count: *mut i64) -> c_int;
    pub fn gaast_hip_mv_alloc(dim: c_int, mask: u64, batch: i64, dtype: c_int, out: *mut Mv) -> c_int;
    pub fn gaast_hip_mv_free(m: Mv) -> c_int;
    pub fn gaast_hip_mv_upload(m: Mv, grade: c_int, host: *const c_void, count: i64) -> c_int;
    pub fn gaast_hip_mv_download(m: Mv, grade: c_int, host: *mut c_void, count: i64) -> c_int;
    pub fn gaast_hip_eval(p: Program, inputs: *const Mv, n_inputs: c_int, batch: i64, out: Mv) -> c_int;
    // multi-GPU (one process per GPU): the gather of result rows over RCCL, see include/gaast_hip.h
    /// which shared object provides the nccl* entry points (NULL = the system's librccl); before the first comm call
    pub fn gaast_hip_comm_set_library(path: *const std::os::raw::c_char) -> c_int;
    pub fn gaast_hip_comm_unique_id(id_out: *mut c_void) -> c_int;
    pub fn gaast_hip_comm_init(id: *const c_void, rank: c_int, world: c_int) -> c_int;
    pub fn gaast_hip_comm_destroy() -> c_int;
    pub fn gaast_hip_comm_info(rank: *mut c_int, world: *mut c_int) -> c_int;
    pub fn gaast_hip_comm_count_ranks(n_ranks: *mut c_int) -> c_int;
    pub fn gaast_hip_gather_rows(local: Mv, gathered: Mv, counts: *const i64, root: c_int) -> c_int;
    pub fn gaast_hip_eval_gather(p: Program, inputs: *const Mv, n_inputs: c_int, out: Mv, gathered: Mv,
                                 counts: *const i64, root: c_int, n_chunks: c_int) -> c_int;
}

/// Turns a non-zero status into the panic the reference would have raised.
pub fn check(status: c_int, what: &str) {
    if status != GAAST_OK {
        let msg = unsafe { std::ffi::CStr::from_ptr(gaast_hip_last_error()) };
        panic!("gaast_hip: {} failed with status {}: {}", what, status, msg.to_string_lossy());
    }
}
