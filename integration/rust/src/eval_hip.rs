//! `SpecializedAst::eval` on an MI355X through `libgaast_hip.so`.  Replaces `src/eval.rs` when the cargo
//! feature `hip` is on.  NOT COMPILED in this repository (no Rust toolchain in the build image); it only
//! uses the reference's public API, and the C side of every call is covered by this repository's tests.
use std::collections::HashMap;
use std::os::raw::c_void;

use crate::algebra::{n_choose_k, Algebra, MetricAlgebra};
use crate::ffi::*;
use crate::{ast::*, graded::*, Grade, GradeSet};
use AstNode as N;

fn mask_of(gs: &GradeSet) -> u64 {
    gs.iter().fold(0u64, |m, k| m | (1u64 << k))
}

/// The dimension d with C(d, k) == grade_slice(k).len() (scalar literals are built with d = 0,
/// `expr.rs:231-240`); `fallback` (the node's `vec_space_dim`) when only grade 0 is present.
fn storage_dim<T: GradedData>(x: &T, fallback: usize) -> usize {
    for k in x.grade_set().iter() {
        if k == 0 {
            continue;
        }
        let len = x.grade_slice(k).len();
        return (k..=64).find(|d| n_choose_k(*d, k) == len).expect("slice length is not a binomial");
    }
    if x.grade_set().contains(0) && fallback > 0 { fallback } else { 0 }
}

/// The flat program of one `SpecializedAst`, its device handle, and the order of its inputs.
pub struct HipProgram<'a, T> {
    handle: Program,
    out_mask: u64,
    out_dim: usize,
    inputs: Vec<&'a T>,
    input_descs: Vec<(u64, usize)>, // (grade mask, storage dim) per slot
}

impl<'a, T> Drop for HipProgram<'a, T> {
    fn drop(&mut self) {
        unsafe { gaast_hip_program_destroy(self.handle) };
    }
}

impl<T: GradedData + std::fmt::Debug> SpecializedAst<T> {
    fn post_order(&self, id: NodeId, order: &mut Vec<NodeId>, index: &mut HashMap<NodeId, i32>) {
        if index.contains_key(&id) {
            return; // shared sub-expression: one node, as in the reference's cache (eval.rs:21-33)
        }
        match self.get_node(id).ast_node() {
            N::GradedObj(_) => {}
            N::Addition(l, r) => {
                self.post_order(*l, order, index);
                self.post_order(*r, order, index);
            }
            N::Product(p) => {
                self.post_order(p.left_expr, order, index);
                self.post_order(p.right_expr, order, index);
            }
            N::Negation(e) | N::Exponential(e) | N::Logarithm(e) | N::GradeProjection(e)
            | N::Reverse(e) | N::GradeInvolution(e) | N::ScalarUnaryOp(_, e) => {
                self.post_order(*e, order, index);
            }
        }
        index.insert(id, order.len() as i32);
        order.push(id);
    }

    /// Phases 1-3 stay as they are; this flattens their result (once per `SpecializedAst`).
    /// `metric` = the diagonal of the algebra given to `specialize` (`MetricAlgebra::base_vec_dot(i, i)`).
    pub fn to_hip(&self, alg: &impl MetricAlgebra, flags: u32) -> HipProgram<'_, T> {
        let mut order = vec![];
        let mut index = HashMap::new();
        self.post_order(self.root_id(), &mut order, &mut index);
        let mut lists: Vec<Vec<GaastCompMul>> = vec![]; // keeps the entry arrays alive until program_create
        let mut inputs: Vec<&T> = vec![];
        let mut input_node_dims: Vec<usize> = vec![];
        let mut nodes: Vec<GaastNodeDesc> = vec![];
        for id in &order {
            let n = self.get_node(*id);
            let mut d = GaastNodeDesc {
                opcode: 0, child0: -1, child1: -1,
                minimal_grade_mask: mask_of(&n.grade_set()),
                vec_space_dim: n.vec_space_dim() as i32,
                input_slot: -1, product_kind: GAAST_PROD_EXPLICIT, n_comp_muls: 0, comp_muls: std::ptr::null(),
            };
            match n.ast_node() {
                N::GradedObj(x) => {
                    d.opcode = OP_INPUT;
                    d.input_slot = inputs.len() as i32;
                    inputs.push(x);
                    input_node_dims.push(n.vec_space_dim());
                }
                N::Addition(l, r) => { d.opcode = OP_ADD; d.child0 = index[l]; d.child1 = index[r]; }
                N::Product(p) => {
                    d.opcode = OP_PRODUCT; d.child0 = index[&p.left_expr]; d.child1 = index[&p.right_expr];
                    let v: Vec<GaastCompMul> = p.individual_comp_muls.iter().map(|m| GaastCompMul {
                        left_grade: m.left_comp.grade as u32, left_index: m.left_comp.index as u32,
                        right_grade: m.right_comp.grade as u32, right_index: m.right_comp.index as u32,
                        result_grade: m.result_comp.grade as u32, result_index: m.result_comp.index as u32,
                        coeff: m.coeff,
                    }).collect();
                    d.n_comp_muls = v.len() as u64;
                    lists.push(v);
                    d.comp_muls = lists.last().unwrap().as_ptr();
                }
                N::Negation(e) => { d.opcode = OP_NEG; d.child0 = index[e]; }
                N::Exponential(e) => { d.opcode = OP_EXP; d.child0 = index[e]; }
                N::Logarithm(e) => { d.opcode = OP_LOG; d.child0 = index[e]; }
                N::GradeProjection(e) => { d.opcode = OP_PROJ; d.child0 = index[e]; }
                N::Reverse(e) => { d.opcode = OP_REVERSE; d.child0 = index[e]; }
                N::GradeInvolution(e) => { d.opcode = OP_GINVOL; d.child0 = index[e]; }
                N::ScalarUnaryOp(ScalarUnaryOp::Inversion, e) => { d.opcode = OP_SINV; d.child0 = index[e]; }
                N::ScalarUnaryOp(ScalarUnaryOp::SquareRoot, e) => { d.opcode = OP_SSQRT; d.child0 = index[e]; }
            }
            nodes.push(d);
        }
        let root = self.get_node(self.root_id());
        let dim = alg.vec_space_dim();
        let metric: Vec<f64> = (0..dim).map(|i| alg.base_vec_dot(i, i)).collect();
        let input_descs: Vec<(u64, usize)> = inputs.iter().zip(&input_node_dims)
            .map(|(x, node_dim)| (mask_of(&x.grade_set()), storage_dim(*x, *node_dim))).collect();
        let c_inputs: Vec<GaastInputDesc> = input_descs.iter().map(|(m, d)| GaastInputDesc {
            grade_mask: *m, storage_dim: *d as i32, is_const: 0, const_row: std::ptr::null(),
        }).collect();
        let desc = GaastProgramDesc {
            vec_space_dim: dim as i32, metric_diag: metric.as_ptr(), dtype: GAAST_F64,
            n_nodes: nodes.len() as i32, nodes: nodes.as_ptr(), root: index[&self.root_id()],
            n_inputs: c_inputs.len() as i32, inputs: c_inputs.as_ptr(), flags,
        };
        let mut handle: Program = std::ptr::null_mut();
        check(unsafe { gaast_hip_program_create(&desc, &mut handle) }, "program_create");
        HipProgram { handle, out_mask: mask_of(&root.grade_set()), out_dim: root.vec_space_dim(), inputs, input_descs }
    }

    /// Same signature and result as the CPU `eval` (src/eval.rs:12-19).
    pub fn eval_hip<R: GradedDataMut>(&self, alg: &impl MetricAlgebra) -> R {
        let prog = self.to_hip(alg, FLAG_EXACT_ORDER);
        let bound: Vec<Vec<&T>> = vec![prog.inputs.clone()];
        prog.eval_batch::<R>(&bound).pop().unwrap()
    }
}

impl<'a, T: GradedData> HipProgram<'a, T> {
    /// Evaluates the program once per entry of `items`; `items[i][slot]` re-binds the `GradedObj` of that
    /// slot for item i (same grade set and slice lengths as the value the AST was built with).
    pub fn eval_batch<R: GradedDataMut>(&self, items: &[Vec<&T>]) -> Vec<R> {
        let batch = items.len() as i64;
        let mut mvs: Vec<Mv> = vec![];
        for (slot, (mask, dim)) in self.input_descs.iter().enumerate() {
            let mut mv: Mv = std::ptr::null_mut();
            check(unsafe { gaast_hip_mv_alloc(*dim as i32, *mask, batch, GAAST_F64, &mut mv) }, "mv_alloc");
            for k in (0..64usize).filter(|k| (mask >> k) & 1 == 1) {
                // [batch][C(dim, k)], item-major: what gaast_hip_mv_upload expects
                let mut staging: Vec<f64> = Vec::with_capacity(items.len() * n_choose_k(*dim, k));
                for it in items {
                    staging.extend_from_slice(it[slot].grade_slice(k as Grade));
                }
                check(unsafe { gaast_hip_mv_upload(mv, k as i32, staging.as_ptr() as *const c_void, staging.len() as i64) }, "mv_upload");
            }
            mvs.push(mv);
        }
        let mut out: Mv = std::ptr::null_mut();
        check(unsafe { gaast_hip_mv_alloc(self.out_dim as i32, self.out_mask, batch, GAAST_F64, &mut out) }, "mv_alloc");
        check(unsafe { gaast_hip_eval(self.handle, mvs.as_ptr(), mvs.len() as i32, batch, out) }, "eval");
        // root result -> R::init_null_mv(dim, root.grade_set()), slice by slice (eval.rs:18)
        let gs = (0..64usize).filter(|k| (self.out_mask >> k) & 1 == 1).fold(GradeSet::empty(), |g, k| g.add_grade(k));
        let mut results: Vec<R> = (0..items.len()).map(|_| R::init_null_mv(self.out_dim, &gs)).collect();
        for k in gs.iter() {
            let len = n_choose_k(self.out_dim, k);
            let mut staging = vec![0.0f64; items.len() * len];
            check(unsafe { gaast_hip_mv_download(out, k as i32, staging.as_mut_ptr() as *mut c_void, staging.len() as i64) }, "mv_download");
            for (i, r) in results.iter_mut().enumerate() {
                r.grade_slice_mut(k).copy_from_slice(&staging[i * len..(i + 1) * len]);
            }
        }
        for mv in mvs {
            unsafe { gaast_hip_mv_free(mv) };
        }
        unsafe { gaast_hip_mv_free(out) };
        results
    }
}
