//! `SpecializedAst::eval` on an MI355X through `libgaast_hip.so`.  Replaces `src/eval.rs` when the cargo
//! feature `hip` is on.  NOT COMPILED in this repository (no Rust toolchain in the build image); it only
//! uses the reference's public API, and the C side of every call is covered by this repository's tests.
use std::cell::RefCell;
use std::collections::HashMap;
use std::os::raw::c_void;
use std::rc::Rc;

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

/// Number of (left component, right component) pairs with |left| in `lmask`, |right| in `rmask` whose product
/// blade has a grade in `omask`: blades a, b of grades kl, kr sharing j basis vectors give grade kl + kr - 2j,
/// and there are C(n,kl) C(kl,j) C(n-kl,kr-j) such pairs.
fn geometric_pair_count(n: usize, lmask: u64, rmask: u64, omask: u64) -> u128 {
    let mut total: u128 = 0;
    for kl in (0..=n).filter(|k| (lmask >> k) & 1 == 1) {
        for kr in (0..=n).filter(|k| (rmask >> k) & 1 == 1) {
            for j in 0..=kl.min(kr) {
                let g = kl + kr - 2 * j;
                if kr - j > n - kl || (omask >> g) & 1 == 0 {
                    continue;
                }
                total += (n_choose_k(n, kl) as u128) * (n_choose_k(kl, j) as u128) * (n_choose_k(n - kl, kr - j) as u128);
            }
        }
    }
    total
}

/// The device program of one `SpecializedAst` (launch plan, tables, compiled kernels): built once, shared by every
/// later `eval_hip` / `to_hip` of the same AST with the same flags.
struct DeviceProgram {
    handle: Program,
    out_mask: u64,
    out_dim: usize,
    input_descs: Vec<(u64, usize)>, // (grade mask, storage dim) per slot
}

impl Drop for DeviceProgram {
    fn drop(&mut self) {
        unsafe { gaast_hip_program_destroy(self.handle) };
    }
}

thread_local! {
    // SpecializedAst is !Send + !Sync (Rc closures, raw-pointer NodeIds): one cache per thread is the whole story.
    // Key: (address of the AST, root NodeId, flags, fingerprint of the node list) -- the fingerprint keeps a new AST
    // that happens to reuse a dropped one's address from hitting a stale entry.
    static PROGRAM_CACHE: RefCell<HashMap<(usize, usize, u32, u64), Rc<DeviceProgram>>> = RefCell::new(HashMap::new());
}

/// Forget every cached device program of this thread (e.g. before `gaast_hip_shutdown`).
pub fn clear_program_cache() {
    PROGRAM_CACHE.with(|c| c.borrow_mut().clear());
}

/// One `SpecializedAst` bound to its device program and to the values its `GradedObj` nodes hold right now.
pub struct HipProgram<'a, T> {
    prog: Rc<DeviceProgram>,
    inputs: Vec<&'a T>,
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

    /// Phases 1-3 stay as they are; this flattens their result.  The device program is built ONCE per
    /// (`SpecializedAst`, flags) and cached; later calls only re-collect the `GradedObj` values.
    /// `metric` = the diagonal of the algebra given to `specialize` (`MetricAlgebra::base_vec_dot(i, i)`).
    pub fn to_hip(&self, alg: &impl MetricAlgebra, flags: u32) -> HipProgram<'_, T> {
        let mut order = vec![];
        let mut index = HashMap::new();
        self.post_order(self.root_id(), &mut order, &mut index);
        // the values bound to the GradedObj nodes, in slot order (cheap: no table is touched)
        let mut inputs: Vec<&T> = vec![];
        let mut input_node_dims: Vec<usize> = vec![];
        let mut fingerprint: u64 = order.len() as u64;
        for id in &order {
            let n = self.get_node(*id);
            fingerprint = fingerprint.rotate_left(7) ^ mask_of(&n.grade_set());
            match n.ast_node() {
                N::GradedObj(x) => {
                    inputs.push(x);
                    input_node_dims.push(n.vec_space_dim());
                }
                N::Product(p) => fingerprint ^= (p.individual_comp_muls.len() as u64).wrapping_mul(0x9e3779b97f4a7c15),
                _ => {}
            }
        }
        let key = (self as *const Self as usize, index[&self.root_id()] as usize ^ (order.len() << 8), flags, fingerprint);
        if let Some(p) = PROGRAM_CACHE.with(|c| c.borrow().get(&key).cloned()) {
            return HipProgram { prog: p, inputs };
        }

        let dim = alg.vec_space_dim();
        let mut lists: Vec<Vec<GaastCompMul>> = vec![]; // keeps the entry arrays alive until program_create
        let mut nodes: Vec<GaastNodeDesc> = vec![];
        let mut slot = 0i32;
        for id in &order {
            let n = self.get_node(*id);
            let mut d = GaastNodeDesc {
                opcode: 0, child0: -1, child1: -1,
                minimal_grade_mask: mask_of(&n.grade_set()),
                vec_space_dim: n.vec_space_dim() as i32,
                input_slot: -1, product_kind: GAAST_PROD_EXPLICIT, n_comp_muls: 0, comp_muls: std::ptr::null(),
            };
            match n.ast_node() {
                N::GradedObj(_) => {
                    d.opcode = OP_INPUT;
                    d.input_slot = slot;
                    slot += 1;
                }
                N::Addition(l, r) => { d.opcode = OP_ADD; d.child0 = index[l]; d.child1 = index[r]; }
                N::Product(p) => {
                    d.opcode = OP_PRODUCT; d.child0 = index[&p.left_expr]; d.child1 = index[&p.right_expr];
                    d.n_comp_muls = p.individual_comp_muls.len() as u64;
                    // `grades_to_produce` is opaque here, but the list proves itself: every entry is a distinct
                    // (left component, right component) pair, the geometric product keeps EVERY pair whose blade
                    // has a wanted grade (expr.rs:180-183 selects all of |k1-k2|..k1+k2) and any other product a
                    // subset -- so a list as long as the geometric product's IS the geometric product's (order
                    // and coefficients follow from specialize.rs:162-183 / algebra.rs:73-83 alone).  The library
                    // then regenerates it (or, for dense products, never materialises it): nothing is converted
                    // or uploaded -- 537 MB at n = 12.
                    let lmask = mask_of(&self.get_node(p.left_expr).grade_set());
                    let rmask = mask_of(&self.get_node(p.right_expr).grade_set());
                    let same_dim = self.get_node(p.left_expr).vec_space_dim() == dim && self.get_node(p.right_expr).vec_space_dim() == dim;
                    if same_dim && geometric_pair_count(dim, lmask, rmask, d.minimal_grade_mask) == p.individual_comp_muls.len() as u128 {
                        d.product_kind = GAAST_PROD_GEOMETRIC;
                    } else {
                        let v: Vec<GaastCompMul> = p.individual_comp_muls.iter().map(|m| GaastCompMul {
                            left_grade: m.left_comp.grade as u32, left_index: m.left_comp.index as u32,
                            right_grade: m.right_comp.grade as u32, right_index: m.right_comp.index as u32,
                            result_grade: m.result_comp.grade as u32, result_index: m.result_comp.index as u32,
                            coeff: m.coeff,
                        }).collect();
                        lists.push(v);
                        d.comp_muls = lists.last().unwrap().as_ptr();
                    }
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
        let prog = Rc::new(DeviceProgram { handle, out_mask: mask_of(&root.grade_set()), out_dim: root.vec_space_dim(), input_descs });
        PROGRAM_CACHE.with(|c| c.borrow_mut().insert(key, prog.clone()));
        HipProgram { prog, inputs }
    }

    /// Same signature and result as the CPU `eval` (src/eval.rs:12-19).  The first call builds the device program
    /// (launch plan, tables, hiprtc kernel); later calls on the same AST reuse it.
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
        for (slot, (mask, dim)) in self.prog.input_descs.iter().enumerate() {
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
        let (out_dim, out_mask) = (self.prog.out_dim, self.prog.out_mask);
        check(unsafe { gaast_hip_mv_alloc(out_dim as i32, out_mask, batch, GAAST_F64, &mut out) }, "mv_alloc");
        check(unsafe { gaast_hip_eval(self.prog.handle, mvs.as_ptr(), mvs.len() as i32, batch, out) }, "eval");
        // root result -> R::init_null_mv(dim, root.grade_set()), slice by slice (eval.rs:18)
        let gs = (0..64usize).filter(|k| (out_mask >> k) & 1 == 1).fold(GradeSet::empty(), |g, k| g.add_grade(k));
        let mut results: Vec<R> = (0..items.len()).map(|_| R::init_null_mv(out_dim, &gs)).collect();
        for k in gs.iter() {
            let len = n_choose_k(out_dim, k);
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
