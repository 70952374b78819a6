/* BASELINE config 5 (R X ~R in R^{4,1}, f64) built BY HAND as the flat program the reference's public read API yields
 * (specialize.rs:17-24, base_types.rs:8-55): shared by the C hosts of the ABI (abi_host.c, gather_host.c). */
#ifndef GAAST_TESTS_CFG5_PROGRAM_H
#define GAAST_TESTS_CFG5_PROGRAM_H
#include <string.h>

#include "gaast_hip.h"

#define CFG5_N 5
#define CFG5_EVEN 0x15u /* grades {0,2,4} */
#define CFG5_VEC 0x2u   /* {1} */
#define CFG5_ODD 0x2Au  /* {1,3,5} */

typedef struct cfg5_program {
    double metric[5];
    gaast_node_desc nodes[5];
    gaast_input_desc inputs[2];
    gaast_program_desc desc;
} cfg5_program;

/* post-order node list: R, X, R*X, ~R, (R*X)*~R; R is one shared node (Expr::clone shares the Rc, expr.rs:47-53) */
static void cfg5_fill(cfg5_program *p) {
    const double metric[5] = {1.0, 1.0, 1.0, 1.0, -1.0};
    memset(p, 0, sizeof *p);
    memcpy(p->metric, metric, sizeof metric);
    for (int i = 0; i < 5; ++i) {
        p->nodes[i].child0 = p->nodes[i].child1 = -1;
        p->nodes[i].vec_space_dim = CFG5_N;
        p->nodes[i].input_slot = -1;
        p->nodes[i].product_kind = GAAST_PROD_EXPLICIT;
    }
    p->nodes[0].opcode = GAAST_OP_INPUT;   p->nodes[0].minimal_grade_mask = CFG5_EVEN; p->nodes[0].input_slot = 0;
    p->nodes[1].opcode = GAAST_OP_INPUT;   p->nodes[1].minimal_grade_mask = CFG5_VEC;  p->nodes[1].input_slot = 1;
    p->nodes[2].opcode = GAAST_OP_PRODUCT; p->nodes[2].minimal_grade_mask = CFG5_ODD;  p->nodes[2].child0 = 0; p->nodes[2].child1 = 1;
    p->nodes[2].product_kind = GAAST_PROD_GEOMETRIC; p->nodes[2].n_comp_muls = 80;   /* compact: the library regenerates the list */
    p->nodes[3].opcode = GAAST_OP_REVERSE; p->nodes[3].minimal_grade_mask = CFG5_EVEN; p->nodes[3].child0 = 0;
    p->nodes[4].opcode = GAAST_OP_PRODUCT; p->nodes[4].minimal_grade_mask = CFG5_ODD;  p->nodes[4].child0 = 2; p->nodes[4].child1 = 3;
    p->nodes[4].product_kind = GAAST_PROD_GEOMETRIC; p->nodes[4].n_comp_muls = 256;
    p->inputs[0].grade_mask = CFG5_EVEN; p->inputs[0].storage_dim = CFG5_N;
    p->inputs[1].grade_mask = CFG5_VEC;  p->inputs[1].storage_dim = CFG5_N;
    p->desc.vec_space_dim = CFG5_N;
    p->desc.metric_diag = p->metric;
    p->desc.dtype = GAAST_F64;
    p->desc.n_nodes = 5;
    p->desc.nodes = p->nodes;
    p->desc.root = 4;
    p->desc.n_inputs = 2;
    p->desc.inputs = p->inputs;
    p->desc.flags = 0;
}
#endif
