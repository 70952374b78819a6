/* Prints sizeof / offsetof of every struct of include/gaast_hip.h as JSON (plain C, no GPU).  The CPU test
 * tests/test_abi_layout.py compares it with the ctypes mirrors (gaast_amd/_lib.py) and with the constants committed
 * next to the Rust bindings (integration/rust/abi_layout.json, asserted at compile time in ffi.rs). */
#include <stddef.h>
#include <stdio.h>

#include "gaast_hip.h"

#define F(type, field) printf("    \"%s\": %zu,\n", #field, offsetof(type, field))
#define BEGIN(type, name) printf("  \"%s\": {\n", name)
#define END(type, last) printf("    \"sizeof\": %zu\n  }%s\n", sizeof(type), last ? "" : ",")

int main(void) {
    printf("{\n");
    BEGIN(gaast_comp_mul, "gaast_comp_mul");
    F(gaast_comp_mul, left_grade); F(gaast_comp_mul, left_index); F(gaast_comp_mul, right_grade);
    F(gaast_comp_mul, right_index); F(gaast_comp_mul, result_grade); F(gaast_comp_mul, result_index);
    F(gaast_comp_mul, coeff);
    END(gaast_comp_mul, 0);
    BEGIN(gaast_node_desc, "gaast_node_desc");
    F(gaast_node_desc, opcode); F(gaast_node_desc, child0); F(gaast_node_desc, child1);
    F(gaast_node_desc, minimal_grade_mask); F(gaast_node_desc, vec_space_dim); F(gaast_node_desc, input_slot);
    F(gaast_node_desc, product_kind); F(gaast_node_desc, n_comp_muls); F(gaast_node_desc, comp_muls);
    END(gaast_node_desc, 0);
    BEGIN(gaast_input_desc, "gaast_input_desc");
    F(gaast_input_desc, grade_mask); F(gaast_input_desc, storage_dim); F(gaast_input_desc, is_const);
    F(gaast_input_desc, const_row);
    END(gaast_input_desc, 0);
    BEGIN(gaast_program_desc, "gaast_program_desc");
    F(gaast_program_desc, vec_space_dim); F(gaast_program_desc, metric_diag); F(gaast_program_desc, dtype);
    F(gaast_program_desc, n_nodes); F(gaast_program_desc, nodes); F(gaast_program_desc, root);
    F(gaast_program_desc, n_inputs); F(gaast_program_desc, inputs); F(gaast_program_desc, flags);
    END(gaast_program_desc, 1);
    printf("}\n");
    return 0;
}
