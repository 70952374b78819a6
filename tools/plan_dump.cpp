// CPU-only: prints the launch plan (step names) gaast::build_plan produces for the whole-AST workloads of bench.py -- no GPU, no HIP.
// Build: g++ -std=c++17 -O1 -I include -I gaast_amd/csrc/device -I gaast_amd/csrc/common -I gaast_amd/csrc/host tools/plan_dump.cpp
//        gaast_amd/csrc/host/{expr,c_api_host,wire}.cpp gaast_amd/csrc/device/plan.cpp -o /tmp/plan_dump
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gaast_expr.h"
#include "plan.hpp"

static const char* g_dump_dir = nullptr;   // argv[2]: write the generated kernel sources there
static uint64_t full_mask(int n) { return (uint64_t(2) << n) - 1; }
static uint64_t even_mask(int n) { return 0x5555555555555555ull & full_mask(n); }

static void dump(const char* what, gaast_expr_t e, int n, int dtype, uint32_t flags) {
    std::vector<double> metric(size_t(n), 1.0);
    gaast_spec_t spec = gaast_expr_specialize(e, n, metric.data(), uint64_t(1) << 22);
    if (!spec) {
        std::printf("%s: specialization failed\n", what);
        return;
    }
    gaast_program_desc desc;
    gaast_spec_program_desc(spec, dtype, flags, &desc);
    gaast::Plan plan;
    gaast::build_plan(desc, plan);
    std::printf("%s (n = %d, %s, flags 0x%x): %zu step(s), %zu cache buffer(s)%s\n", what, n, dtype == GAAST_F32 ? "f32" : "f64", flags,
                plan.steps.size(), plan.node_buffers.size(), plan.unsupported.empty() ? "" : (" UNSUPPORTED: " + plan.unsupported).c_str());
    for (const gaast::Step& s : plan.steps) {
        std::printf("    %s%s%s\n", s.name.c_str(), s.chain_jit ? "  [gaast_chain]" : "", s.jit_source.empty() ? "" : (s.jit_items ? "  [gaast_jit, slab in LDS]" : "  [gaast_jit]"));
        if (g_dump_dir && !s.chain_jit_source.empty()) {
            char path[512];
            std::snprintf(path, sizeof path, "%s/%s_n%d_chain.hip", g_dump_dir, what, n);
            if (FILE* f = std::fopen(path, "w")) {
                std::fputs(s.chain_jit_source.c_str(), f);
                std::fclose(f);
            }
        }
        if (g_dump_dir && !s.jit_source.empty()) {
            char path[512];
            std::snprintf(path, sizeof path, "%s/%s_n%d.hip", g_dump_dir, what, n);
            if (FILE* f = std::fopen(path, "w")) {
                std::fputs(s.jit_source.c_str(), f);
                std::fclose(f);
            }
        }
    }
    gaast_spec_free(spec);
}

int main(int argc, char** argv) {
    const uint32_t flags = argc > 1 ? uint32_t(std::strtoul(argv[1], nullptr, 0)) : 0u;
    g_dump_dir = argc > 2 ? argv[2] : nullptr;
    {   // BASELINE configs[4]: the rotor sandwich R X ~R at n = 5 (one fused launch)
        const int n = 5;
        gaast_expr_t r = gaast_expr_input(0, even_mask(n), n), x = gaast_expr_input(1, 0x2, n);
        gaast_expr_t e = gaast_expr_product(gaast_expr_product(r, x, GAAST_PROD_GEOMETRIC), gaast_expr_rev(r), GAAST_PROD_GEOMETRIC);
        dump("sandwich", e, n, GAAST_F64, flags);
    }
    for (int n : {8, 12}) {
        {   // vinv: a.rev() * a.norm_sq().sinv(), a even (expr.rs:363-371)
            gaast_expr_t a = gaast_expr_input(0, even_mask(n), n);
            dump("vinv", gaast_expr_vinv(a), n, GAAST_F64, flags);
        }
        {   // the projection KAT of eval.rs:152-163 on batched inputs: (v & bv) & bv.vinv()
            gaast_expr_t v = gaast_expr_input(0, 0x2, n), bv = gaast_expr_input(1, 0x4, n);
            gaast_expr_t e = gaast_expr_product(gaast_expr_product(v, bv, GAAST_PROD_INNER), gaast_expr_vinv(bv), GAAST_PROD_INNER);
            dump("proj", e, n, GAAST_F64, flags);
        }
        {   // element-wise arms on rows too big to fuse, then a scaling product
            const int k = n / 2;
            gaast_expr_t a = gaast_expr_input(0, uint64_t(1) << k, n), b = gaast_expr_input(1, uint64_t(1) << k, n), sc = gaast_expr_input(2, 1, n);
            gaast_expr_t e = gaast_expr_product(gaast_expr_rev(gaast_expr_add(gaast_expr_neg(gaast_expr_rev(a)), gaast_expr_ginvol(b))), sc, GAAST_PROD_GEOMETRIC);
            dump("unary", e, n, GAAST_F64, flags);
        }
        {   // README.md:20-22: d = (a + b * c).g(2), full operands
            gaast_expr_t a = gaast_expr_input(0, full_mask(n), n), b = gaast_expr_input(1, full_mask(n), n), c = gaast_expr_input(2, full_mask(n), n);
            gaast_expr_t e = gaast_expr_g(gaast_expr_add(a, gaast_expr_product(b, c, GAAST_PROD_GEOMETRIC)), 2);
            dump("cfg1", e, n, GAAST_F64, flags);
        }
    }
    return 0;
}
