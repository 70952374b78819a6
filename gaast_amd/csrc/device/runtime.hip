// C ABI of the gfx950 back end (include/gaast_hip.h): device storage of graded rows, program
// objects (launch plans) and the batched evaluator.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "comm.hpp"
#include "gaast_hip.h"
#include "kernels.hip.hpp"
#include "plan.hpp"

using namespace gaast;

// ------------------------------------------------------------------------------------------
// state
// ------------------------------------------------------------------------------------------
namespace {

thread_local std::string g_err;
bool g_init = false;
int g_device = -1;
hipStream_t g_stream = nullptr;
int g_num_cu = 256;
bool g_chain_too_big = false;      // program_create: a chained step went beyond the LDS the plan builder had budgeted
size_t g_max_lds = 160 * 1024;
Comm g_comm;                       // the gather communicator (gaast_hip_comm_init), if any
std::vector<hipEvent_t> g_events;  // chunk-done events of gaast_hip_eval_gather, created on demand
hipEvent_t g_comm_done = nullptr;
int64_t* g_comm_flag = nullptr;    // device word of gaast_hip_eval_gather's collective error flag

int set_err(int status, const std::string& msg) {
    g_err = msg;
    return status;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess)                                                                \
            return set_err(GAAST_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

// Every entry point that touches the GPU starts here: HIP's current device is per host thread, the library's
// device is per process (include/gaast_hip.h, conventions).
int ensure_init() {
    if (!g_init) return set_err(GAAST_ERR_NO_DEVICE, "gaast_hip_init() has not been called (or found no GPU)");
    HIP_TRY(hipSetDevice(g_device));
    return GAAST_OK;
}

size_t dtype_size(int dtype) { return dtype == GAAST_F32 ? 4 : 8; }

}  // namespace

struct gaast_hip_mv_s {
    Layout layout;
    int64_t batch = 0;
    int dtype = GAAST_F64;
    int64_t row_stride = 0;  // elements
    void* ptr = nullptr;
    bool owns = false;
};

struct gaast_hip_program_s;

namespace {
void mv_free_impl(gaast_hip_mv_t m);
void release_plan_resources(Plan& plan);
}  // namespace

struct gaast_hip_program_s {
    Plan plan;
    std::vector<gaast_hip_mv_t> const_mvs;  // per input slot (nullptr for bound slots)
    std::vector<gaast_hip_mv_t> scratch;    // per node buffer, sized for scratch_batch
    int64_t scratch_batch = 0;
    std::vector<std::string> launch_names;
    void* d_domain = nullptr;               // exp / log extension: items refused by the domain check (unsigned long long)
    gaast_hip_program_s() = default;
    gaast_hip_program_s(const gaast_hip_program_s&) = delete;
    gaast_hip_program_s& operator=(const gaast_hip_program_s&) = delete;
    ~gaast_hip_program_s() {  // every failure path of program_create and program_destroy end here
        release_plan_resources(plan);
        if (d_domain) (void)hipFree(d_domain);
        for (gaast_hip_mv_t m : const_mvs) mv_free_impl(m);
        for (gaast_hip_mv_t m : scratch) mv_free_impl(m);
    }
};

namespace {

// device tables and hiprtc modules of a plan (also called before a plan is rebuilt)
void release_plan_resources(Plan& plan) {
    for (Step& s : plan.steps) {
        for (void** p : {&s.d_a, &s.d_b, &s.d_c, &s.d_coeff, &s.d_i32, &s.d_coeff_b, &s.d_coeff_c, &s.d_pre_row_start, &s.d_pre_entries,
                         &s.d_pre_coeff, &s.d_pre_row_map, &s.d_pre_row_scale, &s.d_cj_ent1, &s.d_cj_pos1, &s.d_cj_ent2, &s.d_cj_out2}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
        if (s.jit_module) (void)hipModuleUnload(static_cast<hipModule_t>(s.jit_module));
        if (s.jit_module_fma) (void)hipModuleUnload(static_cast<hipModule_t>(s.jit_module_fma));
        s.jit_module = s.jit_module_fma = nullptr;
        s.jit_function = s.jit_function_fma = nullptr;
        std::vector<char>().swap(s.jit_code);   // only after the unload: the image outlives the module built from it
        std::vector<char>().swap(s.jit_code_fma);
    }
}

template <typename T>
int upload_vec(const std::vector<T>& v, void** dptr) {
    *dptr = nullptr;
    if (v.empty()) return GAAST_OK;
    HIP_TRY(hipMalloc(dptr, v.size() * sizeof(T)));
    HIP_TRY(hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return GAAST_OK;
}

int mv_alloc_impl(int dim, uint64_t mask, int64_t batch, int dtype, gaast_hip_mv_t* out) {
    auto* m = new gaast_hip_mv_s;
    m->layout = make_layout(dim, mask);
    m->batch = batch;
    m->dtype = dtype;
    m->row_stride = m->layout.row_len;
    m->owns = true;
    const size_t bytes = size_t(batch) * size_t(m->layout.row_len) * dtype_size(dtype);
    if (bytes) {
        hipError_t e = hipMalloc(&m->ptr, bytes);
        if (e == hipSuccess) e = hipMemsetAsync(m->ptr, 0, bytes, g_stream);
        if (e != hipSuccess) {
            if (m->ptr) (void)hipFree(m->ptr);
            delete m;
            return set_err(GAAST_ERR_HIP, std::string("mv_alloc: ") + hipGetErrorString(e));
        }
    }
    *out = m;
    return GAAST_OK;
}

void mv_free_impl(gaast_hip_mv_t m) {
    if (!m) return;
    if (m->owns && m->ptr) (void)hipFree(m->ptr);
    delete m;
}

int grid_for(int64_t total, int block) {
    int64_t g = (total + block - 1) / block;
    const int64_t cap = int64_t(g_num_cu) * 8;  // grid-stride beyond 8 blocks per CU
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return int(g);
}

struct Bound {  // a buffer resolved for one eval call
    void* ptr;
    int64_t stride;
};

// kernels whose dynamic LDS exceeds the 64 KiB default need the attribute raised, once
int allow_lds(const void* kern, size_t lds) {
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    return GAAST_OK;
}

int resident_blocks(const void* kern, int threads, size_t lds, int* per_cu) {
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, kern, threads, lds));
    if (*per_cu < 1) *per_cu = 1;
    return GAAST_OK;
}

// ------------------------------------------------------------------------------------------
// Launch configuration of one step, decided ONCE at gaast_hip_program_create: which kernel instantiation,
// block size, dynamic LDS, persistent grid.  A step no kernel of this back end can run (operands beyond the LDS
// budget) makes program_create fail with UNIMPLEMENTED: an eval then either runs every step or none.
// ------------------------------------------------------------------------------------------
// k_gp_mfma16x4<T, DEG, n, MODE, SC, CH> by run-time n and degeneracy.  f64 runs n = 8 ... 12 on it, f32 only n = 8, 9 (f32 at
// n >= 10 is k_gp_mfma32p's): the f32 instantiations for n = 10 ... 12 would be unreachable, so they are not built.
template <typename T, int MODE, bool SC, bool CH>
void (*mfma16x4_kernel(int n, bool degenerate))(DenseArgs<T>) {
    switch (n) {
    case 8: return degenerate ? &k_gp_mfma16x4<T, true, 8, MODE, SC, CH> : &k_gp_mfma16x4<T, false, 8, MODE, SC, CH>;
    case 9: return degenerate ? &k_gp_mfma16x4<T, true, 9, MODE, SC, CH> : &k_gp_mfma16x4<T, false, 9, MODE, SC, CH>;
    default: break;
    }
    if constexpr (std::is_same<T, double>::value) {
        switch (n) {
        case 10: return degenerate ? &k_gp_mfma16x4<T, true, 10, MODE, SC, CH> : &k_gp_mfma16x4<T, false, 10, MODE, SC, CH>;
        case 11: return degenerate ? &k_gp_mfma16x4<T, true, 11, MODE, SC, CH> : &k_gp_mfma16x4<T, false, 11, MODE, SC, CH>;
        case 12: return degenerate ? &k_gp_mfma16x4<T, true, 12, MODE, SC, CH> : &k_gp_mfma16x4<T, false, 12, MODE, SC, CH>;
        default: break;
        }
    }
    return nullptr;
}

// picks the (SCALED, CHAINED) instantiation of a dense kernel family: f(std::bool_constant<SC>, std::bool_constant<CH>)
template <typename F>
auto pick_variant(bool scaled, bool chained, F&& f) {
    if (scaled && chained) return f(std::true_type{}, std::true_type{});
    if (scaled) return f(std::true_type{}, std::false_type{});
    if (chained) return f(std::false_type{}, std::true_type{});
    return f(std::false_type{}, std::false_type{});
}

template <typename T>
int prepare_step(Step& s, const Layout& la, const Layout& lb, int n) {
    constexpr bool is_f64 = std::is_same<T, double>::value;
    const std::string tn = is_f64 ? "double" : "float";
    const std::string dg = s.degenerate ? "true" : "false";
    // (SCALED, CHAINED) template arguments as they appear in the kernel's name: ",true" = rescaled basis, ",false,true" = chained
    const std::string vs = s.chained ? (s.scaled ? ",true,true" : ",false,true") : (s.scaled ? ",true" : "");
    switch (s.kind) {
    case Step::PRODUCT_CSR: {
        const size_t per_item = size_t(la.row_len + lb.row_len) * sizeof(T);
        if (per_item > g_max_lds)
            return set_err(GAAST_ERR_UNIMPLEMENTED,
                           "product operands of " + std::to_string(per_item) + " bytes per item do not fit the " +
                               std::to_string(g_max_lds) + "-byte LDS of the list kernels (" + s.name + ")");
        if (s.chain_jit == 2) {
            // the chain (or single long-row list) specialised through hiprtc (plan.cpp: make_chain_jit): static LDS, persistent workgroups
            s.threads = s.cj_threads;
            s.lds = s.cj_lds;
            s.hip_kernel = "gaast_chain<" + tn + ">[" + (s.list_jit ? "one list, " : "") + std::to_string(s.cj_ipb) + " items, " + std::to_string(s.cj_threads) + " threads" +
                           (s.cj_split > 1 ? ", rows in " + std::to_string(s.cj_split) + " slices: re-ordered sums" : "") +
                           (s.cj_fmt[1] >= 3 ? ", sign-sorted terms" : "") + "]";
            int per_cu = 0;
            HIP_TRY(hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, static_cast<hipFunction_t>(s.jit_function), s.threads, 0));
            s.blocks_per_cu = per_cu < 1 ? 1 : per_cu;
            return GAAST_OK;
        }
        if (s.list_chain) {
            // two lists in one launch, the mid row in LDS (plan.cpp: chain_list_into_list): IPB items per workgroup
            s.lds = size_t(s.chain_ent2_lds) + size_t(s.chain_item_stride) * size_t(s.chain_ipb) * sizeof(T);
            if (s.lds > g_max_lds) {
                g_chain_too_big = true;
                return set_err(GAAST_ERR_UNIMPLEMENTED, "list chain does not fit in LDS (" + s.name + ")");
            }
            const int64_t pairs2 = int64_t(s.u32_b.size()) * s.chain_ipb;
            s.threads = int(std::min<int64_t>(512, std::max<int64_t>(256, (pairs2 + 63) / 64 * 64)));
            s.kern[0] = reinterpret_cast<const void*>(&k_product_ell_chain<T>);
            s.hip_kernel = "k_product_ell_chain<" + tn + ">";
            if (int st = allow_lds(s.kern[0], s.lds)) return st;
            return resident_blocks(s.kern[0], s.threads, s.lds, &s.blocks_per_cu);   // persistent workgroups
        }
        s.lds = per_item;
        s.threads = 256;
        if (s.ell_width > 0) {
            // items per pass over the list: as many as a 64 KiB share of LDS holds (at least one), at most 8
            int items = int((64 * 1024) / per_item);
            items = items >= 8 ? 8 : items >= 4 ? 4 : items >= 2 ? 2 : 1;
            s.max_items = items;
            using KernE = void (*)(EllArgs<T>);
            const KernE tab[2][4] = {{&k_product_ell<T, 1, false>, &k_product_ell<T, 2, false>, &k_product_ell<T, 4, false>, &k_product_ell<T, 8, false>},
                                     {&k_product_ell<T, 1, true>, &k_product_ell<T, 2, true>, &k_product_ell<T, 4, true>, &k_product_ell<T, 8, true>}};
            s.hip_kernel = "k_product_ell<" + tn + ",1.." + std::to_string(items) + "," + (s.ell_bytes ? "true" : "false") + ">";
            for (int l2 = 0; (1 << l2) <= items; ++l2) {
                s.kern[l2] = reinterpret_cast<const void*>(tab[s.ell_bytes ? 1 : 0][l2]);
                if (int st = allow_lds(s.kern[l2], per_item << l2)) return st;
            }
            return GAAST_OK;
        }
        // enough items per block to give 256 threads work, within a 64 KiB LDS budget
        const int n_rows = int(s.u32_b.size());
        int items = int((256 + n_rows - 1) / (n_rows > 0 ? n_rows : 1));
        const size_t budget = 64 * 1024;
        if (per_item * size_t(items) > budget) items = int(budget / per_item);
        if (items < 1) items = 1;
        s.max_items = items;
        s.kern[0] = reinterpret_cast<const void*>(&k_product_csr<T>);
        s.hip_kernel = "k_product_csr<" + tn + ">";
        return allow_lds(s.kern[0], per_item * size_t(items));
    }
    case Step::PRODUCT_DENSE: {
        if (s.use_spinor) {
            using KernS = void (*)(SpinorArgs);
            const int m = s.use_spinor;
            const size_t D = size_t(1) << m, plane = (D * (D + 1) + 63) / 64 * 64;
            s.lds = (m == 6 ? 2 * plane + (is_f64 ? 0 : 16) : 2 * D * (D + 1)) * sizeof(T);   // k_gp_spinor12s: second plane 16 words further
            const int lb5 = s.spinor_lam_bit;
            KernS kern = nullptr;
            if (is_f64 && m == 6) kern = lb5 == 5 ? &k_gp_spinor12d<5> : lb5 == 4 ? &k_gp_spinor12d<4> : &k_gp_spinor12d<-1>;
            else if (is_f64 && m == 5) kern = lb5 == 4 ? &k_gp_spinor_wave1d<5, 4> : lb5 == 3 ? &k_gp_spinor_wave1d<5, 3> : &k_gp_spinor_wave1d<5, -1>;
            else if (is_f64) kern = lb5 == 3 ? &k_gp_spinor_wave1d<4, 3> : lb5 == 2 ? &k_gp_spinor_wave1d<4, 2> : &k_gp_spinor_wave1d<4, -1>;
            else if (m == 6) {
                kern = lb5 == 5 ? &k_gp_spinor12s<5, false> : lb5 == 4 ? &k_gp_spinor12s<4, false> : &k_gp_spinor12s<-1, false>;
                const KernS fast = lb5 == 5 ? &k_gp_spinor12s<5, true> : lb5 == 4 ? &k_gp_spinor12s<4, true> : &k_gp_spinor12s<-1, true>;
                s.kern[1] = reinterpret_cast<const void*>(fast);   // full, aligned rows on both sides and in the result
                if (int st = allow_lds(s.kern[1], s.lds)) return st;
            }
            else if (m == 5) kern = lb5 == 4 ? &k_gp_spinor_wave1<5, 4> : lb5 == 3 ? &k_gp_spinor_wave1<5, 3> : &k_gp_spinor_wave1<5, -1>;
            else kern = lb5 == 3 ? &k_gp_spinor_wave1<4, 3> : lb5 == 2 ? &k_gp_spinor_wave1<4, 2> : &k_gp_spinor_wave1<4, -1>;
            s.kern[0] = reinterpret_cast<const void*>(kern);
            s.hip_kernel = (m == 6 ? (is_f64 ? "k_gp_spinor12d<" : "k_gp_spinor12s<") : (is_f64 ? "k_gp_spinor_wave1d<" : "k_gp_spinor_wave1<") + std::to_string(m) + ",") +
                           std::to_string(lb5) + ">";
            s.threads = m == 6 ? 256 : 64;
            if (int st = allow_lds(s.kern[0], s.lds)) return st;
            return resident_blocks(s.kern[0], s.threads, s.lds, &s.blocks_per_cu);
        }
        if (s.use_mfma) {
            if constexpr (!is_f64) {
                const int wpi = 1 << (n - 10);                 // waves per item
                s.threads = wpi > 4 ? wpi * 64 : 256;
                s.items_per_block = (s.threads / 64) / wpi;
                if (s.mfma32_pairs) {
                    s.lds = (size_t(s.items_per_block) * size_t(4 << n) + 16) * sizeof(float);
                    if (s.lds > g_max_lds) return set_err(GAAST_ERR_UNIMPLEMENTED, "dense product does not fit in LDS");
                    using KernD = void (*)(DenseArgs<float>);
                    const KernD kernp = pick_variant(s.scaled, s.chained, [&](auto sc, auto ch) -> KernD {
                        constexpr bool SC = decltype(sc)::value, CH = decltype(ch)::value;
                        return n == 10   ? (s.degenerate ? &k_gp_mfma32p<true, 10, SC, CH> : &k_gp_mfma32p<false, 10, SC, CH>)
                               : n == 11 ? (s.degenerate ? &k_gp_mfma32p<true, 11, SC, CH> : &k_gp_mfma32p<false, 11, SC, CH>)
                               : n == 12 ? (s.degenerate ? &k_gp_mfma32p<true, 12, SC, CH> : &k_gp_mfma32p<false, 12, SC, CH>)
                                         : (s.degenerate ? &k_gp_mfma32p<true, 13, SC, CH> : &k_gp_mfma32p<false, 13, SC, CH>);
                    });
                    s.kern[0] = reinterpret_cast<const void*>(kernp);
                    s.hip_kernel = "k_gp_mfma32p<" + dg + "," + std::to_string(n) + vs + ">";
                    if (int st = allow_lds(s.kern[0], s.lds)) return st;
                    return resident_blocks(s.kern[0], s.threads, s.lds, &s.blocks_per_cu);   // persistent workgroups
                }
                s.lds = size_t(s.items_per_block) * size_t(2 << n) * sizeof(float);
                if (s.lds > g_max_lds) return set_err(GAAST_ERR_UNIMPLEMENTED, "dense product does not fit in LDS");
                using KernD = void (*)(DenseArgs<float>);
                // (k_gp_mfma32 serves n = 14 only -- 16 waves and 128 KiB of LDS per item; n = 10 ... 13 run on k_gp_mfma32p)
                if (s.threads != 1024) return set_err(GAAST_ERR_UNIMPLEMENTED, "k_gp_mfma32 is built for n = 14 only");
                const KernD kern = pick_variant(s.scaled, s.chained, [&](auto sc, auto ch) -> KernD {
                    constexpr bool SC = decltype(sc)::value, CH = decltype(ch)::value;
                    return s.degenerate ? &k_gp_mfma32<true, 1024, SC, CH> : &k_gp_mfma32<false, 1024, SC, CH>;
                });
                s.kern[0] = reinterpret_cast<const void*>(kern);
                s.hip_kernel = "k_gp_mfma32<" + dg + "," + std::to_string(s.threads) + vs + ">";
                return allow_lds(s.kern[0], s.lds);
            }
        }
        if (s.use_mfma16 && s.use_mfma16d) {
            // k_gp_mfma16x4<T>: one wave per 16 result columns, one item per workgroup (f64: n = 8 ... 12; f32: build switch)
            s.threads = 64 << (n - 8);
            s.items_per_block = 1;
            s.lds = size_t(4 * (size_t(1) << n) + 32) * sizeof(T);   // +B, -B, +A, 16 spare, -A images, 16 zeros
            if (s.lds > g_max_lds) return set_err(GAAST_ERR_UNIMPLEMENTED, "dense product does not fit in LDS");
            using KernD = void (*)(DenseArgs<T>);
            // [0]: general staging; [1]: register prefetch (full, contiguous, 16-byte aligned rows at launch); [2]: ... and
            // every blade produced, nothing accumulated: straight-line result stores
            auto pick = [&](auto mode_tag) -> KernD { return mfma16x4_kernel<T, decltype(mode_tag)::value, false, false>(n, s.degenerate != 0); };
            // a rescaled basis (general diagonal metric): general staging and stores only; a chained product: every mode
            const KernD kd = pick_variant(s.scaled, s.chained, [&](auto sc, auto ch) -> KernD {
                return mfma16x4_kernel<T, 0, decltype(sc)::value, decltype(ch)::value>(n, s.degenerate != 0);
            });
            auto pick_chained = [&](auto mode_tag) -> KernD { return mfma16x4_kernel<T, decltype(mode_tag)::value, false, true>(n, s.degenerate != 0); };
            const bool chained_fast = s.chained && !s.scaled;
            const KernD kf = chained_fast ? pick_chained(std::integral_constant<int, 1>{}) : pick(std::integral_constant<int, 1>{}),
                        kw = chained_fast ? pick_chained(std::integral_constant<int, 2>{}) : pick(std::integral_constant<int, 2>{});
            if (!kd || !kf || !kw) return set_err(GAAST_ERR_UNIMPLEMENTED, "no k_gp_mfma16x4 instantiation for this dimension and value type");
            s.kern[0] = reinterpret_cast<const void*>(kd);
            s.kern[1] = reinterpret_cast<const void*>(kf);
            s.kern[2] = reinterpret_cast<const void*>(kw);
            s.hip_kernel = "k_gp_mfma16x4<" + tn + "," + dg + "," + std::to_string(n) + (s.scaled ? ",0" + vs + ">" : ",0|1|2" + vs + ">");   // staging / store mode: by alignment at launch
            for (int v = 0; v < 3; ++v)
                if (int st = allow_lds(s.kern[v], s.lds)) return st;
            return resident_blocks(s.kern[1], s.threads, s.lds, &s.blocks_per_cu);   // persistent workgroups
        }
        if (s.use_mfma6) {
            // k_gp_mfma6<T>: one wave per item, persistent single-wave workgroups, 2 KiB (f32) / 4 KiB (f64) of operand images
            s.threads = 64 * GAAST_MFMA6_WAVES;
            s.items_per_block = GAAST_MFMA6_WAVES;
            s.lds = (size_t(is_f64 ? 4096 : 2048) + 64 * sizeof(T)) * GAAST_MFMA6_WAVES;   // + a dummy element per lane (stores of vanishing slots)
            using KernD = void (*)(DenseArgs<T>);
            // [0]: any operands (partial grade sets, projected or accumulated results); [1]: full operands, every blade produced, nothing
            // accumulated -- straight-line item loop with counted waits
            const KernD k6 = s.scaled ? &k_gp_mfma6<T, true, false> : &k_gp_mfma6<T, false, false>;
            const KernD k6f = s.scaled ? &k_gp_mfma6<T, true, true> : &k_gp_mfma6<T, false, true>;
            s.kern[0] = reinterpret_cast<const void*>(k6);
            s.kern[1] = reinterpret_cast<const void*>(k6f);
            s.hip_kernel = "k_gp_mfma6<" + tn + (s.scaled ? ",true,0|1>" : ",false,0|1>");
            return resident_blocks(s.kern[1], s.threads, s.lds, &s.blocks_per_cu);
        }
        if (s.use_mfma7) {
            // k_gp_mfma7<T>: one wave per item, single-wave workgroups
            s.threads = 64;
            s.items_per_block = 1;
            s.lds = size_t(560) * sizeof(T);   // +B, -B, +A (u = 1 half 72 further), -A 144 further, 16 zeros
            using KernD = void (*)(DenseArgs<T>);
            const KernD kd = pick_variant(s.scaled, s.chained, [&](auto sc, auto ch) -> KernD {
                return &k_gp_mfma7<T, 0, decltype(sc)::value, decltype(ch)::value>;
            });
            const bool chained_fast = s.chained && !s.scaled;
            const KernD kf = chained_fast ? &k_gp_mfma7<T, 1, false, true> : &k_gp_mfma7<T, 1>;
            const KernD kw = chained_fast ? &k_gp_mfma7<T, 2, false, true> : &k_gp_mfma7<T, 2>;
            s.kern[0] = reinterpret_cast<const void*>(kd);
            s.kern[1] = reinterpret_cast<const void*>(kf);
            s.kern[2] = reinterpret_cast<const void*>(kw);
            s.hip_kernel = "k_gp_mfma7<" + tn + (s.scaled ? ",0" + vs + ">" : ",0|1|2" + vs + ">");   // (null vectors: run-time, no instantiation of their own)
            return resident_blocks(s.kern[1], s.threads, s.lds, &s.blocks_per_cu);   // persistent single-wave workgroups
        }
        const int lpi = 1 << (n - 4);
        s.threads = lpi > 256 ? lpi : 256;
        s.items_per_block = s.threads / lpi;
        s.lds = size_t(s.items_per_block) * size_t(2 * (1 << n) + (s.items_per_block > 1 ? 4 : 0)) * sizeof(T);
        if (s.lds > g_max_lds)
            return set_err(GAAST_ERR_UNIMPLEMENTED, "dense product of dimension " + std::to_string(n) + " does not fit in LDS");
        using KernD = void (*)(DenseArgs<T>);
        const KernD kern = pick_variant(s.scaled, s.chained, [&](auto sc, auto ch) -> KernD {
            constexpr bool SC = decltype(sc)::value, CH = decltype(ch)::value;
            if (s.neg_lo_all)
                return s.threads == 256 ? (s.degenerate ? &k_gp_dense<T, true, 256, true, SC, CH> : &k_gp_dense<T, false, 256, true, SC, CH>)
                                        : (s.degenerate ? &k_gp_dense<T, true, 512, true, SC, CH> : &k_gp_dense<T, false, 512, true, SC, CH>);
            return s.threads == 256 ? (s.degenerate ? &k_gp_dense<T, true, 256, false, SC, CH> : &k_gp_dense<T, false, 256, false, SC, CH>)
                                    : (s.degenerate ? &k_gp_dense<T, true, 512, false, SC, CH> : &k_gp_dense<T, false, 512, false, SC, CH>);
        });
        s.kern[0] = reinterpret_cast<const void*>(kern);
        s.hip_kernel = "k_gp_dense<" + tn + "," + dg + "," + std::to_string(s.threads) + "," + (s.neg_lo_all ? "true" : "false") + vs + ">";
        if (int st = allow_lds(s.kern[0], s.lds)) return st;
        // persistent workgroups: as many as are resident at once (register- and LDS-limited)
        return resident_blocks(s.kern[0], s.threads, s.lds, &s.blocks_per_cu);
    }
    case Step::ELEMENTWISE:
        s.hip_kernel = "k_elementwise<" + tn + (s.ew_ops <= 4 ? ",4>" : ",8>");
        return GAAST_OK;
    case Step::AXPY: s.hip_kernel = "k_axpy_map<" + tn + ">"; return GAAST_OK;
    case Step::FLIP: s.hip_kernel = "k_flip<" + tn + ">"; return GAAST_OK;
    case Step::SUNARY: s.hip_kernel = "k_scalar_unary<" + tn + ">"; return GAAST_OK;
    case Step::REDUCE_SCALE: {
        s.threads = 256;
        s.kern[0] = reinterpret_cast<const void*>(&k_reduce_scale<T>);
        s.hip_kernel = "k_reduce_scale<" + tn + ">";
        if (s.rs_wave) {   // tolerance mode: one wave per item, the row read once -- taken at launch when the three rows are one (run_step)
            using KernW = void (*)(ReduceScaleArgs<T>, const uint32_t*);
            KernW kw = nullptr;
            switch (s.rs_wave) {
            case 1: kw = &k_reduce_scale_wave<T, 1>; break;
            case 2: kw = &k_reduce_scale_wave<T, 2>; break;
            case 4: kw = &k_reduce_scale_wave<T, 4>; break;
            case 8: kw = &k_reduce_scale_wave<T, 8>; break;
            case 16:
                if constexpr (sizeof(T) == 8) kw = &k_reduce_scale_wave<T, 16>;
                break;
            default: break;
            }
            s.kern[1] = reinterpret_cast<const void*>(kw);
            if (kw) s.hip_kernel += " | k_reduce_scale_wave<" + tn + "," + std::to_string(s.rs_wave) + "> (lane-parallel sums) when the rows are one";
            else s.rs_wave = 0;
        }
        return resident_blocks(s.kern[0], s.threads, 0, &s.blocks_per_cu);
    }
    case Step::FUSED: {
        if (s.jit_function) return GAAST_OK;
        const size_t lds = (size_t(s.fused_slab) * FUSED_ITEMS + 8) * sizeof(T);
        return allow_lds(reinterpret_cast<const void*>(&k_ast_fused<T>), lds);
    }
    default: return GAAST_OK;
    }
}

template <typename T>
int run_step(const Step& s, const Bound& res, const Bound& a, const Bound& b, const Layout& la,
             const Layout& lb, int64_t batch, int n, const Bound& pre_a = Bound{nullptr, 0}, const Bound& pre_b = Bound{nullptr, 0}) {
    switch (s.kind) {
    case Step::ZERO: return GAAST_OK;  // handled by the caller (needs the row length)
    case Step::AXPY: {
        const int nm = int(s.u32_a.size());
        hipLaunchKernelGGL(k_axpy_map<T>, dim3(grid_for(batch * nm, 256)), dim3(256), 0, g_stream,
                           static_cast<T*>(res.ptr), res.stride, static_cast<const T*>(a.ptr), a.stride,
                           static_cast<const uint32_t*>(s.d_a), nm, batch, s.beta);
        break;
    }
    case Step::FLIP: {
        const int nm = int(s.u32_a.size());
        hipLaunchKernelGGL(k_flip<T>, dim3(grid_for(batch * nm, 256)), dim3(256), 0, g_stream,
                           static_cast<T*>(res.ptr), res.stride, static_cast<const uint32_t*>(s.d_a), nm, batch);
        break;
    }
    case Step::SUNARY:
        hipLaunchKernelGGL(k_scalar_unary<T>, dim3(grid_for(batch, 256)), dim3(256), 0, g_stream,
                           static_cast<T*>(res.ptr), res.stride, s.sunary_off, s.sunary_op, batch);
        break;
    case Step::REDUCE_SCALE: {
        ReduceScaleArgs<T> q;
        q.l1 = static_cast<const T*>(a.ptr);
        q.r1 = static_cast<const T*>(b.ptr);
        q.x = static_cast<const T*>(pre_a.ptr);
        q.out = static_cast<T*>(res.ptr);
        q.l1_stride = a.stride;
        q.r1_stride = b.stride;
        q.x_stride = pre_a.stride;
        q.out_stride = res.stride;
        q.ent1 = static_cast<const uint32_t*>(s.d_a);
        q.coeff1 = static_cast<const T*>(s.d_coeff);
        q.ent2 = static_cast<const uint32_t*>(s.d_b);
        q.coeff2 = static_cast<const T*>(s.d_coeff_b);
        q.n1 = int(s.u32_a.size());
        q.n2 = int(s.u32_b.size());
        q.canon_l1 = s.canon_a;
        q.canon_r1 = s.canon_b;
        q.canon_x = s.pre_canon_a;
        q.canon_s = s.rs_canon_s;
        q.s_is_left = s.list_chain == 1;
        q.op = s.rs_op;
        q.batch = batch;
        if (s.rs_wave && s.kern[1] && a.ptr == b.ptr && a.ptr == pre_a.ptr && a.stride == b.stride && a.stride == pre_a.stride &&
            (reinterpret_cast<uintptr_t>(a.ptr) & 15u) == 0 && (size_t(a.stride) * sizeof(T)) % 16 == 0 &&
            (reinterpret_cast<uintptr_t>(res.ptr) & 15u) == 0 && (size_t(res.stride) * sizeof(T)) % 16 == 0 && res.ptr != a.ptr) {
            // one wave per item, four per workgroup, persistent
            using KernW = void (*)(ReduceScaleArgs<T>, const uint32_t*);
            int64_t blocks = (batch + 3) / 4;
            blocks = std::min<int64_t>(blocks, int64_t(g_num_cu) * 8);
            hipLaunchKernelGGL(reinterpret_cast<KernW>(const_cast<void*>(s.kern[1])), dim3(unsigned(blocks)), dim3(256), 0, g_stream, q,
                               static_cast<const uint32_t*>(s.d_c));
            break;
        }
        // sixteen items per wave, four waves per workgroup, persistent: as many workgroups as are resident at once
        int64_t blocks = (batch + 63) / 64;
        blocks = std::min<int64_t>(blocks, int64_t(g_num_cu) * (s.blocks_per_cu > 0 ? s.blocks_per_cu : 8));
        hipLaunchKernelGGL(k_reduce_scale<T>, dim3(unsigned(blocks)), dim3(256), 0, g_stream, q);
        break;
    }
    case Step::EXPLOG: {
        ExpLogArgs<T> q;
        q.res = static_cast<T*>(res.ptr);
        q.arg = static_cast<const T*>(a.ptr);
        q.res_stride = res.stride;
        q.arg_stride = a.stride;
        q.op = s.explog_op;
        q.m = s.explog_m;
        q.m_res = s.explog_mres;
        q.arg_k = s.explog_arg_k;
        q.arg_0 = s.explog_arg_0;
        q.res_k = s.explog_res_k;
        q.res_0 = s.explog_res_0;
        q.sq = static_cast<const T*>(s.d_coeff);
        q.row_start = static_cast<const uint32_t*>(s.d_a);
        q.pairs = static_cast<const uint32_t*>(s.d_c);
        q.pair_coeff = static_cast<const T*>(s.d_coeff_b);
        q.n_rows = int(s.u32_a.size()) - 1;
        q.dom = static_cast<unsigned long long*>(s.d_domain);
        q.batch = batch;
        hipLaunchKernelGGL(k_exp_log<T>, dim3(grid_for(batch, 256)), dim3(256), 0, g_stream, q);
        break;
    }
    case Step::PRODUCT_CSR: {
        if (s.chain_jit == 2) {
            const bool mid_left = s.list_chain == 1;
            const Bound& other = s.list_jit ? b : s.chain_alias ? pre_a : (mid_left ? b : a);
            // a single list: its left operand is staged as the "mid" row (pointer l1), its right one is list 2's own operand (r2)
            const void *l1 = s.list_jit ? a.ptr : pre_a.ptr, *r1 = s.list_jit ? nullptr : pre_b.ptr, *r2 = other.ptr;
            long long s_l1 = s.list_jit ? a.stride : pre_a.stride, s_r1 = s.list_jit ? 0 : pre_b.stride, s_r2 = other.stride, s_out = res.stride, nb = batch;
            void* optr = res.ptr;
            const void *e1 = s.d_cj_ent1, *p1 = s.d_cj_pos1, *e2 = s.d_cj_ent2, *o2 = s.d_cj_out2;
            const void* init = (s.list_jit && s.fold_prev) ? pre_a.ptr : nullptr;
            long long s_init = (s.list_jit && s.fold_prev) ? pre_a.stride : 0;
            void* args[] = {&l1, &s_l1, &r1, &s_r1, &r2, &s_r2, &optr, &s_out, &e1, &p1, &e2, &o2, &nb, &init, &s_init};
            int64_t blocks = (batch + s.cj_ipb - 1) / s.cj_ipb;
            blocks = std::min<int64_t>(blocks, int64_t(g_num_cu) * s.blocks_per_cu);
            // (the argument block is copied into the dispatch packet at call time, like run_jit's)
            HIP_TRY(hipModuleLaunchKernel(static_cast<hipFunction_t>(s.jit_function), unsigned(blocks), 1, 1, unsigned(s.threads), 1, 1, 0, g_stream,
                                          args, nullptr));
            break;
        }
        if (s.list_chain) {
            EllChainArgs<T> q;
            const bool mid_left = s.list_chain == 1;
            const Bound& other = mid_left ? b : a;
            const Layout& lo = mid_left ? lb : la;
            q.l1 = static_cast<const T*>(pre_a.ptr);
            q.r1 = static_cast<const T*>(pre_b.ptr);
            q.r2 = static_cast<const T*>(other.ptr);
            q.out = static_cast<T*>(res.ptr);
            q.l1_stride = pre_a.stride;
            q.r1_stride = pre_b.stride;
            q.r2_stride = other.stride;
            q.out_stride = res.stride;
            q.l1_len = s.pre_left_len;
            q.r1_len = s.pre_right_len;
            q.r2_len = int(lo.row_len);
            q.mid_len = s.chain_mid_len;
            q.canon_l1 = s.pre_canon_a;
            q.canon_r1 = s.pre_canon_b;
            q.canon_r2 = mid_left ? s.canon_b : s.canon_a;
            q.canon_mid = s.chain_canon_mid;
            q.ent1 = static_cast<const uint32_t*>(s.d_pre_entries);
            q.pos1 = static_cast<const uint32_t*>(s.d_pre_row_map);
            q.rows1 = int(s.pre_row_map.size());
            q.width1 = s.pre_width;
            q.ent2 = static_cast<const uint32_t*>(s.d_c);
            q.out2 = static_cast<const uint32_t*>(s.d_b);
            q.rows2 = int(s.u32_b.size());
            q.width2 = s.ell_width;
            q.mid_is_left = mid_left ? 1 : 0;
            q.r2_alias = s.chain_alias;
            q.mid_covered = s.chain_covered;
            q.beta = s.beta;
            q.ipb = s.chain_ipb;
            q.item_stride = s.chain_item_stride;
            q.batch = batch;
            q.ent2_lds_bytes = s.chain_ent2_lds;
            int64_t blocks = (batch + s.chain_ipb - 1) / s.chain_ipb;
            if (s.blocks_per_cu > 0) blocks = std::min<int64_t>(blocks, int64_t(g_num_cu) * s.blocks_per_cu);
            hipLaunchKernelGGL(k_product_ell_chain<T>, dim3(unsigned(blocks)), dim3(unsigned(s.threads)), s.lds, g_stream, q);
            break;
        }
        if (s.ell_width > 0) {
            EllArgs<T> q;
            q.left = static_cast<const T*>(a.ptr);
            q.right = static_cast<const T*>(b.ptr);
            q.out = static_cast<T*>(res.ptr);
            q.left_stride = a.stride;
            q.right_stride = b.stride;
            q.out_stride = res.stride;
            q.left_len = int(la.row_len);
            q.right_len = int(lb.row_len);
            q.canon_left = s.canon_a;
            q.canon_right = s.canon_b;
            q.row_out = static_cast<const uint32_t*>(s.d_b);
            q.entries = static_cast<const uint32_t*>(s.d_c);
            q.n_rows = int(s.u32_b.size());
            q.width = s.ell_width;
            q.beta = s.beta;
            q.batch = batch;
            int l2 = s.max_items >= 8 ? 3 : s.max_items >= 4 ? 2 : s.max_items >= 2 ? 1 : 0;
            while (l2 > 0 && (int64_t(1) << l2) > batch) --l2;
            using KernE = void (*)(EllArgs<T>);
            const int64_t blocks = (batch + (int64_t(1) << l2) - 1) >> l2;
            hipLaunchKernelGGL(reinterpret_cast<KernE>(const_cast<void*>(s.kern[l2])), dim3(unsigned(blocks)), dim3(256),
                               s.lds << l2, g_stream, q);
            break;
        }
        CsrArgs<T> p;
        p.left = static_cast<const T*>(a.ptr);
        p.right = static_cast<const T*>(b.ptr);
        p.out = static_cast<T*>(res.ptr);
        p.left_stride = a.stride;
        p.right_stride = b.stride;
        p.out_stride = res.stride;
        p.left_len = int(la.row_len);
        p.right_len = int(lb.row_len);
        p.canon_left = s.canon_a;
        p.canon_right = s.canon_b;
        p.row_start = static_cast<const uint32_t*>(s.d_a);
        p.row_out = static_cast<const uint32_t*>(s.d_b);
        p.entries = static_cast<const uint32_t*>(s.d_c);
        p.coeff = static_cast<const T*>(s.d_coeff);
        p.n_rows = int(s.u32_b.size());
        p.beta = s.beta;
        p.batch = batch;
        int items = s.max_items;
        if (int64_t(items) > batch) items = int(batch);
        p.items = items;
        const int64_t blocks = (batch + items - 1) / items;
        hipLaunchKernelGGL(k_product_csr<T>, dim3(unsigned(blocks)), dim3(256), s.lds * size_t(items), g_stream, p);
        break;
    }
    case Step::FUSED: return GAAST_OK;  // launched by run_fused (needs every bound buffer)
    case Step::PRODUCT_DENSE: {
        if (s.use_spinor) {
            SpinorArgs q;
            q.left = a.ptr;
            q.right = b.ptr;
            q.out = res.ptr;
            q.left_stride = a.stride;
            q.right_stride = b.stride;
            q.out_stride = res.stride;
            q.left_map = static_cast<const uint16_t*>(s.d_a);
            q.right_map = static_cast<const uint16_t*>(s.d_b);
            q.left_full = s.left_full;
            q.right_full = s.right_full;
            q.out_map = static_cast<const uint16_t*>(s.d_c);
            q.out_full = s.out_full;
            q.left_len = int(la.row_len);
            q.right_len = int(lb.row_len);
            q.canon_left = s.canon_a;
            q.canon_right = s.canon_b;
            q.beta = s.beta;
            q.batch = batch;
            q.has_alpha = s.spinor_has_alpha;
            using KernS = void (*)(SpinorArgs);
            int64_t blocks = int64_t(g_num_cu) * s.blocks_per_cu;
            if (blocks > batch) blocks = batch;
            auto aligned16 = [](const void* ptr, int64_t stride) {
                return (reinterpret_cast<uintptr_t>(ptr) % 16 == 0) && ((size_t(stride) * sizeof(T)) % 16 == 0);
            };
            const bool fast = s.kern[1] && s.left_full && s.right_full && s.out_full && !s.beta && q.left_len == 4096 && q.right_len == 4096 &&
                              aligned16(a.ptr, a.stride) && aligned16(b.ptr, b.stride) && aligned16(res.ptr, res.stride);
            hipLaunchKernelGGL(reinterpret_cast<KernS>(const_cast<void*>(s.kern[fast ? 1 : 0])), dim3(unsigned(blocks)),
                               dim3(unsigned(s.threads)), s.lds, g_stream, q);
            break;
        }
        DenseArgs<T> p;
        p.left = static_cast<const T*>(a.ptr);
        p.right = static_cast<const T*>(b.ptr);
        p.out = static_cast<T*>(res.ptr);
        p.left_stride = a.stride;
        p.right_stride = b.stride;
        p.out_stride = res.stride;
        p.left_map = static_cast<const uint32_t*>(s.d_a);
        p.right_map = static_cast<const uint32_t*>(s.d_b);
        p.left_count = int(s.u32_a.size());
        p.right_count = int(s.u32_b.size());
        p.left_full = s.left_full;
        p.right_full = s.right_full;
        // vector loads need 16-byte aligned rows: base pointer and row stride
        auto aligned = [](const void* ptr, int64_t stride) {
            return (reinterpret_cast<uintptr_t>(ptr) % 16 == 0) && ((size_t(stride) * sizeof(T)) % 16 == 0);
        };
        p.left_contig = s.left_contig && aligned(a.ptr, a.stride);
        p.right_contig = s.right_contig && aligned(b.ptr, b.stride);
        p.out_map = static_cast<const int32_t*>(s.d_i32);
        p.canon_left = s.canon_a;
        p.canon_right = s.canon_b;
        p.n = n;
        p.neg_hi = s.neg_hi;
        p.zero_hi = s.zero_hi;
        p.neg_lo = s.neg_lo;
        p.beta = s.beta;
        p.batch = batch;
        using KernD = void (*)(DenseArgs<T>);
        const int64_t groups = (batch + s.items_per_block - 1) / s.items_per_block;
        int64_t blocks = groups;
        if (s.blocks_per_cu > 0) {  // persistent workgroups (vector-FMA form)
            blocks = int64_t(g_num_cu) * s.blocks_per_cu;
            if (blocks > groups) blocks = groups;
        }
        p.left_signs = s.left_signs;
        p.out_signs = s.out_signs;
        p.pre_left = p.pre_right = nullptr;
        p.pre_entries = nullptr;
        if (s.chained) {   // the left operand is a comp-mul list over two other rows, evaluated in LDS while staging
            p.pre_left = static_cast<const T*>(pre_a.ptr);
            p.pre_right = static_cast<const T*>(pre_b.ptr);
            p.pre_left_stride = pre_a.stride;
            p.pre_right_stride = pre_b.stride;
            p.pre_left_len = s.pre_left_len;
            p.pre_right_len = s.pre_right_len;
            p.pre_canon_left = s.pre_canon_a;
            p.pre_canon_right = s.pre_canon_b;
            p.pre_row_start = static_cast<const uint32_t*>(s.d_pre_row_start);
            p.pre_entries = static_cast<const uint32_t*>(s.d_pre_entries);
            p.pre_coeff = static_cast<const T*>(s.d_pre_coeff);
            p.pre_row_map = static_cast<const uint32_t*>(s.d_pre_row_map);
            p.pre_row_scale = static_cast<const T*>(s.d_pre_row_scale);
            p.pre_rows = int(s.pre_row_map.size());
            p.pre_width = s.pre_width;
            p.pre_scratch = int(s.pre_scratch_off / sizeof(T));
            p.left_count = 0;
        }
        p.left_scale = s.scaled ? static_cast<const T*>(s.d_coeff) : nullptr;
        p.right_scale = s.scaled ? static_cast<const T*>(s.d_coeff_b) : nullptr;
        p.out_scale = s.scaled ? static_cast<const T*>(s.d_coeff_c) : nullptr;
        // register-prefetch staging: full, contiguous, aligned operand rows; a chained step computes its left operand from a list
        // (then only the right row is prefetched)
        const bool prefetch = s.use_mfma6 ? (p.left_full && p.right_full && s.out_full && !s.beta)   // k_gp_mfma6's straight-line instantiation
                              : s.use_mfma7 ? (s.kern[1] && p.right_full && !s.scaled && (s.chained || p.left_full))   // one component per lane and load: no alignment needed
                                          : (s.use_mfma16 && s.kern[1] && p.right_contig && p.right_full && !s.scaled &&
                                             (s.chained ? true : (p.left_contig && p.left_full)));
        const bool whole_rows = prefetch && s.kern[2] && s.out_full && !s.beta;   // k_gp_mfma16x4: straight-line result stores
        hipLaunchKernelGGL(reinterpret_cast<KernD>(const_cast<void*>(s.kern[whole_rows ? 2 : prefetch ? 1 : 0])), dim3(unsigned(blocks)),
                           dim3(unsigned(s.threads)), s.lds, g_stream, p);
        break;
    }
    }
    HIP_TRY(hipGetLastError());
    return GAAST_OK;
}

// hiprtc specialisation of a fused plan; on any failure the LDS interpreter kernel stays in charge
// (contract: the variant with fused multiply-adds, kept beside the exact one -- Step::jit_function_fma)
bool jit_compile(Step& s, const std::string& source, const char* entry, std::string* log, bool contract = false) {
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, source.c_str(), "gaast_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return false;
    const char* opts[] = {"--offload-arch=gfx950", "-O3", contract ? "-ffp-contract=fast" : "-ffp-contract=off"};
    const hiprtcResult res = hiprtcCompileProgram(prog, 3, opts);
    if (res != HIPRTC_SUCCESS) {
        size_t n = 0;
        hiprtcGetProgramLogSize(prog, &n);
        log->assign(n, 0);
        if (n) hiprtcGetProgramLog(prog, &(*log)[0]);
        hiprtcDestroyProgram(&prog);
        return false;
    }
    size_t cs = 0;
    std::vector<char> code;
    bool got = hiprtcGetCodeSize(prog, &cs) == HIPRTC_SUCCESS && cs > 0;
    if (got) {
        code.resize(cs);
        got = hiprtcGetCode(prog, code.data()) == HIPRTC_SUCCESS;
    }
    hiprtcDestroyProgram(&prog);
    if (!got) {
        *log = "hiprtcGetCodeSize / hiprtcGetCode failed";
        return false;
    }
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    // HIP's header does not say that hipModuleLoadData copies the image: it stays alive, in the step, for as long as the
    // module does (release_plan_resources frees it after hipModuleUnload).  std::move keeps code.data() where it is.
    if (hipModuleLoadData(&mod, code.data()) != hipSuccess) return false;
    if (hipModuleGetFunction(&fn, mod, entry) != hipSuccess) {
        (void)hipModuleUnload(mod);
        return false;
    }
    if (contract) {
        s.jit_code_fma = std::move(code);
        s.jit_module_fma = mod;
        s.jit_function_fma = fn;
    } else {
        s.jit_code = std::move(code);
        s.jit_module = mod;
        s.jit_function = fn;
    }
    return true;
}

// When does the contracted variant of a specialised kernel pay?  An item costs ~2 vector instructions per comp-mul (4 cycles per
// wave instruction and SIMD, 64 items per wave, 1,024 SIMDs at 2.4 GHz) and `bytes` of HBM traffic at 8 TB/s; below 0.6 of the
// traffic time the arithmetic hides behind the rows (cl41: 336 comp-muls, 296 B: 0.46 -- measured 0.78-0.84 of HBM with the
// exact code), above it the vector ALUs are the bound (cl41 with one shared rotor: 168 B: 0.81 -- 79 % busy, 0.66 of HBM).
bool arithmetic_bound(uint64_t comp_muls, double bytes) {
    const double valu_ps = double(comp_muls) * 2.0 * 4.0 / 64.0 / (1024.0 * 2.4e9) * 1e12, hbm_ps = bytes / 8e12 * 1e12;
    return valu_ps > 0.6 * hbm_ps;
}

template <typename T>
int run_jit(const Step& s, const Plan& plan, const std::vector<Bound>& in_bound, const Bound& out, int64_t batch) {
    // argument block: (ptr, stride) per staged input image, then out, out stride, batch
    std::vector<void*> args;
    std::vector<const void*> ptrs(s.fused_inputs.size());
    std::vector<long long> strides(s.fused_inputs.size());
    for (size_t i = 0; i < s.fused_inputs.size(); ++i) {
        ptrs[i] = in_bound[size_t(s.fused_inputs[i].slot)].ptr;
        strides[i] = in_bound[size_t(s.fused_inputs[i].slot)].stride;
    }
    for (size_t i = 0; i < s.fused_inputs.size(); ++i) {
        args.push_back(&ptrs[i]);
        args.push_back(&strides[i]);
    }
    void* optr = out.ptr;
    long long ostride = out.stride, b = batch;
    args.push_back(&optr);
    args.push_back(&ostride);
    args.push_back(&b);
    void* dom = s.d_domain;
    if (plan.has_explog) args.push_back(&dom);
    const unsigned threads = unsigned(s.jit_threads);
    const unsigned per_block = unsigned(s.jit_items > 0 ? s.jit_items : s.jit_threads);   // (the slab-in-LDS form: 64 items per 512 threads)
    unsigned blocks = unsigned((batch + per_block - 1) / per_block);
    if (s.jit_persistent > 0) blocks = unsigned(std::min<int64_t>(blocks, int64_t(g_num_cu) * std::min(s.jit_persistent, 4)));   // persistent workgroups
    // (the argument block -- args, ptrs, strides and the locals they point at -- only has to live until this call returns:
    //  hipModuleLaunchKernel copies the kernel arguments into the dispatch packet's kernarg segment at call time)
    void* fn = s.jit_function;
    if (s.jit_function_fma) {
        // tolerance mode: the rows this launch really moves -- an operand shared by all items moves none, and only then does the
        // contracted variant run (a program over batched operands only keeps the reference's bits by default, as before)
        double bytes = double(plan.out_layout.row_len) * sizeof(T);
        bool shared = false;
        for (size_t i = 0; i < s.fused_inputs.size(); ++i) {
            if (strides[i] != 0) bytes += double(plan.input_layouts[size_t(s.fused_inputs[i].slot)].row_len) * sizeof(T);
            else shared = true;
        }
        if (shared && arithmetic_bound(s.n_entries, bytes)) fn = s.jit_function_fma;
    }
    HIP_TRY(hipModuleLaunchKernel(static_cast<hipFunction_t>(fn), blocks, 1, 1, threads, 1, 1, 0, g_stream, args.data(), nullptr));
    return GAAST_OK;
}

template <typename T>
int run_fused(const Step& s, const Plan& plan, const std::vector<Bound>& in_bound, const Bound& out, int64_t batch) {
    if (s.jit_function) return run_jit<T>(s, plan, in_bound, out, batch);
    FusedArgs<T> p;
    std::memset(&p, 0, sizeof(p));
    p.prog = static_cast<const uint32_t*>(s.d_a);
    p.phase_tab = static_cast<const uint32_t*>(s.d_b);
    p.n_phases = int(s.u32_b.size() / (2 * FUSED_GROUPS));
    for (size_t i = 0; i < s.coeff_host.size() && i < 6; ++i) p.coeff[i] = T(s.coeff_host[i]);
    p.slab = s.fused_slab;
    p.zero_slot = s.fused_zero_slot;
    p.n_in = int(s.fused_inputs.size());
    for (int i = 0; i < p.n_in; ++i) {
        const Step::FusedInput& fi = s.fused_inputs[size_t(i)];
        p.in_ptr[i] = static_cast<const T*>(in_bound[size_t(fi.slot)].ptr);
        p.in_stride[i] = in_bound[size_t(fi.slot)].stride;
        p.in_len[i] = int(plan.input_layouts[size_t(fi.slot)].row_len);
        p.in_base[i] = fi.base;
        p.in_canon[i] = fi.canon;
    }
    p.out_ptr = static_cast<T*>(out.ptr);
    p.out_stride = out.stride;
    p.out_len = int(plan.out_layout.row_len);
    p.out_base = s.fused_out_base;
    p.batch = batch;
    const size_t lds = (size_t(p.slab) * FUSED_ITEMS + 8) * sizeof(T);
    const int64_t blocks = (batch + FUSED_ITEMS - 1) / FUSED_ITEMS;
    hipLaunchKernelGGL(k_ast_fused<T>, dim3(unsigned(blocks)), dim3(FUSED_THREADS), lds, g_stream, p);
    HIP_TRY(hipGetLastError());
    return GAAST_OK;
}

// a run of element-wise arms (and the scaling product after it) in one pass: plan.cpp: fuse_elementwise_runs
template <typename T, typename Resolve>
int run_elementwise(const Step& s, const Bound& res, Resolve&& resolve, int64_t batch) {
    ElementwiseArgs<T> q;
    std::memset(&q, 0, sizeof(q));
    Layout unused;
    for (size_t i = 0; i < s.ew_src.size() && i < size_t(ELEMENTWISE_MAX_SRC); ++i) {
        const Bound b = resolve(s.ew_src[i], &unused);
        q.src[i] = static_cast<const T*>(b.ptr);
        q.src_stride[i] = b.stride;
    }
    q.ops = static_cast<const uint32_t*>(s.d_a);
    q.comp_off = static_cast<const uint32_t*>(s.d_b);
    q.n_ops = s.ew_ops;
    q.n_comp = int(s.u32_b.size());
    q.load_first = s.ew_load_first;
    q.batch = batch;
    if (s.ew_scale) {
        const Bound sc = resolve(s.b, &unused);
        q.out = static_cast<T*>(res.ptr);
        q.out_stride = res.stride;
        q.out_off = static_cast<const uint32_t*>(s.d_c);
        q.coeff = static_cast<const T*>(s.d_coeff);
        q.scalar = static_cast<const T*>(sc.ptr);
        q.scalar_stride = sc.stride;
        q.scalar_off = s.ew_scalar_off;
        q.canon_v = s.ew_canon_v;
        q.canon_s = s.ew_canon_s;
        q.s_is_left = s.ew_s_is_left;
    } else {
        q.res = static_cast<T*>(res.ptr);
        q.res_stride = res.stride;
    }
    // x: the components (a thread keeps one), y: strides over the items -- enough workgroups to fill the chip a few times over
    const unsigned gx = unsigned((q.n_comp + 255) / 256);
    const int64_t want_y = std::max<int64_t>(1, int64_t(g_num_cu) * 16 / gx);
    const unsigned gy = unsigned(std::min<int64_t>(std::min<int64_t>((batch + 3) / 4, want_y), 65535));   // (a thread takes four items per step)
    if (q.n_ops <= 4) hipLaunchKernelGGL((k_elementwise<T, 4>), dim3(gx, gy), dim3(256), 0, g_stream, q);
    else hipLaunchKernelGGL((k_elementwise<T, ELEMENTWISE_MAX_OPS>), dim3(gx, gy), dim3(256), 0, g_stream, q);
    HIP_TRY(hipGetLastError());
    return GAAST_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

const char* gaast_hip_last_error(void) { return g_err.c_str(); }
#ifndef GAAST_KERNELS_REV
#define GAAST_KERNELS_REV "unknown"
#endif
const char* gaast_hip_version(void) { return "gaast-hip 0.2 (gfx950) kernels " GAAST_KERNELS_REV; }

int gaast_hip_init(const int* device_ids, int n_dev) {
    if (n_dev != 1 || !device_ids)
        return set_err(GAAST_ERR_INVALID_ARGUMENT, "one process drives one GPU: pass exactly one device id");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return set_err(GAAST_ERR_NO_DEVICE, "no HIP device visible");
    if (device_ids[0] < 0 || device_ids[0] >= count) return set_err(GAAST_ERR_INVALID_ARGUMENT, "device id out of range");
    HIP_TRY(hipSetDevice(device_ids[0]));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_ids[0]));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_err(GAAST_ERR_NO_DEVICE, std::string("kernels are built for gfx950 only, found ") + prop.gcnArchName);
    g_num_cu = prop.multiProcessorCount;
    g_max_lds = prop.maxSharedMemoryPerMultiProcessor ? size_t(prop.maxSharedMemoryPerMultiProcessor) : size_t(160 * 1024);
    if (g_max_lds > 160 * 1024) g_max_lds = 160 * 1024;
    g_device = device_ids[0];
    g_stream = nullptr;
    // Load the library's code object NOW (HIP loads a fat binary lazily, at the first launch or attribute query of one of its
    // kernels): twice in round 3 a process that had compiled kernels through hiprtc first and launched its first statically
    // compiled kernel afterwards (tests/test_gpu_explog.py run alone: three hiprtc programs, then k_axpy_map / k_exp_log)
    // ended in a silent abort() inside that first launch; with the static code object resident before hiprtc is ever
    // initialised the order that preceded both aborts cannot occur.
    {
        hipFuncAttributes fa;
        HIP_TRY(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_flip<double>)));
    }
    g_init = true;
    return GAAST_OK;
}

int gaast_hip_shutdown(void) {
    if (g_init) {
        (void)hipSetDevice(g_device);
        std::string err;
        (void)comm_destroy(g_comm, &err);
        for (hipEvent_t ev : g_events) (void)hipEventDestroy(ev);
        g_events.clear();
        if (g_comm_done) (void)hipEventDestroy(g_comm_done);
        g_comm_done = nullptr;
        if (g_comm_flag) (void)hipFree(g_comm_flag);
        g_comm_flag = nullptr;
    }
    g_init = false;
    return GAAST_OK;
}

int gaast_hip_set_stream(void* hip_stream) {
    g_stream = static_cast<hipStream_t>(hip_stream);
    return GAAST_OK;
}

int gaast_hip_synchronize(void) {
    if (int st = ensure_init()) return st;
    HIP_TRY(hipStreamSynchronize(g_stream));
    if (g_comm.active()) HIP_TRY(hipStreamSynchronize(g_comm.stream));
    return GAAST_OK;
}

static int program_create_impl(const gaast_program_desc* desc, gaast_hip_program_t* out);

int gaast_hip_program_create(const gaast_program_desc* desc, gaast_hip_program_t* out) {
    if (!desc || !out) return set_err(GAAST_ERR_INVALID_ARGUMENT, "null argument");
    if (int st = ensure_init()) return st;
    g_chain_too_big = false;
    const int st = program_create_impl(desc, out);
    // The plan builder sizes a chain's LDS with its own estimate of the kernel's images; prepare_step checks the real figure
    // against the device.  On a mismatch the unchained plan of the same program still runs: rebuild without chains
    // (as the hiprtc-failure fallback does) instead of refusing the program.
    if (st == GAAST_ERR_UNIMPLEMENTED && g_chain_too_big && !(desc->flags & GAAST_FLAG_DEBUG_NO_CHAIN)) {
        gaast_program_desc d2 = *desc;
        d2.flags |= GAAST_FLAG_DEBUG_NO_CHAIN;
        return program_create_impl(&d2, out);
    }
    return st;
}

static int program_create_impl(const gaast_program_desc* desc, gaast_hip_program_t* out) {
    auto prog = std::make_unique<gaast_hip_program_s>();  // its destructor releases whatever a failure leaves behind
    try {
        build_plan(*desc, prog->plan);
    } catch (const std::exception& ex) {
        return set_err(GAAST_ERR_INVALID_PROGRAM, ex.what());
    }
    // valid in the reference, beyond this back end: refused whole, here, never half-evaluated
    if (!prog->plan.unsupported.empty()) return set_err(GAAST_ERR_UNIMPLEMENTED, prog->plan.unsupported);
    // hiprtc specialisation of fused plans, before anything is uploaded: a plan that can only run as the
    // specialised kernel is rebuilt without run-time compilation if the compiler is not available
    uint32_t rebuild_flags = 0;
    for (int attempt = 0; attempt < 3; ++attempt) {
        bool rebuild = false;
        for (Step& s : prog->plan.steps) {
            if (s.kind != Step::FUSED || s.jit_source.empty()) continue;
            if (desc->flags & GAAST_FLAG_DEBUG_KEEP_JIT_SOURCE) prog->plan.jit_source_kept += s.jit_source;
            std::string log;
            const bool ok = (desc->flags & GAAST_FLAG_DEBUG_JIT_FAILS) ? false : jit_compile(s, s.jit_source, "gaast_jit", &log);
            if (ok)
                s.name = "ast_jit" + s.name.substr(s.name.find('[')) + (s.jit_items ? " slab in LDS" : "");
            else if (!log.empty())
                g_err = "hiprtc: " + log;  // informational: the interpreter kernel (or an unfused plan) runs instead
            bool trial_failed = false;
            if (ok && s.jit_reg_trial) {
                // a slab beyond 160 / 200 elements in registers, on trial (plan.cpp: try_fuse): the compiled kernel has to leave two
                // waves per SIMD (eight single-wave workgroups per CU), else the plan is rebuilt with the slabs in LDS
                int per_cu = 0;
                trial_failed = hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, static_cast<hipFunction_t>(s.jit_function), s.jit_threads, 0) != hipSuccess ||
                               per_cu * (s.jit_threads / 64) < 8;
                if (trial_failed) {
                    rebuild = true;
                    rebuild_flags |= GAAST_FLAG_INTERNAL_SMALL_REG_SLAB;
                }
            }
            // tolerance mode, one item per thread, and arithmetic-bound at least when every operand is shared by all items: the
            // contracted variant too (run_jit picks per launch, by the operands bound)
            if (ok && !trial_failed && !(desc->flags & GAAST_FLAG_EXACT_ORDER) && !s.jit_items && !prog->plan.has_explog &&
                arithmetic_bound(s.n_entries, double(prog->plan.out_layout.row_len) * dtype_size(prog->plan.dtype))) {
                std::string log2;
                if (jit_compile(s, s.jit_source, "gaast_jit", &log2, true)) s.name += " | fused multiply-adds under shared operands";
            }
            std::string().swap(s.jit_source);
            if (!ok && s.fused_jit_only) {
                rebuild = true;
                rebuild_flags |= GAAST_FLAG_NO_JIT;
            }
        }
        if (!rebuild) break;
        gaast_program_desc d2 = *desc;
        d2.flags |= rebuild_flags;
        release_plan_resources(prog->plan);  // modules already loaded for other fused steps
        prog->plan = Plan();
        try {
            build_plan(d2, prog->plan);
        } catch (const std::exception& ex) {
            return set_err(GAAST_ERR_INVALID_PROGRAM, ex.what());
        }
        if (!prog->plan.unsupported.empty()) return set_err(GAAST_ERR_UNIMPLEMENTED, prog->plan.unsupported);
    }
    // list chains specialised per program; on any failure the generic k_product_ell_chain stays in charge
    {
        std::vector<char> drop(prog->plan.steps.size(), 0);
        for (size_t i = 0; i < prog->plan.steps.size(); ++i) {
            Step& s = prog->plan.steps[i];
            if (s.chain_jit != 1) continue;
            if (desc->flags & GAAST_FLAG_DEBUG_KEEP_JIT_SOURCE) prog->plan.jit_source_kept += s.chain_jit_source;
            std::string log;
            const bool ok = jit_compile(s, s.chain_jit_source, "gaast_chain", &log);
            if (!ok && !log.empty()) g_err = "hiprtc: " + log;
            s.chain_jit = ok ? 2 : 0;
            std::string().swap(s.chain_jit_source);
            if (ok && s.fold_prev && i > 0) {   // the copy_grades_from step before a single list is evaluated by the specialised kernel
                drop[i - 1] = 1;
                s.name += " <- " + prog->plan.steps[i - 1].name;
            }
            if (!ok) s.list_jit = s.fold_prev = 0;
        }
        std::vector<Step> kept;
        for (size_t i = 0; i < prog->plan.steps.size(); ++i)
            if (!drop[i]) kept.push_back(std::move(prog->plan.steps[i]));
        prog->plan.steps = std::move(kept);
    }
    Plan& plan = prog->plan;
    auto layout_of = [&](BufRef r) -> Layout {
        if (r.idx < 0) return Layout();
        switch (r.kind) {
        case BufKind::NODE: return plan.node_buffers[size_t(r.idx)];
        case BufKind::INPUT: return plan.input_layouts[size_t(r.idx)];
        default: return plan.out_layout;
        }
    };
    plan.slot_used.assign(plan.inputs.size(), 0);
    if (plan.has_explog) {
        HIP_TRY(hipMalloc(&prog->d_domain, sizeof(unsigned long long)));
        HIP_TRY(hipMemset(prog->d_domain, 0, sizeof(unsigned long long)));
    }
    for (Step& s : plan.steps) {
        s.d_domain = prog->d_domain;
        // kernel choice, LDS budget, persistent grid: fixed here, and a program no kernel can run is refused whole
        const Layout la = layout_of(s.a), lb = layout_of(s.b);
        const int step_n = (s.kind == Step::PRODUCT_DENSE && s.dense_n) ? s.dense_n : plan.n;   // parity-pure products run in Cl(n - 1)
        if (int st = plan.dtype == GAAST_F32 ? prepare_step<float>(s, la, lb, step_n) : prepare_step<double>(s, la, lb, step_n))
            return st;
        if (s.chained) {
            // the list's operand rows of every item a workgroup stages at once, after the kernel's own images
            s.pre_scratch_off = (s.lds + 15) / 16 * 16;
            const size_t items = size_t(s.items_per_block > 0 ? s.items_per_block : 1);
            // + the zero pair; the one-item matrix kernels keep the list's right row twice (+x, -x: a term's sign is an address)
            s.lds = s.pre_scratch_off + (items * size_t(s.pre_left_len + s.pre_right_len + 1) + (items == 1 ? size_t(s.pre_right_len) : 0)) * dtype_size(plan.dtype);
            if (s.lds > g_max_lds) {
                g_chain_too_big = true;
                return set_err(GAAST_ERR_UNIMPLEMENTED, "chained product does not fit in LDS (" + s.name + ")");
            }
            for (int v = 0; v < 3; ++v)
                if (s.kern[v])
                    if (int st = allow_lds(s.kern[v], s.lds)) return st;
            if (s.blocks_per_cu > 0)
                if (int st = resident_blocks(s.kern[0], s.threads, s.lds, &s.blocks_per_cu)) return st;
            if (s.pre_a.kind == BufKind::INPUT) plan.slot_used[size_t(s.pre_a.idx)] = 1;
            if (s.pre_b.kind == BufKind::INPUT) plan.slot_used[size_t(s.pre_b.idx)] = 1;
            if (int st = upload_vec(s.pre_row_start, &s.d_pre_row_start)) return st;
            if (int st = upload_vec(s.pre_entries, &s.d_pre_entries)) return st;
            if (int st = upload_vec(s.pre_row_map, &s.d_pre_row_map)) return st;
            auto upload_t = [&](const std::vector<double>& v, void** dptr) -> int {
                if (plan.dtype == GAAST_F32) {
                    std::vector<float> cf(v.begin(), v.end());
                    return upload_vec(cf, dptr);
                }
                return upload_vec(v, dptr);
            };
            if (int st = upload_t(s.pre_coeff, &s.d_pre_coeff)) return st;
            if (int st = upload_t(s.pre_row_scale, &s.d_pre_row_scale)) return st;
        }
        if (s.kind == Step::ELEMENTWISE) {
            for (const BufRef& b : s.ew_src)
                if (b.kind == BufKind::INPUT) plan.slot_used[size_t(b.idx)] = 1;
        } else if (s.kind == Step::REDUCE_SCALE) {
            if (s.pre_a.idx >= 0 && s.pre_a.kind == BufKind::INPUT) plan.slot_used[size_t(s.pre_a.idx)] = 1;
        } else if (s.list_chain || s.list_jit) {
            if (s.pre_a.idx >= 0 && s.pre_a.kind == BufKind::INPUT) plan.slot_used[size_t(s.pre_a.idx)] = 1;
            if (s.list_chain && s.pre_b.kind == BufKind::INPUT) plan.slot_used[size_t(s.pre_b.idx)] = 1;
            if (s.chain_jit == 2) {
                if (int st = upload_vec(s.cj_ent1, &s.d_cj_ent1)) return st;
                if (int st = upload_vec(s.cj_pos1, &s.d_cj_pos1)) return st;
                if (int st = upload_vec(s.cj_ent2, &s.d_cj_ent2)) return st;
                if (int st = upload_vec(s.cj_out2, &s.d_cj_out2)) return st;
            } else {
                if (int st = upload_vec(s.pre_entries, &s.d_pre_entries)) return st;
                if (int st = upload_vec(s.pre_row_map, &s.d_pre_row_map)) return st;
            }
            std::vector<uint32_t>().swap(s.cj_ent1);
            std::vector<uint32_t>().swap(s.cj_ent2);
        }
        if (s.a.idx >= 0 && s.a.kind == BufKind::INPUT) plan.slot_used[size_t(s.a.idx)] = 1;
        if (s.b.idx >= 0 && s.b.kind == BufKind::INPUT) plan.slot_used[size_t(s.b.idx)] = 1;
        for (const Step::FusedInput& fi : s.fused_inputs) plan.slot_used[size_t(fi.slot)] = 1;
        if (int st = upload_vec(s.u32_a, &s.d_a)) return st;
        if (int st = upload_vec(s.u32_b, &s.d_b)) return st;
        if (int st = upload_vec(s.u32_c, &s.d_c)) return st;
        if (int st = upload_vec(s.i32_a, &s.d_i32)) return st;
        if (!s.coeff.empty() && s.kind != Step::FUSED) {
            if (plan.dtype == GAAST_F32) {
                std::vector<float> cf(s.coeff.begin(), s.coeff.end());
                if (int st = upload_vec(cf, &s.d_coeff)) return st;
            } else {
                if (int st = upload_vec(s.coeff, &s.d_coeff)) return st;
            }
        }
        if (!s.coeff_b.empty()) {
            if (plan.dtype == GAAST_F32) {
                std::vector<float> cf(s.coeff_b.begin(), s.coeff_b.end());
                if (int st = upload_vec(cf, &s.d_coeff_b)) return st;
            } else {
                if (int st = upload_vec(s.coeff_b, &s.d_coeff_b)) return st;
            }
        }
        if (!s.coeff_c.empty()) {
            if (plan.dtype == GAAST_F32) {
                std::vector<float> cf(s.coeff_c.begin(), s.coeff_c.end());
                if (int st = upload_vec(cf, &s.d_coeff_c)) return st;
            } else {
                if (int st = upload_vec(s.coeff_c, &s.d_coeff_c)) return st;
            }
        }
        // the host images of the big tables are no longer needed
        s.coeff_host = s.kind == Step::FUSED ? s.coeff : std::vector<double>();
        std::vector<uint32_t>().swap(s.u32_c);
        std::vector<double>().swap(s.coeff);
        // launch label: what the step is, then WHICH HIP kernel runs it (the name rocprofv3 reports)
        prog->launch_names.push_back(s.hip_kernel.empty() ? s.name : s.name + " :: " + s.hip_kernel);
    }
    prog->const_mvs.assign(plan.inputs.size(), nullptr);
    for (size_t i = 0; i < plan.inputs.size(); ++i) {
        if (!plan.inputs[i].is_const) continue;
        const Layout& l = plan.input_layouts[i];
        gaast_hip_mv_t m = nullptr;
        if (int st = mv_alloc_impl(l.dim, l.mask, 1, plan.dtype, &m)) return st;
        prog->const_mvs[i] = m;
        if (l.row_len) {
            if (plan.dtype == GAAST_F32) {
                std::vector<float> r(plan.const_rows[i].begin(), plan.const_rows[i].end());
                HIP_TRY(hipMemcpy(m->ptr, r.data(), r.size() * 4, hipMemcpyHostToDevice));
            } else {
                HIP_TRY(hipMemcpy(m->ptr, plan.const_rows[i].data(), plan.const_rows[i].size() * 8,
                                  hipMemcpyHostToDevice));
            }
        }
    }
    *out = prog.release();
    return GAAST_OK;
}

int gaast_hip_program_destroy(gaast_hip_program_t prog) {
    if (!prog) return GAAST_OK;
    if (g_init) {
        (void)hipSetDevice(g_device);
        // launches of this program may still be in flight: a hiprtc module must not be unloaded (nor a table freed) under them
        (void)hipStreamSynchronize(g_stream);
    }
    delete prog;
    return GAAST_OK;
}

int gaast_hip_program_domain_errors(gaast_hip_program_t prog, int64_t* count) {
    if (!prog || !count) return set_err(GAAST_ERR_INVALID_ARGUMENT, "null argument");
    if (int st = ensure_init()) return st;
    *count = 0;
    if (!prog->d_domain) return GAAST_OK;
    HIP_TRY(hipStreamSynchronize(g_stream));
    unsigned long long v = 0;
    HIP_TRY(hipMemcpy(&v, prog->d_domain, sizeof(v), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(prog->d_domain, 0, sizeof(v)));
    *count = int64_t(v);
    return GAAST_OK;
}

const char* gaast_hip_program_jit_source(gaast_hip_program_t prog) {
    return prog ? prog->plan.jit_source_kept.c_str() : "";
}

int gaast_hip_program_output_info(gaast_hip_program_t prog, uint64_t* grade_mask, int64_t* row_len) {
    if (!prog) return set_err(GAAST_ERR_INVALID_ARGUMENT, "null program");
    if (grade_mask) *grade_mask = prog->plan.out_layout.mask;
    if (row_len) *row_len = prog->plan.out_layout.row_len;
    return GAAST_OK;
}

int gaast_hip_program_num_launches(gaast_hip_program_t prog) { return prog ? int(prog->launch_names.size()) : 0; }
const char* gaast_hip_program_launch_name(gaast_hip_program_t prog, int i) {
    if (!prog || i < 0 || i >= int(prog->launch_names.size())) return "";
    return prog->launch_names[size_t(i)].c_str();
}

int gaast_hip_mv_alloc(int dim, uint64_t grade_mask, int64_t batch, int dtype, gaast_hip_mv_t* out) {
    if (!out || dim < 0 || dim > GAAST_MAX_DIM || batch < 0 || (dtype != GAAST_F64 && dtype != GAAST_F32))
        return set_err(GAAST_ERR_INVALID_ARGUMENT, "bad mv_alloc argument");
    if (int st = ensure_init()) return st;
    return mv_alloc_impl(dim, grade_mask, batch, dtype, out);
}

int gaast_hip_mv_wrap(void* device_ptr, int dim, uint64_t grade_mask, int64_t batch, int dtype,
                      int64_t row_stride, gaast_hip_mv_t* out) {
    if (!out || dim < 0 || dim > GAAST_MAX_DIM || batch < 0 || (dtype != GAAST_F64 && dtype != GAAST_F32))
        return set_err(GAAST_ERR_INVALID_ARGUMENT, "bad mv_wrap argument");
    auto* m = new gaast_hip_mv_s;
    m->layout = make_layout(dim, grade_mask);
    if (row_stride < m->layout.row_len && !(batch <= 1)) {
        delete m;
        return set_err(GAAST_ERR_INVALID_ARGUMENT, "row_stride shorter than the row");
    }
    // a single row has no stride to speak of: keep the 2-D copies / memsets well-formed (pitch >= width)
    if (batch <= 1 && row_stride < m->layout.row_len) row_stride = m->layout.row_len;
    m->batch = batch;
    m->dtype = dtype;
    m->row_stride = row_stride;
    m->ptr = device_ptr;
    m->owns = false;
    *out = m;
    return GAAST_OK;
}

int gaast_hip_mv_free(gaast_hip_mv_t mv) {
    mv_free_impl(mv);
    return GAAST_OK;
}

int gaast_hip_mv_info(gaast_hip_mv_t mv, int* dim, uint64_t* grade_mask, int64_t* batch, int* dtype,
                      int64_t* row_len, int64_t* row_stride, void** device_ptr) {
    if (!mv) return set_err(GAAST_ERR_INVALID_ARGUMENT, "null mv");
    if (dim) *dim = mv->layout.dim;
    if (grade_mask) *grade_mask = mv->layout.mask;
    if (batch) *batch = mv->batch;
    if (dtype) *dtype = mv->dtype;
    if (row_len) *row_len = mv->layout.row_len;
    if (row_stride) *row_stride = mv->row_stride;
    if (device_ptr) *device_ptr = mv->ptr;
    return GAAST_OK;
}

static int mv_copy_grade(gaast_hip_mv_t mv, int grade, void* host, int64_t count, bool upload) {
    if (!mv || !host) return set_err(GAAST_ERR_INVALID_ARGUMENT, "null argument");
    if (int st = ensure_init()) return st;
    if (grade < 0 || grade > 63 || !((mv->layout.mask >> grade) & 1ULL))
        return set_err(GAAST_ERR_MISSING_GRADE, "grade " + std::to_string(grade) + " absent from this multivector");
    const int64_t glen = mv->layout.grade_len(grade);
    if (count != glen * mv->batch) return set_err(GAAST_ERR_INVALID_ARGUMENT, "count must be C(dim,k) * batch");
    if (count == 0) return GAAST_OK;
    const size_t sz = dtype_size(mv->dtype);
    char* dev = static_cast<char*>(mv->ptr) + size_t(mv->layout.offset(grade)) * sz;
    HIP_TRY(hipStreamSynchronize(g_stream));
    if (upload)
        HIP_TRY(hipMemcpy2D(dev, size_t(mv->row_stride) * sz, host, size_t(glen) * sz, size_t(glen) * sz,
                            size_t(mv->batch), hipMemcpyHostToDevice));
    else
        HIP_TRY(hipMemcpy2D(host, size_t(glen) * sz, dev, size_t(mv->row_stride) * sz, size_t(glen) * sz,
                            size_t(mv->batch), hipMemcpyDeviceToHost));
    return GAAST_OK;
}

int gaast_hip_mv_upload(gaast_hip_mv_t mv, int grade, const void* host, int64_t count) {
    return mv_copy_grade(mv, grade, const_cast<void*>(host), count, true);
}
int gaast_hip_mv_download(gaast_hip_mv_t mv, int grade, void* host, int64_t count) {
    return mv_copy_grade(mv, grade, host, count, false);
}

static int mv_copy_rows(gaast_hip_mv_t mv, void* host, int64_t count, bool upload) {
    if (!mv || (!host && count)) return set_err(GAAST_ERR_INVALID_ARGUMENT, "null argument");
    if (int st = ensure_init()) return st;
    const int64_t rl = mv->layout.row_len;
    if (count != rl * mv->batch) return set_err(GAAST_ERR_INVALID_ARGUMENT, "count must be row_len * batch");
    if (count == 0) return GAAST_OK;
    const size_t sz = dtype_size(mv->dtype);
    HIP_TRY(hipStreamSynchronize(g_stream));
    if (upload)
        HIP_TRY(hipMemcpy2D(mv->ptr, size_t(mv->row_stride) * sz, host, size_t(rl) * sz, size_t(rl) * sz,
                            size_t(mv->batch), hipMemcpyHostToDevice));
    else
        HIP_TRY(hipMemcpy2D(host, size_t(rl) * sz, mv->ptr, size_t(mv->row_stride) * sz, size_t(rl) * sz,
                            size_t(mv->batch), hipMemcpyDeviceToHost));
    return GAAST_OK;
}

int gaast_hip_mv_upload_rows(gaast_hip_mv_t mv, const void* host, int64_t count) {
    return mv_copy_rows(mv, const_cast<void*>(host), count, true);
}
int gaast_hip_mv_download_rows(gaast_hip_mv_t mv, void* host, int64_t count) {
    return mv_copy_rows(mv, host, count, false);
}

int gaast_hip_mv_zero(gaast_hip_mv_t mv) {
    if (!mv) return set_err(GAAST_ERR_INVALID_ARGUMENT, "null mv");
    if (int st = ensure_init()) return st;
    const size_t sz = dtype_size(mv->dtype);
    if (mv->batch && mv->layout.row_len)
        HIP_TRY(hipMemset2DAsync(mv->ptr, size_t(mv->row_stride) * sz, 0, size_t(mv->layout.row_len) * sz,
                                 size_t(mv->batch), g_stream));
    return GAAST_OK;
}

}  // extern "C"

namespace {

// argument checks of an eval + the (pointer, stride) of every bound input, for items [0, batch)
int bind_eval(gaast_hip_program_t prog, const gaast_hip_mv_t* inputs, int n_inputs, int64_t batch, gaast_hip_mv_t out,
              std::vector<Bound>& in_bound) {
    if (!prog || !out || batch < 0) return set_err(GAAST_ERR_INVALID_ARGUMENT, "null argument");
    Plan& plan = prog->plan;
    if (plan.error != GAAST_OK) return set_err(plan.error, plan.error_msg);  // the reference panics here
    if (n_inputs < 0 || (n_inputs && !inputs)) return set_err(GAAST_ERR_INVALID_ARGUMENT, "bad inputs");
    in_bound.assign(plan.inputs.size(), Bound{nullptr, 0});
    for (size_t i = 0; i < plan.inputs.size(); ++i) {
        gaast_hip_mv_t m = plan.inputs[i].is_const ? prog->const_mvs[i] : (int(i) < n_inputs ? inputs[i] : nullptr);
        const Layout& want = plan.input_layouts[i];
        if (!m) {
            // slots no launch reads may stay unbound
            if (plan.slot_used[i]) return set_err(GAAST_ERR_INVALID_ARGUMENT, "input slot " + std::to_string(i) + " is not bound");
            continue;
        }
        if (m->layout.mask != want.mask || m->layout.dim != want.dim)
            return set_err(GAAST_ERR_INVALID_ARGUMENT, "input slot " + std::to_string(i) + ": grade set / dimension differ from the program's");
        if (m->dtype != plan.dtype) return set_err(GAAST_ERR_INVALID_ARGUMENT, "input dtype differs from the program's");
        if (m->batch != batch && m->batch != 1)
            return set_err(GAAST_ERR_INVALID_ARGUMENT, "input batch must equal the eval batch or be 1 (shared)");
        in_bound[i] = Bound{m->ptr, (m->batch == 1 && batch != 1) ? 0 : m->row_stride};
    }
    if (out->layout.mask != plan.out_layout.mask || out->layout.dim != plan.out_layout.dim)
        return set_err(GAAST_ERR_INVALID_ARGUMENT, "output grade set / dimension differ from the root's");
    if (out->dtype != plan.dtype || out->batch != batch)
        return set_err(GAAST_ERR_INVALID_ARGUMENT, "output dtype / batch mismatch");
    return GAAST_OK;
}

// the launches of one evaluation over items [first, first + count) of the bound buffers
int eval_range(gaast_hip_program_t prog, const std::vector<Bound>& in_bound0, gaast_hip_mv_t out, int64_t first, int64_t count) {
    Plan& plan = prog->plan;
    if (plan.flags & GAAST_FLAG_DEBUG_FAIL_EVAL) return set_err(GAAST_ERR_HIP, "injected evaluation failure (GAAST_FLAG_DEBUG_FAIL_EVAL)");
    if (count == 0) return GAAST_OK;
    const size_t sz = dtype_size(plan.dtype);
    auto shifted = [&](Bound b) {
        if (b.ptr && first) b.ptr = static_cast<char*>(b.ptr) + size_t(first) * size_t(b.stride) * sz;
        return b;
    };
    std::vector<Bound> in_bound(in_bound0.size());
    for (size_t i = 0; i < in_bound0.size(); ++i) in_bound[i] = shifted(in_bound0[i]);
    const Bound out_b = shifted(Bound{out->ptr, out->row_stride});

    // cache buffers of the product operands (the per-eval HashMap<NodeId, R> of eval.rs:16)
    if (prog->scratch_batch < count || prog->scratch.size() != plan.node_buffers.size()) {
        if (!prog->scratch.empty()) HIP_TRY(hipStreamSynchronize(g_stream));  // launches may still read the old ones
        for (gaast_hip_mv_t m : prog->scratch) mv_free_impl(m);
        prog->scratch.clear();
        for (size_t bi = 0; bi < plan.node_buffers.size(); ++bi) {
            const Layout& l = plan.node_buffers[bi];
            gaast_hip_mv_t m = nullptr;
            // a cache buffer a chained product made unnecessary is never allocated
            const bool dead = bi < plan.node_dead.size() && plan.node_dead[bi];
            if (int st = mv_alloc_impl(l.dim, l.mask, dead ? 0 : count, plan.dtype, &m)) return st;
            prog->scratch.push_back(m);
        }
        prog->scratch_batch = count;
    }

    auto resolve = [&](BufRef r, Layout* lay) -> Bound {
        switch (r.kind) {
        case BufKind::NODE: {
            gaast_hip_mv_t m = prog->scratch[size_t(r.idx)];
            *lay = m->layout;
            return Bound{m->ptr, m->row_stride};
        }
        case BufKind::INPUT: *lay = plan.input_layouts[size_t(r.idx)]; return in_bound[size_t(r.idx)];
        default: *lay = out->layout; return out_b;
        }
    };
    for (const Step& s : plan.steps) {
        Layout lres, la, lb;
        const Bound res = resolve(s.res, &lres);
        if (s.kind == Step::FUSED) {
            const int st = plan.dtype == GAAST_F32 ? run_fused<float>(s, plan, in_bound, res, count)
                                                   : run_fused<double>(s, plan, in_bound, res, count);
            if (st != GAAST_OK) return st;
            continue;
        }
        if (s.kind == Step::ELEMENTWISE) {
            const int st = plan.dtype == GAAST_F32 ? run_elementwise<float>(s, res, resolve, count) : run_elementwise<double>(s, res, resolve, count);
            if (st != GAAST_OK) return st;
            continue;
        }
        if (s.kind == Step::ZERO) {
            if (lres.row_len)
                HIP_TRY(hipMemset2DAsync(res.ptr, size_t(res.stride) * sz, 0, size_t(lres.row_len) * sz,
                                         size_t(count), g_stream));
            continue;
        }
        Bound a{nullptr, 0}, b{nullptr, 0};
        if (s.a.idx >= 0) a = resolve(s.a, &la);
        if (s.b.idx >= 0) b = resolve(s.b, &lb);
        const int step_n = (s.kind == Step::PRODUCT_DENSE && s.dense_n) ? s.dense_n : plan.n;
        Bound pa{nullptr, 0}, pb{nullptr, 0};
        if (s.kind == Step::REDUCE_SCALE) {
            Layout unused;
            pa = resolve(s.pre_a, &unused);
        } else if (s.chained || s.list_chain) {
            Layout unused;
            pa = resolve(s.pre_a, &unused);
            pb = resolve(s.pre_b, &unused);
        } else if (s.list_jit && s.fold_prev) {
            Layout unused;
            pa = resolve(s.pre_a, &unused);   // the folded copy's source
        }
        const int st = plan.dtype == GAAST_F32 ? run_step<float>(s, res, a, b, la, lb, count, step_n, pa, pb)
                                               : run_step<double>(s, res, a, b, la, lb, count, step_n, pa, pb);
        if (st != GAAST_OK) return st;
    }
    return GAAST_OK;
}

int rccl_err(const std::string& msg) { return set_err(GAAST_ERR_RCCL, msg); }

// items [lo, hi) of chunk c when `count` items are cut into n_chunks contiguous chunks
void chunk_span(int64_t count, int n_chunks, int c, int64_t* lo, int64_t* hi) {
    const int64_t per = (count + n_chunks - 1) / n_chunks;
    *lo = std::min<int64_t>(int64_t(c) * per, count);
    *hi = std::min<int64_t>(*lo + per, count);
}

// chunk c of every rank's rows travels to root on the communicator's stream (the caller has made that stream
// wait for the chunk's compute)
int gather_chunk(gaast_hip_mv_t local, gaast_hip_mv_t gathered, const int64_t* counts, int root, int n_chunks, int c) {
    std::string err;
    const size_t sz = dtype_size(local->dtype);
    const size_t row = size_t(local->layout.row_len);
    if (g_comm.rank != root) {
        int64_t lo, hi;
        chunk_span(counts[g_comm.rank], n_chunks, c, &lo, &hi);
        if (hi > lo && row)
            if (comm_send(g_comm, static_cast<const char*>(local->ptr) + size_t(lo) * row * sz, size_t(hi - lo) * row, int(sz),
                          root, &err))
                return rccl_err(err);
        return GAAST_OK;
    }
    // root: one receive per peer, grouped, so that the transfers of all xGMI links are in flight together
    if (comm_group_start(&err)) return rccl_err(err);
    int64_t first = 0;
    int failed = 0;
    for (int r = 0; r < g_comm.world; ++r) {
        int64_t lo, hi;
        chunk_span(counts[r], n_chunks, c, &lo, &hi);
        if (r != root && hi > lo && row && !failed)
            failed = comm_recv(g_comm, static_cast<char*>(gathered->ptr) + size_t(first + lo) * row * sz, size_t(hi - lo) * row,
                               int(sz), r, &err);
        first += counts[r];
    }
    std::string err2;
    if (comm_group_end(&err2) && !failed) return rccl_err(err2);
    if (failed) return rccl_err(err);
    // the root's own rows: a device copy, unless `out` already is that part of `gathered`
    int64_t lo, hi, first_root = 0;
    for (int r = 0; r < root; ++r) first_root += counts[r];
    chunk_span(counts[root], n_chunks, c, &lo, &hi);
    char* dst = static_cast<char*>(gathered->ptr) + size_t(first_root + lo) * row * sz;
    const char* src = static_cast<const char*>(local->ptr) + size_t(lo) * row * sz;
    if (hi > lo && row && dst != src)
        HIP_TRY(hipMemcpyAsync(dst, src, size_t(hi - lo) * row * sz, hipMemcpyDeviceToDevice, g_comm.stream));
    return GAAST_OK;
}

int check_gather_args(gaast_hip_mv_t local, gaast_hip_mv_t gathered, const int64_t* counts, int root) {
    if (!g_comm.active()) return rccl_err("no communicator: call gaast_hip_comm_init first");
    if (!local || !counts || root < 0 || root >= g_comm.world) return set_err(GAAST_ERR_INVALID_ARGUMENT, "bad gather argument");
    int64_t total = 0;
    for (int r = 0; r < g_comm.world; ++r) {
        if (counts[r] < 0) return set_err(GAAST_ERR_INVALID_ARGUMENT, "negative row count");
        total += counts[r];
    }
    if (counts[g_comm.rank] > local->batch) return set_err(GAAST_ERR_INVALID_ARGUMENT, "counts[rank] exceeds the local batch");
    if (local->row_stride != local->layout.row_len && local->batch > 1)
        return set_err(GAAST_ERR_INVALID_ARGUMENT, "gather needs contiguous rows (row_stride == row_len)");
    if (g_comm.rank == root) {
        if (!gathered) return set_err(GAAST_ERR_INVALID_ARGUMENT, "root needs a destination");
        if (gathered->layout.mask != local->layout.mask || gathered->layout.dim != local->layout.dim || gathered->dtype != local->dtype)
            return set_err(GAAST_ERR_INVALID_ARGUMENT, "gather destination: grade set / dimension / dtype differ");
        if (gathered->batch < total) return set_err(GAAST_ERR_INVALID_ARGUMENT, "gather destination holds fewer rows than the ranks send");
        if (gathered->row_stride != gathered->layout.row_len && gathered->batch > 1)
            return set_err(GAAST_ERR_INVALID_ARGUMENT, "gather needs contiguous rows (row_stride == row_len)");
    }
    return GAAST_OK;
}

int chunk_event(int c, hipEvent_t* ev) {
    while (int(g_events.size()) <= c) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        g_events.push_back(e);
    }
    *ev = g_events[size_t(c)];
    return GAAST_OK;
}

// after the last transfer: later work on the library stream (and gaast_hip_synchronize) sees the gathered rows
int join_comm_stream() {
    if (!g_comm_done) HIP_TRY(hipEventCreateWithFlags(&g_comm_done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(g_comm_done, g_comm.stream));
    HIP_TRY(hipStreamWaitEvent(g_stream, g_comm_done, 0));
    return GAAST_OK;
}

}  // namespace

extern "C" {

int gaast_hip_eval(gaast_hip_program_t prog, const gaast_hip_mv_t* inputs, int n_inputs, int64_t batch,
                   gaast_hip_mv_t out) {
    if (int st = ensure_init()) return st;
    std::vector<Bound> in_bound;
    if (int st = bind_eval(prog, inputs, n_inputs, batch, out, in_bound)) return st;
    return eval_range(prog, in_bound, out, 0, batch);
}

int gaast_hip_comm_set_library(const char* path) {
    std::string err;
    if (comm_set_library(path, &err)) return rccl_err(err);
    return GAAST_OK;
}

int gaast_hip_comm_unique_id(void* id_out) {
    if (!id_out) return set_err(GAAST_ERR_INVALID_ARGUMENT, "null argument");
    if (int st = ensure_init()) return st;
    std::string err;
    if (comm_unique_id(id_out, &err)) return rccl_err(err);
    return GAAST_OK;
}

int gaast_hip_comm_init(const void* id, int rank, int world) {
    if (!id || world < 1 || rank < 0 || rank >= world) return set_err(GAAST_ERR_INVALID_ARGUMENT, "bad communicator argument");
    if (int st = ensure_init()) return st;
    std::string err;
    if (comm_init(g_comm, id, rank, world, &err)) return rccl_err(err);
    return GAAST_OK;
}

int gaast_hip_comm_destroy(void) {
    if (int st = ensure_init()) return st;
    std::string err;
    if (comm_destroy(g_comm, &err)) return rccl_err(err);
    return GAAST_OK;
}

int gaast_hip_comm_info(int* rank, int* world) {
    if (!g_comm.active()) return rccl_err("no communicator: call gaast_hip_comm_init first");
    if (rank) *rank = g_comm.rank;
    if (world) *world = g_comm.world;
    return GAAST_OK;
}

int gaast_hip_comm_count_ranks(int* n_ranks) {
    if (!n_ranks) return set_err(GAAST_ERR_INVALID_ARGUMENT, "null argument");
    if (int st = ensure_init()) return st;
    if (!g_comm.active()) return rccl_err("no communicator: call gaast_hip_comm_init first");
    int64_t* d = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), sizeof(int64_t)));
    const int64_t one = 1;
    int64_t got = 0;
    std::string err;
    hipError_t e = hipMemcpy(d, &one, sizeof(one), hipMemcpyHostToDevice);
    int failed = 0;
    if (e == hipSuccess) failed = comm_allreduce_sum_i64(g_comm, d, 1, &err);
    if (e == hipSuccess && !failed) e = hipStreamSynchronize(g_comm.stream);
    if (e == hipSuccess && !failed) e = hipMemcpy(&got, d, sizeof(got), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (failed) return rccl_err(err);
    if (e != hipSuccess) return set_err(GAAST_ERR_HIP, std::string("comm_count_ranks: ") + hipGetErrorString(e));
    *n_ranks = int(got);
    return GAAST_OK;
}

int gaast_hip_gather_rows(gaast_hip_mv_t local, gaast_hip_mv_t gathered, const int64_t* counts, int root) {
    if (int st = ensure_init()) return st;
    if (int st = check_gather_args(local, gathered, counts, root)) return st;
    hipEvent_t ev;
    if (int st = chunk_event(0, &ev)) return st;
    HIP_TRY(hipEventRecord(ev, g_stream));               // the rows are produced on the library stream
    HIP_TRY(hipStreamWaitEvent(g_comm.stream, ev, 0));
    if (int st = gather_chunk(local, gathered, counts, root, 1, 0)) return st;
    return join_comm_stream();
}

int gaast_hip_eval_gather(gaast_hip_program_t prog, const gaast_hip_mv_t* inputs, int n_inputs, gaast_hip_mv_t out,
                          gaast_hip_mv_t gathered, const int64_t* counts, int root, int n_chunks) {
    if (int st = ensure_init()) return st;
    if (n_chunks < 1 || n_chunks > 1024) return set_err(GAAST_ERR_INVALID_ARGUMENT, "n_chunks out of range");
    if (int st = check_gather_args(out, gathered, counts, root)) return st;
    const int64_t batch = counts[g_comm.rank];
    if (out->batch != batch) return set_err(GAAST_ERR_INVALID_ARGUMENT, "out must hold counts[rank] items");
    std::vector<Bound> in_bound;
    if (int st = bind_eval(prog, inputs, n_inputs, batch, out, in_bound)) return st;
    // chunk c is computed on the library stream; its rows leave on the communicator's stream while chunk c + 1 is
    // being computed.  Every rank walks all n_chunks steps (a rank with fewer items has empty chunks) so that
    // sends and receives pair up -- also after a local failure: the remaining transfers are still posted (of whatever
    // the rows hold) so that no peer is left waiting in a receive.  The failure is then made COLLECTIVE: every rank
    // all-reduces an error flag and returns non-zero if any rank failed (the root must not hand out stale rows as GAAST_OK).
    int first_error = GAAST_OK;
    std::string first_msg;
    auto note = [&](int st) {
        if (st != GAAST_OK && first_error == GAAST_OK) {
            first_error = st;
            first_msg = g_err;
        }
        return st;
    };
    for (int c = 0; c < n_chunks; ++c) {
        int64_t lo, hi;
        chunk_span(batch, n_chunks, c, &lo, &hi);
        if (first_error == GAAST_OK) note(eval_range(prog, in_bound, out, lo, hi - lo));
        hipEvent_t ev;
        if (note(chunk_event(c, &ev)) == GAAST_OK) {
            if (hipEventRecord(ev, g_stream) != hipSuccess || hipStreamWaitEvent(g_comm.stream, ev, 0) != hipSuccess)
                note(set_err(GAAST_ERR_HIP, "gaast_hip_eval_gather: chunk event"));
        }
        note(gather_chunk(out, gathered, counts, root, n_chunks, c));
    }
    note(join_comm_stream());
    // the collective error flag (sum over the ranks of "I failed"), on the communicator's stream behind the transfers
    int64_t failed_ranks = first_error != GAAST_OK ? 1 : 0;
    {
        std::string err;
        hipError_t e = g_comm_flag ? hipSuccess : hipMalloc(reinterpret_cast<void**>(&g_comm_flag), sizeof(int64_t));
        if (e == hipSuccess) e = hipMemcpy(g_comm_flag, &failed_ranks, sizeof(failed_ranks), hipMemcpyHostToDevice);
        int rc = 0;
        if (e == hipSuccess) rc = comm_allreduce_sum_i64(g_comm, g_comm_flag, 1, &err);
        if (e == hipSuccess && !rc) e = hipStreamSynchronize(g_comm.stream);
        if (e == hipSuccess && !rc) e = hipMemcpy(&failed_ranks, g_comm_flag, sizeof(failed_ranks), hipMemcpyDeviceToHost);
        if (rc) note(rccl_err(err));
        else if (e != hipSuccess) note(set_err(GAAST_ERR_HIP, std::string("gaast_hip_eval_gather: error flag: ") + hipGetErrorString(e)));
    }
    if (first_error != GAAST_OK) return set_err(first_error, first_msg);
    if (failed_ranks > 0)
        return rccl_err("gaast_hip_eval_gather: another rank failed (" + std::to_string(failed_ranks) + " of " + std::to_string(g_comm.world) +
                        "): the gathered rows are not valid");
    return GAAST_OK;
}

}  // extern "C"
