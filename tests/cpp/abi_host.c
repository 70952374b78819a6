/* A plain C host of the ABI (what a non-Python caller links against): #include "gaast_hip.h", -lgaast_hip, no ctypes,
 * no torch.  Builds BASELINE config 5 (R X ~R in R^{4,1}, f64) BY HAND as the flat program the reference's public read
 * API yields (specialize.rs:17-24, base_types.rs:8-55), binds per-grade slices the way GradedData::grade_slice hands
 * them over (graded.rs:43-47), evaluates and writes the root's rows.
 *
 *     abi_host <in.bin> <out.bin> <batch> [gather]
 *
 * in.bin : batch x 16 doubles (R: grades 0, 2, 4 concatenated per item) then batch x 5 doubles (X: grade 1)
 * out.bin: batch x 16 doubles (root: grades 1, 3, 5)
 * gather : go through the multi-GPU entry points with a one-rank communicator (gaast_hip_eval_gather, 4 chunks)
 * hiprtc_first : the ORDER that preceded round 3's two aborts, made deterministic: this process compiles, loads and launches
 *          three kernels through hiprtc (dlopen'd, no library of ours involved) BEFORE gaast_hip_init, then evaluates the
 *          unfused plan (GAAST_FLAG_NO_FUSION: statically compiled kernels only -- the first launches out of the fat binary)
 * tests/test_gpu_abi_c_host.py compares out.bin with the oracle, bit for bit. */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gaast_hip.h"
#include "cfg5_program.h"

#define TRY(call)                                                                             \
    do {                                                                                      \
        int st__ = (call);                                                                    \
        if (st__ != GAAST_OK) {                                                               \
            fprintf(stderr, "%s -> status %d: %s\n", #call, st__, gaast_hip_last_error());    \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

/* hiprtc and the HIP module API through dlopen (plain C: no HIP header needed for a handful of entry points) */
static int use_hiprtc_before_the_library(void) {
    void *rtc = dlopen("libhiprtc.so", RTLD_NOW | RTLD_GLOBAL), *hip = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!rtc || !hip) {
        fprintf(stderr, "dlopen: %s\n", dlerror());
        return 1;
    }
    typedef int (*create_t)(void **, const char *, const char *, int, const char **, const char **);
    typedef int (*compile_t)(void *, int, const char **);
    typedef int (*size_t_fn)(void *, size_t *);
    typedef int (*code_t)(void *, char *);
    typedef int (*destroy_t)(void **);
    typedef int (*load_t)(void **, const void *);
    typedef int (*getfn_t)(void **, void *, const char *);
    typedef int (*launch_t)(void *, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, void *, void **, void **);
    typedef int (*malloc_t)(void **, size_t);
    typedef int (*memcpy_t)(void *, const void *, size_t, int);
    typedef int (*sync_t)(void);
    typedef int (*unload_t)(void *);
    typedef int (*free_t)(void *);
    create_t create = (create_t)dlsym(rtc, "hiprtcCreateProgram");
    compile_t compile = (compile_t)dlsym(rtc, "hiprtcCompileProgram");
    size_t_fn code_size = (size_t_fn)dlsym(rtc, "hiprtcGetCodeSize");
    code_t get_code = (code_t)dlsym(rtc, "hiprtcGetCode");
    destroy_t destroy = (destroy_t)dlsym(rtc, "hiprtcDestroyProgram");
    load_t load = (load_t)dlsym(hip, "hipModuleLoadData");
    getfn_t getfn = (getfn_t)dlsym(hip, "hipModuleGetFunction");
    launch_t launch = (launch_t)dlsym(hip, "hipModuleLaunchKernel");
    malloc_t dmalloc = (malloc_t)dlsym(hip, "hipMalloc");
    memcpy_t dmemcpy = (memcpy_t)dlsym(hip, "hipMemcpy");
    sync_t dsync = (sync_t)dlsym(hip, "hipDeviceSynchronize");
    unload_t unload = (unload_t)dlsym(hip, "hipModuleUnload");
    free_t dfree = (free_t)dlsym(hip, "hipFree");
    if (!create || !compile || !code_size || !get_code || !destroy || !load || !getfn || !launch || !dmalloc || !dmemcpy || !dsync || !unload || !dfree) return 1;
    int *d = NULL;
    if (dmalloc((void **)&d, 3 * sizeof(int))) return 1;
    for (int k = 0; k < 3; ++k) {
        char src[256];
        snprintf(src, sizeof src, "extern \"C\" __global__ void probe%d(int* p) { if (threadIdx.x == 0) p[%d] = %d; }\n", k, k, 100 + k);
        void *prog = NULL;
        const char *opts[] = {"--offload-arch=gfx950"};
        if (create(&prog, src, "probe.hip", 0, NULL, NULL) || compile(prog, 1, opts)) return 1;
        size_t cs = 0;
        if (code_size(prog, &cs) || !cs) return 1;
        char *image = malloc(cs);
        if (get_code(prog, image) || destroy(&prog)) return 1;
        void *mod = NULL, *fn = NULL;
        char name[16];
        snprintf(name, sizeof name, "probe%d", k);
        if (load(&mod, image) || getfn(&fn, mod, name)) return 1;
        void *arg = d, *args[1];
        args[0] = &arg;
        if (launch(fn, 1, 1, 1, 64, 1, 1, 0, NULL, args, NULL) || dsync()) return 1;
        if (unload(mod)) return 1;
        free(image);
    }
    int got[3] = {0, 0, 0};
    if (dmemcpy(got, d, sizeof got, 2 /* hipMemcpyDeviceToHost */) || dfree(d)) return 1;
    if (got[0] != 100 || got[1] != 101 || got[2] != 102) return 1;
    printf("hiprtc first: three kernels compiled, loaded and run before gaast_hip_init\n");
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: abi_host in.bin out.bin batch [gather | hiprtc_first]\n");
        return 2;
    }
    const int64_t batch = atoll(argv[3]);
    const int use_gather = argc > 4 && strcmp(argv[4], "gather") == 0;
    const int hiprtc_first = argc > 4 && strcmp(argv[4], "hiprtc_first") == 0;
    if (hiprtc_first && use_hiprtc_before_the_library()) {
        fprintf(stderr, "the hiprtc preamble failed\n");
        return 1;
    }
    const int n = CFG5_N;
    const uint64_t EVEN = CFG5_EVEN, VEC = CFG5_VEC, ODD = CFG5_ODD; /* grades {0,2,4}, {1}, {1,3,5} */
    static cfg5_program cfg;
    cfg5_fill(&cfg);
#define desc cfg.desc
    if (hiprtc_first) desc.flags |= GAAST_FLAG_NO_FUSION;   /* statically compiled kernels only */

    const int dev = 0;
    TRY(gaast_hip_init(&dev, 1));
    printf("%s\n", gaast_hip_version());
    gaast_hip_program_t prog = NULL;
    TRY(gaast_hip_program_create(&desc, &prog));
    uint64_t out_mask = 0;
    int64_t out_len = 0;
    TRY(gaast_hip_program_output_info(prog, &out_mask, &out_len));
    if (out_mask != ODD || out_len != 16) {
        fprintf(stderr, "unexpected root: mask %llx len %lld\n", (unsigned long long)out_mask, (long long)out_len);
        return 1;
    }
    for (int i = 0; i < gaast_hip_program_num_launches(prog); ++i) printf("launch %d: %s\n", i, gaast_hip_program_launch_name(prog, i));

    /* inputs: item-major rows in the file; handed over grade by grade, as grade_slice(k) would be */
    double *R = malloc(sizeof(double) * 16 * (size_t)batch), *X = malloc(sizeof(double) * 5 * (size_t)batch);
    double *out_rows = malloc(sizeof(double) * 16 * (size_t)batch);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(R, sizeof(double), 16 * (size_t)batch, f) != 16 * (size_t)batch ||
        fread(X, sizeof(double), 5 * (size_t)batch, f) != 5 * (size_t)batch) {
        fprintf(stderr, "cannot read %s\n", argv[1]);
        return 1;
    }
    fclose(f);
    gaast_hip_mv_t mR = NULL, mX = NULL, mOut = NULL, mAll = NULL;
    TRY(gaast_hip_mv_alloc(n, EVEN, batch, GAAST_F64, &mR));
    TRY(gaast_hip_mv_alloc(n, VEC, batch, GAAST_F64, &mX));
    TRY(gaast_hip_mv_alloc(n, ODD, batch, GAAST_F64, &mOut));
    const int r_grades[3] = {0, 2, 4}, r_len[3] = {1, 10, 5}, r_off[3] = {0, 1, 11};
    for (int g = 0; g < 3; ++g) {
        double *slab = malloc(sizeof(double) * (size_t)r_len[g] * (size_t)batch);
        for (int64_t i = 0; i < batch; ++i) memcpy(slab + i * r_len[g], R + i * 16 + r_off[g], sizeof(double) * (size_t)r_len[g]);
        TRY(gaast_hip_mv_upload(mR, r_grades[g], slab, (int64_t)r_len[g] * batch));
        free(slab);
    }
    TRY(gaast_hip_mv_upload(mX, 1, X, 5 * batch));

    gaast_hip_mv_t ins[2];
    ins[0] = mR;
    ins[1] = mX;
    gaast_hip_mv_t result = mOut;
    if (use_gather) {
        unsigned char id[GAAST_COMM_ID_BYTES];
        int n_ranks = 0, rank = -1, world = -1;
        TRY(gaast_hip_comm_unique_id(id));
        TRY(gaast_hip_comm_init(id, 0, 1));
        TRY(gaast_hip_comm_info(&rank, &world));
        TRY(gaast_hip_comm_count_ranks(&n_ranks));
        printf("communicator: rank %d of %d, %d rank(s) counted\n", rank, world, n_ranks);
        if (n_ranks != 1) return 1;
        TRY(gaast_hip_mv_alloc(n, ODD, batch, GAAST_F64, &mAll));
        const int64_t counts[1] = {batch};
        TRY(gaast_hip_eval_gather(prog, ins, 2, mOut, mAll, counts, 0, 4));
        TRY(gaast_hip_gather_rows(mOut, mAll, counts, 0));     /* and once more as the blocking form */
        result = mAll;
    } else {
        TRY(gaast_hip_eval(prog, ins, 2, batch, mOut));
    }
    TRY(gaast_hip_synchronize());
    TRY(gaast_hip_mv_download_rows(result, out_rows, 16 * batch));
    f = fopen(argv[2], "wb");
    if (!f || fwrite(out_rows, sizeof(double), 16 * (size_t)batch, f) != 16 * (size_t)batch) {
        fprintf(stderr, "cannot write %s\n", argv[2]);
        return 1;
    }
    fclose(f);
    if (use_gather) TRY(gaast_hip_comm_destroy());
    TRY(gaast_hip_mv_free(mR));
    TRY(gaast_hip_mv_free(mX));
    TRY(gaast_hip_mv_free(mOut));
    if (mAll) TRY(gaast_hip_mv_free(mAll));
    TRY(gaast_hip_program_destroy(prog));
    TRY(gaast_hip_shutdown());
    free(R); free(X); free(out_rows);
    printf("OK\n");
    return 0;
}
