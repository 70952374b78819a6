/* One RANK of a multi-rank job driving the library's own gather from plain C (no torch, no Python in the process):
 * what the Rust host of `north_star` would do per GPU.  Started once per rank by gaast_amd.launch.spawn_ranks
 * (RANK / WORLD_SIZE in the environment); every rank is a fresh process.
 *
 *     gather_host <in.bin> <out.bin> <idfile> <transport.so|-> <root> <n_chunks> <alias 0|1> <count_0> ... <count_{W-1}>
 *
 * in.bin   : B x 16 doubles (R: grades 0, 2, 4 per item) then B x 5 doubles (X: grade 1), B = sum of the counts; rank r
 *            evaluates items [sum_{q<r} count_q, + count_r) of BASELINE config 5 (tests/cpp/cfg5_program.h)
 * out.bin  : written by the root: B x 16 doubles gathered by gaast_hip_eval_gather (n_chunks chunks, overlapped), after
 *            checking that a second, blocking gaast_hip_gather_rows delivers the same bytes
 * idfile   : the 128-byte communicator id: written by rank 0 (atomic rename), polled by the others
 * transport: the shared object with the nccl* entry points (tests/cpp/rccl_stub.c when ranks share a GPU), "-" = librccl
 * alias    : 1 = the root's `out` IS its row range of `gathered` (gaast_hip_mv_wrap on the same memory: no local copy)
 * GAAST_TEST_ROOT_MISCOUNT=1: the root under-counts one peer's rows (negative test: must FAIL, not hang).
 * Every rank uses device (RANK mod visible devices).  Exit code 0 only if every step succeeded. */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "gaast_hip.h"
#include "cfg5_program.h"

#define TRY(call)                                                                                         \
    do {                                                                                                  \
        int st__ = (call);                                                                                \
        if (st__ != GAAST_OK) {                                                                           \
            fprintf(stderr, "rank %d: %s -> status %d: %s\n", rank, #call, st__, gaast_hip_last_error()); \
            return 1;                                                                                     \
        }                                                                                                 \
    } while (0)

int main(int argc, char **argv) {
    const char *er = getenv("RANK"), *ew = getenv("WORLD_SIZE");
    const int rank = er ? atoi(er) : 0, world = ew ? atoi(ew) : 1;
    if (argc < 8 + world || world < 1 || world > 16) {
        fprintf(stderr, "usage: gather_host in.bin out.bin idfile transport.so|- root n_chunks alias count_0 ... (WORLD_SIZE counts)\n");
        return 2;
    }
    const char *transport = argv[4];
    const int root = atoi(argv[5]), n_chunks = atoi(argv[6]), alias = atoi(argv[7]);
    int64_t counts[16], total = 0, first = 0;
    for (int r = 0; r < world; ++r) {
        counts[r] = atoll(argv[8 + r]);
        if (r < rank) first += counts[r];
        total += counts[r];
    }
    const int64_t mine = counts[rank];
    const int n = CFG5_N;

    static cfg5_program cfg;
    cfg5_fill(&cfg);
    /* ranks share the box's GPU(s): GAAST_TEST_DEVICES = how many are visible */
    const char *nd = getenv("GAAST_TEST_DEVICES");
    const int dev = rank % (nd && atoi(nd) > 0 ? atoi(nd) : 1);
    TRY(gaast_hip_init(&dev, 1));
    if (strcmp(transport, "-") != 0) TRY(gaast_hip_comm_set_library(transport));

    /* the communicator id travels through a file */
    unsigned char id[GAAST_COMM_ID_BYTES];
    if (rank == 0) {
        char tmp[1024];
        TRY(gaast_hip_comm_unique_id(id));
        snprintf(tmp, sizeof tmp, "%s.tmp", argv[3]);
        FILE *f = fopen(tmp, "wb");
        if (!f || fwrite(id, 1, sizeof id, f) != sizeof id || fclose(f) || rename(tmp, argv[3])) {
            fprintf(stderr, "rank 0: cannot write %s\n", argv[3]);
            return 1;
        }
    } else {
        int ok = 0;
        for (int tries = 0; tries < 60000 && !ok; ++tries) {
            FILE *f = fopen(argv[3], "rb");
            if (f) {
                ok = fread(id, 1, sizeof id, f) == sizeof id;
                fclose(f);
            }
            if (!ok) {
                struct timespec ts = {0, 2000000};
                nanosleep(&ts, NULL);
            }
        }
        if (!ok) {
            fprintf(stderr, "rank %d: no communicator id in %s\n", rank, argv[3]);
            return 1;
        }
    }
    int n_ranks = 0, crank = -1, cworld = -1;
    TRY(gaast_hip_comm_init(id, rank, world));
    TRY(gaast_hip_comm_info(&crank, &cworld));
    TRY(gaast_hip_comm_count_ranks(&n_ranks));
    if (crank != rank || cworld != world || n_ranks != world) {
        fprintf(stderr, "rank %d: communicator says rank %d of %d, %d counted\n", rank, crank, cworld, n_ranks);
        return 1;
    }

    /* negative test: ONE rank's evaluation fails (GAAST_TEST_FAIL_EVAL_RANK = that rank); every rank must be told */
    const char *fail_rank = getenv("GAAST_TEST_FAIL_EVAL_RANK");
    if (fail_rank && atoi(fail_rank) == rank) cfg.desc.flags |= GAAST_FLAG_DEBUG_FAIL_EVAL;
    gaast_hip_program_t prog = NULL;
    TRY(gaast_hip_program_create(&cfg.desc, &prog));

    /* this rank's rows of the global input */
    double *R = malloc(sizeof(double) * 16 * (size_t)(mine ? mine : 1)), *X = malloc(sizeof(double) * 5 * (size_t)(mine ? mine : 1));
    FILE *f = fopen(argv[1], "rb");
    if (!f || fseek(f, (long)(sizeof(double) * 16 * (size_t)first), SEEK_SET) ||
        fread(R, sizeof(double), 16 * (size_t)mine, f) != 16 * (size_t)mine ||
        fseek(f, (long)(sizeof(double) * (16 * (size_t)total + 5 * (size_t)first)), SEEK_SET) ||
        fread(X, sizeof(double), 5 * (size_t)mine, f) != 5 * (size_t)mine) {
        fprintf(stderr, "rank %d: cannot read %s\n", rank, argv[1]);
        return 1;
    }
    fclose(f);
    gaast_hip_mv_t mR = NULL, mX = NULL, mOut = NULL, mAll = NULL, mAll2 = NULL;
    TRY(gaast_hip_mv_alloc(n, CFG5_EVEN, mine, GAAST_F64, &mR));
    TRY(gaast_hip_mv_alloc(n, CFG5_VEC, mine, GAAST_F64, &mX));
    TRY(gaast_hip_mv_upload_rows(mR, R, 16 * mine));
    TRY(gaast_hip_mv_upload_rows(mX, X, 5 * mine));
    if (rank == root) {
        TRY(gaast_hip_mv_alloc(n, CFG5_ODD, total, GAAST_F64, &mAll));
        TRY(gaast_hip_mv_alloc(n, CFG5_ODD, total, GAAST_F64, &mAll2));
    }
    if (rank == root && alias) {
        void *base = NULL;
        TRY(gaast_hip_mv_info(mAll, NULL, NULL, NULL, NULL, NULL, NULL, &base));
        TRY(gaast_hip_mv_wrap((char *)base + sizeof(double) * 16 * (size_t)first, n, CFG5_ODD, mine, GAAST_F64, 16, &mOut));
    } else {
        TRY(gaast_hip_mv_alloc(n, CFG5_ODD, mine, GAAST_F64, &mOut));
    }
    gaast_hip_mv_t ins[2];
    ins[0] = mR;
    ins[1] = mX;
    /* negative test: the root believes the last rank sends one row less than it does -> the transport must report
     * that a receive met a send of another size (tests/test_gpu_multirank_gather.py) */
    if (getenv("GAAST_TEST_ROOT_MISCOUNT") && rank == root && world > 1) counts[root == world - 1 ? 0 : world - 1] -= 1;
    if (fail_rank) {
        /* the failure of one rank's evaluation is COLLECTIVE (include/gaast_hip.h): every rank returns non-zero, the
         * failing rank its own status, the others GAAST_ERR_RCCL; exit code 42 = "reported as specified" */
        const int st = gaast_hip_eval_gather(prog, ins, 2, mOut, mAll, counts, root, n_chunks);
        const int want = atoi(fail_rank) == rank ? GAAST_ERR_HIP : GAAST_ERR_RCCL;
        fprintf(stderr, "rank %d: injected failure on rank %s -> status %d (%s)\n", rank, fail_rank, st, gaast_hip_last_error());
        return st == want ? 42 : 1;
    }
    /* twice: the second pass runs over warm buffers and must deliver the same rows */
    for (int pass = 0; pass < 2; ++pass) TRY(gaast_hip_eval_gather(prog, ins, 2, mOut, mAll, counts, root, n_chunks));
    TRY(gaast_hip_gather_rows(mOut, mAll2, counts, root));   /* the blocking form, into a second buffer */
    TRY(gaast_hip_synchronize());
    if (rank == root) {
        double *a = malloc(sizeof(double) * 16 * (size_t)(total ? total : 1)), *b = malloc(sizeof(double) * 16 * (size_t)(total ? total : 1));
        TRY(gaast_hip_mv_download_rows(mAll, a, 16 * total));
        TRY(gaast_hip_mv_download_rows(mAll2, b, 16 * total));
        if (memcmp(a, b, sizeof(double) * 16 * (size_t)total) != 0) {
            fprintf(stderr, "rank %d: overlapped and blocking gather differ\n", rank);
            return 1;
        }
        f = fopen(argv[2], "wb");
        if (!f || fwrite(a, sizeof(double), 16 * (size_t)total, f) != 16 * (size_t)total || fclose(f)) {
            fprintf(stderr, "cannot write %s\n", argv[2]);
            return 1;
        }
        free(a);
        free(b);
    }
    TRY(gaast_hip_comm_destroy());   /* the test transport reports here when sends and receives did not pair up */
    TRY(gaast_hip_mv_free(mR));
    TRY(gaast_hip_mv_free(mX));
    TRY(gaast_hip_mv_free(mOut));
    if (mAll) TRY(gaast_hip_mv_free(mAll));
    if (mAll2) TRY(gaast_hip_mv_free(mAll2));
    TRY(gaast_hip_program_destroy(prog));
    TRY(gaast_hip_shutdown());
    free(R);
    free(X);
    if (rank == root) printf("rank %d (root) of %d: %lld rows gathered in %d chunks, %d rank(s) counted: OK\n", rank, world,
                             (long long)total, n_chunks, n_ranks);
    return 0;
}
