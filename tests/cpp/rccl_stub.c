/* rccl_stub.c -- TEST-ONLY transport with the nine nccl* entry points libgaast_hip.so resolves
 * (gaast_amd/csrc/device/comm.cpp), for boxes where several ranks must share ONE GPU: RCCL refuses two ranks on one
 * device, so without this the library's own multi-rank gather (gaast_hip_gather_rows / gaast_hip_eval_gather:
 * ncclSend / ncclRecv pairing, chunk events, the join of the communicator's stream) could never execute before an
 * 8-GPU node runs it.  NOT a product component and never loaded unless a host calls
 * gaast_hip_comm_set_library(<this .so>).
 *
 * Semantics kept from RCCL (what the library relies on):
 *  - point-to-point operations are STREAM-ORDERED: an operation starts when the work enqueued on its stream before it
 *    has finished (an event recorded at the call), and work enqueued on that stream after it starts only when the
 *    transfer has completed (a host function on the stream waits for it);
 *  - the call itself returns at once; operations between ncclGroupStart / ncclGroupEnd are posted together at
 *    ncclGroupEnd and progress concurrently;
 *  - every send must meet a receive of the same byte count on the peer: a mismatch is an error (reported on stderr,
 *    by the next call's return value, and by ncclCommDestroy).
 * Transport: device -> host copy on a private stream, one Unix-domain stream socket per pair of ranks (framed:
 * magic, byte count), host -> device copy.  The 128-byte unique id carries the socket directory prefix.
 *
 *   gcc -shared -fPIC -O1 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include rccl_stub.c -L/opt/rocm/lib -lamdhip64 -lpthread
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <poll.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <sys/types.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>

/* the slice of rccl.h this file implements (same values as <rccl/rccl.h>) */
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5, ncclRemoteError = 6 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6,
               ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct stub_comm* ncclComm_t;

#define STUB_MAGIC 0x67617374u /* "gast" */
#define STUB_TIMEOUT_S stub_timeout_s()
#define STUB_MAX_OP_BYTES ((size_t)1 << 31)

enum { OP_SEND = 0, OP_RECV = 1, OP_ALLREDUCE = 2 };

typedef struct stub_op {
    int kind, peer;
    void* buf;            /* device */
    const void* src;      /* all-reduce: device source */
    size_t bytes;
    int dtype;
    hipEvent_t ready;     /* recorded on the caller's stream at the call */
    int group;            /* operations of one group progress together */
    /* progress */
    char* host;
    size_t hdr_done, done_bytes;
    int finished, failed;
    struct stub_op* next;       /* queue */
    struct stub_op* all_next;   /* every op of the communicator, freed at destroy */
} stub_op;

struct stub_comm {
    int rank, world, device;
    int* fd;                    /* per peer */
    char prefix[100];
    pthread_t worker;
    pthread_mutex_t mu;
    pthread_cond_t cv_work, cv_done;
    stub_op *q_head, *q_tail, *all;
    int stop, async_error, next_group;
    hipStream_t copy_stream;
    long n_sends, n_recvs;
};

/* group state of the calling thread */
static __thread int t_group_depth = 0;
static __thread stub_op* t_group_head = NULL;
static __thread stub_op* t_group_tail = NULL;
static __thread hipStream_t t_group_stream[64];
static __thread int t_group_n = 0;

static int stub_timeout_s(void) {   /* GAAST_RCCL_STUB_TIMEOUT_S: how long a transfer waits for its peer (default 120) */
    const char* e = getenv("GAAST_RCCL_STUB_TIMEOUT_S");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 120;
}

static size_t dtype_size(int dt) {
    switch (dt) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
    }
}

static void stub_log(const struct stub_comm* c, const char* fmt, const char* a, long x, long y) {
    fprintf(stderr, "[rccl_stub rank %d/%d] ", c ? c->rank : -1, c ? c->world : 0);
    fprintf(stderr, fmt, a, x, y);
    fputc('\n', stderr);
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ---- sockets ---------------------------------------------------------------------------------------------- */
static void sock_path(const struct stub_comm* c, int rank, struct sockaddr_un* sa) {
    memset(sa, 0, sizeof *sa);
    sa->sun_family = AF_UNIX;
    snprintf(sa->sun_path, sizeof sa->sun_path, "%s.r%d", c->prefix, rank);
}

static int full_io(int fd, void* p, size_t n, int wr) {   /* blocking, with the socket's timeouts */
    char* b = (char*)p;
    while (n) {
        ssize_t k = wr ? send(fd, b, n, MSG_NOSIGNAL) : recv(fd, b, n, 0);
        if (k < 0 && errno == EINTR) continue;
        if (k <= 0) return -1;
        b += k;
        n -= (size_t)k;
    }
    return 0;
}

static int connect_all(struct stub_comm* c) {
    struct sockaddr_un sa;
    int lfd = socket(AF_UNIX, SOCK_STREAM, 0);
    if (lfd < 0) return -1;
    sock_path(c, c->rank, &sa);
    unlink(sa.sun_path);
    if (bind(lfd, (struct sockaddr*)&sa, sizeof sa) < 0 || listen(lfd, c->world + 4) < 0) {
        close(lfd);
        return -1;
    }
    struct timeval tv = {STUB_TIMEOUT_S, 0};
    /* lower ranks: connect (their listener may not exist yet) */
    for (int p = 0; p < c->rank; ++p) {
        int fd = -1;
        const double t_end = now_s() + STUB_TIMEOUT_S;
        sock_path(c, p, &sa);
        for (;;) {
            fd = socket(AF_UNIX, SOCK_STREAM, 0);
            if (fd < 0) break;
            if (connect(fd, (struct sockaddr*)&sa, sizeof sa) == 0) break;
            close(fd);
            fd = -1;
            if (now_s() > t_end) break;
            usleep(2000);
        }
        if (fd < 0) {
            close(lfd);
            return -1;
        }
        int32_t me = c->rank;
        if (full_io(fd, &me, sizeof me, 1)) {
            close(fd);
            close(lfd);
            return -1;
        }
        c->fd[p] = fd;
    }
    /* higher ranks: accept */
    for (int k = c->rank + 1; k < c->world; ++k) {
        struct pollfd pf = {lfd, POLLIN, 0};
        if (poll(&pf, 1, STUB_TIMEOUT_S * 1000) <= 0) {
            close(lfd);
            return -1;
        }
        int fd = accept(lfd, NULL, NULL);
        int32_t who = -1;
        if (fd < 0) {
            close(lfd);
            return -1;
        }
        setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
        if (full_io(fd, &who, sizeof who, 0) || who <= c->rank || who >= c->world || c->fd[who] >= 0) {
            close(fd);
            close(lfd);
            return -1;
        }
        c->fd[who] = fd;
    }
    close(lfd);
    sock_path(c, c->rank, &sa);
    unlink(sa.sun_path);
    for (int p = 0; p < c->world; ++p) {
        if (p == c->rank) continue;
        setsockopt(c->fd[p], SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
        setsockopt(c->fd[p], SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
    }
    return 0;
}

/* ---- worker ------------------------------------------------------------------------------------------------ */
static int copy_d2h(struct stub_comm* c, void* host, const void* dev, size_t bytes) {
    if (!bytes) return 0;
    if (hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->copy_stream) != hipSuccess) return -1;
    return hipStreamSynchronize(c->copy_stream) == hipSuccess ? 0 : -1;
}
static int copy_h2d(struct stub_comm* c, void* dev, const void* host, size_t bytes) {
    if (!bytes) return 0;
    if (hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, c->copy_stream) != hipSuccess) return -1;
    return hipStreamSynchronize(c->copy_stream) == hipSuccess ? 0 : -1;
}

/* the point-to-point operations of one group: all of them progress together (non-blocking sockets, poll) */
static void run_p2p_group(struct stub_comm* c, stub_op** ops, int n) {
    struct { uint32_t magic, pad; uint64_t bytes; } hdr[64];
    for (int i = 0; i < n; ++i) {
        stub_op* o = ops[i];
        if (hipEventSynchronize(o->ready) != hipSuccess) o->failed = 1;
        o->host = (char*)malloc(o->bytes ? o->bytes : 1);
        if (!o->host) o->failed = 1;
        if (!o->failed && o->kind == OP_SEND && copy_d2h(c, o->host, o->buf, o->bytes)) o->failed = 1;
        hdr[i].magic = STUB_MAGIC;
        hdr[i].pad = 0;
        hdr[i].bytes = o->bytes;
        if (o->failed) o->finished = 1;
    }
    /* Operations towards one peer share its socket: they proceed in posting order, one at a time per (peer, direction) */
    const double t_end = now_s() + STUB_TIMEOUT_S;
    for (;;) {
        struct pollfd pf[64];
        int who[64], np = 0, left = 0;
        for (int i = 0; i < n; ++i) {
            stub_op* o = ops[i];
            if (o->finished) continue;
            ++left;
            int blocked = 0;   /* an earlier unfinished op on the same socket and direction goes first */
            for (int j = 0; j < i; ++j)
                if (!ops[j]->finished && ops[j]->peer == o->peer && ops[j]->kind == o->kind) blocked = 1;
            if (blocked) continue;
            pf[np].fd = c->fd[o->peer];
            pf[np].events = o->kind == OP_SEND ? POLLOUT : POLLIN;
            pf[np].revents = 0;
            who[np++] = i;
        }
        if (!left) break;
        const int pr = poll(pf, (nfds_t)np, 1000);
        if (pr < 0 && errno != EINTR) {
            for (int i = 0; i < n; ++i)
                if (!ops[i]->finished) ops[i]->failed = ops[i]->finished = 1;
            break;
        }
        if (now_s() > t_end) {
            for (int i = 0; i < n; ++i)
                if (!ops[i]->finished) {
                    stub_log(c, "%s with peer %ld timed out after %ld bytes (no matching operation on the peer?)",
                             ops[i]->kind == OP_SEND ? "send" : "recv", ops[i]->peer, (long)ops[i]->done_bytes);
                    ops[i]->failed = ops[i]->finished = 1;
                }
            break;
        }
        for (int k = 0; k < np; ++k) {
            if (!(pf[k].revents & (POLLIN | POLLOUT | POLLERR | POLLHUP))) continue;
            const int i = who[k];
            stub_op* o = ops[i];
            const int fd = pf[k].fd;
            if (o->hdr_done < sizeof hdr[0]) {
                char* hp = (char*)&hdr[i] + o->hdr_done;
                const size_t want = sizeof hdr[0] - o->hdr_done;
                const ssize_t g = o->kind == OP_SEND ? send(fd, hp, want, MSG_NOSIGNAL | MSG_DONTWAIT) : recv(fd, hp, want, MSG_DONTWAIT);
                if (g < 0 && (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR)) continue;
                if (g <= 0) {
                    stub_log(c, "%s: connection to peer %ld lost (%ld)", o->kind == OP_SEND ? "send" : "recv", o->peer, (long)errno);
                    o->failed = o->finished = 1;
                    continue;
                }
                o->hdr_done += (size_t)g;
                if (o->hdr_done == sizeof hdr[0] && o->kind == OP_RECV && (hdr[i].magic != STUB_MAGIC || hdr[i].bytes != o->bytes)) {
                    stub_log(c, "%s: receive of %ld bytes met a send of %ld bytes: sends and receives do not pair up", "MISMATCH",
                             (long)o->bytes, (long)hdr[i].bytes);
                    o->failed = o->finished = 1;
                }
                if (o->hdr_done < sizeof hdr[0]) continue;
                if (o->bytes == 0) o->finished = 1;
                continue;
            }
            const size_t want = o->bytes - o->done_bytes;
            const ssize_t g = o->kind == OP_SEND ? send(fd, o->host + o->done_bytes, want, MSG_NOSIGNAL | MSG_DONTWAIT)
                                                 : recv(fd, o->host + o->done_bytes, want, MSG_DONTWAIT);
            if (g < 0 && (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR)) continue;
            if (g <= 0) {
                stub_log(c, "%s: connection to peer %ld lost (%ld)", o->kind == OP_SEND ? "send" : "recv", o->peer, (long)errno);
                o->failed = o->finished = 1;
                continue;
            }
            o->done_bytes += (size_t)g;
            if (o->done_bytes == o->bytes) o->finished = 1;
        }
    }
    for (int i = 0; i < n; ++i) {
        stub_op* o = ops[i];
        if (!o->failed && o->kind == OP_RECV && copy_h2d(c, o->buf, o->host, o->bytes)) o->failed = 1;
        free(o->host);
        o->host = NULL;
    }
}

/* all-reduce(sum) through rank 0: small counts only (the library counts ranks with one int64) */
static void run_allreduce(struct stub_comm* c, stub_op* o) {
    const size_t es = dtype_size(o->dtype), cnt = o->bytes / es;
    char* acc = (char*)malloc(o->bytes ? o->bytes : 1);
    char* tmp = (char*)malloc(o->bytes ? o->bytes : 1);
    if (!acc || !tmp || hipEventSynchronize(o->ready) != hipSuccess || copy_d2h(c, acc, o->src, o->bytes)) o->failed = 1;
    if (!o->failed) {
        if (c->rank == 0) {
            for (int p = 1; p < c->world && !o->failed; ++p) {
                if (full_io(c->fd[p], tmp, o->bytes, 0)) { o->failed = 1; break; }
                for (size_t i = 0; i < cnt; ++i) {
                    if (o->dtype == ncclFloat32) ((float*)acc)[i] += ((float*)tmp)[i];
                    else if (o->dtype == ncclFloat64) ((double*)acc)[i] += ((double*)tmp)[i];
                    else if (es == 8) ((int64_t*)acc)[i] += ((int64_t*)tmp)[i];
                    else if (es == 4) ((int32_t*)acc)[i] += ((int32_t*)tmp)[i];
                    else o->failed = 1;
                }
            }
            for (int p = 1; p < c->world && !o->failed; ++p)
                if (full_io(c->fd[p], acc, o->bytes, 1)) o->failed = 1;
        } else {
            if (full_io(c->fd[0], acc, o->bytes, 1) || full_io(c->fd[0], acc, o->bytes, 0)) o->failed = 1;
        }
    }
    if (!o->failed && copy_h2d(c, o->buf, acc, o->bytes)) o->failed = 1;
    free(acc);
    free(tmp);
}

static void* worker_main(void* arg) {
    struct stub_comm* c = (struct stub_comm*)arg;
    (void)hipSetDevice(c->device);
    for (;;) {
        stub_op* batch[64];
        int n = 0;
        pthread_mutex_lock(&c->mu);
        while (!c->q_head && !c->stop) pthread_cond_wait(&c->cv_work, &c->mu);
        if (!c->q_head && c->stop) {
            pthread_mutex_unlock(&c->mu);
            return NULL;
        }
        const int g = c->q_head->group;
        while (c->q_head && c->q_head->group == g && n < 64) {
            batch[n++] = c->q_head;
            c->q_head = c->q_head->next;
        }
        if (!c->q_head) c->q_tail = NULL;
        pthread_mutex_unlock(&c->mu);
        if (batch[0]->kind == OP_ALLREDUCE) {
            for (int i = 0; i < n; ++i) run_allreduce(c, batch[i]);
        } else {
            run_p2p_group(c, batch, n);
        }
        pthread_mutex_lock(&c->mu);
        for (int i = 0; i < n; ++i) {
            if (batch[i]->failed) c->async_error = 1;
            batch[i]->finished = 2;   /* visible to the waiting host function */
        }
        pthread_cond_broadcast(&c->cv_done);
        pthread_mutex_unlock(&c->mu);
    }
}

/* runs on the caller's stream (hipLaunchHostFunc): later work on that stream starts after the transfer */
struct wait_arg { struct stub_comm* c; stub_op* o; };
static void wait_for_op(void* p) {
    struct wait_arg* w = (struct wait_arg*)p;
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    ts.tv_sec += 2 * STUB_TIMEOUT_S;
    pthread_mutex_lock(&w->c->mu);
    while (w->o->finished != 2)
        if (pthread_cond_timedwait(&w->c->cv_done, &w->c->mu, &ts) == ETIMEDOUT) break;
    pthread_mutex_unlock(&w->c->mu);
    free(w);
}

static ncclResult_t post(struct stub_comm* c, stub_op* first, hipStream_t* streams) {
    /* `first` is a list (->next) of operations of one group, streams[i] the stream of the i-th */
    int i = 0;
    for (stub_op* o = first; o; o = o->next, ++i) {
        if (hipEventCreateWithFlags(&o->ready, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
        if (hipEventRecord(o->ready, streams[i]) != hipSuccess) return ncclUnhandledCudaError;
    }
    pthread_mutex_lock(&c->mu);
    const int g = c->next_group++;
    stub_op* last = first;
    for (stub_op* o = first; o; o = o->next) {
        o->group = g;
        o->all_next = c->all;
        c->all = o;
        last = o;
    }
    if (c->q_tail) c->q_tail->next = first;
    else c->q_head = first;
    c->q_tail = last;
    pthread_cond_signal(&c->cv_work);
    pthread_mutex_unlock(&c->mu);
    i = 0;
    for (stub_op* o = first; o; ++i) {
        stub_op* nx = (o == last) ? NULL : o->next;   /* ->next may already be spliced onto later posts */
        struct wait_arg* w = (struct wait_arg*)malloc(sizeof *w);
        if (!w) return ncclSystemError;
        w->c = c;
        w->o = o;
        if (hipLaunchHostFunc(streams[i], wait_for_op, w) != hipSuccess) return ncclUnhandledCudaError;
        o = nx;
    }
    return ncclSuccess;
}

static ncclResult_t submit(struct stub_comm* c, stub_op* o, hipStream_t stream) {
    if (c->async_error) {
        free(o);
        return ncclRemoteError;
    }
    if (t_group_depth > 0) {
        if (t_group_n >= 64) {
            free(o);
            return ncclInvalidUsage;
        }
        o->next = NULL;
        if (t_group_tail) t_group_tail->next = o;
        else t_group_head = o;
        t_group_tail = o;
        t_group_stream[t_group_n++] = stream;
        return ncclSuccess;
    }
    o->next = NULL;
    return post(c, o, &stream);
}

/* every op of a pending group belongs to one communicator in this stub (the library has one) */
static struct stub_comm* t_group_comm = NULL;

/* ---- the nine entry points ------------------------------------------------------------------------------------ */
__attribute__((visibility("default"))) const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
    case ncclSuccess: return "no error (rccl_stub)";
    case ncclUnhandledCudaError: return "unhandled HIP error (rccl_stub)";
    case ncclSystemError: return "system error: socket / rendezvous (rccl_stub)";
    case ncclInvalidArgument: return "invalid argument (rccl_stub)";
    case ncclInvalidUsage: return "invalid usage (rccl_stub)";
    case ncclRemoteError: return "an earlier transfer failed or did not pair up (rccl_stub)";
    default: return "internal error (rccl_stub)";
    }
}

__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    unsigned long long r = 0;
    int fd = open("/dev/urandom", O_RDONLY);
    if (fd >= 0) {
        if (read(fd, &r, sizeof r) != (ssize_t)sizeof r) r = 0;
        close(fd);
    }
    r ^= (unsigned long long)getpid() << 32 ^ (unsigned long long)time(NULL);
    const char* dir = getenv("TMPDIR");
    snprintf(id->internal, sizeof id->internal, "%s/gaast_stub_%016llx", (dir && *dir && strlen(dir) < 40) ? dir : "/tmp", r);
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks || nranks > 64) return ncclInvalidArgument;
    id.internal[sizeof id.internal - 1] = 0;
    if (strncmp(id.internal, "/", 1) != 0 || !strstr(id.internal, "gaast_stub_")) return ncclInvalidArgument;
    struct stub_comm* c = (struct stub_comm*)calloc(1, sizeof *c);
    if (!c) return ncclSystemError;
    c->rank = rank;
    c->world = nranks;
    snprintf(c->prefix, sizeof c->prefix, "%.95s", id.internal);
    c->fd = (int*)malloc(sizeof(int) * (size_t)nranks);
    for (int p = 0; p < nranks; ++p) c->fd[p] = -1;
    if (hipGetDevice(&c->device) != hipSuccess || hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) {
        free(c->fd);
        free(c);
        return ncclUnhandledCudaError;
    }
    if (connect_all(c)) {
        stub_log(c, "rendezvous failed (%s, errno %ld, world %ld)", c->prefix, (long)errno, (long)nranks);
        for (int p = 0; p < nranks; ++p)
            if (c->fd[p] >= 0) close(c->fd[p]);
        (void)hipStreamDestroy(c->copy_stream);
        free(c->fd);
        free(c);
        return ncclSystemError;
    }
    pthread_mutex_init(&c->mu, NULL);
    pthread_cond_init(&c->cv_work, NULL);
    pthread_cond_init(&c->cv_done, NULL);
    if (pthread_create(&c->worker, NULL, worker_main, c)) {
        free(c->fd);
        free(c);
        return ncclSystemError;
    }
    *comm = c;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclInvalidArgument;
    pthread_mutex_lock(&c->mu);
    c->stop = 1;
    pthread_cond_signal(&c->cv_work);
    pthread_mutex_unlock(&c->mu);
    pthread_join(c->worker, NULL);
    const int err = c->async_error;
    if (getenv("GAAST_RCCL_STUB_VERBOSE"))
        stub_log(c, "%s: %ld sends, %ld receives", "destroy", c->n_sends, c->n_recvs);
    for (stub_op* o = c->all; o;) {
        stub_op* nx = o->all_next;
        if (o->ready) (void)hipEventDestroy(o->ready);
        free(o);
        o = nx;
    }
    for (int p = 0; p < c->world; ++p)
        if (c->fd[p] >= 0) close(c->fd[p]);
    (void)hipStreamDestroy(c->copy_stream);
    free(c->fd);
    free(c);
    return err ? ncclRemoteError : ncclSuccess;
}

static ncclResult_t p2p(int kind, void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t stream) {
    if (!c || peer < 0 || peer >= c->world || peer == c->rank) return ncclInvalidArgument;
    const size_t bytes = count * dtype_size(dt);
    if (bytes > STUB_MAX_OP_BYTES) return ncclInvalidArgument;
    stub_op* o = (stub_op*)calloc(1, sizeof *o);
    if (!o) return ncclSystemError;
    o->kind = kind;
    o->peer = peer;
    o->buf = buf;
    o->bytes = bytes;
    o->dtype = dt;
    if (kind == OP_SEND) c->n_sends++;
    else c->n_recvs++;
    t_group_comm = c;
    return submit(c, o, stream);
}

__attribute__((visibility("default"))) ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c,
                                                          hipStream_t stream) {
    return p2p(OP_SEND, (void*)buf, count, dt, peer, c, stream);
}

__attribute__((visibility("default"))) ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c,
                                                          hipStream_t stream) {
    return p2p(OP_RECV, buf, count, dt, peer, c, stream);
}

__attribute__((visibility("default"))) ncclResult_t ncclAllReduce(const void* src, void* dst, size_t count, ncclDataType_t dt,
                                                               ncclRedOp_t op, ncclComm_t c, hipStream_t stream) {
    if (!c || op != ncclSum || t_group_depth > 0) return ncclInvalidUsage;
    const size_t bytes = count * dtype_size(dt);
    if (bytes > ((size_t)1 << 20)) return ncclInvalidArgument;
    stub_op* o = (stub_op*)calloc(1, sizeof *o);
    if (!o) return ncclSystemError;
    o->kind = OP_ALLREDUCE;
    o->peer = -1;
    o->buf = dst;
    o->src = src;
    o->bytes = bytes;
    o->dtype = dt;
    return submit(c, o, stream);
}

__attribute__((visibility("default"))) ncclResult_t ncclGroupStart(void) {
    ++t_group_depth;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclGroupEnd(void) {
    if (t_group_depth <= 0) return ncclInvalidUsage;
    if (--t_group_depth > 0) return ncclSuccess;
    stub_op* first = t_group_head;
    t_group_head = t_group_tail = NULL;
    const int n = t_group_n;
    t_group_n = 0;
    (void)n;
    if (!first) return ncclSuccess;   /* an empty group */
    return post(t_group_comm, first, t_group_stream);
}
