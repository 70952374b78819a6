// comm.cpp -- see comm.hpp.  Only the handful of RCCL entry points the gather needs, resolved with dlsym.
#include "comm.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

namespace gaast {
namespace {

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::string g_library;   // comm_set_library: the file to load instead of the system's librccl

template <typename F>
bool sym(void* so, const char* name, F* out) {
    *out = reinterpret_cast<F>(dlsym(so, name));
    return *out != nullptr;
}

int load(std::string* err) {
    if (g_rccl.so) return 0;
    void* so = nullptr;
    if (!g_library.empty()) {
        // the host named the transport (gaast_hip_comm_set_library): that file or nothing; RTLD_LOCAL keeps its
        // nccl* symbols away from a librccl the process may also have mapped
        so = dlopen(g_library.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!so) {
            *err = std::string("cannot load ") + g_library + ": " + dlerror();
            return 1;
        }
    } else {
        // the soname first: a copy already mapped by the host process (torch ships one) is reused
        so = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!so) so = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!so) {
            *err = std::string("cannot load librccl: ") + dlerror();
            return 1;
        }
    }
    Rccl r;
    r.so = so;
    const bool ok = sym(so, "ncclGetUniqueId", &r.GetUniqueId) && sym(so, "ncclCommInitRank", &r.CommInitRank) &&
                    sym(so, "ncclCommDestroy", &r.CommDestroy) && sym(so, "ncclSend", &r.Send) &&
                    sym(so, "ncclRecv", &r.Recv) && sym(so, "ncclAllReduce", &r.AllReduce) &&
                    sym(so, "ncclGroupStart", &r.GroupStart) && sym(so, "ncclGroupEnd", &r.GroupEnd) &&
                    sym(so, "ncclGetErrorString", &r.GetErrorString);
    if (!ok) {
        *err = (g_library.empty() ? std::string("librccl") : g_library) + " lacks an expected nccl* symbol";
        dlclose(so);
        return 1;
    }
    g_rccl = r;
    return 0;
}

int fail(const char* what, ncclResult_t r, std::string* err) {
    *err = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error");
    return 1;
}

#define RCCL_TRY(call, what)                          \
    do {                                              \
        const ncclResult_t r__ = (call);              \
        if (r__ != ncclSuccess) return fail(what, r__, err); \
    } while (0)

ncclDataType_t dt(int elem_size) { return elem_size == 4 ? ncclFloat32 : ncclFloat64; }

}  // namespace

int comm_set_library(const char* path, std::string* err) {
    const std::string want = path ? path : "";
    if (g_rccl.so && want != g_library) {
        *err = "a collective library is already loaded (" + (g_library.empty() ? std::string("librccl") : g_library) + ")";
        return 1;
    }
    g_library = want;
    return 0;
}

const char* comm_library() { return g_library.c_str(); }

int comm_unique_id(void* id128, std::string* err) {
    if (load(err)) return 1;
    static_assert(sizeof(ncclUniqueId) == 128, "GAAST_COMM_ID_BYTES");
    ncclUniqueId id;
    RCCL_TRY(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(id128, &id, sizeof(id));
    return 0;
}

int comm_init(Comm& c, const void* id128, int rank, int world, std::string* err) {
    if (load(err)) return 1;
    if (c.handle) {
        *err = "a communicator already exists (gaast_hip_comm_destroy first)";
        return 1;
    }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    RCCL_TRY(g_rccl.CommInitRank(&comm, world, id, rank), "ncclCommInitRank");
    if (hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking) != hipSuccess) {
        g_rccl.CommDestroy(comm);
        *err = "hipStreamCreateWithFlags failed";
        return 1;
    }
    c.handle = comm;
    c.rank = rank;
    c.world = world;
    return 0;
}

int comm_destroy(Comm& c, std::string* err) {
    if (!c.handle) return 0;
    (void)hipStreamSynchronize(c.stream);
    const ncclResult_t r = g_rccl.CommDestroy(static_cast<ncclComm_t>(c.handle));
    (void)hipStreamDestroy(c.stream);
    c = Comm();
    if (r != ncclSuccess) return fail("ncclCommDestroy", r, err);
    return 0;
}

int comm_group_start(std::string* err) {
    RCCL_TRY(g_rccl.GroupStart(), "ncclGroupStart");
    return 0;
}
int comm_group_end(std::string* err) {
    RCCL_TRY(g_rccl.GroupEnd(), "ncclGroupEnd");
    return 0;
}
int comm_send(Comm& c, const void* buf, size_t count, int elem_size, int peer, std::string* err) {
    RCCL_TRY(g_rccl.Send(buf, count, dt(elem_size), peer, static_cast<ncclComm_t>(c.handle), c.stream), "ncclSend");
    return 0;
}
int comm_recv(Comm& c, void* buf, size_t count, int elem_size, int peer, std::string* err) {
    RCCL_TRY(g_rccl.Recv(buf, count, dt(elem_size), peer, static_cast<ncclComm_t>(c.handle), c.stream), "ncclRecv");
    return 0;
}
int comm_allreduce_sum_i64(Comm& c, void* buf, size_t count, std::string* err) {
    RCCL_TRY(g_rccl.AllReduce(buf, buf, count, ncclInt64, ncclSum, static_cast<ncclComm_t>(c.handle), c.stream),
             "ncclAllReduce");
    return 0;
}

}  // namespace gaast
