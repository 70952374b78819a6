// RCCL behind a dlopen: the one inter-rank exchange of the path (gather of result rows, include/gaast_hip.h
// "multi-GPU").  librccl is resolved at first use so that libgaast_hip.so loads on hosts without it and shares
// the copy a host process (e.g. torch) may already have mapped.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>
#include <string>

namespace gaast {

struct Comm {
    void* handle = nullptr;   // ncclComm_t
    int rank = -1, world = 0;
    hipStream_t stream = nullptr;   // transfers run here, compute on the library stream
    bool active() const { return handle != nullptr; }
};

// All return 0 on success; on failure *err describes the RCCL / loader error.
// Which shared object provides the nccl* entry points: NULL / "" = the system's librccl (default).  Must come before the
// first use; fails once another library has been loaded.
int comm_set_library(const char* path, std::string* err);
const char* comm_library();
int comm_unique_id(void* id128, std::string* err);
int comm_init(Comm& c, const void* id128, int rank, int world, std::string* err);
int comm_destroy(Comm& c, std::string* err);
int comm_group_start(std::string* err);
int comm_group_end(std::string* err);
// elem_size 4 (f32) or 8 (f64)
int comm_send(Comm& c, const void* buf, size_t count, int elem_size, int peer, std::string* err);
int comm_recv(Comm& c, void* buf, size_t count, int elem_size, int peer, std::string* err);
// in-place all-reduce(sum) of `count` int64 values on c.stream
int comm_allreduce_sum_i64(Comm& c, void* buf, size_t count, std::string* err);

}  // namespace gaast
