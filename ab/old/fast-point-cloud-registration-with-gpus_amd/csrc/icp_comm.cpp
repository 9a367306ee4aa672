// icp_comm.cpp -- the loop's one collective, issued by the library itself: an in-place SUM all-reduce of the
// ICP_NMOM-double moment vector over RCCL (xGMI inside a node), on the stream the loop runs on, directly behind
// the finalize kernel -- no Python, no host synchronisation between the kernels and the collective.
// librccl is resolved at run time (dlopen): the copy already loaded by the host application (e.g. torch's
// bundled librccl) is reused when there is one, so the library itself keeps no link-time RCCL dependency and
// single-GPU users never load it.  The reference has no collective at all; this is new design.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>

#include "../../include/icp_mi355x.h"
#include "icp_comm.h"

namespace icp {

namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
    bool ok() const { return handle && GetUniqueId && CommInitRank && CommDestroy && AllReduce && GetErrorString; }
};

Rccl& rccl()
{
    static Rccl r;
    if (r.handle || !r.error.empty()) return r;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)  // a copy that is already in the process wins (one RCCL per process)
        if ((r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!r.handle)
        for (const char* n : names)
            if ((r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!r.handle) {
        r.error = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?");
        return r;
    }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
    if (!r.ok()) r.error = "librccl is missing a required symbol";
    return r;
}
}  // namespace

static_assert(sizeof(ncclUniqueId) == ICP_COMM_ID_BYTES, "ICP_COMM_ID_BYTES must match ncclUniqueId");

int comm_unique_id(void* out, std::string& err)
{
    Rccl& r = rccl();
    if (!r.ok()) { err = r.error; return ICP_ERR_HIP; }
    ncclUniqueId id;
    const ncclResult_t rc = r.GetUniqueId(&id);
    if (rc != ncclSuccess) { err = std::string("ncclGetUniqueId: ") + r.GetErrorString(rc); return ICP_ERR_HIP; }
    std::memcpy(out, &id, sizeof id);
    return ICP_OK;
}

int comm_init(const void* id_bytes, int rank, int world, void** comm_out, std::string& err)
{
    Rccl& r = rccl();
    if (!r.ok()) { err = r.error; return ICP_ERR_HIP; }
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof id);
    ncclComm_t comm = nullptr;
    const ncclResult_t rc = r.CommInitRank(&comm, world, id, rank);
    if (rc != ncclSuccess) { err = std::string("ncclCommInitRank: ") + r.GetErrorString(rc); return ICP_ERR_HIP; }
    *comm_out = comm;
    return ICP_OK;
}

void comm_destroy(void* comm)
{
    Rccl& r = rccl();
    if (comm && r.ok()) (void)r.CommDestroy((ncclComm_t)comm);
}

int comm_allreduce_sum_f64(void* comm, double* dev_buf, int count, hipStream_t stream, std::string& err)
{
    Rccl& r = rccl();
    if (!r.ok()) { err = r.error; return ICP_ERR_HIP; }
    const ncclResult_t rc = r.AllReduce(dev_buf, dev_buf, (size_t)count, ncclFloat64, ncclSum, (ncclComm_t)comm, stream);
    if (rc != ncclSuccess) { err = std::string("ncclAllReduce: ") + r.GetErrorString(rc); return ICP_ERR_HIP; }
    return ICP_OK;
}

}  // namespace icp
