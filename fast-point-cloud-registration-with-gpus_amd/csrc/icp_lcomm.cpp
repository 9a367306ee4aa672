// icp_lcomm.cpp -- the loop's one exchange step between the ranks of a node, through POSIX shared memory.
//
// Why not the device collective: in the single-node fast path the per-row moment sums already arrive in HOST
// memory (the host adds the rows while the resident kernel finishes), and what the ranks exchange is
// one vector of 32 doubles.  A 256-byte all-reduce is pure latency: ~15-20 us through RCCL (launch + protocol) --
// more than the whole 13 us iteration -- against ~1 us for a few cache lines between processes of one node.  Every
// rank adds the slots in rank order, so all ranks hold bit-identical sums (and then solve the same 3x3 / 6x6).
// The RCCL route (icp_comm_init) stays for vectors that must remain on the device or ranks on different nodes.
//
// Segment: world slots x 2 buffers (sequence parity).  A rank publishes {v, seq} into buffer seq & 1, then waits for
// every rank's slot of that buffer to carry seq.  Writing sequence s + 2 into the same buffer is safe: a rank gets
// there only after it has seen every other rank's s + 1, which they publish after having read all of s.
#include "icp_lcomm.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/icp_mi355x.h"

namespace icp {

namespace {
constexpr int kMaxCount = ICP_NMOM;
struct alignas(64) Slot {
    std::atomic<uint64_t> seq;
    double v[kMaxCount];
};
struct Header {
    std::atomic<uint32_t> attached;
    uint32_t world;
    char pad[56];
};
}  // namespace

struct LocalComm {
    int rank = 0, world = 1;
    uint64_t seq = 0;
    size_t bytes = 0;
    void* base = nullptr;
    char name[64] = {0};
    Slot* slot(int buf, int r) const { return reinterpret_cast<Slot*>((char*)base + sizeof(Header)) + (size_t)buf * world + r; }
    Header* header() const { return reinterpret_cast<Header*>(base); }
};

int lcomm_create(const void* id, int rank, int world, LocalComm** out, std::string& err)
{
    if (!id || !out || world < 1 || rank < 0 || rank >= world) { err = "bad local communicator arguments"; return ICP_ERR_INVALID; }
    LocalComm* c = new (std::nothrow) LocalComm();
    if (!c) { err = "out of memory"; return ICP_ERR_NOMEM; }
    c->rank = rank;
    c->world = world;
    const unsigned char* b = static_cast<const unsigned char*>(id);
    std::snprintf(c->name, sizeof c->name, "/icp_mi355x_%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x", b[0], b[1], b[2], b[3], b[4], b[5],
                  b[6], b[7], b[8], b[9], b[10], b[11]);
    c->bytes = sizeof(Header) + 2 * (size_t)world * sizeof(Slot);
    const auto t0 = std::chrono::steady_clock::now();
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd >= 0 && ftruncate(fd, (off_t)c->bytes) != 0) { close(fd); shm_unlink(c->name); fd = -1; }
    } else {
        // rank 0 creates the segment; the others wait for it to reach its size
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 60.0) {
            fd = shm_open(c->name, O_RDWR, 0600);
            if (fd >= 0) {
                struct stat st;
                if (fstat(fd, &st) == 0 && (size_t)st.st_size >= c->bytes) break;
                close(fd);
                fd = -1;
            }
            usleep(200);
        }
    }
    if (fd < 0) { err = std::string("shared memory segment ") + c->name + " unavailable"; delete c; return ICP_ERR_HIP; }
    c->base = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->base == MAP_FAILED) { err = "mmap of the shared segment failed"; if (rank == 0) shm_unlink(c->name); delete c; return ICP_ERR_HIP; }
    // a fresh segment is zero-filled: sequence numbers start at 0, the first exchange uses 1
    if (rank == 0) c->header()->world = (uint32_t)world;
    c->header()->attached.fetch_add(1, std::memory_order_acq_rel);
    while (c->header()->attached.load(std::memory_order_acquire) < (uint32_t)world) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 60.0) {
            err = "not every rank attached to the shared segment";
            lcomm_destroy(c);
            return ICP_ERR_HIP;
        }
        usleep(100);
    }
    if (rank == 0) shm_unlink(c->name);  // everybody holds a mapping: the name can go
    *out = c;
    return ICP_OK;
}

void lcomm_destroy(LocalComm* c)
{
    if (!c) return;
    if (c->base && c->base != MAP_FAILED) munmap(c->base, c->bytes);
    delete c;
}

int lcomm_allreduce_sum_f64(LocalComm* c, double* v, int count, std::string& err)
{
    if (!c || !v || count < 0 || count > kMaxCount) { err = "bad all-reduce arguments"; return ICP_ERR_INVALID; }
    if (c->world == 1) return ICP_OK;
    const uint64_t s = ++c->seq;
    const int buf = (int)(s & 1);
    Slot* mine = c->slot(buf, c->rank);
    std::memcpy(mine->v, v, (size_t)count * sizeof(double));
    mine->seq.store(s, std::memory_order_release);
    double sum[kMaxCount];
    for (int k = 0; k < count; ++k) sum[k] = 0.0;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    // Wait for ALL slots first, sweeping over them: the lines of the ranks that are already there are pulled into
    // this core's cache while the last rank is still on its way, so what is left after its arrival is one line
    // transfer, not world - 1 of them one after the other.  The sum is then taken in rank order.
    for (int pending = c->world; pending > 0;) {
        pending = 0;
        for (int r = 0; r < c->world; ++r) {
            const Slot* sl = c->slot(buf, r);
            if (sl->seq.load(std::memory_order_acquire) != s) { ++pending; continue; }
            __builtin_prefetch(&sl->v[7]); __builtin_prefetch(&sl->v[15]); __builtin_prefetch(&sl->v[23]); __builtin_prefetch(&sl->v[31]);
        }
        if (pending && (++spins & 0xfff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.0) {
            err = "a rank did not reach the exchange (30 s)";
            return ICP_ERR_HIP;
        }
    }
    for (int r = 0; r < c->world; ++r) {
        const Slot* sl = c->slot(buf, r);
        for (int k = 0; k < count; ++k) sum[k] += sl->v[k];   // rank order: the same bits on every rank
    }
    std::memcpy(v, sum, (size_t)count * sizeof(double));
    return ICP_OK;
}

}  // namespace icp
