// mailbox_probe.hip -- how long does a host -> waiting-kernel message take on this machine?
// A kernel spins on a flag the host bumps and answers into pinned host memory; the host measures the round trip.
//   flag in pinned host memory (GPU polls over PCIe)   vs   flag in fine-grained device memory written by the
//   CPU through the BAR (GPU polls locally), with 1 or 128 polling blocks.
// build: hipcc --offload-arch=gfx950 -O2 -o bin/mailbox_probe tools/mailbox_probe.hip
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void ping(const volatile unsigned long long* in, volatile unsigned long long* out, int rounds, int budget)
{
    for (int i = 1; i <= rounds; ++i) {
        int spins = 0;
        while (__hip_atomic_load((const unsigned long long*)in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned long long)i) {
            if (++spins > budget) return;  // the host went away
            __builtin_amdgcn_s_sleep(2);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0)
            __hip_atomic_store((unsigned long long*)out, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }

static int run(const char* name, volatile unsigned long long* in_host_view, unsigned long long* in_dev, int blocks)
{
    unsigned long long* out = nullptr;
    CK(hipHostMalloc((void**)&out, 64, hipHostMallocMapped));
    *out = 0;
    *in_host_view = 0;
    const int rounds = 2000;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipLaunchKernelGGL(ping, dim3(blocks), dim3(64), 0, st, in_dev, out, rounds, 1 << 22);
    CK(hipGetLastError());
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 1; i <= rounds; ++i) {
        *in_host_view = (unsigned long long)i;
        std::atomic_thread_fence(std::memory_order_seq_cst);
        const auto w0 = std::chrono::steady_clock::now();
        while (*(volatile unsigned long long*)out < (unsigned long long)i)
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() > 2.0) { std::printf("%s: timeout at %d\n", name, i); *in_host_view = ~0ull; hipStreamSynchronize(st); return 1; }
    }
    const double us = 1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / rounds;
    CK(hipStreamSynchronize(st));
    std::printf("%-44s blocks %4d: round trip %.2f us\n", name, blocks, us);
    CK(hipStreamDestroy(st));
    CK(hipHostFree(out));
    return 0;
}

int main()
{
    unsigned long long* hp = nullptr;
    CK(hipHostMalloc((void**)&hp, 64, hipHostMallocMapped));
    run("flag in pinned host memory", hp, hp, 1);
    run("flag in pinned host memory", hp, hp, 128);
    unsigned long long* fg = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&fg, 4096, hipDeviceMallocFinegrained);
    std::printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
    if (e == hipSuccess) {
        std::signal(SIGSEGV, on_segv);
        std::signal(SIGBUS, on_segv);
        if (sigsetjmp(jb, 1) == 0) {
            *(volatile unsigned long long*)fg = 0;  // is it CPU-addressable at all?
            std::printf("fine-grained device memory is CPU-writable\n");
            run("flag in fine-grained device memory (BAR)", fg, fg, 1);
            run("flag in fine-grained device memory (BAR)", fg, fg, 128);
        } else {
            std::printf("fine-grained device memory is NOT CPU-addressable here\n");
        }
    }
    unsigned long long* mg = nullptr;
    e = hipMallocManaged((void**)&mg, 4096);
    std::printf("hipMallocManaged: %s\n", hipGetErrorString(e));
    if (e == hipSuccess) {
        (void)hipMemAdvise(mg, 4096, hipMemAdviseSetCoarseGrain, 0);
        if (sigsetjmp(jb, 1) == 0) {
            *(volatile unsigned long long*)mg = 0;
            run("flag in managed memory", mg, mg, 1);
        } else std::printf("managed memory write faulted\n");
    }
    return 0;
}
