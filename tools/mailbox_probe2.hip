// mailbox_probe2.hip -- which ingredient of the resident-kernel mailbox makes the second message invisible?
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <chrono>
#include <cstdio>
#include <cstring>
struct MB { float rt[12]; double seq; int cmd; int pad; };
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k(const MB* mb, volatile double* out, int rounds, int mode, float* scratch)
{
    for (int i = 1; i <= rounds; ++i) {
        double s = 0;
        int spins = 0;
        for (; spins < (1 << 21); ++spins) {
            s = __hip_atomic_load(&mb->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (s == (double)i) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (s != (double)i) { if (threadIdx.x == 0) *out = -(double)i; return; }
        if (mode & 1) __atomic_thread_fence(__ATOMIC_ACQUIRE);
        float v = 0;
        if (mode & 2) { if (threadIdx.x < 12) v = __hip_atomic_load(&mb->rt[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
        if (mode & 4) scratch[threadIdx.x] = v;
        if (mode & 8) __threadfence_system();
        if (threadIdx.x == 0) __hip_atomic_store((double*)out, (double)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static int run(MB* mb, int mode, int hostmode)
{
    double* out = nullptr; float* scratch = nullptr;
    CK(hipHostMalloc((void**)&out, 64, hipHostMallocMapped));
    CK(hipMalloc((void**)&scratch, 4096));
    *out = 0;
    std::memset((void*)mb, 0, sizeof(MB)); _mm_sfence();
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int rounds = 200;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, st, mb, out, rounds, mode, scratch);
    int reached = 0;
    for (int i = 1; i <= rounds; ++i) {
        if (hostmode & 1) for (int q = 0; q < 12; ++q) mb->rt[q] = (float)(i + q);
        if (hostmode & 2) mb->cmd = i;
        _mm_sfence();
        *(volatile double*)&mb->seq = (double)i;
        _mm_sfence();
        const auto w0 = std::chrono::steady_clock::now();
        bool ok = false;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() < 1.0) { const double o = *(volatile double*)out; if (o == (double)i) { ok = true; break; } if (o < 0) break; }
        if (!ok) break;
        reached = i;
    }
    CK(hipStreamSynchronize(st));
    std::printf("kernel mode %2d host mode %d: reached round %d of %d (out=%.0f)\n", mode, hostmode, reached, rounds, *out);
    CK(hipStreamDestroy(st)); CK(hipHostFree(out)); CK(hipFree(scratch));
    return 0;
}
int main()
{
    MB* mb = nullptr;
    CK(hipExtMallocWithFlags((void**)&mb, 256, hipDeviceMallocFinegrained));
    for (int hm = 0; hm < 4; ++hm)
        for (int m : {0, 1, 2, 3, 8, 15}) run(mb, m, hm);
    return 0;
}
