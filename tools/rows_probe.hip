// rows_probe.hip -- the communication floor of one resident pass: the host posts a 64-byte line (BAR-visible device
// memory), N blocks poll it, each answers with one ROW of `slots` doubles + a tag into pinned coherent host memory
// (system-scope stores, drained, then the tag -- the protocol of tail_close_row), the host adds the rows up in block
// order as their tags appear.  No computation at all: what is measured is message flight + detection + row flight +
// the host's ingest of N rows.  Variants: rows of 32 doubles with the tag in slot 31 (the full format), rows of 16
// with the tag in slot 0 (the compact format), and 16 + a separate 8-byte error array.
// build: hipcc --offload-arch=gfx950 -O2 -mavx -o bin/rows_probe tools/rows_probe.hip
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void answer(const unsigned int* mb, double* rows, double* errs, int stride, int tag_slot, int nslots, int use_err, int rounds)
{
    const int lane = threadIdx.x;
    double* row = rows + (size_t)blockIdx.x * stride;
    for (int i = 1; i <= rounds; ++i) {
        unsigned int word = 0;
        const long long give_up = (long long)wall_clock64() + 200000000ll;
        for (unsigned int spins = 1;; ++spins) {
            word = __hip_atomic_load(mb + (lane & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((unsigned int)__builtin_amdgcn_readlane((int)word, 7) == (unsigned int)i && (unsigned int)__builtin_amdgcn_readlane((int)word, 14) == (unsigned int)i) break;
            if ((spins & 63u) == 0u && (long long)wall_clock64() > give_up) return;
            __builtin_amdgcn_s_sleep(2);
        }
        if (lane < nslots && lane != tag_slot) __hip_atomic_store(&row[lane], (double)(blockIdx.x + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (use_err && lane == 0) __hip_atomic_store(&errs[blockIdx.x], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(&row[tag_slot], (double)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Device-side combine (round 3): the blocks of a group leave their rows in DEVICE memory (agent-scope stores,
// drained), draw a ticket on the group's counter, and the last arriver adds the group's rows in FIXED member order (the sums do
// not depend on who came last) and sends ONE row to the host.  groups = 8: group g = the blocks b with b % 8 == g (one XCD's
// blocks when workgroups are dealt to the XCDs round-robin) or b / (blocks / 8) (contiguous).  The host ingests 8 rows.
__global__ void answer_combine(const unsigned int* mb, double* dev_rows, unsigned int* tickets, double* host_rows, int groups, int by_xcd, int rounds)
{
    const int lane = threadIdx.x;
    const int per = gridDim.x / groups;
    const int g = by_xcd ? (int)blockIdx.x % groups : (int)blockIdx.x / per;
    const int member = by_xcd ? (int)blockIdx.x / groups : (int)blockIdx.x % per;
    double* row = dev_rows + (size_t)blockIdx.x * 16;
    for (int i = 1; i <= rounds; ++i) {
        unsigned int word = 0;
        const long long give_up = (long long)wall_clock64() + 200000000ll;
        for (unsigned int spins = 1;; ++spins) {
            word = __hip_atomic_load(mb + (lane & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((unsigned int)__builtin_amdgcn_readlane((int)word, 7) == (unsigned int)i && (unsigned int)__builtin_amdgcn_readlane((int)word, 14) == (unsigned int)i) break;
            if ((spins & 63u) == 0u && (long long)wall_clock64() > give_up) return;
            __builtin_amdgcn_s_sleep(2);
        }
        if (lane < 16) __hip_atomic_store(&row[lane], (double)(blockIdx.x + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned int ticket = 0;
        if (lane == 0) ticket = __hip_atomic_fetch_add(&tickets[g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = (unsigned int)__builtin_amdgcn_readfirstlane((int)ticket);
        if (ticket != (unsigned int)(i * per - 1)) continue;     // (the counter runs on: round i ends at i * per)
        // the group's last arriver: lane l adds slot l % 16 of the members l / 16, l / 16 + 4, ... in that order; then the four
        // partial sums of a slot are added in lane order -- the same order whoever closes the group
        double sum = 0.0;
        for (int m = lane >> 4; m < per; m += 4) {
            const int b = by_xcd ? m * groups + g : g * per + m;
            sum += __hip_atomic_load(&dev_rows[(size_t)b * 16 + (lane & 15)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        sum += __shfl_down(sum, 32, 64);
        sum += __shfl_down(sum, 16, 64);
        double* out = host_rows + (size_t)g * 16;
        if (lane >= 1 && lane < 16) __hip_atomic_store(&out[lane], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(&out[0], (double)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        (void)member;
    }
}

__attribute__((target("avx"))) static void post(unsigned int* mb, unsigned int tag)
{
    alignas(32) unsigned int line[16] = {0};
    line[7] = tag; line[14] = tag;
    _mm256_store_si256((__m256i*)mb, _mm256_load_si256((const __m256i*)line));
    _mm256_store_si256((__m256i*)(mb + 8), _mm256_load_si256((const __m256i*)(line + 8)));
    _mm_sfence();
}

static int g_mode = 0;   // 0: poll the rows in block order; 1: the same with the next rows prefetched; 2: sweep over all missing tags, then add
static int run(unsigned int* mb, int blocks, int stride, int tag_slot, int nslots, int use_err, const char* name)
{
    const int rounds = 3000;
    double *rows = nullptr, *errs = nullptr;
    CK(hipHostMalloc((void**)&rows, (size_t)blocks * stride * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostMalloc((void**)&errs, (size_t)blocks * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(rows, 0, (size_t)blocks * stride * sizeof(double));
    std::memset(errs, 0, (size_t)blocks * sizeof(double));
    std::memset(mb, 0, 64); _mm_sfence();
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipLaunchKernelGGL(answer, dim3(blocks), dim3(64), 0, st, mb, rows, errs, stride, tag_slot, nslots, use_err, rounds);
    CK(hipGetLastError());
    std::vector<double> rt, first, ingest;
    double sink = 0;
    for (int i = 1; i <= rounds; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        post(mb, (unsigned int)i);
        double mom[32] = {0};
        double t_first = 0;
        if (g_mode == 2) {
            // sweep: touch every missing tag in turn until all are there (the loads of different rows overlap), then add in order
            static unsigned char seen[1024];
            std::memset(seen, 0, sizeof seen);
            int left = blocks;
            while (left > 0) {
                for (int b = 0; b < blocks; ++b) {
                    if (seen[b]) continue;
                    const volatile double* tg = rows + (size_t)b * stride + tag_slot;
                    if (*tg == (double)i) { seen[b] = 1; --left; _mm_prefetch((const char*)(rows + (size_t)b * stride) + 64, _MM_HINT_T0); }
                }
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) { std::printf("%s: timeout round %d\n", name, i); return 1; }
            }
            t_first = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            for (int b = 0; b < blocks; ++b) {
                const double* row = rows + (size_t)b * stride;
                for (int k = 0; k < nslots; ++k) if (k != tag_slot) mom[k] += row[k];
            }
        } else
        for (int b = 0; b < blocks; ++b) {
            const volatile double* tg = rows + (size_t)b * stride + tag_slot;
            if (g_mode == 1)
                for (int a = 1; a <= 4; ++a)
                    if (b + a < blocks) _mm_prefetch((const char*)(rows + (size_t)(b + a) * stride + tag_slot), _MM_HINT_T0);
            while (*tg != (double)i)
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) { std::printf("%s: timeout round %d row %d\n", name, i, b); return 1; }
            if (b == 0) t_first = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            const double* row = rows + (size_t)b * stride;
            for (int k = 0; k < nslots; ++k) if (k != tag_slot) mom[k] += row[k];
            if (use_err) mom[31] += errs[b];
        }
        const double t1 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        sink += mom[1] + mom[31];
        if (i > rounds / 3) { rt.push_back(1e6 * t1); first.push_back(1e6 * t_first); }
    }
    CK(hipStreamSynchronize(st));
    std::sort(rt.begin(), rt.end()); std::sort(first.begin(), first.end());
    std::printf("%-46s blocks %4d: row 0 after %5.2f us, all rows added after %5.2f us (medians; p90 %5.2f)  [%g]\n", name, blocks, first[first.size() / 2],
                rt[rt.size() / 2], rt[rt.size() * 9 / 10], sink > 0 ? 0.0 : 1.0);
    CK(hipStreamDestroy(st)); CK(hipHostFree(rows)); CK(hipHostFree(errs));
    return 0;
}

static int run_combine(unsigned int* mb, int blocks, int groups, int by_xcd, const char* name)
{
    const int rounds = 3000;
    double *host_rows = nullptr, *dev_rows = nullptr;
    unsigned int* tickets = nullptr;
    CK(hipHostMalloc((void**)&host_rows, (size_t)groups * 16 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipMalloc((void**)&dev_rows, (size_t)blocks * 16 * sizeof(double)));
    CK(hipMalloc((void**)&tickets, 64 * sizeof(unsigned int)));
    CK(hipMemset(tickets, 0, 64 * sizeof(unsigned int)));
    std::memset(host_rows, 0, (size_t)groups * 16 * sizeof(double));
    std::memset(mb, 0, 64); _mm_sfence();
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipLaunchKernelGGL(answer_combine, dim3(blocks), dim3(64), 0, st, mb, dev_rows, tickets, host_rows, groups, by_xcd, rounds);
    CK(hipGetLastError());
    std::vector<double> rt;
    double sink = 0, want = 0;
    for (int b = 0; b < blocks; ++b) want += (double)(b + 1);
    for (int i = 1; i <= rounds; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        post(mb, (unsigned int)i);
        double mom[16] = {0};
        for (int g = 0; g < groups; ++g) {
            const volatile double* tg = host_rows + (size_t)g * 16;
            while (*tg != (double)i)
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) { std::printf("%s: timeout round %d group %d\n", name, i, g); return 1; }
            const double* row = host_rows + (size_t)g * 16;
            for (int k = 1; k < 16; ++k) mom[k] += row[k];
        }
        const double t1 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (mom[1] != want) { std::printf("%s: wrong sum %g, expected %g (round %d)\n", name, mom[1], want, i); return 1; }
        sink += mom[1];
        if (i > rounds / 3) rt.push_back(1e6 * t1);
    }
    CK(hipStreamSynchronize(st));
    std::sort(rt.begin(), rt.end());
    std::printf("%-46s blocks %4d: all sums on the host after %5.2f us (median; p90 %5.2f)  [%g]\n", name, blocks, rt[rt.size() / 2], rt[rt.size() * 9 / 10], sink > 0 ? 0.0 : 1.0);
    CK(hipStreamDestroy(st)); CK(hipHostFree(host_rows)); CK(hipFree(dev_rows)); CK(hipFree(tickets));
    return 0;
}

int main(int argc, char** argv)
{
    unsigned int* mb = nullptr;
    CK(hipExtMallocWithFlags((void**)&mb, 256, hipDeviceMallocFinegrained));
    if (argc > 1 && std::strcmp(argv[1], "combine") == 0) {
        // the device-side combine against the host's own ingest of every row (the shipped form: compact rows, sweep)
        for (int rep = 0; rep < 2; ++rep)
            for (int blocks : {128, 256}) {
                g_mode = 2;
                run(mb, blocks, 16, 0, 16, 0, "rows of 16 to the host, sweep (shipped form)");
                run_combine(mb, blocks, 8, 1, "combine on the device: 8 groups, b % 8 (per XCD)");
                run_combine(mb, blocks, 8, 0, "combine on the device: 8 groups, contiguous");
                run_combine(mb, blocks, 16, 1, "combine on the device: 16 groups, b % 16");
                run_combine(mb, blocks, 4, 1, "combine on the device: 4 groups, b % 4");
                run_combine(mb, blocks, 1, 0, "combine on the device: 1 group");
            }
        return 0;
    }
    for (g_mode = 1; g_mode <= 2; ++g_mode)
        for (int blocks : {128, 256}) {
            run(mb, blocks, 16, 0, 16, 0, g_mode == 1 ? "rows of 16, next 4 tag lines prefetched" : "rows of 16, sweep over missing tags then add");
            run(mb, blocks, 32, 31, 19, 0, g_mode == 1 ? "rows of 32, next 4 tag lines prefetched" : "rows of 32, sweep over missing tags then add");
        }
    g_mode = 0;
    for (int blocks : {1, 32, 128, 256, 512}) {
        run(mb, blocks, 32, 31, 19, 0, "rows of 32 doubles, 19 used, tag in slot 31");
        run(mb, blocks, 16, 0, 16, 0, "rows of 16 doubles, tag in slot 0");
        run(mb, blocks, 16, 0, 16, 1, "rows of 16 doubles + error array");
        run(mb, blocks, 8, 0, 8, 0, "rows of 8 doubles, tag in slot 0");
    }
    return 0;
}
