// Latency of the node-local all-reduce (icp_lcomm_*, shared host memory): `world` processes, pinned to CPUs spread over
// the NUMA nodes the way one rank per GPU would be, 32 doubles per exchange.
//   g++ -O2 -o bin/lcomm_bench tools/lcomm_bench.cpp -ldl && bin/lcomm_bench <libicp_mi355x.so> [world 8] [iters 200000]
#include <dlfcn.h>
#include <sched.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef int (*create_fn)(const void*, int, int, void**);
typedef int (*reduce_fn)(void*, double*, int);
typedef void (*destroy_fn)(void*);

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s lib [world] [iters]\n", argv[0]); return 2; }
    const int world = argc > 2 ? std::atoi(argv[2]) : 8, iters = argc > 3 ? std::atoi(argv[3]) : 200000;
    const long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    unsigned char id[128];
    std::memset(id, 0, sizeof id);
    const unsigned seed = (unsigned)getpid() * 2654435761u;
    for (int k = 0; k < 16; ++k) id[k] = (unsigned char)(seed >> (k % 4 * 8)) ^ (unsigned char)(k * 37);
    for (int r = 0; r < world; ++r) {
        if (fork() != 0) continue;
        // spread: rank r on CPU r * (ncpu / 2) / world * ... -- first half of the ranks on the first half of the physical
        // CPUs (socket 0 on the two-socket GPU hosts: CPUs 0-63), the second half on the second (64-127)
        const long phys = ncpu >= 4 ? ncpu / 2 : ncpu;   // (SMT siblings are the upper half of the numbering)
        cpu_set_t set; CPU_ZERO(&set); CPU_SET((int)((long)r * phys / world), &set);
        sched_setaffinity(0, sizeof set, &set);
        void* h = dlopen(argv[1], RTLD_NOW);
        if (!h) { std::fprintf(stderr, "%s\n", dlerror()); _exit(3); }
        create_fn cr = (create_fn)dlsym(h, "icp_lcomm_create");
        reduce_fn rd = (reduce_fn)dlsym(h, "icp_lcomm_allreduce");
        destroy_fn ds = (destroy_fn)dlsym(h, "icp_lcomm_destroy");
        void* lc = nullptr;
        if (!cr || !rd || !ds || cr(id, r, world, &lc) != 0) { std::fprintf(stderr, "rank %d: create failed\n", r); _exit(4); }
        double v[32];
        for (int w = 0; w < 2; ++w) {
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < iters; ++i) {
                for (int k = 0; k < 32; ++k) v[k] = r + k;
                if (rd(lc, v, 32) != 0) _exit(5);
            }
            const double us = 1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / iters;
            if (w == 1 && r == 0) { std::printf("%d ranks: %.3f us per all-reduce of 32 doubles (sum[0] = %.0f)\n", world, us, v[0]); std::fflush(stdout); }
        }
        ds(lc);
        _exit(0);
    }
    int st = 0, bad = 0;
    while (wait(&st) > 0) bad |= st;
    return bad ? 1 : 0;
}
