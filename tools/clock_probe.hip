// clock_probe.hip -- at what shader clock does a kernel run that mostly WAITS?
// The resident registration kernel spends most of its life in s_sleep polling loops with short bursts of work in between.
// This probe imitates that: `blocks` blocks of `waves` waves; per round every wave idles ~idle_us (s_sleep poll loop on the
// constant 100 MHz counter, or -- busy = 1 -- a spin of dependent VALU ops), then runs a burst of 2000 dependent v_add_f32
// and stamps both counters around it: s_memtime (shader clock) and s_memrealtime (100 MHz).  Prints the shader clock
// seen over the burst and the burst's duration.
// build: hipcc --offload-arch=gfx950 -O2 -o bin/clock_probe tools/clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void probe(long long* out, int rounds, int idle_ticks, int busy)
{
    float v = (float)threadIdx.x;
    for (int r = 0; r < rounds; ++r) {
        const long long t_end = (long long)wall_clock64() + idle_ticks;
        if (busy) {
            while ((long long)wall_clock64() < t_end) {
#pragma unroll
                for (int k = 0; k < 64; ++k) v = v + 1.0f;
            }
        } else {
            while ((long long)wall_clock64() < t_end) __builtin_amdgcn_s_sleep(2);
        }
        const long long c0 = clock64(), w0 = wall_clock64();
#pragma unroll 100
        for (int k = 0; k < 2000; ++k) asm volatile("v_add_f32 %0, %0, 1.0" : "+v"(v));
        const long long c1 = clock64(), w1 = wall_clock64();
        if (threadIdx.x == 0 && blockIdx.x == 0) { out[2 * r] = c1 - c0; out[2 * r + 1] = w1 - w0; }
    }
    if (v == 12345.f) out[0] = 0;
}

int main(int argc, char** argv)
{
    const int rounds = 400;
    long long* d = nullptr;
    CK(hipMalloc((void**)&d, 2 * rounds * sizeof(long long)));
    std::vector<long long> h(2 * rounds);
    for (int busy = 0; busy < 2; ++busy)
        for (int blocks : {1, 128, 256})
            for (int waves : {1, 8, 16})
                for (int idle_us : {1, 8}) {
                    hipLaunchKernelGGL(probe, dim3(blocks), dim3(64 * waves), 0, 0, d, rounds, idle_us * 100, busy);
                    CK(hipDeviceSynchronize());
                    CK(hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
                    std::vector<double> mhz, us;
                    for (int r = rounds / 2; r < rounds; ++r) {
                        us.push_back(h[2 * r + 1] / 100.0);
                        mhz.push_back(h[2 * r + 1] > 0 ? 100.0 * (double)h[2 * r] / (double)h[2 * r + 1] : 0.0);
                    }
                    std::sort(mhz.begin(), mhz.end());
                    std::sort(us.begin(), us.end());
                    std::printf("%s wait, %3d blocks x %2d waves, idle %d us: burst of 2000 dependent v_add_f32 takes %6.2f us (median), s_memtime runs at %7.1f MHz\n",
                                busy ? "busy " : "sleep", blocks, waves, idle_us, us[us.size() / 2], mhz[mhz.size() / 2]);
                }
    return 0;
}
