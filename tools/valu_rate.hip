// tools/valu_rate.hip -- gfx950 VALU issue-rate probe used to size the matching kernel.
// For each instruction kind: 16 independent accumulators, `iters` x 16 instructions per wave, W blocks
// (of 4 waves) per CU.  Reports shader cycles (s_memtime) per wave-instruction as seen by one wave and the
// aggregate lane-ops/s.  Build: hipcc --offload-arch=gfx950 -O2 tools/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* cyc, int iters)
{
    float a[16];
    f2 p[16];
    double d[16];
    for (int i = 0; i < 16; ++i) {
        a[i] = threadIdx.x * 0.001f + i;
        p[i] = f2{a[i], a[i] + 1.f};
        d[i] = a[i];
    }
    const float c = 1.0001f + blockIdx.x * 1e-9f;
    const f2 c2 = f2{c, c};
    const double cd = c;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            REP16(X)
#undef X
        } else if constexpr (MODE == 1) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
            REP16(X)
#undef X
        } else if constexpr (MODE == 2) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            REP16(X)
#undef X
        } else if constexpr (MODE == 3) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
            REP16(X)
#undef X
        } else if constexpr (MODE == 4) {
#define X(i) asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
            REP16(X)
#undef X
        } else if constexpr (MODE == 5) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
            REP16(X)
#undef X
        } else if constexpr (MODE == 6) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
            REP16(X)
#undef X
        } else if constexpr (MODE == 7) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
            REP16(X)
#undef X
        } else if constexpr (MODE == 8) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(c2));
            REP16(X)
#undef X
        } else if constexpr (MODE == 9) {
#define X(i) asm volatile("v_min_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
            REP16(X)
#undef X
        } else if constexpr (MODE == 10) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c));
            REP16(X)
#undef X
        } else if constexpr (MODE == 11) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(c) : "vcc");
            REP16(X)
#undef X
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y + (float)d[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int lanes_per_instr, int cus)
{
    const int iters = 40000;
    for (int bpc : {1, 2, 4, 8}) {
        const int blocks = cus * bpc;
        float* out;
        unsigned long long* cyc;
        hipMalloc(&out, blocks * 256 * sizeof(float));
        hipMalloc(&cyc, blocks * sizeof(unsigned long long));
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        probe<MODE><<<blocks, 256>>>(out, cyc, 2000);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        probe<MODE><<<blocks, 256>>>(out, cyc, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double avg = 0;
        for (auto v : h) avg += (double)v;
        avg /= blocks;
        const double instr = (double)iters * 16;
        const double wave_instr_total = instr * blocks * 4;
        std::printf("%-14s waves/SIMD=%d  cyc/instr(per wave)=%6.2f  => SIMD issue interval=%5.2f cyc  |  %7.2f T lane-ops/s  (%.3f ms)\n",
                    name, bpc, avg / instr, avg / instr / bpc, wave_instr_total * lanes_per_instr / (ms * 1e-3) / 1e12, ms);
        hipFree(out);
        hipFree(cyc);
    }
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    std::printf("%s  CUs=%d  clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    const int cus = p.multiProcessorCount;
    run<0>("v_add_f32", 64, cus);
    run<1>("v_pk_add_f32", 128, cus);
    run<2>("v_mul_f32", 64, cus);
    run<3>("v_pk_mul_f32", 128, cus);
    run<4>("v_min3_f32", 64, cus);
    run<7>("v_fma_f32", 64, cus);
    run<8>("v_pk_fma_f32", 128, cus);
    run<5>("v_add_f64", 64, cus);
    run<6>("v_mul_f64", 64, cus);
    run<9>("v_min_f64", 64, cus);
    run<10>("v_cndmask_b32", 64, cus);
    run<11>("v_cmp_lt_f32", 64, cus);
    return 0;
}
