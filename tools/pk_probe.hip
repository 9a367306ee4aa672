// pk_probe: issue rate of the packed fp32 forms the matching kernels are made of (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o bin/pk_probe tools/pk_probe.hip && bin/pk_probe
// Every variant: 8 independent accumulators per lane, 64 instructions per loop turn, 4 waves per block, `bpc` blocks per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

template <int HI, bool SG>
__device__ __forceinline__ f2 sub_b(f2 q, f2 p)
{
    f2 r;
    if constexpr (SG) {
        if constexpr (HI == 0) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "s"(q), "v"(p));
        else asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "s"(q), "v"(p));
    } else {
        if constexpr (HI == 0) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(q), "v"(p));
        else asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(q), "v"(p));
    }
    return r;
}
template <int HI, bool SG>
__device__ __forceinline__ f2 dist2p(f2 qx, f2 qy, f2 qz, f2 px, f2 py, f2 pz)
{
    f2 dx = sub_b<HI, SG>(qx, px), dy = sub_b<HI, SG>(qy, py), dz = sub_b<HI, SG>(qz, pz);
    dx = dx * dx; dy = dy * dy; dz = dz * dz;
    f2 d = dx + dy;
    return d + dz;
}
// the matching kernels' own chunk: 8 model points (registers) against the lane's two points -- 64 packed ops + 8 min3
template <bool SG>
__global__ __launch_bounds__(256) void probe_chunk(float* out, int iters, float seed, unsigned long long sq)
{
    f2 px = f2{seed + threadIdx.x, seed * 0.5f}, py = f2{seed * 0.25f + threadIdx.x, seed * 0.125f}, pz = f2{seed * 3.f, seed * 5.f};
    f2 q[12];
    unsigned lo = (unsigned)sq, hi = (unsigned)(sq >> 32);
    for (int k = 0; k < 12; ++k) {
        if constexpr (SG) q[k] = f2{__uint_as_float(__builtin_amdgcn_readfirstlane(lo + k)), __uint_as_float(__builtin_amdgcn_readfirstlane(hi + 3 * k))};
        else q[k] = f2{__uint_as_float(lo + k) + threadIdx.x, __uint_as_float(hi + 3 * k)};
    }
    float b0 = 1e30f, b1 = 1e30f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f2 d0 = dist2p<0, SG>(q[k], q[4 + k], q[8 + k], px, py, pz);
            const f2 d1 = dist2p<1, SG>(q[k], q[4 + k], q[8 + k], px, py, pz);
            b0 = __builtin_fminf(__builtin_fminf(b0, d0.x), d1.x);
            b1 = __builtin_fminf(__builtin_fminf(b1, d0.y), d1.y);
        }
        asm volatile("" : "+v"(px), "+v"(py), "+v"(pz));   // (a new turn: nothing is hoisted out of the loop)
    }
    if (b0 + b1 == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = b0;
}

// a SHORT kernel of the same arithmetic (what a hall-sized dense launch is: 32 chunks per wave, 8 waves per SIMD): the shader
// clock (s_memtime) against the constant 100 MHz counter tells at what frequency it actually ran
__global__ __launch_bounds__(256) void probe_short(long long* clk, int iters, float seed, unsigned long long sq)
{
    const long long c0 = clock64(), w0 = wall_clock64();
    f2 px = f2{seed + threadIdx.x, seed * 0.5f}, py = f2{seed * 0.25f + threadIdx.x, seed * 0.125f}, pz = f2{seed * 3.f, seed * 5.f};
    f2 q[12];
    unsigned lo = (unsigned)sq, hi = (unsigned)(sq >> 32);
    for (int k = 0; k < 12; ++k) q[k] = f2{__uint_as_float(lo + k) + threadIdx.x, __uint_as_float(hi + 3 * k)};
    float b0 = 1e30f, b1 = 1e30f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f2 d0 = dist2p<0, false>(q[k], q[4 + k], q[8 + k], px, py, pz);
            const f2 d1 = dist2p<1, false>(q[k], q[4 + k], q[8 + k], px, py, pz);
            b0 = __builtin_fminf(__builtin_fminf(b0, d0.x), d1.x);
            b1 = __builtin_fminf(__builtin_fminf(b1, d0.y), d1.y);
        }
        asm volatile("" : "+v"(px), "+v"(py), "+v"(pz));
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
        clk[2 * wv] = c1 - c0;
        clk[2 * wv + 1] = (b0 + b1 == 12345.678f) ? 0 : w1 - w0;
    }
}

static void run_short(int iters, bool back_to_back)
{
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = cus * 8, waves = blocks * 4;
    long long* d = nullptr;
    (void)hipMalloc(&d, (size_t)waves * 2 * sizeof(long long));
    long long* h = (long long*)std::malloc((size_t)waves * 2 * sizeof(long long));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) { probe_short<<<blocks, 256>>>(d, iters, 1.5f, 0x3fc000003f800000ull); (void)hipDeviceSynchronize(); }
    float ms = 0.f;
    const int reps = back_to_back ? 20 : 1;
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) probe_short<<<blocks, 256>>>(d, iters, 1.5f, 0x3fc000003f800000ull);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, d, (size_t)waves * 2 * sizeof(long long), hipMemcpyDeviceToHost);
    double mhz = 0, us_max = 0, us_min = 1e30, us_sum = 0;
    for (int w = 0; w < waves; ++w) {
        const double us = h[2 * w + 1] / 100.0;
        mhz += h[2 * w] / us;
        us_max = us > us_max ? us : us_max; us_min = us < us_min ? us : us_min; us_sum += us;
    }
    std::printf("short kernel, %3d chunks per wave, 8 waves/SIMD, %s: %7.2f us per launch (events); a wave's loop lasts %6.2f / %6.2f / %6.2f us (min / mean / max), "
                "shader clock while it runs %6.0f MHz; %5.2f T pairs/s by the events\n", iters, back_to_back ? "20 launches back to back" : "one launch after a synchronisation",
                1e3 * ms / reps, us_min, us_sum / waves, us_max, mhz / waves, (double)waves * iters * 1024 / (1e-3 * ms / reps) / 1e12);
    (void)hipFree(d);
    std::free(h);
}

template <bool SG>
static void run_chunk(const char* name, int bpc, float* out)
{
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int iters = 4096 * 8;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe_chunk<SG><<<cus * bpc, 256>>>(out, 64, 1.5f, 0x3fc000003f800000ull);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    probe_chunk<SG><<<cus * bpc, 256>>>(out, iters, 1.5f, 0x3fc000003f800000ull);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double chunks_per_simd = (double)bpc * iters;     // (4 waves of a block: one per SIMD)
    std::printf("%-58s waves/SIMD %d: %7.3f ms  %6.1f cycles of a SIMD per chunk (64 packed ops + 8 min3; 2.4 GHz)  %5.2f T pairs/s\n", name, bpc, ms,
                ms * 1e-3 * 2.4e9 / chunks_per_simd, (double)cus * bpc * 4 * iters * 1024 / (ms * 1e-3) / 1e12);
}

template <int V>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed, unsigned long long sq)
{
    f2 a[8];
    float s[8];
    for (int k = 0; k < 8; ++k) { a[k] = f2{seed + k + threadIdx.x, seed * 0.5f + k}; s[k] = seed + k; }
    f2 p = f2{seed * 0.25f, seed * 0.125f};
    f2 qs;   // wave-uniform pair in SGPRs
    {
        unsigned lo = (unsigned)sq, hi = (unsigned)(sq >> 32);
        qs = f2{__uint_as_float(__builtin_amdgcn_readfirstlane(lo)), __uint_as_float(__builtin_amdgcn_readfirstlane(hi))};
    }
    for (int it = 0; it < iters; ++it) {
        if constexpr (V == 0) {         // plain packed add
#define X(k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(p));
            REP64(X)
#undef X
        } else if constexpr (V == 1) {  // packed add, src0 half broadcast by op_sel, src1 negated (the kernels' subtraction)
#define X(k) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a[k]) : "v"(p));
            REP64(X)
#undef X
        } else if constexpr (V == 2) {  // packed mul (square)
#define X(k) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(a[k]));
            REP64(X)
#undef X
        } else if constexpr (V == 3) {  // packed add, src0 an SGPR pair with op_sel broadcast
#define X(k) asm volatile("v_pk_add_f32 %0, %1, %0 op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a[k]) : "s"(qs));
            REP64(X)
#undef X
        } else if constexpr (V == 4) {  // min3
#define X(k) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(s[k]) : "v"(a[k].x), "v"(a[k].y));
            REP64(X)
#undef X
        } else if constexpr (V == 5) {  // plain scalar add
#define X(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[k]) : "v"(p.x));
            REP64(X)
#undef X
        } else if constexpr (V == 6) {  // packed add with negation only (no broadcast)
#define X(k) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a[k]) : "v"(p));
            REP64(X)
#undef X
        } else if constexpr (V == 7) {  // packed add, SGPR pair src0, no modifiers
#define X(k) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[k]) : "s"(qs));
            REP64(X)
#undef X
        } else if constexpr (V == 8) {  // plain scalar mul
#define X(k) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(s[k]));
            REP64(X)
#undef X
        } else if constexpr (V == 9) {  // packed fma (the FMA-counted roof's instruction)
#define X(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[k]) : "v"(p));
            REP64(X)
#undef X
        } else if constexpr (V == 10) { // packed add broadcasting the HIGH half
#define X(k) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a[k]) : "v"(p));
            REP64(X)
#undef X
        } else if constexpr (V == 11) { // scalar sub with an SGPR operand (one model coordinate against one moving point)
#define X(k) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(s[k]) : "s"(qs.x));
            REP64(X)
#undef X
        }
    }
    float r = 0.f;
    for (int k = 0; k < 8; ++k) r += a[k].x + a[k].y + s[k];
    if (r == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int V>
static void run(const char* name, int bpc, float* out)
{
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int iters = 4096;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<V><<<cus * bpc, 256>>>(out, 64, 1.5f, 0x3fc000003f800000ull);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    probe<V><<<cus * bpc, 256>>>(out, iters, 1.5f, 0x3fc000003f800000ull);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)cus * bpc * 4 * iters * 64;                 // wave-instructions
    const double per_simd_cycles = ms * 1e-3 * 2.4e9 / (winstr / (cus * 4.0));  // SIMD cycles per wave-instruction at 2.4 GHz
    std::printf("%-58s waves/SIMD %d: %7.3f ms  %6.2f T lane-instr/s  %5.2f cycles per wave-instruction (2.4 GHz)\n", name, bpc, ms,
                winstr * 64 / (ms * 1e-3) / 1e12, per_simd_cycles);
}

int main()
{
    float* out = nullptr;
    (void)hipMalloc(&out, 1 << 24);
    for (int iters : {32, 128, 512, 4096}) { run_short(iters, false); run_short(iters, true); }
    for (int bpc : {8, 4, 2, 1}) {
        run_chunk<false>("chunk of the matching kernels, model operand in VGPRs", bpc, out);
        run_chunk<true>("chunk of the matching kernels, model operand in SGPRs", bpc, out);
    }
    for (int bpc : {8}) {
        run<0>("v_pk_add_f32 plain", bpc, out);
        run<6>("v_pk_add_f32 neg", bpc, out);
        run<1>("v_pk_add_f32 op_sel lo-broadcast + neg (kernel form)", bpc, out);
        run<10>("v_pk_add_f32 op_sel hi-broadcast + neg (kernel form)", bpc, out);
        run<3>("v_pk_add_f32 SGPR-pair src0, lo-broadcast + neg", bpc, out);
        run<7>("v_pk_add_f32 SGPR-pair src0, plain", bpc, out);
        run<2>("v_pk_mul_f32 (square)", bpc, out);
        run<9>("v_pk_fma_f32", bpc, out);
        run<4>("v_min3_f32", bpc, out);
        run<5>("v_add_f32", bpc, out);
        run<8>("v_mul_f32", bpc, out);
        run<11>("v_sub_f32 SGPR src0", bpc, out);
    }
    return 0;
}
