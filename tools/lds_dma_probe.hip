// gfx950: where do the 64 x 16 bytes of a global_load_lds_dwordx4 land?  (expected: M0 base + lane * 16, masked lanes skipped)
//   hipcc --offload-arch=gfx950 -O3 -o bin/lds_dma_probe tools/lds_dma_probe.hip && bin/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
__global__ void k(const float* __restrict__ g, const int* __restrict__ src, float* out)
{
    __shared__ __attribute__((aligned(16))) float buf[2][64 * 4];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) (&buf[0][0])[i] = -1.f;
    __syncthreads();
    __builtin_amdgcn_global_load_lds((gptr_t*)(g + (size_t)src[lane] * 4), (lptr_t*)buf[0], 16, 0, 0);
    if (lane < 16) __builtin_amdgcn_global_load_lds((gptr_t*)(g + (size_t)src[lane] * 4 + 4096), (lptr_t*)buf[1], 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = (&buf[0][0])[i];
}
int main()
{
    std::vector<float> hg(8192); for (int i = 0; i < 8192; ++i) hg[i] = (float)i;
    std::vector<int> hs(64); for (int i = 0; i < 64; ++i) hs[i] = (i * 37 + 11) % 1000;
    float *g, *o; int* s;
    hipMalloc(&g, 8192 * 4); hipMalloc(&o, 512 * 4); hipMalloc(&s, 64 * 4);
    hipMemcpy(g, hg.data(), 8192 * 4, hipMemcpyHostToDevice); hipMemcpy(s, hs.data(), 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, s, o);
    std::vector<float> ho(512); hipMemcpy(ho.data(), o, 512 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) if (ho[l * 4 + q] != (float)(hs[l] * 4 + q)) ++bad;
    int bad2 = 0, untouched = 0;
    for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) {
        const float v = ho[256 + l * 4 + q];
        if (l < 16) { if (v != (float)(hs[l] * 4 + q + 4096)) ++bad2; } else if (v == -1.f) ++untouched;
    }
    printf("all lanes: %d mismatches of 256 (lane l's 16 bytes at base + l * 16); 16 active lanes: %d mismatches of 64, %d of the other 192 words untouched\n", bad, bad2, untouched);
    printf("first words: %g %g %g %g | %g %g\n", ho[0], ho[1], ho[4], ho[5], ho[256], ho[260]);
    return bad || bad2 || untouched != 192;
}
