// icp_k_setup.hip -- once per icp_set_model / icp_set_moving: exact-duplicate flags, Hilbert / Morton order and the extent test, chunk
// boxes, samples, per-chunk records; and between two passes of a large cloud: the order and the roles of the next launch's rows.
#include "icp_device.h"
#include <math.h>
#include <stdlib.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace icp {

// ------------------------------------------------------------------------------------------------
// preparation of a cloud for the sparse kernel, on the device (once per icp_set_model / icp_set_moving):
// exact-duplicate flags (lexicographic order of the raw coordinate bits: three stable radix passes),
// Morton order, and the grouped-extent test that decides whether the Morton-ordered view is used.
// Everything is deterministic (fixed-order reductions): the decision must not change from run to run.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned int canon_bits(float v) { return __float_as_uint(v == 0.0f ? 0.0f : v); }
__device__ __forceinline__ bool finite3(float x, float y, float z)
{
    return x - x == 0.f && y - y == 0.f && z - z == 0.f;   // false for NaN and +-inf
}

// keys[s] = raw bits of coordinate `axis` of point order[s] (order == NULL: identity, and vals is initialised)
__global__ void prep_axis_keys_kernel(const float* __restrict__ X, int n, int n_pad, int axis, const int32_t* __restrict__ order,
                                      unsigned int* __restrict__ keys, int32_t* __restrict__ vals)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int i = order ? order[s] : s;
    keys[s] = canon_bits(X[(size_t)axis * n_pad + i]);
    if (!order) vals[s] = s;
}

// lex[s] ascending in (x, y, z, index): a point equal to its predecessor has a lower-index twin
__global__ void prep_mark_duplicates_kernel(const float* __restrict__ X, int n, int n_pad, const int32_t* __restrict__ lex,
                                            unsigned char* __restrict__ voided, int* __restrict__ count)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    bool v = false;
    if (s > 0) {
        const int a = lex[s - 1], b = lex[s];
        const float bx = X[b], by = X[(size_t)n_pad + b], bz = X[2 * (size_t)n_pad + b];
        const bool same = canon_bits(X[a]) == canon_bits(bx) && canon_bits(X[(size_t)n_pad + a]) == canon_bits(by) &&
                          canon_bits(X[2 * (size_t)n_pad + a]) == canon_bits(bz);
        const bool nan = bx != bx || by != by || bz != bz;
        v = same && !nan;
    }
    voided[lex[s]] = v ? 1 : 0;
    if (v) atomicAdd(count, 1);
}

// scan copy: the cloud with its flagged points (and the padding) voided to +inf
__global__ void prep_scan_copy_kernel(const float* __restrict__ X, int n, int n_pad, const unsigned char* __restrict__ voided,
                                      float* __restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_pad) return;
    const bool keep = j < n && !voided[j];
#pragma unroll
    for (int a = 0; a < 3; ++a) out[(size_t)a * n_pad + j] = keep ? X[(size_t)a * n_pad + j] : inf_<float>();
}

// ---- the same for a cloud in double: a 64-bit key is two stable 32-bit passes (low word, then high word) --------------
__device__ __forceinline__ unsigned long long canon_bits64(double v) { return (unsigned long long)__double_as_longlong(v == 0.0 ? 0.0 : v); }

__global__ void prep_axis_keys_f64_kernel(const double* __restrict__ X, int n, int n_pad, int axis, int high, const int32_t* __restrict__ order,
                                          unsigned int* __restrict__ keys, int32_t* __restrict__ vals)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int i = order ? order[s] : s;
    const unsigned long long b = canon_bits64(X[(size_t)axis * n_pad + i]);
    keys[s] = high ? (unsigned int)(b >> 32) : (unsigned int)b;
    if (!order) vals[s] = s;
}

__global__ void prep_mark_duplicates_f64_kernel(const double* __restrict__ X, int n, int n_pad, const int32_t* __restrict__ lex,
                                                unsigned char* __restrict__ voided, int* __restrict__ count)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    bool v = false;
    if (s > 0) {
        const int a = lex[s - 1], b = lex[s];
        const double bx = X[b], by = X[(size_t)n_pad + b], bz = X[2 * (size_t)n_pad + b];
        const bool same = canon_bits64(X[a]) == canon_bits64(bx) && canon_bits64(X[(size_t)n_pad + a]) == canon_bits64(by) &&
                          canon_bits64(X[2 * (size_t)n_pad + a]) == canon_bits64(bz);
        const bool nan = bx != bx || by != by || bz != bz;
        v = same && !nan;
    }
    voided[lex[s]] = v ? 1 : 0;
    if (v) atomicAdd(count, 1);
}

__global__ void prep_scan_copy_f64_kernel(const double* __restrict__ X, int n, int n_pad, const unsigned char* __restrict__ voided,
                                          double* __restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_pad) return;
    const bool keep = j < n && !voided[j];
#pragma unroll
    for (int a = 0; a < 3; ++a) out[(size_t)a * n_pad + j] = keep ? X[(size_t)a * n_pad + j] : inf_<double>();
}

// bounding cube of the finite points: box[0..2] = lo, box[3] = largest extent.  One block, or (partial != NULL) a grid of them
// that leave {lo, hi} per block for prep_bbox_final_kernel -- minima and maxima: the result does not depend on the split
// (a 10 M-point cloud through one block was 3.8 ms of a 37 ms set-up, twice)
__global__ __launch_bounds__(1024) void prep_bbox_kernel(const float* __restrict__ X, int n, int n_pad, float* __restrict__ box, float* __restrict__ partial)
{
    __shared__ float red[6][1024];
    float lo[3] = {inf_<float>(), inf_<float>(), inf_<float>()}, hi[3] = {-inf_<float>(), -inf_<float>(), -inf_<float>()};
    for (int i = blockIdx.x * 1024 + threadIdx.x; i < n; i += 1024 * gridDim.x) {
        const float x = X[i], y = X[(size_t)n_pad + i], z = X[2 * (size_t)n_pad + i];
        if (!finite3(x, y, z)) continue;
        lo[0] = fminf(lo[0], x); lo[1] = fminf(lo[1], y); lo[2] = fminf(lo[2], z);
        hi[0] = fmaxf(hi[0], x); hi[1] = fmaxf(hi[1], y); hi[2] = fmaxf(hi[2], z);
    }
    for (int a = 0; a < 3; ++a) { red[a][threadIdx.x] = lo[a]; red[3 + a][threadIdx.x] = hi[a]; }
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w)
            for (int a = 0; a < 3; ++a) {
                red[a][threadIdx.x] = fminf(red[a][threadIdx.x], red[a][threadIdx.x + w]);
                red[3 + a][threadIdx.x] = fmaxf(red[3 + a][threadIdx.x], red[3 + a][threadIdx.x + w]);
            }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (partial != nullptr) {
            for (int a = 0; a < 6; ++a) partial[blockIdx.x * 6 + a] = red[a][0];
            return;
        }
        float ext = 0.f;
        for (int a = 0; a < 3; ++a) { box[a] = red[a][0]; ext = fmaxf(ext, red[3 + a][0] - red[a][0]); }
        box[3] = ext;   // -inf / NaN when there is no finite point: the codes below then all take the "last" value
    }
}

__global__ __launch_bounds__(64) void prep_bbox_final_kernel(const float* __restrict__ partial, int blocks, float* __restrict__ box)
{
    float v[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) v[a] = a < 3 ? inf_<float>() : -inf_<float>();
    for (int b = threadIdx.x; b < blocks; b += 64)
#pragma unroll
        for (int a = 0; a < 6; ++a) v[a] = a < 3 ? fminf(v[a], partial[b * 6 + a]) : fmaxf(v[a], partial[b * 6 + a]);
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(v[a], off, 64); v[a] = a < 3 ? fminf(v[a], o) : fmaxf(v[a], o); }
    if (threadIdx.x == 0) {
        float ext = 0.f;
        for (int a = 0; a < 3; ++a) { box[a] = v[a]; ext = fmaxf(ext, v[3 + a] - v[a]); }
        box[3] = ext;
    }
}

__device__ __forceinline__ unsigned int spread10(unsigned int v)
{
    v &= 1023u;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// position of cell (x, y, z) of a 1024^3 grid along the Hilbert curve (J. Skilling, "Programming the Hilbert curve", AIP Conf.
// Proc. 707, 2004: axes -> transposed index, then the bits interleaved).  Consecutive positions are neighbouring cells -- a
// Z-order range of 128 points straddles the curve's jumps, and the group box is the union: on the 10 M-point surface the rows
// of 128 measure 0.027 x 0.027 x 0.031 in this order against 0.038 x 0.034 x 0.031 in Z-order, and every level of the search
// lists 10-19 % fewer boxes (tools/s5_hits_model.py)
__device__ __forceinline__ unsigned int hilbert30(unsigned int x, unsigned int y, unsigned int z)
{
    unsigned int X[3] = {x & 1023u, y & 1023u, z & 1023u};
#pragma unroll
    for (unsigned int Q = 512u; Q > 1u; Q >>= 1) {
        const unsigned int P = Q - 1u;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) X[0] ^= P;
            else { const unsigned int t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    X[1] ^= X[0];
    X[2] ^= X[1];
    unsigned int t = 0u;
#pragma unroll
    for (unsigned int Q = 512u; Q > 1u; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1u;
    return (spread10(X[0] ^ t) << 2) | (spread10(X[1] ^ t) << 1) | spread10(X[2] ^ t);
}

// 30-bit space-filling-curve codes in one cube for all axes (cells stay cubic): the Hilbert curve, or Z-order (hilbert == 0:
// ICP_ORDER=morton, A/B runs); non-finite points go last
__global__ void prep_morton_keys_kernel(const float* __restrict__ X, int n, int n_pad, const float* __restrict__ box,
                                        unsigned int* __restrict__ keys, int32_t* __restrict__ vals, int hilbert)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = X[i], y = X[(size_t)n_pad + i], z = X[2 * (size_t)n_pad + i];
    unsigned int code = 0x7fffffffu;
    const float ext = box[3];
    if (finite3(x, y, z) && ext >= 0.f) {
        const double scale = ext > 0.f ? 1023.0 / (double)ext : 0.0;
        const unsigned int qx = (unsigned int)fmin(1023.0, fmax(0.0, ((double)x - (double)box[0]) * scale));
        const unsigned int qy = (unsigned int)fmin(1023.0, fmax(0.0, ((double)y - (double)box[1]) * scale));
        const unsigned int qz = (unsigned int)fmin(1023.0, fmax(0.0, ((double)z - (double)box[2]) * scale));
        code = hilbert ? hilbert30(qx, qy, qz) : (spread10(qx) | (spread10(qy) << 1) | (spread10(qz) << 2));
    }
    keys[i] = code;
    vals[i] = i;
}

// per group of `group` consecutive entries of an order (NULL: the cloud's own): extent dx + dy + dz of its finite points.
// One wave per group (a thread per group walks 128 gathered points one after the other: 50 us for 128 groups).
__global__ __launch_bounds__(64) void prep_group_extent_kernel(const float* __restrict__ X, int n, int n_pad, const int32_t* __restrict__ order,
                                                               int group, double* __restrict__ ext)
{
    const int g = blockIdx.x, g0 = g * group, lane = threadIdx.x;
    float lo[3] = {inf_<float>(), inf_<float>(), inf_<float>()}, hi[3] = {-inf_<float>(), -inf_<float>(), -inf_<float>()};
    for (int k = g0 + lane; k < min(n, g0 + group); k += 64) {
        const int i = order ? order[k] : k;
        const float p[3] = {X[i], X[(size_t)n_pad + i], X[2 * (size_t)n_pad + i]};
        if (!finite3(p[0], p[1], p[2])) continue;
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], p[a]); hi[a] = fmaxf(hi[a], p[a]); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off, 64));
        }
    if (lane == 0) ext[g] = hi[0] >= lo[0] ? (double)(hi[0] - lo[0]) + (double)(hi[1] - lo[1]) + (double)(hi[2] - lo[2]) : 0.0;
}

// out[which] = sum of ext[0..groups) in a fixed order (one block)
__global__ __launch_bounds__(256) void prep_sum_kernel(const double* __restrict__ ext, int groups, double* __restrict__ out, int which)
{
    __shared__ double red[256];
    double s = 0.0;
    for (int g = threadIdx.x; g < groups; g += 256) s += ext[g];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[which] = red[0];
}

// ------------------------------------------------------------------------------------------------
// Round 4: the set-up of a cloud in a handful of launches (what a sensor's NEXT pair costs: the hall pair's set-up was ~60 launches
// of 3-10 us -- five rocPRIM radix sorts in 25 of them -- 0.35 ms against a registration of 0.12 ms).
//   * exact duplicates by HASHING instead of three stable radix sorts: every point looks up its coordinates' slot in an
//     open-addressing table (one 32-bit word per slot: generation << 24 | index of the lowest-index point seen so far with these
//     coordinates -- any member of a group serves to compare coordinates with, they are all the same); kernel 1 inserts (CAS on an
//     empty or stale slot, atomicMin on a slot of its own coordinates), kernel 2 looks up: a point is voided iff the minimum of its
//     slot is not itself -- and writes the scan copy in the same pass.  The answer is the unique minimum: deterministic.  The
//     generation makes clearing the table unnecessary (an entry of another generation is empty; a stale entry that survives a
//     wrap-around of the 8-bit counter names a point whose coordinates are compared like any other's: harmless).
//   * the bounding cube comes out of the layout kernel (ordered-integer atomics), the group extents are summed in 2^-36 fixed point
//     by 64-bit integer atomics (any order of additions gives the same bits: the decision must not change from run to run),
//     given and sorted order in one launch.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float ord_decode(unsigned int e) { return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e); }
// box[0..2] = lo, box[3] = the largest extent (-inf when the cloud has no finite point), from the layout kernel's six encodings
// {~ord(min x), ~ord(min y), ~ord(min z), ord(max x), ord(max y), ord(max z)} (0 = nothing seen)
__device__ __forceinline__ void box_from_enc(const unsigned int* __restrict__ enc, float (&box)[4])
{
    float ext = -inf_<float>();
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const unsigned int el = enc[a], eh = enc[3 + a];
        const float lo = el ? ord_decode(~el) : inf_<float>(), hi = eh ? ord_decode(eh) : -inf_<float>();
        box[a] = lo;
        ext = fmaxf(ext, hi - lo);
    }
    box[3] = (enc[0] && enc[3]) ? ext : -inf_<float>();
}

__device__ __forceinline__ unsigned int hash3(unsigned int a, unsigned int b, unsigned int c)
{
    unsigned int h = a * 0x9E3779B1u ^ b * 0x85EBCA77u ^ c * 0xC2B2AE3Du;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}

__global__ void prep_dup_insert_kernel(const float* __restrict__ X, int n, int n_pad, unsigned int* __restrict__ table, unsigned int mask, unsigned int gen)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < n;
    const float x = in ? X[i] : 0.f, y = in ? X[(size_t)n_pad + i] : 0.f, z = in ? X[2 * (size_t)n_pad + i] : 0.f;
    const bool live = in && !(x != x || y != y || z != z);   // (NaN equals nothing; such a cloud is refused anyway)
    const unsigned int bx = canon_bits(x), by = canon_bits(y), bz = canon_bits(z), me = (gen << 24) | (unsigned int)i;
    // One insert per wave and coordinate triple: the hall scan's 4 361 no-return points all hash to ONE slot, and thousands of atomics on
    // one address were 60 us of a 107 us set-up.  The lanes of a wave are consecutive indices, so the lowest lane of a group of equal
    // points holds the group's minimum within the wave: it alone goes to the table (a loop over the wave's distinct points).
    bool leader = false;
    {
        const int lane = (int)(threadIdx.x & 63);
        unsigned long long todo = __builtin_amdgcn_ballot_w64(live);
        while (todo != 0ull) {
            const int first = (int)__builtin_ctzll(todo);
            const unsigned int fx = (unsigned int)__builtin_amdgcn_readlane((int)bx, first), fy = (unsigned int)__builtin_amdgcn_readlane((int)by, first),
                               fz = (unsigned int)__builtin_amdgcn_readlane((int)bz, first);
            const unsigned long long same = __builtin_amdgcn_ballot_w64(live && bx == fx && by == fy && bz == fz) & todo;
            if (lane == first) leader = true;
            todo &= ~same;
        }
    }
    if (!leader) return;
    unsigned int h = hash3(bx, by, bz) & mask;
    for (unsigned int probes = 0; probes <= mask; ++probes) {
        unsigned int v = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (;;) {
            if ((v >> 24) != gen) {   // empty (or another upload's): claim it
                unsigned int expect = v;
                if (__hip_atomic_compare_exchange_strong(&table[h], &expect, me, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
                v = expect;           // somebody else got there first: look at what is there now
                continue;
            }
            break;
        }
        const int j = (int)(v & 0xffffffu);
        if (canon_bits(X[j]) == bx && canon_bits(X[(size_t)n_pad + j]) == by && canon_bits(X[2 * (size_t)n_pad + j]) == bz) {
            if (i < j) __hip_atomic_fetch_min(&table[h], me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (same generation: the index decides)
            return;
        }
        h = (h + 1u) & mask;
    }
}

// voided[j] = 1 for every point with an exact lower-index twin, *count += their number; and the scan copy (voided points and
// the padding at +inf) in the same pass
__global__ void prep_dup_lookup_kernel(const float* __restrict__ X, int n, int n_pad, const unsigned int* __restrict__ table, unsigned int mask, unsigned int gen,
                                       unsigned char* __restrict__ voided, int* __restrict__ count, float* __restrict__ scan_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    bool v = false;
    float x = inf_<float>(), y = x, z = x;
    if (i < n) {
        x = X[i]; y = X[(size_t)n_pad + i]; z = X[2 * (size_t)n_pad + i];
        if (x == x && y == y && z == z) {
            const unsigned int bx = canon_bits(x), by = canon_bits(y), bz = canon_bits(z);
            unsigned int h = hash3(bx, by, bz) & mask;
            for (unsigned int probes = 0; probes <= mask; ++probes) {
                const unsigned int e = table[h];   // (written by the kernel before: the kernel boundary orders it)
                if ((e >> 24) != gen) break;       // (cannot happen: every finite point was inserted)
                const int j = (int)(e & 0xffffffu);
                if (canon_bits(X[j]) == bx && canon_bits(X[(size_t)n_pad + j]) == by && canon_bits(X[2 * (size_t)n_pad + j]) == bz) { v = j != i; break; }
                h = (h + 1u) & mask;
            }
        }
        voided[i] = v ? 1 : 0;
    }
    const bool keep = i < n && !v;
    scan_out[i] = keep ? x : inf_<float>();
    scan_out[(size_t)n_pad + i] = keep ? y : inf_<float>();
    scan_out[2 * (size_t)n_pad + i] = keep ? z : inf_<float>();
    const unsigned long long m = __builtin_amdgcn_ballot_w64(v);
    if (m != 0ull && (threadIdx.x & 63) == 0) atomicAdd(count, (int)__builtin_popcountll(m));
}

hipError_t launch_duplicates_hashed(const float* X, int n, int n_pad, unsigned int* table, unsigned int table_entries, unsigned int gen, unsigned char* voided,
                                    int* count_dev, float* scan_out, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    if (table_entries == 0 || (table_entries & (table_entries - 1)) != 0 || table_entries < 2u * (unsigned int)n || n >= (1 << 24)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(prep_dup_insert_kernel, dim3((n + 255) / 256), dim3(256), 0, st, X, n, n_pad, table, table_entries - 1u, gen & 0xffu);
    hipLaunchKernelGGL(prep_dup_lookup_kernel, dim3((n_pad + 255) / 256), dim3(256), 0, st, X, n, n_pad, (const unsigned int*)table, table_entries - 1u, gen & 0xffu,
                       voided, count_dev, scan_out);
    return hipGetLastError();
}

// extent dx + dy + dz of every group of `group` consecutive entries of an order (blockIdx.y == 0: the cloud's own; 1: `order`),
// relative to the cloud's largest extent, summed in 2^-36 fixed point into out[2 * which + blockIdx.y]: four groups (one wave
// each) per block, one integer atomic per block
// (report: the LAST block of the launch -- a ticket -- copies the sums, the duplicate count and `seq` into pinned host memory: the host
// spins on that word instead of a copy back and a stream synchronisation, 15-20 us per cloud)
__global__ __launch_bounds__(256) void prep_extent_fixed_kernel(const float* __restrict__ X, int n, int n_pad, const int32_t* __restrict__ order, int group, int groups,
                                                                const unsigned int* __restrict__ enc, unsigned long long* __restrict__ out, int which,
                                                                unsigned int* __restrict__ ticket, const int* __restrict__ voided_count, PrepReport* __restrict__ report, unsigned int seq)
{
    __shared__ unsigned long long part[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = blockIdx.x * 4 + w, g0 = g * group;
    const bool sorted = blockIdx.y != 0;
    float lo[3] = {inf_<float>(), inf_<float>(), inf_<float>()}, hi[3] = {-inf_<float>(), -inf_<float>(), -inf_<float>()};
    if (g < groups)
        for (int k = g0 + lane; k < min(n, g0 + group); k += 64) {
            const int i = sorted ? order[k] : k;
            const float p[3] = {X[i], X[(size_t)n_pad + i], X[2 * (size_t)n_pad + i]};
            if (!finite3(p[0], p[1], p[2])) continue;
#pragma unroll
            for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], p[a]); hi[a] = fmaxf(hi[a], p[a]); }
        }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off, 64));
        }
    if (lane == 0) {
        float box[4];
        box_from_enc(enc, box);
        const double e = hi[0] >= lo[0] ? (double)(hi[0] - lo[0]) + (double)(hi[1] - lo[1]) + (double)(hi[2] - lo[2]) : 0.0;
        const double rel = box[3] > 0.f ? e / (double)box[3] : 0.0;   // <= 3
        part[w] = (unsigned long long)(rel * 68719476736.0);          // 2^36
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = part[0] + part[1] + part[2] + part[3];
        if (t != 0ull) atomicAdd(&out[2 * which + (sorted ? 1 : 0)], t);
        if (report != nullptr) {
            __threadfence();
            const unsigned int blocks = gridDim.x * gridDim.y;
            if (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == blocks - 1u) {
                __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int k = 0; k < 4; ++k)
                    __hip_atomic_store(&report->fixed[k], __hip_atomic_load(&out[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&report->voided, __hip_atomic_load(voided_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __threadfence_system();
                __hip_atomic_store(&report->seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

hipError_t launch_extents_fixed(const float* X, int n, int n_pad, const int32_t* order, int group, const unsigned int* enc, unsigned long long* out, int which, hipStream_t st,
                                unsigned int* ticket, const int* voided_count, PrepReport* report, unsigned int seq)
{
    if (n <= 0 || group <= 0) return hipSuccess;
    const int groups = (n + group - 1) / group;
    hipLaunchKernelGGL(prep_extent_fixed_kernel, dim3((groups + 3) / 4, order ? 2 : 1), dim3(256), 0, st, X, n, n_pad, order, group, groups, enc, out, which, ticket,
                       voided_count, report, seq);
    return hipGetLastError();
}

// the space-filling-curve order of a small cloud: keys from the layout kernel's bounding cube, one radix sort; perm_out[k] = k-th point
__global__ void prep_curve_keys_enc_kernel(const float* __restrict__ X, int n, int n_pad, const unsigned int* __restrict__ enc,
                                           unsigned int* __restrict__ keys, int32_t* __restrict__ vals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float box[4];
    box_from_enc(enc, box);
    const float x = X[i], y = X[(size_t)n_pad + i], z = X[2 * (size_t)n_pad + i];
    unsigned int code = 0x7fffffffu;
    const float ext = box[3];
    if (finite3(x, y, z) && ext >= 0.f) {
        const double scale = ext > 0.f ? 1023.0 / (double)ext : 0.0;
        const unsigned int qx = (unsigned int)fmin(1023.0, fmax(0.0, ((double)x - (double)box[0]) * scale));
        const unsigned int qy = (unsigned int)fmin(1023.0, fmax(0.0, ((double)y - (double)box[1]) * scale));
        const unsigned int qz = (unsigned int)fmin(1023.0, fmax(0.0, ((double)z - (double)box[2]) * scale));
        code = hilbert30(qx, qy, qz);
    }
    keys[i] = code;
    vals[i] = i;
}

// Morton-ordered view of a scan copy + its permutation, padded (+inf / 0x7fffffff)
__global__ void prep_gather_sorted_kernel(const float* __restrict__ Qs, int m, int m_pad, const int32_t* __restrict__ perm,
                                          float* __restrict__ out, int32_t* __restrict__ perm_pad)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m_pad) return;
    const int j = k < m ? perm[k] : -1;
#pragma unroll
    for (int a = 0; a < 3; ++a) out[(size_t)a * m_pad + k] = j >= 0 ? Qs[(size_t)a * m_pad + j] : inf_<float>();
    perm_pad[k] = j >= 0 ? j : 0x7fffffff;
}

// slot -> point map of the moving cloud: the Morton order, padding slots keep themselves
__global__ void prep_slot_map_kernel(const int32_t* __restrict__ perm, int n, int n_pad, int32_t* __restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_pad) out[k] = k < n ? perm[k] : k;
}

// The order's weight classes (round 4): a count's class is its leading one and `bits` (0..3) bits behind it -- exact below 8 --, i.e.
// classes 2^-bits wide; the rows of one class keep the order they have on the Hilbert curve.  Heaviest-first is about the END of a
// launch (no long row left for last); an exact order bought that with neighbours on the curve -- rows that list mostly the same
// chunks -- scattered over the launch and over the eight XCDs, every one of which then fetched every record for itself.
__device__ __forceinline__ unsigned int order_class(unsigned int h, int bits)
{
    if (h < 8u) return h;
    const int ex = 31 - __builtin_clz(h);
    const unsigned int cmask = (bits >= 0 && bits < 3) ? (7u << (3 - bits)) & 7u : 7u;
    return 8u + (unsigned int)(ex - 3) * 8u + ((h >> (ex - 3)) & 7u & cmask);   // <= 143 (20-bit counts)
}

__global__ void row_order_keys_kernel(unsigned int* __restrict__ hits, int rows, unsigned int* __restrict__ keys, int32_t* __restrict__ vals,
                                      unsigned long long* __restrict__ total_add, unsigned long long* __restrict__ total_zero, int coarse)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned int h = 0u;
    if (r < rows) {
        h = hits[r];
        hits[r] = 0u;                                          // (the next launch counts afresh)
        h = h > 0xfffffu ? 0xfffffu : h;
        // bits 20..27: the class, inverted (the sort looks at these alone: ascending = heaviest class first, and it is stable -- the rows
        // of a class keep the curve's order); bits 0..19: the exact count, for the split of the head (row_roles_kernel)
        keys[r] = ((255u - order_class(h, coarse)) << 20) | h;
        vals[r] = r;
    }
    // the sum of the counters (the target of the split rows derives from it): two words, this launch adds to one and clears
    // the other for the next
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) h += (unsigned int)__shfl_xor((int)h, off, 64);
    __shared__ unsigned int wsum[4];   // (one add per block: 1200 adds to one address were 17 us of every pass)
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = (unsigned long long)wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (t != 0ull) atomicAdd(total_add, t);
    }
    if (r == 0) *total_zero = 0ull;
}

// the roles of an ordered launch's blocks (NN_ORDER_*, icp_kernels.h): every block works the split of the NN_ORDER_HEAD heaviest
// rows out for itself (one scan of 1024 counters), block 0 writes their roles, all write the roles behind them
__global__ __launch_bounds__(1024) void row_roles_kernel(const unsigned int* __restrict__ keys, const int32_t* __restrict__ vals, int rows,
                                                         const unsigned long long* __restrict__ total, int min_part, int total_div, int32_t* __restrict__ roles)
{
    __shared__ int wsum[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int head = rows < NN_ORDER_HEAD ? rows : NN_ORDER_HEAD;
    const unsigned int h = t < head ? keys[t] & 0xfffffu : 0u;
    auto block_sum = [&](int v) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        __syncthreads();   // (the words are free again)
        if (lane == 0) wsum[w] = v;
        __syncthreads();
        int s = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += wsum[k];
        return s;
    };
    unsigned long long T = *total / (unsigned long long)(total_div > 0 ? total_div : 1024);
    if (T < (unsigned long long)min_part) T = (unsigned long long)min_part;
    int parts = t < head ? 1 : 0, E = head;
    if (min_part > 0) {
        for (int it = 0; it < 24; ++it) {   // (the target doubles until the parts fit the spare blocks: at most 20 times, the counters have 20 bits)
            unsigned int p = 1u;
            if ((unsigned long long)h > T) {
                const unsigned long long want = ((unsigned long long)h + T - 1ull) / T;   // >= 2
                p = want >= 64ull ? 64u : 1u << (32 - __builtin_clz((unsigned int)want - 1u));
            }
            parts = t < head ? (int)p : 0;
            E = block_sum(parts);
            if (E - head <= NN_ORDER_EXTRA) break;
            T *= 2ull;
            parts = t < head ? 1 : 0;
            E = head;
        }
    }
    int v = parts;   // inclusive running sum within the wave, then across the waves
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(v, o, 64); v += lane >= o ? u : 0; }
    __syncthreads();
    if (lane == 63) wsum[w] = v;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) base += k < w ? wsum[k] : 0;
    const int excl = base + v - parts;
    if (blockIdx.x == 0 && t < head) {
        const int row = vals[t];
        const int lg = 31 - __builtin_clz((unsigned int)parts);
        for (int p = 0; p < parts; ++p) roles[excl + p] = row | (p << NN_ROLE_ROW_BITS) | (lg << (NN_ROLE_ROW_BITS + NN_ROLE_PART_BITS));
    }
    const int j = E + (int)blockIdx.x * 1024 + t;   // the roles behind the head: one block each, in the sorted order
    if (j < rows + NN_ORDER_EXTRA) {
        const int src = head + (j - E);
        roles[j] = src < rows ? vals[src] : -1;
    }
}

// ONE launch instead of nine (round 4; up to NN_CONTROL_MAX_ROWS rows -- the share of one rank of eight of configs[4] has 9 768): the
// counters of the launch before are read and zeroed, sorted and dealt as roles by a single workgroup.  The order need not be exact --
// any order is exact for the RESULT; what it decides is how even the load is -- so the 20-bit counts are quantised to 8 bits (exact
// below 8, then 3 bits of mantissa per power of two: steps of 12 %) and sorted by two stable 4-bit counting passes in LDS: thread t
// owns 16 consecutive entries, counts them per digit in REGISTERS (sixteen 5-bit counters in two words: no read-modify-write chain
// through LDS), leaves the counts in its column of cnt[digit][thread]; one block-wide exclusive scan over the 16 x 1024 counters
// gives every (digit, thread) its first output slot, and an entry's slot is that plus its rank among the thread's earlier entries
// of the same digit -- the counter's value when it was counted.  Ties keep the row order: the same roles every run.
// The split of the heaviest rows follows row_roles_kernel statement by statement (on the exact counts).
constexpr int NN_CONTROL_MAX_ROWS = 16384;
__global__ __launch_bounds__(1024) void pass_control_kernel(unsigned int* __restrict__ hits, int rows, unsigned int* __restrict__ exact, int min_part, int total_div,
                                                            int32_t* __restrict__ roles, int coarse)
{
    __shared__ __attribute__((aligned(16))) unsigned char key[NN_CONTROL_MAX_ROWS];        // quantised count, inverted: ascending = heaviest first
    __shared__ __attribute__((aligned(16))) unsigned short ids[2][NN_CONTROL_MAX_ROWS];    // ping-pong: the order so far
    __shared__ __attribute__((aligned(16))) unsigned short cnt[16 * 1024];                 // [digit][thread]
    __shared__ int wsum[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    auto block_sum = [&](int v) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        __syncthreads();   // (the words are free again)
        if (lane == 0) wsum[w] = v;
        __syncthreads();
        int s = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += wsum[k];
        return s;
    };
    // ---- read + zero the counters (thread t: rows 16 t .. 16 t + 15, four 16-byte loads in flight), quantise ----
    unsigned long long mine = 0ull;
    {
        unsigned int h[16];
        const int r0 = t * 16;
        if (r0 + 16 <= rows) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint4 v = reinterpret_cast<const uint4*>(hits + r0)[q];
                h[4 * q] = v.x; h[4 * q + 1] = v.y; h[4 * q + 2] = v.z; h[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) reinterpret_cast<uint4*>(hits + r0)[q] = uint4{0u, 0u, 0u, 0u};   // (the next launch counts afresh)
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) { h[e] = r0 + e < rows ? hits[r0 + e] : 0u; if (r0 + e < rows) hits[r0 + e] = 0u; }
        }
        unsigned int packed[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            h[e] = h[e] > 0xfffffu ? 0xfffffu : h[e];
            mine += h[e];
            const unsigned int q = order_class(h[e], coarse);   // <= 143
            const unsigned int kq = r0 + e < rows ? 255u - q : 255u;   // (entries beyond the cloud sort behind everything; they are never dealt)
            packed[e >> 2] |= kq << (8 * (e & 3));
        }
        if (r0 + 16 <= rows) {
#pragma unroll
            for (int q = 0; q < 4; ++q) reinterpret_cast<uint4*>(exact + r0)[q] = uint4{h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]};
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) if (r0 + e < rows) exact[r0 + e] = h[e];
        }
        *reinterpret_cast<uint4*>(key + r0) = uint4{packed[0], packed[1], packed[2], packed[3]};
        unsigned int idp[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) idp[e] = (unsigned int)(r0 + 2 * e) | ((unsigned int)(r0 + 2 * e + 1) << 16);
        reinterpret_cast<uint4*>(&ids[0][r0])[0] = uint4{idp[0], idp[1], idp[2], idp[3]};
        reinterpret_cast<uint4*>(&ids[0][r0])[1] = uint4{idp[4], idp[5], idp[6], idp[7]};
    }
    // (the sum of the counters: at most 16 384 x 2^20 = 2^34 -- in two halves through the int reduction)
    const unsigned long long total = ((unsigned long long)(unsigned int)block_sum((int)(mine >> 17)) << 17) + (unsigned long long)(unsigned int)block_sum((int)(mine & 0x1ffffull));
    __syncthreads();
    // ---- two stable counting passes of 4 bits ----
    int cur = 0;
#pragma unroll 1
    for (int shift = 0; shift < 8; shift += 4) {
        // the thread's sixteen entries (two 16-byte reads), their digits, and -- in two words of sixteen 5-bit fields (<= 16 each) --
        // how many of each digit: an entry's rank among the thread's earlier entries of its digit is the field's value as it is counted
        unsigned int id16[16], rk = 0u, dg16[2] = {0u, 0u};
        {
            const uint4 a = reinterpret_cast<const uint4*>(&ids[cur][t * 16])[0], b = reinterpret_cast<const uint4*>(&ids[cur][t * 16])[1];
            const unsigned int wds[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int e = 0; e < 16; ++e) id16[e] = (wds[e >> 1] >> (16 * (e & 1))) & 0xffffu;
        }
        unsigned long long c_lo = 0ull, c_hi = 0ull;   // digits 0..11 in c_lo (5 bits each), 12..15 in c_hi
        unsigned int kv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) kv[e] = key[id16[e]];   // (independent reads: one LDS latency for all)
        unsigned int ranks[2] = {0u, 0u};   // sixteen 4-bit ranks ... a rank can be 15 at most (the sixteenth entry of one digit)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const unsigned int d = (kv[e] >> shift) & 15u;
            const unsigned int sh5 = 5u * (d < 12u ? d : d - 12u);
            const unsigned long long cw = d < 12u ? c_lo : c_hi;
            const unsigned int r = (unsigned int)(cw >> sh5) & 31u;
            if (d < 12u) c_lo += 1ull << sh5; else c_hi += 1ull << sh5;
            ranks[e >> 3] |= r << (4 * (e & 7));
            dg16[e >> 3] |= d << (4 * (e & 7));
        }
        (void)rk;
#pragma unroll
        for (int d = 0; d < 16; ++d) cnt[d * 1024 + t] = (unsigned short)((d < 12 ? (c_lo >> (5 * d)) : (c_hi >> (5 * (d - 12)))) & 31ull);
        __syncthreads();
        // exclusive scan of the 16 384 counters in (digit, thread) order: thread t scans entries [16 t, 16 t + 16)
        unsigned int cw8[8];
        {
            const uint4 a = reinterpret_cast<const uint4*>(&cnt[t * 16])[0], b = reinterpret_cast<const uint4*>(&cnt[t * 16])[1];
            cw8[0] = a.x; cw8[1] = a.y; cw8[2] = a.z; cw8[3] = a.w; cw8[4] = b.x; cw8[5] = b.y; cw8[6] = b.z; cw8[7] = b.w;
        }
        int loc[16], run = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) { loc[e] = run; run += (int)((cw8[e >> 1] >> (16 * (e & 1))) & 0xffffu); }
        int v = run;   // inclusive scan of the threads' sums
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(v, o, 64); v += lane >= o ? u : 0; }
        if (lane == 63) wsum[w] = v;
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) base += k < w ? wsum[k] : 0;
        const int excl = base + v - run;
        {
            unsigned int o8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o8[e] = (unsigned int)(excl + loc[2 * e]) | ((unsigned int)(excl + loc[2 * e + 1]) << 16);   // (< 16 384: fits)
            reinterpret_cast<uint4*>(&cnt[t * 16])[0] = uint4{o8[0], o8[1], o8[2], o8[3]};
            reinterpret_cast<uint4*>(&cnt[t * 16])[1] = uint4{o8[4], o8[5], o8[6], o8[7]};
        }
        __syncthreads();
        unsigned int pos[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const unsigned int d = (dg16[e >> 3] >> (4 * (e & 7))) & 15u;
            pos[e] = (unsigned int)cnt[d * 1024 + t] + ((ranks[e >> 3] >> (4 * (e & 7))) & 15u);   // (independent reads)
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) ids[cur ^ 1][pos[e]] = (unsigned short)id16[e];
        __syncthreads();
        cur ^= 1;
    }
    const unsigned short* order = ids[cur];   // order[k] = the k-th heaviest row
    // ---- the roles (row_roles_kernel, on the exact counts of the head) ----
    const int head = rows < NN_ORDER_HEAD ? rows : NN_ORDER_HEAD;
    const unsigned int h = t < head ? exact[order[t]] : 0u;   // (written by this block above; the barriers order it)
    unsigned long long T = total / (unsigned long long)(total_div > 0 ? total_div : 1024);
    if (T < (unsigned long long)min_part) T = (unsigned long long)min_part;
    int parts = t < head ? 1 : 0, E = head;
    if (min_part > 0) {
        for (int it = 0; it < 24; ++it) {   // (the target doubles until the parts fit the spare blocks)
            unsigned int p = 1u;
            if ((unsigned long long)h > T) {
                const unsigned long long want = ((unsigned long long)h + T - 1ull) / T;   // >= 2
                p = want >= 64ull ? 64u : 1u << (32 - __builtin_clz((unsigned int)want - 1u));
            }
            parts = t < head ? (int)p : 0;
            E = block_sum(parts);
            if (E - head <= NN_ORDER_EXTRA) break;
            T *= 2ull;
            parts = t < head ? 1 : 0;
            E = head;
        }
    }
    int v = parts;   // inclusive running sum within the wave, then across the waves
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(v, o, 64); v += lane >= o ? u : 0; }
    __syncthreads();
    if (lane == 63) wsum[w] = v;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) base += k < w ? wsum[k] : 0;
    const int excl = base + v - parts;
    if (t < head) {
        const int row = order[t];
        const int lg = 31 - __builtin_clz((unsigned int)parts);
        for (int p = 0; p < parts; ++p) roles[excl + p] = row | (p << NN_ROLE_ROW_BITS) | (lg << (NN_ROLE_ROW_BITS + NN_ROLE_PART_BITS));
    }
    for (int j = E + t; j < rows + NN_ORDER_EXTRA; j += 1024) {   // the roles behind the head: one block each, in the sorted order
        const int src = head + (j - E);
        roles[j] = src < rows ? (int)order[src] : -1;
    }
}

size_t row_order_temp_bytes(int rows)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (unsigned int*)nullptr, (unsigned int*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr,
                                    (unsigned int)(rows > 0 ? rows : 1), 20, 28);
    return bytes;
}

hipError_t launch_row_order(const RowOrderBuffers& b, unsigned int* hits, int rows, const int32_t** roles_out, hipStream_t st)
{
    if (rows <= 0 || rows >= (1 << NN_ROLE_ROW_BITS) || b.roles == nullptr || b.totals == nullptr) return hipErrorInvalidValue;
    if (rows <= NN_CONTROL_MAX_ROWS && b.control) {
        // one workgroup does it all (the exact counts go through keys[0], which the sort below would use)
        hipLaunchKernelGGL(pass_control_kernel, dim3(1), dim3(1024), 0, st, hits, rows, b.keys[0], b.min_part, b.total_div, b.roles, b.coarse);
        *roles_out = b.roles;
        return hipGetLastError();
    }
    unsigned long long* tot = b.totals + (b.seq & 1ull);
    hipLaunchKernelGGL(row_order_keys_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, hits, rows, b.keys[0], b.vals[0], tot, b.totals + ((b.seq + 1ull) & 1ull), b.coarse);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    size_t bytes = b.temp_bytes;
    e = rocprim::radix_sort_pairs(b.temp, bytes, b.keys[0], b.keys[1], b.vals[0], b.vals[1], (unsigned int)rows, 20, 28, st);   // (the class bits alone; stable)
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(row_roles_kernel, dim3((rows + NN_ORDER_EXTRA + 1023) / 1024), dim3(1024), 0, st, b.keys[1], b.vals[1], rows,
                       (const unsigned long long*)tot, b.min_part, b.total_div, b.roles);
    *roles_out = b.roles;
    return hipGetLastError();
}

size_t prep_sort_temp_bytes(int count)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (unsigned int*)nullptr, (unsigned int*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr,
                                    (unsigned int)(count > 0 ? count : 1));
    return bytes;
}

static hipError_t sort_pairs(const PrepBuffers& b, int count, int from, int end_bit, hipStream_t st)
{
    size_t bytes = b.temp_bytes;
    return rocprim::radix_sort_pairs(b.temp, bytes, b.keys[from], b.keys[from ^ 1], b.vals[from], b.vals[from ^ 1], (unsigned int)count, 0,
                                     (unsigned int)end_bit, st);
}

// voided[j] = 1 for every point with an exact lower-index twin, *count_dev += their number; then the scan copy
hipError_t launch_duplicates_and_scan_copy(const PrepBuffers& b, const float* X, int n, int n_pad, unsigned char* voided, int* count_dev,
                                           float* scan_out, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    const dim3 blk(256), grd((n + 255) / 256);
    int cur = 0;
    for (int axis = 2; axis >= 0; --axis) {   // least significant key first, stable passes
        hipLaunchKernelGGL(prep_axis_keys_kernel, grd, blk, 0, st, X, n, n_pad, axis, axis == 2 ? (const int32_t*)nullptr : b.vals[cur],
                           b.keys[cur], b.vals[cur]);
        if (hipError_t e = sort_pairs(b, n, cur, 32, st)) return e;
        cur ^= 1;
    }
    hipLaunchKernelGGL(prep_mark_duplicates_kernel, grd, blk, 0, st, X, n, n_pad, b.vals[cur], voided, count_dev);
    hipLaunchKernelGGL(prep_scan_copy_kernel, dim3((n_pad + 255) / 256), blk, 0, st, X, n, n_pad, voided, scan_out);
    return hipGetLastError();
}

// the same for a cloud in double (six stable 32-bit passes: z low, z high, y low, ...)
hipError_t launch_duplicates_and_scan_copy_f64(const PrepBuffers& b, const double* X, int n, int n_pad, unsigned char* voided, int* count_dev,
                                               double* scan_out, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    const dim3 blk(256), grd((n + 255) / 256);
    int cur = 0;
    bool first = true;
    for (int axis = 2; axis >= 0; --axis)
        for (int high = 0; high < 2; ++high) {
            hipLaunchKernelGGL(prep_axis_keys_f64_kernel, grd, blk, 0, st, X, n, n_pad, axis, high, first ? (const int32_t*)nullptr : b.vals[cur],
                               b.keys[cur], b.vals[cur]);
            if (hipError_t e = sort_pairs(b, n, cur, 32, st)) return e;
            cur ^= 1;
            first = false;
        }
    hipLaunchKernelGGL(prep_mark_duplicates_f64_kernel, grd, blk, 0, st, X, n, n_pad, b.vals[cur], voided, count_dev);
    hipLaunchKernelGGL(prep_scan_copy_f64_kernel, dim3((n_pad + 255) / 256), blk, 0, st, X, n, n_pad, voided, scan_out);
    return hipGetLastError();
}

// perm_out[k] = k-th point in Morton order; totals[0] / totals[1] = summed group extents of the given / the Morton order
// (group2 > 0: totals[2] / totals[3] = the same for groups of group2 -- the upper search level of a large model)
hipError_t launch_morton_order(const PrepBuffers& b, const float* X, int n, int n_pad, int group, int group2, int32_t* perm_out,
                               double* totals, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    const dim3 blk(256), grd((n + 255) / 256);
    {
        // (large clouds: a grid of blocks; their partial boxes lie in the sort's second key buffer, idle until the sort)
        const int bb = n >= (1 << 18) ? 256 : 1;
        float* partial = bb > 1 ? reinterpret_cast<float*>(b.keys[1]) : nullptr;
        hipLaunchKernelGGL(prep_bbox_kernel, dim3(bb), dim3(1024), 0, st, X, n, n_pad, b.box, partial);
        if (bb > 1) hipLaunchKernelGGL(prep_bbox_final_kernel, dim3(1), dim3(64), 0, st, (const float*)partial, bb, b.box);
    }
    const int hilbert = 1;   // (Z-order, round 3's A/B: rows of 128 on the 10 M-point surface 10-19 % looser at every level; its switch is gone)
    hipLaunchKernelGGL(prep_morton_keys_kernel, grd, blk, 0, st, X, n, n_pad, b.box, b.keys[0], b.vals[0], hilbert);
    if (hipError_t e = sort_pairs(b, n, 0, 31, st)) return e;
    if (hipError_t e = hipMemcpyAsync(perm_out, b.vals[1], (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, st)) return e;
    const int groups = (n + group - 1) / group;
    const dim3 ggrd(groups);
    hipLaunchKernelGGL(prep_group_extent_kernel, ggrd, dim3(64), 0, st, X, n, n_pad, (const int32_t*)nullptr, group, b.ext);
    hipLaunchKernelGGL(prep_sum_kernel, dim3(1), dim3(256), 0, st, b.ext, groups, totals, 0);
    hipLaunchKernelGGL(prep_group_extent_kernel, ggrd, dim3(64), 0, st, X, n, n_pad, (const int32_t*)perm_out, group, b.ext);
    hipLaunchKernelGGL(prep_sum_kernel, dim3(1), dim3(256), 0, st, b.ext, groups, totals, 1);
    if (group2 > 0) {
        const int groups2 = (n + group2 - 1) / group2;
        hipLaunchKernelGGL(prep_group_extent_kernel, dim3(groups2), dim3(64), 0, st, X, n, n_pad, (const int32_t*)nullptr, group2, b.ext);
        hipLaunchKernelGGL(prep_sum_kernel, dim3(1), dim3(256), 0, st, b.ext, groups2, totals, 2);
        hipLaunchKernelGGL(prep_group_extent_kernel, dim3(groups2), dim3(64), 0, st, X, n, n_pad, (const int32_t*)perm_out, group2, b.ext);
        hipLaunchKernelGGL(prep_sum_kernel, dim3(1), dim3(256), 0, st, b.ext, groups2, totals, 3);
    }
    return hipGetLastError();
}

hipError_t launch_curve_order_small(const PrepBuffers& b, const float* X, int n, int n_pad, const unsigned int* enc, int32_t* perm_out, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(prep_curve_keys_enc_kernel, dim3((n + 255) / 256), dim3(256), 0, st, X, n, n_pad, enc, b.keys[0], b.vals[0]);
    if (hipError_t e = sort_pairs(b, n, 0, 31, st)) return e;
    return hipMemcpyAsync(perm_out, b.vals[1], (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, st);
}

hipError_t launch_gather_sorted(const float* Qs, int m, int m_pad, const int32_t* perm, float* out, int32_t* perm_pad, hipStream_t st)
{
    if (m_pad <= 0) return hipSuccess;
    hipLaunchKernelGGL(prep_gather_sorted_kernel, dim3((m_pad + 255) / 256), dim3(256), 0, st, Qs, m, m_pad, perm, out, perm_pad);
    return hipGetLastError();
}

hipError_t launch_slot_map(const int32_t* perm, int n, int n_pad, int32_t* out, hipStream_t st)
{
    if (n_pad <= 0) return hipSuccess;
    hipLaunchKernelGGL(prep_slot_map_kernel, dim3((n_pad + 255) / 256), dim3(256), 0, st, perm, n, n_pad, out);
    return hipGetLastError();
}

// bounding box of every 8-point chunk of the duplicate-voided scan copy (voided = +inf entries are ignored; an
// all-void chunk gets lo = +inf, hi = -inf and is skipped by construction).  Once per model.
__global__ void model_boxes_kernel(const float* __restrict__ Qs, int m_pad, float* __restrict__ boxes)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c * 8 >= m_pad) return;
    float lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = inf_<float>(); hi[a] = -inf_<float>(); }
    for (int k = 0; k < 8; ++k) {
        const int j = c * 8 + k;
        if (j >= m_pad) break;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = Qs[(size_t)a * m_pad + j];
            if (v < inf_<float>() && v > -inf_<float>()) { lo[a] = __builtin_fminf(lo[a], v); hi[a] = __builtin_fmaxf(hi[a], v); }
        }
    }
    float* o = boxes + (size_t)c * 8;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2]; o[6] = 0.f; o[7] = 0.f;
}

// one representative per chunk of the scan copy (its first point that is not voided; +inf if there is none),
// SoA over round_up(m_pad / 8, 8) entries: the thinned-out model of the sparse kernel's cold start
__global__ void model_samples_kernel(const float* __restrict__ Qs, int m_pad, int ns_pad, float* __restrict__ samples)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ns_pad) return;
    float v[3] = {inf_<float>(), inf_<float>(), inf_<float>()};
    for (int k = 0; k < 8; ++k) {
        const int j = c * 8 + k;
        if (j >= m_pad) break;
        const float x = Qs[j];
        if (x < inf_<float>() && x > -inf_<float>()) { v[0] = x; v[1] = Qs[(size_t)m_pad + j]; v[2] = Qs[2 * (size_t)m_pad + j]; break; }
    }
    samples[c] = v[0];
    samples[(size_t)ns_pad + c] = v[1];
    samples[2 * (size_t)ns_pad + c] = v[2];
}

// the same two tables in double, for the fp64 form of the search (the model itself is the scan copy: nothing is voided)
__global__ void model_boxes_f64_kernel(const double* __restrict__ Qs, int m_pad, double* __restrict__ boxes)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c * 8 >= m_pad) return;
    double lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = inf_<double>(); hi[a] = -inf_<double>(); }
    for (int k = 0; k < 8; ++k) {
        const int j = c * 8 + k;
        if (j >= m_pad) break;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double v = Qs[(size_t)a * m_pad + j];
            if (v < inf_<double>() && v > -inf_<double>()) { lo[a] = __builtin_fmin(lo[a], v); hi[a] = __builtin_fmax(hi[a], v); }
        }
    }
    double* o = boxes + (size_t)c * 8;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2]; o[6] = 0.0; o[7] = 0.0;
}

__global__ void model_samples_f64_kernel(const double* __restrict__ Qs, int m_pad, int ns_pad, double* __restrict__ samples)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ns_pad) return;
    double v[3] = {inf_<double>(), inf_<double>(), inf_<double>()};
    for (int k = 0; k < 8; ++k) {
        const int j = c * 8 + k;
        if (j >= m_pad) break;
        const double x = Qs[j];
        if (x < inf_<double>() && x > -inf_<double>()) { v[0] = x; v[1] = Qs[(size_t)m_pad + j]; v[2] = Qs[2 * (size_t)m_pad + j]; break; }
    }
    samples[c] = v[0];
    samples[(size_t)ns_pad + c] = v[1];
    samples[2 * (size_t)ns_pad + c] = v[2];
}

hipError_t launch_model_tables_f64(const void* Q_soa, int m_pad, void* boxes, void* samples, hipStream_t st)
{
    if (m_pad <= 0) return hipSuccess;
    const int chunks = (m_pad + 7) / 8, ns_pad = (chunks + 7) / 8 * 8;
    hipLaunchKernelGGL(model_boxes_f64_kernel, dim3((chunks + 255) / 256), dim3(256), 0, st, (const double*)Q_soa, m_pad, (double*)boxes);
    hipLaunchKernelGGL(model_samples_f64_kernel, dim3((ns_pad + 255) / 256), dim3(256), 0, st, (const double*)Q_soa, m_pad, ns_pad, (double*)samples);
    return hipGetLastError();
}
size_t model_boxes_f64_bytes(int m_pad) { return (size_t)((m_pad + 7) / 8) * 8 * sizeof(double); }
size_t model_samples_f64_bytes(int m_pad) { return 3 * (size_t)(((m_pad / 8) + 7) / 8 * 8) * sizeof(double); }

size_t model_samples_bytes(int m_pad) { return 3 * (size_t)(((m_pad / 8) + 7) / 8 * 8) * sizeof(float); }

hipError_t launch_model_samples(const void* Qs_soa, int m_pad, float* samples, hipStream_t st)
{
    if (m_pad <= 0) return hipSuccess;
    const int ns_pad = ((m_pad / 8) + 7) / 8 * 8;
    hipLaunchKernelGGL(model_samples_kernel, dim3((ns_pad + 255) / 256), dim3(256), 0, st, (const float*)Qs_soa, m_pad, ns_pad, samples);
    return hipGetLastError();
}

// upper levels: the box of every 64 boxes of the level below (512, 32 768 model points), one wave per box; stored
// behind the chunk boxes, level after level
__global__ __launch_bounds__(256) void model_superboxes_kernel(const float* __restrict__ boxes, int chunks, int supers, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int sidx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (sidx >= supers) return;
    const int c = sidx * 64 + lane;
    float lo[3] = {inf_<float>(), inf_<float>(), inf_<float>()}, hi[3] = {-inf_<float>(), -inf_<float>(), -inf_<float>()};
    if (c < chunks) {
        const float4* bp = reinterpret_cast<const float4*>(boxes + (size_t)c * 8);
        const float4 b0 = bp[0], b1 = bp[1];
        lo[0] = b0.x; lo[1] = b0.y; lo[2] = b0.z; hi[0] = b0.w; hi[1] = b1.x; hi[2] = b1.y;
    }
    wave_box(lo, hi);
    if (lane == 0) {
        float* o = out + (size_t)sidx * 8;
        o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2]; o[6] = 0.f; o[7] = 0.f;
    }
}

size_t model_boxes_bytes(int m_pad)
{
    const size_t chunks = (size_t)(m_pad + 7) / 8, supers = (chunks + 63) / 64, thirds = (supers + 63) / 64;
    return (chunks + supers + thirds) * 8 * sizeof(float);
}

// chunk boxes and chunk samples of a model that is searched flat, in ONE launch (the upper box levels are read by the hierarchical
// search only): thread c does chunk c's box and -- the samples array is padded to a multiple of 8 chunks -- sample c
__global__ void model_boxes_samples_kernel(const float* __restrict__ Qs, int m_pad, int ns_pad, float* __restrict__ boxes, float* __restrict__ samples)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ns_pad) return;
    float lo[3], hi[3], sv[3] = {inf_<float>(), inf_<float>(), inf_<float>()};
    bool have = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = inf_<float>(); hi[a] = -inf_<float>(); }
    for (int k = 0; k < 8; ++k) {
        const int j = c * 8 + k;
        if (j >= m_pad) break;
        const float p[3] = {Qs[j], Qs[(size_t)m_pad + j], Qs[2 * (size_t)m_pad + j]};
#pragma unroll
        for (int a = 0; a < 3; ++a)
            if (p[a] < inf_<float>() && p[a] > -inf_<float>()) { lo[a] = __builtin_fminf(lo[a], p[a]); hi[a] = __builtin_fmaxf(hi[a], p[a]); }
        if (!have && p[0] < inf_<float>() && p[0] > -inf_<float>()) { sv[0] = p[0]; sv[1] = p[1]; sv[2] = p[2]; have = true; }   // (model_samples_kernel's rule)
    }
    if (c * 8 < m_pad) {
        float* o = boxes + (size_t)c * 8;
        o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2]; o[6] = 0.f; o[7] = 0.f;
    }
    samples[c] = sv[0];
    samples[(size_t)ns_pad + c] = sv[1];
    samples[2 * (size_t)ns_pad + c] = sv[2];
}

hipError_t launch_model_boxes_samples(const void* Qs_soa, int m_pad, float* boxes, float* samples, hipStream_t st)
{
    if (m_pad <= 0) return hipSuccess;
    const int ns_pad = ((m_pad / 8) + 7) / 8 * 8;
    hipLaunchKernelGGL(model_boxes_samples_kernel, dim3((ns_pad + 255) / 256), dim3(256), 0, st, (const float*)Qs_soa, m_pad, ns_pad, boxes, samples);
    return hipGetLastError();
}

hipError_t launch_model_boxes(const void* Qs_soa, int m_pad, float* boxes, hipStream_t st)
{
    if (m_pad <= 0) return hipSuccess;
    const int chunks = (m_pad + 7) / 8;
    hipLaunchKernelGGL(model_boxes_kernel, dim3((chunks + 255) / 256), dim3(256), 0, st, (const float*)Qs_soa, m_pad, boxes);
    const int supers = (chunks + 63) / 64;
    float* l2 = boxes + (size_t)chunks * 8;
    hipLaunchKernelGGL(model_superboxes_kernel, dim3((supers + 3) / 4), dim3(256), 0, st, (const float*)boxes, chunks, supers, l2);
    const int thirds = (supers + 63) / 64;
    hipLaunchKernelGGL(model_superboxes_kernel, dim3((thirds + 3) / 4), dim3(256), 0, st, (const float*)l2, supers, thirds,
                       l2 + (size_t)supers * 8);
    return hipGetLastError();
}

// One record per chunk for the hierarchical search of a large model: {box lo.xyz hi.x | hi.yz - - | x[8] | y[8] | z[8] | index[8]}
// = 40 words = 160 contiguous bytes, the layout of a hit's stage in LDS.  A hit is then two cache lines instead of five
// 32-byte pieces of five arrays (a model of millions of points is not L2-resident: 10 M x 10 M fetched 23 GB per early pass).
__global__ void model_records_kernel(const float* __restrict__ Qs, const float* __restrict__ boxes, const int32_t* __restrict__ perm, int m_pad,
                                     float* __restrict__ rec)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one 16-byte piece each
    const long long chunks = m_pad >> 3;
    if (t >= chunks * 10) return;
    const long long ch = t / 10;
    const int part = (int)(t % 10);
    float4 v;
    if (part < 2) v = *reinterpret_cast<const float4*>(boxes + ch * 8 + part * 4);
    else if (part < 8) v = *reinterpret_cast<const float4*>(Qs + (size_t)((part - 2) >> 1) * m_pad + ch * 8 + (part & 1) * 4);
    else {
        const int k0 = (int)(ch * 8) + (part - 8) * 4;
        int4 iv = perm ? *reinterpret_cast<const int4*>(perm + k0) : int4{k0, k0 + 1, k0 + 2, k0 + 3};
        v = float4{__int_as_float(iv.x), __int_as_float(iv.y), __int_as_float(iv.z), __int_as_float(iv.w)};
    }
    *reinterpret_cast<float4*>(rec + ch * NN_REC_WORDS + part * 4) = v;
}

size_t model_records_bytes(int m_pad) { return (size_t)(m_pad >> 3) * NN_REC_WORDS * sizeof(float); }

hipError_t launch_model_records(const void* Qs_soa, const float* boxes, const int32_t* perm, int m_pad, float* rec, hipStream_t st)
{
    if (m_pad <= 0) return hipSuccess;
    const long long pieces = (long long)(m_pad >> 3) * 10;
    hipLaunchKernelGGL(model_records_kernel, dim3((unsigned int)((pieces + 255) / 256)), dim3(256), 0, st, (const float*)Qs_soa, boxes, perm, m_pad, rec);
    return hipGetLastError();
}

}  // namespace icp
