// icp_k_plane.hip -- point-to-plane front end (kNN(4) + PCA normals: src/CUDA/GPU_point_to_plane_real.cu:54-188,413-423) and the hall
// ingest (Conversion<<<>>> :20-36, packet decode :432-488).
#include "icp_device.h"
#include <math.h>
#include <stdlib.h>
#include <cstring>

namespace icp {

// ------------------------------------------------------------------------------------------------
// point-to-plane front end: 4 nearest model neighbours of every model point (self dropped).
// One lane per model point, whole model streamed through LDS; a sorted (d, j) top-5 lives in
// registers, insertion happens under a (rare) wave-level branch.  Candidates arrive in ascending
// j, so "insert after every entry with d_e <= d" reproduces the reference's k+1 passes of
// first-arg-min with overwrite (src/CUDA/GPU_point_to_plane_real.cu:83-89).
// ------------------------------------------------------------------------------------------------
template <typename F, int TQ>
__global__ __launch_bounds__(NN_BLOCK) void knn4_kernel(const F* __restrict__ Q, int m, int m_pad,
                                                        int32_t* __restrict__ nbr)
{
    using V = typename Vec16<F>::type;
    constexpr int VN = Vec16<F>::N;
    __shared__ __attribute__((aligned(16))) F sq[3 * TQ];
    const int i = blockIdx.x * NN_BLOCK + threadIdx.x;
    const int is = i < m ? i : m - 1;
    const F px = Q[is], py = Q[(size_t)m_pad + is], pz = Q[2 * (size_t)m_pad + is];
    F bd[5];
    int bj[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) { bd[r] = inf_<F>(); bj[r] = 0; }

    for (int tile = 0; tile < m; tile += TQ) {
        const int len = min(TQ, m_pad - tile);
        __syncthreads();
        for (int e = threadIdx.x * VN; e < len; e += NN_BLOCK * VN) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
                *reinterpret_cast<V*>(&sq[a * TQ + e]) =
                    *reinterpret_cast<const V*>(&Q[(size_t)a * m_pad + tile + e]);
        }
        __syncthreads();
        const int real = min(len, m - tile);  // padded duplicates must not enter a top-k
        for (int c = 0; c < real; ++c) {
            const F d = dist2<F>(px, py, pz, sq[c], sq[TQ + c], sq[2 * TQ + c]);
            if (d < bd[4]) {
                const int j = tile + c;
                // insert keeping (d, j) ascending; equal d keeps the earlier (lower) j first
                F cd = d;
                int cj = j;
                bool shifting = false;  // once the new entry is placed, everything below moves down one slot
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    const bool sw = shifting || (cd < bd[r]);
                    shifting = sw;
                    const F td = bd[r];
                    const int tj = bj[r];
                    bd[r] = sw ? cd : td;
                    bj[r] = sw ? cj : tj;
                    cd = sw ? td : cd;
                    cj = sw ? tj : cj;
                }
            }
        }
    }
    if (i < m) {
#pragma unroll
        for (int r = 1; r < 5; ++r) nbr[(size_t)i * 4 + (r - 1)] = bj[r];
    }
}

// ------------------------------------------------------------------------------------------------
// kNN(4), fp32, v2: the matching kernel's machinery (packed distances over two query points per lane, four
// waves splitting the block's model segment, grid.y segments, 8-point chunks with a wave-uniform early-out)
// carrying a sorted (d, j) top-5 per query instead of a single minimum.  A chunk is examined element-wise only
// when some lane's chunk minimum beats that lane's threshold = min(5th best so far, seeded bound).  The seeded
// bound is the largest distance to five DISTINCT model points around the query's own index, bumped one ulp: at
// least five points lie strictly under it, so the exact top-5 survives; the seed only prunes work.
// Per-wave lists are merged through LDS (ties -> the lower wave = lower indices), per-segment lists by
// knn4_merge_kernel (ties -> the lower segment).  Rank 0 (self or an equal-distance lower index) is dropped there.
// ------------------------------------------------------------------------------------------------
struct Top5 {
    float d[5];
    int j[5];
};

__device__ __forceinline__ void top5_insert(Top5& L, float d, int j)
{
    float cd = d;
    int cj = j;
    bool shifting = false;  // once placed, everything below moves down one slot (keeps equal-d entries index-ordered)
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const bool sw = shifting || (cd < L.d[r]);
        shifting = sw;
        const float td = L.d[r];
        const int tj = L.j[r];
        L.d[r] = sw ? cd : td;
        L.j[r] = sw ? cj : tj;
        cd = sw ? td : cd;
        cj = sw ? tj : cj;
    }
}

constexpr int KNN_C = 8;

__global__ __launch_bounds__(NN_BLOCK, 4) void knn4_f32_v2(const float* __restrict__ Q, int m, int m_pad, int n_pad,
                                                           int seg_len, float* __restrict__ part_d,
                                                           int32_t* __restrict__ part_j)
{
    constexpr int C = KNN_C;
    __shared__ __attribute__((aligned(16))) float sq[4][3][NN2_TQW];
    __shared__ float ld[4][128][5];
    __shared__ int lj[4][128][5];

    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int wseg = seg_len >> 2;
    const int q0 = blockIdx.y * seg_len;
    const int my0 = q0 + w * wseg;
    const int my1 = min(my0 + wseg, m_pad);
    const int ibase = blockIdx.x * 128 + lane;
    const int i0 = min(ibase, m - 1), i1 = min(ibase + 64, m - 1);  // queries are model points; padding lanes repeat the last

    const f2 px = f2{Q[i0], Q[i1]}, py = f2{Q[(size_t)m_pad + i0], Q[(size_t)m_pad + i1]},
             pz = f2{Q[2 * (size_t)m_pad + i0], Q[2 * (size_t)m_pad + i1]};
    Top5 L[2];
    float bound[2], thr[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 5; ++r) { L[t].d[r] = inf_<float>(); L[t].j[r] = 0x7fffffff; }
        const int i = t ? i1 : i0;
        const float x = t ? px.y : px.x, y = t ? py.y : py.x, z = t ? pz.y : pz.x;
        const int lo = max(0, min(i - 2, m - 5));  // five distinct indices around the query's own
        float mx = 0.f;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int j = lo + k;
            mx = fmaxf(mx, dist2<float>(x, y, z, Q[j], Q[(size_t)m_pad + j], Q[2 * (size_t)m_pad + j]));
        }
        bound[t] = (mx < inf_<float>()) ? __uint_as_float(__float_as_uint(mx) + 1u) : mx;
        thr[t] = bound[t];
    }

    const int ntile = (wseg + NN2_TQW - 1) / NN2_TQW;
    for (int k = 0; k < ntile; ++k) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int v = threadIdx.x + r * NN_BLOCK;
            const int ww = v / 192, rem = v % 192;
            const int a = rem / 64, e = (rem % 64) * 4;
            const int off = k * NN2_TQW + e;
            const int src = q0 + ww * wseg + off;
            if (off < wseg && src < m_pad)
                *reinterpret_cast<float4*>(&sq[ww][a][e]) = *reinterpret_cast<const float4*>(&Q[(size_t)a * m_pad + src]);
        }
        __syncthreads();
        const int tile0 = my0 + k * NN2_TQW;
        const int len = min(NN2_TQW, my1 - tile0);
        for (int c = 0; c < len; c += C) {
            f2 dd[C];
            float cmin0 = inf_<float>(), cmin1 = inf_<float>();
#pragma unroll
            for (int kk = 0; kk < C; kk += 4) {
                const float4 qx4 = *reinterpret_cast<const float4*>(&sq[w][0][c + kk]);
                const float4 qy4 = *reinterpret_cast<const float4*>(&sq[w][1][c + kk]);
                const float4 qz4 = *reinterpret_cast<const float4*>(&sq[w][2][c + kk]);
                const f2 qxa = f2{qx4.x, qx4.y}, qxb = f2{qx4.z, qx4.w};
                const f2 qya = f2{qy4.x, qy4.y}, qyb = f2{qy4.z, qy4.w};
                const f2 qza = f2{qz4.x, qz4.y}, qzb = f2{qz4.z, qz4.w};
                dd[kk + 0] = pk_dist2<0>(qxa, qya, qza, px, py, pz);
                dd[kk + 1] = pk_dist2<1>(qxa, qya, qza, px, py, pz);
                dd[kk + 2] = pk_dist2<0>(qxb, qyb, qzb, px, py, pz);
                dd[kk + 3] = pk_dist2<1>(qxb, qyb, qzb, px, py, pz);
                cmin0 = fmin_(fmin_(cmin0, dd[kk].x), dd[kk + 1].x);
                cmin0 = fmin_(fmin_(cmin0, dd[kk + 2].x), dd[kk + 3].x);
                cmin1 = fmin_(fmin_(cmin1, dd[kk].y), dd[kk + 1].y);
                cmin1 = fmin_(fmin_(cmin1, dd[kk + 2].y), dd[kk + 3].y);
            }
            const bool need = (cmin0 < thr[0]) | (cmin1 < thr[1]);
            if (__builtin_amdgcn_ballot_w64(need) == 0ull) continue;
#pragma unroll
            for (int kk = 0; kk < C; ++kk) {
                const int j = tile0 + c + kk;
                const bool real = j < m;  // padded duplicates of the last point must not enter a top-k
                if (real && dd[kk].x < thr[0]) { top5_insert(L[0], dd[kk].x, j); thr[0] = fmin_(bound[0], L[0].d[4]); }
                if (real && dd[kk].y < thr[1]) { top5_insert(L[1], dd[kk].y, j); thr[1] = fmin_(bound[1], L[1].d[4]); }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            ld[w][lane + t * 64][r] = L[t].d[r];
            lj[w][lane + t * 64][r] = L[t].j[r];
        }
    __syncthreads();
    if (threadIdx.x < 128) {
        // 4-way merge of sorted lists; on equal d the lower wave (lower indices) goes first
        int h[4] = {0, 0, 0, 0};
        const size_t o = ((size_t)blockIdx.y * n_pad + (size_t)blockIdx.x * 128 + threadIdx.x) * 5;
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            float bd = inf_<float>();
            int bw = 0;
#pragma unroll
            for (int ww = 3; ww >= 0; --ww) {
                const float d = h[ww] < 5 ? ld[ww][threadIdx.x][h[ww]] : inf_<float>();
                if (d <= bd) { bd = d; bw = ww; }   // descending ww with <= : the lowest wave wins ties
            }
            const int hj = h[bw] < 5 ? lj[bw][threadIdx.x][h[bw]] : 0x7fffffff;
            part_d[o + r] = bd;
            part_j[o + r] = bd < inf_<float>() ? hj : 0x7fffffff;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) h[ww] += (ww == bw) ? 1 : 0;
        }
    }
}

// merge the S per-segment top-5 lists of every query (ascending segments, earlier segment first on equal d),
// drop rank 0, store the 4 neighbour indices
__global__ void knn4_merge_kernel(const float* __restrict__ part_d, const int32_t* __restrict__ part_j, int S, int n_pad,
                                  int m, int32_t* __restrict__ nbr)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    Top5 L;
#pragma unroll
    for (int r = 0; r < 5; ++r) { L.d[r] = part_d[(size_t)i * 5 + r]; L.j[r] = part_j[(size_t)i * 5 + r]; }
    for (int s = 1; s < S; ++s) {
        const size_t o = ((size_t)s * n_pad + i) * 5;
        for (int r = 0; r < 5; ++r) {
            const float d = part_d[o + r];
            if (!(d < L.d[4])) break;  // lists are sorted: nothing further in this segment can enter
            top5_insert(L, d, part_j[o + r]);
        }
    }
#pragma unroll
    for (int r = 1; r < 5; ++r) nbr[(size_t)i * 4 + (r - 1)] = L.j[r];
}

// PCA normal of every model point from its 4 neighbours, entirely on the device: float covariance in the
// order of src/CUDA/CPU_ICP_point_to-plane.cpp:217-246 (bar = sum * 0.25f, A += (x-bar)(y-bar), not divided by
// k), then a cyclic-Jacobi eigen-solve in fp64 registers (stands in for the reference's HOST loop of
// LAPACKE_ssyev, src/ICP_point_to_plane.cu:429-438) and the eigenvector of the eigenvalue of smallest magnitude
// (cblas_isamin over the ascending eigenvalues, first on ties).  Writes the padded SoA normal cloud directly.
template <typename F>
__global__ void normals_kernel(const F* __restrict__ Q, int m, int m_pad, const int32_t* __restrict__ nbr,
                               F* __restrict__ Nrm)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m_pad) return;
    const int src = i < m ? i : m - 1;  // padding replicates the last point's normal (never referenced)
    float x[4], y[4], z[4];
    float bx = 0.f, by = 0.f, bz = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int s = nbr[(size_t)src * 4 + j];
        x[j] = (float)Q[s];
        y[j] = (float)Q[(size_t)m_pad + s];
        z[j] = (float)Q[2 * (size_t)m_pad + s];
        bx += x[j]; by += y[j]; bz += z[j];
    }
    const float qa = 1.0f / 4.0f;
    bx *= qa; by *= qa; bz *= qa;
    float A[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float dx = x[j] - bx, dy = y[j] - by, dz = z[j] - bz;
        A[0] += dx * dx; A[1] += dx * dy; A[2] += dx * dz;
        A[3] += dy * dy; A[4] += dy * dz; A[5] += dz * dz;
    }
    // symmetric 3x3 in named scalars (no runtime-indexed arrays -> no scratch)
    double a00 = A[0], a01 = A[1], a02 = A[2], a11 = A[3], a12 = A[4], a22 = A[5];
    double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
    for (int sweep = 0; sweep < 32; ++sweep) {
        const double off = a01 * a01 + a02 * a02 + a12 * a12;
        const double dia = a00 * a00 + a11 * a11 + a22 * a22;
        if (off <= 1e-34 * dia || off == 0.0) break;
        // rotation (p,q) = (0,1): r = 2
        if (a01 != 0.0) {
            const double th = (a11 - a00) / (2.0 * a01);
            const double t = copysign(1.0, th) / (fabs(th) + sqrt(th * th + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            a00 -= t * a01; a11 += t * a01; a01 = 0.0;
            const double rp = a02, rq = a12;
            a02 = c * rp - s * rq; a12 = s * rp + c * rq;
            double p, q;
            p = v00; q = v01; v00 = c * p - s * q; v01 = s * p + c * q;
            p = v10; q = v11; v10 = c * p - s * q; v11 = s * p + c * q;
            p = v20; q = v21; v20 = c * p - s * q; v21 = s * p + c * q;
        }
        // (0,2): r = 1
        if (a02 != 0.0) {
            const double th = (a22 - a00) / (2.0 * a02);
            const double t = copysign(1.0, th) / (fabs(th) + sqrt(th * th + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            a00 -= t * a02; a22 += t * a02; a02 = 0.0;
            const double rp = a01, rq = a12;
            a01 = c * rp - s * rq; a12 = s * rp + c * rq;
            double p, q;
            p = v00; q = v02; v00 = c * p - s * q; v02 = s * p + c * q;
            p = v10; q = v12; v10 = c * p - s * q; v12 = s * p + c * q;
            p = v20; q = v22; v20 = c * p - s * q; v22 = s * p + c * q;
        }
        // (1,2): r = 0
        if (a12 != 0.0) {
            const double th = (a22 - a11) / (2.0 * a12);
            const double t = copysign(1.0, th) / (fabs(th) + sqrt(th * th + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            a11 -= t * a12; a22 += t * a12; a12 = 0.0;
            const double rp = a01, rq = a02;
            a01 = c * rp - s * rq; a02 = s * rp + c * rq;
            double p, q;
            p = v01; q = v02; v01 = c * p - s * q; v02 = s * p + c * q;
            p = v11; q = v12; v11 = c * p - s * q; v12 = s * p + c * q;
            p = v21; q = v22; v21 = c * p - s * q; v22 = s * p + c * q;
        }
    }
    // ascending eigenvalues (stable w.r.t. the original slot), then the first of smallest |w| as floats
    double w0 = a00, w1 = a11, w2 = a22;
    double e0x = v00, e0y = v10, e0z = v20, e1x = v01, e1y = v11, e1z = v21, e2x = v02, e2y = v12, e2z = v22;
#define ICP_SWAP_EIG(wa, ax, ay, az, wb, bx_, by_, bz_) \
    if (wb < wa) { double tw = wa; wa = wb; wb = tw; double tx = ax; ax = bx_; bx_ = tx; double ty = ay; ay = by_; by_ = ty; double tz = az; az = bz_; bz_ = tz; }
    ICP_SWAP_EIG(w0, e0x, e0y, e0z, w1, e1x, e1y, e1z)
    ICP_SWAP_EIG(w0, e0x, e0y, e0z, w2, e2x, e2y, e2z)
    ICP_SWAP_EIG(w1, e1x, e1y, e1z, w2, e2x, e2y, e2z)
#undef ICP_SWAP_EIG
    double nx = e0x, ny = e0y, nz = e0z;
    float wm = fabsf((float)w0);
    if (fabsf((float)w1) < wm) { wm = fabsf((float)w1); nx = e1x; ny = e1y; nz = e1z; }
    if (fabsf((float)w2) < wm) { nx = e2x; ny = e2y; nz = e2z; }
    Nrm[i] = (F)nx;
    Nrm[(size_t)m_pad + i] = (F)ny;
    Nrm[2 * (size_t)m_pad + i] = (F)nz;
}

// ------------------------------------------------------------------------------------------------
// OS1-16 polar -> Cartesian (mm), one range per lane
// ------------------------------------------------------------------------------------------------
__global__ void os1_conversion_kernel(const uint32_t* __restrict__ r, int n, uint32_t encoder0,
                                      const float* __restrict__ altitude, const float* __restrict__ azimuth,
                                      float* __restrict__ xyz)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int azimuth_block = i / 16, channel = i % 16;
    const unsigned long long counter = ((unsigned long long)encoder0 + (unsigned long long)azimuth_block * 88ull) % 90112ull;
    const float theta = (float)(2.0 * M_PI * ((double)counter / 90112.0 + (double)azimuth[channel] / 360.0));
    const float phi = (float)(2.0 * M_PI * (double)altitude[channel] / 360.0);
    const float rr = (float)r[i];
    const float ct = cosf(theta), st = sinf(theta), cp = cosf(phi), sp = sinf(phi);
    xyz[3 * (size_t)i + 0] = rr * ct * cp;
    xyz[3 * (size_t)i + 1] = -rr * st * cp;
    xyz[3 * (size_t)i + 2] = rr * sp;
}

// raw OS1-16 packets (12 608 B each: 16 azimuth blocks x [16 B header | 64 channels x 12 B | 4 B status]) ->
// ranges [mm] + Cartesian points [mm] in one pass; one lane per (packet, block, beam).  Replaces the host
// parse loop + H2D + Conversion of src/CUDA/GPU_point_to_point_real.cu:457-487,538-563.  Byte-granular reads
// (the 20-bit range sits at an arbitrary byte offset); 3 bytes per lane, ~0.8 MB for the hall dump.
__global__ void os1_packets_kernel(const uint8_t* __restrict__ packets, int n_packets, const float* __restrict__ altitude,
                                   const float* __restrict__ azimuth, uint32_t* __restrict__ ranges,
                                   float* __restrict__ xyz)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_packets * 256) return;
    const int packet = i / 256, blk = (i / 16) % 16, beam = i % 16;
    const int ch = 2 + 4 * beam;  // the 16 lasers of an OS1-16 sit in channels 2, 6, ..., 62
    const size_t w = (size_t)packet * 12608 + (size_t)blk * 788 + 16 + 12 * (size_t)ch;
    const uint32_t r = (uint32_t)packets[w] | ((uint32_t)packets[w + 1] << 8) | (((uint32_t)packets[w + 2] & 0xFu) << 16);
    const uint32_t encoder0 = (uint32_t)packets[12] | ((uint32_t)packets[13] << 8);  // first block of the first packet
    ranges[i] = r;
    const int azimuth_block = i / 16;
    const unsigned long long counter = ((unsigned long long)encoder0 + (unsigned long long)azimuth_block * 88ull) % 90112ull;
    const float theta = (float)(2.0 * M_PI * ((double)counter / 90112.0 + (double)azimuth[beam] / 360.0));
    const float phi = (float)(2.0 * M_PI * (double)altitude[beam] / 360.0);
    const float rr = (float)r;
    const float ct = cosf(theta), st = sinf(theta), cp = cosf(phi), sp = sinf(phi);
    xyz[3 * (size_t)i + 0] = rr * ct * cp;
    xyz[3 * (size_t)i + 1] = -rr * st * cp;
    xyz[3 * (size_t)i + 2] = rr * sp;
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
void knn4_v2_geometry(int m, int num_cus, int* n_pad, int* blocks_x, int* splits, int* seg_len)
{
    const int m_pad = pad_model(m);
    *n_pad = round_up(m, 128);
    *blocks_x = *n_pad / 128;
    if (num_cus <= 0) num_cus = 256;
    int S = (num_cus * 4 + *blocks_x - 1) / *blocks_x;        // 4 blocks (16 waves) per CU
    const int max_S = (m_pad + 511) / 512;
    if (S > max_S) S = max_S;
    if (S < 1) S = 1;
    int seg = round_up((m_pad + S - 1) / S, 4 * KNN_C);
    *splits = (m_pad + seg - 1) / seg;
    *seg_len = seg;
}

hipError_t launch_knn4_v2(const void* Q, int m, int num_cus, float* part_d, int32_t* part_j, int32_t* nbr, hipStream_t st)
{
    if (m <= 0) return hipSuccess;
    int n_pad, bx, S, seg;
    knn4_v2_geometry(m, num_cus, &n_pad, &bx, &S, &seg);
    hipLaunchKernelGGL(knn4_f32_v2, dim3(bx, S), dim3(NN_BLOCK), 0, st, (const float*)Q, m, pad_model(m), n_pad, seg, part_d,
                       part_j);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(knn4_merge_kernel, dim3((m + 255) / 256), dim3(256), 0, st, (const float*)part_d,
                       (const int32_t*)part_j, S, n_pad, m, nbr);
    return hipGetLastError();
}

hipError_t launch_knn4(const NNPlan& pl, const void* Q, int32_t* nbr, hipStream_t st)
{
    if (pl.m <= 0) return hipSuccess;
    const int blocks = (pl.m + NN_BLOCK - 1) / NN_BLOCK;
    if (pl.precision == ICP_F64)
        hipLaunchKernelGGL((knn4_kernel<double, 1024>), dim3(blocks), dim3(NN_BLOCK), 0, st, (const double*)Q, pl.m,
                           pl.m_pad, nbr);
    else
        hipLaunchKernelGGL((knn4_kernel<float, 2048>), dim3(blocks), dim3(NN_BLOCK), 0, st, (const float*)Q, pl.m,
                           pl.m_pad, nbr);
    return hipGetLastError();
}

hipError_t launch_normals(int precision, const void* Q, int m, int m_pad, const int32_t* nbr, void* Nrm_soa,
                          hipStream_t st)
{
    if (m <= 0) return hipSuccess;
    const int blocks = (m_pad + 127) / 128;
    if (precision == ICP_F64)
        hipLaunchKernelGGL((normals_kernel<double>), dim3(blocks), dim3(128), 0, st, (const double*)Q, m, m_pad, nbr,
                           (double*)Nrm_soa);
    else
        hipLaunchKernelGGL((normals_kernel<float>), dim3(blocks), dim3(128), 0, st, (const float*)Q, m, m_pad, nbr,
                           (float*)Nrm_soa);
    return hipGetLastError();
}

hipError_t launch_os1_packets(const uint8_t* packets, int n_packets, const float* alt16, const float* az16,
                              uint32_t* ranges, float* xyz_aos, hipStream_t st)
{
    if (n_packets <= 0) return hipSuccess;
    const int n = n_packets * 256;
    hipLaunchKernelGGL(os1_packets_kernel, dim3((n + 255) / 256), dim3(256), 0, st, packets, n_packets, alt16, az16, ranges,
                       xyz_aos);
    return hipGetLastError();
}

hipError_t launch_os1_conversion(const uint32_t* ranges, int n, uint32_t encoder0, const float* alt16,
                                 const float* az16, float* xyz_aos, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(os1_conversion_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ranges, n, encoder0, alt16, az16,
                       xyz_aos);
    return hipGetLastError();
}

}  // namespace icp
