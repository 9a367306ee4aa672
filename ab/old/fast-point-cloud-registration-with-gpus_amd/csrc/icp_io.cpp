// icp_io.cpp -- the reference's input formats and synthetic generators (host side).
//   synthetic grid / models   src/ICP_point_to_point.cu:103-190, src/ICP_CPU.c:51-149, src/ICP_standard.cu:150-262
//   "x y z" / "x;y;z" text    src/CUDA/GPU_point_to_point_bunny.cu:463-497
//   OS1-16 packet dump        src/CUDA/GPU_point_to_point_real.cu:432-527
// Compiled with -ffp-contract=off so that the model clouds carry the reference's float roundings.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/icp_mi355x.h"

namespace {

constexpr int OS1_PACKET_BYTES = 12608;  // 16 azimuth blocks x 788 B
constexpr int OS1_BLOCK_BYTES = 788;     // 16 B header + 64 channels x 12 B + 4 B status
constexpr int OS1_BLOCKS = 16;
constexpr int OS1_CHANNELS = 64;
constexpr int OS1_FIRST_BEAM = 2;        // beams 2, 6, ..., 62 carry the 16 lasers of an OS1-16
constexpr int OS1_BEAM_STRIDE = 4;

bool read_file(const char* path, std::vector<unsigned char>& buf)
{
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    buf.resize(sz > 0 ? (size_t)sz : 0);
    const size_t got = sz > 0 ? std::fread(buf.data(), 1, (size_t)sz, f) : 0;
    std::fclose(f);
    return got == buf.size();
}

bool looks_like_text(const std::vector<unsigned char>& b)
{
    const size_t probe = b.size() < 4096 ? b.size() : 4096;
    for (size_t i = 0; i < probe; ++i) {
        const unsigned char c = b[i];
        if (!((c >= '0' && c <= '9') || c == '\n' || c == '\r' || c == ' ' || c == '\t')) return false;
    }
    return probe > 0;
}

}  // namespace

extern "C" {

int icp_synthetic_grid_f32(int W, float xy_min, float xy_max, float* D)
{
    if (W < 2 || !D) return ICP_ERR_INVALID;
    const float length = xy_max - xy_min;
    std::vector<float> lin((size_t)W);
    for (int i = 0; i < W; ++i) lin[i] = xy_min + ((float)i * length) / ((float)W - 1.0f);
    for (int k = 0; k < W; ++k)      // outer index walks x
        for (int j = 0; j < W; ++j) {  // inner index walks y
            const size_t i = (size_t)k * W + j;
            const double x = lin[k], y = lin[j];
            D[3 * i + 0] = lin[k];
            D[3 * i + 1] = lin[j];
            D[3 * i + 2] = (float)(x * x - y * y);  // pow(float, 2) promotes to double in the reference
        }
    return ICP_OK;
}

int icp_synthetic_grid_f64(int W, double xy_min, double xy_max, double* D)
{
    if (W < 2 || !D) return ICP_ERR_INVALID;
    const double length = xy_max - xy_min;
    std::vector<double> lin((size_t)W);
    for (int i = 0; i < W; ++i) lin[i] = xy_min + (double)i * length / ((double)W - 1.0);
    for (int k = 0; k < W; ++k)
        for (int j = 0; j < W; ++j) {
            const size_t i = (size_t)k * W + j;
            D[3 * i + 0] = lin[k];
            D[3 * i + 1] = lin[j];
            D[3 * i + 2] = lin[k] * lin[k] - lin[j] * lin[j];
        }
    return ICP_OK;
}

static void apply_colmajor_f32(const float r[9], const float t[3], const float* D, int n, float* M)
{
    for (int i = 0; i < n; ++i) {
        const float* d = D + 3 * (size_t)i;
        float* o = M + 3 * (size_t)i;
        for (int j = 0; j < 3; ++j) {
            float acc = 0.0f;
            acc += r[j + 0] * d[0];
            acc += r[j + 3] * d[1];
            acc += r[j + 6] * d[2];
            o[j] = acc;
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < 3; ++j) M[3 * (size_t)i + j] += t[j];
}

int icp_make_model_f32(const float* D, int n, const float ang[3], const float t[3], float* M)
{
    if (!D || !M || !ang || !t || n < 0) return ICP_ERR_INVALID;
    const float cx = (float)std::cos((double)ang[0]), cy = (float)std::cos((double)ang[1]), cz = (float)std::cos((double)ang[2]);
    const float sx = (float)std::sin((double)ang[0]), sy = (float)std::sin((double)ang[1]), sz = (float)std::sin((double)ang[2]);
    float r[9];  // column-major, src/ICP_point_to_point.cu:170-172
    r[0] = cy * cz;  r[1] = (cz * sx * sy) + (cx * sz);  r[2] = -(cx * cz * sy) + (sx * sz);
    r[3] = -cy * sz; r[4] = (cx * cz) - (sx * sy * sz);  r[5] = (cx * sy * sz) + (cz * sx);
    r[6] = sy;       r[7] = -cy * sx;                    r[8] = cx * cy;
    apply_colmajor_f32(r, t, D, n, M);
    return ICP_OK;
}

int icp_make_model_standard_f32(const float* D, int n, float* M)
{
    if (!D || !M || n < 0) return ICP_ERR_INVALID;
    // src/ICP_standard.cu:247-249 ships this literal matrix instead of rx*ry*rz
    static const float r[9] = {0.876485812f, -0.37591464f, 0.300767018f, -0.04386084f, 0.559789799f,
                               0.827473024f, -0.47942553f, -0.73846026f, 0.474159881f};
    static const float t[3] = {1.0f, -0.3f, 0.2f};
    apply_colmajor_f32(r, t, D, n, M);
    return ICP_OK;
}

int icp_make_model_cpu_f64(const double* D, int n, const double ang[3], const double t[3], double* M)
{
    if (!D || !M || !ang || !t || n < 0) return ICP_ERR_INVALID;
    const double c0 = std::cos(ang[0]), s0 = std::sin(ang[0]);
    const double c1 = std::cos(ang[1]), s1 = std::sin(ang[1]);
    const double c2 = std::cos(ang[2]), s2 = std::sin(ang[2]);
    // src/ICP_CPU.c:110-126: +sin above the diagonal
    const double rx[3][3] = {{1, 0, 0}, {0, c0, s0}, {0, -s0, c0}};
    const double ry[3][3] = {{c1, 0, -s1}, {0, 1, 0}, {s1, 0, c1}};
    const double rz[3][3] = {{c2, s2, 0}, {-s2, c2, 0}, {0, 0, 1}};
    double a[3][3], r[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += rx[i][k] * ry[k][j];
            a[i][j] = s;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += a[i][k] * rz[k][j];
            r[i][j] = s;
        }
    for (int i = 0; i < n; ++i) {
        const double* d = D + 3 * (size_t)i;
        for (int j = 0; j < 3; ++j) {
            double s = r[j][0] * d[0];
            s += r[j][1] * d[1];
            s += r[j][2] * d[2];
            M[3 * (size_t)i + j] = s + t[j];
        }
    }
    return ICP_OK;
}

int icp_read_xyz_text(const char* path, float* out, int cap_points)
{
    if (!path || (!out && cap_points > 0) || cap_points < 0) return ICP_ERR_INVALID;
    std::vector<unsigned char> buf;
    if (!read_file(path, buf)) return ICP_ERR_IO;
    buf.push_back('\0');
    const char* p = reinterpret_cast<const char*>(buf.data());
    const char* end = p + buf.size() - 1;
    long nvals = 0;
    while (p < end) {
        while (p < end && (*p == ' ' || *p == ';' || *p == ',' || *p == '\n' || *p == '\r' || *p == '\t')) ++p;
        if (p >= end) break;
        char* q = nullptr;
        const float v = std::strtof(p, &q);
        if (q == p) return ICP_ERR_IO;  // not a number
        if (nvals / 3 < cap_points) out[nvals] = v;
        ++nvals;
        p = q;
    }
    if (nvals % 3 != 0) return ICP_ERR_IO;
    return (int)(nvals / 3);
}

int icp_read_os1_ranges(const char* path, uint32_t* ranges, int cap, uint32_t* encoder_count0)
{
    if (!path || (!ranges && cap > 0) || cap < 0) return ICP_ERR_INVALID;
    std::vector<unsigned char> raw;
    if (!read_file(path, raw)) return ICP_ERR_IO;
    std::vector<unsigned char> bytes;
    if (looks_like_text(raw)) {  // one decimal byte value per line
        bytes.reserve(raw.size() / 3);
        size_t i = 0;
        while (i < raw.size()) {
            while (i < raw.size() && (raw[i] < '0' || raw[i] > '9')) ++i;
            if (i >= raw.size()) break;
            unsigned v = 0;
            while (i < raw.size() && raw[i] >= '0' && raw[i] <= '9') v = v * 10 + (raw[i++] - '0');
            if (v > 255) return ICP_ERR_IO;
            bytes.push_back((unsigned char)v);
        }
    } else {
        bytes.swap(raw);
    }
    const size_t packets = bytes.size() / OS1_PACKET_BYTES;
    if (packets == 0) return ICP_ERR_IO;
    // encoder count: bytes 12..13 of the first azimuth block header (lines 13-14 of the dump)
    if (encoder_count0) *encoder_count0 = (uint32_t)bytes[12] | ((uint32_t)bytes[13] << 8);
    long o = 0;
    for (size_t p = 0; p < packets; ++p)
        for (int blk = 0; blk < OS1_BLOCKS; ++blk) {
            const size_t base = p * OS1_PACKET_BYTES + (size_t)blk * OS1_BLOCK_BYTES + 16;
            for (int ch = OS1_FIRST_BEAM; ch < OS1_CHANNELS; ch += OS1_BEAM_STRIDE) {
                const unsigned char* w = &bytes[base + 12 * (size_t)ch];
                const uint32_t range = (uint32_t)w[0] | ((uint32_t)w[1] << 8) | (((uint32_t)w[2] & 0xFu) << 16);
                if (o < cap) ranges[o] = range;
                ++o;
            }
        }
    return (int)o;
}

int icp_read_os1_intrinsics(const char* path, float alt16[16], float az16[16])
{
    if (!path || !alt16 || !az16) return ICP_ERR_INVALID;
    std::vector<unsigned char> buf;
    if (!read_file(path, buf)) return ICP_ERR_IO;
    buf.push_back('\0');
    std::vector<double> vals;
    const char* p = reinterpret_cast<const char*>(buf.data());
    const char* end = p + buf.size() - 1;
    while (p < end) {
        // take lines that start with a number, skip header / blank lines
        const char* eol = (const char*)std::memchr(p, '\n', (size_t)(end - p));
        if (!eol) eol = end;
        char* q = nullptr;
        const double v = std::strtod(p, &q);
        if (q != p && q <= eol) vals.push_back(v);
        p = eol < end ? eol + 1 : end;
    }
    if (vals.size() != 128) return ICP_ERR_IO;
    for (int b = 0; b < 16; ++b) {
        alt16[b] = (float)vals[OS1_FIRST_BEAM + OS1_BEAM_STRIDE * b];
        az16[b] = (float)vals[64 + OS1_FIRST_BEAM + OS1_BEAM_STRIDE * b];
    }
    return ICP_OK;
}

}  // extern "C"
