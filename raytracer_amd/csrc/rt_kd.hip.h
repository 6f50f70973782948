// Photon-map build on the device: the kd-tree of Kdtree.cpp:11-52 (widest axis of the range, median
// as the pivot) in the implicit in-order layout the gather walks (a node is a range [lo, hi) of the
// photon array, its pivot sits at lo + (hi - lo) / 2 with the split axis beside it, ranges of <= 64
// photons are leaves).
//
// The ranges of that layout depend on the photon count alone, so every array position knows, for any
// level, which range it belongs to or that it already holds a pivot.  One level of the tree is
//   1. per range: bounds of the three coordinates (wave-reduced, then ordered-integer atomics), which
//      give the split axis (first maximum of the extents, Kdtree.cpp:14-22);
//   2. a key per position -- (start of the range, coordinate on the range's axis) for the positions of a
//      range, (position, 0) for pivots of earlier levels and for leaves, which therefore stay put;
//   3. one stable radix sort of (key, photon) pairs over the whole array.
// The reference sorts each range and takes element n/2 (Kdtree.cpp:24-35); the sort here puts the
// same median at the same position, and photons of equal coordinate may land on either side, as
// with any sort.  k-NN results do not depend on such choices.
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

namespace rtkd {

constexpr int LEAF = 64;     // a range of at most this many photons is scanned by one wave

struct Where {
    int lo, hi;       // the range the position belongs to at this level
    int path;         // ordinal of the range among the 2^depth of its depth
    int depth;        // levels descended: `level`, or fewer where the position was fixed as a pivot or its range became a leaf
    bool fixed;       // the position holds the pivot of an earlier level
};

__device__ __forceinline__ Where locate(int i, int n, int level)
{
    Where w{0, n, 0, 0, false};
    for (int l = 0; l < level; l++) {
        if (w.hi - w.lo <= LEAF) break;
        const int mid = w.lo + (w.hi - w.lo) / 2;
        if (i == mid) { w.fixed = true; break; }
        if (i < mid) { w.hi = mid; w.path = 2 * w.path; }
        else { w.lo = mid + 1; w.path = 2 * w.path + 1; }
        w.depth = l + 1;
    }
    return w;
}

// Where a position's segment -- a range of this level, a leaf, or a pivot fixed earlier -- stands among the segments of
// the level, in array order: the high part of the sort key (level + 2 bits where the position itself would take as
// many bits as the photon count).  Ranges of the level: 2 x path.  A leaf that stopped at depth d covers the paths
// [path << (level - d), (path + 1) << (level - d)) and takes twice the first.  The pivot fixed at depth d stands between its
// node's two halves: one below twice the first path of the right half.
__device__ __forceinline__ unsigned segment_ordinal(const Where& w, int level)
{
    const int up = level - w.depth;
    if (w.fixed) return 2u * (((unsigned)w.path << up) + (1u << (up - 1))) - 1u;      // (fixed at depth < level: up >= 1)
    return 2u * ((unsigned)w.path << up);
}
__host__ __device__ inline int key_bits_of_level(int level) { return 32 + level + 2; }

__device__ __forceinline__ unsigned order_f32(float v)
{
    unsigned u = __float_as_uint(v);
    return (u >> 31) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unorder_f32(unsigned u)
{
    return __uint_as_float((u >> 31) ? (u & 0x7FFFFFFFu) : ~u);
}

struct Source {
    const double* rec;   // photon records: position, power, direction (9 doubles) every `stride` doubles
    int stride;
    // the positions as floats, one 16-byte entry per photon (filled once by compact_kernel): every level reads a photon's
    // coordinates through the level's permutation, and 16 bytes from a dense array cost half the lines of three doubles
    // out of a 72-byte record
    const float4* xyz;
    __device__ float coord(unsigned photon, int ax) const { const float4 p = xyz[photon]; return ax == 0 ? p.x : (ax == 1 ? p.y : p.z); }
    __device__ float4 point(unsigned photon) const { return xyz[photon]; }
};
__global__ void compact_kernel(const double* __restrict__ rec, int stride, int n, float4* __restrict__ xyz)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* r = rec + (size_t)i * stride;
    xyz[i] = make_float4((float)r[0], (float)r[1], (float)r[2], 0.f);
}

// mins[path * 3 + c] / maxs[path * 3 + c]: bounds of coordinate c (ordered bits) over range `path`, the upper one stored
// inverted, so that one fill value -- all ones -- presets both arrays (and those of every level, which the host lays out
// behind one another and fills once).  A wave takes RUN consecutive steps of 64 positions and keeps the bounds
// in registers for as long as the steps lie inside one range, so the top levels -- one or a few ranges
// holding all the photons -- cost a few atomics per wave instead of a few per 64 photons (which took
// 0.3 ms per level on 1.1 M photons: every atomic of a level lands on the same six words).
// The deeper levels have many ranges (their atomics land on many words) and want many waves instead: a step's fetch
// through the permutation is a round trip to memory, and RUN of them in a row per wave left the device a fifth full.
__host__ __device__ inline int run_of_level(int level) { return level < 4 ? 32 : (level < 8 ? 8 : 2); }
__global__ void bounds_kernel(Source src, int n, int level, const unsigned* __restrict__ photon, unsigned* __restrict__ mins, unsigned* __restrict__ maxs)
{
    const int RUN = run_of_level(level);
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    unsigned lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    int held = -1;      // range whose bounds the registers hold (-1: none)
    auto flush = [&]() {
        if (held < 0) return;
        for (int c = 0; c < 3; c++)
            for (int off = 32; off > 0; off >>= 1) {
                lo[c] = min(lo[c], (unsigned)__shfl_xor((int)lo[c], off));
                hi[c] = max(hi[c], (unsigned)__shfl_xor((int)hi[c], off));
            }
        if (lane == 0)
            for (int c = 0; c < 3; c++) {
                atomicMin(&mins[(size_t)held * 3 + c], lo[c]);
                atomicMin(&maxs[(size_t)held * 3 + c], ~hi[c]);
            }
        for (int c = 0; c < 3; c++) { lo[c] = 0xFFFFFFFFu; hi[c] = 0u; }
        held = -1;
    };
    for (int r = 0; r < RUN; r++) {
        const long long i = (wave * RUN + r) * 64 + lane;
        if ((wave * RUN + r) * 64 >= n) break;
        const bool in = i < n;
        Where w = locate(in ? (int)i : n - 1, n, level);
        const bool part = in && !w.fixed && w.hi - w.lo > LEAF;
        unsigned v[3] = {0, 0, 0};
        if (part) {
            const float4 q = src.point(photon[i]);
            v[0] = order_f32(q.x); v[1] = order_f32(q.y); v[2] = order_f32(q.z);
        }
        // Ranges that still split hold more than LEAF = 64 positions, so a step of 64 meets at most two of them: the one its
        // first such lane is in, which the registers may hold already, and the one its last such lane is in.  Each leaves
        // the registers by one reduction over the wave and six atomics of one lane (a lane's own atomics -- 64 lanes on two or
        // three words, six times a step -- made the deepest level of 1.1 M photons take 0.4 ms).
        const unsigned long long m = __ballot(part);
        if (!m) continue;
        const int path_a = __shfl(w.path, __builtin_ctzll(m)), path_b = __shfl(w.path, 63 - __builtin_clzll(m));
        if (held != path_a) flush();
        held = path_a;
        const bool in_a = part && w.path == path_a;
        for (int c = 0; c < 3; c++) { lo[c] = min(lo[c], in_a ? v[c] : 0xFFFFFFFFu); hi[c] = max(hi[c], in_a ? v[c] : 0u); }
        if (path_b != path_a) {
            flush();
            held = path_b;
            const bool in_b = part && w.path == path_b;
            for (int c = 0; c < 3; c++) { lo[c] = min(lo[c], in_b ? v[c] : 0xFFFFFFFFu); hi[c] = max(hi[c], in_b ? v[c] : 0u); }
        }
    }
    flush();
}

__global__ void key_kernel(Source src, int n, int level, const unsigned* __restrict__ photon, const unsigned* __restrict__ mins,
                           const unsigned* __restrict__ maxs, unsigned long long* __restrict__ keys, unsigned char* __restrict__ axis_at)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Where w = locate(i, n, level);
    if (w.fixed || w.hi - w.lo <= LEAF) {
        // (stays put: a pivot is alone in its segment, a leaf's positions keep their order by their place in it)
        keys[i] = (unsigned long long)segment_ordinal(w, level) << 32 | (unsigned)(w.fixed ? 0 : i - w.lo);
        return;
    }
    float best = -1.f;
    int ax = 0;
    for (int c = 0; c < 3; c++) {
        float span = unorder_f32(~maxs[(size_t)w.path * 3 + c]) - unorder_f32(mins[(size_t)w.path * 3 + c]);
        if (span > best) { best = span; ax = c; }     // first maximum wins (Kdtree.cpp:14-22)
    }
    keys[i] = ((unsigned long long)segment_ordinal(w, level) << 32) | order_f32(src.coord(photon[i], ax));
    if (i == w.lo + (w.hi - w.lo) / 2) axis_at[i] = (unsigned char)ax;   // a property of the position: the median arrives by the sort
}

__global__ void iota_kernel(int n, unsigned* __restrict__ v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (unsigned)i;
}

// the gather's arrays: position + axis, power + direction, original index
__global__ void emit_kernel(Source src, int n, const unsigned* __restrict__ photon, const unsigned char* __restrict__ axis_at, float4* __restrict__ pos,
                            float* __restrict__ aux, int* __restrict__ order)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned p = photon[i];
    const double* r = src.rec + (size_t)p * src.stride;
    pos[i] = make_float4((float)r[0], (float)r[1], (float)r[2], (float)axis_at[i]);
    for (int c = 0; c < 6; c++) aux[(size_t)i * 6 + c] = (float)r[3 + c];
    order[i] = (int)p;
}

// one thread per position; the one at the start of a leaf writes the leaf's box (slot lo / 32: leaves
// hold at least 32 photons unless the whole map is one leaf)
__global__ void leaf_box_kernel(int n, int levels, const float4* __restrict__ pos, float* __restrict__ leaf_box)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Where w = locate(i, n, levels);
    if (w.fixed || i != w.lo) return;
    float lo[3] = {pos[i].x, pos[i].y, pos[i].z}, hi[3] = {pos[i].x, pos[i].y, pos[i].z};
    for (int j = w.lo + 1; j < w.hi; j++) {
        const float4 p = pos[j];
        lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
        hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
    }
    float* b = leaf_box + 6 * (size_t)(w.lo >> 5);
    for (int c = 0; c < 3; c++) { b[c] = lo[c]; b[3 + c] = hi[c]; }
}

// levels until every range is a leaf: the larger child of a range of s photons holds s / 2
inline int levels_for(long long n)
{
    int l = 0;
    while (n > LEAF) { n /= 2; l++; }
    return l;
}

// photons in emission order: records sorted by their (photon, bounce, slot) key
__global__ void permute_kernel(int n, const unsigned* __restrict__ from, const double* __restrict__ in, int in_stride, double* __restrict__ out)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long long)n * 9) return;
    const int i = (int)(g / 9), c = (int)(g % 9);
    out[g] = in[(size_t)from[i] * in_stride + c];
}
__global__ void gather_keys_kernel(int n, const double* __restrict__ in, int in_stride, int key_at, unsigned long long* __restrict__ keys, unsigned* __restrict__ vals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = (unsigned long long)__double_as_longlong(in[(size_t)i * in_stride + key_at]);
    vals[i] = (unsigned)i;
}

}  // namespace rtkd
