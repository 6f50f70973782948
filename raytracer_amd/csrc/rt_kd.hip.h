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
    int path;         // ordinal of the range among the 2^level of its level
    bool fixed;       // the position holds the pivot of an earlier level
};

__device__ __forceinline__ Where locate(int i, int n, int level)
{
    Where w{0, n, 0, false};
    for (int l = 0; l < level; l++) {
        if (w.hi - w.lo <= LEAF) break;
        const int mid = w.lo + (w.hi - w.lo) / 2;
        if (i == mid) { w.fixed = true; break; }
        if (i < mid) { w.hi = mid; w.path = 2 * w.path; }
        else { w.lo = mid + 1; w.path = 2 * w.path + 1; }
    }
    return w;
}

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
    __device__ float coord(unsigned photon, int ax) const { return (float)rec[(size_t)photon * stride + ax]; }
};

// mins[path * 3 + c] / maxs[path * 3 + c]: bounds of coordinate c (ordered bits) over range `path`.
// Preset to all ones / zero.  A wave takes RUN consecutive steps of 64 positions and keeps the bounds
// in registers for as long as the steps lie inside one range, so the top levels -- one or a few ranges
// holding all the photons -- cost a few atomics per wave instead of a few per 64 photons (which took
// 0.3 ms per level on 1.1 M photons: every atomic of a level lands on the same six words).
constexpr int RUN = 32;
__global__ void bounds_kernel(Source src, int n, int level, const unsigned* __restrict__ photon, unsigned* __restrict__ mins, unsigned* __restrict__ maxs)
{
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
                atomicMax(&maxs[(size_t)held * 3 + c], hi[c]);
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
            const unsigned p = photon[i];
            for (int c = 0; c < 3; c++) v[c] = order_f32(src.coord(p, c));
        }
        // a step whose first and last lanes sit in the same range lies inside it entirely
        const int path0 = __shfl(part ? w.path : -1, 0), path63 = __shfl(part ? w.path : -2, 63);
        if (path0 == path63) {
            if (held != path0) flush();
            held = path0;
            for (int c = 0; c < 3; c++) { lo[c] = min(lo[c], v[c]); hi[c] = max(hi[c], v[c]); }
        } else {
            flush();
            if (part)
                for (int c = 0; c < 3; c++) {
                    atomicMin(&mins[(size_t)w.path * 3 + c], v[c]);
                    atomicMax(&maxs[(size_t)w.path * 3 + c], v[c]);
                }
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
        keys[i] = (unsigned long long)(unsigned)i << 32;
        return;
    }
    float best = -1.f;
    int ax = 0;
    for (int c = 0; c < 3; c++) {
        float span = unorder_f32(maxs[(size_t)w.path * 3 + c]) - unorder_f32(mins[(size_t)w.path * 3 + c]);
        if (span > best) { best = span; ax = c; }     // first maximum wins (Kdtree.cpp:14-22)
    }
    keys[i] = ((unsigned long long)(unsigned)w.lo << 32) | order_f32(src.coord(photon[i], ax));
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
