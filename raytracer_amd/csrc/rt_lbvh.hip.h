// Opt-in fast scene build on the device: a linear BVH over Morton-sorted primitives in place of the
// reference's host-built tree (BVH.cpp:57-96).
//
// The reference's split rule (sort by box centre, sum-of-extents cost over the middle third) is a
// serial O(n log^2 n) host algorithm and defines the tree the bit-exact parity is stated for.  This
// build produces a DIFFERENT tree in milliseconds.  Traversal of either tree returns the closest
// primitive, so the hits agree except where the reference's own box culling is decisive (a hit
// within rounding of a box face) and on exact ties; it is validated against exhaustive
// intersection, not against the goldens, and nothing uses it unless asked to.
//
//   1. box + centroid per primitive, scene bounds of the centroids (ordered-integer atomics)
//   2. 63-bit Morton key per primitive, radix sort of (key, primitive) pairs (hipCUB)
//   3. binary radix tree over the sorted keys (Karras 2012): every inner node covers a contiguous
//      range of the sorted primitives
//   4. boxes bottom-up: the second thread to reach a node merges its children's boxes
//   5. inner nodes covering <= 15 primitives become leaves (the leaf format of the flattened tree:
//      a range of the primitive array); the kept nodes are renumbered by a prefix sum and written as
//      Node<double> / Node<float> records, the primitives in sorted order
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "rt_flat.h"

namespace rtl {

constexpr uint32_t KLEAF = 0x80000000u;     // child reference of the radix tree: KLEAF | sorted primitive, or inner node
constexpr int LEAF_MAX = 15;                // BVH.h:45: the reference's leaves hold fewer than 16 objects

struct Box6 { double lo[3], hi[3]; };

__device__ __forceinline__ unsigned long long order_bits(double v)
{
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__host__ __device__ inline double unorder_bits(unsigned long long u)
{
    u = (u >> 63) ? (u & 0x7FFFFFFFFFFFFFFFull) : ~u;
    double d;
    __builtin_memcpy(&d, &u, 8);
    return d;
}

// box of a primitive record (triangle A, A + e1, A + e2; sphere centre +- radius), widened by a margin
// that covers the rounding of the intersection routines in either precision
__device__ __forceinline__ Box6 prim_box(const rtf::Prim<double>& p)
{
    Box6 b;
    for (int i = 0; i < 3; i++) {
        double lo, hi;
        if (p.kind == rtf::PRIM_TRI) {
            double a = p.a[i], v1 = p.a[i] + p.e1[i], v2 = p.a[i] + p.e2[i];
            lo = fmin(a, fmin(v1, v2));
            hi = fmax(a, fmax(v1, v2));
        } else {
            lo = p.a[i] - p.e1[0];
            hi = p.a[i] + p.e1[0];
        }
        double m = 1e-5 * fmax(1.0, fmax(fabs(lo), fabs(hi)));
        b.lo[i] = lo - m;
        b.hi[i] = hi + m;
    }
    return b;
}

__global__ void bounds_kernel(const rtf::Prim<double>* __restrict__ prims, long long n, Box6* __restrict__ boxes, unsigned long long* __restrict__ cb)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Box6 b = prim_box(prims[i]);
    boxes[i] = b;
    for (int c = 0; c < 3; c++) {
        double ctr = 0.5 * (b.lo[c] + b.hi[c]);
        atomicMin(&cb[c], order_bits(ctr));
        atomicMax(&cb[3 + c], order_bits(ctr));
    }
}

__device__ __forceinline__ unsigned long long spread21(unsigned long long x)
{
    x &= 0x1FFFFFull;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ void morton_kernel(const Box6* __restrict__ boxes, long long n, const unsigned long long* __restrict__ cb, unsigned long long* __restrict__ keys,
                              uint32_t* __restrict__ vals)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // one scale for the three axes (the longest extent of the centroid bounds): the cells of the key are
    // cubes, so a flat or elongated scene is not sliced along its short axis before its long ones
    double ext = 0;
    for (int c = 0; c < 3; c++) ext = fmax(ext, unorder_bits(cb[3 + c]) - unorder_bits(cb[c]));
    unsigned long long key = 0;
    for (int c = 0; c < 3; c++) {
        double lo = unorder_bits(cb[c]);
        double ctr = 0.5 * (boxes[i].lo[c] + boxes[i].hi[c]);
        double t = ext > 0 ? (ctr - lo) / ext : 0.0;
        t = fmin(fmax(t, 0.0), 1.0);
        unsigned long long q = (unsigned long long)(t * 2097151.0);
        key |= spread21(q) << c;
    }
    keys[i] = key;
    vals[i] = (uint32_t)i;
}

struct Radix {
    const unsigned long long* key;   // sorted
    long long n;
    __device__ int delta(long long i, long long j) const
    {
        if (j < 0 || j >= n) return -1;
        unsigned long long a = key[i], b = key[j];
        if (a == b) return 64 + __clz((unsigned)(i ^ j));     // equal keys: fall back to the position
        return __clzll((long long)(a ^ b));
    }
};

// Karras, "Maximizing parallelism in the construction of BVHs, octrees, and k-d trees" (2012), one thread per inner node
__global__ void radix_tree_kernel(Radix r, uint32_t* __restrict__ left, uint32_t* __restrict__ right, uint32_t* __restrict__ first, uint32_t* __restrict__ last,
                                  uint32_t* __restrict__ parent_inner, uint32_t* __restrict__ parent_leaf)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= r.n - 1) return;
    const int d = r.delta(i, i + 1) - r.delta(i, i - 1) > 0 ? 1 : -1;
    const int dmin = r.delta(i, i - d);
    long long lmax = 2;
    while (r.delta(i, i + lmax * d) > dmin) lmax *= 2;
    long long l = 0;
    for (long long t = lmax / 2; t >= 1; t /= 2)
        if (r.delta(i, i + (l + t) * d) > dmin) l += t;
    const long long j = i + l * d;
    const int dnode = r.delta(i, j);
    long long s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (r.delta(i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const long long gamma = i + s * d + (d < 0 ? -1 : 0);
    const long long lo = i < j ? i : j, hi = i < j ? j : i;
    const bool lleaf = lo == gamma, rleaf = hi == gamma + 1;
    left[i] = lleaf ? (KLEAF | (uint32_t)gamma) : (uint32_t)gamma;
    right[i] = rleaf ? (KLEAF | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
    first[i] = (uint32_t)lo;
    last[i] = (uint32_t)hi;
    if (lleaf) parent_leaf[gamma] = (uint32_t)i; else parent_inner[gamma] = (uint32_t)i;
    if (rleaf) parent_leaf[gamma + 1] = (uint32_t)i; else parent_inner[gamma + 1] = (uint32_t)i;
}

__device__ __forceinline__ Box6 load_box_coherent(const Box6* p)
{
    // written by another workgroup moments ago: read past this CU's L1
    Box6 b;
    for (int c = 0; c < 3; c++) {
        b.lo[c] = __hip_atomic_load(&p->lo[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b.hi[c] = __hip_atomic_load(&p->hi[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return b;
}
__device__ __forceinline__ void store_box_coherent(Box6* p, const Box6& b)
{
    for (int c = 0; c < 3; c++) {
        __hip_atomic_store(&p->lo[c], b.lo[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&p->hi[c], b.hi[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// one thread per sorted primitive climbs the tree; at each inner node the first arrival stops, the second
// (which knows both children are finished) merges their boxes and goes on
__global__ void refit_kernel(long long n, const uint32_t* __restrict__ vals, const Box6* __restrict__ prim_boxes, Box6* __restrict__ leaf_boxes,
                             Box6* __restrict__ node_boxes, const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
                             const uint32_t* __restrict__ parent_inner, const uint32_t* __restrict__ parent_leaf, unsigned* __restrict__ arrived)
{
    long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    Box6 b = prim_boxes[vals[g]];
    store_box_coherent(&leaf_boxes[g], b);
    if (n < 2) return;
    uint32_t node = parent_leaf[g];
    while (true) {
        __threadfence();
        if (atomicAdd(&arrived[node], 1u) == 0u) return;     // the sibling subtree is not finished yet
        __threadfence();
        const uint32_t l = left[node], r = right[node];
        Box6 bl = (l & KLEAF) ? load_box_coherent(&leaf_boxes[l & ~KLEAF]) : load_box_coherent(&node_boxes[l]);
        Box6 br = (r & KLEAF) ? load_box_coherent(&leaf_boxes[r & ~KLEAF]) : load_box_coherent(&node_boxes[r]);
        for (int c = 0; c < 3; c++) { b.lo[c] = fmin(bl.lo[c], br.lo[c]); b.hi[c] = fmax(bl.hi[c], br.hi[c]); }
        store_box_coherent(&node_boxes[node], b);
        if (node == 0) return;
        node = parent_inner[node];
    }
}

__global__ void keep_kernel(long long n_inner, const uint32_t* __restrict__ first, const uint32_t* __restrict__ last, uint32_t* __restrict__ keep)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_inner) keep[i] = (last[i] - first[i] + 1u > (uint32_t)LEAF_MAX) ? 1u : 0u;
}

struct EmitArgs {
    const uint32_t *left, *right, *first, *last, *keep, *newidx, *parent_inner;
    const Box6 *leaf_boxes, *node_boxes;
};

__device__ __forceinline__ uint32_t child_ref(const EmitArgs& a, uint32_t c, Box6& box)
{
    if (c & KLEAF) {
        const uint32_t g = c & ~KLEAF;
        box = a.leaf_boxes[g];
        return rtf::make_leaf(g, 1u);
    }
    box = a.node_boxes[c];
    if (a.keep[c]) return a.newidx[c];
    return rtf::make_leaf(a.first[c], a.last[c] - a.first[c] + 1u);
}

__global__ void emit_nodes_kernel(long long n_inner, EmitArgs a, rtf::Node<double>* __restrict__ out64, rtf::Node<float>* __restrict__ out32, int* __restrict__ depth)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inner || !a.keep[i]) return;
    Box6 b0, b1;
    rtf::Node<double> nd;
    nd.c0 = child_ref(a, a.left[i], b0);
    nd.c1 = child_ref(a, a.right[i], b1);
    rtf::Node<float> nf;
    for (int c = 0; c < 3; c++) {
        nd.lo0[c] = b0.lo[c]; nd.hi0[c] = b0.hi[c]; nd.lo1[c] = b1.lo[c]; nd.hi1[c] = b1.hi[c];
        nf.lo0[c] = (float)b0.lo[c]; nf.hi0[c] = (float)b0.hi[c]; nf.lo1[c] = (float)b1.lo[c]; nf.hi1[c] = (float)b1.hi[c];
    }
    nf.c0 = nd.c0;
    nf.c1 = nd.c1;
    const uint32_t me = a.newidx[i];
    out64[me] = nd;
    out32[me] = nf;
    int d = 1;                               // levels from the root down to this node's children
    for (uint32_t p = (uint32_t)i; p != 0; p = a.parent_inner[p]) d++;
    atomicMax(depth, d);
}

__global__ void emit_prims_kernel(long long n, const uint32_t* __restrict__ vals, const rtf::Prim<double>* __restrict__ in, rtf::Prim<double>* __restrict__ out64,
                                  rtf::Prim<float>* __restrict__ out32)
{
    long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    rtf::Prim<double> p = in[vals[g]];
    out64[g] = p;
    rtf::Prim<float> q;
    for (int c = 0; c < 3; c++) { q.a[c] = (float)p.a[c]; q.e1[c] = (float)p.e1[c]; q.e2[c] = (float)p.e2[c]; }
    q.id = p.id;
    q.kind = p.kind;
    out32[g] = q;
}

}  // namespace rtl
