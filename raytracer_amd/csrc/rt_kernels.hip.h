// HIP kernels of the closest-hit path (gfx950, wave64).  Included by rt_device.hip only.
//
// K2  closest-hit traversal = BVH::Shoot (BVH.cpp:108-179) + the primitive tests, one ray per lane,
//     one 64-ray chunk / 8x8 pixel tile per wave (the hardware dispatcher places the blocks; there
//     is no shared work counter), per-lane traversal stack in LDS laid out [depth][lane] (bank =
//     lane, conflict free).
// K1  primary-ray generation (main.cpp:753-790) is fused into the primary-ray kernel.
#pragma once
#include <hip/hip_runtime.h>

#include "rt_flat.h"
#include "rt_math.h"

namespace rtk {

constexpr int WAVE = 64;
constexpr int BLOCK = 256;
constexpr int WAVES_PER_BLOCK = BLOCK / WAVE;
constexpr int MAX_BLOCK = 768;   // register budget of the traversal kernels: compiled for 3 waves per SIMD (<= 168 VGPRs)

// How a traversal kernel keeps its nodes and its stack:
//   TRAV_PLAIN    nodes read through L1; stack entries (child ref, entry distance)
//   TRAV_NODES    small trees: node array copied into the block's LDS; same stack
//   TRAV_RETEST   deep trees: stack entries are 4 bytes (parent node, which child) and the deferred
//                 child's box is tested again when it is popped -- the reference's own entry
//                 re-test (BVH.cpp:132) -- so a third of the LDS per wave buys three times the
//                 resident waves where the stack depth limits occupancy
//   TRAV_SCENE    small scenes, persistent kernels: node array AND primitive records copied into the block's
//                 LDS once (one block per CU); rays that diverge fetch a different primitive record per lane,
//                 which LDS serves at 256 B per clock where the L1's data return path serves 64
constexpr int TRAV_PLAIN = 0, TRAV_NODES = 1, TRAV_RETEST = 2, TRAV_SCENE = 3;
constexpr bool nodes_in_lds(int mode) { return mode == TRAV_NODES || mode == TRAV_SCENE; }

// 16 bytes read from LDS by address space, so the read is a ds_read_b128 whatever the compiler can prove about the pointer
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u4v lds_u4;

struct LaneStack {
    uint32_t* ref;   // [depth][64] for this wave
    void* t0;        // [depth][64] of R
    const uint4* lds_nodes;   // node image in LDS (TRAV_NODES / TRAV_SCENE kernels), else unused
    const uint4* lds_prims;   // primitive records in LDS (TRAV_SCENE kernels), else unused
    int lane;
    int cap;
};

// Inner nodes staged in LDS (small trees): only the payload of a node is kept (float 56 -> 64 B read as
// 4 chunks, double 104 -> 112 B as 7 chunks of 16 B) at a stride of an ODD number of 16-byte slots
// (5 / 7), so that lanes reading different nodes spread over all 16 slots of the 256-byte bank row.
template <class R> struct NodeLds;
template <> struct NodeLds<float> { static constexpr int CHUNKS = 4, STRIDE = 5; };
template <> struct NodeLds<double> { static constexpr int CHUNKS = 7, STRIDE = 7; };

template <class R>
__host__ __device__ inline size_t node_lds_bytes(size_t n_nodes) { return n_nodes * NodeLds<R>::STRIDE * 16; }

template <class R>
__device__ __forceinline__ void stage_nodes(uint4* dst, const rtf::Node<R>* nodes, int n_nodes)
{
    constexpr int CH = NodeLds<R>::CHUNKS, ST = NodeLds<R>::STRIDE, SRC = sizeof(rtf::Node<R>) / 16;
    const uint4* src = reinterpret_cast<const uint4*>(nodes);
    for (int i = threadIdx.x; i < n_nodes * CH; i += blockDim.x) {
        int n = i / CH, c = i - n * CH;
        dst[n * ST + c] = src[n * SRC + c];
    }
    __syncthreads();
}

// Primitive records staged in LDS as they are (float 48 B, double 80 B: 3 / 5 slots of 16 bytes, an odd number, so
// lanes reading different records spread over the 16 slots of a bank row).
template <class R>
__host__ __device__ inline size_t prim_lds_bytes(size_t n_prims) { return n_prims * sizeof(rtf::Prim<R>); }

template <class R>
__device__ __forceinline__ void stage_prims(uint4* dst, const rtf::Prim<R>* prims, int n_prims)
{
    const uint4* src = reinterpret_cast<const uint4*>(prims);
    const int n16 = n_prims * (int)(sizeof(rtf::Prim<R>) / 16);
    for (int i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}

template <class R, int MODE>
__device__ __forceinline__ rtf::Node<R> fetch_node(const rtf::Node<R>* __restrict__ nodes, const uint4* lds_nodes, uint32_t idx)
{
    constexpr int CH = NodeLds<R>::CHUNKS, ST = NodeLds<R>::STRIDE;
    uint4 v[sizeof(rtf::Node<R>) / 16];
    if (nodes_in_lds(MODE)) {
        const lds_u4* src = (const lds_u4*)(lds_nodes + (size_t)idx * ST);
#pragma unroll
        for (int i = 0; i < CH; i++) { const u4v t = src[i]; v[i] = uint4{t.x, t.y, t.z, t.w}; }
    } else {
        const uint4* src = reinterpret_cast<const uint4*>(nodes + idx);
#pragma unroll
        for (int i = 0; i < CH; i++) v[i] = src[i];
    }
    rtf::Node<R> out;
    __builtin_memcpy(&out, v, CH * 16);
    return out;
}

template <class R> __device__ __forceinline__ bool rt_finite(R v);
template <> __device__ __forceinline__ bool rt_finite<float>(float v) { return __builtin_isfinite(v); }
template <> __device__ __forceinline__ bool rt_finite<double>(double v) { return __builtin_isfinite(v); }

struct WorkCount { unsigned long long inner = 0, prims = 0; };

// whole primitive record in 16-byte pieces issued back to back (one round trip instead of one per field)
template <class R, bool LDS = false>
__device__ __forceinline__ rtf::Prim<R> load_prim(const rtf::Prim<R>* p)
{
    constexpr int N = sizeof(rtf::Prim<R>) / 16;
    static_assert(sizeof(rtf::Prim<R>) % 16 == 0, "primitive record is a multiple of 16 bytes");
    uint4 v[N];
    if (LDS) {
        const lds_u4* src = (const lds_u4*)p;
#pragma unroll
        for (int i = 0; i < N; i++) { const u4v t = src[i]; v[i] = uint4{t.x, t.y, t.z, t.w}; }
    } else {
        const uint4* src = reinterpret_cast<const uint4*>(p);
#pragma unroll
        for (int i = 0; i < N; i++) v[i] = src[i];
    }
    rtf::Prim<R> out;
    __builtin_memcpy(&out, v, sizeof(out));
    return out;
}

template <class R, bool TRI_ONLY>
__device__ __forceinline__ void test_prim(const rtf::Prim<R>& p, const R* o, const R* d, R& dist, int& hit)
{
    R t;
    if (TRI_ONLY) t = rtm::tri_intersect_staged<R>(p.a, p.e1, p.e2, o, d);
    else t = p.kind == rtf::PRIM_TRI ? rtm::tri_intersect_staged<R>(p.a, p.e1, p.e2, o, d) : rtm::prim_intersect<R>(p, o, d);
#ifdef RT_TRI_TWICE
    // (measurement build: the triangle test's arithmetic a second time on operands the compiler cannot see to be the same,
    // its result kept alive -- what the kernel's time owes to that arithmetic is the difference to the plain build)
    if (TRI_ONLY && sizeof(R) == 4) {
        R o2[3] = {o[0], o[1], o[2]};
        asm volatile("" : "+v"(o2[0]), "+v"(o2[1]), "+v"(o2[2]));
        const R t2 = rtm::tri_intersect_staged<R>(p.a, p.e1, p.e2, o2, d);
        if (t2 != t && t2 == t2 && t == t) hit = -3;
    }
#endif
    if (t >= R(0) && t < dist) { dist = t; hit = (int)p.id; }
}

// Leaf: test every primitive in leaf order, accept iff t >= 0 && t < dist (BVH.cpp:135-149).
// The next record is fetched while the current one is tested; two register sets take turns so that
// no record is ever copied from one set to the other.
template <class R, bool COUNT, bool TRI_ONLY, bool LDS = false>
__device__ __forceinline__ void test_leaf(const rtf::Prim<R>* __restrict__ prims, uint32_t leaf, const R* o, const R* d, R& dist, int& hit,
                                          WorkCount* wc)
{
    const rtf::Prim<R>* p = prims + (leaf & rtf::LEAF_FIRST_MASK);
    const uint32_t cnt = (leaf >> rtf::LEAF_COUNT_SHIFT) & 15u;
    if (COUNT) wc->prims += cnt;
    if (cnt == 0) return;
    // the fetches are unconditional (index clamped to the last record) so that they stay in flight
    // across the test instead of being waited for at the join of a divergent branch
    const uint32_t last = cnt - 1;
    rtf::Prim<R> even = load_prim<R, LDS>(p), odd;
    for (uint32_t k = 0;; k += 2) {
        odd = load_prim<R, LDS>(p + (k + 1 < last ? k + 1 : last));
        test_prim<R, TRI_ONLY>(even, o, d, dist, hit);
        if (k + 1 > last) break;
        even = load_prim<R, LDS>(p + (k + 2 < last ? k + 2 : last));
        test_prim<R, TRI_ONLY>(odd, o, d, dist, hit);
        if (k + 2 > last) break;
    }
}

// BVH::Shoot for one ray.  `dist` is in/out exactly like the reference's `distance`.
// Visiting order, tie handling and culling are the reference's:
//   - both child boxes are tested against the current best (BVH.cpp:155-156), the child with the
//     smaller entry distance is entered first (child 1 on ties, BVH.cpp:159-176);
//   - the reference re-tests a child's box when it enters it.  For the child entered first nothing
//     has changed, so the re-test is skipped; for the deferred child the only thing that can have
//     changed is `dist`, and for finite operands the re-test then fails iff t0 > dist, which is
//     what the pop compares (t0 travels on the stack);
//   - rays with a non-finite reciprocal direction or origin (axis-parallel rays: 0 * inf = NaN in
//     the slab test) take the literal path below, which re-runs the full slab test on entry.
// Entry distance of a deferred child as kept on the stack: a float not above the value.  The pop test
// "entry distance > current best" skips a subtree; with the distance rounded DOWN the test can only fail to
// skip, and a subtree entered needlessly holds no hit closer than the best (its hits lie beyond its entry
// distance), so results are unchanged while a stack entry takes 8 bytes instead of 12.
template <class R>
__device__ __forceinline__ float entry_floor(R t)
{
    float f = (float)t;
    if ((R)f > t) f = __uint_as_float(__float_as_uint(f) + (f > 0.f ? -1 : 1));     // next float toward -inf (f != 0 here: f > t >= 0 or t < 0)
    return f;
}

// any_hit: the caller only asks WHETHER something lies within `dist` (an occlusion query): the walk ends at the first
// primitive found.  The same boxes are entered in the same order and the same primitive tests decide, so "something
// was hit" is the answer the full walk gives; which primitive, and how far, is then not the closest one.
template <class R, bool COUNT = false, bool TRI_ONLY = false, int MODE = TRAV_PLAIN>
__device__ int trace_ray(const rtf::SceneView<R>& sv, const R* o, const R* d, R& dist, const LaneStack& st, uint32_t* overflow,
                         WorkCount* wc = nullptr, bool any_hit = false)
{
    if (d[0] != d[0]) return -1;   // BVH.cpp:110-115
    int hit = -1;
    uint32_t* s_ref = st.ref;
    float* s_t0 = (float*)st.t0;
    const int lane = st.lane;
    constexpr bool PRIMS_LDS = MODE == TRAV_SCENE;
    const rtf::Prim<R>* const prims = PRIMS_LDS ? reinterpret_cast<const rtf::Prim<R>*>(st.lds_prims) : sv.prims;

    if (sv.root_ref != rtf::NO_NODE) {
        R inv[3] = {R(1) / d[0], R(1) / d[1], R(1) / d[2]};
        bool finite = rt_finite<R>(inv[0]) && rt_finite<R>(inv[1]) && rt_finite<R>(inv[2]) && rt_finite<R>(o[0]) && rt_finite<R>(o[1]) &&
                      rt_finite<R>(o[2]) && rt_finite<R>(dist);
        if (finite) {
            if (!(rtm::slab_finite<R>(sv.root_lo, sv.root_hi, o, inv, dist) < R(0))) {   // BVH.cpp:121,132
                uint32_t cur = sv.root_ref;
                int sp = 0;
                while (true) {
                    if (!(cur & rtf::LEAF_BIT)) {
                        const rtf::Node<R> nd = fetch_node<R, MODE>(sv.nodes, st.lds_nodes, cur);
                        if (COUNT) wc->inner++;
                        R td0 = rtm::slab_finite<R>(nd.lo0, nd.hi0, o, inv, dist);
                        R td1 = rtm::slab_finite<R>(nd.lo1, nd.hi1, o, inv, dist);
                        uint32_t c0 = nd.c0, c1 = nd.c1;
                        bool h0 = !(td0 < R(0)), h1 = !(td1 < R(0));
                        if (h0 && h1) {
                            bool first0 = td0 < td1;
                            if (sp < st.cap) {
                                if (MODE == TRAV_RETEST) {
                                    s_ref[sp * WAVE + lane] = (cur << 1) | (first0 ? 1u : 0u);
                                } else {
                                    s_ref[sp * WAVE + lane] = first0 ? c1 : c0;
                                    s_t0[sp * WAVE + lane] = entry_floor<R>(first0 ? td1 : td0);
                                }
                                sp++;
                            } else {
                                *overflow = 1;
                            }
                            cur = first0 ? c0 : c1;
                            continue;
                        }
                        if (h0) { cur = c0; continue; }
                        if (h1) { cur = c1; continue; }
                    } else {
                        test_leaf<R, COUNT, TRI_ONLY, PRIMS_LDS>(prims, cur, o, d, dist, hit, wc);
                        if (any_hit && hit >= 0) break;
                    }
                    bool found = false;
                    while (sp > 0) {
                        sp--;
                        if (MODE == TRAV_RETEST) {
                            const uint32_t e = s_ref[sp * WAVE + lane];
                            const rtf::Node<R>& pn = sv.nodes[e >> 1];
                            const bool second = e & 1u;
                            R lo[3], hi[3];
                            for (int i = 0; i < 3; i++) { lo[i] = second ? pn.lo1[i] : pn.lo0[i]; hi[i] = second ? pn.hi1[i] : pn.hi0[i]; }
                            if (!(rtm::slab_finite<R>(lo, hi, o, inv, dist) < R(0))) { cur = second ? pn.c1 : pn.c0; found = true; break; }   // BVH.cpp:132
                        } else {
                            const R t0 = (R)s_t0[sp * WAVE + lane];
                            const uint32_t ref = s_ref[sp * WAVE + lane];     // read with t0, not after the comparison
                            if (!(t0 > dist)) { cur = ref; found = true; break; }
                        }
                    }
                    if (!found) break;
                }
            }
        } else {
            // literal emulation of the recursion; stack entries are (parent node << 1 | which child).  At an inner
            // node the second child waits on the stack and the first is entered at once, so a path of `depth`
            // inner nodes holds at most `depth` entries, as in the walk above.
            if (!(rtm::slab_literal<R>(sv.root_lo, sv.root_hi, o, d, dist) < R(0))) {
                int sp = 0;
                uint32_t cur = sv.root_ref;
                bool have = true;
                while (true) {
                    uint32_t e = 0;
                    bool direct = false;
                    if (have) {
                        if (cur & rtf::LEAF_BIT) {
                            test_leaf<R, COUNT, TRI_ONLY, PRIMS_LDS>(prims, cur, o, d, dist, hit, wc);
                            if (any_hit && hit >= 0) break;
                        } else {
                            const rtf::Node<R>& nd = sv.nodes[cur];
                            if (COUNT) wc->inner++;
                            R td0 = rtm::slab_literal<R>(nd.lo0, nd.hi0, o, d, dist);
                            R td1 = rtm::slab_literal<R>(nd.lo1, nd.hi1, o, d, dist);
                            uint32_t first = td0 < td1 ? 0u : 1u;
                            if (sp < st.cap) {
                                s_ref[sp * WAVE + lane] = (cur << 1) | (1u - first);
                                sp++;
                            } else {
                                *overflow = 1;
                            }
                            e = (cur << 1) | first;
                            direct = true;
                        }
                    }
                    if (!direct) {
                        if (sp == 0) break;
                        sp--;
                        e = s_ref[sp * WAVE + lane];
                    }
                    const rtf::Node<R>& pn = sv.nodes[e >> 1];
                    bool second = e & 1u;
                    const R* lo = second ? pn.lo1 : pn.lo0;
                    const R* hi = second ? pn.hi1 : pn.hi0;
                    have = !(rtm::slab_literal<R>(lo, hi, o, d, dist) < R(0));   // entry test, BVH.cpp:132
                    cur = second ? pn.c1 : pn.c0;
                }
            }
        }
    }
    // planes: side list, same accept rule (no counterpart in the reference, where Plane is abstract)
    for (int k = 0; k < sv.n_planes && !(any_hit && hit >= 0); k++) {
        const rtf::Prim<R>& p = sv.planes[k];
        R t = rtm::plane_intersect<R>(p.a, p.e1[0], o, d);
        if (t >= R(0) && t < dist) { dist = t; hit = (int)p.id; }
    }
    return hit;
}

// LDS layout of a block: [node image (node_bytes, may be 0)] [primitive records (prim_bytes, may be 0)]
// [refs: waves x cap x 64 x 4 B] [t0: waves x cap x 64 x 4 B]
template <class R>
__device__ __forceinline__ LaneStack make_stack(unsigned char* lds, int cap, size_t node_bytes = 0, size_t prim_bytes = 0)
{
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int waves = blockDim.x >> 6;
    LaneStack st;
    st.lds_nodes = reinterpret_cast<const uint4*>(lds);
    st.lds_prims = reinterpret_cast<const uint4*>(lds + node_bytes);
    unsigned char* base = lds + node_bytes + prim_bytes;
    st.ref = (uint32_t*)base + (size_t)wave * cap * WAVE;
    st.t0 = (void*)((float*)(base + (size_t)waves * cap * WAVE * sizeof(uint32_t)) + (size_t)wave * cap * WAVE);
    st.lane = lane;
    st.cap = cap;
    return st;
}

// real_size = 0 for TRAV_RETEST (no entry distances)
inline size_t stack_lds_bytes(int cap, size_t real_size, int waves = WAVES_PER_BLOCK) { return (size_t)waves * cap * WAVE * (sizeof(uint32_t) + real_size); }

// Overflow latch of a context: two words -- [0] traversal stack, [1] bounce / gather / photon queue -- at a fixed
// offset behind the per-call counter block every kernel of the context gets (`counters` below is that block's base).
// Kernels raise it next to the per-call counters; no launch clears it, so calls that do not wait for their
// kernels (no stats requested) still get their overflow reported, by the next rt_context_synchronize of THAT context
// (contexts sharing a device do not see each other's).
constexpr size_t LATCH_OFFSET = 64 * 1024;
__device__ __forceinline__ void latch_overflow(void* counters, int which)
{
    atomicMax(reinterpret_cast<unsigned long long*>(static_cast<char*>(counters) + LATCH_OFFSET) + which, 1ull);
}

// next: unused (kept for layout); overflow: stack-overflow flag; shadow: shadow rays traced; inner / prims: work of the counting build
struct TraceCounters { unsigned long long next, overflow, shadow, inner, prims, pad[3]; };

// Batch form of BVH::Shoot: rays in SoA doubles (x[n], y[n], z[n]).
template <class R, bool TRI_ONLY, int MODE>
__global__ __launch_bounds__(MAX_BLOCK) void trace_batch_kernel(rtf::SceneView<R> sv, long long n, const double* __restrict__ org,
                                                                const double* __restrict__ dir, double* __restrict__ dist_io,
                                                                int* __restrict__ ids, TraceCounters* ctr, int cap, int n_nodes)
{
    extern __shared__ __align__(16) unsigned char lds[];
    const size_t node_bytes = MODE == TRAV_NODES ? node_lds_bytes<R>((size_t)n_nodes) : 0;
    if (MODE == TRAV_NODES) stage_nodes<R>(reinterpret_cast<uint4*>(lds), sv.nodes, n_nodes);
    LaneStack st = make_stack<R>(lds, cap, node_bytes);
    uint32_t overflow = 0;
    // one chunk of 64 rays per wave; the dispatcher balances the load
    long long i = ((long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * WAVE + st.lane;
    if (i < n) {
        R o[3] = {(R)org[i], (R)org[n + i], (R)org[2 * n + i]};
        R d[3] = {(R)dir[i], (R)dir[n + i], (R)dir[2 * n + i]};
        R dist = (R)dist_io[i];
        int id = trace_ray<R, false, TRI_ONLY, MODE>(sv, o, d, dist, st, &overflow);
        ids[i] = id;
        if (id >= 0) dist_io[i] = (double)dist;
    }
    if (overflow) { atomicMax(&ctr->overflow, 1ull); latch_overflow(ctr, 0); }
}

// shadow == 2 and no shadow-id buffer: an occluded hit gets this bit set in its id (ids stay below 2^27, see rt_flat.h)
constexpr int SHADOWED_BIT = 1 << 30;
struct PrimaryArgs {
    double cam_pos[3], cam_dir[3], cam_right[3], cam_down[3];
    double light[3];
    int width, height, row_begin, row_end;
    int shadow;
    int band_rows, n_parts, part;   // row interleave (band_rows == 0: contiguous)
    int strata;                     // compact tile order: strata an XCD's sequence of super-blocks is taken from in turn (tile_of_wave)
};

// Which 8x8 pixel tile a wave of the pixel kernels takes; two orders, both measured.
//  * compact (scenes whose node image sits in LDS): the frame is cut into super-blocks of 16 x 8 tiles
//    (128 x 64 pixels) walked in Morton order, so the waves of a block look at neighbouring pixels and share
//    triangle records in L1.  Workgroups are handed to the 8 XCDs round-robin (block b runs on XCD b % 8, each
//    with its own L2): XCD x takes super-blocks x, x + 8, ... and finishes each before the next.  3-5 % faster
//    than rows of tiles on the teapot frames.
//    How long a launch takes depends on WHICH tiles are resident together, by more than one would think: with `strata`
//    = 2 (tuning aid RT_TILE_STRATA) an XCD's sequence is taken from its upper and its lower half in turn, so that the
//    frame's cheap part (ceiling, far wall) and its heavy part (objects, floor) are under way at the same time.  The
//    headline frame in double then takes 0.283 instead of 0.301 ms (vector units 73 % busy instead of 67 %) -- and
//    the same frame without the occlusion queries 0.151 instead of 0.128, with closest-hit shadow rays 0.367 instead of
//    0.335, the other teapot mesh 0.311 instead of 0.291, while a camera near the floor gains 6-15 % in every form
//    (tools/strata_sweep.py: 36 combinations of mesh, camera, shadow query and precision, -15 % to +27 %).  No rule
//    that holds across frames came out of it (4, 8, 16 strata, the halves dealt out 4 to 32 tiles at a time, either
//    half walked upward, two tiles per wave: each better on some frames and worse on most), so the default stays the
//    plain sequence.
//  * rows of tiles (large scenes): consecutive waves take consecutive tiles of a row.  On the 10 M-triangle grid
//    the compact order lowers L1/L2 misses but leaves fewer waves resident on average (3.0 instead of 3.6 per
//    SIMD) and the frame takes 9 % longer, so such scenes keep this order.
constexpr int SB_TX = 16, SB_TY = 8, SB_TILES = SB_TX * SB_TY, N_XCD = 8;
// (`split` consecutive waves share a tile: wave `*part` of them takes that share of the tile's samples)
__host__ __device__ inline long long tile_grid_blocks(int tiles_x, int tiles_y, int waves_per_block, bool compact, int split = 1, int strata = 1)
{
    if (!compact) return ((long long)tiles_x * tiles_y * split + waves_per_block - 1) / waves_per_block;
    const long long n_sb = (long long)((tiles_x + SB_TX - 1) / SB_TX) * ((tiles_y + SB_TY - 1) / SB_TY);
    const long long n_seq = ((n_sb + N_XCD - 1) / N_XCD + strata - 1) / strata * strata;      // super-blocks (with padding) in one XCD's sequence
    const long long per_xcd = n_seq * SB_TILES * split;                                        // waves one XCD runs
    return (per_xcd + waves_per_block - 1) / waves_per_block * N_XCD;
}
template <bool COMPACT>
__device__ __forceinline__ bool tile_of_wave(int tiles_x, int tiles_y, int* tx, int* ty, unsigned split = 1, int* part = nullptr, unsigned strata = 1, unsigned bid = blockIdx.x)
{
    const unsigned wpb = blockDim.x >> 6;
    if (!COMPACT) {
        const unsigned long long w = (unsigned long long)bid * wpb + (threadIdx.x >> 6);
        const unsigned long long tile = w / split;
        if (part) *part = (int)(w % split);
        if (tile >= (unsigned long long)tiles_x * tiles_y) return false;
        *tx = (int)(tile % (unsigned)tiles_x);
        *ty = (int)(tile / (unsigned)tiles_x);
        return true;
    }
    const unsigned xcd = bid % N_XCD, slot = bid / N_XCD;
    const unsigned long long w = (unsigned long long)slot * wpb + (threadIdx.x >> 6);
    const unsigned long long q = w / split;
    if (part) *part = (int)(w % split);
    const unsigned sbx = (unsigned)(tiles_x + SB_TX - 1) / SB_TX, sby = (unsigned)(tiles_y + SB_TY - 1) / SB_TY;
    unsigned long long seq = q / SB_TILES;
    if (strata > 1) {
        const unsigned long long per = (((unsigned long long)sbx * sby + N_XCD - 1) / N_XCD + strata - 1) / strata;      // places in a stratum
        if (seq >= per * strata) return false;
        seq = (seq % strata) * per + seq / strata;
    }
    const unsigned long long sb = seq * N_XCD + xcd;
    const unsigned m = (unsigned)(q % SB_TILES);
    if (sb >= (unsigned long long)sbx * sby) return false;
    // m = x3 y2 x2 y1 x1 y0 x0
    const unsigned mx = (m & 1) | ((m >> 1) & 2) | ((m >> 2) & 4) | ((m >> 3) & 8);
    const unsigned my = ((m >> 1) & 1) | ((m >> 2) & 2) | ((m >> 3) & 4);
    *tx = (int)((unsigned)(sb % sbx) * SB_TX + mx);
    *ty = (int)((unsigned)(sb / sbx) * SB_TY + my);
    return *tx < tiles_x && *ty < tiles_y;
}

// Pixel loop front end: camera ray (always generated in double, then rounded to R) + BVH::Shoot
// with distance 1e9 (main.cpp:812-813); optional shadow ray toward the light.
// Work unit = 8x8 pixel tile per wave.
// ROUNDS = 2: a wave takes two tiles, consecutive places of its XCD's sequence, one after the other (unrolled: a loop around
// the traversal costs some forty registers) -- half as many blocks stage the node image.  One of the orders the launch
// code measures for a view in single precision, where it wins a few views by 8-15 % and loses most by more; in double it
// costs a wave per SIMD (120 registers) and never won.
template <class R, bool COUNT, bool TRI_ONLY, int MODE, int ROUNDS = 1>
__global__ __launch_bounds__(MAX_BLOCK) void trace_primary_kernel(rtf::SceneView<R> sv, PrimaryArgs a, int* __restrict__ ids,
                                                                  double* __restrict__ t_out, int* __restrict__ shadow_ids,
                                                                  TraceCounters* ctr, int cap, int n_nodes)
{
    extern __shared__ __align__(16) unsigned char lds[];
    const size_t node_bytes = MODE == TRAV_NODES ? node_lds_bytes<R>((size_t)n_nodes) : 0;
    if (MODE == TRAV_NODES) stage_nodes<R>(reinterpret_cast<uint4*>(lds), sv.nodes, n_nodes);
    LaneStack st = make_stack<R>(lds, cap, node_bytes);
    uint32_t overflow = 0;
    unsigned long long n_shadow = 0;
    WorkCount wc;
    const int rows = a.row_end - a.row_begin;
    const int tiles_x = (a.width + 7) >> 3, tiles_y = (rows + 7) >> 3;
    // one 8x8 tile per wave, no work counter: the hardware dispatcher balances the load by starting
    // blocks as earlier ones retire (a shared counter serves only ~90 requests per microsecond
    // chip-wide, which is fewer than the tiles this kernel retires)
    int tx, ty;
#pragma unroll
    for (int round = 0; round < ROUNDS; round++)
    if (tile_of_wave<MODE == TRAV_NODES>(tiles_x, tiles_y, &tx, &ty, 1, nullptr, (unsigned)a.strata, ROUNDS == 1 ? blockIdx.x : ((blockIdx.x / N_XCD) * ROUNDS + round) * N_XCD + blockIdx.x % N_XCD)) {
        int x = tx * 8 + (st.lane & 7), yl = ty * 8 + (st.lane >> 3);
        if (x < a.width && yl < rows) {
            int yc = a.row_begin + yl;   // row in this call's (compact) row space
            int y = a.band_rows > 0 ? ((yc / a.band_rows) * a.n_parts + a.part) * a.band_rows + yc % a.band_rows : yc;
            rtm::D3 dd = rtm::camera_ray_dir(a.cam_dir, a.cam_right, a.cam_down, a.width, a.height, 1, x, y, 0);
            R o[3] = {(R)a.cam_pos[0], (R)a.cam_pos[1], (R)a.cam_pos[2]};
            R d[3] = {(R)dd.x, (R)dd.y, (R)dd.z};
            R dist = R(1e9);
#ifdef RT_TILE_END_TIMES
            const unsigned long long c_begin = clock64();
#endif
            int id = trace_ray<R, COUNT, TRI_ONLY, MODE>(sv, o, d, dist, st, &overflow, &wc);
#ifdef RT_TILE_END_TIMES
            __builtin_amdgcn_wave_barrier();
            const unsigned long long c_mid = clock64();
#endif
            long long px = (long long)yl * a.width + x;
            ids[px] = id;
            if (t_out) t_out[px] = (double)dist;
            if (a.shadow) {
                int sid = -2;
                if (id >= 0) {
                    // occlusion query toward the light through the same BVH::Shoot semantics:
                    // P = o + d * t; L = normalize(light - P); origin = P + L * 1e-4; range = |light - P| - 1e-4
                    double t = (double)dist;
                    rtm::D3 P = rtm::vadd(rtm::D3{a.cam_pos[0], a.cam_pos[1], a.cam_pos[2]}, rtm::vmul(dd, t));
                    rtm::D3 Lv = rtm::vsub(rtm::D3{a.light[0], a.light[1], a.light[2]}, P);
                    double len = sqrt(rtm::vdot(Lv, Lv));
                    rtm::D3 L = {Lv.x / len, Lv.y / len, Lv.z / len};
                    rtm::D3 so = rtm::vadd(P, rtm::vmul(L, 1e-4));
                    R o2[3] = {(R)so.x, (R)so.y, (R)so.z};
                    R d2[3] = {(R)L.x, (R)L.y, (R)L.z};
                    R range = (R)(len - 1e-4);
                    // (the folded form reports occlusion only: the walk may stop at the first occluder, and it may run either way)
                    const bool any = a.shadow == 2 && !shadow_ids;
                    if (any) {
                        // the same segment walked from the light: the rays of a tile then leave one point side by side, like its
                        // camera rays, and meet what stands near the light -- where the occluders of most pixels are -- first
                        o2[0] = (R)a.light[0]; o2[1] = (R)a.light[1]; o2[2] = (R)a.light[2];
                        d2[0] = (R)(-L.x); d2[1] = (R)(-L.y); d2[2] = (R)(-L.z);
                    }
                    sid = trace_ray<R, COUNT, TRI_ONLY, MODE>(sv, o2, d2, range, st, &overflow, &wc, any);
                    n_shadow++;
                }
                if (shadow_ids) shadow_ids[px] = sid;
                else if (a.shadow == 2 && sid >= 0) ids[px] = id | SHADOWED_BIT;      // folded form: one word per pixel
            }
#ifdef RT_TILE_END_TIMES
            // (measurement build: when the pixel's wave was done; with rt_render_params.part = -7, instead, the cycles of the
            // wave's camera phase in the even columns and of its shadow phase in the odd ones)
            if (t_out) t_out[px] = a.part == -7 ? (double)((x & 1) ? clock64() - c_mid : c_mid - c_begin) : (double)wall_clock64();
#endif
        }
    }
    if (overflow) { atomicMax(&ctr->overflow, 1ull); latch_overflow(ctr, 0); }
    if (a.shadow) {
        // one atomic per wave
        for (int off = 32; off > 0; off >>= 1) n_shadow += __shfl_down(n_shadow, off);
        if (st.lane == 0 && n_shadow) atomicAdd(&ctr->shadow, n_shadow);
    }
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) { wc.inner += __shfl_down(wc.inner, off); wc.prims += __shfl_down(wc.prims, off); }
        if (st.lane == 0) { atomicAdd(&ctr->inner, wc.inner); atomicAdd(&ctr->prims, wc.prims); }
    }
}

}  // namespace rtk
