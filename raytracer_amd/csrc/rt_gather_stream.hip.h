// K4, the form the pixel path uses: the photon gather for a GROUP of up to 64 neighbouring requests per wave, one
// request per lane (KDTree::findKNN, Kdtree.cpp:73-158, + the estimate of main.cpp:260-295), over a PACKED list of
// the group's photons.
//
// The requests one shading wave queues are hit points a few pixel footprints apart; their k-nearest balls overlap
// almost entirely.  A wave takes such a window and
//   1. picks the middle request as the round's centre and finds its k-th nearest distance r_c with all 64 lanes
//      working on that one request: the tree is walked for a box of a guessed radius (the wave's last r_c, grown
//      or shrunk until it holds k photons and fits the lists), the photons of the listed leaves that lie within
//      reach of the centre are written, in tree order, to the wave's packed list in global memory, and four
//      radix passes over that list (lane j measuring photon j) give the exact bit pattern of the k-th smallest
//      squared distance;
//   2. derives every other request's search range from r_c (triangle inequality, narrowed to r_c (1 -+ rho) and
//      checked by counting), and filters the packed list in place down to the photons inside the box around the
//      round's balls and within reach of the centre -- a list that depends on the window alone, not on the guess,
//      so the sums below run in the same order whichever wave takes the window;
//   3. streams the packed list three times, every lane measuring its own distance to the same photon (a step's
//      64 photons are staged in LDS and read back as broadcasts, four at a time):
//        a. a histogram per lane over the shell between its two radii locates the bin of its k-th distance,
//        b. the distances of that bin (a few dozen) are kept and ranked: the exact k-th distance and its ties,
//        c. the radiance estimate is accumulated over the set, which is never stored.
// A group lists the photons of every leaf its box touches (about 5 500 at k = 1000 on a surface map of 47 000
// photons per unit area) but only a third of them lie inside some request's ball; packing makes the three passes
// stream that third, in full steps of 64, from one contiguous list.  Nothing here needs the per-request search,
// whose candidate buffer and registers would cap the kernel at two waves per SIMD.
//
// Same exact k-nearest set as the per-request form (rt_gather.hip.h): the search radius is exclusive in both,
// photons at exactly the k-th distance either all belong or the request is handed on.  Requests a group cannot
// take (rounds of a few requests, a range that fails its check, a crowded bin, lists that overflow) go to the
// hand-over list, which the per-request kernel works off.  Subsurface requests (main.cpp:297-388: one gather in
// the surface map, three in the map of photons scattered inside the material) are answered whole by the
// subsurface instantiation of the kernel, in up to three more phases of rounds over the volume map per window.
#pragma once
#include "rt_gather_tile.hip.h"

namespace rtg {

constexpr int S_PACK_CAP = 8192;      // photons the packed list of a wave holds at most (16 bytes each); TileArgs::pack_cap is the size in use
constexpr int S_ATTEMPTS = 7;         // guesses at the centre's radius per round before the round is handed on
constexpr int S_ATTEMPTS_VOLUME = 4;  // the same in a volume phase of a subsurface window (whose wave has three more phases to see to)
constexpr int CONT_SHARDS = 8;        // lists of window parts handed from wave to wave (TileArgs::cont)
constexpr int S_LEVEL = 384;          // subtrees of one tree level a group's box may touch
constexpr int S_LEAVES = 448;         // leaves it may touch (what the per-lane tables of the passes leave room for: see StreamLds)
constexpr int S_SSS_ATTEMPTS = 10;    // ... walks for a guessed radius, all its rounds together (one per round is the common case)
constexpr int S_SSS_STEPS = 320;      // ... and steps of 64 photons its rounds may stream per pass, all together, before another round is begun
constexpr int S_SSS_TRIES = 8;        // rounds a window with subsurface requests may take, its surface and volume phases together

// Per-wave LDS of the kernel: 12.5 KB, so that twelve waves share a CU.  The per-lane tables of the passes lie over
// what the walk and the sort of its lists need, which is dead once the packed list exists.
struct StreamLds {
    union {
        uint32_t table[(T_BINS / 2) * WAVE];      // histograms: [bin][32 words], two 16-bit counts each; critical bin: [T_CRIT][64]
        struct {
            uint32_t leaf[S_LEAVES], piv[T_PIVOTS];           // as the walk lists them: by tree level, left to right
            int a_lo[S_LEVEL], a_hi[S_LEVEL];                 // the subtrees of the level being looked at
            int b_lo[S_LEVEL], b_hi[S_LEVEL];                 // those of the level below, in the making
        } w;
    };
    float stage[9 * WAVE];                        // a step's photons: x[64] y[64] z[64] (+ power, direction in a sum pass); the radix histogram of step 1
};
static_assert(T_CRIT * WAVE <= (T_BINS / 2) * WAVE && sizeof(StreamLds) <= 13 * 1024, "twelve waves per CU");
__shared__ StreamLds g_S;

__device__ __forceinline__ unsigned long long bcast64(unsigned long long v, int src)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src);
    return ((unsigned long long)hi << 32) | lo;
}

// ---- the walk: leaves and pivots inside the box [gl, gh] -> g_S.w.leaf / piv; n_leaf | n_piv << 16, or -1 when the
// box touches more of the tree than the lists hold.  The tree is taken level by level, every level from left to right
// (a lane's children go to the next level's list side by side, in the lanes' order), so the lists come out ordered by
// (level, tree position): two walks for different boxes list what they have in common in the same order, which is
// what makes the packed list -- and the order of the sums over it -- a function of the final box alone.
// Of the leaves inside the box only those within sqrt(reach2) of the centre c are listed (by their bounding boxes: the distance
// formed like a photon's from offsets no larger in magnitude, so it never exceeds that of a photon inside): the packing
// keeps nothing beyond that reach, and the corners of the box hold a fifth of its leaves.
__device__ __attribute__((noinline)) int stream_walk(PhotonMapView pm, float glx, float gly, float glz, float ghx, float ghy, float ghz, float cx, float cy, float cz, float reach2)
{
    StreamLds& S = g_S;
    const int lane = lane_id();
    int n_leaf = 0, n_piv = 0, n_cur = 1;
    int *c_lo = S.w.a_lo, *c_hi = S.w.a_hi, *x_lo = S.w.b_lo, *x_hi = S.w.b_hi;
    if (lane == 0) { c_lo[0] = 0; c_hi[0] = (int)pm.n; }
    __builtin_amdgcn_wave_barrier();
    while (n_cur > 0) {
        int n_next = 0;
        for (int base = 0; base < n_cur; base += WAVE) {
            const int take = n_cur - base < WAVE ? n_cur - base : WAVE;
            int rlo = 0, rhi = 0;
            if (lane < take) { rlo = c_lo[base + lane]; rhi = c_hi[base + lane]; }
            const bool live = lane < take && rhi > rlo;
            const bool small = live && rhi - rlo <= WAVE;
            bool is_leaf = small;
            if (is_leaf) {
                const float* b = pm.leaf_box + 6 * (size_t)(rlo >> 5);
                is_leaf = !(b[0] > ghx || b[3] < glx || b[1] > ghy || b[4] < gly || b[2] > ghz || b[5] < glz);
                const float ex = fmaxf(fmaxf(b[0] - cx, cx - b[3]), 0.f), ey = fmaxf(fmaxf(b[1] - cy, cy - b[4]), 0.f), ez = fmaxf(fmaxf(b[2] - cz, cz - b[5]), 0.f);
                is_leaf = is_leaf && ex * ex + ey * ey + ez * ez <= reach2;
            }
            const bool is_inner = live && !small;
            int mid = 0;
            bool piv_in = false, go_l = false, go_r = false;
            if (is_inner) {
                mid = rlo + (rhi - rlo) / 2;
                const float4 pv = pm.pos[mid];
                const int ax = (int)pv.w;
                piv_in = !(pv.x > ghx || pv.x < glx || pv.y > ghy || pv.y < gly || pv.z > ghz || pv.z < glz);
                const float split = ax == 0 ? pv.x : (ax == 1 ? pv.y : pv.z);
                const float lo_ax = ax == 0 ? glx : (ax == 1 ? gly : glz), hi_ax = ax == 0 ? ghx : (ax == 1 ? ghy : ghz);
                go_l = lo_ax <= split && mid > rlo;           // the left side holds coordinates <= split
                go_r = hi_ax >= split && rhi > mid + 1;       // the right side holds coordinates >= split
            }
            const unsigned long long m_leaf = __ballot(is_leaf), m_piv = __ballot(piv_in), m_l = __ballot(go_l), m_r = __ballot(go_r);
            const int c_leaf = __builtin_popcountll(m_leaf), c_piv = __builtin_popcountll(m_piv), c_l = __builtin_popcountll(m_l), c_r = __builtin_popcountll(m_r);
            if (n_leaf + c_leaf > S_LEAVES || n_piv + c_piv > T_PIVOTS || n_next + c_l + c_r > S_LEVEL) return -1;
            if (is_leaf) S.w.leaf[n_leaf + (int)prefix_rank(m_leaf)] = ((uint32_t)rlo << 7) | (uint32_t)(rhi - rlo);
            if (piv_in) S.w.piv[n_piv + (int)prefix_rank(m_piv)] = (uint32_t)mid;
            const int at = n_next + (int)prefix_rank(m_l) + (int)prefix_rank(m_r);      // children of the lanes before this one
            if (go_l) { x_lo[at] = rlo; x_hi[at] = mid; }
            if (go_r) { x_lo[at + (go_l ? 1 : 0)] = mid + 1; x_hi[at + (go_l ? 1 : 0)] = rhi; }
            n_leaf += c_leaf; n_piv += c_piv; n_next += c_l + c_r;
        }
        __builtin_amdgcn_wave_barrier();
        int* t = c_lo; c_lo = x_lo; x_lo = t;
        t = c_hi; c_hi = x_hi; x_hi = t;
        n_cur = n_next;
    }
    return n_leaf | (n_piv << 16);
}

// Step s of a pair of lists (leaves, pivots): lane j's tree position (clamped to a valid one) and the photons in the step.
__device__ __forceinline__ int list_step(const uint32_t* leaf, const uint32_t* piv, int s, int n_leaf, int n_piv, int lane, int& n)
{
    if (s < n_leaf) {
        const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)leaf[s]);
        n = (int)(e & 127u);
        return (int)(e >> 7) + (lane < n ? lane : n - 1);
    }
    const int first = (s - n_leaf) * WAVE;
    n = n_piv - first < WAVE ? n_piv - first : WAVE;
    return (int)piv[first + (lane < n ? lane : n - 1)];
}

// a step of the packed list: lane j's entry (clamped) and the entries in the step
__device__ __forceinline__ float4 pack_step(const float4* pack, int s, int n_pack, int lane, int& n)
{
    const int first = s * WAVE;
    n = n_pack - first < WAVE ? n_pack - first : WAVE;
    return load_global4(pack + first + (lane < n ? lane : n - 1));
}

// ---- packing, first stage: the photons of the listed leaves and pivots within sqrt(reach2) of the centre, in tree
// order, as (x, y, z, tree position).  Returns their number, or -1 when they exceed the list.
// On the way, the photons within sqrt(lim) of the centre (strictly) are counted into *inside and their squared distances
// binned into 256 linear bins over [0, lim) (C_HIST words of the staging area).
constexpr int C_BINS = 256;
__device__ __forceinline__ uint32_t* centre_hist() { return reinterpret_cast<uint32_t*>(g_S.stage); }
__device__ __forceinline__ int centre_bin(float d2, float hscale) { const int b = (int)(d2 * hscale); return b < C_BINS - 1 ? b : C_BINS - 1; }
__device__ __attribute__((noinline)) int pack_listed(const float4* __restrict__ pos, float4* pack, int pack_cap, float cx, float cy, float cz, float reach2, float lim, int n_leaf,
                                                     int n_piv, int* inside)
{
    const int lane = lane_id();
    uint32_t* const hist = centre_hist();
    for (int b = lane; b < C_BINS; b += WAVE) hist[b] = 0;
    __builtin_amdgcn_wave_barrier();
    const float hscale = (float)C_BINS / lim;
    int count = 0;
    n_leaf = __builtin_amdgcn_readfirstlane(n_leaf); n_piv = __builtin_amdgcn_readfirstlane(n_piv);
    const int n_steps = n_leaf + (n_piv + WAVE - 1) / WAVE;
    int out = 0;
    // S_FLIGHT steps are fetched together (every fetch unconditional, the step clamped to a valid one: a load inside a
    // branch is waited for at the branch's end): one memory round trip per S_FLIGHT x 64 photons
    for (int s0 = 0; s0 < n_steps; s0 += S_FLIGHT) {
        float4 p[S_FLIGHT];
        int at[S_FLIGHT], n[S_FLIGHT];
#pragma unroll
        for (int u = 0; u < S_FLIGHT; u++) {
            at[u] = list_step(g_S.w.leaf, g_S.w.piv, s0 + u < n_steps ? s0 + u : n_steps - 1, n_leaf, n_piv, lane, n[u]);
            if (s0 + u >= n_steps) n[u] = 0;
            p[u] = load_global4(pos + at[u]);
        }
#pragma unroll
        for (int u = 0; u < S_FLIGHT; u++) {
            const float dx = cx - p[u].x, dy = cy - p[u].y, dz = cz - p[u].z;      // the operation order of every other distance here
            const float d2 = dx * dx + dy * dy + dz * dz;
            const bool keep = lane < n[u] && d2 <= reach2;
            const unsigned long long m = __ballot(keep);
            const int c = __builtin_popcountll(m);
            if (out + c > pack_cap) return -1;
            if (keep) pack[out + (int)prefix_rank(m)] = float4{p[u].x, p[u].y, p[u].z, __uint_as_float((uint32_t)at[u])};
            out += c;
            const bool in = lane < n[u] && d2 < lim;
            if (in) atomicAdd(&hist[centre_bin(d2, hscale)], 1u);
            count += __builtin_popcountll(__ballot(in));
        }
    }
    __builtin_amdgcn_wave_barrier();
    *inside = count;
    return out;
}

// The k-th smallest squared distance (k <= the count pack_listed returned) from the histogram it left: the bin that holds
// it, then that bin's entries ranked.  Returns the bit pattern, or 0xFFFFFFFF when the bin holds more entries than the
// list (many equal distances): the radix passes of centre_kth decide then.
__device__ __attribute__((noinline)) uint32_t centre_select(const float4* pack, int n_pack, float cx, float cy, float cz, float lim, int k)
{
    uint32_t* const hist = centre_hist();
    uint32_t* const list = hist + C_BINS;                     // C_BINS more words of the staging area
    const int lane = lane_id();
    // lane l owns bins 4l .. 4l + 3
    const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
    const uint32_t sum = h0 + h1 + h2 + h3;
    uint32_t incl = sum;
    for (int off = 1; off < WAVE; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    const uint32_t excl = incl - sum;
    const bool mine = (uint32_t)k > excl && (uint32_t)k <= incl;      // exactly one lane
    uint32_t b = 0, bl = 0, ib = 0;
    if (mine) {
        const uint32_t w = (uint32_t)k - excl;
        if (w <= h0) { b = 0; bl = excl; ib = h0; }
        else if (w <= h0 + h1) { b = 1; bl = excl + h0; ib = h1; }
        else if (w <= h0 + h1 + h2) { b = 2; bl = excl + h0 + h1; ib = h2; }
        else { b = 3; bl = excl + h0 + h1 + h2; ib = h3; }
        b += 4 * lane;
    }
    const int src = __builtin_ctzll(__ballot(mine));
    const int bin = (int)__shfl(b, src);
    const int need = k - (int)__shfl(bl, src);                // 1-based rank inside the bin
    const int in_bin = (int)__shfl(ib, src);
    if (in_bin > C_BINS) return 0xFFFFFFFFu;
    __builtin_amdgcn_wave_barrier();
    const float hscale = (float)C_BINS / lim;
    n_pack = __builtin_amdgcn_readfirstlane(n_pack);
    const int n_steps = (n_pack + WAVE - 1) / WAVE;
    int m = 0;
    for (int s0 = 0; s0 < n_steps; s0 += S_FLIGHT) {
        float4 p[S_FLIGHT];
        int n[S_FLIGHT];
#pragma unroll
        for (int u = 0; u < S_FLIGHT; u++) {
            p[u] = pack_step(pack, s0 + u < n_steps ? s0 + u : n_steps - 1, n_pack, lane, n[u]);
            if (s0 + u >= n_steps) n[u] = 0;
        }
#pragma unroll
        for (int u = 0; u < S_FLIGHT; u++) {
            const float dx = cx - p[u].x, dy = cy - p[u].y, dz = cz - p[u].z;
            const float d2 = dx * dx + dy * dy + dz * dz;
            const bool hit = lane < n[u] && d2 < lim && centre_bin(d2, hscale) == bin;
            const unsigned long long mh = __ballot(hit);
            if (hit) list[m + (int)prefix_rank(mh)] = __float_as_uint(d2);
            m += __builtin_popcountll(mh);
        }
    }
    __builtin_amdgcn_wave_barrier();
    // m == in_bin; the need-th smallest of the list (entries may repeat: ties broken by position)
    uint32_t found = 0;
    bool have = false;
    for (int i0 = 0; i0 < m; i0 += WAVE) {
        const int i = i0 + lane;
        const uint32_t e = list[i < m ? i : m - 1];
        int rank = 0;
        for (int j = 0; j < m; j++) { const uint32_t o = list[j]; rank += (o < e || (o == e && j < i)) ? 1 : 0; }
        if (i < m && rank == need - 1) { found = e; have = true; }
    }
    const unsigned long long mf = __ballot(have);
    __builtin_amdgcn_wave_barrier();
    return mf ? (uint32_t)__shfl((int)found, __builtin_ctzll(mf)) : 0xFFFFFFFFu;
}

// ---- packing, second stage, in place: what lies inside the box [gl, gh] and within sqrt(reach2) of the centre
struct Box6f { float lx, ly, lz, hx, hy, hz; };
__device__ __attribute__((noinline)) int pack_refilter(float4* pack, int n_pack, float cx, float cy, float cz, float reach2, Box6f bx)
{
    const int lane = lane_id();
    n_pack = __builtin_amdgcn_readfirstlane(n_pack);
    const int n_steps = (n_pack + WAVE - 1) / WAVE;
    int out = 0;
    for (int s0 = 0; s0 < n_steps; s0 += S_FLIGHT) {
        // a batch's entries are all in registers before any of its outputs is written, and outputs never pass the batch
        float4 p[S_FLIGHT];
        int n[S_FLIGHT];
#pragma unroll
        for (int u = 0; u < S_FLIGHT; u++) {
            p[u] = pack_step(pack, s0 + u < n_steps ? s0 + u : n_steps - 1, n_pack, lane, n[u]);
            if (s0 + u >= n_steps) n[u] = 0;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < S_FLIGHT; u++) {
            const float dx = p[u].x - cx, dy = p[u].y - cy, dz = p[u].z - cz;
            const bool inside = !(p[u].x > bx.hx || p[u].x < bx.lx || p[u].y > bx.hy || p[u].y < bx.ly || p[u].z > bx.hz || p[u].z < bx.lz);
            const bool keep = lane < n[u] && inside && dx * dx + dy * dy + dz * dz <= reach2;
            const unsigned long long m = __ballot(keep);
            if (keep) pack[out + (int)prefix_rank(m)] = p[u];
            out += __builtin_popcountll(m);
        }
        __builtin_amdgcn_wave_barrier();
    }
    return out;
}

// ---- the centre's k-th smallest squared distance among the packed photons with d2 < lim, by all lanes together:
// most-significant-byte-first radix passes over the bit patterns (non-negative floats order like their bits).
// Returns the bit pattern; *total = photons with d2 < lim (fewer than k: the return value means nothing).
__device__ __attribute__((noinline)) uint32_t centre_kth(const float4* pack, int n_pack, float cx, float cy, float cz, float lim, int k, int* total)
{
    uint32_t* const hist = reinterpret_cast<uint32_t*>(g_S.stage);      // 256 words of the staging area, idle here
    const int lane = lane_id();
    n_pack = __builtin_amdgcn_readfirstlane(n_pack);
    const int n_steps = (n_pack + WAVE - 1) / WAVE;
    uint32_t prefix = 0, pmask = 0;
    int want = k;
    for (int pass = 3; pass >= 0; pass--) {
        for (int b = lane; b < 256; b += WAVE) hist[b] = 0;
        __builtin_amdgcn_wave_barrier();
        for (int s0 = 0; s0 < n_steps; s0 += S_FLIGHT) {
            float4 p[S_FLIGHT];
            int n[S_FLIGHT];
#pragma unroll
            for (int u = 0; u < S_FLIGHT; u++) {
                p[u] = pack_step(pack, s0 + u < n_steps ? s0 + u : n_steps - 1, n_pack, lane, n[u]);
                if (s0 + u >= n_steps) n[u] = 0;
            }
#pragma unroll
            for (int u = 0; u < S_FLIGHT; u++) {
                const float dx = cx - p[u].x, dy = cy - p[u].y, dz = cz - p[u].z;      // the operation order of every other distance here
                const float d2 = dx * dx + dy * dy + dz * dz;
                const uint32_t bits = __float_as_uint(d2);
                if (lane < n[u] && d2 < lim && (bits & pmask) == prefix) atomicAdd(&hist[(bits >> (8 * pass)) & 255u], 1u);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // lane l owns bins 4l .. 4l + 3: the bin holding the want-th entry
        const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
        const uint32_t sum = h0 + h1 + h2 + h3;
        uint32_t incl = sum;
        for (int off = 1; off < WAVE; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        const uint32_t all = __shfl(incl, WAVE - 1);
        if (pass == 3) {
            *total = (int)all;
            if ((int)all < k) return 0u;
        }
        const uint32_t excl = incl - sum;
        const bool mine = (uint32_t)want > excl && (uint32_t)want <= incl;      // exactly one lane
        uint32_t b = 0, bl = 0;
        if (mine) {
            const uint32_t w = (uint32_t)want - excl;
            if (w <= h0) { b = 0; bl = excl; }
            else if (w <= h0 + h1) { b = 1; bl = excl + h0; }
            else if (w <= h0 + h1 + h2) { b = 2; bl = excl + h0 + h1; }
            else { b = 3; bl = excl + h0 + h1 + h2; }
            b += 4 * lane;
        }
        const int src = __builtin_ctzll(__ballot(mine));
        const uint32_t bin = __shfl(b, src);
        want -= (int)__shfl(bl, src);
        prefix |= bin << (8 * pass);
        pmask |= 255u << (8 * pass);
        __builtin_amdgcn_wave_barrier();
    }
    return prefix;
}

// ---- the three passes over the packed list (see rt_gather_tile.hip.h for the arithmetic they share)
// A pass fetches step s + 1 (or s + 2) from the packed list while it works on step s.  The registers that receive a
// fetch must not be needed for anything else meanwhile -- a register of the destination reused as a temporary makes
// the compiler wait for the fetch at once -- so the histogram and critical-bin passes, which have no use for an
// entry's fourth word, fetch three words only.
struct P3 { float x, y, z; };
typedef float v3f __attribute__((ext_vector_type(3)));
__device__ __forceinline__ P3 pack_step_xyz(const float4* pack, int s, int n_pack, int lane, int& n)
{
    const int first = s * WAVE;
    n = n_pack - first < WAVE ? n_pack - first : WAVE;
    const v3f v = *(const v3f __attribute__((address_space(1)))*)(pack + first + (lane < n ? lane : n - 1));
    return P3{v.x, v.y, v.z};
}
__device__ __forceinline__ void s_stage_put(float* st, float x, float y, float z, int lane, int n)
{
    __builtin_amdgcn_wave_barrier();       // (the reads of the step before are done: LDS operations of a wave complete in order)
    st[lane] = lane < n ? x : 1.0e18f; st[WAVE + lane] = y; st[2 * WAVE + lane] = z;      // (a filler no ball reaches: the loops over a step need no bound checks)
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ Xyz4 s_quad_load(const float* st, int j)
{
    return Xyz4{*reinterpret_cast<const float4*>(st + j), *reinterpret_cast<const float4*>(st + WAVE + j), *reinterpret_cast<const float4*>(st + 2 * WAVE + j)};
}
// f(what load(j) returned) for the quads j of a staged step, the next quad's data read from LDS before the current one is
// worked on.  A full step -- all but a list's last -- is straight-line code: no wait for LDS at a loop edge.
template <class L, class F>
__device__ __forceinline__ void for_quads(int n_now, L&& load, F&& f)
{
    auto cur = load(0);
    if (n_now == WAVE) {
#pragma unroll
        for (int j = 0; j < WAVE; j += 4) {
            const auto nxt = load((j + 4) & (WAVE - 1));
            __builtin_amdgcn_sched_barrier(0);      // (the reads stay ahead of the arithmetic: left to itself the scheduler sinks them to their first use)
            f(cur);
            cur = nxt;
        }
    } else {
        for (int j = 0; j < n_now; j += 4) {
            const auto nxt = load((j + 4) & (WAVE - 1));
            __builtin_amdgcn_sched_barrier(0);
            f(cur);
            cur = nxt;
        }
    }
}

template <bool any_radius>
__device__ __attribute__((noinline)) void stream_hist(const float4* pack, int n_pack, float qx, float qy, float qz, Shell sh, float radius2)
{
    uint32_t* const cell = g_S.table;
    float* const st = g_S.stage;
    const int lane = lane_id();
#pragma unroll 8
    for (int b = 0; b < T_BINS / 2; b++) cell[b * WAVE + lane] = 0u;
    __builtin_amdgcn_wave_barrier();
    n_pack = __builtin_amdgcn_readfirstlane(n_pack);
    const int n_steps = (n_pack + WAVE - 1) / WAVE;
    uint32_t* const mine = hist_mine(cell, lane);
    const uint32_t unit = hist_unit(lane);
    // (the entries of steps s + 1 .. s + 3 are on their way while step s is binned: the packed lists of all waves together
    // are larger than the caches, and a fetch from memory takes longer than a step)
    if (n_steps == 0) return;
    int n = 0, n1 = 0, n2 = 0, n3 = 0;
    P3 p = pack_step_xyz(pack, 0, n_pack, lane, n);
    P3 p1 = pack_step_xyz(pack, 1 < n_steps ? 1 : n_steps - 1, n_pack, lane, n1);
    P3 p2 = pack_step_xyz(pack, 2 < n_steps ? 2 : n_steps - 1, n_pack, lane, n2);
    for (int s = 0; s < n_steps; s++) {
        s_stage_put(st, p.x, p.y, p.z, lane, n);
        const int n_now = __builtin_amdgcn_readfirstlane(n);
        const P3 p3 = pack_step_xyz(pack, s + 3 < n_steps ? s + 3 : n_steps - 1, n_pack, lane, n3);
        for_quads(n_now, [&](int j) { return s_quad_load(st, j); }, [&](const Xyz4& c) {
            const D2x4 d = quad_d2(c, qx, qy, qz);
            const f2 at_a = shell_pos(d.a, sh), at_b = shell_pos(d.b, sh);
            int b0 = shell_bin_of(at_a.x), b1 = shell_bin_of(at_a.y), b2 = shell_bin_of(at_b.x), b3 = shell_bin_of(at_b.y);
            if (any_radius) {
                b0 = d.a.x < radius2 ? b0 : T_BINS - 1; b1 = d.a.y < radius2 ? b1 : T_BINS - 1;
                b2 = d.b.x < radius2 ? b2 : T_BINS - 1; b3 = d.b.y < radius2 ? b3 : T_BINS - 1;
            }
            hist_add(mine, unit, b0);
            hist_add(mine, unit, b1);
            hist_add(mine, unit, b2);
            hist_add(mine, unit, b3);
        });
        p = p1; n = n1; p1 = p2; n1 = n2; p2 = p3; n2 = n3;
    }
    __builtin_amdgcn_wave_barrier();
}

template <bool any_radius>
__device__ __attribute__((noinline)) int stream_crit(const float4* pack, int n_pack, float qx, float qy, float qz, Shell sh, float radius2, int crit)
{
    uint32_t* const cell = g_S.table;
    float* const st = g_S.stage;
    const int lane = lane_id();
    int cnt = 0;
    n_pack = __builtin_amdgcn_readfirstlane(n_pack);
    const int n_steps = (n_pack + WAVE - 1) / WAVE;
    // (bin membership by one unsigned comparison: see pass_crit)
    const float crit_f = crit >= 1 ? (float)crit : 1.0e30f;
    const f2 crit_2 = {crit_f, crit_f};
    constexpr uint32_t ONE = 0x3F800000u;
    if (n_steps == 0) return 0;
    int n = 0, n1 = 0, n2 = 0, n3 = 0;
    P3 p = pack_step_xyz(pack, 0, n_pack, lane, n);
    P3 p1 = pack_step_xyz(pack, 1 < n_steps ? 1 : n_steps - 1, n_pack, lane, n1);
    P3 p2 = pack_step_xyz(pack, 2 < n_steps ? 2 : n_steps - 1, n_pack, lane, n2);
    for (int s = 0; s < n_steps; s++) {
        s_stage_put(st, p.x, p.y, p.z, lane, n);
        const int n_now = __builtin_amdgcn_readfirstlane(n);
        const P3 p3 = pack_step_xyz(pack, s + 3 < n_steps ? s + 3 : n_steps - 1, n_pack, lane, n3);
        for_quads(n_now, [&](int j) { return s_quad_load(st, j); }, [&](const Xyz4& c) {
            const D2x4 d = quad_d2(c, qx, qy, qz);
            const f2 at_a = shell_pos(d.a, sh), at_b = shell_pos(d.b, sh);
            const f2 t_a = at_a - crit_2, t_b = at_b - crit_2;
            uint32_t u0 = __float_as_uint(t_a.x), u1 = __float_as_uint(t_a.y), u2 = __float_as_uint(t_b.x), u3 = __float_as_uint(t_b.y);
            if (any_radius) {      // (what the histogram put into the last bin is not in this one)
                u0 = d.a.x < radius2 ? u0 : ONE; u1 = d.a.y < radius2 ? u1 : ONE; u2 = d.b.x < radius2 ? u2 : ONE; u3 = d.b.y < radius2 ? u3 : ONE;
            }
            const uint32_t least = min(min(u0, u1), min(u2, u3));
            if (__ballot(least < ONE)) {           // rare: a few dozen photons per request
                if (u0 < ONE && cnt < T_CRIT) { cell[cnt * WAVE + lane] = __float_as_uint(d.a.x); cnt++; }
                if (u1 < ONE && cnt < T_CRIT) { cell[cnt * WAVE + lane] = __float_as_uint(d.a.y); cnt++; }
                if (u2 < ONE && cnt < T_CRIT) { cell[cnt * WAVE + lane] = __float_as_uint(d.b.x); cnt++; }
                if (u3 < ONE && cnt < T_CRIT) { cell[cnt * WAVE + lane] = __float_as_uint(d.b.y); cnt++; }
            }
        });
        p = p1; n = n1; p1 = p2; n1 = n2; p2 = p3; n2 = n3;
    }
    __builtin_amdgcn_wave_barrier();
    return cnt;
}

// the sums of main.cpp:262-291 over the photons with d2 <= t_k (see pass_sum).  Power and direction of a photon are
// fetched by the tree position its entry carries, so the entries run two steps ahead of the arithmetic and the records one.
struct Aux6 { float b0, b1, b2, b3, b4, b5; };
__device__ __forceinline__ Aux6 aux_fetch(const float* aux, uint32_t at)
{
    const float __attribute__((address_space(1)))* ax = as_global(aux) + 6 * (size_t)at;
    return Aux6{ax[0], ax[1], ax[2], ax[3], ax[4], ax[5]};
}
// ONE_NORMAL: every lane that sums has the same surface normal (the requests of a window on one flat wall, which is
// most windows of a scene whose diffuse surfaces are planes): |n . u| of a photon is then the same for all lanes, and the
// lane that stages the photon forms it once -- same operands, same order, same value as every lane would get -- instead
// of 64 lanes forming it each (a fifth of the pass's instructions).
template <bool ONE_NORMAL>
__device__ __attribute__((noinline)) Sum3 stream_sum(const float4* pack, int n_pack, const float* __restrict__ aux, float qx, float qy, float qz, float t_k, bool some,
                                                     float nx, float ny, float nz)
{
    float* const st = g_S.stage;
    float* const sa = g_S.stage + 3 * WAVE;
    const int lane = lane_id();
    const TermCoef tc = term_coef(t_k);
    double s0 = 0, s1 = 0, s2 = 0;
    n_pack = __builtin_amdgcn_readfirstlane(n_pack);
    const int n_steps = (n_pack + WAVE - 1) / WAVE;
    if (n_steps == 0) return Sum3{0, 0, 0};
    const float lim = some ? t_k : -1.f;
    int n = 0, n1 = 0, n2 = 0;
    float4 p = pack_step(pack, 0, n_pack, lane, n);                                    // step s
    float4 p1 = pack_step(pack, n_steps > 1 ? 1 : 0, n_pack, lane, n1);               // step s + 1
    Aux6 a = aux_fetch(aux, __float_as_uint(p.w));                                     // records of step s
    for (int s = 0; s < n_steps; s++) {
        s_stage_put(st, p.x, p.y, p.z, lane, n);
        {
            sa[lane] = a.b0; sa[WAVE + lane] = a.b1; sa[2 * WAVE + lane] = a.b2;
            if (ONE_NORMAL) {
                sa[3 * WAVE + lane] = fabsf(nx * a.b3 + ny * a.b4 + nz * a.b5);      // (nx, ny, nz: the one normal, in every lane)
            } else {
                sa[3 * WAVE + lane] = a.b3; sa[4 * WAVE + lane] = a.b4; sa[5 * WAVE + lane] = a.b5;
            }
            __builtin_amdgcn_wave_barrier();
        }
        const int n_now = __builtin_amdgcn_readfirstlane(n);
        // on their way while this step is summed (unconditional, clamped to the last step: a fetch inside a branch is waited for
        // at the branch's end): the records of step s + 1, whose entries arrived a step ago, and the entries of step s + 2
        a = aux_fetch(aux, __float_as_uint(p1.w));
        const float4 p2 = pack_step(pack, s + 2 < n_steps ? s + 2 : n_steps - 1, n_pack, lane, n2);
        struct Quad { Xyz4 c; float4 pr, pg, pb, ux, uy, uz; };
        auto quad_all = [&](int j) {
            if (ONE_NORMAL)
                return Quad{s_quad_load(st, j), *reinterpret_cast<const float4*>(sa + j), *reinterpret_cast<const float4*>(sa + WAVE + j), *reinterpret_cast<const float4*>(sa + 2 * WAVE + j),
                            *reinterpret_cast<const float4*>(sa + 3 * WAVE + j), float4{0.f, 0.f, 0.f, 0.f}, float4{0.f, 0.f, 0.f, 0.f}};
            return Quad{s_quad_load(st, j), *reinterpret_cast<const float4*>(sa + j), *reinterpret_cast<const float4*>(sa + WAVE + j), *reinterpret_cast<const float4*>(sa + 2 * WAVE + j),
                        *reinterpret_cast<const float4*>(sa + 3 * WAVE + j), *reinterpret_cast<const float4*>(sa + 4 * WAVE + j), *reinterpret_cast<const float4*>(sa + 5 * WAVE + j)};
        };
        for_quads(n_now, quad_all, [&](const Quad& q) {
            const D2x4 d = quad_d2(q.c, qx, qy, qz);
            // four photons at a time without a branch per photon or per quad: one outside the lane's ball contributes a
            // factor 0 (the packed list holds little that lies outside every ball); the four terms of a channel are added in
            // float before they join the double sum
            const bool in0 = d.a.x <= lim, in1 = d.a.y <= lim, in2 = d.b.x <= lim, in3 = d.b.y <= lim;
            {
                const float4 &pr4 = q.pr, &pg4 = q.pg, &pb4 = q.pb, &ux4 = q.ux, &uy4 = q.uy, &uz4 = q.uz;
                const f2 c2 = {tc.c2, tc.c2}, A = {tc.A, tc.A}, B = {tc.B, tc.B};
                const f2 ea = c2 * d.a, eb = c2 * d.b;
                const f2 wa = A + B * f2{__builtin_amdgcn_exp2f(ea.x), __builtin_amdgcn_exp2f(ea.y)}, wb = A + B * f2{__builtin_amdgcn_exp2f(eb.x), __builtin_amdgcn_exp2f(eb.y)};
                float t0, t1, t2, t3;      // |n . u| of the four photons
                if (ONE_NORMAL) {
                    t0 = ux4.x; t1 = ux4.y; t2 = ux4.z; t3 = ux4.w;      // (staged in the place of the direction's first component)
                } else {
                    const f2 n_x = {nx, nx}, n_y = {ny, ny}, n_z = {nz, nz};
                    const f2 ca = n_x * f2{ux4.x, ux4.y} + n_y * f2{uy4.x, uy4.y} + n_z * f2{uz4.x, uz4.y};
                    const f2 cb = n_x * f2{ux4.z, ux4.w} + n_y * f2{uy4.z, uy4.w} + n_z * f2{uz4.z, uz4.w};
                    t0 = fabsf(ca.x); t1 = fabsf(ca.y); t2 = fabsf(cb.x); t3 = fabsf(cb.y);
                }
                const float sc0 = in0 ? t0 * wa.x : 0.f, sc1 = in1 ? t1 * wa.y : 0.f;
                const float sc2 = in2 ? t2 * wb.x : 0.f, sc3 = in3 ? t3 * wb.y : 0.f;
                const f2 sa2 = {sc0, sc1}, sb2 = {sc2, sc3};
                const f2 r = f2{pr4.x, pr4.y} * sa2 + f2{pr4.z, pr4.w} * sb2, g = f2{pg4.x, pg4.y} * sa2 + f2{pg4.z, pg4.w} * sb2, b = f2{pb4.x, pb4.y} * sa2 + f2{pb4.z, pb4.w} * sb2;
                s0 += (double)(r.x + r.y);                   // main.cpp:286 without the hit colour, which the caller multiplies in
                s1 += (double)(g.x + g.y);
                s2 += (double)(b.x + b.y);
            }
        });
        p = p1; n = n1;
        p1 = p2; n1 = n2;
    }
    return Sum3{s0, s1, s2};
}

// The sums of a subsurface step (main.cpp:310-334, 346-373) over the photons with d2 <= t_k of the VOLUME map: the powers
// added up, and the position of the set's farthest photon -- the one at the k-th distance itself; of several at exactly
// that distance the first in the tree, as volume_step picks it -- from which the caller takes the step's normaliser.
// (Only that photon's tree position is noted on the way -- its coordinates, kept at hand for every quad, would cost the loop
// the registers its fetches run ahead in -- and its position is read from the map at the end.)
struct VolSum { double s0, s1, s2; float fx, fy, fz; };
__device__ __attribute__((noinline)) VolSum stream_sum_volume(const float4* pack, int n_pack, const float4* __restrict__ pos, const float* __restrict__ aux, float qx, float qy, float qz,
                                                              float t_k, bool some)
{
    float* const st = g_S.stage;
    float* const sa = g_S.stage + 3 * WAVE;      // power r[64] g[64] b[64], tree position[64]
    const int lane = lane_id();
    double s0 = 0, s1 = 0, s2 = 0;
    uint32_t f_at = 0xFFFFFFFFu;
    n_pack = __builtin_amdgcn_readfirstlane(n_pack);
    const int n_steps = (n_pack + WAVE - 1) / WAVE;
    if (n_steps == 0) return VolSum{0, 0, 0, 0.f, 0.f, 0.f};
    const float lim = some ? t_k : -1.f;
    int n = 0, n1 = 0, n2 = 0;
    float4 p = pack_step(pack, 0, n_pack, lane, n);                                    // step s
    float4 p1 = pack_step(pack, n_steps > 1 ? 1 : 0, n_pack, lane, n1);               // step s + 1
    Aux6 a = aux_fetch(aux, __float_as_uint(p.w));                                     // records of step s
    for (int s = 0; s < n_steps; s++) {
        s_stage_put(st, p.x, p.y, p.z, lane, n);
        sa[lane] = a.b0; sa[WAVE + lane] = a.b1; sa[2 * WAVE + lane] = a.b2; sa[3 * WAVE + lane] = p.w;
        __builtin_amdgcn_wave_barrier();
        const int n_now = __builtin_amdgcn_readfirstlane(n);
        a = aux_fetch(aux, __float_as_uint(p1.w));
        const float4 p2 = pack_step(pack, s + 2 < n_steps ? s + 2 : n_steps - 1, n_pack, lane, n2);
        struct Quad { Xyz4 c; float4 pr, pg, pb, at; };
        auto quad_all = [&](int j) {
            return Quad{s_quad_load(st, j), *reinterpret_cast<const float4*>(sa + j), *reinterpret_cast<const float4*>(sa + WAVE + j), *reinterpret_cast<const float4*>(sa + 2 * WAVE + j),
                        *reinterpret_cast<const float4*>(sa + 3 * WAVE + j)};
        };
        for_quads(n_now, quad_all, [&](const Quad& q) {
            const D2x4 d = quad_d2(q.c, qx, qy, qz);
            // (as in stream_sum: a photon outside the lane's ball contributes a factor 0; four terms of a channel are added in float)
            const f2 sa2 = {d.a.x <= lim ? 1.f : 0.f, d.a.y <= lim ? 1.f : 0.f}, sb2 = {d.b.x <= lim ? 1.f : 0.f, d.b.y <= lim ? 1.f : 0.f};
            const f2 r = f2{q.pr.x, q.pr.y} * sa2 + f2{q.pr.z, q.pr.w} * sb2, g = f2{q.pg.x, q.pg.y} * sa2 + f2{q.pg.z, q.pg.w} * sb2, b = f2{q.pb.x, q.pb.y} * sa2 + f2{q.pb.z, q.pb.w} * sb2;
            s0 += (double)(r.x + r.y);                   // main.cpp:329, 368 without the hit colour and the 0.2, which the caller multiplies in
            s1 += (double)(g.x + g.y);
            s2 += (double)(b.x + b.y);
            // the photon at the k-th distance (no branch around this, rare as it is: with one the compiler splits the step in
            // two and parks what the next quad has read ahead in scratch memory, whose reads wait for the fetches in flight)
            const uint32_t none = 0xFFFFFFFFu;
            f_at = min(f_at, min(min(d.a.x == lim ? __float_as_uint(q.at.x) : none, d.a.y == lim ? __float_as_uint(q.at.y) : none),
                                 min(d.b.x == lim ? __float_as_uint(q.at.z) : none, d.b.y == lim ? __float_as_uint(q.at.w) : none)));
        });
        p = p1; n = n1;
        p1 = p2; n1 = n2;
    }
    const float4 f = load_global4(pos + (f_at != 0xFFFFFFFFu ? f_at : 0u));
    return VolSum{s0, s1, s2, f.x, f.y, f.z};
}

// Slots of the cycle accounting that mean something else here than in gather_tile_kernel: TP_KTH = walk + packing + the
// centre's radix passes of all attempts, TP_WALK = the in-place filter, TP_N_HIST = attempts, TP_N_CRIT = photons packed
// by the last attempt, TP_N_SUM = photons streamed by the passes.
// PROF: the instantiation with the cycle accounting (its counters take some ninety registers), launched only when
// TileArgs::prof is given.
// SSS: the instantiation that also answers the VOLUME terms of subsurface requests (the three findKNN calls in the map of
// photons scattered inside the material, main.cpp:310, 346): after a window's rounds over the surface map its subsurface
// requests go through up to three more phases of rounds over the volume map `vm` -- at the hit point, 0.02 along the
// refracted direction, and one random step further -- with the same machinery; a request whose every phase succeeded is
// finished here (surface + volume terms, clipped, main.cpp:389-391), any other is handed on with its surface estimate.
// The state of a request between its phases is in its lane's registers, so such windows are not split into parts.
template <bool PROF, bool SSS>
__device__ __forceinline__ void gather_stream_body(const PhotonMapView& pm, const PhotonMapView& vm, const TileArgs& ta, const GatherParams& gp, double* __restrict__ acc, float4* pack_all)
{
    uint32_t* const cell = g_S.table;
    const int lane = lane_id();
    float4* const pack = pack_all + (size_t)blockIdx.x * (size_t)ta.pack_cap;
    const rtq::Map map = rtq::map_units(ta.cnt, ta.shard_cap, WAVE);
    const uint32_t n_windows = map.total;
    // draws of the window counter that may name a window: with late_first, places in the longest shard x shards
    uint32_t n_draws = n_windows;
    if (ta.late_first) {
        const uint32_t units = lane < rtq::SHARDS ? (map.n + WAVE - 1) / WAVE : 0u;
        n_draws = (uint32_t)wave_max_f((float)units) * rtq::SHARDS;      // (window counts are far below 2^24: exact in float)
    }
    // With volume terms to answer, the windows are gone through twice: first those that hold subsurface requests -- four
    // phases of rounds each, several times the work of any other window, and a wave that drew one near the end would set the
    // kernel's length -- then the rest.  A look at a window's flags tells which kind it is (a few microseconds per window skipped).
    const bool two_sweeps = SSS && ta.sss_surface > 1;
    const uint32_t n_all = two_sweeps ? 2u * n_draws : n_draws;
    if (lane == 0) atomicAdd(ta.windows_done + 1, 1ull);
    const unsigned long long span_begin = ta.span ? wall_clock64() : 0ull;
    unsigned span_windows = 0, span_spins = 0;
    // Parts of windows are handed from wave to wave (TileArgs::cont) only while the kernel is drawing its last windows:
    // before that every wave has work anyway.  One list per CONT_SHARDS-th of the waves, so that the words the waves
    // count on are spread (a single word takes some sixty additions a microsecond).
    unsigned long long* const my_cont = ta.cont ? ta.cont + (size_t)(blockIdx.x % CONT_SHARDS) * (4 + 2 * (size_t)ta.cont_cap) : nullptr;
    bool sharing = false;
    unsigned long long pr[TP_SLOTS] = {};      // (dead, and gone, in the instantiation without the accounting)
    const unsigned long long t_begin = PROF ? clock64() : 0ull;
    const float r_max = sqrtf(gp.r2);
    float prev_r = -1.f;                 // the last centre's k-th distance: the next guess
    float prev_r_vol = -1.f;             // (the same in the volume map)
#define TILE_PROF(slot, expr) do { if (PROF) { const unsigned long long t_ = clock64(); expr; pr[slot] += clock64() - t_; } else { expr; } } while (0)
#define TILE_COUNT(slot, cond) do { if (PROF) pr[slot] += __builtin_popcountll(__ballot(cond)); } while (0)
    while (true) {
        // ---- what to work on: a part of a window that another wave (or this one) has handed on, else the next window
        unsigned long long c_mask = 0, c_word = 0;
        if (sharing && lane == 0) {
            // (every access a read-modify-write -- a count read by adding zero: those are served in a few microseconds, where a
            // device-coherent load of a word that other waves keep changing was measured to take 150 and more; and no
            // compare-and-swap loops: with a part per window, as at k = 100, hundreds of waves retrying took 8 us per part)
            if ((long long)atomicAdd(my_cont + 1, 0ull) > 0) {
                if ((long long)atomicAdd(my_cont + 1, ~0ull) > 0) {
                    const unsigned long long h = atomicAdd(my_cont, 1ull);
                    // announced only once written, and the announcements before it were for entries before it -- though such
                    // an entry's own writer may be a moment behind: a short wait at most.  Taken and zeroed in one.
                    unsigned long long* const e = my_cont + 4 + 2 * h;
                    while (!(c_mask = atomicExch(e, 0ull))) { __builtin_amdgcn_s_sleep(2); span_spins++; }
                    while (!(c_word = atomicExch(e + 1, 0ull))) { __builtin_amdgcn_s_sleep(2); span_spins++; }
                } else {
                    atomicAdd(my_cont + 1, 1ull);      // (another wave was quicker: the count goes back up)
                }
            }
        }
        c_word = bcast64(c_word, 0);
        c_mask = bcast64(c_mask, 0);
        const bool part = c_word != 0;                // a handed-on part: its window, the lanes waiting, the rounds the window has had
        const int round_first = part ? (int)((c_word >> 32) & 0xFFu) : 0;
        uint32_t win;
        bool first_sweep = false;
        if (part) {
            win = (uint32_t)c_word;
            sharing = true;
        } else {
            unsigned long long w0 = 0;
            if (lane == 0) w0 = atomicAdd(ta.next, 1ull);
            uint32_t drawn = (uint32_t)__builtin_amdgcn_readfirstlane((unsigned)w0);
            const uint32_t drawn_hi = (uint32_t)__builtin_amdgcn_readfirstlane((unsigned)(w0 >> 32));
            if (drawn_hi || drawn >= n_all) break;       // (the high half is zero for any real queue); parts handed on later stay with their wave
            sharing = ta.cont != nullptr && (unsigned long long)drawn + (unsigned long long)ta.part_tail * gridDim.x >= n_all;
            first_sweep = two_sweeps && drawn < n_draws;
            if (two_sweeps && !first_sweep) drawn -= n_draws;
            if (ta.late_first) {
                // The shards' windows from the last appended to the first, the shards taken in turn: what the later shading
                // passes queued -- hit points of rays that left a refracting or reflecting surface, scattered where a tile of
                // camera rays lands in one patch -- makes the windows of several rounds and most of the requests handed on,
                // and those are best met while every wave still has work to turn to.
                const int s = (int)(drawn % rtq::SHARDS);
                const uint32_t j = drawn / rtq::SHARDS;
                const uint32_t units_s = (__shfl(map.n, s) + WAVE - 1) / WAVE;
                if (j >= units_s) continue;               // (a shorter shard than the longest: nothing at this place)
                win = __shfl(map.incl, s) - 1 - j;
            } else {
                win = drawn;
            }
        }
        unsigned shard;
        uint32_t first, n_shard;
        rtq::locate(map, win, WAVE, shard, first, n_shard);
        const unsigned long long slot0 = (unsigned long long)shard * ta.shard_cap + first;
        const int n_req = (int)((ta.recs[slot0].flags >> GATHER_COUNT_SHIFT) & 127u);
        const bool have = lane < n_req;
        const GatherRec& r = ta.recs[slot0 + (have ? lane : 0)];
        float qx = (float)r.pos[0], qy = (float)r.pos[1], qz = (float)r.pos[2];
        const float hx = qx, hy = qy, hz = qz;      // the hit point (a volume phase asks elsewhere)
        const uint32_t flags = r.flags & 0xFFu;
        // Subsurface requests: their surface estimate is one of four gathers (main.cpp:260-388) and the only one in this
        // map; it is formed here with the group's.  The subsurface instantiation answers the other three as well; otherwise,
        // and where it cannot, the estimate is written into the record and the request handed on for the rest.
        const bool sss = (flags & GATHER_SSS) != 0;
        const bool keep_whole = SSS && __ballot(have && sss) != 0;      // (a window with subsurface requests stays with this wave: see the phases)
        if (two_sweeps && !part && keep_whole != first_sweep) continue;      // (the other sweep's)
        span_windows++;
        bool redo = have && ((flags & ~(GATHER_SSS | GATHER_MARCH)) != 0 || (sss && (!SSS || !ta.sss_surface)) || ta.all_redo || pm.n >= (1ll << 24));      // (subsurface requests: the subsurface instantiation's)
        bool todo = have && !redo;
        if (part) { redo = false; todo = have && ((c_mask >> lane) & 1ull) != 0; }      // (the parts before this one saw to the other lanes)
        if (PROF) pr[TP_WINDOWS]++;
        TILE_COUNT(TP_R_FLAGS, redo);
        const unsigned long long t_window = PROF ? clock64() : 0ull;
        int rounds_done = 0;
        // A window mostly holds one patch of neighbouring hit points, but it may hold a few (a tile across a silhouette):
        // each round takes the requests close to one of those still waiting; what is left after the last round, and
        // rounds of only a few requests, go to the per-request kernel.
        const int ROUNDS = gp.max_rounds;
        // (the fewest requests worth a round: measured on the water frame with windows taken late-first and parts handed on --
        // k = 1000: 1 / 2 / 4 / 6 / 8 / 12 requests give 8.3 / 8.1 / 8.0 / 7.4 / 7.5 / 7.6 ms; k = 100: 1 / 2 / 4 / 6 / 8 give 4.7 / 4.8 / 4.9 / 5.0 / 5.1)
        const int MIN_LANES = gp.min_lanes > 0 ? gp.min_lanes : (gp.k >= 400 ? 6 : 1);
        // subsurface requests between their phases: the surface estimate is formed (sss_wait), the volume terms so far
        const bool volume_here = SSS && ta.sss_surface > 1;
        bool sss_wait = false;
        int tries = 0;      // rounds begun, all phases together
        int work = 0;       // steps of 64 photons per pass, all rounds together
        int w_attempts = 0; // guesses at a centre's radius, all rounds together
        float vol_r_here = -1.f, vol_cx = 0.f, vol_cy = 0.f, vol_cz = 0.f;      // the last volume round of this window: its centre and k-th distance
        double surf0 = 0, surf1 = 0, surf2 = 0, vol0 = 0, vol1 = 0, vol2 = 0;
        for (int phase = 0; phase < (SSS ? 4 : 1); phase++) {
            const bool volume = SSS && phase > 0;
            double e_pass = 1.0;                        // e^pass of main.cpp:371
            if (volume) {
                // phase 1: around the hit point (main.cpp:310); 2 and 3: the march of main.cpp:337-376, where refraction is possible
                const bool in_phase = volume_here && sss_wait && !redo && vm.n > 0 && vm.n < (1ll << 24) && (phase == 1 || (flags & GATHER_MARCH) != 0);
                if (!__ballot(in_phase)) break;
                todo = in_phase;
                if (phase >= 2) {
                    double px = r.pos[0] + r.march[0] * 0.02, py = r.pos[1] + r.march[1] * 0.02, pz = r.pos[2] + r.march[2] * 0.02;      // main.cpp:341
                    e_pass = 2.71828;
                    if (phase == 3) {
                        // (the step length as answer_request draws it: keyed by pixel, hit point and pass)
                        const uint32_t where = __float_as_uint(hx) * 0x9E3779B1u ^ __float_as_uint(hy) * 0x85EBCA77u ^ __float_as_uint(hz) * 0xC2B2AE3Du;
                        const double d = -log(unit_random(r.pixel, where + 1u, gp.seed)) / 200;      // main.cpp:374
                        px += r.march[0] * d; py += r.march[1] * d; pz += r.march[2] * d;
                        e_pass *= 2.71828;
                    }
                    qx = (float)px; qy = (float)py; qz = (float)pz;
                }
            }
            const PhotonMapView& map = volume ? vm : pm;
            for (int round = (phase == 0 ? round_first : 0); round < ROUNDS; round++) {
                const unsigned long long m_todo = __ballot(todo);
                if (!m_todo) break;
                // ---- the round's centre: the middle one of the requests waiting
                int c_lane = 0;
                {
                    const int want_rank = __builtin_popcountll(m_todo) / 2;
                    unsigned long long mm = m_todo;
                    for (int i = 0; i < want_rank; i++) mm &= mm - 1;
                    c_lane = __builtin_ctzll(mm);
                }
                const float cx = bcast(qx, c_lane), cy = bcast(qy, c_lane), cz = bcast(qz, c_lane);
                const float ddx = qx - cx, ddy = qy - cy, ddz = qz - cz;
                const float d_c = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
                // a first look at who is near, with the radius unknown: if only a few are within the widest possible ball they
                // are not worth a round of their own
                const bool near = todo && d_c <= r_max;
                if (__builtin_popcountll(__ballot(near)) < MIN_LANES) {
                    TILE_COUNT(TP_R_FEW, near);
                    redo = redo || near;
                    todo = todo && !near;
                    continue;
                }
                // A window of subsurface requests is one wave's from its first phase to its last, and the longest such window is the
                // least the kernel takes: past S_SSS_TRIES rounds (four phases of one round each is the common case, of two across a
                // silhouette), or S_SSS_STEPS steps of 64 photons streamed by a pass (120 in the common case; several times that where
                // the balls are wide -- the unlit side of an object, whose few photons lie far apart), what still waits goes to the
                // per-request kernel, which spreads it over as many waves as there are requests.
                if (keep_whole && (++tries > S_SSS_TRIES || work > S_SSS_STEPS)) {
                    TILE_COUNT(TP_R_LEFT, todo);
                    redo = redo || todo;
                    todo = false;
                    break;
                }
                // ---- 1. the centre's k-th distance r_c (< 0: fewer than k photons within the search radius)
                float r_c = -1.f, R_found = 0.f;
                int n_pack = 0;
                bool found = false;
                {
                    const unsigned long long t_kth = PROF ? clock64() : 0ull;
                    // The guess only has to be within a factor of a few: the photons actually counted inside it correct it
                    // (density ~ count / R^2 on a surface), so a second walk, where one is needed, starts from a good radius.
                    float guess;
                    if (volume && phase >= 2 && vol_r_here > 0.f) {
                        // (the march moves the centre a few hundredths: the radius found a phase ago, and what the move can add)
                        const float mx = cx - vol_cx, my = cy - vol_cy, mz = cz - vol_cz;
                        guess = (vol_r_here + sqrtf(mx * mx + my * my + mz * mz)) * 1.02f;
                    } else {
                        TILE_PROF(TP_C_GUESS, guess = radius_guess(map, cx, cy, cz, gp.k, volume) * 1.15f);
                    }
                    const float prev = volume ? prev_r_vol : prev_r;
                    if (!(guess > 0.f)) { guess = prev > 0.f ? prev * 1.2f : r_max * 0.3f; if (PROF) pr[TP_A_NOGUESS]++; }
                    float too_small = 0.f, too_large = 3.0e38f;      // radii that held fewer than k photons / overflowed the lists
                    for (int attempt = 0; attempt < (volume ? S_ATTEMPTS_VOLUME : S_ATTEMPTS) && !found; attempt++) {
                        if (PROF) pr[TP_N_HIST]++;
                        // (a subsurface window: no more than S_SSS_ATTEMPTS walks, all its rounds together -- where the balls are so wide
                        // that the lists barely hold them every walk is a long one, and a failed round costs several)
                        if (keep_whole && ++w_attempts > S_SSS_ATTEMPTS) break;
                        const float R = fminf(guess, r_max);
                        const bool whole = !(R < r_max);                            // the search radius itself: nothing larger to try
                        const bool cand = near && d_c <= R;                         // who can be in the round if r_c <= R
                        // every such request's ball lies within its point +- rad (its bound never exceeds the search radius)
                        const float rad = fminf(R * (1.f + gp.opt_range) * 1.002f + 1e-5f, r_max * 1.0002f + 2e-6f);
                        const float big = 3.0e38f;
                        const float glx = wave_min_f(cand ? qx - rad : big), gly = wave_min_f(cand ? qy - rad : big), glz = wave_min_f(cand ? qz - rad : big);
                        const float ghx = wave_max_f(cand ? qx + rad : -big), ghy = wave_max_f(cand ? qy + rad : -big), ghz = wave_max_f(cand ? qz + rad : -big);
                        const float reach = wave_max_f(cand ? d_c : 0.f) + rad;
                        __builtin_amdgcn_wave_barrier();
                        int lists;
                        TILE_PROF(TP_C_WALK, lists = stream_walk(map, glx, gly, glz, ghx, ghy, ghz, cx, cy, cz, reach * reach * 1.0001f));
                        if (lists < 0) {
                            if (PROF) pr[TP_A_LISTS]++;
                            // too much of the tree: a smaller radius, unless that is known to hold fewer than k photons
                            too_large = fminf(too_large, R);
                            // (the photons in reach go with the radius squared on a surface, cubed in a volume: a smaller step back there)
                            guess = fmaxf(R * (volume ? 0.8f : 0.6f), sqrtf(too_small * R));
                            if (guess <= too_small * 1.02f || guess >= R * 0.98f) break;
                            continue;
                        }
                        const int n_leaf = lists & 0xFFFF, n_piv = lists >> 16;
                        const float lim = whole ? gp.r2 : R * R;
                        int packed, inside = 0;
                        TILE_PROF(TP_C_PACK, packed = pack_listed(map.pos, pack, ta.pack_cap, cx, cy, cz, reach * reach * 1.0001f, lim, n_leaf, n_piv, &inside));
                        __builtin_amdgcn_wave_barrier();
                        if (packed < 0) {
                            if (PROF) pr[TP_A_PACK]++;
                            too_large = fminf(too_large, R);
                            guess = fmaxf(R * (volume ? 0.85f : 0.7f), sqrtf(too_small * R));
                            if (guess <= too_small * 1.02f || guess >= R * 0.98f) break;
                            continue;
                        }
                        if (inside < gp.k && !whole) {
                            // too few: the radius that would hold k at the density just counted, and a margin
                            if (PROF) pr[TP_A_FEW]++;
                            too_small = fmaxf(too_small, R);
                            const float ratio = (float)gp.k / (float)(inside > 0 ? inside : 1);
                            const float grow = (volume ? cbrtf(ratio) : sqrtf(ratio)) * 1.15f;      // (the count grows with the radius squared on a surface, cubed in a volume)
                            guess = fminf(R * fminf(grow, 3.f), too_large < 3.0e38f ? sqrtf(R * too_large) : 3.0e38f);
                            if (guess >= too_large * 0.98f || guess <= R * 1.02f) break;
                            continue;
                        }
                        if (PROF && inside > 2 * gp.k) pr[TP_A_MANY]++;      // (more than needed: the filter below cuts the list down)
                        uint32_t bits = 0;
                        if (inside >= gp.k) {
                            TILE_PROF(TP_C_KTH, bits = centre_select(pack, packed, cx, cy, cz, lim, gp.k));
                            if (bits == 0xFFFFFFFFu) {
                                int total = 0;
                                bits = centre_kth(pack, packed, cx, cy, cz, lim, gp.k, &total);
                            }
                            __builtin_amdgcn_wave_barrier();
                        }
                        n_pack = packed;
                        found = true;
                        R_found = R;
                        r_c = inside >= gp.k ? sqrtf(__uint_as_float(bits)) : -1.f;      // (fewer than k: only within the search radius itself)
                    }
                    if (PROF) pr[TP_KTH] += clock64() - t_kth;
                }
                if (!found) {      // the lists cannot hold what the centre's ball touches
                    TILE_COUNT(TP_R_WALK, near);
                    redo = redo || near;
                    todo = todo && !near;
                    continue;
                }
                bool act = near;
                float bound = gp.r2, lower = 0.f;
                if (r_c >= 0.f) {
                    if (volume) { prev_r_vol = r_c; vol_r_here = r_c; vol_cx = cx; vol_cy = cy; vol_cz = cz; } else prev_r = r_c;
                    // this round: the requests inside the centre's own ball and a little beyond, as far as the packed list reaches
                    act = todo && d_c <= fminf(r_c * gp.act_reach, R_found);
                    // r_k(q) lies within |q - q_c| of r_c (triangle inequality).  For a request far from the centre that
                    // range is wide while the photon density, and with it r_k, changes little over a ball's width: such a
                    // request is given the range r_c * (1 -+ opt_range) instead.  Both ends are checked by the counts below
                    // (k photons inside the outer radius, fewer than k inside the inner one); a request that fails either
                    // goes to the per-request kernel.
                    const float dd = fminf(d_c, gp.opt_range * r_c);
                    const float reach = r_c + dd;
                    const float b = reach * reach * 1.001f + 1e-12f;      // slack for the rounding of the distances
                    if (b < bound) {
                        bound = b;
                        // and r_k(q) >= r_c - |q - q_c|: closer photons belong to the set whatever their order
                        const float inner = fmaxf(r_c - dd, 0.f);
                        lower = inner * inner * 0.999f;
                    }
                }
                if (__builtin_popcountll(__ballot(act)) < MIN_LANES) {
                    // (the centre itself is always among them, so every round retires at least one request)
                    TILE_COUNT(TP_R_FEW, act);
                    redo = redo || act;
                    todo = todo && !act;
                    continue;
                }
                todo = todo && !act;
                // The requests still waiting become a part of their own, for whichever wave is free first (this one, if none is
                // sooner): a window of several patches -- a tile across a silhouette -- is then worked on by several waves at once
                // instead of holding one wave, and the kernel's end, for several rounds.  What a round computes depends on the
                // window and the lanes waiting alone, so the results do not depend on who takes the part.
                {
                    const unsigned long long m_left = __ballot(todo);
                    if (sharing && m_left && round + 1 < ROUNDS && !keep_whole) {
                        unsigned long long at = ~0ull;
                        if (lane == 0) {
                            at = atomicAdd(my_cont + 2, 1ull);
                            if (at < ta.cont_cap) {
                                // (one part more to finish before the window counts as done: noted before this part is)
                                atomicAdd(ta.windows_done, ~0ull);
                                __hip_atomic_store(my_cont + 4 + 2 * at, m_left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                __hip_atomic_store(my_cont + 5 + 2 * at, (unsigned long long)win | ((unsigned long long)(round + 1) << 32) | (1ull << 63), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
                                atomicAdd(my_cont + 1, 1ull);
                            }
                        }
                        at = bcast64(at, 0);
                        if (at < ta.cont_cap) todo = false;      // (a full list: the rounds stay here)
                    }
                }
                // ---- 2. the packed list cut down to the box around the round's balls and the centre's reach: what it holds now
                // depends on the window alone
                {
                    const float rad = act ? sqrtf(bound) * 1.0001f + 1e-6f : 0.f;
                    const float big = 3.0e38f;
                    Box6f bx;
                    bx.lx = wave_min_f(act ? qx - rad : big); bx.ly = wave_min_f(act ? qy - rad : big); bx.lz = wave_min_f(act ? qz - rad : big);
                    bx.hx = wave_max_f(act ? qx + rad : -big); bx.hy = wave_max_f(act ? qy + rad : -big); bx.hz = wave_max_f(act ? qz + rad : -big);
                    const float reach = wave_max_f(act ? d_c + rad : 0.f);
                    if (PROF) pr[TP_N_CRIT] += n_pack;
                    TILE_PROF(TP_WALK, n_pack = pack_refilter(pack, n_pack, cx, cy, cz, reach * reach * 1.0001f, bx));
                    __builtin_amdgcn_wave_barrier();
                }
                const int n_steps = (n_pack + WAVE - 1) / WAVE;
                work += n_steps;
                rounds_done++;
                if (PROF) pr[TP_ROUNDS]++; if (PROF) pr[TP_LANES] += __builtin_popcountll(__ballot(act)); pr[TP_STEPS] += n_steps; if (PROF) pr[TP_N_SUM] += n_pack;

                const Shell sh = shell_of(bound, lower);
                const float radius2 = act && !(bound < gp.r2) ? gp.r2 : 3.0e38f;      // lanes bounded by the search radius itself
                const bool any_radius = __ballot(radius2 < 3.0e38f) != 0;
                // ---- 3a. per-lane histogram of the squared distances below the lane's bound
                if (any_radius) TILE_PROF(TP_HIST, stream_hist<true>(pack, n_pack, qx, qy, qz, sh, radius2));
                else TILE_PROF(TP_HIST, stream_hist<false>(pack, n_pack, qx, qy, qz, sh, radius2));
                uint32_t* const mine = hist_mine(cell, lane);
                // the bin holding the lane's k-th smallest distance (or its largest, if fewer than k lie within the bound)
                uint32_t total = 0;                                   // (the last bin holds what lies beyond the bound)
    #pragma unroll 4
                for (int b = 0; b < T_BINS - 1; b++) total += hist_count(mine, lane, b);
                const uint32_t k_eff = total < (uint32_t)gp.k ? total : (uint32_t)gp.k;
                // a bound derived from the neighbour must contain k photons; if it does not, the per-request kernel decides
                TILE_COUNT(TP_R_SHORT, act && total < (uint32_t)gp.k && bound < gp.r2);
                if (act && total < (uint32_t)gp.k && bound < gp.r2) { redo = true; act = false; }
                int crit = -1;
                uint32_t below = 0, in_bin = 0;
                {
                    uint32_t cum = 0;
    #pragma unroll 4
                    for (int b = 0; b < T_BINS - 1; b++) {
                        const uint32_t h = hist_count(mine, lane, b);
                        if (crit < 0 && k_eff > 0 && cum + h >= k_eff) { crit = b; below = cum; in_bin = h; }
                        cum += h;
                    }
                }
                // (k photons already inside the inner radius: the range was guessed too high -- the rigorous range never is)
                const bool crowded = act && k_eff > 0 && (in_bin > (uint32_t)T_CRIT || crit == 0);
                TILE_COUNT(TP_R_CROWD, crowded);
                if (crowded) { redo = true; act = false; }
                const bool some = act && k_eff > 0;
                const int need = (int)(k_eff - below);                 // 1-based rank of the k-th smallest inside its bin
                __builtin_amdgcn_wave_barrier();
                // ---- 3b. the distances of that bin, ranked
                int cnt;
                if (any_radius) TILE_PROF(TP_CRIT, cnt = stream_crit<true>(pack, n_pack, qx, qy, qz, sh, radius2, some ? crit : -1));
                else TILE_PROF(TP_CRIT, cnt = stream_crit<false>(pack, n_pack, qx, qy, qz, sh, radius2, some ? crit : -1));
                const unsigned long long t_rank = PROF ? clock64() : 0ull;
                const int max_cnt = (int)wave_max_f((float)cnt);
                uint32_t t_bits = 0;
                for (int i0 = 0; i0 < max_cnt; i0 += 8) {
                    // eight entries in registers per sweep over the lane's column
                    uint32_t e[8];
                    int rank[8];
    #pragma unroll
                    for (int u = 0; u < 8; u++) { e[u] = cell[(i0 + u < T_CRIT ? i0 + u : T_CRIT - 1) * WAVE + lane]; rank[u] = 0; }
    #pragma unroll 4
                    for (int j = 0; j < max_cnt; j++) {
                        const uint32_t o = cell[j * WAVE + lane];
                        const bool in_col = j < cnt;
    #pragma unroll
                        for (int u = 0; u < 8; u++) rank[u] += (in_col && (o < e[u] || (o == e[u] && j < i0 + u))) ? 1 : 0;
                    }
    #pragma unroll
                    for (int u = 0; u < 8; u++)
                        if (i0 + u < cnt && rank[u] == need - 1) t_bits = e[u];
                }
                int less = 0, equal = 0;
                for (int j = 0; j < max_cnt; j++) {
                    const uint32_t o = cell[j * WAVE + lane];
                    less += (j < cnt && o < t_bits) ? 1 : 0;
                    equal += (j < cnt && o == t_bits) ? 1 : 0;
                }
                const float t_k = __uint_as_float(t_bits);             // the k-th distance = maxDistSqr (main.cpp:266-267)
                const int want_eq = need - less;                       // photons at exactly that distance that belong to the set
                // Several photons at exactly the k-th distance of which only some belong: the per-request kernel decides which
                // (it has a rule that does not depend on the order the photons are met in); here every one at that distance counts.
                const bool split_tie = some && equal != want_eq;
                TILE_COUNT(TP_R_TIE, split_tie);
                if (split_tie) redo = true;
                const bool summed = some && !split_tie;
                __builtin_amdgcn_wave_barrier();
                if (PROF) pr[TP_RANK] += clock64() - t_rank;
                // ---- 3c. a volume phase: the step's sums over the set (main.cpp:310-334, 346-373)
                if (volume) {
                    VolSum vs;
                    TILE_PROF(TP_SUM, vs = stream_sum_volume(pack, n_pack, map.pos, map.aux, qx, qy, qz, t_k, summed));
                    if (summed) {
                        // normalised by the distance from the HIT POINT to the farthest photon of the set (main.cpp:317, 353)
                        const float dx = hx - vs.fx, dy = hy - vs.fy, dz = hz - vs.fz;
                        const float m = dx * dx + dy * dy + dz * dz;
                        const double scale = 1.0 / (e_pass * gp.photon_num / 8 * 3.1415926 * m * sqrtf(m));      // main.cpp:332, 371
                        vol0 += (r.color[0] * vs.s0) * 0.2 * scale; vol1 += (r.color[1] * vs.s1) * 0.2 * scale; vol2 += (r.color[2] * vs.s2) * 0.2 * scale;
                    }
                    __builtin_amdgcn_wave_barrier();
                    continue;
                }
                // ---- 3c. the estimate over the set (main.cpp:262-295)
                Sum3 sum;
                {
                    const float nx = (float)r.normal[0], ny = (float)r.normal[1], nz = (float)r.normal[2];
                    const unsigned long long m_sum = __ballot(summed);
                    const int l0 = m_sum ? __builtin_ctzll(m_sum) : 0;
                    const float ax = bcast(nx, l0), ay = bcast(ny, l0), az = bcast(nz, l0);
                    if (m_sum && !__ballot(summed && (nx != ax || ny != ay || nz != az)))
                        TILE_PROF(TP_SUM, sum = stream_sum<true>(pack, n_pack, pm.aux, qx, qy, qz, t_k, summed, ax, ay, az));
                    else
                        TILE_PROF(TP_SUM, sum = stream_sum<false>(pack, n_pack, pm.aux, qx, qy, qz, t_k, summed, nx, ny, nz));
                }
                if (summed) {
                    const double sc = 1.0 / (gp.norm * t_k);           // main.cpp:293
                    double rgb[3] = {r.color[0] * sum.s0 * sc, r.color[1] * sum.s1 * sc, r.color[2] * sum.s2 * sc};
                    if (SSS && sss) {
                        surf0 = rgb[0]; surf1 = rgb[1]; surf2 = rgb[2];
                    } else {
                        clip3(rgb);                                    // main.cpp:391
                        for (int c = 0; c < 3; c++) atomicAdd(&acc[3 * (size_t)r.pixel + c], r.weight * rgb[c]);
                    }
                }
                // (a subsurface request with no surface photon in reach has a surface estimate, too: zero)
                if (SSS && sss && act && !split_tie) sss_wait = true;
                __builtin_amdgcn_wave_barrier();
            }
            TILE_COUNT(TP_R_LEFT, todo);
            redo = redo || todo;
            todo = false;
        }
        if (SSS && sss_wait) {
            if (volume_here && !redo) {
                // every phase answered: surface + volume terms (main.cpp:389), clipped (main.cpp:391)
                double rgb[3] = {surf0 + vol0, surf1 + vol1, surf2 + vol2};
                clip3(rgb);
                for (int c = 0; c < 3; c++) atomicAdd(&acc[3 * (size_t)r.pixel + c], r.weight * rgb[c]);
            } else {
                // handed on with the surface estimate: the per-request kernel adds the volume terms
                GatherRec* const w = const_cast<GatherRec*>(&r);
                w->normal[0] = surf0; w->normal[1] = surf1; w->normal[2] = surf2;
                w->flags = r.flags | GATHER_SURFACE_DONE;
                redo = true;
            }
        }
        if (SSS && __ballot(sss_wait)) __threadfence();      // (a completed record is visible before the entry that hands it over: the kernel alongside reads it at once)
        if (PROF) {
            const unsigned long long dt = clock64() - t_window;
            if (rounds_done <= 1) { pr[TP_W_ROUNDS1]++; pr[TP_W_CYC1] += dt; }
            else if (rounds_done == 2) { pr[TP_W_ROUNDS2]++; pr[TP_W_CYC2] += dt; }
            else if (rounds_done == 3) { pr[TP_W_ROUNDS3]++; pr[TP_W_CYC3] += dt; }
            else { pr[TP_W_ROUNDS4]++; pr[TP_W_CYC4] += dt; }
            // (the longest window with what it did, in one word that a maximum can be taken of: cycles, rounds completed, attempts at a radius, steps / 4)
            const unsigned long long what = (dt << 24) | ((unsigned long long)(rounds_done & 255) << 16) | ((unsigned long long)(w_attempts & 255) << 8) | (unsigned long long)(work / 4 > 255 ? 255 : work / 4);
            if (what > pr[TP_W_MAX]) pr[TP_W_MAX] = what;
        }
        // ---- requests left to the per-request kernel
        const unsigned long long m_redo = __ballot(redo);
        if (m_redo) {
            unsigned long long at = 0;
            const int leader = __builtin_ctzll(m_redo);
            if (lane == leader) at = atomicAdd(ta.n_redo, (unsigned long long)__builtin_popcountll(m_redo));
            at = ((unsigned long long)(unsigned)__shfl((unsigned)(at >> 32), leader) << 32) | (unsigned)__shfl((unsigned)at, leader);
            if (redo) __hip_atomic_store(&ta.redo[at + prefix_rank(m_redo)], (uint32_t)(slot0 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __threadfence();                                        // the entries are visible before the window counts as done
        if (lane == 0) atomicAdd(ta.windows_done, 1ull);
        __builtin_amdgcn_wave_barrier();
    }
#undef TILE_PROF
#undef TILE_COUNT
    if (ta.span && lane == 0) {
        ta.span[3 * (size_t)blockIdx.x] = span_begin;
        ta.span[3 * (size_t)blockIdx.x + 1] = wall_clock64();
        ta.span[3 * (size_t)blockIdx.x + 2] = span_windows | ((unsigned long long)span_spins << 32);
    }
    if (PROF && lane == 0) {
        pr[TP_BUSY] = clock64() - t_begin;
#pragma unroll
        for (int i = 0; i < TP_BUSY_MAX; i++)
            if (i != TP_W_MAX && i != TP_W_FIRST_MIN && i != TP_W_LAST_MAX) atomicAdd(&ta.prof[i], pr[i]);
        atomicMax(&ta.prof[TP_BUSY_MAX], pr[TP_BUSY]);
        atomicMax(&ta.prof[TP_W_MAX], pr[TP_W_MAX]);
        atomicMax(&ta.prof[TP_W_LAST_MAX], (unsigned long long)clock64());                   // when the last wave ended
        atomicMin(&ta.prof[TP_W_FIRST_MIN], t_begin);                                        // when the first wave began (the slot starts at all ones)
    }
}

__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(3, 3))) void gather_stream_kernel(PhotonMapView pm, TileArgs ta, GatherParams gp, double* __restrict__ acc,
                                                                                                         float4* pack_all)
{
    gather_stream_body<false, false>(pm, pm, ta, gp, acc, pack_all);
}
// the same with the volume terms of subsurface requests (see gather_stream_body): launched for scenes that have such materials
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(3, 3))) void gather_stream_sss_kernel(PhotonMapView pm, PhotonMapView vm, TileArgs ta, GatherParams gp,
                                                                                                             double* __restrict__ acc, float4* pack_all)
{
    gather_stream_body<false, true>(pm, vm, ta, gp, acc, pack_all);
}
// (the accounting's registers would spill at three waves per SIMD and distort what it measures)
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(2, 2))) void gather_stream_prof_kernel(PhotonMapView pm, PhotonMapView vm, TileArgs ta, GatherParams gp,
                                                                                                              double* __restrict__ acc, float4* pack_all)
{
    gather_stream_body<true, true>(pm, vm, ta, gp, acc, pack_all);
}

}  // namespace rtg
