// K4: bounded k-NN photon gather + radiance estimate (KDTree::findKNN, Kdtree.cpp:73-158, and the
// estimate of main.cpp:260-295), one query per wave64.
//
// The reference walks a pointer kd-tree keeping a max-heap of k photon pointers.  Here the tree is
// implicit over one in-order array (see FlatPhotonMap: a subtree is a contiguous range, its pivot
// the middle element, ranges of <= 64 photons are leaves).  A serial walk would pay one dependent
// memory round trip per tree level and leaf, so the wave works on many subtrees at once:
//   1. it walks down the near side from the root to the first range of <= 2k photons, testing the
//      pivots on the way and remembering the far sides; that range is scanned linearly (coalesced,
//      64 photons per step, several steps in flight), which yields a first bound: the k-th
//      smallest distance found so far;
//   2. the remembered far sides form a frontier of ranges in LDS.  Each round the lanes take up
//      to 64 ranges from its top, drop those whose plane distance exceeds the current bound
//      (the pruning of Kdtree.cpp:121-132 with the up-to-date bound), test their pivots and put
//      the children back (far side only if its plane distance is within the bound); ranges that
//      are leaves are scanned by the whole wave as in step 1;
//   3. photons with d2 < bound go to a candidate buffer in LDS through a ballot prefix; when it is
//      nearly full, a wave-wide radix select keeps exactly the k nearest candidates and tightens
//      the bound to the k-th distance.
// The result is the exact k-nearest set within the radius (ties at the k-th distance arbitrary,
// as with the reference's heap); only max d2 and the weighted sum leave the kernel.
#pragma once
#include <hip/hip_runtime.h>

#include "rt_queue.hip.h"
#include <stdint.h>

namespace rtg {

constexpr int WAVE = 64;
constexpr int CAP = 2048;        // candidate buffer entries per wave
constexpr int GROUP = 4;         // 64-photon steps of a linear scan between two checks for buffer room
constexpr int FLIGHT = 16;       // 64-photon steps of a linear scan whose loads are in flight together
constexpr int MAX_K = CAP - GROUP * WAVE;
constexpr int STACK = 48;        // > log2(2^31 / 64) levels
constexpr int FRONT = 256;       // frontier capacity (ranges); beyond it subtrees are walked serially

struct PhotonMapView {
    const float4* pos;           // per photon, tree order: x, y, z, split axis of the node whose pivot it is (0/1/2 as a float)
    const float* aux;            // 6 per photon: power rgb, dir xyz
    const int* order;            // tree position -> index in the caller's photon array
    const float* leaf_box;       // 6 per slot: bounding box of the leaf whose first photon is 32 * slot .. 32 * slot + 31
    long long n;
};

// one gather request = one diffuse hit
struct GatherRec {
    double pos[3];
    double normal[3];
    double color[3];
    double weight;               // path throughput the estimate is scaled by
    uint32_t pixel;
    uint32_t flags;              // GATHER_SSS: subsurface material (main.cpp:297); GATHER_MARCH: refraction possible (main.cpp:337)
    double march[3];             // refracted direction the volume march follows (main.cpp:339)
};
constexpr uint32_t GATHER_SSS = 1u, GATHER_MARCH = 2u;
// A subsurface request whose surface estimate (main.cpp:260-295, before the clip) the group kernel has already formed: it
// stands in the record's `normal`, which nothing else needs any more; the per-request kernel adds the volume terms.
constexpr uint32_t GATHER_SURFACE_DONE = 4u;

// Optional cycle accounting (development aid): slots of the array a kernel is given.
enum ProfSlot { P_DESCENT, P_SCAN0, P_SELECT, P_N_SELECT, P_ROUNDS, P_N_ROUNDS, P_LEAVES, P_N_LEAVES, P_ESTIMATE, P_QUERIES, P_SLOTS };

// Per-wave working set in LDS.  The gather kernels run one wave per block, so one file-scope
// instance serves; helper functions that are not inlined address it directly.
struct WaveLds {
    float key[CAP];              // squared distances of the candidates
    uint32_t hist[256];
    int c_lo[WAVE], c_n[WAVE];      // steps of the current linear scan: first photon, count (<= 64)
    int s_lo[STACK], s_hi[STACK];   // stack of the serial walk
    float s_d2[STACK];
    // (everything above is free between two searches: the group form lays its per-lane tables over it)
    int f_lo[FRONT], f_hi[FRONT];   // frontier of pending subtrees: range and squared plane distance
    float f_d2[FRONT];
    unsigned long long prof[P_SLOTS];
};
__shared__ WaveLds g_L;
// Tree positions of the candidates, beside g_L.key.  Only code that needs to know WHICH photons are the nearest
// touches it (the helpers' IDS = true forms); a kernel that asks for the k-th distance alone does not pay its 8 KB.
__shared__ uint32_t g_idx[CAP];

struct Prof {
    unsigned long long* out;     // null: no accounting
    __device__ void init() const { if (out && (threadIdx.x & 63) < P_SLOTS) g_L.prof[threadIdx.x & 63] = 0; }
    __device__ void add(int slot, unsigned long long x) const { if (out && (threadIdx.x & 63) == 0) g_L.prof[slot] += x; }
    __device__ unsigned long long now() const { return out ? (unsigned long long)clock64() : 0ull; }
    __device__ void flush() const { if (out && (threadIdx.x & 63) < P_SLOTS) atomicAdd(&out[threadIdx.x & 63], g_L.prof[threadIdx.x & 63]); }
};

// candidate-set state carried through the helpers by value
struct Cand {
    int count;       // entries in g_L.key / g_idx
    float bound;     // candidates need d2 < bound
    int tight;       // the keys reach up to the bound (a select or a neighbour's hint set it): the
                     // select may bin over [0, bound] without looking for the largest key first.  Such a bound
                     // is INCLUSIVE: a photon found later at exactly the k-th distance is still a candidate, so
                     // which of several photons at that distance stay does not depend on the order they are met in
                     // (select_k decides by tree position).  The search radius itself is exclusive (Kdtree.cpp:77,82).
};
__device__ __forceinline__ bool within(float d2, const Cand& c) { return c.tight ? d2 <= c.bound : d2 < c.bound; }

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ unsigned prefix_rank(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// k-th smallest (1-based) of key[0..count): wave-wide MSB-first radix select on the float bits
// (all keys are non-negative, so the unsigned bit pattern orders like the value).  Afterwards the
// buffer holds exactly the k smallest entries.  Returns the k-th key.
// Find the owner of the `want`-th element in a 256-bin histogram held in L.hist: returns the bin,
// the number of elements in lower bins and the bin's own count.  All lanes get the same values.
__device__ __forceinline__ void locate_bin(uint32_t want, uint32_t& bin, uint32_t& below, uint32_t& in_bin)
{
    const WaveLds& L = g_L;
    const int lane = lane_id();
    // lane l owns bins 4l .. 4l+3
    uint32_t h0 = L.hist[4 * lane], h1 = L.hist[4 * lane + 1], h2 = L.hist[4 * lane + 2], h3 = L.hist[4 * lane + 3];
    uint32_t s = h0 + h1 + h2 + h3;
    uint32_t incl = s;
    for (int off = 1; off < WAVE; off <<= 1) {
        uint32_t v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    uint32_t excl = incl - s;
    bool mine = want > excl && want <= incl;   // exactly one lane
    uint32_t b = 0, bl = 0, ib = 0;
    if (mine) {
        uint32_t w = want - excl;
        if (w <= h0) { b = 0; bl = excl; ib = h0; }
        else if (w <= h0 + h1) { b = 1; bl = excl + h0; ib = h1; }
        else if (w <= h0 + h1 + h2) { b = 2; bl = excl + h0 + h1; ib = h2; }
        else { b = 3; bl = excl + h0 + h1 + h2; ib = h3; }
        b += 4 * lane;
    }
    int src = __builtin_ctzll(__ballot(mine));
    bin = __shfl(b, src);
    below = __shfl(bl, src);
    in_bin = __shfl(ib, src);
}

// Bit pattern of the k-th smallest key by most-significant-digit radix passes (any key distribution).
// `want` returns how many entries equal to it belong to the k smallest, `n_eq` how many entries equal it.
__device__ uint32_t radix_threshold(int count, int k, int& want, int& n_eq)
{
    WaveLds& L = g_L;
    const int lane = lane_id();
    uint32_t prefix = 0, pmask = 0;
    want = k;
    for (int pass = 3; pass >= 0; pass--) {
        for (int b = lane; b < 256; b += WAVE) L.hist[b] = 0;
        __builtin_amdgcn_wave_barrier();
        for (int j = lane; j < count; j += WAVE) {
            uint32_t u = __float_as_uint(L.key[j]);
            if ((u & pmask) == prefix) atomicAdd(&L.hist[(u >> (8 * pass)) & 255u], 1u);
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t bin, below, in_bin;
        locate_bin((uint32_t)want, bin, below, in_bin);
        want -= (int)below;
        n_eq = (int)in_bin;                 // (after the last pass: the entries with exactly this bit pattern)
        prefix |= bin << (8 * pass);
        pmask |= 255u << (8 * pass);
        __builtin_amdgcn_wave_barrier();
    }
    return prefix;
}

// Keep the k smallest of L.key[0..count) (count > k).  Squared distances of points spread over a
// surface are nearly uniformly distributed, so one histogram with 256 bins linear in the key value
// leaves only a handful of entries in the bin that holds the k-th smallest; those are ranked
// directly.  (Bins are monotonic in the key, so every entry of a lower bin is smaller.)  Crowded
// bins -- many equal keys -- fall back to the radix passes.  Returns the k-th key.
template <bool IDS>
__device__ __attribute__((noinline)) Cand select_k(int count, int k, float key_limit)
{
    WaveLds& L = g_L;
    const int lane = lane_id();
    // every pass reads UNR x 64 keys before using them, so the LDS latency is paid once per 512 entries
    constexpr int UNR = 8;
    float mx = key_limit;          // > 0: no key exceeds it
    if (!(mx > 0.f)) {
        mx = 0.f;
        for (int base = 0; base < count; base += UNR * WAVE) {
            float kv[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) { int j = base + u * WAVE + lane; kv[u] = j < count ? L.key[j] : 0.f; }
#pragma unroll
            for (int u = 0; u < UNR; u++) mx = fmaxf(mx, kv[u]);
        }
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    }
    const float scale = mx > 0.f ? 255.5f / mx : 0.f;
    for (int b = lane; b < 256; b += WAVE) L.hist[b] = 0;
    __builtin_amdgcn_wave_barrier();
    for (int base = 0; base < count; base += UNR * WAVE) {
        float kv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) { int j = base + u * WAVE + lane; kv[u] = j < count ? L.key[j] : -1.f; }
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            if (kv[u] >= 0.f) {
                int b = (int)(kv[u] * scale);
                atomicAdd(&L.hist[b > 255 ? 255 : b], 1u);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t bin, below, in_bin;
    locate_bin((uint32_t)k, bin, below, in_bin);
    __builtin_amdgcn_wave_barrier();
    uint32_t T;
    int want, n_eq = 0;
    if (in_bin <= (uint32_t)WAVE) {
        // the bin's entries, in buffer order, into L.hist (its counts are no longer needed)
        int m = 0;
        for (int base = 0; base < count; base += UNR * WAVE) {
            float kv[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) { int j = base + u * WAVE + lane; kv[u] = j < count ? L.key[j] : -1.f; }
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                int b = (int)(kv[u] * scale);
                bool hit = kv[u] >= 0.f && (b > 255 ? 255 : b) == (int)bin;
                unsigned long long mh = __ballot(hit);
                if (hit) L.hist[m + (int)prefix_rank(mh)] = __float_as_uint(kv[u]);
                m += __builtin_popcountll(mh);
            }
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t e = lane < m ? L.hist[lane] : 0xFFFFFFFFu;
        int rank = 0;
        for (int j = 0; j < m; j++) {
            uint32_t kj = L.hist[j];
            rank += (kj < e || (kj == e && j < lane)) ? 1 : 0;
        }
        const int need = k - (int)below;            // 1-based position of the k-th smallest inside the bin
        int src = __builtin_ctzll(__ballot(lane < m && rank == need - 1));
        T = __shfl(e, src);
        want = __builtin_popcountll(__ballot(lane < m && e == T && rank < need));
        n_eq = __builtin_popcountll(__ballot(lane < m && e == T));
        __builtin_amdgcn_wave_barrier();
    } else {
        T = radix_threshold(count, k, want, n_eq);
    }
    // Several photons at exactly the k-th distance of which only some belong to the set (float distances of photons a
    // rounding error apart coincide): those at the lowest tree positions stay, whatever order the walk met them in.
    bool by_position = false;
    uint32_t last_position = 0;
    if (IDS && n_eq > want && n_eq <= 256) {
        int m = 0;
        for (int base = 0; base < count; base += WAVE) {
            const int j = base + lane;
            const bool eq = j < count && __float_as_uint(L.key[j]) == T;
            const unsigned long long me = __ballot(eq);
            if (eq) L.hist[m + (int)prefix_rank(me)] = g_idx[j];
            m += __builtin_popcountll(me);
        }
        __builtin_amdgcn_wave_barrier();
        bool found = false;
        uint32_t mine = 0;
        for (int t = lane; t < n_eq; t += WAVE) {
            const uint32_t p = L.hist[t];
            int rank = 0;
            for (int j = 0; j < n_eq; j++) rank += L.hist[j] < p ? 1 : 0;
            if (rank == want - 1) { found = true; mine = p; }
        }
        const unsigned long long mf = __ballot(found);      // tree positions are distinct: exactly one lane
        if (mf) { last_position = (uint32_t)__shfl((int)mine, __builtin_ctzll(mf)); by_position = true; }
        __builtin_amdgcn_wave_barrier();
    }
    // T = bit pattern of the k-th smallest key; `want` = how many entries equal to T to keep
    // stable in-place compaction (outputs never pass the read cursor: a block of entries is in registers
    // before any of its outputs is written)
    int out = 0, eq_taken = 0;
    for (int base = 0; base < count; base += UNR * WAVE) {
        uint32_t ku[UNR], id[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            int j = base + u * WAVE + lane;
            ku[u] = j < count ? __float_as_uint(L.key[j]) : 0xFFFFFFFFu;   // larger than any key: never kept
            id[u] = IDS && j < count ? g_idx[j] : 0u;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            bool less = ku[u] < T;
            bool eq = ku[u] == T;
            unsigned long long meq = __ballot(eq);
            bool keep = less || (eq && (by_position ? id[u] <= last_position : (eq_taken + (int)prefix_rank(meq)) < want));
            unsigned long long mk = __ballot(keep);
            if (keep) {
                int p = out + (int)prefix_rank(mk);
                L.key[p] = __uint_as_float(ku[u]);
                if (IDS) g_idx[p] = id[u];
            }
            out += __builtin_popcountll(mk);
            eq_taken += __builtin_popcountll(meq);
        }
        __builtin_amdgcn_wave_barrier();
    }
    return Cand{out, __uint_as_float(T), 1};
}

template <bool IDS>
__device__ __forceinline__ void append(int& count, bool take, float d2, uint32_t id)
{
    WaveLds& L = g_L;
    unsigned long long m = __ballot(take);
    if (take) {
        int p = count + (int)prefix_rank(m);
        L.key[p] = d2;
        if (IDS) g_idx[p] = id;
    }
    count += __builtin_popcountll(m);
}

// Make room for `need` more candidates (need <= GROUP * WAVE): if they might not fit, keep the k
// nearest and tighten the bound.  count > k holds whenever this triggers (k <= MAX_K).
template <bool IDS>
__device__ __forceinline__ void make_room(Cand& c, int k, int need, Prof pf)
{
    // also once the set has grown to twice k (small k): an early, cheap select tightens the bound
    // long before the buffer fills
    const int early = k + (k > GROUP * WAVE ? k : GROUP * WAVE);
    if (c.count > CAP - need || c.count > early) {
        __builtin_amdgcn_wave_barrier();
        unsigned long long t = pf.now();
        c = select_k<IDS>(c.count, k, c.tight ? c.bound : 0.f);
        pf.add(P_SELECT, pf.now() - t); pf.add(P_N_SELECT, 1);
    }
}

// Linear scan of the n_steps (<= 64) steps listed in L.c_lo / L.c_n.  The loads of FLIGHT steps are
// issued together (one memory round trip per FLIGHT x 64 photons); room in the candidate buffer is
// checked before every GROUP steps.
template <bool IDS>
__device__ __attribute__((noinline)) Cand scan_steps(PhotonMapView pm, float qx, float qy, float qz, int n_steps, int k, Cand c, Prof pf)
{
    const WaveLds& L = g_L;
    const int lane = lane_id();
    for (int g = 0; g < n_steps; g += FLIGHT) {
        float px[FLIGHT], py[FLIGHT], pz[FLIGHT];
        int id[FLIGHT];
        bool valid[FLIGHT];
        // every load is unconditional (step and lane clamped to valid ones): a load inside a divergent
        // branch is waited for at the branch's join, which would serialise the round trips
#pragma unroll
        for (int u = 0; u < FLIGHT; u++) {
            const int step = g + u < n_steps ? g + u : n_steps - 1;
            const int lo = L.c_lo[step], n = L.c_n[step];
            valid[u] = g + u < n_steps && lane < n;
            id[u] = lo + (lane < n ? lane : (n > 0 ? n - 1 : 0));     // n >= 1 for every listed step
            const float4 p = pm.pos[id[u]];
            px[u] = p.x; py[u] = p.y; pz[u] = p.z;
        }
#pragma unroll
        for (int v = 0; v < FLIGHT; v += GROUP) {
            if (g + v < n_steps) {
                make_room<IDS>(c, k, GROUP * WAVE, pf);
#pragma unroll
                for (int w = 0; w < GROUP; w++) {
                    const int u = v + w;
                    float d2 = 3.0e38f;
                    if (valid[u]) {
                        float dx = qx - px[u], dy = qy - py[u], dz = qz - pz[u];
                        d2 = dx * dx + dy * dy + dz * dz;
                    }
                    append<IDS>(c.count, valid[u] && within(d2, c), d2, (uint32_t)id[u]);
                }
            }
        }
    }
    return c;
}

// Linear scan of one contiguous range (a whole subtree), any length.
template <bool IDS>
__device__ __forceinline__ Cand scan_range(const PhotonMapView& pm, float qx, float qy, float qz, int lo, int hi, int k, Cand c, Prof pf)
{
    WaveLds& L = g_L;
    const int lane = lane_id();
    while (lo < hi) {
        int steps = (hi - lo + WAVE - 1) / WAVE;
        if (steps > WAVE) steps = WAVE;
        __builtin_amdgcn_wave_barrier();
        if (lane < steps) {
            int first = lo + lane * WAVE;
            L.c_lo[lane] = first;
            L.c_n[lane] = hi - first < WAVE ? hi - first : WAVE;
        }
        __builtin_amdgcn_wave_barrier();
        c = scan_steps<IDS>(pm, qx, qy, qz, steps, k, c, pf);
        lo += steps * WAVE;
    }
    return c;
}

// Serial depth-first walk of one subtree (all lanes in step); used when the frontier is full.
template <bool IDS>
__device__ __attribute__((noinline)) Cand walk_serial(PhotonMapView pm, float qx, float qy, float qz, int lo, int hi, int k, Cand c, Prof pf)
{
    WaveLds& L = g_L;
    const int lane = lane_id();
    int sp = 0;
    float dfar = 0.f;
    bool have = true;
    while (true) {
        if (!have) {
            if (sp == 0) break;
            sp--;
            lo = L.s_lo[sp]; hi = L.s_hi[sp]; dfar = L.s_d2[sp];
            if (dfar > c.bound) continue;          // Kdtree.cpp:123-125: far side only if delta^2 <= bound
        }
        have = false;
        while (hi - lo > WAVE) {
            int mid = lo + (hi - lo) / 2;
            const float4 pv = pm.pos[mid];
            const int ax = (int)pv.w;
            float px = pv.x, py = pv.y, pz = pv.z;
            float dx = qx - px, dy = qy - py, dz = qz - pz;
            float d2 = dx * dx + dy * dy + dz * dz;
            make_room<IDS>(c, k, WAVE, pf);
            append<IDS>(c.count, lane == 0 && within(d2, c), d2, (uint32_t)mid);   // the pivot itself (Kdtree.cpp:119)
            float split = ax == 0 ? px : (ax == 1 ? py : pz);
            float delta = (ax == 0 ? qx : (ax == 1 ? qy : qz)) - split;
            int nlo, nhi, flo, fhi;
            if (delta < 0.f) { nlo = lo; nhi = mid; flo = mid + 1; fhi = hi; }      // Kdtree.cpp:121
            else { nlo = mid + 1; nhi = hi; flo = lo; fhi = mid; }
            float dd = delta * delta;
            if (fhi > flo && dd <= c.bound && sp < STACK) {
                if (lane == 0) { L.s_lo[sp] = flo; L.s_hi[sp] = fhi; L.s_d2[sp] = dd; }
                sp++;
            }
            lo = nlo; hi = nhi;
        }
        c = scan_range<IDS>(pm, qx, qy, qz, lo, hi, k, c, pf);
        __builtin_amdgcn_wave_barrier();
    }
    return c;
}

// k nearest photons within sqrt(r2) of one query held identically by all lanes.  Returns the
// number of photons found; g_L.key / g_idx[0..count) is the set (IDS = false: the distances only).
// `hint` > 0: a bound known to contain at least k photons (from a neighbouring query, see
// knn_collect); then steps 1's scan is skipped and the whole tree goes through the frontier.
template <bool IDS>
__device__ __forceinline__ Cand knn_collect_from(const PhotonMapView& pm, float qx, float qy, float qz, int k, float r2, float hint, int front_cap,
                                                 Prof pf)
{
    WaveLds& L = g_L;
    const int lane = lane_id();
    if (pm.n <= 0) return Cand{0, r2, 0};
    Cand c{0, r2, 0};            // candidates need d2 < bound (Kdtree.cpp:77,82: strict)
    const bool hinted = hint > 0.f && hint < r2;
    if (hinted) { c.bound = hint; c.tight = 1; }

    // ---- 1. near side down to a range of <= 2k photons, far sides onto the frontier
    int nf = 0;
    int lo = 0, hi = (int)pm.n;
    unsigned long long t0 = pf.now();
    const int limit = 2 * k > WAVE ? 2 * k : WAVE;
    if (hinted) {
        if (lane == 0) { L.f_lo[0] = 0; L.f_hi[0] = hi; L.f_d2[0] = 0.f; }
        nf = 1;
        lo = hi = 0;
    }
    while (hi - lo > limit) {
        int mid = lo + (hi - lo) / 2;
        const float4 pv = pm.pos[mid];
        const int ax = (int)pv.w;
        float px = pv.x, py = pv.y, pz = pv.z;
        float dx = qx - px, dy = qy - py, dz = qz - pz;
        float d2 = dx * dx + dy * dy + dz * dz;
        append<IDS>(c.count, lane == 0 && within(d2, c), d2, (uint32_t)mid);   // the pivot itself (Kdtree.cpp:119); < 32 of these
        float split = ax == 0 ? px : (ax == 1 ? py : pz);
        float delta = (ax == 0 ? qx : (ax == 1 ? qy : qz)) - split;
        int nlo, nhi, flo, fhi;
        if (delta < 0.f) { nlo = lo; nhi = mid; flo = mid + 1; fhi = hi; }      // Kdtree.cpp:121
        else { nlo = mid + 1; nhi = hi; flo = lo; fhi = mid; }
        float dd = delta * delta;
        if (fhi > flo && dd <= c.bound) {          // at most one per level: nf < 32 <= FRONT
            if (lane == 0) { L.f_lo[nf] = flo; L.f_hi[nf] = fhi; L.f_d2[nf] = dd; }
            nf++;
        }
        lo = nlo; hi = nhi;
    }
    unsigned long long t1 = pf.now();
    pf.add(P_DESCENT, t1 - t0);
    c = scan_range<IDS>(pm, qx, qy, qz, lo, hi, k, c, pf);
    t0 = pf.now();
    pf.add(P_SCAN0, t0 - t1);
    if (c.count > k) {
        __builtin_amdgcn_wave_barrier();
        c = select_k<IDS>(c.count, k, c.tight ? c.bound : 0.f);
        t1 = pf.now();
        pf.add(P_SELECT, t1 - t0); pf.add(P_N_SELECT, 1);
    }
    __builtin_amdgcn_wave_barrier();

    // ---- 2. frontier rounds.  Leaves found on the way are listed and scanned together, 16 at a time in
    // flight, when the list is full or the frontier is empty: one memory round trip per batch of
    // leaves instead of one per round.
    int n_pend = 0;
    while (nf > 0) {
        t0 = pf.now();
        const int take = nf < WAVE ? nf : WAVE;
        const int base = nf - take;
        nf = base;
        int rlo = 0, rhi = 0;
        float rd2 = 0.f;
        if (lane < take) { rlo = L.f_lo[base + lane]; rhi = L.f_hi[base + lane]; rd2 = L.f_d2[base + lane]; }
        const bool live = lane < take && rhi > rlo && !(rd2 > c.bound);
        const bool small = live && rhi - rlo <= WAVE;
        bool is_leaf = small;
        if (is_leaf) {
            // a leaf whose box lies outside the bound holds no candidate.  The box distance is formed like a
            // photon's distance from per-axis offsets that are no larger in magnitude, so in float, too,
            // it never exceeds the distance of any photon inside.
            const float* b = pm.leaf_box + 6 * (size_t)(rlo >> 5);
            float ex = fmaxf(fmaxf(b[0] - qx, qx - b[3]), 0.f), ey = fmaxf(fmaxf(b[1] - qy, qy - b[4]), 0.f), ez = fmaxf(fmaxf(b[2] - qz, qz - b[5]), 0.f);
            if (!within(ex * ex + ey * ey + ez * ez, c)) is_leaf = false;
        }
        const bool is_inner = live && !small;
        int mid = 0, nlo = 0, nhi = 0, flo = 0, fhi = 0;
        float dd = 0.f, d2 = 3.0e38f;
        if (is_inner) {
            mid = rlo + (rhi - rlo) / 2;
            const float4 pv = pm.pos[mid];
            const int ax = (int)pv.w;
            float px = pv.x, py = pv.y, pz = pv.z;
            float dx = qx - px, dy = qy - py, dz = qz - pz;
            d2 = dx * dx + dy * dy + dz * dz;
            float split = ax == 0 ? px : (ax == 1 ? py : pz);
            float delta = (ax == 0 ? qx : (ax == 1 ? qy : qz)) - split;
            if (delta < 0.f) { nlo = rlo; nhi = mid; flo = mid + 1; fhi = rhi; }
            else { nlo = mid + 1; nhi = rhi; flo = rlo; fhi = mid; }
            dd = delta * delta;
        }
        make_room<IDS>(c, k, WAVE, pf);
        append<IDS>(c.count, is_inner && within(d2, c), d2, (uint32_t)mid);
        const bool push_far = is_inner && fhi > flo && dd <= c.bound;
        const bool push_near = is_inner && nhi > nlo;
        const unsigned long long m_far = __ballot(push_far), m_near = __ballot(push_near), m_leaf = __ballot(is_leaf);
        const int n_far = __builtin_popcountll(m_far), n_near = __builtin_popcountll(m_near), n_leaf = __builtin_popcountll(m_leaf);
        __builtin_amdgcn_wave_barrier();
        const bool fits = nf + n_far + n_near <= front_cap;
        if (n_pend > 0 && (n_pend + n_leaf > WAVE || !fits)) {
            // (the serial walk of a full frontier uses the leaf list as its own scratch)
            unsigned long long ts = pf.now();
            c = scan_steps<IDS>(pm, qx, qy, qz, n_pend, k, c, pf);
            pf.add(P_LEAVES, pf.now() - ts); pf.add(P_N_LEAVES, n_pend);
            n_pend = 0;
            __builtin_amdgcn_wave_barrier();
        }
        if (is_leaf) {
            int r = n_pend + (int)prefix_rank(m_leaf);
            L.c_lo[r] = rlo;
            L.c_n[r] = rhi - rlo;
        }
        n_pend += n_leaf;
        if (fits) {
            // far sides below near sides: the near sides are taken first
            if (push_far) { int p = nf + (int)prefix_rank(m_far); L.f_lo[p] = flo; L.f_hi[p] = fhi; L.f_d2[p] = dd; }
            nf += n_far;
            if (push_near) { int p = nf + (int)prefix_rank(m_near); L.f_lo[p] = nlo; L.f_hi[p] = nhi; L.f_d2[p] = 0.f; }
            nf += n_near;
        }
        __builtin_amdgcn_wave_barrier();
        t1 = pf.now();
        pf.add(P_ROUNDS, t1 - t0); pf.add(P_N_ROUNDS, 1);
        if (!fits) {
            // frontier full: walk this round's children serially instead, lane by lane
            if (n_pend > 0) {
                c = scan_steps<IDS>(pm, qx, qy, qz, n_pend, k, c, pf);
                pf.add(P_N_LEAVES, n_pend);
                n_pend = 0;
                __builtin_amdgcn_wave_barrier();
            }
            unsigned long long todo = m_far | m_near;
            while (todo) {
                int src = __builtin_ctzll(todo);
                todo &= todo - 1;
                int a_lo = __shfl(nlo, src), a_hi = __shfl(nhi, src), b_lo = __shfl(flo, src), b_hi = __shfl(fhi, src);
                float b_dd = __shfl(dd, src);
                if ((m_near >> src) & 1ull) c = walk_serial<IDS>(pm, qx, qy, qz, a_lo, a_hi, k, c, pf);
                if (((m_far >> src) & 1ull) && !(b_dd > c.bound)) c = walk_serial<IDS>(pm, qx, qy, qz, b_lo, b_hi, k, c, pf);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (n_pend > 0) {
        t0 = pf.now();
        c = scan_steps<IDS>(pm, qx, qy, qz, n_pend, k, c, pf);
        pf.add(P_LEAVES, pf.now() - t0); pf.add(P_N_LEAVES, n_pend);
    }
    if (c.count > k) {
        __builtin_amdgcn_wave_barrier();
        t0 = pf.now();
        c = select_k<IDS>(c.count, k, c.tight ? c.bound : 0.f);
        pf.add(P_SELECT, pf.now() - t0); pf.add(P_N_SELECT, 1);
    }
    return c;
}

constexpr int S_FLIGHT = 8;           // 64-photon steps of a streaming loop whose loads are in flight together
constexpr int S_CELL = 512;           // photons in the tree cell whose density gives a first guess at a k-th distance

__device__ __forceinline__ float wave_min_f(float v)
{
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v)
{
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

// 16 bytes read as GLOBAL memory (a pointer of unknown address space makes it a flat load, which every wait for an LDS
// read in the loops would wait for, too)
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load_global4(const float4* p)
{
    const v4f v = *(const v4f __attribute__((address_space(1)))*)p;
    return float4{v.x, v.y, v.z, v.w};
}

// ---- a first guess at a point's k-th nearest distance from the photon density around it: the cell (subtree) of at most
// S_CELL photons that holds the centre is a box whose photons -- points on a surface -- fill its bounding box, so their
// number over the box's face area (its two largest extents) is the density there, and r_k ~ sqrt(k / (pi density)).
// The guess only has to be close: what the walk then lists is counted, and a second walk corrects a bad one.
// (`volume`: the map of photons scattered inside a subsurface material.  Where they fill the cell's box the density is
// their number over its volume and r_k ~ cbrt(3 k / (4 pi density)); where they lie in a thin layer under the lit surface,
// as they do when the free path is short, the surface form holds.  The larger of the two: a guess too small costs a
// second walk, one too large only a longer list for the filter to cut down.)
__device__ __attribute__((noinline)) float radius_guess(PhotonMapView pm, float cx, float cy, float cz, int k, bool volume)
{
    const int lane = lane_id();
    int lo = 0, hi = (int)pm.n;
    while (hi - lo > S_CELL) {
        const int mid = lo + (hi - lo) / 2;
        const float4 pv = pm.pos[mid];
        const int ax = (int)pv.w;
        const float split = ax == 0 ? pv.x : (ax == 1 ? pv.y : pv.z);
        const float delta = (ax == 0 ? cx : (ax == 1 ? cy : cz)) - split;
        if (delta < 0.f) hi = mid; else lo = mid + 1;
    }
    if (hi - lo < 16) return 0.f;
    static_assert(S_CELL <= S_FLIGHT * WAVE, "one batch of loads covers the cell");
    float4 p[S_FLIGHT];
#pragma unroll
    for (int u = 0; u < S_FLIGHT; u++) {
        const int i = lo + u * WAVE + lane;
        p[u] = load_global4(pm.pos + (i < hi ? i : hi - 1));      // (a repeated photon changes no bound)
    }
    float nx = p[0].x, ny = p[0].y, nz = p[0].z, xx = nx, xy = ny, xz = nz;
#pragma unroll
    for (int u = 1; u < S_FLIGHT; u++) {
        nx = fminf(nx, p[u].x); ny = fminf(ny, p[u].y); nz = fminf(nz, p[u].z);
        xx = fmaxf(xx, p[u].x); xy = fmaxf(xy, p[u].y); xz = fmaxf(xz, p[u].z);
    }
    const float ex = wave_max_f(xx) - wave_min_f(nx), ey = wave_max_f(xy) - wave_min_f(ny), ez = wave_max_f(xz) - wave_min_f(nz);
    const float e_max = fmaxf(ex, fmaxf(ey, ez)), e_min = fminf(ex, fminf(ey, ez)), e_mid = ex + ey + ez - e_max - e_min;
    const float area = e_max * e_mid;
    if (volume) {
        if (!(area > 0.f)) return 0.f;
        const float vol = area * e_min;
        return fmaxf(cbrtf(3.f * (float)k * vol / (4.f * 3.14159265f * (float)(hi - lo))), sqrtf((float)k * area / (3.14159265f * (float)(hi - lo))));
    }
    if (!(area > 0.f)) return 0.f;
    return sqrtf((float)k * area / (3.14159265f * (float)(hi - lo)));
}

// State a wave carries from one query to the next.  Queries arrive in runs of neighbouring hit
// points (one 8x8 pixel tile appends its requests together), and the k-th nearest distance obeys
// the triangle inequality: r_k(q') <= r_k(q) + |q' - q|.  So a query that found its k photons hands
// the next one a bound that already contains k photons -- no first scan, no intermediate selects,
// and only the leaves that touch the final ball.
struct Neighbour {
    float x, y, z;
    float radius;     // k-th nearest distance of the previous query; < 0: none
};

// A query with no neighbour to go by -- an isolated hit point, or the first of a run -- starts from a GUESSED bound instead:
// the radius the photon density around it suggests (radius_guess) and a margin.  A guess that holds k photons saves the
// first scan and the selects that shrink an unknown bound step by step (216 k -> about half the cycles per isolated
// query at k = 1000); one that holds fewer is widened by the count just taken, and the search radius itself is the last
// resort, so the set returned is the exact one in every case.
template <bool IDS = true>
__device__ __forceinline__ int knn_collect(const PhotonMapView& pm, float qx, float qy, float qz, int k, float r2, int front_cap, Prof pf,
                                           Neighbour* nb = nullptr, bool volume_map = false)
{
    float hint = 0.f;
    if (nb && nb->radius >= 0.f) {
        float dx = qx - nb->x, dy = qy - nb->y, dz = qz - nb->z;
        float reach = nb->radius + sqrtf(dx * dx + dy * dy + dz * dz);
        hint = reach * reach * 1.001f + 1e-12f;       // slack for the rounding of the distances
    }
    bool guessed = false;
    if (!(hint > 0.f && hint < r2)) {
        const float g = radius_guess(pm, qx, qy, qz, k, volume_map) * 1.15f;
        if (g > 0.f && g * g < r2) { hint = g * g; guessed = true; }
    }
    Cand c{0, r2, 0};
    for (int attempt = 0;; attempt++) {      // (one call site: the search is inlined)
        c = knn_collect_from<IDS>(pm, qx, qy, qz, k, r2, hint, front_cap, pf);
        if (c.count >= k || !(hint > 0.f && hint < r2)) break;      // (a search bounded by the search radius alone is final)
        if (guessed && attempt < 2) {
            // a guess that holds too few: widened by the count just taken (the photons in reach go with the radius squared
            // on a surface -- with `hint`, that is; a volume's go faster)
            const float ratio = (float)k / (float)(c.count > 0 ? c.count : 1);
            hint *= fminf(ratio * 1.3f, 9.f);
        } else {
            // the last resort; for a neighbour's bound this cannot happen in exact arithmetic -- never a short set because of a hint
            hint = 0.f;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (nb) {
        nb->x = qx; nb->y = qy; nb->z = qz;
        // after the final select the bound is the k-th smallest distance; a set of fewer than k says nothing
        nb->radius = c.count >= k ? sqrtf(c.bound) : -1.f;
    }
    return c.count;
}

struct GatherParams {
    int k;
    float r2;
    double norm;      // PHOTONMUM / 16 * PI  (integer division first, main.cpp:293)
    int front_cap;    // frontier entries in use (<= FRONT; smaller values only to test the serial walk)
    double photon_num;   // PHOTONMUM as a double: the volume normaliser divides in floating point (main.cpp:332)
    uint32_t seed;       // of the march's step lengths (main.cpp:374)
    float opt_range = 0.3f;   // group form: widest relative range around the centre's k-th distance searched for a neighbour's (>= 1: the rigorous range); the launch sets it from k
    int max_rounds = 6;       // group form: rounds per window before the rest is handed on
    int min_lanes = 0;        // group form: fewest requests worth a round of their own (0: chosen from k)
    float act_reach = 1.4f;   // packed group form: a round takes the requests within act_reach x the centre's k-th distance of the centre (1.0: 3 % slower on the water frame at k = 1000, 7 % at k = 100; 2.0 as 1.4)
};

// reduce a double over the wave (all lanes get lane 0's total after the final shuffle)
__device__ __forceinline__ double wave_sum(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return __shfl(v, 0);
}
__device__ __forceinline__ float wave_max(float v)
{
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off));
    return __shfl(v, 0);
}

// One photon's share of the radiance estimate (main.cpp:271-286) without the colours: |cosine| * weight.
//   weight = 0.918 * (1 - (1 - e^(-1.953 * d2 / (2 * maxDistSqr))) / (1 - e^-1.953)),  e = 2.71828    (main.cpp:279)
//          = A + B * 2^(c2 * d2)
//   cosine = normal . (-photon direction)                                                              (main.cpp:281-284)
// The reference evaluates both in double and rounds each to float before multiplying them; here they are evaluated
// in float throughout (a few units in the last place of a float apart; the sums they enter are compared at 8-bit
// resolution).  Both forms of the gather use this one function, so they add up the same terms.
struct TermCoef { float c2, A, B; };
__device__ __forceinline__ TermCoef term_coef(float maxDistSqr)
{
    const double ln_base = 0.999999327347282;              // ln 2.71828
    const double inv_denom = 1.0 / (1 - exp(-1.953 * ln_base));
    TermCoef t;
    t.c2 = (float)(-1.953 * ln_base * 1.4426950408889634 / (2 * (double)maxDistSqr));
    t.A = (float)(0.918 * (1 - inv_denom));
    t.B = (float)(0.918 * inv_denom);
    return t;
}
__device__ __forceinline__ float photon_term(const TermCoef& t, float d2, float nx, float ny, float nz, float ux, float uy, float uz)
{
    const float weight = t.A + t.B * __builtin_amdgcn_exp2f(t.c2 * d2);
    const float cosine = nx * ux + ny * uy + nz * uz;        // of the angle to the photon's direction; the sign drops out below
    return fabsf(cosine) * weight;
}

// Radiance estimate over the gathered set (main.cpp:262-295), before the final clip.
__device__ void estimate_raw(const PhotonMapView& pm, int count, const double* nrm, const double* col, double norm, double* rgb,
                         float* max_d2)
{
    const WaveLds& L = g_L;
    const int lane = lane_id();
    rgb[0] = rgb[1] = rgb[2] = 0.0;
    *max_d2 = -1.f;
    if (count == 0) return;                       // maxDistSqr stays -1, nothing is added (main.cpp:266,288)
    float m = 0.f;
    for (int j = lane; j < count; j += WAVE) m = fmaxf(m, L.key[j]);
    const float maxDistSqr = wave_max(m);          // distance of the farthest returned photon (main.cpp:266-267)
    *max_d2 = maxDistSqr;
    const TermCoef tc = term_coef(maxDistSqr);
    const float nx = (float)nrm[0], ny = (float)nrm[1], nz = (float)nrm[2];
    double s0 = 0, s1 = 0, s2 = 0;
    constexpr int UNR = 4;       // photons per lane whose records are fetched together
    for (int base = 0; base < count; base += UNR * WAVE) {
        float dsq[UNR], a[UNR][6];
        bool valid[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            int j = base + u * WAVE + lane;
            valid[u] = j < count;
            const int jj = valid[u] ? j : count - 1;      // unconditional fetches (clamped): all in flight together
            dsq[u] = L.key[jj];
            const float* src = pm.aux + 6 * (size_t)g_idx[jj];
#pragma unroll
            for (int c = 0; c < 6; c++) a[u][c] = src[c];
        }
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            if (valid[u]) {
                const float sc = photon_term(tc, dsq[u], nx, ny, nz, a[u][3], a[u][4], a[u][5]);
                s0 += (double)(a[u][0] * sc);             // main.cpp:286 without the hit colour, multiplied in below
                s1 += (double)(a[u][1] * sc);
                s2 += (double)(a[u][2] * sc);
            }
        }
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
    double scale = 1.0 / (norm * maxDistSqr);      // main.cpp:293
    rgb[0] = col[0] * s0 * scale; rgb[1] = col[1] * s1 * scale; rgb[2] = col[2] * s2 * scale;
}

__device__ __forceinline__ void clip3(double* v)   // Color::clip, Color.h:39-57
{
    for (int c = 0; c < 3; c++) {
        double x = v[c];
        if (x > 1) x = 1;
        if (x < 0) x = 0;
        v[c] = x;
    }
}

__device__ __forceinline__ void estimate(const PhotonMapView& pm, int count, const double* nrm, const double* col, double norm, double* rgb,
                                         float* max_d2)
{
    estimate_raw(pm, count, nrm, col, norm, rgb, max_d2);
    clip3(rgb);
}

// One step of the subsurface term (main.cpp:310-334, 346-373) over the volume photons gathered
// around `centre`: sum of hit_color * power * 0.2, normalised by e^pass * PHOTONMUM / 8 * PI * m * sqrt(m)
// where m is the squared distance from HIT_POS to the photon farthest from the centre (the
// reference measures it against hit_pos in the march too, main.cpp:353).  Adds to sss[3].
__device__ void volume_step(const PhotonMapView& vm, int count, const float* hit_pos, const double* col, double e_pass, double photon_num,
                            double* sss)
{
    const WaveLds& L = g_L;
    const int lane = lane_id();
    if (count == 0) return;
    double p0 = 0, p1 = 0, p2 = 0;
    float far_key = -1.f;
    uint32_t far_idx = 0;
    for (int j = lane; j < count; j += WAVE) {
        float key = L.key[j];
        uint32_t id = g_idx[j];
        const float* a = vm.aux + 6 * (size_t)id;
        p0 += (col[0] * (double)a[0]) * 0.2;      // main.cpp:329, 368
        p1 += (col[1] * (double)a[1]) * 0.2;
        p2 += (col[2] * (double)a[2]) * 0.2;
        if (key > far_key) { far_key = key; far_idx = id; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        float ok = __shfl_xor(far_key, off);
        uint32_t oi = __shfl_xor(far_idx, off);
        if (ok > far_key || (ok == far_key && oi < far_idx)) { far_key = ok; far_idx = oi; }
    }
    p0 = wave_sum(p0); p1 = wave_sum(p1); p2 = wave_sum(p2);
    const float4 fp = vm.pos[far_idx];
    float dx = hit_pos[0] - fp.x, dy = hit_pos[1] - fp.y, dz = hit_pos[2] - fp.z;
    float m = dx * dx + dy * dy + dz * dz;        // maxDistSqr, main.cpp:317, 353
    double scale = 1.0 / (e_pass * photon_num / 8 * 3.1415926 * m * sqrtf(m));   // main.cpp:332, 371
    sss[0] += p0 * scale; sss[1] += p1 * scale; sss[2] += p2 * scale;
}

// uniform in (0, 1) from a counter: the march's step length needs one per pass and sample
__device__ __forceinline__ double unit_random(uint32_t a, uint32_t b, uint32_t seed)
{
    uint32_t x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ seed * 0xC2B2AE3Du;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return ((double)x + 0.5) * (1.0 / 4294967296.0);
}

constexpr uint32_t REDO_EMPTY = 0xFFFFFFFFu;    // a list entry nobody has written yet (the list is all REDO_EMPTY between launches)

// the estimate of one queued request, added to its pixel (one wave)
// (nb / nbv: what the wave's previous search in the surface / volume map found, see Neighbour)
__device__ __forceinline__ void answer_request(const PhotonMapView& pm, const PhotonMapView& vm, const GatherRec& r, const GatherParams& gp, Neighbour& nb, Neighbour& nbv,
                                               double* __restrict__ acc)
{
    const int lane = lane_id();
    const Prof pf{nullptr};
    double rgb[3];
    if (r.flags & GATHER_SURFACE_DONE) {
        rgb[0] = r.normal[0]; rgb[1] = r.normal[1]; rgb[2] = r.normal[2];
    } else {
        const int count = knn_collect(pm, (float)r.pos[0], (float)r.pos[1], (float)r.pos[2], gp.k, gp.r2, gp.front_cap, pf, &nb);
        __builtin_amdgcn_wave_barrier();
        float md2;
        estimate_raw(pm, count, r.normal, r.color, gp.norm, rgb, &md2);
    }
    if ((r.flags & GATHER_SSS) && vm.n > 0) {                       // main.cpp:297-388
        const float hp[3] = {(float)r.pos[0], (float)r.pos[1], (float)r.pos[2]};
        double sss[3] = {0, 0, 0};
        __builtin_amdgcn_wave_barrier();
        int vc = knn_collect(vm, hp[0], hp[1], hp[2], gp.k, gp.r2, gp.front_cap, pf, &nbv, true);
        __builtin_amdgcn_wave_barrier();
        volume_step(vm, vc, hp, r.color, 1.0, gp.photon_num, sss);
        if (r.flags & GATHER_MARCH) {
            double px = r.pos[0] + r.march[0] * 0.02, py = r.pos[1] + r.march[1] * 0.02, pz = r.pos[2] + r.march[2] * 0.02;   // main.cpp:341
            double e_pass = 2.71828;
            for (int pass = 1; pass < 3; pass++) {
                __builtin_amdgcn_wave_barrier();
                vc = knn_collect(vm, (float)px, (float)py, (float)pz, gp.k, gp.r2, gp.front_cap, pf, &nbv, true);
                __builtin_amdgcn_wave_barrier();
                volume_step(vm, vc, hp, r.color, e_pass, gp.photon_num, sss);
                // the stream is keyed by what identifies the request whatever the launch: pixel, hit point, pass
                const uint32_t where = __float_as_uint(hp[0]) * 0x9E3779B1u ^ __float_as_uint(hp[1]) * 0x85EBCA77u ^ __float_as_uint(hp[2]) * 0xC2B2AE3Du;
                double d = -log(unit_random(r.pixel, where + (uint32_t)pass, gp.seed)) / 200;   // main.cpp:374
                px += r.march[0] * d; py += r.march[1] * d; pz += r.march[2] * d;
                e_pass *= 2.71828;
            }
        }
        rgb[0] += sss[0]; rgb[1] += sss[1]; rgb[2] += sss[2];      // main.cpp:389
    }
    clip3(rgb);                                                    // main.cpp:391
    if (lane < 3) atomicAdd(&acc[3 * (size_t)r.pixel + lane], r.weight * (lane == 0 ? rgb[0] : (lane == 1 ? rgb[1] : rgb[2])));
    __builtin_amdgcn_wave_barrier();
}

// Per-request consumer: acc[3*pixel..] += weight * clip(estimate) for the requests whose queue slots
// are listed in `list` (those the group form hands over: subsurface materials, crowded or
// incoherent groups).  One request per wave at a time.  Requests differ a lot in cost, so the
// (single-wave) blocks of the persistent grid draw them from a shared counter -- several per draw
// while plenty are left, one near the end; a request takes tens of microseconds, which keeps the
// draw rate far below what one counter word sustains.  Every entry read is reset to REDO_EMPTY.
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(2, 2))) void gather_kernel(PhotonMapView pm, PhotonMapView vm, const GatherRec* __restrict__ recs, uint32_t* list,
                                                      const unsigned long long* n_list, unsigned long long* next, GatherParams gp,
                                                      double* __restrict__ acc)
{
    const int lane = lane_id();
    const unsigned long long n_all = *n_list;
    const uint32_t n = n_all > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)n_all;
    uint32_t q_end = 0;
    Neighbour nb{0.f, 0.f, 0.f, -1.f}, nbv{0.f, 0.f, 0.f, -1.f};
    while (true) {
        uint32_t left = n > q_end ? n - q_end : 0;
        uint32_t want = left / (4u * gridDim.x);
        want = want < 1 ? 1 : (want > 16 ? 16 : want);
        unsigned long long q0 = 0;
        if (lane == 0) q0 = atomicAdd(next, (unsigned long long)want);
        q0 = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(q0 >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)q0);
        if (q0 >= n) break;
        q_end = q0 + want < n ? (uint32_t)q0 + want : n;
        for (uint32_t q = (uint32_t)q0; q < q_end; q++) {
            const uint32_t slot = list[q];
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) list[q] = REDO_EMPTY;
            answer_request(pm, vm, recs[slot], gp, nb, nbv, acc);
        }
    }
}

// The same consumer started ALONGSIDE the group kernel, on a few blocks per CU: it takes the hand-overs as they are
// written (an entry is there once it is no longer REDO_EMPTY) and so fills the compute units the group kernel's last
// windows leave idle.  It draws from the same counter as the launch above, which follows the group kernel and takes
// what is left.  A wave that has drawn an entry not yet written waits for it, or for the group kernel's last window:
// after that the list is final, and an entry still empty lies beyond its end.  A wave of this kernel stays only if
// every wave of the group kernel has begun (windows_done[1] counts them), i.e. it is not holding a place one of those
// still needs: the group kernel is launched first and fills the device, and these waves move in as its waves leave.
// `fresh_records`: the group kernel completes records before it hands them over (the surface estimate of a subsurface
// request), on whichever part of the device its wave runs; a record is then read word by word with read-modify-write
// operations that change nothing -- served where the device's writes meet, not from a cache that may hold the line as
// the shading pass left it -- after the entry that announces it (the writer orders the two with a fence).
__device__ __forceinline__ GatherRec fresh_record(const GatherRec* rec)
{
    static_assert(sizeof(GatherRec) == 14 * sizeof(unsigned long long), "fourteen words");
    const int lane = lane_id();
    unsigned long long w = 0;
    if (lane < 14) w = atomicAdd(reinterpret_cast<unsigned long long*>(const_cast<GatherRec*>(rec)) + lane, 0ull);
    union { GatherRec r; unsigned long long u[14]; } x;
#pragma unroll
    for (int i = 0; i < 14; i++)
        x.u[i] = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(w >> 32), i) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)w, i);
    return x.r;
}
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(2, 2))) void gather_alongside_kernel(PhotonMapView pm, PhotonMapView vm, const GatherRec* __restrict__ recs, uint32_t* list,
                                                      uint32_t list_cap, const rtq::Counts* cnt, unsigned long long shard_cap, const unsigned long long* windows_done,
                                                      unsigned tile_waves, unsigned long long* next, GatherParams gp, double* __restrict__ acc, int fresh_records)
{
    const int lane = lane_id();
    const unsigned long long n_windows = rtq::map_units(cnt, shard_cap, WAVE).total;
    if (__hip_atomic_load(windows_done + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < tile_waves) return;
    Neighbour nb{0.f, 0.f, 0.f, -1.f}, nbv{0.f, 0.f, 0.f, -1.f};
    while (true) {
        unsigned long long q0 = 0;
        if (lane == 0) q0 = atomicAdd(next, 1ull);
        q0 = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(q0 >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)q0);
        if (q0 >= list_cap) break;
        // (One lane asks, by read-modify-write operations that change nothing: those are served in a few microseconds,
        // where a device-coherent load of a word that other waves are writing was measured to take 150 and more.)
        uint32_t slot = REDO_EMPTY;
        if (lane == 0) {
            while (true) {
                slot = atomicOr(&list[q0], 0u);
                if (slot != REDO_EMPTY) break;
                // (signed: the count runs below zero while parts of windows handed from wave to wave outnumber the windows finished)
                if ((long long)atomicAdd(const_cast<unsigned long long*>(windows_done), 0ull) >= (long long)n_windows) {
                    __threadfence();
                    slot = atomicOr(&list[q0], 0u);      // final now
                    break;
                }
                __builtin_amdgcn_s_sleep(64);
            }
        }
        slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
        if (slot == REDO_EMPTY) break;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) list[q0] = REDO_EMPTY;
        if (fresh_records) {
            const GatherRec r = fresh_record(recs + slot);
            answer_request(pm, vm, r, gp, nb, nbv, acc);
        } else {
            answer_request(pm, vm, recs[slot], gp, nb, nbv, acc);
        }
    }
}

// Batch form for the ABI test hooks: query q -> count, max d2, clipped estimate
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(2, 2))) void knn_batch_kernel(PhotonMapView pm, long long n, const double* __restrict__ pos, const double* __restrict__ nrm,
                                                         const double* __restrict__ col, GatherParams gp, int* __restrict__ counts,
                                                         double* __restrict__ r2_out, double* __restrict__ rgb_out, int* __restrict__ idx_out,
                                                         unsigned long long* next, unsigned long long* prof)
{
    const WaveLds& L = g_L;
    const int lane = lane_id();
    const Prof pf{prof};
    pf.init();
    __builtin_amdgcn_wave_barrier();
    // runs of 8 consecutive queries per draw, like the gather kernel: neighbouring queries hand their bound on
    Neighbour nb{0.f, 0.f, 0.f, -1.f};
    unsigned long long q = 0, q_end = 0;
    while (true) {
        if (++q >= q_end) {
            unsigned long long q0 = 0;
            if (lane == 0) q0 = atomicAdd(next, 8ull);
            q = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(q0 >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)q0);
            q_end = q + 8 < (unsigned long long)n ? q + 8 : (unsigned long long)n;
            nb.radius = -1.f;
        }
        if (q >= (unsigned long long)n) break;
        const int count = knn_collect(pm, (float)pos[3 * q], (float)pos[3 * q + 1], (float)pos[3 * q + 2], gp.k, gp.r2, gp.front_cap, pf, &nb);
        __builtin_amdgcn_wave_barrier();
        pf.add(P_QUERIES, 1);
        double rgb[3] = {0, 0, 0};
        float md2 = -1.f;
        if (nrm && col) {
            double nn[3] = {nrm[3 * q], nrm[3 * q + 1], nrm[3 * q + 2]}, cc[3] = {col[3 * q], col[3 * q + 1], col[3 * q + 2]};
            unsigned long long t = pf.now();
            estimate(pm, count, nn, cc, gp.norm, rgb, &md2);
            pf.add(P_ESTIMATE, pf.now() - t);
        } else if (count > 0) {
            float m = 0.f;
            for (int j = lane; j < count; j += WAVE) m = fmaxf(m, L.key[j]);
            md2 = wave_max(m);
        }
        if (lane == 0) {
            counts[q] = count;
            if (r2_out) r2_out[q] = (double)md2;
        }
        if (rgb_out && lane < 3) rgb_out[3 * q + lane] = lane == 0 ? rgb[0] : (lane == 1 ? rgb[1] : rgb[2]);
        if (idx_out) {   // the set itself, in the caller's numbering, -1 padded to k
            for (int j = lane; j < gp.k; j += WAVE) idx_out[(size_t)q * gp.k + j] = j < count ? pm.order[g_idx[j]] : -1;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_wave_barrier();
    pf.flush();
}

}  // namespace rtg
