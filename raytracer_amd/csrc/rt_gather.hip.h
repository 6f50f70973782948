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
constexpr int GROUP = 4;         // 64-photon steps of a linear scan kept in flight together
constexpr int MAX_K = CAP - GROUP * WAVE;
constexpr int STACK = 48;        // > log2(2^31 / 64) levels
constexpr int FRONT = 256;       // frontier capacity (ranges); beyond it subtrees are walked serially

struct PhotonMapView {
    const float* pos;            // 3 per photon, tree order
    const float* aux;            // 6 per photon: power rgb, dir xyz
    const uint8_t* axis;
    const int* order;            // tree position -> index in the caller's photon array
    long long n;
};

// one gather request = one diffuse hit
struct GatherRec {
    double pos[3];
    double normal[3];
    double color[3];
    double weight;               // path throughput the estimate is scaled by
    uint32_t pixel;
    uint32_t pad;
};

struct WaveLds {
    float key[CAP];
    uint32_t idx[CAP];
    uint32_t hist[256];
    int f_lo[FRONT], f_hi[FRONT];   // frontier of pending subtrees: range and squared plane distance
    float f_d2[FRONT];
    int c_lo[WAVE], c_n[WAVE];      // steps of the current linear scan: first photon, count (<= 64)
    int s_lo[STACK], s_hi[STACK];   // stack of the serial walk
    float s_d2[STACK];
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ unsigned prefix_rank(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// k-th smallest (1-based) of key[0..count): wave-wide MSB-first radix select on the float bits
// (all keys are non-negative, so the unsigned bit pattern orders like the value).  Afterwards the
// buffer holds exactly the k smallest entries.  Returns the k-th key.
__device__ float select_k(WaveLds& L, int& count, int k)
{
    const int lane = lane_id();
    uint32_t prefix = 0, pmask = 0;
    int want = k;
    for (int pass = 3; pass >= 0; pass--) {
        for (int b = lane; b < 256; b += WAVE) L.hist[b] = 0;
        __builtin_amdgcn_wave_barrier();
        for (int j = lane; j < count; j += WAVE) {
            uint32_t u = __float_as_uint(L.key[j]);
            if ((u & pmask) == prefix) atomicAdd(&L.hist[(u >> (8 * pass)) & 255u], 1u);
        }
        __builtin_amdgcn_wave_barrier();
        // lane l owns bins 4l .. 4l+3
        uint32_t h0 = L.hist[4 * lane], h1 = L.hist[4 * lane + 1], h2 = L.hist[4 * lane + 2], h3 = L.hist[4 * lane + 3];
        uint32_t s = h0 + h1 + h2 + h3;
        uint32_t incl = s;
        for (int off = 1; off < WAVE; off <<= 1) {
            uint32_t v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        uint32_t excl = incl - s;
        bool mine = (uint32_t)want > excl && (uint32_t)want <= incl;   // exactly one lane
        uint32_t bin = 0, below = 0;
        if (mine) {
            uint32_t w = (uint32_t)want - excl;
            if (w <= h0) { bin = 0; below = excl; }
            else if (w <= h0 + h1) { bin = 1; below = excl + h0; }
            else if (w <= h0 + h1 + h2) { bin = 2; below = excl + h0 + h1; }
            else { bin = 3; below = excl + h0 + h1 + h2; }
            bin += 4 * lane;
        }
        unsigned long long who = __ballot(mine);
        int src = __builtin_ctzll(who);
        bin = __shfl(bin, src);
        below = __shfl(below, src);
        want -= (int)below;
        prefix |= bin << (8 * pass);
        pmask |= 255u << (8 * pass);
        __builtin_amdgcn_wave_barrier();
    }
    const uint32_t T = prefix;     // bit pattern of the k-th smallest key; `want` = how many == T to keep
    // stable in-place compaction, one 64-entry chunk at a time (outputs never pass the read cursor)
    int out = 0, eq_taken = 0;
    for (int base = 0; base < count; base += WAVE) {
        int j = base + lane;
        bool valid = j < count;
        uint32_t u = valid ? __float_as_uint(L.key[j]) : 0xFFFFFFFFu;
        uint32_t id = valid ? L.idx[j] : 0u;
        bool less = valid && u < T;
        bool eq = valid && u == T;
        unsigned long long meq = __ballot(eq);
        bool keep = less || (eq && (eq_taken + (int)prefix_rank(meq)) < want);
        unsigned long long mk = __ballot(keep);
        __builtin_amdgcn_wave_barrier();
        if (keep) {
            int p = out + (int)prefix_rank(mk);
            L.key[p] = __uint_as_float(u);
            L.idx[p] = id;
        }
        out += __builtin_popcountll(mk);
        eq_taken += __builtin_popcountll(meq);
        __builtin_amdgcn_wave_barrier();
    }
    count = out;
    return __uint_as_float(T);
}

__device__ __forceinline__ void append(WaveLds& L, int& count, bool take, float d2, uint32_t id)
{
    unsigned long long m = __ballot(take);
    if (take) {
        int p = count + (int)prefix_rank(m);
        L.key[p] = d2;
        L.idx[p] = id;
    }
    count += __builtin_popcountll(m);
}

// Make room for `need` more candidates (need <= GROUP * WAVE): if they might not fit, keep the k
// nearest and tighten the bound.  count > k holds whenever this triggers (k <= MAX_K).
__device__ __forceinline__ void make_room(WaveLds& L, int& count, int k, float& bound, int need)
{
    if (count > CAP - need) {
        __builtin_amdgcn_wave_barrier();
        bound = select_k(L, count, k);
    }
}

// Linear scan of the n_steps (<= 64) steps listed in L.c_lo / L.c_n, GROUP steps in flight.
__device__ void scan_steps(const PhotonMapView& pm, WaveLds& L, const float* q, int n_steps, int k, float& bound, int& count)
{
    const int lane = lane_id();
    for (int g = 0; g < n_steps; g += GROUP) {
        float px[GROUP], py[GROUP], pz[GROUP];
        int id[GROUP];
        bool valid[GROUP];
#pragma unroll
        for (int u = 0; u < GROUP; u++) {
            valid[u] = false;
            id[u] = 0;
            if (g + u < n_steps) {
                int lo = L.c_lo[g + u], n = L.c_n[g + u];
                valid[u] = lane < n;
                id[u] = lo + lane;
                if (valid[u]) {
                    const float* p = pm.pos + 3 * (size_t)id[u];
                    px[u] = p[0]; py[u] = p[1]; pz[u] = p[2];
                }
            }
        }
        make_room(L, count, k, bound, GROUP * WAVE);
#pragma unroll
        for (int u = 0; u < GROUP; u++) {
            float d2 = 3.0e38f;
            if (valid[u]) {
                float dx = q[0] - px[u], dy = q[1] - py[u], dz = q[2] - pz[u];
                d2 = dx * dx + dy * dy + dz * dz;
            }
            append(L, count, valid[u] && d2 < bound, d2, (uint32_t)id[u]);
        }
    }
}

// Linear scan of one contiguous range (a whole subtree), any length.
__device__ void scan_range(const PhotonMapView& pm, WaveLds& L, const float* q, int lo, int hi, int k, float& bound, int& count)
{
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
        scan_steps(pm, L, q, steps, k, bound, count);
        lo += steps * WAVE;
    }
}

// Serial depth-first walk of one subtree (all lanes in step); used when the frontier is full.
__device__ void walk_serial(const PhotonMapView& pm, WaveLds& L, const float* q, int lo, int hi, int k, float& bound, int& count)
{
    const int lane = lane_id();
    int sp = 0;
    float dfar = 0.f;
    bool have = true;
    while (true) {
        if (!have) {
            if (sp == 0) break;
            sp--;
            lo = L.s_lo[sp]; hi = L.s_hi[sp]; dfar = L.s_d2[sp];
            if (dfar > bound) continue;          // Kdtree.cpp:123-125: far side only if delta^2 <= bound
        }
        have = false;
        while (hi - lo > WAVE) {
            int mid = lo + (hi - lo) / 2;
            int ax = pm.axis[mid];
            float px = pm.pos[3 * (size_t)mid], py = pm.pos[3 * (size_t)mid + 1], pz = pm.pos[3 * (size_t)mid + 2];
            float dx = q[0] - px, dy = q[1] - py, dz = q[2] - pz;
            float d2 = dx * dx + dy * dy + dz * dz;
            make_room(L, count, k, bound, WAVE);
            append(L, count, lane == 0 && d2 < bound, d2, (uint32_t)mid);   // the pivot itself (Kdtree.cpp:119)
            float split = ax == 0 ? px : (ax == 1 ? py : pz);
            float delta = q[ax] - split;
            int nlo, nhi, flo, fhi;
            if (delta < 0.f) { nlo = lo; nhi = mid; flo = mid + 1; fhi = hi; }      // Kdtree.cpp:121
            else { nlo = mid + 1; nhi = hi; flo = lo; fhi = mid; }
            float dd = delta * delta;
            if (fhi > flo && dd <= bound && sp < STACK) {
                if (lane == 0) { L.s_lo[sp] = flo; L.s_hi[sp] = fhi; L.s_d2[sp] = dd; }
                sp++;
            }
            lo = nlo; hi = nhi;
        }
        scan_range(pm, L, q, lo, hi, k, bound, count);
        __builtin_amdgcn_wave_barrier();
    }
}

// k nearest photons within sqrt(r2) of one query held identically by all lanes.  On return
// L.key/L.idx[0..count) is the set.
__device__ void knn_collect(const PhotonMapView& pm, WaveLds& L, float qx, float qy, float qz, int k, float r2, int& count)
{
    const int lane = lane_id();
    count = 0;
    if (pm.n <= 0) return;
    float bound = r2;            // candidates need d2 < bound (Kdtree.cpp:77,82: strict)
    const float q[3] = {qx, qy, qz};

    // ---- 1. near side down to a range of <= 2k photons, far sides onto the frontier
    int nf = 0;
    int lo = 0, hi = (int)pm.n;
    const int limit = 2 * k > WAVE ? 2 * k : WAVE;
    while (hi - lo > limit) {
        int mid = lo + (hi - lo) / 2;
        int ax = pm.axis[mid];
        float px = pm.pos[3 * (size_t)mid], py = pm.pos[3 * (size_t)mid + 1], pz = pm.pos[3 * (size_t)mid + 2];
        float dx = qx - px, dy = qy - py, dz = qz - pz;
        float d2 = dx * dx + dy * dy + dz * dz;
        append(L, count, lane == 0 && d2 < bound, d2, (uint32_t)mid);   // the pivot itself (Kdtree.cpp:119); < 32 of these
        float split = ax == 0 ? px : (ax == 1 ? py : pz);
        float delta = q[ax] - split;
        int nlo, nhi, flo, fhi;
        if (delta < 0.f) { nlo = lo; nhi = mid; flo = mid + 1; fhi = hi; }      // Kdtree.cpp:121
        else { nlo = mid + 1; nhi = hi; flo = lo; fhi = mid; }
        float dd = delta * delta;
        if (fhi > flo && dd <= bound) {          // at most one per level: nf < 32 <= FRONT
            if (lane == 0) { L.f_lo[nf] = flo; L.f_hi[nf] = fhi; L.f_d2[nf] = dd; }
            nf++;
        }
        lo = nlo; hi = nhi;
    }
    scan_range(pm, L, q, lo, hi, k, bound, count);
    if (count > k) {
        __builtin_amdgcn_wave_barrier();
        bound = select_k(L, count, k);
    }
    __builtin_amdgcn_wave_barrier();

    // ---- 2. frontier rounds
    while (nf > 0) {
        const int take = nf < WAVE ? nf : WAVE;
        const int base = nf - take;
        nf = base;
        int rlo = 0, rhi = 0;
        float rd2 = 0.f;
        if (lane < take) { rlo = L.f_lo[base + lane]; rhi = L.f_hi[base + lane]; rd2 = L.f_d2[base + lane]; }
        const bool live = lane < take && rhi > rlo && !(rd2 > bound);
        const bool is_leaf = live && rhi - rlo <= WAVE;
        const bool is_inner = live && !is_leaf;
        int mid = 0, nlo = 0, nhi = 0, flo = 0, fhi = 0;
        float dd = 0.f, d2 = 3.0e38f;
        if (is_inner) {
            mid = rlo + (rhi - rlo) / 2;
            int ax = pm.axis[mid];
            float px = pm.pos[3 * (size_t)mid], py = pm.pos[3 * (size_t)mid + 1], pz = pm.pos[3 * (size_t)mid + 2];
            float dx = qx - px, dy = qy - py, dz = qz - pz;
            d2 = dx * dx + dy * dy + dz * dz;
            float split = ax == 0 ? px : (ax == 1 ? py : pz);
            float delta = (ax == 0 ? qx : (ax == 1 ? qy : qz)) - split;
            if (delta < 0.f) { nlo = rlo; nhi = mid; flo = mid + 1; fhi = rhi; }
            else { nlo = mid + 1; nhi = rhi; flo = rlo; fhi = mid; }
            dd = delta * delta;
        }
        make_room(L, count, k, bound, WAVE);
        append(L, count, is_inner && d2 < bound, d2, (uint32_t)mid);
        const bool push_far = is_inner && fhi > flo && dd <= bound;
        const bool push_near = is_inner && nhi > nlo;
        const unsigned long long m_far = __ballot(push_far), m_near = __ballot(push_near), m_leaf = __ballot(is_leaf);
        const int n_far = __builtin_popcountll(m_far), n_near = __builtin_popcountll(m_near), n_leaf = __builtin_popcountll(m_leaf);
        __builtin_amdgcn_wave_barrier();
        if (is_leaf) {
            int r = (int)prefix_rank(m_leaf);
            L.c_lo[r] = rlo;
            L.c_n[r] = rhi - rlo;
        }
        const bool fits = nf + n_far + n_near <= FRONT;
        if (fits) {
            // far sides below near sides: the near sides are taken first
            if (push_far) { int p = nf + (int)prefix_rank(m_far); L.f_lo[p] = flo; L.f_hi[p] = fhi; L.f_d2[p] = dd; }
            nf += n_far;
            if (push_near) { int p = nf + (int)prefix_rank(m_near); L.f_lo[p] = nlo; L.f_hi[p] = nhi; L.f_d2[p] = 0.f; }
            nf += n_near;
        }
        __builtin_amdgcn_wave_barrier();
        scan_steps(pm, L, q, n_leaf, k, bound, count);
        if (!fits) {
            // frontier full: walk this round's children serially instead, lane by lane
            unsigned long long todo = m_far | m_near;
            while (todo) {
                int src = __builtin_ctzll(todo);
                todo &= todo - 1;
                int a_lo = __shfl(nlo, src), a_hi = __shfl(nhi, src), b_lo = __shfl(flo, src), b_hi = __shfl(fhi, src);
                float b_dd = __shfl(dd, src);
                if ((m_near >> src) & 1ull) walk_serial(pm, L, q, a_lo, a_hi, k, bound, count);
                if (((m_far >> src) & 1ull) && !(b_dd > bound)) walk_serial(pm, L, q, b_lo, b_hi, k, bound, count);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (count > k) {
        __builtin_amdgcn_wave_barrier();
        select_k(L, count, k);
    }
}

struct GatherParams {
    int k;
    float r2;
    double norm;      // PHOTONMUM / 16 * PI  (integer division first, main.cpp:293)
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

// Radiance estimate over the gathered set (main.cpp:262-295).  rgb is the clipped estimate.
__device__ void estimate(const PhotonMapView& pm, const WaveLds& L, int count, const double* nrm, const double* col, double norm, double* rgb,
                         float* max_d2)
{
    const int lane = lane_id();
    rgb[0] = rgb[1] = rgb[2] = 0.0;
    *max_d2 = -1.f;
    if (count == 0) return;                       // maxDistSqr stays -1, nothing is added (main.cpp:266,288)
    float m = 0.f;
    for (int j = lane; j < count; j += WAVE) m = fmaxf(m, L.key[j]);
    const float maxDistSqr = wave_max(m);          // distance of the farthest returned photon (main.cpp:266-267)
    *max_d2 = maxDistSqr;
    const double denom = 1 - pow(2.71828, -1.953);
    double s0 = 0, s1 = 0, s2 = 0;
    for (int j = lane; j < count; j += WAVE) {
        float distanceSqr = L.key[j];
        const float* a = pm.aux + 6 * (size_t)L.idx[j];
        // gaussian filter, main.cpp:279 (float result of a double expression)
        float weight = (float)(0.918 * (1 - (1 - pow(2.71828, -1.953 * (distanceSqr) / (2 * maxDistSqr))) / denom));
        float cosine = (float)(nrm[0] * (-(double)a[3]) + nrm[1] * (-(double)a[4]) + nrm[2] * (-(double)a[5]));   // main.cpp:281-283
        if (cosine < 0) cosine = -cosine;
        double sc = (double)(cosine * weight);
        s0 += (col[0] * (double)a[0]) * sc;        // main.cpp:286
        s1 += (col[1] * (double)a[1]) * sc;
        s2 += (col[2] * (double)a[2]) * sc;
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
    double scale = 1.0 / (norm * maxDistSqr);      // main.cpp:293
    double v[3] = {s0 * scale, s1 * scale, s2 * scale};
    for (int c = 0; c < 3; c++) {                  // Color::clip, Color.h:39-57
        double x = v[c];
        if (x > 1) x = 1;
        if (x < 0) x = 0;
        rgb[c] = x;
    }
}

// Gather queue consumer: acc[3*pixel..] += weight * clip(estimate).  One request per wave at a time.
// Requests differ a lot in cost, so the (single-wave) blocks of the persistent grid draw them from
// a shared counter -- several per draw while plenty are left, one near the end; a request takes
// tens of microseconds, which keeps the draw rate far below what one counter word sustains.
__global__ __launch_bounds__(WAVE) void gather_kernel(PhotonMapView pm, const GatherRec* __restrict__ recs, const rtq::Counts* cnt,
                                                      unsigned long long shard_cap, unsigned long long* next, GatherParams gp,
                                                      double* __restrict__ acc)
{
    __shared__ WaveLds L;
    const int lane = lane_id();
    const rtq::Map map = rtq::map_units(cnt, shard_cap, 1);
    const uint32_t n = map.total;
    uint32_t q_end = 0;
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
            unsigned shard;
            uint32_t first, n_shard;
            rtq::locate(map, q, 1, shard, first, n_shard);
            const GatherRec& r = recs[(unsigned long long)shard * shard_cap + first];
            int count;
            knn_collect(pm, L, (float)r.pos[0], (float)r.pos[1], (float)r.pos[2], gp.k, gp.r2, count);
            __builtin_amdgcn_wave_barrier();
            double rgb[3];
            float md2;
            estimate(pm, L, count, r.normal, r.color, gp.norm, rgb, &md2);
            if (lane < 3) atomicAdd(&acc[3 * (size_t)r.pixel + lane], r.weight * (lane == 0 ? rgb[0] : (lane == 1 ? rgb[1] : rgb[2])));
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// Batch form for the ABI test hooks: query q -> count, max d2, clipped estimate
__global__ __launch_bounds__(WAVE) void knn_batch_kernel(PhotonMapView pm, long long n, const double* __restrict__ pos, const double* __restrict__ nrm,
                                                         const double* __restrict__ col, GatherParams gp, int* __restrict__ counts,
                                                         double* __restrict__ r2_out, double* __restrict__ rgb_out, int* __restrict__ idx_out,
                                                         unsigned long long* next)
{
    __shared__ WaveLds L;
    const int lane = lane_id();
    while (true) {
        unsigned long long q = 0;
        if (lane == 0) q = atomicAdd(next, 1ull);
        q = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(q >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)q);
        if (q >= (unsigned long long)n) break;
        int count;
        knn_collect(pm, L, (float)pos[3 * q], (float)pos[3 * q + 1], (float)pos[3 * q + 2], gp.k, gp.r2, count);
        __builtin_amdgcn_wave_barrier();
        double rgb[3] = {0, 0, 0};
        float md2 = -1.f;
        if (nrm && col) {
            double nn[3] = {nrm[3 * q], nrm[3 * q + 1], nrm[3 * q + 2]}, cc[3] = {col[3 * q], col[3 * q + 1], col[3 * q + 2]};
            estimate(pm, L, count, nn, cc, gp.norm, rgb, &md2);
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
            for (int j = lane; j < gp.k; j += WAVE) idx_out[(size_t)q * gp.k + j] = j < count ? pm.order[L.idx[j]] : -1;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace rtg
