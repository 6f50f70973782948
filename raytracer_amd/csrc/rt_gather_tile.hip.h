// K4, second form: the photon gather for a GROUP of up to 64 neighbouring requests per wave, one
// request per lane (KDTree::findKNN, Kdtree.cpp:73-158, + the estimate of main.cpp:260-295).
//
// The one-query-per-wave form (rt_gather.hip.h) spends its time on bookkeeping around a candidate
// set in LDS.  Requests appended together by one shading wave are hit points a few pixel footprints
// apart, whose k-nearest balls overlap almost entirely.  A wave therefore
//   1. answers ONE request of the group exactly with the per-query machinery; its k-th distance r_c
//      bounds every other request's by the triangle inequality, r_k(q) <= r_c + |q - q_c|;
//   2. walks the kd-tree once for the box around all those balls, listing the leaves and inner-node
//      pivots that touch it;
//   3. streams the listed photons three times, every lane measuring its own distance to the same
//      photon (the photon's coordinates travel through scalar registers):
//        a. a histogram of the squared distances per lane (64 bins over [0, bound_q)) locates the
//           bin that holds the lane's k-th smallest distance,
//        b. the distances falling into that bin are kept (a few dozen) and ranked, which gives the
//           exact k-th distance and how many photons at exactly that distance belong to the set,
//        c. the radiance estimate is accumulated over the photons of the set, which is never stored.
// The set is the same exact k-nearest set (ties at the k-th distance broken by list order instead of
// by buffer order -- arbitrary in both forms and in the reference's heap); the sums run in another
// order.  Requests the group form does not take (subsurface materials, a crowded bin, a group whose
// box touches too much of the tree) go to a list the per-query kernel works off afterwards.
#pragma once
#include "rt_gather.hip.h"

namespace rtg {

constexpr int T_BINS = 80;        // histogram bins per request
constexpr int T_CRIT = 40;        // distances of the critical bin kept per request
constexpr int T_LEAVES = 320;     // leaves a group's box may touch
constexpr int T_PIVOTS = 512;     // inner-node pivots inside a group's box
// The per-lane tables lie over what the search for the centre's k-th distance leaves free: the candidates' distances
// and the arrays after them, up to the frontier.  (The histogram holds two 16-bit counts per word: a group lists at
// most T_LEAVES * 64 + T_PIVOTS photons.)
constexpr size_t T_TABLE_BYTES = offsetof(WaveLds, f_lo) - offsetof(WaveLds, key);
static_assert(offsetof(WaveLds, key) == 0 && T_BINS * WAVE * 2 <= T_TABLE_BYTES && T_CRIT * WAVE * 4 <= T_TABLE_BYTES, "the per-lane tables fit");
static_assert(T_LEAVES * 64 + T_PIVOTS < 65536, "histogram counts fit 16 bits");

struct TileLds {
    uint32_t leaf[T_LEAVES];      // first photon << 7 | photons (<= 64): the leaves some request's ball touches
    uint32_t part[T_LEAVES];      // the subset of them one pass has to look at photon by photon
    uint32_t piv[T_PIVOTS];       // tree positions of pivots
};
__shared__ TileLds g_T;

constexpr uint32_t GATHER_COUNT_SHIFT = 8;    // GatherRec::flags bits 8..14 of a window's first record: requests in the window

// Append the requests of one wave as a window of 64 slots (the group the gather works on).  The
// first record carries the number of requests.  cnt->n[shard][0] counts slots, [1] requests.
__device__ __forceinline__ void push_window(GatherRec* q, rtq::Counts* cnt, unsigned long long shard_cap, bool want, GatherRec& rec,
                                            unsigned long long* overflow)
{
    unsigned long long m = __ballot(want);
    if (m == 0) return;
    const unsigned shard = rtq::wave_index() & (rtq::SHARDS - 1);
    const int leader = __builtin_ctzll(m);
    const unsigned n_req = (unsigned)__builtin_popcountll(m);
    unsigned long long base = 0;
    if ((int)rtq::lane_id() == leader) {
        base = atomicAdd(&cnt->n[shard][0], (unsigned long long)WAVE);
        atomicAdd(&cnt->n[shard][1], (unsigned long long)n_req);
        rec.flags |= n_req << GATHER_COUNT_SHIFT;
    }
    unsigned lo = __shfl((unsigned)base, leader), hi = __shfl((unsigned)(base >> 32), leader);
    base = ((unsigned long long)hi << 32) | lo;
    if (want) {
        if (base + WAVE <= shard_cap) q[(unsigned long long)shard * shard_cap + base + rtq::prefix_rank(m)] = rec;
        else *overflow = 1;
    }
}

struct TileArgs {
    const GatherRec* recs;
    const rtq::Counts* cnt;          // n[s][0]: slots appended to shard s (a multiple of 64)
    unsigned long long shard_cap;
    unsigned long long* next;        // draw counter
    uint32_t* redo;                  // slots of the requests left to the per-query kernel (entries not yet written hold REDO_EMPTY)
    unsigned long long* n_redo;
    unsigned long long* windows_done;   // [0] windows finished, their hand-overs written; [1] waves of this kernel that have begun (the per-query kernel that runs alongside goes by both)
    int all_redo;                    // nonzero: hand every request to the per-query kernel (comparison runs)
    int sss_surface;                 // nonzero: form the surface estimate of subsurface requests here and hand the rest on (the per-request
                                     // kernel must then run AFTER this one: the result travels in the request's record); 2: the packed
                                     // form's subsurface instantiation answers their volume terms as well and hands on what it cannot finish
    unsigned long long* prof;        // optional cycle accounting (development aid): TP_* slots
    unsigned long long* span;        // optional (development aid): per wave of the packed form, when it began and ended (100 MHz clock) and the windows it took
    // The packed form's lists of window parts handed from wave to wave (see gather_stream_body), CONT_SHARDS of them, each:
    // [0] entries taken, [1] entries written and not yet taken (signed), [2] entries reserved, [3] unused, then two words
    // per entry -- the lanes still waiting (never zero), and window | round << 32 | 1 << 63.  Entries are zero until
    // written and zeroed again by the wave that takes them.  Null: rounds stay with their wave.
    unsigned long long* cont;
    unsigned cont_cap;               // entries one list holds
    int part_tail;                   // parts are handed on while the kernel draws its last part_tail x (waves of the grid) windows
    int late_first;                  // packed form: windows are taken from each shard's last to its first (see gather_stream_body)
    int pack_cap;                    // photons a wave's packed list holds (<= S_PACK_CAP)
};
enum TileProf { TP_WINDOWS, TP_ROUNDS, TP_LANES, TP_STEPS, TP_KTH, TP_WALK, TP_HIST, TP_CRIT, TP_RANK, TP_SUM, TP_N_HIST, TP_N_CRIT, TP_N_SUM,
                TP_R_FLAGS, TP_R_FEW, TP_R_WALK, TP_R_SHORT, TP_R_CROWD, TP_R_LEFT, TP_R_TIE,      // TP_R_*: requests handed on, by reason
                // the packed form's search for the centre's radius: cycles by part, and attempts by outcome
                TP_C_GUESS, TP_C_WALK, TP_C_COUNT, TP_C_SORT, TP_C_PACK, TP_C_KTH, TP_A_NOGUESS, TP_A_LISTS, TP_A_FEW, TP_A_MANY, TP_A_PACK,
                TP_W_ROUNDS1, TP_W_ROUNDS2, TP_W_ROUNDS3, TP_W_ROUNDS4, TP_W_CYC1, TP_W_CYC2, TP_W_CYC3, TP_W_CYC4, TP_W_MAX, TP_W_FIRST_MIN, TP_W_LAST_MAX,      // windows by rounds taken: count, cycles; longest window
                TP_BUSY, TP_BUSY_MAX, TP_SLOTS };

typedef float f2 __attribute__((ext_vector_type(2)));   // two photons per packed instruction

__device__ __forceinline__ float bcast(float v, int src) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), src)); }

// a pointer of unknown address space as one into global memory
__device__ __forceinline__ const float __attribute__((address_space(1)))* as_global(const float* p)
{
    return (const float __attribute__((address_space(1)))*)p;
}

// Step s of the group's photon list: the tree position lane j reads (clamped to a valid one) and the
// number of photons in the step (the same in every lane, returned as a scalar).
__device__ __forceinline__ int step_index(const uint32_t* leaf, int s, int n_leaf, int n_piv, int lane, int& n)
{
    const TileLds& T = g_T;
    if (s < n_leaf) {
        const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)leaf[s]);
        n = (int)(e & 127u);
        return (int)(e >> 7) + (lane < n ? lane : n - 1);
    }
    const int first = (s - n_leaf) * WAVE;
    n = n_piv - first < WAVE ? n_piv - first : WAVE;
    return (int)T.piv[first + (lane < n ? lane : n - 1)];
}

// the step's photons, lane j holding photon j (lanes beyond the step re-read its last photon; stage_put replaces those
// by a point no ball reaches, so the loops over a step need no bound checks)
__device__ __forceinline__ float4 step_photons(const float4* __restrict__ pos, const uint32_t* leaf, int s, int n_leaf, int n_piv, int lane, int& n, int& at)
{
    at = step_index(leaf, s, n_leaf, n_piv, lane, n);
    // used only when it is staged, a step later: the fetch stays in flight meanwhile.  Read as GLOBAL memory: a pointer of
    // unknown address space makes it a flat load, which every wait for an LDS read in the loops would wait for, too.
    const float __attribute__((address_space(1)))* g = as_global(reinterpret_cast<const float*>(pos + at));
    return float4{g[0], g[1], g[2], g[3]};
}

// k-th nearest distance around one point by the per-query machinery (< 0: fewer than k photons within the radius).
// Not inlined: its register needs stay out of the group kernel's loops.
__device__ __attribute__((noinline)) float kth_distance(PhotonMapView pm, float cx, float cy, float cz, int k, float r2, int front_cap)
{
    Neighbour nb{0.f, 0.f, 0.f, -1.f};
    knn_collect<false>(pm, cx, cy, cz, k, r2, front_cap, Prof{nullptr}, &nb);      // (the distances alone: no photon ids kept)
    return nb.radius;
}

// The leaves and pivots inside the box [gl, gh]: fills g_T.leaf / g_T.piv and returns
// n_leaf | n_piv << 16, or -1 when the box touches more of the tree than the lists hold.
__device__ __attribute__((noinline)) int group_walk(PhotonMapView pm, float glx, float gly, float glz, float ghx, float ghy, float ghz)
{
    WaveLds& L = g_L;
    TileLds& T = g_T;
    const int lane = lane_id();
    int n_leaf = 0, n_piv = 0, nf = 1;
    if (lane == 0) { L.f_lo[0] = 0; L.f_hi[0] = (int)pm.n; }
    __builtin_amdgcn_wave_barrier();
    while (nf > 0) {
        const int take = nf < WAVE ? nf : WAVE;
        const int base = nf - take;
        nf = base;
        int rlo = 0, rhi = 0;
        if (lane < take) { rlo = L.f_lo[base + lane]; rhi = L.f_hi[base + lane]; }
        const bool live = lane < take && rhi > rlo;
        const bool small = live && rhi - rlo <= WAVE;
        bool is_leaf = small;
        if (is_leaf) {
            const float* b = pm.leaf_box + 6 * (size_t)(rlo >> 5);
            is_leaf = !(b[0] > ghx || b[3] < glx || b[1] > ghy || b[4] < gly || b[2] > ghz || b[5] < glz);
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
        __builtin_amdgcn_wave_barrier();
        if (n_leaf + c_leaf > T_LEAVES || n_piv + c_piv > T_PIVOTS || nf + c_l + c_r > FRONT) return -1;
        if (is_leaf) T.leaf[n_leaf + (int)prefix_rank(m_leaf)] = ((uint32_t)rlo << 7) | (uint32_t)(rhi - rlo);
        if (piv_in) T.piv[n_piv + (int)prefix_rank(m_piv)] = (uint32_t)mid;
        if (go_l) { const int p = nf + (int)prefix_rank(m_l); L.f_lo[p] = rlo; L.f_hi[p] = mid; }
        if (go_r) { const int p = nf + c_l + (int)prefix_rank(m_r); L.f_lo[p] = mid + 1; L.f_hi[p] = rhi; }
        n_leaf += c_leaf; n_piv += c_piv; nf += c_l + c_r;
        __builtin_amdgcn_wave_barrier();
    }
    return n_leaf | (n_piv << 16);
}

// The photons of a step are handed from the lanes that fetched them to every lane through LDS: each lane writes its
// photon's coordinates (x[64], y[64], z[64]; a sum pass also power and direction), then all lanes read the same
// addresses -- a broadcast read, which costs one LDS instruction per four photons and component where a
// lane-to-scalar move costs one vector instruction per photon and component.  The staging area lies over the
// frontier arrays of the tree walks (f_lo, f_hi, f_d2: 3 KB, contiguous), which no pass uses.
static_assert(offsetof(WaveLds, f_hi) == offsetof(WaveLds, f_lo) + FRONT * 4 && offsetof(WaveLds, f_d2) == offsetof(WaveLds, f_hi) + FRONT * 4 && FRONT * 12 >= WAVE * 9 * 4,
              "the staging area of the passes lies over the three frontier arrays");
__device__ __forceinline__ float* stage_xyz() { return reinterpret_cast<float*>(g_L.f_lo); }          // x[64] y[64] z[64]
__device__ __forceinline__ float* stage_aux() { return reinterpret_cast<float*>(g_L.f_lo) + 3 * WAVE; } // [6][64]: power r, g, b, direction x, y, z
__device__ __forceinline__ void stage_put(const float4& p, int lane, int n)
{
    float* s = stage_xyz();
    __builtin_amdgcn_wave_barrier();       // (the reads of the step before are done: LDS operations of a wave complete in order)
    s[lane] = lane < n ? p.x : 1.0e18f; s[WAVE + lane] = p.y; s[2 * WAVE + lane] = p.z;
    __builtin_amdgcn_wave_barrier();
}

struct D2x4 { f2 a, b; };      // squared distances to photons j .. j + 3
struct Xyz4 { float4 x, y, z; };   // coordinates of photons j .. j + 3 of the staged step
__device__ __forceinline__ Xyz4 quad_load(int j)          // j a multiple of 4, j + 3 < 64
{
    const float* s = stage_xyz();
    return Xyz4{*reinterpret_cast<const float4*>(s + j), *reinterpret_cast<const float4*>(s + WAVE + j), *reinterpret_cast<const float4*>(s + 2 * WAVE + j)};
}
// squared distances from (qx, qy, qz) to four photons; the operation order per photon is that of the per-request
// kernel, so the values are the same floats
__device__ __forceinline__ D2x4 quad_d2(const Xyz4& c, float qx, float qy, float qz)
{
    const f2 q_x = {qx, qx}, q_y = {qy, qy}, q_z = {qz, qz};
    const f2 dxa = q_x - f2{c.x.x, c.x.y}, dya = q_y - f2{c.y.x, c.y.y}, dza = q_z - f2{c.z.x, c.z.y};
    const f2 dxb = q_x - f2{c.x.z, c.x.w}, dyb = q_y - f2{c.y.z, c.y.w}, dzb = q_z - f2{c.z.z, c.z.w};
    return D2x4{dxa * dxa + dya * dya + dza * dza, dxb * dxb + dyb * dyb + dzb * dzb};
}

// Histogram bin of a squared distance: bin 0 holds everything below `lower` (distances the triangle inequality
// already puts inside the set), bins 1 .. T_BINS - 2 are linear over [lower, bound), the last bin takes everything
// from `bound` on (and the filler photons of a short step), so no photon needs a comparison with the bound.
// The position d2 * scale + offset is monotone in d2 as evaluated (one fused multiply-add with a positive factor: a single
// rounding of a monotone function), which is all the selection needs; every pass forms it with this one function.
struct Shell { float scale, offset; };
__device__ __forceinline__ Shell shell_of(float bound, float lower)
{
    Shell sh;
    sh.scale = ((float)(T_BINS - 2) * 0.999999f) / (bound - lower);
    sh.offset = 1.f - lower * sh.scale;
    return sh;
}
__device__ __forceinline__ f2 shell_pos(f2 d2, const Shell& sh) { return __builtin_elementwise_fma(d2, f2{sh.scale, sh.scale}, f2{sh.offset, sh.offset}); }
__device__ __forceinline__ int shell_bin_of(float pos) { return (int)__builtin_amdgcn_fmed3f(pos, 0.f, (float)(T_BINS - 1)); }
// The histograms: lanes l and l + 32 share the word bin * 32 + l, the lower lane counting in its low half.  (A
// wave's LDS operation goes through in two halves of 32 lanes, so the two never meet in one pass.)
__device__ __forceinline__ uint32_t* hist_mine(uint32_t* cell, int lane) { return cell + (lane & 31); }
__device__ __forceinline__ uint32_t hist_unit(int lane) { return lane < 32 ? 1u : 0x10000u; }
__device__ __forceinline__ void hist_add(uint32_t* mine, uint32_t unit, int bin) { atomicAdd(mine + bin * 32, unit); }
__device__ __forceinline__ uint32_t hist_count(const uint32_t* mine, int lane, int bin) { const uint32_t w = mine[bin * 32]; return lane < 32 ? w & 0xFFFFu : w >> 16; }

// Copy to `dst` the leaves of `src` that some lane has to look at photon by photon: those with a photon
// possibly at a squared distance in [lo_lim, hi_lim] from the lane's point (hi_lim < 0: the lane takes no
// part).  The distance range of a box is taken a little wide, so rounding never drops a leaf that matters.
// Returns the number kept; `dropped` gets the photons in the leaves left out.  dst may be src.
// 64 leaves at a time: lane j fetches the j-th box, then every lane measures its own point against each
// box in turn.
__device__ __attribute__((noinline)) int select_leaves(const float* __restrict__ leaf_box, const uint32_t* src, uint32_t* dst, float qx, float qy, float qz,
                                                       float lo_lim, float hi_lim, int n_leaf, int* dropped)
{
    const int lane = lane_id();
    int out = 0, left_out = 0;
    for (int s0 = 0; s0 < n_leaf; s0 += WAVE) {
        const int n = n_leaf - s0 < WAVE ? n_leaf - s0 : WAVE;
        const uint32_t e = src[s0 + (lane < n ? lane : n - 1)];
        const float* b = leaf_box + 6 * (size_t)((e >> 7) >> 5);
        const float b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3], b4 = b[4], b5 = b[5];
        unsigned long long keep = 0;
        for (int j = 0; j < n; j++) {
            const float lx = bcast(b0, j) - qx, hx = qx - bcast(b3, j), ly = bcast(b1, j) - qy, hy = qy - bcast(b4, j), lz = bcast(b2, j) - qz,
                        hz = qz - bcast(b5, j);
            const float nx = fmaxf(fmaxf(lx, hx), 0.f), ny = fmaxf(fmaxf(ly, hy), 0.f), nz = fmaxf(fmaxf(lz, hz), 0.f);       // to the nearest point of the box
            const float fx = fmaxf(-lx, -hx), fy = fmaxf(-ly, -hy), fz = fmaxf(-lz, -hz);                                    // to its farthest corner
            const float dmin2 = (nx * nx + ny * ny + nz * nz) * 0.9999f, dmax2 = (fx * fx + fy * fy + fz * fz) * 1.0001f;
            if (__ballot(dmin2 <= hi_lim && dmax2 >= lo_lim)) keep |= 1ull << j;
        }
        __builtin_amdgcn_wave_barrier();
        if ((keep >> lane) & 1ull) dst[out + (int)prefix_rank(keep)] = e;      // out <= s0: never ahead of the entries still to be read
        if (lane < n && !((keep >> lane) & 1ull)) left_out += (int)(e & 127u);
        out += __builtin_popcountll(keep);
        __builtin_amdgcn_wave_barrier();
    }
    for (int off = 32; off > 0; off >>= 1) left_out += __shfl_xor(left_out, off);
    *dropped = left_out;
    return out;
}

__device__ __forceinline__ uint32_t* lane_tables() { return reinterpret_cast<uint32_t*>(g_L.key); }   // [row][lane] over the candidates' distances

// 3a: the lane's histogram (see hist_add) of the squared distances from its point to the photons of the list
// `radius2` < 3e38 in some lane (`any_radius`): that lane's bound is the search radius itself, which is exclusive
// (Kdtree.cpp:77,82) -- the bins' last edge lies a rounding error beyond it, so such a lane compares explicitly.
template <bool any_radius>
__device__ __attribute__((noinline)) void pass_hist(const float4* __restrict__ pos, const uint32_t* leaf, float qx, float qy, float qz, Shell sh, float radius2,
                                                    int n_leaf, int n_piv)
{
    uint32_t* const cell = lane_tables();
    const int lane = lane_id();
#pragma unroll 8
    for (int b = 0; b < T_BINS / 2; b++) cell[b * WAVE + lane] = 0u;
    __builtin_amdgcn_wave_barrier();
    n_leaf = __builtin_amdgcn_readfirstlane(n_leaf); n_piv = __builtin_amdgcn_readfirstlane(n_piv);      // (the same in every lane: let the compiler know)
    const int n_steps = n_leaf + (n_piv + WAVE - 1) / WAVE;
    uint32_t* const mine = hist_mine(cell, lane);
    const uint32_t unit = hist_unit(lane);
    int n = 0, at = 0, n_next = 0;
    float4 p = n_steps > 0 ? step_photons(pos, leaf, 0, n_leaf, n_piv, lane, n, at) : float4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < n_steps; s++, n = n_next) {
        stage_put(p, lane, n);
        const int n_now = __builtin_amdgcn_readfirstlane(n);
        // the next step's photons are on their way while this step's are used
        p = step_photons(pos, leaf, s + 1 < n_steps ? s + 1 : s, n_leaf, n_piv, lane, n_next, at);      // (unconditional: a fetch inside a branch is waited for at its end)
        // (the coordinates of the next four photons are read from LDS while the current four are binned; two register
        // sets take turns, so nothing is copied)
        auto bin4 = [&](const Xyz4& c) {
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
        };
        Xyz4 c0 = quad_load(0), c1;
        for (int j0 = 0; j0 < n_now; j0 += 8) {
            c1 = quad_load(j0 + 4);
            bin4(c0);
            c0 = quad_load((j0 + 8) & (WAVE - 1));
            if (j0 + 4 < n_now) bin4(c1);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// 3b: cell[i][lane] = the i-th squared distance of the lane that falls into its bin `crit` (1 .. T_BINS - 2; < 0: the
// lane takes no part); returns their number
template <bool any_radius>
__device__ __attribute__((noinline)) int pass_crit(const float4* __restrict__ pos, const uint32_t* leaf, float qx, float qy, float qz, Shell sh, float radius2, int crit,
                                                   int n_leaf, int n_piv)
{
    uint32_t* const cell = lane_tables();
    const int lane = lane_id();
    int cnt = 0;
    n_leaf = __builtin_amdgcn_readfirstlane(n_leaf); n_piv = __builtin_amdgcn_readfirstlane(n_piv);      // (the same in every lane: let the compiler know)
    const int n_steps = n_leaf + (n_piv + WAVE - 1) / WAVE;
    // A position lies in bin `crit` (1 <= crit <= T_BINS - 2) when pos - crit is in [0, 1).  The difference is exact there
    // (both are multiples of 2^-23 below 128), and a float in [0, 1) is one whose bit pattern, read as an unsigned
    // number, is below that of 1.0: negative values and everything from 1.0 on compare above.  An idle lane subtracts
    // a number no position reaches.
    const float crit_f = crit >= 1 ? (float)crit : 1.0e30f;
    const f2 crit_2 = {crit_f, crit_f};
    constexpr uint32_t ONE = 0x3F800000u;
    int n = 0, at = 0, n_next = 0;
    float4 p = n_steps > 0 ? step_photons(pos, leaf, 0, n_leaf, n_piv, lane, n, at) : float4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < n_steps; s++, n = n_next) {
        stage_put(p, lane, n);
        const int n_now = __builtin_amdgcn_readfirstlane(n);
        // the next step's photons are on their way while this step's are used
        p = step_photons(pos, leaf, s + 1 < n_steps ? s + 1 : s, n_leaf, n_piv, lane, n_next, at);      // (unconditional: a fetch inside a branch is waited for at its end)
        // (two register sets take turns: a set is read from LDS while the other is worked on)
        auto pick4 = [&](const Xyz4& c) {
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
        };
        Xyz4 c0 = quad_load(0), c1;
        for (int j0 = 0; j0 < n_now; j0 += 8) {
            c1 = quad_load(j0 + 4);
            pick4(c0);
            c0 = quad_load((j0 + 8) & (WAVE - 1));
            if (j0 + 4 < n_now) pick4(c1);
        }
    }
    __builtin_amdgcn_wave_barrier();
    return cnt;
}

struct Sum3 { double s0, s1, s2; };

// 3c: the sums of main.cpp:262-291 (power * |cosine| * weight per channel) over the photons with d2 <= t_k: every
// photon at exactly the k-th distance belongs to the set (the caller has counted them), so the test is one comparison.
__device__ __attribute__((noinline)) Sum3 pass_sum(const float4* __restrict__ pos, const float* __restrict__ aux, const uint32_t* leaf, float qx, float qy, float qz, float t_k,
                                                   bool some, float nx, float ny, float nz, int n_leaf, int n_piv)
{
    const int lane = lane_id();
    const TermCoef tc = term_coef(t_k);
    double s0 = 0, s1 = 0, s2 = 0;
    n_leaf = __builtin_amdgcn_readfirstlane(n_leaf); n_piv = __builtin_amdgcn_readfirstlane(n_piv);      // (the same in every lane: let the compiler know)
    const int n_steps = n_leaf + (n_piv + WAVE - 1) / WAVE;
    const float lim = some ? t_k : -1.f;
    int n = 0, at = 0, n_next = 0;
    float4 p = float4{0.f, 0.f, 0.f, 0.f};
    float b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0, b5 = 0;
    if (n_steps > 0) {
        p = step_photons(pos, leaf, 0, n_leaf, n_piv, lane, n, at);
        const float __attribute__((address_space(1)))* ax = as_global(aux) + 6 * (size_t)at;
        b0 = ax[0]; b1 = ax[1]; b2 = ax[2]; b3 = ax[3]; b4 = ax[4]; b5 = ax[5];
    }
    float* const sa = stage_aux();
    for (int s = 0; s < n_steps; s++, n = n_next) {
        stage_put(p, lane, n);
        {
            sa[lane] = b0; sa[WAVE + lane] = b1; sa[2 * WAVE + lane] = b2; sa[3 * WAVE + lane] = b3; sa[4 * WAVE + lane] = b4; sa[5 * WAVE + lane] = b5;
            __builtin_amdgcn_wave_barrier();
        }
        const int n_now = __builtin_amdgcn_readfirstlane(n);
        // the next step's photons are on their way while this step's are used
        {      // (unconditional: a fetch inside a branch is waited for at its end)
            p = step_photons(pos, leaf, s + 1 < n_steps ? s + 1 : s, n_leaf, n_piv, lane, n_next, at);
            const float __attribute__((address_space(1)))* ax = as_global(aux) + 6 * (size_t)at;
            b0 = ax[0]; b1 = ax[1]; b2 = ax[2]; b3 = ax[3]; b4 = ax[4]; b5 = ax[5];
        }
        // power and direction of four photons are fetched with the coordinates, not inside the branch that uses them; two
        // register sets take turns, one being read from LDS while the other is worked on
        struct Aux4 { float4 pr, pg, pb, ux, uy, uz; };
        auto aux_load = [&](int j) {
            return Aux4{*reinterpret_cast<const float4*>(sa + j), *reinterpret_cast<const float4*>(sa + WAVE + j), *reinterpret_cast<const float4*>(sa + 2 * WAVE + j),
                        *reinterpret_cast<const float4*>(sa + 3 * WAVE + j), *reinterpret_cast<const float4*>(sa + 4 * WAVE + j), *reinterpret_cast<const float4*>(sa + 5 * WAVE + j)};
        };
        auto sum4 = [&](const Xyz4& c, const Aux4& a) {
            const D2x4 d = quad_d2(c, qx, qy, qz);
            const float4 &pr4 = a.pr, &pg4 = a.pg, &pb4 = a.pb, &ux4 = a.ux, &uy4 = a.uy, &uz4 = a.uz;
            {
                // Four photons at a time without a branch per photon: a photon outside the lane's ball contributes a
                // factor 0.  The four terms of a channel are added in float before they join the double sum.
                const bool in0 = d.a.x <= lim, in1 = d.a.y <= lim, in2 = d.b.x <= lim, in3 = d.b.y <= lim;
                if (__ballot(in0 || in1 || in2 || in3)) {
                    const f2 c2 = {tc.c2, tc.c2}, A = {tc.A, tc.A}, B = {tc.B, tc.B};
                    const f2 ea = c2 * d.a, eb = c2 * d.b;
                    const f2 wa = A + B * f2{__builtin_amdgcn_exp2f(ea.x), __builtin_amdgcn_exp2f(ea.y)}, wb = A + B * f2{__builtin_amdgcn_exp2f(eb.x), __builtin_amdgcn_exp2f(eb.y)};
                    const f2 n_x = {nx, nx}, n_y = {ny, ny}, n_z = {nz, nz};
                    const f2 ca = n_x * f2{ux4.x, ux4.y} + n_y * f2{uy4.x, uy4.y} + n_z * f2{uz4.x, uz4.y};
                    const f2 cb = n_x * f2{ux4.z, ux4.w} + n_y * f2{uy4.z, uy4.w} + n_z * f2{uz4.z, uz4.w};
                    const float sc0 = in0 ? fabsf(ca.x) * wa.x : 0.f, sc1 = in1 ? fabsf(ca.y) * wa.y : 0.f;
                    const float sc2 = in2 ? fabsf(cb.x) * wb.x : 0.f, sc3 = in3 ? fabsf(cb.y) * wb.y : 0.f;
                    const f2 sa = {sc0, sc1}, sb = {sc2, sc3};
                    const f2 r = f2{pr4.x, pr4.y} * sa + f2{pr4.z, pr4.w} * sb, g = f2{pg4.x, pg4.y} * sa + f2{pg4.z, pg4.w} * sb, b = f2{pb4.x, pb4.y} * sa + f2{pb4.z, pb4.w} * sb;
                    s0 += (double)(r.x + r.y);                   // main.cpp:286 without the hit colour, which the caller multiplies in
                    s1 += (double)(g.x + g.y);
                    s2 += (double)(b.x + b.y);
                }
            }
        };
        Xyz4 c0 = quad_load(0), c1;
        Aux4 a0 = aux_load(0), a1;
        for (int j0 = 0; j0 < n_now; j0 += 8) {
            c1 = quad_load(j0 + 4); a1 = aux_load(j0 + 4);
            sum4(c0, a0);
            c0 = quad_load((j0 + 8) & (WAVE - 1)); a0 = aux_load((j0 + 8) & (WAVE - 1));
            if (j0 + 4 < n_now) sum4(c1, a1);
        }
    }
    return Sum3{s0, s1, s2};
}

__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(2, 2))) void gather_tile_kernel(PhotonMapView pm, TileArgs ta, GatherParams gp, double* __restrict__ acc)
{
    uint32_t* const cell = lane_tables();
    const int lane = lane_id();
    const rtq::Map map = rtq::map_units(ta.cnt, ta.shard_cap, WAVE);
    const uint32_t n_windows = map.total;
    if (lane == 0) atomicAdd(ta.windows_done + 1, 1ull);
    unsigned long long pr[TP_SLOTS] = {};
    const unsigned long long t_begin = ta.prof ? clock64() : 0ull;
#define TILE_PROF(slot, expr) do { if (ta.prof) { const unsigned long long t_ = clock64(); expr; pr[slot] += clock64() - t_; } else { expr; } } while (0)
    while (true) {
        unsigned long long w0 = 0;
        if (lane == 0) w0 = atomicAdd(ta.next, 1ull);
        const uint32_t win = (uint32_t)__builtin_amdgcn_readfirstlane((unsigned)w0);
        const uint32_t win_hi = (uint32_t)__builtin_amdgcn_readfirstlane((unsigned)(w0 >> 32));
        if (win_hi || win >= n_windows) break;       // (the high half is zero for any real queue)
        unsigned shard;
        uint32_t first, n_shard;
        rtq::locate(map, win, WAVE, shard, first, n_shard);
        const unsigned long long slot0 = (unsigned long long)shard * ta.shard_cap + first;
        const int n_req = (int)((ta.recs[slot0].flags >> GATHER_COUNT_SHIFT) & 127u);
        const bool have = lane < n_req;
        const GatherRec& r = ta.recs[slot0 + (have ? lane : 0)];
        const float qx = (float)r.pos[0], qy = (float)r.pos[1], qz = (float)r.pos[2];
        const uint32_t flags = r.flags & 0xFFu;
        // Subsurface requests: their surface estimate is one of four gathers (main.cpp:260-388) and the only one in this
        // map; it is formed here with the group's, written into the record, and the request handed on for the rest.
        const bool sss = (flags & GATHER_SSS) != 0;
        bool redo = have && ((flags & ~(GATHER_SSS | GATHER_MARCH)) != 0 || (sss && !ta.sss_surface) || ta.all_redo || pm.n >= (1ll << 24));
        bool todo = have && !redo;
        pr[TP_WINDOWS]++;
#define TILE_COUNT(slot, cond) do { if (ta.prof) pr[slot] += __builtin_popcountll(__ballot(cond)); } while (0)
        TILE_COUNT(TP_R_FLAGS, redo);
        // A window mostly holds one patch of neighbouring hit points, but it may hold a few (a tile across a
        // silhouette, two producers' requests in one window): each round takes the requests close to one of
        // those still waiting; what is left after ROUNDS rounds, and rounds of only a few requests, go to the
        // per-request kernel.
        // (measured on the water frame: fastest at 12 requests for k = 1000 -- 8: +8 %, 16: +2 %, 24: +9 % -- and flat from 4 to 8 for k = 100)
        const int ROUNDS = gp.max_rounds;
        const int MIN_LANES = gp.min_lanes > 0 ? gp.min_lanes : (gp.k >= 400 ? 12 : 6);
        for (int round = 0; round < ROUNDS; round++) {
            const unsigned long long m_todo = __ballot(todo);
            if (!m_todo) break;
            // ---- 1. one request answered exactly: the middle one of those waiting
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
            // a first look at who is near, with the radius unknown: if only a few are within the widest possible
            // ball they are not worth a round of their own
            bool act = todo && d_c <= sqrtf(gp.r2);
            if (__builtin_popcountll(__ballot(act)) < MIN_LANES) {
                TILE_COUNT(TP_R_FEW, act);
                redo = redo || act;
                todo = todo && !act;
                continue;
            }
            __builtin_amdgcn_wave_barrier();
            float r_c;
            TILE_PROF(TP_KTH, r_c = kth_distance(pm, cx, cy, cz, gp.k, gp.r2, gp.front_cap));
            __builtin_amdgcn_wave_barrier();
            float bound = gp.r2, lower = 0.f;
            if (r_c >= 0.f) {
                act = todo && d_c <= r_c;                            // this round: the requests inside the centre's own ball
                // r_k(q) lies within |q - q_c| of r_c (triangle inequality).  For a request far from the centre that
                // range is wide -- up to four times the ball's area to search -- while the photon density, and with it
                // r_k, changes little over a ball's width: such a request is given the range r_c * (1 -+ OPT_RANGE)
                // instead.  Both ends are checked by the counts below (k photons inside the outer radius, fewer than k
                // inside the inner one); a request that fails either goes to the per-request kernel.
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
            // ---- 2. the leaves and pivots inside the box around the round's balls
            const float rad = act ? sqrtf(bound) * 1.0001f + 1e-6f : 0.f;
            const float big = 3.0e38f;
            const float glx = wave_min_f(act ? qx - rad : big), gly = wave_min_f(act ? qy - rad : big), glz = wave_min_f(act ? qz - rad : big);
            const float ghx = wave_max_f(act ? qx + rad : -big), ghy = wave_max_f(act ? qy + rad : -big), ghz = wave_max_f(act ? qz + rad : -big);
            int lists;
            TILE_PROF(TP_WALK, lists = group_walk(pm, glx, gly, glz, ghx, ghy, ghz));
            if (lists < 0) { TILE_COUNT(TP_R_WALK, act); redo = redo || act; continue; }
            int n_leaf = lists & 0xFFFF;
            const int n_piv = lists >> 16;
            // the walk lists what touches the box around the balls; the corners of that box touch no ball
            int n_out;
            n_leaf = select_leaves(pm.leaf_box, g_T.leaf, g_T.leaf, qx, qy, qz, 0.f, act ? bound : -1.f, n_leaf, &n_out);
            pr[TP_ROUNDS]++; pr[TP_LANES] += __builtin_popcountll(__ballot(act)); pr[TP_STEPS] += n_leaf + (n_piv + WAVE - 1) / WAVE;

            const Shell sh = shell_of(bound, lower);
            const float radius2 = act && !(bound < gp.r2) ? gp.r2 : 3.0e38f;      // lanes bounded by the search radius itself
            const bool any_radius = __ballot(radius2 < 3.0e38f) != 0;
            // ---- 3a. per-lane histogram of the squared distances below the lane's bound
            // leaves that lie inside every request's inner ball count as a whole; the others go photon by photon
            int n_inside;
            const int n_hist = select_leaves(pm.leaf_box, g_T.leaf, g_T.part, qx, qy, qz, act ? lower : 3.0e38f, act ? 3.0e38f : -1.f, n_leaf, &n_inside);
            if (any_radius) TILE_PROF(TP_HIST, pass_hist<true>(pm.pos, g_T.part, qx, qy, qz, sh, radius2, n_hist, n_piv));
            else TILE_PROF(TP_HIST, pass_hist<false>(pm.pos, g_T.part, qx, qy, qz, sh, radius2, n_hist, n_piv));
            pr[TP_N_HIST] += n_hist + (n_piv + WAVE - 1) / WAVE;
            uint32_t* const mine = hist_mine(cell, lane);
            if (act) hist_add(mine, hist_unit(lane) * (uint32_t)n_inside, 0);
            // the bin holding the lane's k-th smallest distance (or its largest, if fewer than k lie within the bound)
            __builtin_amdgcn_wave_barrier();
            uint32_t total = 0;                                   // (the last bin holds what lies beyond the bound)
#pragma unroll 4
            for (int b = 0; b < T_BINS - 1; b++) total += hist_count(mine, lane, b);
            const uint32_t k_eff = total < (uint32_t)gp.k ? total : (uint32_t)gp.k;
            // a bound derived from the neighbour must contain k photons; if it somehow does not, the per-query kernel decides
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
            // only the leaves that reach into some request's critical shell can hold a distance of its bin
            const float shell_lo = crit > 0 ? lower + (float)(crit - 1) / sh.scale : 0.f, shell_hi = crit > 0 ? lower + (float)crit / sh.scale : lower;
            const int n_crit = select_leaves(pm.leaf_box, g_T.leaf, g_T.part, qx, qy, qz, some ? shell_lo * 0.9999f : 3.0e38f, some ? shell_hi * 1.0001f : -1.f, n_leaf,
                                             &n_out);
            if (any_radius) TILE_PROF(TP_CRIT, cnt = pass_crit<true>(pm.pos, g_T.part, qx, qy, qz, sh, radius2, some ? crit : -1, n_crit, n_piv));
            else TILE_PROF(TP_CRIT, cnt = pass_crit<false>(pm.pos, g_T.part, qx, qy, qz, sh, radius2, some ? crit : -1, n_crit, n_piv));
            pr[TP_N_CRIT] += n_crit + (n_piv + WAVE - 1) / WAVE;
            const unsigned long long t_rank = ta.prof ? clock64() : 0ull;
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
                    const bool mine = j < cnt;
#pragma unroll
                    for (int u = 0; u < 8; u++) rank[u] += (mine && (o < e[u] || (o == e[u] && j < i0 + u))) ? 1 : 0;
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
            if (ta.prof) pr[TP_RANK] += clock64() - t_rank;
            // ---- 3c. the estimate over the set (main.cpp:262-295)
            Sum3 sum;
            const unsigned long long t_sum = ta.prof ? clock64() : 0ull;
            n_leaf = select_leaves(pm.leaf_box, g_T.leaf, g_T.leaf, qx, qy, qz, 0.f, summed ? t_k : -1.f, n_leaf, &n_out);      // the final balls are smaller than the bounds
            sum = pass_sum(pm.pos, pm.aux, g_T.leaf, qx, qy, qz, t_k, summed, (float)r.normal[0], (float)r.normal[1], (float)r.normal[2], n_leaf, n_piv);
            pr[TP_N_SUM] += n_leaf + (n_piv + WAVE - 1) / WAVE;
            if (ta.prof) pr[TP_SUM] += clock64() - t_sum;
            if (summed) {
                const double sc = 1.0 / (gp.norm * t_k);           // main.cpp:293
                double rgb[3] = {r.color[0] * sum.s0 * sc, r.color[1] * sum.s1 * sc, r.color[2] * sum.s2 * sc};
                if (sss) {
                    GatherRec* const w = const_cast<GatherRec*>(&r);
                    w->normal[0] = rgb[0]; w->normal[1] = rgb[1]; w->normal[2] = rgb[2];
                    w->flags = r.flags | GATHER_SURFACE_DONE;
                    redo = true;
                } else {
                    clip3(rgb);                                    // main.cpp:391
                    for (int c = 0; c < 3; c++) atomicAdd(&acc[3 * (size_t)r.pixel + c], r.weight * rgb[c]);
                }
            } else if (sss && act && !split_tie) {
                redo = true;      // (no surface photon in reach: nothing to note, but the volume terms are still to come)
            }
            __builtin_amdgcn_wave_barrier();
        }
        TILE_COUNT(TP_R_LEFT, todo);
        redo = redo || todo;
        // ---- requests left to the per-query kernel
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
    if (ta.prof && lane == 0) {
        pr[TP_BUSY] = clock64() - t_begin;
        for (int i = 0; i < TP_BUSY_MAX; i++) atomicAdd(&ta.prof[i], pr[i]);
        atomicMax(&ta.prof[TP_BUSY_MAX], pr[TP_BUSY]);
    }
}

}  // namespace rtg
