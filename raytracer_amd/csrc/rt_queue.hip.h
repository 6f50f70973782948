// Queues in HBM that many waves append to at once (bounce queue, gather queue).
//
// One returning atomic on a single word serves only ~90 requests per microsecond chip-wide, fewer
// than the waves of a frame issue, so a queue is split into SHARDS regions of equal capacity whose
// fill counters sit on separate 64-byte lines.  A wave appends (ballot-compacted, one atomic per
// wave) to the shard its grid-wide wave index selects.  Consumers read all counters once, number
// the work units shard by shard and map a unit index back to (shard, element).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtq {

constexpr int SHARDS = 32;

struct Counts {
    unsigned long long n[SHARDS][8];   // n[s][0] = elements appended to shard s
};

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ unsigned wave_index() { return blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); }
__device__ __forceinline__ unsigned n_waves() { return gridDim.x * (blockDim.x >> 6); }

__device__ __forceinline__ unsigned prefix_rank(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// Append `rec` of every lane with `want` set.  Must be called by all lanes of the wave.
template <class T>
__device__ __forceinline__ void push(T* q, Counts* cnt, unsigned long long shard_cap, bool want, const T& rec, unsigned long long* overflow)
{
    unsigned long long m = __ballot(want);
    if (m == 0) return;
    const unsigned shard = wave_index() & (SHARDS - 1);
    const int leader = __builtin_ctzll(m);
    unsigned long long base = 0;
    if ((int)lane_id() == leader) base = atomicAdd(&cnt->n[shard][0], (unsigned long long)__builtin_popcountll(m));
    unsigned lo = __shfl((unsigned)base, leader), hi = __shfl((unsigned)(base >> 32), leader);
    base = ((unsigned long long)hi << 32) | lo;
    if (want) {
        unsigned long long p = base + prefix_rank(m);
        if (p < shard_cap) q[(unsigned long long)shard * shard_cap + p] = rec;
        else *overflow = 1;
    }
}

// Consumer view, held per lane: lane s describes shard s.
struct Map {
    uint32_t incl;    // work units in shards 0..s
    uint32_t n;       // elements in shard s
    uint32_t total;   // work units in the whole queue
};

// unit = elements per work unit (64: one element per lane; 1: one element per wave).
// Must be called by all lanes of the wave.
__device__ __forceinline__ Map map_units(const Counts* cnt, unsigned long long shard_cap, uint32_t unit)
{
    const unsigned lane = lane_id();
    unsigned long long n = lane < (unsigned)SHARDS ? cnt->n[lane][0] : 0ull;
    if (n > shard_cap) n = shard_cap;   // an overflowing shard holds shard_cap elements (the overflow flag is set by the producer)
    Map m;
    m.n = (uint32_t)n;
    uint32_t units = (m.n + unit - 1) / unit;
    uint32_t incl = units;
    for (int off = 1; off < SHARDS; off <<= 1) {
        uint32_t v = __shfl_up(incl, off);
        if ((int)lane >= off) incl += v;
    }
    m.incl = incl;
    m.total = __shfl(incl, SHARDS - 1);
    return m;
}

// Work unit u (< total) -> shard, index of its first element in the shard, elements in the shard.
// Must be called by all lanes of the wave with the same u.
__device__ __forceinline__ void locate(const Map& m, uint32_t u, uint32_t unit, unsigned& shard, uint32_t& first, uint32_t& n_shard)
{
    unsigned long long above = __ballot(lane_id() < (unsigned)SHARDS && m.incl > u);
    shard = (unsigned)__builtin_ctzll(above);
    uint32_t incl_s = __shfl(m.incl, (int)shard);
    n_shard = __shfl(m.n, (int)shard);
    uint32_t units_s = (n_shard + unit - 1) / unit;
    first = (u - (incl_s - units_s)) * unit;
}

// Dynamic assignment of a queue's work units to the waves of a persistent grid, for consumers whose
// units differ a lot in cost.  One draw counter per shard (a single counter word would be the
// bottleneck): a wave draws from its home shard until that is exhausted, then reads all draw
// counters at once and moves to the next shard that still has units.
struct Drawer {
    unsigned home;
};
__device__ __forceinline__ Drawer drawer_init() { return Drawer{wave_index() & (SHARDS - 1)}; }

// Must be called by all lanes of the wave.  Returns false when no unit is left anywhere.
__device__ __forceinline__ bool draw(const Map& m, Counts* next, uint32_t unit, Drawer& d, uint32_t& u)
{
    const unsigned lane = lane_id();
    const uint32_t my_units = (m.n + unit - 1) / unit;   // lane s: units of shard s
    while (true) {
        const uint32_t incl_s = __shfl(m.incl, (int)d.home);
        const uint32_t units_s = __shfl(my_units, (int)d.home);
        if (units_s > 0) {
            unsigned long long k = 0;
            if (lane == 0) k = atomicAdd(&next->n[d.home][0], 1ull);
            k = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(k >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)k);
            if (k < units_s) {
                u = incl_s - units_s + (uint32_t)k;
                return true;
            }
        }
        unsigned long long taken = lane < (unsigned)SHARDS ? __hip_atomic_load(&next->n[lane][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
        const unsigned long long avail = __ballot(lane < (unsigned)SHARDS && taken < my_units);
        if (avail == 0) return false;
        const unsigned long long after = d.home + 1 < 64 ? avail & ~((1ull << (d.home + 1)) - 1ull) : 0ull;
        d.home = (unsigned)__builtin_ctzll(after ? after : avail);
    }
}

inline unsigned long long host_total(const Counts& c, unsigned long long shard_cap)
{
    unsigned long long t = 0;
    for (int s = 0; s < SHARDS; s++) t += c.n[s][0] < shard_cap ? c.n[s][0] : shard_cap;
    return t;
}

}  // namespace rtq
