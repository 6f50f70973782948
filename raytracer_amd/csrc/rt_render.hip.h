// Pixel path as a wavefront of queues in HBM (replaces the recursion of getColorAt, main.cpp:158-392):
//
//   render_primary_kernel   camera rays (main.cpp:753-790) + BVH::Shoot -> hit queue level 1
//   per level d = 1..D:
//   shade_kernel            per queued hit: colour, normal, material branch (main.cpp:164-258);
//                           mirror / glass hits append their reflection / refraction rays to the
//                           ray queue, diffuse hits append a gather request (all ballot-compacted)
//   trace_queue_kernel      per queued ray: BVH::Shoot, append the next level's hit.  Tracing from
//                           its own compacted queue keeps all 64 lanes busy in the expensive part
//                           (in a hit queue only the mirror / glass entries have rays to trace)
//   gather_kernel           photon k-NN + radiance estimate per request (rt_gather.hip.h)
//   resolve_kernel          average, clip, quantise (main.cpp:853-859, IO.cpp:55-65)
//
// A path's contribution is carried as a scalar throughput `weight` (products of (2 - special),
// (special - 1) or 1, main.cpp:200-203,251) and added to its pixel with a double atomic; the
// per-level Color::clip of the recursion only ever sees sums of weights <= 1 times values <= 1.
#pragma once
#include <hip/hip_runtime.h>

#include "rt_gather_tile.hip.h"
#include "rt_kernels.hip.h"
#include "rt_queue.hip.h"

namespace rtr {

using rtk::BLOCK;
using rtk::LaneStack;
using rtk::WAVE;

constexpr int MAX_LEVELS = 8;

struct HitRec {
    double pos[3];      // intersection position handed to getColorAt
    double dir[3];      // direction of the ray that found it
    double weight;
    uint32_t pixel;     // index in the call's row block
    int32_t obj;        // object to shade
};

// a secondary ray waiting to be traced
struct RayRec {
    double org[3];      // hit_pos of the surface it leaves (main.cpp:162)
    double dir[3];      // direction Shoot is called with (a glossy sample's jittered direction, main.cpp:183-188)
    double cont[3];     // direction the path goes on with: the un-jittered one (main.cpp:195-196, 246-247)
    double weight;      // throughput of the path once this ray is followed
    uint32_t pixel;     // bit 31: refraction ray (child position uses accuracy + t, main.cpp:246)
    int32_t obj;        // object the ray leaves (HEAD recursion shades it again, main.cpp:198,249)
};
constexpr uint32_t RAY_REFRACT = 0x80000000u;

// The hit queues are filled in windows of 64 slots, one per producing wave (an 8x8 pixel tile, or the
// hits of 64 queued rays), so that a consuming wave works on hit points that lie together; the
// gather requests it queues then form one group.  The first record of a window carries the number of
// hits in bits 25..31 of `pixel` (a pixel index stays below 2^25: a call handles at most ~4M camera rays).
constexpr uint32_t HIT_PIXEL_MASK = (1u << 25) - 1u;
// (`spread`: what picks the shard.  A producer whose waves take work as they become free passes the number of the unit
// it works on, so the shards fill evenly whichever waves did the work.)
__device__ __forceinline__ void push_hits(HitRec* q, rtq::Counts* cnt, unsigned long long shard_cap, bool want, HitRec& rec, unsigned long long* overflow,
                                          unsigned spread = rtq::wave_index())
{
    unsigned long long m = __ballot(want);
    if (m == 0) return;
    const unsigned shard = spread & (rtq::SHARDS - 1);
    const int leader = __builtin_ctzll(m);
    unsigned long long base = 0;
    if ((int)rtq::lane_id() == leader) {
        base = atomicAdd(&cnt->n[shard][0], (unsigned long long)WAVE);
        rec.pixel |= (uint32_t)__builtin_popcountll(m) << 25;
    }
    unsigned lo = __shfl((unsigned)base, leader), hi = __shfl((unsigned)(base >> 32), leader);
    base = ((unsigned long long)hi << 32) | lo;
    if (want) {
        if (base + WAVE <= shard_cap) q[(unsigned long long)shard * shard_cap + base + rtq::prefix_rank(m)] = rec;
        else *overflow = 1;
    }
}

struct RenderCounters {
    rtq::Counts hits[MAX_LEVELS + 2];   // hits[d] = fill of the bounce queue of level d (1-based)
    rtq::Counts rays[MAX_LEVELS + 2];   // rays[d] = fill of the ray queue produced by shading level d
    rtq::Counts ray_draws[MAX_LEVELS + 2];   // chunks of rays[d] handed out to the tracing waves
    rtq::Counts gather;                 // gather queue: [0] slots (windows of 64 per shading wave), [1] requests
    unsigned long long overflow_stack, overflow_queue;
    unsigned long long secondary_rays;
    unsigned long long next_gather;     // draw counter of the per-request gather kernel
    unsigned long long next_window;     // draw counter of the group gather kernel
    unsigned long long n_redo;          // requests the group kernel left to the per-request kernel
    unsigned long long tile_progress[2];   // of the group gather kernel: [0] windows finished (their hand-overs are written), [1] waves that have begun
    unsigned long long work_inner, work_prims;   // inner-node visits and primitive tests (counting instantiations only)
};

struct ShadeData {
    const rtf::ObjShade* objs;
    const rtf::TexDesc* maps;
    const double* texels;
};

struct RenderArgs {
    double cam_pos[3], cam_dir[3], cam_right[3], cam_down[3];
    int width, height, row_begin, row_end;
    int col_begin, col_end;   // the columns this launch renders (a block of rows is split by columns when its rays outgrow the queues)
    int band_rows, n_parts, part;
    int aadepth;
    double accuracy;
    int head_quirk, use_photon_map;
    uint32_t seed;      // of the counter-based draws (glossy rays, lens jitter)
    int dof_samples;    // main.cpp:35 (1: no depth of field)
    double focal_len, aperture;
    // Without a photon map the shading of a hit does one thing only: spawn the rays of a reflecting or refracting surface
    // (main.cpp:260: the diffuse term IS the photon estimate).  obj_live[object] != 0 marks the objects whose hits spawn
    // rays; hits on the others are not queued for a shading pass that would drop them.  Every ray is traced as before.
    // Null when a photon map is in use: every hit is shaded then.
    const unsigned char* obj_live;
};

__device__ __forceinline__ int rtk_wave_max(int v)
{
    for (int off = 32; off > 0; off >>= 1) { int o = __shfl_xor(v, off); v = o > v ? o : v; }
    return v;
}

// DOF (depth of field compiled in) is a template parameter: its sample loops around the traversal cost the kernel 60
// registers (147 instead of 87 in f64), and the reference's default build has no depth of field (main.cpp:35).
template <class R, bool TRI_ONLY, int MODE, bool DOF, bool COUNT = false>
__global__ __launch_bounds__(rtk::MAX_BLOCK) void render_primary_kernel(rtf::SceneView<R> sv, RenderArgs a, HitRec* __restrict__ hits,
                                                                       unsigned long long shard_cap, RenderCounters* ctr, int cap, int n_nodes,
                                                                       double* __restrict__ acc)
{
    extern __shared__ __align__(16) unsigned char lds[];
    const size_t node_bytes = MODE == rtk::TRAV_NODES ? rtk::node_lds_bytes<R>((size_t)n_nodes) : 0;
    if (MODE == rtk::TRAV_NODES) rtk::stage_nodes<R>(reinterpret_cast<uint4*>(lds), sv.nodes, n_nodes);
    LaneStack st = rtk::make_stack<R>(lds, cap, node_bytes);
    uint32_t ovf_stack = 0;
    unsigned long long ovf_queue = 0;
    rtk::WorkCount wc;
    const int rows = a.row_end - a.row_begin;
    const int tiles_x = (a.col_end - a.col_begin + 7) >> 3, tiles_y = (rows + 7) >> 3;
    // one 8x8 tile and one of its aadepth sample columns per wave (the waves of a tile are neighbours in a block,
    // so they share its nodes and triangles in L1); the dispatcher balances the load
    int tx, ty, aax;
    if (rtk::tile_of_wave<MODE == rtk::TRAV_NODES>(tiles_x, tiles_y, &tx, &ty, (unsigned)a.aadepth, &aax)) {
        // The aadepth waves of a tile share its samples.  With 2, 4 or 8 sample columns a wave takes ALL columns of a part
        // of the tile -- 8 x 4 pixels x 2, 4 x 4 pixels x 4 or a row of 8 pixels x 8 -- instead of one column of the whole
        // tile: the columns of a pixel are almost the same ray, so the wave's rays stay together for longer.
        int lx = st.lane & 7, ly = st.lane >> 3;
        if (a.aadepth == 8) {
            const int part = aax;
            aax = st.lane & 7;
            lx = st.lane >> 3; ly = part;
        } else if (a.aadepth == 4) {
            const int part = aax, px = st.lane >> 2;
            aax = st.lane & 3;
            lx = (part & 1) * 4 + (px & 3); ly = (part >> 1) * 4 + (px >> 2);
        } else if (a.aadepth == 2) {
            const int part = aax, px = st.lane >> 1;
            aax = st.lane & 1;
            lx = px & 7; ly = part * 4 + (px >> 3);
        }
        int x = a.col_begin + tx * 8 + lx, yl = ty * 8 + ly;
        bool inside = x < a.col_end && yl < rows;
        int yc = a.row_begin + yl;
        int y = a.band_rows > 0 ? ((yc / a.band_rows) * a.n_parts + a.part) * a.band_rows + yc % a.band_rows : yc;
        // the pixel's accumulator starts at zero (the kernels that add to it run after this one): no separate fill pass
        if (inside && aax == 0) {
            double* px = acc + ((long long)yl * a.width + x) * 3;
            px[0] = 0.0; px[1] = 0.0; px[2] = 0.0;
        }
        // The reference shoots aadepth^2 samples of which only aadepth are distinct (both offsets use
        // aax, main.cpp:775-785): without depth of field each distinct ray is traced once and weighted by
        // its multiplicity.  With it every one of the aadepth^2 x dof_samples rays has its own lens jitter.
        const bool dof = DOF;
        const int n_aay = dof ? a.aadepth : 1, n_dof = dof ? a.dof_samples : 1;
        for (int aay = 0; aay < n_aay; aay++)
            for (int ds = 0; ds < n_dof; ds++) {
                HitRec rec;
                bool want = false;
                if (inside) {
                    rtm::D3 dd = rtm::camera_ray_dir(a.cam_dir, a.cam_right, a.cam_down, a.width, a.height, a.aadepth, x, y, aax);
                    rtm::D3 org = {a.cam_pos[0], a.cam_pos[1], a.cam_pos[2]}, dj = dd;
                    if (dof) {                                                           // main.cpp:792-810
                        rtm::D3 focal = rtm::vadd(org, rtm::vmul(dd, a.focal_len));
                        rtm::D3 up = {0, 1, 0};
                        rtm::D3 cam_up = rtm::vadd(up, rtm::vneg(rtm::vmul(dd, rtm::vdot(up, dd))));
                        rtm::D3 cam_side = rtm::vcross(cam_up, dd);
                        const uint32_t px = (uint32_t)((long long)y * a.width + x), draw = (uint32_t)(((aay * a.aadepth + aax) * n_dof + ds) * 2);
                        double jx = -a.aperture + 2 * a.aperture * rtm::counter_uniform(px, draw, a.seed);
                        double jy = -a.aperture + 2 * a.aperture * rtm::counter_uniform(px, draw + 1u, a.seed);
                        org = rtm::vadd(rtm::vadd(org, rtm::vmul(cam_up, jx)), rtm::vmul(cam_side, jy));
                        dj = rtm::vnorm(rtm::vadd(focal, rtm::vneg(org)));
                    }
                    R o[3] = {(R)org.x, (R)org.y, (R)org.z};
                    R d[3] = {(R)dj.x, (R)dj.y, (R)dj.z};
                    R dist = R(1e9);
                    int id = rtk::trace_ray<R, COUNT, TRI_ONLY, MODE>(sv, o, d, dist, st, &ovf_stack, &wc);
                    if (id >= 0 && (double)dist > a.accuracy && (!a.obj_live || a.obj_live[id])) {   // main.cpp:815-820
                        double t = (double)dist;
                        // the jittered origin, but the un-jittered direction (main.cpp:821-822)
                        rec.pos[0] = org.x + dd.x * t; rec.pos[1] = org.y + dd.y * t; rec.pos[2] = org.z + dd.z * t;
                        rec.dir[0] = dd.x; rec.dir[1] = dd.y; rec.dir[2] = dd.z;
                        rec.weight = dof ? 1.0 / n_dof : (double)a.aadepth;
                        rec.pixel = (uint32_t)((long long)yl * a.width + x);
                        rec.obj = id;
                        want = true;
                    }
                }
                push_hits(hits, &ctr->hits[1], shard_cap, want, rec, &ovf_queue);
            }
    }
    if (ovf_stack) { atomicMax(&ctr->overflow_stack, 1ull); rtk::latch_overflow(ctr, 0); }
    if (ovf_queue) { atomicMax(&ctr->overflow_queue, 1ull); rtk::latch_overflow(ctr, 1); }
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) { wc.inner += __shfl_down(wc.inner, off); wc.prims += __shfl_down(wc.prims, off); }
        if (st.lane == 0) { atomicAdd(&ctr->work_inner, wc.inner); atomicAdd(&ctr->work_prims, wc.prims); }
    }
}

// The ray queue in windows, too: the rays one shading wave spawns in one sample slot (one kind, one tile's
// hit points) are traced by one wave.  The first record carries count - 1 in bits 25..30 of `pixel`.
__device__ __forceinline__ void push_rays(RayRec* q, rtq::Counts* cnt, unsigned long long shard_cap, bool want, RayRec& rec, unsigned long long* overflow)
{
    unsigned long long m = __ballot(want);
    if (m == 0) return;
    const unsigned shard = rtq::wave_index() & (rtq::SHARDS - 1);
    const int leader = __builtin_ctzll(m);
    unsigned long long base = 0;
    if ((int)rtq::lane_id() == leader) {
        base = atomicAdd(&cnt->n[shard][0], (unsigned long long)WAVE);
        rec.pixel |= (uint32_t)(__builtin_popcountll(m) - 1) << 25;
    }
    unsigned lo = __shfl((unsigned)base, leader), hi = __shfl((unsigned)(base >> 32), leader);
    base = ((unsigned long long)hi << 32) | lo;
    if (want) {
        if (base + WAVE <= shard_cap) q[(unsigned long long)shard * shard_cap + base + rtq::prefix_rank(m)] = rec;
        else *overflow = 1;
    }
}

// One recursion level of getColorAt for every queued hit, up to the point where it would call Shoot.
// GLOSSY: some material of the scene takes the 16-sample loops (main.cpp:177-257); without one the loops are a single
// pass and the kernel needs far fewer registers.
template <bool GLOSSY>
__global__ __launch_bounds__(BLOCK) void shade_kernel(ShadeData sd, RenderArgs a, int level, int last_level, const HitRec* __restrict__ in,
                                                      RayRec* __restrict__ rays, rtg::GatherRec* __restrict__ gq, unsigned long long shard_cap,
                                                      unsigned long long gq_cap, RenderCounters* ctr)
{
    const int lane = threadIdx.x & 63;
    unsigned long long ovf_queue = 0;
    // the queue length lives in device memory, so the grid is persistent; chunks of 64 hits are dealt
    // round-robin to the waves (no shared work counter)
    const rtq::Map map = rtq::map_units(&ctr->hits[level], shard_cap, WAVE);
    for (uint32_t chunk = rtq::wave_index(); chunk < map.total; chunk += rtq::n_waves()) {
        unsigned shard;
        uint32_t first, n_shard;
        rtq::locate(map, chunk, WAVE, shard, first, n_shard);
        const uint32_t i = first + lane;
        const uint32_t in_window = in[(unsigned long long)shard * shard_cap + first].pixel >> 25;
        bool active = lane < (int)in_window;
        // what the reflection / refraction loops of main.cpp:181-257 need, per hit
        int n_samp = 0;                      // 0: no rays; 1; GLOSSY_SAMPLE for a glossy material
        bool refr = false, refr_rest = false;
        rtm::D3 hp{0, 0, 0}, Rd{0, 0, 0}, T0{0, 0, 0}, T1{0, 0, 0}, Nr{0, 0, 0}, Nf{0, 0, 0};
        double w_refl = 0, w_refr = 0;
        uint32_t pix = 0, key = 0;
        int32_t from = 0;
        rtg::GatherRec g;
        bool want_g = false;
        if (active) {
            HitRec rec = in[(unsigned long long)shard * shard_cap + i];
            rec.pixel &= HIT_PIXEL_MASK;
            rtm::D3 hit_dir = {rec.dir[0], rec.dir[1], rec.dir[2]};
            rtm::D3 hit_pos = rtm::vadd(rtm::D3{rec.pos[0], rec.pos[1], rec.pos[2]}, rtm::vmul(hit_dir, a.accuracy));   // main.cpp:162
            const rtf::ObjShade& obj = sd.objs[rec.obj];
            double col[4];
            rtm::obj_color(obj, sd.maps, sd.texels, hit_pos, col);                 // main.cpp:164
            rtm::D3 hit_normal = rtm::obj_normal(obj, hit_pos);                    // main.cpp:165
            double special = col[3];
            if (special > 0) {
                if (!last_level) {   // children of the last level return black (main.cpp:161): not traced
                    hp = hit_pos; pix = rec.pixel; from = rec.obj;
                    key = rtm::path_key(hit_pos);
                    n_samp = GLOSSY && rtm::glossy_special(special) ? rtm::GLOSSY_SAMPLE : 1;          // main.cpp:177-179
                    Rd = rtm::reflect_dir(hit_normal, hit_dir);                            // main.cpp:170
                    Nr = hit_normal;
                    w_refl = special > 1 ? rec.weight * (2 - special) : rec.weight;          // main.cpp:200-203
                    if (special > 1) {                                                     // main.cpp:210-258
                        w_refr = rec.weight * (special - 1);
                        double n1n2 = 1.0 / obj.ior;
                        double dot1 = rtm::vdot(hit_normal, rtm::vneg(hit_dir));
                        Nf = hit_normal;
                        if (dot1 < 0) {
                            Nf = rtm::vneg(hit_normal);
                            dot1 = rtm::vdot(Nf, rtm::vneg(hit_dir));
                            n1n2 = 1.0 / n1n2;
                        }
                        double nNL = n1n2 * dot1;
                        double underSQRT = 1 - n1n2 * n1n2 * (1 - dot1 * dot1);
                        if (underSQRT > 0) {
                            double coeffN = nNL - sqrt(underSQRT);
                            T0 = rtm::vadd(rtm::vmul(Nf, coeffN), rtm::vneg(rtm::vmul(rtm::vneg(hit_dir), n1n2)));   // main.cpp:233
                            refr = true;
                        }
                        // The loop of main.cpp:214 flips hit_normal in place, so from its second pass on the normal already
                        // faces the ray and n1n2 stays 1 / ior even for a ray leaving the medium.
                        {
                            double n1n2b = 1.0 / obj.ior;
                            double dot1b = rtm::vdot(Nf, rtm::vneg(hit_dir));
                            double nNLb = n1n2b * dot1b;
                            double under_b = 1 - n1n2b * n1n2b * (1 - dot1b * dot1b);
                            if (under_b > 0) {
                                double coeffN = nNLb - sqrt(under_b);
                                T1 = rtm::vadd(rtm::vmul(Nf, coeffN), rtm::vneg(rtm::vmul(rtm::vneg(hit_dir), n1n2b)));
                                refr_rest = true;
                            }
                        }
                    }
                }
            } else if (a.use_photon_map) {                                         // main.cpp:260-295
                g.pos[0] = hit_pos.x; g.pos[1] = hit_pos.y; g.pos[2] = hit_pos.z;
                g.normal[0] = hit_normal.x; g.normal[1] = hit_normal.y; g.normal[2] = hit_normal.z;
                g.color[0] = col[0]; g.color[1] = col[1]; g.color[2] = col[2];
                g.weight = rec.weight;
                g.pixel = rec.pixel;
                g.flags = 0;
                g.march[0] = g.march[1] = g.march[2] = 0;
                if (special < 0) {                                                 // main.cpp:297-308, 337-339
                    g.flags = rtg::GATHER_SSS;
                    double n1n2 = 1.0 / obj.ior;
                    double dot1 = rtm::vdot(hit_normal, rtm::vneg(hit_dir));
                    rtm::D3 n = hit_normal;      // the estimate above keeps the un-flipped normal
                    if (dot1 < 0) {
                        n = rtm::vneg(n);
                        dot1 = rtm::vdot(n, rtm::vneg(hit_dir));
                        n1n2 = 1.0 / n1n2;
                    }
                    double nNL = n1n2 * dot1;
                    double underSQRT = 1 - n1n2 * n1n2 * (1 - dot1 * dot1);
                    if (underSQRT > 0) {
                        double coeffN = nNL - sqrt(underSQRT);
                        rtm::D3 T = rtm::vadd(rtm::vmul(n, coeffN), rtm::vneg(rtm::vmul(rtm::vneg(hit_dir), n1n2)));
                        g.march[0] = T.x; g.march[1] = T.y; g.march[2] = T.z;
                        g.flags |= rtg::GATHER_MARCH;
                    }
                }
                want_g = true;
            }
        }
        // the sample loops: one reflection ray, and one refraction ray where there is one, per sample
        const int wave_samples = GLOSSY ? __builtin_amdgcn_readfirstlane((int)rtk_wave_max(n_samp)) : (__ballot(n_samp > 0) ? 1 : 0);
        for (int i = 0; i < wave_samples; i++) {
            RayRec next;
            const bool mine = i < n_samp;
            const bool glossy = GLOSSY && n_samp > 1;
            bool want = mine;
            if (mine) {
                rtm::D3 dir = Rd;
                if (glossy) {
                    const uint32_t draw = (uint32_t)(level * 1024 + i * 2);
                    dir = rtm::glossy_ray(Rd, Nr, 0.2, rtm::counter_uniform(key, draw, a.seed), rtm::counter_uniform(key, draw + 1u, a.seed));   // main.cpp:183-184
                }
                next.org[0] = hp.x; next.org[1] = hp.y; next.org[2] = hp.z;
                next.dir[0] = dir.x; next.dir[1] = dir.y; next.dir[2] = dir.z;                                // main.cpp:188
                next.cont[0] = Rd.x; next.cont[1] = Rd.y; next.cont[2] = Rd.z;                                // main.cpp:195-196
                next.weight = w_refl * (1.0 / n_samp);                                                        // main.cpp:207
                next.pixel = pix;
                next.obj = from;
            }
            push_rays(rays, &ctr->rays[level], shard_cap, want, next, &ovf_queue);
            const bool has = i == 0 ? refr : refr_rest;
            want = mine && has;
            if (want) {
                const rtm::D3 T = i == 0 ? T0 : T1;
                rtm::D3 dir = T;
                if (glossy) {
                    const uint32_t draw = (uint32_t)(level * 1024 + 512 + i * 2);
                    dir = rtm::glossy_ray(T, Nf, 0.2, rtm::counter_uniform(key, draw, a.seed), rtm::counter_uniform(key, draw + 1u, a.seed));    // main.cpp:235-236
                }
                // main.cpp:255-256 scale the running sum and add it to the result inside the loop: sample i ends up
                // with the weight sum_{t=1..n-i} n^-t (1 for a single sample)
                double gw = 0, pw = 1;
                for (int t = 0; t < n_samp - i; t++) { pw *= 1.0 / n_samp; gw += pw; }
                next.org[0] = hp.x; next.org[1] = hp.y; next.org[2] = hp.z;
                next.dir[0] = dir.x; next.dir[1] = dir.y; next.dir[2] = dir.z;                                // main.cpp:240
                next.cont[0] = T.x; next.cont[1] = T.y; next.cont[2] = T.z;                                   // main.cpp:246-247
                next.weight = w_refr * gw;                                                                    // main.cpp:251
                next.pixel = pix | RAY_REFRACT;
                next.obj = from;
            }
            push_rays(rays, &ctr->rays[level], shard_cap, want, next, &ovf_queue);
        }
        rtg::push_window(gq, &ctr->gather, gq_cap, want_g, g, &ovf_queue);      // (the gather queue takes requests from every level: a capacity of its own)
    }
    if (ovf_queue) { atomicMax(&ctr->overflow_queue, 1ull); rtk::latch_overflow(ctr, 1); }
}

// Shoot for every queued ray of a level; hits become the next level's queue (main.cpp:190-199, 242-250).
// (The draw loop around the traversal costs this kernel about 45 registers.  It is held to 128 -- four waves per SIMD --
// which the compiler reaches without spilling once the ray record is read in two parts, see below.)
template <class R, bool TRI_ONLY, int MODE, bool COUNT = false>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(MODE == rtk::TRAV_SCENE && sizeof(R) == 8 ? 3 : 4))) void trace_queue_kernel(rtf::SceneView<R> sv, RenderArgs a, int level, const RayRec* __restrict__ rays,
                                                                    HitRec* __restrict__ out, unsigned long long shard_cap, RenderCounters* ctr,
                                                                    int cap, int n_nodes, int n_prims)
{
    extern __shared__ __align__(16) unsigned char lds[];
    const size_t node_bytes = rtk::nodes_in_lds(MODE) ? rtk::node_lds_bytes<R>((size_t)n_nodes) : 0;
    const size_t prim_bytes = MODE == rtk::TRAV_SCENE ? rtk::prim_lds_bytes<R>((size_t)n_prims) : 0;
    if (MODE == rtk::TRAV_SCENE) rtk::stage_prims<R>(reinterpret_cast<uint4*>(lds + node_bytes), sv.prims, n_prims);
    if (rtk::nodes_in_lds(MODE)) rtk::stage_nodes<R>(reinterpret_cast<uint4*>(lds), sv.nodes, n_nodes);
    LaneStack st = rtk::make_stack<R>(lds, cap, node_bytes, prim_bytes);
    uint32_t ovf_stack = 0;
    unsigned long long ovf_queue = 0, n_secondary = 0;
    rtk::WorkCount wc;
    // secondary rays differ a lot in cost: chunks of 64 are drawn dynamically (per-shard draw counters)
    const rtq::Map map = rtq::map_units(&ctr->rays[level], shard_cap, WAVE);
    rtq::Drawer drawer = rtq::drawer_init();
    uint32_t chunk;
    while (rtq::draw(map, &ctr->ray_draws[level], WAVE, drawer, chunk)) {
        unsigned shard;
        uint32_t first, n_shard;
        rtq::locate(map, chunk, WAVE, shard, first, n_shard);
        const uint32_t i = first + st.lane;
        const uint32_t in_window = ((rays[(unsigned long long)shard * shard_cap + first].pixel >> 25) & 63u) + 1u;
        HitRec next;
        bool want = false;
        if (st.lane < (int)in_window) {
            // Only origin and direction are read before the walk; what the hit record needs besides (the direction the path
            // continues along, weight, pixel, object) is read after it, from the same cache lines: twenty registers
            // fewer held across the traversal, which is what decides how many waves a SIMD holds.
            const RayRec* const rp = rays + ((unsigned long long)shard * shard_cap + i);
            R o[3] = {(R)rp->org[0], (R)rp->org[1], (R)rp->org[2]};
            R d[3] = {(R)rp->dir[0], (R)rp->dir[1], (R)rp->dir[2]};
            R dist = R(1e9);
            int io = rtk::trace_ray<R, COUNT, TRI_ONLY, MODE>(sv, o, d, dist, st, &ovf_stack, &wc);
            n_secondary++;
            __asm__ volatile("" ::: "memory");          // (keeps the reads below from being moved up across the walk)
            double t = (double)dist;
            bool go = a.head_quirk ? (t > a.accuracy) : (io >= 0 && t > a.accuracy);   // main.cpp:193,245
            const uint32_t ray_pixel = rp->pixel & (HIT_PIXEL_MASK | RAY_REFRACT);
            const int32_t ray_obj = rp->obj;
            if (go && a.obj_live) go = a.obj_live[a.head_quirk ? ray_obj : io] != 0;
            if (go) {
                rtm::D3 org = {rp->org[0], rp->org[1], rp->org[2]}, dir = {rp->cont[0], rp->cont[1], rp->cont[2]};
                const bool refract = (ray_pixel & RAY_REFRACT) != 0;
                rtm::D3 np = rtm::vadd(org, rtm::vmul(dir, refract ? a.accuracy + t : t));   // main.cpp:195, 246
                next.pos[0] = np.x; next.pos[1] = np.y; next.pos[2] = np.z;
                next.dir[0] = dir.x; next.dir[1] = dir.y; next.dir[2] = dir.z;
                next.weight = rp->weight;
                next.pixel = ray_pixel & ~RAY_REFRACT;
                next.obj = a.head_quirk ? ray_obj : io;                          // main.cpp:198, 249
                want = true;
            }
        }
        push_hits(out, &ctr->hits[level + 1], shard_cap, want, next, &ovf_queue, chunk);
    }
    if (ovf_stack) { atomicMax(&ctr->overflow_stack, 1ull); rtk::latch_overflow(ctr, 0); }
    if (ovf_queue) { atomicMax(&ctr->overflow_queue, 1ull); rtk::latch_overflow(ctr, 1); }
    for (int off = 32; off > 0; off >>= 1) n_secondary += __shfl_down(n_secondary, off);
    if (st.lane == 0 && n_secondary) atomicAdd(&ctr->secondary_rays, n_secondary);
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) { wc.inner += __shfl_down(wc.inner, off); wc.prims += __shfl_down(wc.prims, off); }
        if (st.lane == 0) { atomicAdd(&ctr->work_inner, wc.inner); atomicAdd(&ctr->work_prims, wc.prims); }
    }
}

// Pixel average (main.cpp:853-859), clip (Color.h:39-57; a no-op after per-sample clipping) and the
// quantiser of saveBmp: (unsigned char) floor(v * 255), written B, G, R (IO.cpp:58-62).
__global__ void resolve_kernel(long long n_px, int aadepth, double* __restrict__ acc, double* __restrict__ rgb, uint8_t* __restrict__ bgr)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_px) return;
    double v[3];
    for (int c = 0; c < 3; c++) {
        double x = acc[3 * i + c];
        if (aadepth > 1) x = x / (double)(aadepth * aadepth);
        if (x > 1) x = 1;
        if (x < 0) x = 0;
        v[c] = x;
    }
    if (rgb) { rgb[3 * i] = v[0]; rgb[3 * i + 1] = v[1]; rgb[3 * i + 2] = v[2]; }
    if (bgr) {
        bgr[3 * i] = (uint8_t)floor(v[2] * 255);
        bgr[3 * i + 1] = (uint8_t)floor(v[1] * 255);
        bgr[3 * i + 2] = (uint8_t)floor(v[0] * 255);
    }
}

}  // namespace rtr
