// Photon emission on the device: photonEmission (main.cpp:396-598) and the emission loop of
// main.cpp:697-718, one photon per lane.
//
// The reference's emission is order dependent: every photon draws from one shared std::mt19937, so
// only a serial walk reproduces its photon list (the host routine rth::emit_photons does, bit for
// bit).  Here every photon owns a counter-based random stream keyed by (seed, photon number, draw
// number); the walk itself -- Shoot, material branch, Russian roulette, Lambert / mirror /
// refraction continuation, in-volume scatter -- is the reference's, in its arithmetic.  The result is
// the same distribution, not the same list, and is validated statistically.  Photons are appended
// to sharded queues with a sort key (photon number, bounce, slot), so the host can put them in a
// launch-independent order.
#pragma once
#include <hip/hip_runtime.h>

#include "rt_kernels.hip.h"
#include "rt_queue.hip.h"

namespace rte {

using rtk::LaneStack;
using rtk::WAVE;
using rtm::D3;

struct PhotonOut {
    double pos[3], power[3], dir[3];
    unsigned long long key;     // photon number * 64 + bounce * 8 + slot (0: surface store, 1..4: volume scatter step)
};

struct EmitArgs {
    double light[3];
    long long n_emit;
    long long first;            // number of the first photon of this launch
    uint32_t seed;
    int bounce;                 // BOUNCE, main.cpp:25
    double accuracy;
};

constexpr int MAX_EMIT_LEVELS = 32;
struct EmitCounters {
    rtq::Counts surface, volume;
    unsigned long long overflow_stack, overflow_queue, rays, pad[5];
    unsigned long long walking[MAX_EMIT_LEVELS + 1];    // walking[b]: photons that go into bounce b (b >= 1; bounce 0 takes all)
};

// A photon between two bounces of its walk.
struct PhotonState {
    double org[3], dir[3], light[3];
    unsigned long long pn;       // photon number: keys its random stream and its records
    uint32_t draws;              // draws taken from the stream so far
    uint32_t pad;
};

struct ShadeTables {
    const rtf::ObjShade* objs;
    const rtf::TexDesc* maps;
    const double* texels;
};

// one random stream per photon: draw number `ctr` of photon `pn`
struct LaneRng {
    uint32_t lo, hi, ctr;
    __device__ double unit()    // uniform in (0, 1)
    {
        uint32_t x = lo * 0x9E3779B1u ^ (hi + 0x7F4A7C15u) * 0x85EBCA77u ^ (ctr++ + 0x165667B1u) * 0xC2B2AE3Du;
        x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
        uint32_t y = (x ^ lo) * 0x2C1B3C6Du + hi;
        y ^= y >> 15; y *= 0x297A2D39u; y ^= y >> 13;
        // 32 + 20 bits
        return ((double)x * 1048576.0 + (double)(y >> 12) + 0.5) * (1.0 / 4503599627370496.0);
    }
};

// The direction a photon leaves the light with (main.cpp:702-708: uniform in the unit ball by rejection, then normalised);
// takes the first draws of the photon's stream.
__device__ __forceinline__ D3 start_direction(LaneRng& rng)
{
    double x, y, z;
    do { x = 2 * rng.unit() - 1; y = 2 * rng.unit() - 1; z = 2 * rng.unit() - 1; } while (x * x + y * y + z * z > 1);
    return rtm::vnorm(D3{x, y, z});
}

// Which photons share a wave at the first bounce: the photons of a launch ordered by the direction they start with (a
// key per photon: cube face, then a 6 + 6 bit Morton code of the point on the face), so that the 64 rays of a wave leave
// the light side by side like the camera rays of a tile instead of in 64 unrelated directions.  What a photon does does
// not depend on its lane, and the photon list is put into (photon, bounce) order afterwards: the list is the same.
__global__ void emit_order_keys_kernel(EmitArgs a, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_emit) return;
    const unsigned long long pn = (unsigned long long)(a.first + i);
    LaneRng rng{(uint32_t)pn ^ a.seed * 0x9E3779B9u, (uint32_t)(pn >> 32) + a.seed, 0u};
    const D3 d = start_direction(rng);
    const double ax = fabs(d.x), ay = fabs(d.y), az = fabs(d.z);
    uint32_t face;
    double u, v, m;
    if (ax >= ay && ax >= az) { face = d.x < 0 ? 1u : 0u; m = ax; u = d.y; v = d.z; }
    else if (ay >= az) { face = d.y < 0 ? 3u : 2u; m = ay; u = d.x; v = d.z; }
    else { face = d.z < 0 ? 5u : 4u; m = az; u = d.x; v = d.y; }
    const uint32_t qu = (uint32_t)fmin(63.0, (u / m * 0.5 + 0.5) * 64.0), qv = (uint32_t)fmin(63.0, (v / m * 0.5 + 0.5) * 64.0);
    uint32_t mort = 0;
    for (int b = 0; b < 6; b++) mort |= ((qu >> b) & 1u) << (2 * b) | ((qv >> b) & 1u) << (2 * b + 1);
    keys[i] = face << 12 | mort;
    vals[i] = (uint32_t)i;
}
constexpr int EMIT_ORDER_KEY_BITS = 15;

// Which photons share a wave at a later bounce (optional): the survivors of a bounce keyed by the direction they go on in
// (cube face, 4 x 4 cells on it: 7 bits) and, within a direction, by where they start (5 + 5 + 5 bit Morton code in the
// scene's box), so that a wave's rays run roughly side by side.  Valid keys stay below 2^22; the host pads the key array
// with a larger value, and a stable sort leaves the survivors in front.
constexpr int SURVIVOR_KEY_BITS = 23;
template <class R>
__device__ __forceinline__ uint32_t survivor_key(const R* lo, const R* hi, const D3& org, const D3& d)
{
    const double ax = fabs(d.x), ay = fabs(d.y), az = fabs(d.z);
    uint32_t face;
    double u, v, m;
    if (ax >= ay && ax >= az) { face = d.x < 0 ? 1u : 0u; m = ax; u = d.y; v = d.z; }
    else if (ay >= az) { face = d.y < 0 ? 3u : 2u; m = ay; u = d.x; v = d.z; }
    else { face = d.z < 0 ? 5u : 4u; m = az; u = d.x; v = d.y; }
    const uint32_t qu = (uint32_t)fmin(3.0, (u / m * 0.5 + 0.5) * 4.0), qv = (uint32_t)fmin(3.0, (v / m * 0.5 + 0.5) * 4.0);
    const uint32_t dir_bin = face << 4 | qv << 2 | qu;                                       // < 96
    const double p[3] = {org.x, org.y, org.z};
    uint32_t q[3];
    for (int i = 0; i < 3; i++) {
        const double e = (double)hi[i] - (double)lo[i];
        const double t = e > 0 ? (p[i] - (double)lo[i]) / e : 0.0;
        q[i] = (uint32_t)fmin(31.0, fmax(0.0, t * 32.0));
    }
    uint32_t mort = 0;
    for (int b = 0; b < 5; b++) mort |= ((q[0] >> b) & 1u) << (3 * b) | ((q[1] >> b) & 1u) << (3 * b + 1) | ((q[2] >> b) & 1u) << (3 * b + 2);
    return dir_bin << 15 | mort;
}

// One bounce of every photon still walking (main.cpp:396-600), one photon per lane.  Bounce 0 starts the photons at the
// light; a later bounce takes the survivors of the one before, compacted: walks end at very different lengths, and a wave
// that carried its photons from the light to their ends ran with a sixth of its lanes on average.
template <class R, bool TRI_ONLY, int MODE>
__global__ __launch_bounds__(rtk::MAX_BLOCK) void emit_kernel(rtf::SceneView<R> sv, ShadeTables sd, EmitArgs a, int bounce, const PhotonState* __restrict__ from,
                                                             PhotonState* __restrict__ onward, PhotonOut* __restrict__ surf, unsigned long long cap_s,
                                                             PhotonOut* __restrict__ vol, unsigned long long cap_v, EmitCounters* ctr, int cap, int n_nodes,
                                                             const uint32_t* __restrict__ order, const uint32_t* __restrict__ from_order,
                                                             uint32_t* __restrict__ next_keys, uint32_t* __restrict__ next_vals)
{
    extern __shared__ __align__(16) unsigned char lds[];
    const size_t node_bytes = MODE == rtk::TRAV_NODES ? rtk::node_lds_bytes<R>((size_t)n_nodes) : 0;
    if (MODE == rtk::TRAV_NODES) rtk::stage_nodes<R>(reinterpret_cast<uint4*>(lds), sv.nodes, n_nodes);
    LaneStack st = rtk::make_stack<R>(lds, cap, node_bytes);
    uint32_t ovf_stack = 0;
    unsigned long long ovf_queue = 0, n_rays = 0;
    const double PI_REF = 3.1415926;   // main.cpp:24
    const long long local = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n_here = bounce == 0 ? a.n_emit : (long long)ctr->walking[bounce];
    bool alive = local < n_here;
    unsigned long long pn = 0;
    D3 org = {a.light[0], a.light[1], a.light[2]};
    D3 dir = {1, 0, 0};
    double light[3] = {1.0, 1.0, 1.0};   // white_light, main.cpp:637,685
    uint32_t draws = 0;
    if (alive && bounce > 0) {
        const PhotonState& ps = from[from_order ? (long long)from_order[local] : local];      // (see survivor_key)
        org = D3{ps.org[0], ps.org[1], ps.org[2]}; dir = D3{ps.dir[0], ps.dir[1], ps.dir[2]};
        for (int i = 0; i < 3; i++) light[i] = ps.light[i];
        pn = ps.pn; draws = ps.draws;
    } else if (alive) {
        pn = (unsigned long long)(a.first + (order ? (long long)order[local] : local));      // (see emit_order_keys_kernel)
    }
    LaneRng rng{(uint32_t)pn ^ a.seed * 0x9E3779B9u, (uint32_t)(pn >> 32) + a.seed, draws};
    if (alive && bounce == 0) dir = start_direction(rng);

    {      // (main.cpp:398: depth > BOUNCE ends the walk -- the host launches bounces 0 .. BOUNCE)
        int io = -1;
        double dist = 1e9;
        if (alive) {
            R o[3] = {(R)org.x, (R)org.y, (R)org.z}, d[3] = {(R)dir.x, (R)dir.y, (R)dir.z};
            R dd = R(1e9);
            io = rtk::trace_ray<R, false, TRI_ONLY, MODE>(sv, o, d, dd, st, &ovf_stack);      // main.cpp:402
            dist = (double)dd;
            n_rays++;
            if (io < 0 || !(dist > a.accuracy)) alive = false;        // main.cpp:407
        }
        PhotonOut rec;
        bool store = false;
        bool scatter = false;
        D3 ipos = {0, 0, 0};
        if (alive) {
            const rtf::ObjShade& obj = sd.objs[io];
            ipos = rtm::vadd(org, rtm::vmul(dir, a.accuracy + dist));   // main.cpp:409
            D3 normal = rtm::obj_normal(obj, ipos);
            double col[4];
            rtm::obj_color(obj, sd.maps, sd.texels, ipos, col);
            const double special = col[3];
            float reflectance = 1;
            bool mask0 = true;
            bool transmit = true;
            if (special <= 0) {
                // storePhoton (main.cpp:69-96): the photon carries the incoming power; the walk goes on
                // with min(surface colour, power) * 0.8
                rec.pos[0] = ipos.x; rec.pos[1] = ipos.y; rec.pos[2] = ipos.z;
                for (int i = 0; i < 3; i++) rec.power[i] = light[i];
                rec.dir[0] = dir.x; rec.dir[1] = dir.y; rec.dir[2] = dir.z;
                rec.key = pn * 64ull + (unsigned long long)bounce * 8ull;
                store = true;
                for (int i = 0; i < 3; i++) {
                    double v = col[i] < light[i] ? col[i] : light[i];
                    v = v > 1 ? 1 : v;
                    v = v < 0 ? 0 : v;
                    light[i] = v * 0.8;
                }
            } else if (special < 1 && special > 0) {              // mirror: Fresnel roulette, main.cpp:424-433
                float cosTheta = (float)rtm::vdot(normal, rtm::vneg(dir));
                float n2 = obj.ior;
                reflectance = (float)rtm::fresnel(cosTheta, 1, n2, 0);
                int roll = (int)(rng.unit() * 100);
                if (roll > reflectance * 100) transmit = false;
            } else if (special < 2 && special > 0) {              // glass: one roulette per colour band, main.cpp:434-452
                for (int i = 0; i < 3; i++) {
                    float cosTheta = (float)rtm::vdot(normal, rtm::vneg(dir));
                    float n2 = (float)(obj.ior - 0.01 + i / 100.0);
                    if (cosTheta > 0) {
                        reflectance = (float)rtm::fresnel(cosTheta, 1, n2, 1);
                    } else {
                        cosTheta = (float)rtm::vdot(rtm::vneg(normal), rtm::vneg(dir));
                        reflectance = (float)rtm::fresnel(cosTheta, n2, 1, 1);
                    }
                    int roll = (int)(rng.unit() * 100);
                    if (roll > reflectance * 100 && i == 0) mask0 = false;
                }
            }
            bool go_on = false;
            D3 ndir = dir;
            if (transmit) {
                if (special == 0) {                                   // Lambert bounce, main.cpp:464-484
                    double z1 = rng.unit(), z2 = rng.unit();
                    double theta = acos(sqrt(z1)), phi = 2 * PI_REF * z2;
                    double x = sin(theta) * cos(phi), z = sin(theta) * sin(phi), y = cos(theta);
                    D3 u = rtm::obj_tangent(obj, ipos);
                    D3 v = rtm::vcross(normal, u);
                    ndir = rtm::vnorm(rtm::vadd(rtm::vmul(u, x), rtm::vmul(rtm::vadd(rtm::vmul(v, z), normal), y)));
                    go_on = true;
                } else if (special <= 1 && special > 0) {             // mirror, main.cpp:485-503
                    double dot1 = rtm::vdot(normal, rtm::vneg(dir));
                    ndir = rtm::vnorm(rtm::vadd(rtm::vneg(dir), rtm::vmul(rtm::vadd(rtm::vmul(normal, dot1), dir), 2.0)));
                    for (int i = 0; i < 3; i++) light[i] = col[i] < light[i] ? col[i] : light[i];
                    go_on = true;
                } else if (special <= 2 && special > 1) {             // refraction, main.cpp:504-557 (only band 0 continues)
                    if (mask0) {
                        double n1n2 = 1.0 / (obj.ior - 0.01 + 0 / 100.0);
                        double dot1 = rtm::vdot(normal, rtm::vneg(dir));
                        if (dot1 < 0) { normal = rtm::vneg(normal); dot1 = rtm::vdot(normal, rtm::vneg(dir)); n1n2 = 1.0 / n1n2; }
                        double nNL = n1n2 * dot1;
                        double under = 1 - n1n2 * n1n2 * (1 - dot1 * dot1);
                        if (under > 0) {
                            double coeffN = nNL - sqrt(under);
                            ndir = rtm::vadd(rtm::vmul(normal, coeffN), rtm::vneg(rtm::vmul(rtm::vneg(dir), n1n2)));
                            if (!(light[0] == 0 && light[1] == 0 && light[2] == 0)) {
                                for (int i = 0; i < 3; i++) light[i] = col[i] < light[i] ? col[i] : light[i];
                                go_on = true;
                            }
                        }
                    }
                }
                scatter = special < 0;                                // in-volume scatter, main.cpp:559-595
            }
            if (!go_on) alive = false;
            else { org = ipos; }
            if (scatter) ndir = dir;      // the scatter below needs the incoming direction; a subsurface hit does not continue
            dir = go_on ? ndir : dir;
        }
        rtq::push(surf, &ctr->surface, cap_s, store, rec, &ovf_queue);
        if (__any(scatter)) {
            const double step = 0.005;
            D3 pos = rtm::vadd(ipos, rtm::vmul(dir, step));
            double vlight[3] = {light[0], light[1], light[2]};
            for (int k = 0; k < 4; k++) {
                if (!__any(scatter)) break;
                PhotonOut vrec;
                bool vstore = false;
                if (scatter) {
                    double phi = (2 * rng.unit() - 1) * PI_REF;       // uniform in (-PI, PI), main.cpp:566
                    double g = 0.3;
                    double q = ((1 - g * g) / (1 - g + 2 * g * rng.unit()));
                    double newCos = 1 / fabs(2 * g) * (1 + g * g - q * q);
                    double newTheta = acos(newCos);
                    D3 sdir = rtm::vnorm(D3{sin(newTheta) * cos(phi), sin(newTheta) * sin(phi), cos(newTheta)});
                    R po[3] = {(R)pos.x, (R)pos.y, (R)pos.z}, pd[3] = {(R)sdir.x, (R)sdir.y, (R)sdir.z};
                    R range = (R)step;
                    n_rays++;
                    if (rtk::trace_ray<R, false, TRI_ONLY, MODE>(sv, po, pd, range, st, &ovf_stack) >= 0) {   // main.cpp:584-586: left the volume
                        scatter = false;
                    } else {
                        pos = rtm::vadd(pos, rtm::vmul(sdir, step));
                        vrec.pos[0] = pos.x; vrec.pos[1] = pos.y; vrec.pos[2] = pos.z;
                        for (int i = 0; i < 3; i++) vrec.power[i] = vlight[i];
                        vrec.dir[0] = sdir.x; vrec.dir[1] = sdir.y; vrec.dir[2] = sdir.z;
                        vrec.key = pn * 64ull + (unsigned long long)bounce * 8ull + 1ull + (unsigned long long)k;
                        vstore = true;
                        for (int i = 0; i < 3; i++) vlight[i] *= 0.6;
                    }
                }
                rtq::push(vol, &ctr->volume, cap_v, vstore, vrec, &ovf_queue);
            }
        }
    }
    // the photons that walk on: appended to the next bounce's list, one counter update per wave
    if (bounce < a.bounce) {
        const unsigned long long m = __ballot(alive);
        if (m) {
            const int leader = __builtin_ctzll(m);
            unsigned long long base = 0;
            if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(&ctr->walking[bounce + 1], (unsigned long long)__builtin_popcountll(m));
            base = ((unsigned long long)(unsigned)__shfl((unsigned)(base >> 32), leader) << 32) | (unsigned)__shfl((unsigned)base, leader);
            if (alive) {
                PhotonState ps;
                ps.org[0] = org.x; ps.org[1] = org.y; ps.org[2] = org.z;
                ps.dir[0] = dir.x; ps.dir[1] = dir.y; ps.dir[2] = dir.z;
                for (int i = 0; i < 3; i++) ps.light[i] = light[i];
                ps.pn = pn; ps.draws = rng.ctr; ps.pad = 0;
                const unsigned long long at = base + rtq::prefix_rank(m);
                onward[at] = ps;
                if (next_keys) { next_keys[at] = survivor_key(sv.root_lo, sv.root_hi, org, dir); next_vals[at] = (uint32_t)at; }
            }
        }
    }
    if (ovf_stack) { atomicMax(&ctr->overflow_stack, 1ull); rtk::latch_overflow(ctr, 0); }
    if (ovf_queue) { atomicMax(&ctr->overflow_queue, 1ull); rtk::latch_overflow(ctr, 1); }
    for (int off = 32; off > 0; off >>= 1) n_rays += __shfl_down(n_rays, off);
    if ((threadIdx.x & 63) == 0 && n_rays) atomicAdd(&ctr->rays, n_rays);
}

}  // namespace rte
