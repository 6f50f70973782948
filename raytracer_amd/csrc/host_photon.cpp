// Host photon emission and photon-map flattening.  See host_photon.h.
#include "host_photon.h"

#include <math.h>

#include <algorithm>
#include <array>
#include <thread>

#include "rt_math.h"

namespace rth {

using rtm::D3;

// ---------------------------------------------------------------- host closest hit
namespace {

int shoot_node(const Scene& s, uint32_t ref, const double* lo, const double* hi, const double* o, const double* d, double& dist)
{
    if (rtm::slab_literal<double>(lo, hi, o, d, dist) < 0) return -1;          // BVH.cpp:132
    const FlatBVH& b = s.bvh;
    if (ref & rtf::LEAF_BIT) {
        int result = -1;
        uint32_t first = ref & rtf::LEAF_FIRST_MASK, cnt = (ref >> rtf::LEAF_COUNT_SHIFT) & 15u;
        for (uint32_t k = 0; k < cnt; k++) {                                    // BVH.cpp:135-149
            const rtf::Prim<double>& p = b.prims[first + k];
            double t = rtm::prim_intersect<double>(p, o, d);
            if (t >= 0 && t < dist) { result = (int)p.id; dist = t; }
        }
        return result;
    }
    const rtf::Node<double>& nd = b.nodes[ref];
    double td0 = rtm::slab_literal<double>(nd.lo0, nd.hi0, o, d, dist);          // BVH.cpp:155-156
    double td1 = rtm::slab_literal<double>(nd.lo1, nd.hi1, o, d, dist);
    int result = -1, rc;
    if (td0 < td1) {                                                             // BVH.cpp:159-176
        if ((rc = shoot_node(s, nd.c0, nd.lo0, nd.hi0, o, d, dist)) >= 0) result = rc;
        if ((rc = shoot_node(s, nd.c1, nd.lo1, nd.hi1, o, d, dist)) >= 0) result = rc;
    } else {
        if ((rc = shoot_node(s, nd.c1, nd.lo1, nd.hi1, o, d, dist)) >= 0) result = rc;
        if ((rc = shoot_node(s, nd.c0, nd.lo0, nd.hi0, o, d, dist)) >= 0) result = rc;
    }
    return result;
}

}  // namespace

int host_shoot(const Scene& s, const double* o, const double* d, double& dist)
{
    if (d[0] != d[0]) return -1;                                                 // BVH.cpp:110-115
    const FlatBVH& b = s.bvh;
    int result = -1;
    if (b.root_ref != rtf::NO_NODE && !(rtm::slab_literal<double>(b.root_box.lo, b.root_box.hi, o, d, dist) < 0))
        result = shoot_node(s, b.root_ref, b.root_box.lo, b.root_box.hi, o, d, dist);
    for (const auto& p : b.planes) {
        double t = rtm::plane_intersect<double>(p.a, p.e1[0], o, d);
        if (t >= 0 && t < dist) { result = (int)p.id; dist = t; }
    }
    return result;
}

// ---------------------------------------------------------------- emission
namespace {

using rtm::fresnel;

const double PI_REF = 3.1415926;   // main.cpp:24

struct Emitter {
    const Scene& s;
    const EmitParams& ep;
    std::mt19937 gen;
    std::vector<PhotonRec>& surface;
    std::vector<PhotonRec>& volume;
    std::uniform_real_distribution<> unit{0, 1};       // dis_rand, main.cpp:48
    std::uniform_real_distribution<> percent{0, 100};  // dis_rand_100, main.cpp:49
    std::vector<rtf::TexDesc> maps;
    std::vector<double> texels;

    void color_of(const rtf::ObjShade& o, D3 p, double* rgbs) { rtm::obj_color(o, maps.data(), texels.data(), p, rgbs); }

    static void min3(std::array<double, 3>& light, const double* c) { for (int i = 0; i < 3; i++) light[i] = std::min(c[i], light[i]); }

    // photonEmission, main.cpp:396-598
    // `light` is by value: each recursion level owns its copy, as the reference passes Color by value
    void walk(D3 org, D3 dir, std::array<double, 3> light, int bounce)
    {
        if (bounce > ep.bounce) return;
        double dist = 1e9;
        double o[3] = {org.x, org.y, org.z}, d[3] = {dir.x, dir.y, dir.z};
        int io = host_shoot(s, o, d, dist);
        if (io < 0 || !(dist > ep.accuracy)) return;
        const rtf::ObjShade& obj = s.objects[io].sh;
        D3 ipos = rtm::vadd(org, rtm::vmul(dir, ep.accuracy + dist));
        D3 normal = rtm::obj_normal(obj, ipos);
        double col[4];
        color_of(obj, ipos, col);
        double special = col[3];
        float reflectance = 1;
        int mask[3] = {1, 1, 1};
        bool transmit = true;
        if (special <= 0) {
            // storePhoton (main.cpp:69-96): the photon carries the incoming power; the walk goes on
            // with min(surface colour, power) * 0.8
            PhotonRec ph;
            ph.pos[0] = ipos.x; ph.pos[1] = ipos.y; ph.pos[2] = ipos.z;
            for (int i = 0; i < 3; i++) ph.power[i] = light[i];
            ph.dir[0] = dir.x; ph.dir[1] = dir.y; ph.dir[2] = dir.z;
            surface.push_back(ph);
            for (int i = 0; i < 3; i++) {
                double v = std::min(col[i], light[i]);
                v = v > 1 ? 1 : v;
                v = v < 0 ? 0 : v;
                light[i] = v * 0.8;
            }
        } else if (special < 1 && special > 0) {
            float cosTheta = (float)rtm::vdot(normal, rtm::vneg(dir));
            float n2 = obj.ior;
            reflectance = (float)fresnel(cosTheta, 1, n2, 0);
            int roll = (int)percent(gen);
            if (roll > reflectance * 100) transmit = false;
        } else if (special < 2 && special > 0) {
            for (int i = 0; i < 3; i++) {
                float cosTheta = (float)rtm::vdot(normal, rtm::vneg(dir));
                float n2 = (float)(obj.ior - 0.01 + i / 100.0);
                if (cosTheta > 0) {
                    reflectance = (float)fresnel(cosTheta, 1, n2, 1);
                } else {
                    cosTheta = (float)rtm::vdot(rtm::vneg(normal), rtm::vneg(dir));
                    reflectance = (float)fresnel(cosTheta, n2, 1, 1);
                }
                int roll = (int)percent(gen);
                if (roll > reflectance * 100) mask[i] = 0;
            }
        }
        if (!transmit) return;
        if (special == 0) {                                   // Lambert bounce, main.cpp:464-484
            double z1 = unit(gen), z2 = unit(gen);
            double theta = acos(sqrt(z1)), phi = 2 * PI_REF * z2;
            double x = sin(theta) * cos(phi), z = sin(theta) * sin(phi), y = cos(theta);
            D3 u = rtm::obj_tangent(obj, ipos);
            D3 v = rtm::vcross(normal, u);
            D3 w = rtm::vnorm(rtm::vadd(rtm::vmul(u, x), rtm::vmul(rtm::vadd(rtm::vmul(v, z), normal), y)));
            walk(ipos, w, light, bounce + 1);
        } else if (special <= 1 && special > 0) {             // mirror, main.cpp:485-503
            double dot1 = rtm::vdot(normal, rtm::vneg(dir));
            D3 r = rtm::vnorm(rtm::vadd(rtm::vneg(dir), rtm::vmul(rtm::vadd(rtm::vmul(normal, dot1), dir), 2.0)));
            min3(light, col);
            walk(ipos, r, light, bounce + 1);
        } else if (special <= 2 && special > 1) {             // refraction, main.cpp:504-557 (only i = 0 runs)
            if (mask[0]) {
                double n1n2 = 1.0 / (obj.ior - 0.01 + 0 / 100.0);
                double dot1 = rtm::vdot(normal, rtm::vneg(dir));
                if (dot1 < 0) { normal = rtm::vneg(normal); dot1 = rtm::vdot(normal, rtm::vneg(dir)); n1n2 = 1.0 / n1n2; }
                double nNL = n1n2 * dot1;
                double under = 1 - n1n2 * n1n2 * (1 - dot1 * dot1);
                if (under > 0) {
                    double coeffN = nNL - sqrt(under);
                    D3 t = rtm::vadd(rtm::vmul(normal, coeffN), rtm::vneg(rtm::vmul(rtm::vneg(dir), n1n2)));
                    if (!(light[0] == 0 && light[1] == 0 && light[2] == 0)) {
                        std::array<double, 3> next = light;
                        min3(next, col);
                        walk(ipos, t, next, bounce + 1);
                    }
                }
            }
        }
        if (special < 0) {                                    // in-volume scatter, main.cpp:559-595
            double step = 0.005;
            D3 pos = rtm::vadd(ipos, rtm::vmul(dir, step));
            std::uniform_real_distribution<> phi_rand(-PI_REF, PI_REF);
            for (int k = 0; k < 4; k++) {
                double phi = phi_rand(gen);
                double g = 0.3;
                double q = ((1 - g * g) / (1 - g + 2 * g * unit(gen)));
                double newCos = 1 / fabs(2 * g) * (1 + g * g - q * q);
                double newTheta = acos(newCos);
                D3 sd = rtm::vnorm(D3{sin(newTheta) * cos(phi), sin(newTheta) * sin(phi), cos(newTheta)});
                double range = step;
                double po[3] = {pos.x, pos.y, pos.z}, pd[3] = {sd.x, sd.y, sd.z};
                if (host_shoot(s, po, pd, range) >= 0) break;
                pos = rtm::vadd(pos, rtm::vmul(sd, step));
                PhotonRec ph;
                ph.pos[0] = pos.x; ph.pos[1] = pos.y; ph.pos[2] = pos.z;
                for (int i = 0; i < 3; i++) ph.power[i] = light[i];
                ph.dir[0] = sd.x; ph.dir[1] = sd.y; ph.dir[2] = sd.z;
                volume.push_back(ph);
                for (int i = 0; i < 3; i++) light[i] *= 0.6;
            }
        }
    }
};

}  // namespace

void emit_photons(const Scene& s, int64_t n_emit, const double* light_pos, uint32_t seed, const EmitParams& ep,
                  std::vector<PhotonRec>& surface, std::vector<PhotonRec>& volume)
{
    Emitter e{s, ep, std::mt19937(seed), surface, volume};
    for (const auto& t : s.textures) {
        e.maps.push_back({t.x_res, t.y_res, (int64_t)e.texels.size()});
        e.texels.insert(e.texels.end(), t.rgb.begin(), t.rgb.end());
    }
    std::uniform_real_distribution<> dis(-1, 1);          // main.cpp:683
    D3 lp = {light_pos[0], light_pos[1], light_pos[2]};
    for (int64_t pn = 0; pn < n_emit; pn++) {
        double x, y, z;
        do { x = dis(e.gen); y = dis(e.gen); z = dis(e.gen); } while (x * x + y * y + z * z > 1);   // main.cpp:702-706
        D3 d = rtm::vnorm(D3{x, y, z});
        std::array<double, 3> light = {1.0, 1.0, 1.0};      // white_light, main.cpp:637,685
        e.walk(lp, d, light, 0);
    }
}

// ---------------------------------------------------------------- photon map flattening
namespace {

struct KdBuild {
    const double* ph;          // 9 doubles per photon
    std::vector<int32_t>& idx;
    std::vector<uint8_t>& axis;

    double coord(int32_t i, int ax) const { return ph[(size_t)i * 9 + ax]; }

    void build(int64_t lo, int64_t hi, int par_levels)
    {
        while (hi - lo > FlatPhotonMap::LEAF) {
            // widest axis of this range, first maximum wins (Kdtree.cpp:14-22)
            double span[3];
            for (int ax = 0; ax < 3; ax++) {
                double mn = coord(idx[lo], ax), mx = mn;
                for (int64_t i = lo + 1; i < hi; i++) { double v = coord(idx[i], ax); mn = std::min(mn, v); mx = std::max(mx, v); }
                span[ax] = mx - mn;
            }
            int ax = int(std::max_element(span, span + 3) - span);
            // the reference sorts the whole range and takes element n/2 as the pivot (Kdtree.cpp:24-35);
            // only the partition {smaller | pivot | larger} shapes the tree, so selection is enough
            int64_t mid = lo + (hi - lo) / 2;
            std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi,
                             [&](int32_t a, int32_t b) { return coord(a, ax) < coord(b, ax); });
            axis[mid] = (uint8_t)ax;
            if (par_levels > 0) {
                std::thread t([=]() { build(lo, mid, par_levels - 1); });
                build(mid + 1, hi, par_levels - 1);
                t.join();
                return;
            }
            build(lo, mid, 0);
            lo = mid + 1;
        }
    }
};

}  // namespace

void build_photon_map(const double* photons9, int64_t n, FlatPhotonMap& out, int threads)
{
    out = FlatPhotonMap{};
    out.n = n;
    out.order.resize(n);
    out.axis.assign(n, 0);
    for (int64_t i = 0; i < n; i++) out.order[i] = (int32_t)i;
    int levels = 0;
    while ((1 << levels) < threads) levels++;
    KdBuild kb{photons9, out.order, out.axis};
    if (n > 0) kb.build(0, n, levels);
    out.pos.resize((size_t)n * 4);
    out.aux.resize((size_t)n * 6);
    for (int64_t i = 0; i < n; i++) {
        const double* p = photons9 + (size_t)out.order[i] * 9;
        for (int c = 0; c < 3; c++) out.pos[(size_t)i * 4 + c] = (float)p[c];
        out.pos[(size_t)i * 4 + 3] = (float)out.axis[i];
        for (int c = 0; c < 6; c++) out.aux[(size_t)i * 6 + c] = (float)p[3 + c];
    }
    // bounding boxes of the leaves, found by the split rule the device walk uses
    out.leaf_box.assign((size_t)(n / 32 + 1) * 6, 0.f);
    std::vector<std::pair<int64_t, int64_t>> todo;
    if (n > 0) todo.push_back({0, n});
    while (!todo.empty()) {
        auto [lo, hi] = todo.back();
        todo.pop_back();
        if (hi - lo > FlatPhotonMap::LEAF) {
            int64_t mid = lo + (hi - lo) / 2;
            todo.push_back({lo, mid});
            todo.push_back({mid + 1, hi});
            continue;
        }
        float* b = &out.leaf_box[(size_t)(lo / 32) * 6];
        for (int c = 0; c < 3; c++) b[c] = b[3 + c] = out.pos[(size_t)lo * 4 + c];
        for (int64_t i = lo + 1; i < hi; i++)
            for (int c = 0; c < 3; c++) {
                float v = out.pos[(size_t)i * 4 + c];
                b[c] = std::min(b[c], v);
                b[3 + c] = std::max(b[3 + c], v);
            }
    }
}

}  // namespace rth
