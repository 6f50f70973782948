// TEST INFRASTRUCTURE ONLY -- the product library never includes, links or loads this file.
//
// CPU restatement, in double precision, of the reference renderer's per-pixel hot path.  Written
// from the reference's behaviour (index-based, no virtual dispatch, no pointer trees), keeping the
// exact floating-point operation order so results are bit-identical when both sides are compiled
// with -ffp-contract=off.  Citations are file:line of the reference.
//
// Parity status: PINNED against the compiled reference (oracle/_ref) by oracle/make_golden.py and
// against the committed vectors in tests/golden/ by tests/test_oracle_golden.py.
#include "rt_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <queue>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace {

struct V3 {
    double x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(double a, double b, double c) : x(a), y(b), z(c) {}
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
// Vect.h:43-75
inline V3 vnorm(const V3& v) { double m = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); return V3(v.x / m, v.y / m, v.z / m); }
inline V3 vneg(const V3& v) { return V3(-v.x, -v.y, -v.z); }
inline double vdot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 vcross(const V3& a, const V3& b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline V3 vadd(const V3& a, const V3& b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 vsub(const V3& a, const V3& b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 vmul(const V3& a, double s) { return V3(a.x * s, a.y * s, a.z * s); }
inline double vsqd(const V3& a, const V3& b) { return (a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y) + (a.z - b.z) * (a.z - b.z); }

struct Col {
    double r, g, b, s;
    Col() : r(0), g(0), b(0), s(0) {}
    Col(double r_, double g_, double b_, double s_) : r(r_), g(g_), b(b_), s(s_) {}
};
// Color.h:23-57 (the `special` of the left operand is kept)
inline Col cscale(const Col& c, double k) { return Col(c.r * k, c.g * k, c.b * k, c.s); }
inline Col cadd(const Col& a, const Col& b) { return Col(a.r + b.r, a.g + b.g, a.b + b.b, a.s); }
inline Col cmulc(const Col& a, const Col& b) { return Col(a.r * b.r, a.g * b.g, a.b * b.b, a.s); }
inline Col cclip(Col c)
{
    if (c.r > 1) c.r = 1;
    if (c.g > 1) c.g = 1;
    if (c.b > 1) c.b = 1;
    if (c.r < 0) c.r = 0;
    if (c.g < 0) c.g = 0;
    if (c.b < 0) c.b = 0;
    return c;
}

struct Box { double d[3][2]; };

enum Kind { TRI = 0, SPHERE = 1, PLANE = 2 };

struct Prim {
    int kind = TRI;
    V3 A, B, C;        // Triangle.h:15 (after the ctor's A,C,B argument order)
    V3 NA, NB, NC;     // Triangle.h:16
    bool hasSetNormal = false, hasTexture = false;
    double tA[2] = {0, 0}, tB[2] = {0, 0}, tC[2] = {0, 0};
    int map = -1;
    Col color;
    float ior = 1;
    V3 center; double radius = 0;      // Sphere.h:10-11
    V3 pnormal; double pdistance = 0;  // Plane.h:11-12
};

struct TexMap { int x_res = 0, y_res = 0; std::vector<double> rgb; };

struct Node {
    Box box;
    int child[2] = {-1, -1};
    std::vector<int> objs;
};

struct Photon { V3 pos; Col power; V3 dir; };

struct KNode {
    int axis = -1;
    int self = -1;
    std::vector<int> left, right;
    int lb = -1, rb = -1;
};

struct KTree { std::vector<KNode> nodes; int root = -1; };

}  // namespace

struct orc_scene {
    std::vector<Prim> prims;
    std::vector<int> bvh_objs;   // BVH members (planes have no box and are kept out)
    std::vector<int> planes;
    std::vector<TexMap> maps;
    std::vector<Node> nodes;     // nodes[0] = root
    std::vector<Photon> photons[2];
    KTree kd[2];
    std::mt19937 gen;
    std::vector<int32_t> dump_rec;
    std::vector<double> dump_boxes;
};

namespace {

// ---------------------------------------------------------------- primitives
// Triangle.h:223-227
V3 tri_normal(const Prim& p)
{
    V3 CA(p.C.x - p.A.x, p.C.y - p.A.y, p.C.z - p.A.z);
    V3 BA(p.B.x - p.A.x, p.B.y - p.A.y, p.B.z - p.A.z);
    return vnorm(vcross(CA, BA));
}

// Triangle.h:260-279
double tri_intersect(const Prim& p, const V3& o, const V3& d)
{
    V3 e1 = vsub(p.B, p.A);
    V3 e2 = vsub(p.C, p.A);
    V3 q = vcross(d, e2);
    double a = vdot(e1, q);
    if (std::fabs(a) < 0.00001) return -1;
    double f = 1 / a;
    V3 s = vsub(o, p.A);
    double u = f * vdot(s, q);
    if (u < 0.) return -1;
    V3 r = vcross(s, e1);
    double v = f * vdot(d, r);
    if (v < 0. || u + v > 1.) return -1;
    return f * vdot(r, e2);
}

// Sphere.h:44-90 (origin pre-offset by 0.1*dir; second root as written there)
double sphere_intersect(const Prim& p, const V3& o, const V3& d)
{
    double ox = o.x + 0.1 * d.x, oy = o.y + 0.1 * d.y, oz = o.z + 0.1 * d.z;
    double cx = p.center.x, cy = p.center.y, cz = p.center.z;
    double b = (2 * (ox - cx) * d.x) + (2 * (oy - cy) * d.y) + (2 * (oz - cz) * d.z);
    double c = (ox - cx) * (ox - cx) + (oy - cy) * (oy - cy) + (oz - cz) * (oz - cz) - p.radius * p.radius;
    double disc = b * b - 4 * c;
    if (disc > 0) {
        double root_1 = ((-1 * b - std::sqrt(disc)) / 2) - 0.000001;
        if (root_1 > 0) return root_1;
        return ((-1 * b + std::sqrt(disc) - b) / 2) - 0.000001;
    }
    return -1;
}

// Plane.h:26-39
double plane_intersect(const Prim& p, const V3& o, const V3& d)
{
    double a = vdot(d, p.pnormal);
    if (a == 0) return -1;
    double b = vdot(p.pnormal, vadd(o, vneg(vmul(p.pnormal, p.pdistance))));
    return -1 * b / a;
}

double prim_intersect(const Prim& p, const V3& o, const V3& d)
{
    switch (p.kind) {
        case TRI: return tri_intersect(p, o, d);
        case SPHERE: return sphere_intersect(p, o, d);
        default: return plane_intersect(p, o, d);
    }
}

// Triangle.h:115-125, Sphere.h:17-27
Box prim_box(const Prim& p)
{
    Box r;
    if (p.kind == TRI) {
        r.d[0][0] = std::min(p.A.x, std::min(p.B.x, p.C.x)); r.d[0][1] = std::max(p.A.x, std::max(p.B.x, p.C.x));
        r.d[1][0] = std::min(p.A.y, std::min(p.B.y, p.C.y)); r.d[1][1] = std::max(p.A.y, std::max(p.B.y, p.C.y));
        r.d[2][0] = std::min(p.A.z, std::min(p.B.z, p.C.z)); r.d[2][1] = std::max(p.A.z, std::max(p.B.z, p.C.z));
    } else {
        r.d[0][0] = p.center.x - p.radius; r.d[0][1] = p.center.x + p.radius;
        r.d[1][0] = p.center.y - p.radius; r.d[1][1] = p.center.y + p.radius;
        r.d[2][0] = p.center.z - p.radius; r.d[2][1] = p.center.z + p.radius;
    }
    return r;
}

// BVH.cpp:5-24 -- the ternaries are kept verbatim: they define the NaN behaviour
double box_intersect(const Box& bx, const V3& o, const V3& d, double current_best)
{
    double t0 = 0., t1 = current_best;
    for (int i = 0; i < 3; i++) {
        double inv_ray = 1. / d[i];
        double far_t = (bx.d[i][0] - o[i]) * inv_ray;
        double near_t = (bx.d[i][1] - o[i]) * inv_ray;
        if (near_t > far_t) std::swap(near_t, far_t);
        t0 = near_t > t0 ? near_t : t0;
        t1 = far_t > t1 ? t1 : far_t;
        if (t0 > t1) return -1;
    }
    return t0;
}

// BVH.cpp:31-40
Box box_union(const Box& a, const Box& b)
{
    Box r;
    for (int i = 0; i < 3; i++) { r.d[i][0] = std::min(a.d[i][0], b.d[i][0]); r.d[i][1] = std::max(a.d[i][1], b.d[i][1]); }
    return r;
}
// BVH.h:23-28, 33-38
inline double box_center(const Box& b, int ax) { return (b.d[ax][1] + b.d[ax][0]) / 2; }
inline double box_size(const Box& b) { return b.d[0][1] - b.d[0][0] + b.d[1][1] - b.d[1][0] + b.d[2][1] - b.d[2][0]; }

// ---------------------------------------------------------------- BVH build (BVH.cpp:42-106)
struct Builder {
    orc_scene* s;
    std::vector<Box> pbox;

    void calc_box(int n)
    {
        Node& nd = s->nodes[n];
        if (!nd.objs.empty()) {
            Box b = pbox[nd.objs[0]];
            for (int o : nd.objs) b = box_union(b, pbox[o]);
            nd.box = b;
        } else {
            nd.box = box_union(s->nodes[nd.child[0]].box, s->nodes[nd.child[1]].box);
        }
    }

    void split(int n)
    {
        if (s->nodes[n].objs.size() < 16) return;  // BVH.h:45, BVH.cpp:59
        std::vector<int> objs = s->nodes[n].objs;
        double span_axis[3];
        for (int i = 0; i < 3; i++) {
            auto cmp = [&](int a, int b) { return box_center(pbox[a], i) < box_center(pbox[b], i); };
            auto span = std::minmax_element(objs.begin(), objs.end(), cmp);
            span_axis[i] = box_center(pbox[*span.second], i) - box_center(pbox[*span.first], i);
        }
        int axis = int(std::max_element(span_axis, span_axis + 3) - span_axis);
        std::sort(objs.begin(), objs.end(), [&](int a, int b) { return box_center(pbox[a], axis) < box_center(pbox[b], axis); });
        size_t cnt = objs.size();
        std::vector<Box> left_box(cnt, pbox[objs[0]]), right_box(cnt, pbox[objs[cnt - 1]]);
        for (size_t i = 1; i < cnt; i++) {
            left_box[i] = box_union(left_box[i - 1], pbox[objs[i]]);
            right_box[i] = box_union(right_box[i - 1], pbox[objs[cnt - 1 - i]]);
        }
        std::vector<double> quality(cnt, 1e9);
        int bad_zone = std::max(1, (int)cnt / 3);
        for (size_t i = 0; i < cnt; i++) {
            if ((int)i < bad_zone || i > cnt - 1 - bad_zone) continue;
            quality[i] = box_size(left_box[i]) + box_size(right_box[cnt - 1 - i]);
        }
        int split_pos = int(std::min_element(quality.begin(), quality.end()) - quality.begin());
        int c0 = (int)s->nodes.size();
        s->nodes.emplace_back();
        s->nodes[c0].objs.assign(objs.begin(), objs.begin() + split_pos + 1);
        s->nodes[n].child[0] = c0;
        calc_box(c0);
        split(c0);
        int c1 = (int)s->nodes.size();
        s->nodes.emplace_back();
        s->nodes[c1].objs.assign(objs.begin() + split_pos + 1, objs.end());
        s->nodes[n].child[1] = c1;
        calc_box(c1);
        split(c1);
        s->nodes[n].objs.clear();
        calc_box(n);
    }
};

struct Counters { int64_t slab = 0, inner = 0, prim = 0, nodes = 0; };

// BVH.cpp:130-179
int shoot_node(const orc_scene* s, int n, const V3& o, const V3& d, double& distance, Counters* c)
{
    const Node& nd = s->nodes[n];
    if (c) c->slab++;
    if (box_intersect(nd.box, o, d, distance) < 0) return -1;
    if (c) c->nodes++;
    int result = -1;
    for (int obj : nd.objs) {
        double t = prim_intersect(s->prims[obj], o, d);
        if (c) c->prim++;
        if (t >= 0 && t < distance) { result = obj; distance = std::min(distance, t); }
    }
    if (nd.child[0] < 0) return result;
    if (c) { c->inner++; c->slab += 2; }
    double td[2];
    for (int i = 0; i < 2; i++) td[i] = box_intersect(s->nodes[nd.child[i]].box, o, d, distance);
    int first = td[0] < td[1] ? 0 : 1;
    int rc = shoot_node(s, nd.child[first], o, d, distance, c);
    if (rc >= 0) result = rc;
    rc = shoot_node(s, nd.child[1 - first], o, d, distance, c);
    if (rc >= 0) result = rc;
    return result;
}

// BVH.cpp:108-128.  Planes (no bounding box; the reference cannot instantiate them at all) are
// tested after the tree with the same accept rule -- an extension, not reference behaviour.
int shoot(const orc_scene* s, const V3& o, const V3& d, double& distance, Counters* c = nullptr)
{
    if (std::isnan(d.x)) return -1;
    int result = -1;
    if (!s->nodes.empty()) {
        if (c) c->slab++;
        if (!(box_intersect(s->nodes[0].box, o, d, distance) < 0)) result = shoot_node(s, 0, o, d, distance, c);
    }
    for (int pl : s->planes) {
        double t = plane_intersect(s->prims[pl], o, d);
        if (t >= 0 && t < distance) { result = pl; distance = std::min(distance, t); }
    }
    return result;
}

// ---------------------------------------------------------------- normals / colours
// Triangle.h:157-162
V3 tri_project(const Prim& p, const V3& above)
{
    V3 line_ap = vadd(above, vneg(p.A));
    V3 n = tri_normal(p);
    return vadd(vadd(vneg(vmul(n, vdot(line_ap, n))), line_ap), p.A);
}

// barycentric cascade shared by Triangle.h:77-97 and :193-213 (third branch typo kept)
void tri_bary(const Prim& p, const V3& position, double& b, double& c)
{
    double AP_x = position.x - p.A.x, AP_y = position.y - p.A.y, AP_z = position.z - p.A.z;
    double AB_x = p.B.x - p.A.x, AB_y = p.B.y - p.A.y, AB_z = p.B.z - p.A.z;
    double AC_x = p.C.x - p.A.x, AC_y = p.C.y - p.A.y, AC_z = p.C.z - p.A.z;
    if (AB_y * AC_x - AC_y * AB_x != 0) c = (AP_x * AB_y - AP_y * AB_x) / (AB_y * AC_x - AC_y * AB_x);
    else if (AC_y * AB_z - AC_z * AB_y != 0) c = (AP_y * AB_z - AP_z * AB_y) / (AC_y * AB_z - AC_z * AB_y);
    else c = (AP_x * AB_z - AP_z - AB_x) / (AC_x * AB_z - AC_z * AB_x);
    if (AB_x != 0) b = (AP_x - c * AC_x) / AB_x;
    else if (AB_y != 0) b = (AP_y - c * AC_y) / AB_y;
    else b = (AP_z - c * AC_z) / AB_z;
}

// Triangle.h:164-222
V3 tri_smooth_normal(const Prim& p, const V3& above)
{
    V3 position = tri_project(p, above);
    V3 NAB = vadd(p.NB, vneg(p.NA));
    V3 NAC = vadd(p.NC, vneg(p.NA));
    double b, c;
    tri_bary(p, position, b, c);
    V3 result = vnorm(vadd(p.NA, vadd(vmul(NAB, b), vmul(NAC, c))));
    if (vdot(result, tri_normal(p)) < 0) result = vneg(result);
    return result;
}

// Triangle.h:235-240, Sphere.h:33-36, Plane.h:22-24
V3 prim_normal(const Prim& p, const V3& pt)
{
    if (p.kind == TRI) return p.hasSetNormal ? tri_smooth_normal(p, pt) : tri_normal(p);
    if (p.kind == SPHERE) return vnorm(vadd(pt, vneg(p.center)));
    return p.pnormal;
}

// Triangle.h:242-256, Sphere.h:40-42
V3 prim_tangent(const Prim& p, const V3& pt)
{
    if (p.kind == TRI) {
        if (!p.hasSetNormal) return vnorm(V3(p.C.x - p.A.x, p.C.y - p.A.y, p.C.z - p.A.z));
        V3 n = tri_smooth_normal(p, pt);
        if (n.y == 0 && n.z == 0) return V3(0, 1, 0);
        return vnorm(vcross(n, V3(1, 0, 0)));
    }
    return V3(1, 0, 0);
}

// TextureMap.h:127-182
Col tex_get(const TexMap& m, double x, double y)
{
    double dot_x = x, dot_y = y;
    if (dot_x > 0) { while (dot_x - 1 > 0) dot_x -= 1; }
    else if (dot_x < 0) { do { dot_x += 1; } while (dot_x < 0); }
    if (dot_y > 0) { while (dot_y - 1 > 0) dot_y -= 1; }
    else if (dot_y < 0) { do { dot_y += 1; } while (dot_y < 0); }
    dot_x = dot_x * (m.x_res - 1);
    dot_y = dot_y * (m.y_res - 1);
    int floor_x = int(dot_x);
    int ceil_x = (floor_x + 1) >= m.x_res ? m.x_res - 1 : floor_x + 1;
    int floor_y = int(dot_y);
    int ceil_y = (floor_y + 1) >= m.y_res ? m.y_res - 1 : floor_y + 1;
    auto at = [&](int i, int j, int ch) { return m.rgb[((size_t)i * m.y_res + j) * 3 + ch]; };
    double s = dot_x - floor_x, t = dot_y - floor_y;
    double out[3];
    for (int ch = 0; ch < 3; ch++) {
        double lu = at(floor_x, floor_y, ch), ld = at(ceil_x, floor_y, ch), ru = at(floor_x, ceil_y, ch), rd = at(ceil_x, ceil_y, ch);
        out[ch] = lu * (1 - s) * (1 - t) + ld * s * (1 - t) + ru * (1 - s) * t + rd * s * t;
    }
    return Col(out[0], out[1], out[2], 0);
}

// Triangle.h:29-104, Sphere.h:31
Col prim_color(const orc_scene* s, const Prim& p, const V3& above)
{
    if (p.kind == TRI && p.hasTexture) {
        V3 position = tri_project(p, above);
        double b, c;
        tri_bary(p, position, b, c);
        double ABx = p.tB[0] + p.tA[0] * (-1), ABy = p.tB[1] + (-1) * p.tA[1];
        double ACx = p.tC[0] + p.tA[0] * (-1), ACy = p.tC[1] + (-1) * p.tA[1];
        double u = (ABx * b + ACx * c) + p.tA[0];
        double v = (ABy * b + ACy * c) + p.tA[1];
        return tex_get(s->maps[p.map], u, v);
    }
    return p.color;
}

// ---------------------------------------------------------------- kd-tree (Kdtree.cpp:11-61)
int kd_create(const std::vector<Photon>& ph, KTree& t, std::vector<int>& idx)
{
    int me = (int)t.nodes.size();
    t.nodes.emplace_back();
    double span_axis[3];
    for (int i = 0; i < 3; i++) {
        auto span = std::minmax_element(idx.begin(), idx.end(), [&](int a, int b) { return ph[a].pos[i] < ph[b].pos[i]; });
        span_axis[i] = ph[*span.second].pos[i] - ph[*span.first].pos[i];
    }
    int axis = int(std::max_element(span_axis, span_axis + 3) - span_axis);
    t.nodes[me].axis = axis;
    std::sort(idx.begin(), idx.end(), [&](int a, int b) { return ph[a].pos[axis] < ph[b].pos[axis]; });
    int mid = (int)idx.size() / 2;
    t.nodes[me].self = idx[mid];
    std::vector<int> left(idx.begin(), idx.begin() + mid), right(idx.begin() + mid + 1, idx.end());
    if (left.size() <= 8) t.nodes[me].left = left;
    else { int c = kd_create(ph, t, left); t.nodes[me].lb = c; }
    if (right.size() <= 8) t.nodes[me].right = right;
    else { int c = kd_create(ph, t, right); t.nodes[me].rb = c; }
    return me;
}

struct KnnCmp {
    const std::vector<Photon>* ph; V3 c;
    bool operator()(int a, int b) const { return vsqd((*ph)[a].pos, c) < vsqd((*ph)[b].pos, c); }  // Kdtree.h:34-37
};
typedef std::priority_queue<int, std::vector<int>, KnnCmp> KnnQ;

// Kdtree.cpp:73-90
void kd_update(const std::vector<Photon>& ph, int k, int p, const V3& center, KnnQ& q, double& best, double r2)
{
    double dist = vsqd(ph[p].pos, center);
    if ((int)q.size() < k && dist < r2) { q.push(p); return; }
    if (dist < best && dist < r2) {
        int prev_top = q.top();
        q.push(p);
        int top = q.top();
        if (prev_top != top) best = vsqd(ph[top].pos, center);
        q.pop();
    }
}

// Kdtree.cpp:92-133
void kd_find(const std::vector<Photon>& ph, const KTree& t, int n, int k, const V3& center, KnnQ& q, double& best, double r2)
{
    if (n < 0) return;
    const KNode& nd = t.nodes[n];
    float farDistance = (float)std::min(r2 * 1., best);
    for (int p : nd.left) kd_update(ph, k, p, center, q, best, r2);
    for (int p : nd.right) kd_update(ph, k, p, center, q, best, r2);
    kd_update(ph, k, nd.self, center, q, best, r2);
    double delta = center[nd.axis] - ph[nd.self].pos[nd.axis];
    if (center[nd.axis] < ph[nd.self].pos[nd.axis]) {
        kd_find(ph, t, nd.lb, k, center, q, best, r2);
        if (delta * delta <= farDistance) kd_find(ph, t, nd.rb, k, center, q, best, r2);
    } else {
        kd_find(ph, t, nd.rb, k, center, q, best, r2);
        if (delta * delta <= farDistance) kd_find(ph, t, nd.lb, k, center, q, best, r2);
    }
}

// Kdtree.cpp:135-158: nearest first
std::vector<int> kd_knn(const orc_scene* s, int which, int k, const V3& center, double photon_dist)
{
    const auto& ph = s->photons[which];
    double best = 1e9;
    double r2 = photon_dist * photon_dist;
    KnnQ q(KnnCmp{&ph, center});
    kd_find(ph, s->kd[which], s->kd[which].root, k, center, q, best, r2);
    std::vector<int> result;
    while (!q.empty()) { result.push_back(q.top()); q.pop(); }
    std::reverse(result.begin(), result.end());
    return result;
}

// ---------------------------------------------------------------- shading (main.cpp:125-392)
const double PI_REF = 3.1415926;   // main.cpp:24
const double NATUREE = 2.71828;    // main.cpp:26

V3 reflect_dir(const V3& normal, const V3& ray_in)
{
    double dot1 = vdot(normal, vneg(ray_in));
    if (dot1 <= 0.02) return ray_in;
    V3 scalar1 = vmul(normal, dot1);
    V3 add1 = vadd(scalar1, ray_in);
    V3 scalar2 = vmul(add1, 2);
    V3 add2 = vadd(vneg(ray_in), scalar2);
    return vnorm(add2);
}

inline float gauss_weight(float distanceSqr, float maxDistSqr)
{
    return (float)(0.918 * (1 - (1 - std::pow(NATUREE, -1.953 * (distanceSqr) / (2 * maxDistSqr))) / (1 - std::pow(NATUREE, -1.953))));
}

struct ShadeCtx {
    const orc_scene* s;
    const orc_params* p;
    int64_t rays = 0;
    std::mt19937* gen = nullptr;  // eye-path RNG: glossy rays (main.cpp:140-141), SSS march (main.cpp:374), lens jitter (main.cpp:803-804)
};

// counter_rng = 1: uniform in (0, 1) as a function of two counters and the seed
double counter_uniform(uint32_t a, uint32_t b, uint32_t seed)
{
    uint32_t x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ seed * 0xC2B2AE3Du;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return ((double)x + 0.5) * (1.0 / 4294967296.0);
}
uint32_t fbits(double v)
{
    float f = (float)v;
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
uint32_t path_key(const V3& p) { return fbits(p.x) * 0x9E3779B1u ^ fbits(p.y) * 0x85EBCA77u ^ fbits(p.z) * 0xC2B2AE3Du; }

// computeGlossyRay, main.cpp:139-156; a, b are its two draws in call order
V3 glossy_ray(const V3& reflection_direction, const V3& normal, double glossiness, double a, double b)
{
    double theta = std::acos(std::pow(1 - a, glossiness));
    double phi = 2 * PI_REF * b;
    double x = std::sin(theta) * std::cos(phi);
    double z = std::sin(theta) * std::sin(phi);
    V3 u = vcross(reflection_direction, normal);
    V3 v = vcross(reflection_direction, u);
    return vnorm(vadd(vmul(u, x), vadd(vmul(v, z), reflection_direction)));
}
const int GLOSSY_SAMPLE = 16;   // main.cpp:32

Col color_at(ShadeCtx& cx, V3 hit_pos, const V3& hit_dir, int hit_obj, int depth)
{
    const orc_scene* s = cx.s;
    const orc_params& P = *cx.p;
    double accuracy = P.accuracy;
    Col final_color(0, 0, 0, 0);
    if (depth > P.tracing_depth) return cclip(final_color);
    hit_pos = vadd(hit_pos, vmul(hit_dir, accuracy));
    const Prim& obj = s->prims[hit_obj];
    Col hit_color = prim_color(s, obj, hit_pos);
    V3 hit_normal = prim_normal(obj, hit_pos);
    double colorSpecial = hit_color.s;

    if (colorSpecial > 0) {
        V3 R = reflect_dir(hit_normal, hit_dir);
        Col acc(0, 0, 0, 0);
        // main.cpp:177-179: a material whose special * 100 ends in .7654 takes GLOSSY_SAMPLE jittered rays
        double frac = colorSpecial * 100.0 - std::floor(colorSpecial * 100.0);
        int sample = (frac - 0.7654 < 0.0001 && frac - 0.7653 > 0) ? GLOSSY_SAMPLE : 1;
        std::uniform_real_distribution<> dis_rand(0, 1);
        const uint32_t key = path_key(hit_pos);
        for (int i = 0; i < sample; i++) {
            V3 dir = R;
            if (sample != 1) {
                double a, b;
                if (P.counter_rng) {
                    uint32_t draw = (uint32_t)(depth * 1024 + i * 2);
                    a = counter_uniform(key, draw, P.rng_seed); b = counter_uniform(key, draw + 1u, P.rng_seed);
                } else {
                    a = dis_rand(*cx.gen); b = dis_rand(*cx.gen);
                }
                dir = glossy_ray(R, hit_normal, 0.2, a, b);
            }
            double dist = 1e9;
            int io = shoot(s, hit_pos, dir, dist);
            cx.rays++;
            bool go = P.head_quirk ? (dist > accuracy) : (io >= 0 && dist > accuracy);
            if (go) {
                V3 npos = vadd(hit_pos, vmul(R, dist));                       // main.cpp:195: along the un-jittered direction
                Col rc = color_at(cx, npos, R, P.head_quirk ? hit_obj : io, depth + 1);
                if (colorSpecial > 1) acc = cadd(acc, cscale(rc, 2 - hit_color.s));
                else acc = cadd(acc, rc);
            }
        }
        acc = cscale(acc, 1.0 / sample);
        final_color = cadd(final_color, acc);
        if (colorSpecial > 1) {
            Col racc(0, 0, 0, 0);
            for (int i = 0; i < sample; i++) {
                double refractance = hit_color.s - 1;
                double n1n2 = 1.0 / obj.ior;
                double dot1 = vdot(hit_normal, vneg(hit_dir));
                if (dot1 < 0) {                                                 // (flips the normal for the later passes, too)
                    hit_normal = vneg(hit_normal);
                    dot1 = vdot(hit_normal, vneg(hit_dir));
                    n1n2 = 1.0 / n1n2;
                }
                double nNL = n1n2 * dot1;
                double underSQRT = 1 - n1n2 * n1n2 * (1 - dot1 * dot1);
                if (underSQRT > 0) {
                    double coeffN = nNL - std::sqrt(underSQRT);
                    V3 T = vadd(vmul(hit_normal, coeffN), vneg(vmul(vneg(hit_dir), n1n2)));
                    V3 dir = T;
                    if (sample != 1) {
                        double a, b;
                        if (P.counter_rng) {
                            uint32_t draw = (uint32_t)(depth * 1024 + 512 + i * 2);
                            a = counter_uniform(key, draw, P.rng_seed); b = counter_uniform(key, draw + 1u, P.rng_seed);
                        } else {
                            a = dis_rand(*cx.gen); b = dis_rand(*cx.gen);
                        }
                        dir = glossy_ray(T, hit_normal, 0.2, a, b);
                    }
                    double dist = 1e9;
                    int io = shoot(s, hit_pos, dir, dist);
                    cx.rays++;
                    bool go = P.head_quirk ? (dist > accuracy) : (io >= 0 && dist > accuracy);
                    if (go) {
                        V3 npos = vadd(hit_pos, vmul(T, accuracy + dist));
                        Col rc = color_at(cx, npos, T, P.head_quirk ? hit_obj : io, depth + 1);
                        racc = cadd(racc, cscale(rc, refractance));
                    }
                }
                racc = cscale(racc, 1.0 / sample);                              // main.cpp:255-256: inside the loop
                final_color = cadd(final_color, racc);
            }
        }
    } else if (colorSpecial <= 0) {
        Col sssColor(0, 0, 0, 0);
        if (P.use_photon_map && s->kd[0].root >= 0) {
            const auto& ph = s->photons[0];
            std::vector<int> found = kd_knn(s, 0, P.photon_use, hit_pos, P.photon_dist);
            float maxDistSqr = -1;
            if (!found.empty()) maxDistSqr = (float)vsqd(ph[found.back()].pos, hit_pos);
            for (size_t i = 0; i < found.size(); i++) {
                float distanceSqr = (float)vsqd(ph[found[i]].pos, hit_pos);
                float weight = gauss_weight(distanceSqr, maxDistSqr);
                V3 light_direction = vneg(ph[found[i]].dir);
                float cosine_angle = (float)vdot(hit_normal, light_direction);
                if (cosine_angle < 0) cosine_angle = -cosine_angle;
                final_color = cadd(final_color, cscale(cmulc(hit_color, ph[found[i]].power), cosine_angle * weight));
            }
            if (!found.empty()) final_color = cscale(final_color, 1.0 / (P.photon_num / 16 * PI_REF * maxDistSqr));
        }
        if (colorSpecial < 0 && P.use_photon_map && s->kd[1].root >= 0) {  // main.cpp:297-388
            const auto& vp = s->photons[1];
            double n1n2 = 1.0 / obj.ior;
            double dot1 = vdot(hit_normal, vneg(hit_dir));
            if (dot1 < 0) { hit_normal = vneg(hit_normal); dot1 = vdot(hit_normal, vneg(hit_dir)); n1n2 = 1.0 / n1n2; }
            double nNL = n1n2 * dot1;
            double underSQRT = 1 - n1n2 * n1n2 * (1 - dot1 * dot1);
            std::vector<int> vf = kd_knn(s, 1, P.photon_use, hit_pos, P.photon_dist);
            float maxDistSqr = -1;
            Col firstStep(0, 0, 0, 0);
            if (!vf.empty()) maxDistSqr = (float)vsqd(vp[vf.back()].pos, hit_pos);
            for (size_t i = 0; i < vf.size(); i++)
                firstStep = cadd(firstStep, cscale(cmulc(hit_color, vp[vf[i]].power), 0.2));
            if (!vf.empty()) firstStep = cscale(firstStep, 1.0 / (std::pow(NATUREE, 0) * P.photon_num / 8 * PI_REF * maxDistSqr * std::sqrt(maxDistSqr)));
            sssColor = cadd(sssColor, firstStep);
            if (underSQRT > 0) {
                double coeffN = nNL - std::sqrt(underSQRT);
                V3 T = vadd(vmul(hit_normal, coeffN), vneg(vmul(vneg(hit_dir), n1n2)));
                V3 pos = vadd(hit_pos, vmul(T, 0.02));
                std::uniform_real_distribution<> dis_rand(0, 1);
                for (double pass = 1; pass < 3; pass++) {
                    std::vector<int> f2 = kd_knn(s, 1, P.photon_use, pos, P.photon_dist);
                    float m2 = -1;
                    Col cur(0, 0, 0, 0);
                    if (!f2.empty()) m2 = (float)vsqd(vp[f2.back()].pos, hit_pos);
                    for (size_t i = 0; i < f2.size(); i++)
                        cur = cadd(cur, cscale(cmulc(hit_color, vp[f2[i]].power), 0.2));
                    if (!f2.empty()) cur = cscale(cur, 1.0 / (std::pow(NATUREE, pass) * P.photon_num / 8 * PI_REF * m2 * std::sqrt(m2)));
                    sssColor = cadd(sssColor, cur);
                    double d = -std::log(dis_rand(*cx.gen)) / 200;
                    pos = vadd(pos, vmul(T, d));
                }
            }
        }
        final_color = cadd(final_color, sssColor);
    }
    return cclip(final_color);
}

// main.cpp:626-635 camera basis and :753-790 sample position
struct Cam { V3 pos, dir, right, down; };
Cam make_cam(const double* cam)
{
    V3 Y(0, 1, 0);
    V3 camPos(cam[0], cam[1], cam[2]), look_at(cam[3], cam[4], cam[5]);
    V3 diff(camPos.x - look_at.x, camPos.y - look_at.y, camPos.z - look_at.z);
    Cam c;
    c.pos = camPos;
    c.dir = vnorm(vneg(diff));
    c.right = vnorm(vcross(Y, c.dir));
    c.down = vcross(c.right, c.dir);
    return c;
}

V3 cam_ray_dir(const Cam& c, int width, int height, int aadepth, int x, int y, int aax)
{
    double aspectRatio = (double)width / (double)height;
    double xamnt, yamnt;
    if (aadepth == 1) {
        if (width > height) {
            xamnt = ((x + 0.5) / width) * aspectRatio - (((width - height) / (double)height) / 2);
            yamnt = ((height - y) + 0.5) / height;
        } else if (height > width) {
            xamnt = (x + 0.5) / width;
            yamnt = (((height - y) + 0.5) / height) / aspectRatio - (((height - width) / (double)width) / 2);
        } else {
            xamnt = (x + 0.5) / width;
            yamnt = ((height - y) + 0.5) / height;
        }
    } else {  // main.cpp:771-787: aax is used for both offsets
        double off = (double)aax / ((double)aadepth - 1);
        if (width > height) {
            xamnt = ((x + off) / width) * aspectRatio - (((width - height) / (double)height) / 2);
            yamnt = ((height - y) + off) / height;
        } else if (height > width) {
            xamnt = (x + off) / width;
            yamnt = (((height - y) + off) / height) / aspectRatio - (((height - width) / (double)width) / 2);
        } else {
            xamnt = (x + off) / width;
            yamnt = ((height - y) + off) / height;
        }
    }
    return vnorm(vadd(c.dir, vadd(vmul(c.right, xamnt - 0.5), vmul(c.down, yamnt - 0.5))));
}

// ---------------------------------------------------------------- photon emission (main.cpp:69-123, 396-598)
double fresnel(float cosThetaI, float n1, float n2, int type)
{
    float sin2ThetaT = (n1 / n2) * (n1 / n2) * (1 - cosThetaI * cosThetaI);
    float cosThetaT = std::sqrt(1 - sin2ThetaT);
    float Rs = (n1 * cosThetaI - n2 * cosThetaT) / (n1 * cosThetaI + n2 * cosThetaT);
    float Rp = (n2 * cosThetaI - n1 * cosThetaT) / (n2 * cosThetaI + n1 * cosThetaT);
    if (type == 0) return (Rs * Rs + Rp * Rp) / 2;
    return 1 - (Rs * Rs + Rp * Rp) / 2;
}

struct Emit {
    orc_scene* s;
    const orc_params* P;
    std::uniform_real_distribution<> dis_rand{0, 1};
    std::uniform_real_distribution<> dis_rand_100{0, 100};

    void emit(V3 org, V3 dir, Col lightColor, int bounce)
    {
        if (bounce > P->bounce) return;
        double dist = 1e9;
        int io = shoot(s, org, dir, dist);
        if (io < 0) return;
        if (!(dist > P->accuracy)) return;
        const Prim& obj = s->prims[io];
        V3 ipos = vadd(org, vmul(dir, P->accuracy + dist));
        V3 idir = dir;
        V3 normal = prim_normal(obj, ipos);
        float reflectance = 1;
        int refractMask[3] = {1, 1, 1};
        bool canTransmit = true;
        double special = prim_color(s, obj, ipos).s;
        if (special <= 0) {
            // storePhoton (main.cpp:69-96)
            Col oc = prim_color(s, obj, ipos);
            Col fc;
            fc.r = std::min(oc.r, lightColor.r); fc.g = std::min(oc.g, lightColor.g); fc.b = std::min(oc.b, lightColor.b);
            s->photons[0].push_back(Photon{ipos, lightColor, idir});
            lightColor = cclip(fc);
            lightColor = cscale(lightColor, 0.8);
        } else if (special < 1 && special > 0) {
            float cosTheta = (float)vdot(normal, vneg(idir));
            float n2 = obj.ior;
            reflectance = (float)fresnel(cosTheta, 1, n2, 0);
            int roll = (int)dis_rand_100(s->gen);
            if (roll > reflectance * 100) canTransmit = false;
        } else if (special < 2 && special > 0) {
            for (int i = 0; i < 3; i++) {
                float cosTheta = (float)vdot(normal, vneg(idir));
                float n2 = (float)(obj.ior - 0.01 + i / 100.0);
                if (cosTheta > 0) reflectance = (float)fresnel(cosTheta, 1, n2, 1);
                else {
                    cosTheta = (float)vdot(vneg(normal), vneg(idir));
                    reflectance = (float)fresnel(cosTheta, n2, 1, 1);
                }
                int roll = (int)dis_rand_100(s->gen);
                if (roll > reflectance * 100) refractMask[i] = 0;
            }
        }
        if (!canTransmit) return;
        if (prim_color(s, obj, ipos).s == 0) {
            double Zeta1 = dis_rand(s->gen);
            double Zeta2 = dis_rand(s->gen);
            double theta = std::acos(std::sqrt(Zeta1));
            double phi = 2 * PI_REF * Zeta2;
            double x = std::sin(theta) * std::cos(phi);
            double z = std::sin(theta) * std::sin(phi);
            double y = std::cos(theta);
            V3 u = prim_tangent(obj, ipos);
            V3 v = vcross(normal, u);
            V3 w = vadd(vmul(u, x), vmul(vadd(vmul(v, z), normal), y));
            w = vnorm(w);
            emit(ipos, w, lightColor, bounce + 1);
        } else if (special <= 1 && special > 0) {
            double dot1 = vdot(normal, vneg(idir));
            V3 scalar1 = vmul(normal, dot1);
            V3 add1 = vadd(scalar1, idir);
            V3 scalar2 = vmul(add1, 2);
            V3 add2 = vadd(vneg(idir), scalar2);
            V3 rdir = vnorm(add2);
            Col oc = prim_color(s, obj, ipos);
            lightColor.r = std::min(oc.r, lightColor.r); lightColor.g = std::min(oc.g, lightColor.g); lightColor.b = std::min(oc.b, lightColor.b);
            emit(ipos, rdir, lightColor, bounce + 1);
        } else if (special <= 2 && special > 1) {
            for (int i = 0; i < 1; i++) {
                if (refractMask[i] == 0) continue;
                double n1n2 = 1.0 / (obj.ior - 0.01 + i / 100.0);
                double dot1 = vdot(normal, vneg(idir));
                if (dot1 < 0) { normal = vneg(normal); dot1 = vdot(normal, vneg(idir)); n1n2 = 1.0 / n1n2; }
                double nNL = n1n2 * dot1;
                double underSQRT = 1 - n1n2 * n1n2 * (1 - dot1 * dot1);
                if (underSQRT > 0) {
                    double coeffN = nNL - std::sqrt(underSQRT);
                    V3 T = vadd(vmul(normal, coeffN), vneg(vmul(vneg(idir), n1n2)));
                    Col rc = lightColor;
                    if (!(rc.r == 0 && rc.g == 0 && rc.b == 0)) {
                        Col oc = prim_color(s, obj, ipos);
                        rc.r = std::min(oc.r, lightColor.r); rc.g = std::min(oc.g, lightColor.g); rc.b = std::min(oc.b, lightColor.b);
                        emit(ipos, T, rc, bounce + 1);
                    }
                }
            }
        }
        if (special < 0) {  // main.cpp:559-595
            double d = 0.005;
            V3 pos = vadd(ipos, vmul(idir, d));
            std::uniform_real_distribution<> phi_rand(-PI_REF, PI_REF);
            for (int k = 0; k < 4; k++) {
                double phi = phi_rand(s->gen);
                double g = 0.3;
                double q = ((1 - g * g) / (1 - g + 2 * g * dis_rand(s->gen)));
                double newCosTheta = 1 / std::fabs(2 * g) * (1 + g * g - q * q);
                double newTheta = std::acos(newCosTheta);
                double nx = std::sin(newTheta) * std::cos(phi);
                double ny = std::sin(newTheta) * std::sin(phi);
                double nz = std::cos(newTheta);
                V3 sdir = vnorm(V3(nx, ny, nz));
                double id = d;
                int o2 = shoot(s, pos, sdir, id);
                if (o2 < 0) {
                    pos = vadd(pos, vmul(sdir, d));
                    s->photons[1].push_back(Photon{pos, lightColor, sdir});
                    lightColor = cscale(lightColor, 0.6);
                } else break;
            }
        }
    }
};

V3 rd3(const double* p) { return V3(p[0], p[1], p[2]); }
void wr3(double* p, const V3& v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

int add_tri(orc_scene* s, const V3& A, const V3& C, const V3& B, const Col& c, float ior)
{
    Prim p;
    p.kind = TRI;
    p.A = A; p.B = B; p.C = C;  // Triangle.h:137-140
    p.NA = p.NB = p.NC = tri_normal(p);
    p.color = c;
    p.ior = ior;
    s->prims.push_back(p);
    s->bvh_objs.push_back((int)s->prims.size() - 1);
    return (int)s->prims.size() - 1;
}

}  // namespace

extern "C" {

void orc_default_params(orc_params* p)
{
    p->photon_num = 50000;
    p->photon_use = 1000;
    p->tracing_depth = 4;
    p->bounce = 5;
    p->head_quirk = 0;
    p->focal_len = 2.6;
    p->aperture = 0.2;
    p->dof_samples = 1;
    p->counter_rng = 0;
    p->rng_seed = 12345u;
    p->pad_ = 0;
    p->use_photon_map = 1;
    p->accuracy = 0.000001;
    p->photon_dist = 0.28;
}

orc_scene* orc_scene_new(void)
{
    orc_scene* s = new orc_scene();
    s->kd[0].root = s->kd[1].root = -1;
    return s;
}
void orc_scene_free(orc_scene* s) { delete s; }
int32_t orc_scene_size(const orc_scene* s) { return (int32_t)s->prims.size(); }

int32_t orc_add_triangle(orc_scene* s, const double* A, const double* C, const double* B, const double* rgbs, float ior)
{
    return add_tri(s, rd3(A), rd3(C), rd3(B), Col(rgbs[0], rgbs[1], rgbs[2], rgbs[3]), ior);
}
// Triangle.h:150-156
void orc_last_set_normals(orc_scene* s, const double* na, const double* nc, const double* nb)
{
    Prim& p = s->prims.back();
    p.NA = rd3(na); p.NB = rd3(nb); p.NC = rd3(nc);
    p.hasSetNormal = true;
}
// Triangle.h:106-113
void orc_last_set_texture(orc_scene* s, const double* ta, const double* tc, const double* tb, int32_t use, int32_t map_idx)
{
    Prim& p = s->prims.back();
    p.tA[0] = ta[0]; p.tA[1] = ta[1]; p.tB[0] = tb[0]; p.tB[1] = tb[1]; p.tC[0] = tc[0]; p.tC[1] = tc[1];
    p.hasTexture = use != 0;
    p.map = map_idx;
}
int32_t orc_add_sphere(orc_scene* s, const double* c, double radius, const double* rgbs, float ior)
{
    Prim p;
    p.kind = SPHERE; p.center = rd3(c); p.radius = radius; p.color = Col(rgbs[0], rgbs[1], rgbs[2], rgbs[3]); p.ior = ior;
    s->prims.push_back(p);
    s->bvh_objs.push_back((int)s->prims.size() - 1);
    return (int)s->prims.size() - 1;
}
int32_t orc_add_plane(orc_scene* s, const double* normal, double distance, const double* rgbs)
{
    Prim p;
    p.kind = PLANE; p.pnormal = rd3(normal); p.pdistance = distance; p.color = Col(rgbs[0], rgbs[1], rgbs[2], rgbs[3]); p.ior = 1;
    s->prims.push_back(p);
    s->planes.push_back((int)s->prims.size() - 1);
    return (int)s->prims.size() - 1;
}

// IO.cpp:108-154
void orc_add_cornell(orc_scene* s, const double* c1, const double* c2)
{
    V3 corner1 = rd3(c1), corner2 = rd3(c2);
    V3 A(c2[0], c1[1], c1[2]), B(c2[0], c1[1], c2[2]), C(c1[0], c1[1], c2[2]);
    V3 D(c2[0], c2[1], c1[2]), E(c1[0], c2[1], c1[2]), F(c1[0], c2[1], c2[2]);
    Col red(1, 0.25, 0.25, 0), green(0.25, 1, 0.25, 0), white(1.0, 1.0, 1.0, 0);
    add_tri(s, D, A, corner1, green, 20); add_tri(s, corner1, E, D, green, 20);
    add_tri(s, corner2, B, A, white, 20); add_tri(s, A, D, corner2, white, 20);
    add_tri(s, F, C, B, red, 20);         add_tri(s, B, corner2, F, red, 20);
    add_tri(s, D, E, F, white, 20);       add_tri(s, F, corner2, D, white, 20);
    add_tri(s, corner1, A, B, white, 20); add_tri(s, B, C, corner1, white, 20);
}
// IO.cpp:70-106
void orc_add_cube(orc_scene* s, const double* c1, const double* c2, const double* rgbs)
{
    V3 corner1 = rd3(c1), corner2 = rd3(c2);
    V3 A(c2[0], c1[1], c1[2]), B(c2[0], c1[1], c2[2]), C(c1[0], c1[1], c2[2]);
    V3 D(c2[0], c2[1], c1[2]), E(c1[0], c2[1], c1[2]), F(c1[0], c2[1], c2[2]);
    Col col(rgbs[0], rgbs[1], rgbs[2], rgbs[3]);
    add_tri(s, D, A, corner1, col, 2); add_tri(s, corner1, E, D, col, 2);
    add_tri(s, corner2, B, A, col, 2); add_tri(s, A, D, corner2, col, 2);
    add_tri(s, F, C, B, col, 2);       add_tri(s, B, corner2, F, col, 2);
    add_tri(s, E, corner1, C, col, 2); add_tri(s, C, F, E, col, 2);
    add_tri(s, D, E, F, col, 2);       add_tri(s, F, corner2, D, col, 2);
    add_tri(s, corner1, A, B, col, 2); add_tri(s, B, C, corner1, col, 2);
}

// IO.h:61-124.  Same word-by-word tokenisation; unlike the reference, a face corner may omit the
// texture and/or normal index (`v`, `v/vt`, `v//vn`): the triangle then keeps its flat normal /
// stays untextured, which is what the reference constructs before setNormals/setTexture.
int32_t orc_load_obj(orc_scene* s, const char* path, const double* rgbs, float ior, float xo, float yo, float zo,
                     int32_t use_texture, int32_t map_idx)
{
    std::ifstream input(path);
    if (!input) return 0;
    std::vector<V3> points, normals;
    std::vector<std::pair<double, double>> tex;
    std::string head;
    Col col(rgbs[0], rgbs[1], rgbs[2], rgbs[3]);
    while (input >> head) {
        if (head == "v") {
            double a, b, c;
            input >> a >> b >> c;
            points.push_back(V3(a + xo, b + yo, c + zo));
        } else if (head == "vn") {
            double a, b, c;
            input >> a >> b >> c;
            normals.push_back(V3(a, b, c));
        } else if (head == "vt") {
            double a, b;
            input >> a >> b;
            tex.push_back({a, b});
        } else if (head == "f") {
            std::string c[3];
            input >> c[0] >> c[1] >> c[2];
            int vi[3], ti[3], ni[3];
            for (int k = 0; k < 3; k++) {
                vi[k] = ti[k] = ni[k] = -1;
                size_t p1 = c[k].find('/');
                vi[k] = atoi(c[k].substr(0, p1).c_str()) - 1;
                if (p1 != std::string::npos) {
                    size_t p2 = c[k].find('/', p1 + 1);
                    std::string ts = c[k].substr(p1 + 1, p2 == std::string::npos ? std::string::npos : p2 - p1 - 1);
                    if (!ts.empty()) ti[k] = atoi(ts.c_str()) - 1;
                    if (p2 != std::string::npos && p2 + 1 < c[k].size()) ni[k] = atoi(c[k].substr(p2 + 1).c_str()) - 1;
                }
            }
            add_tri(s, points.at(vi[0]), points.at(vi[1]), points.at(vi[2]), col, ior);
            Prim& p = s->prims.back();
            if (ni[0] >= 0 && ni[1] >= 0 && ni[2] >= 0) {
                // setNormals(a, c, b): Triangle.h:150
                p.NA = normals.at(ni[0]); p.NC = normals.at(ni[1]); p.NB = normals.at(ni[2]);
                p.hasSetNormal = true;
            }
            if (ti[0] >= 0 && ti[1] >= 0 && ti[2] >= 0) {
                // setTexture(tA, tC, tB, ...): Triangle.h:106
                p.tA[0] = tex.at(ti[0]).first; p.tA[1] = tex.at(ti[0]).second;
                p.tC[0] = tex.at(ti[1]).first; p.tC[1] = tex.at(ti[1]).second;
                p.tB[0] = tex.at(ti[2]).first; p.tB[1] = tex.at(ti[2]).second;
                p.hasTexture = use_texture != 0;
                p.map = map_idx;
            }
        }
    }
    return 1;
}

// TextureMap.h:20-61
int32_t orc_texture_read(orc_scene* s, const char* path)
{
    FILE* fp = fopen(path, "rb");
    if (!fp) return -1;
    unsigned char header[54];
    if (fread(header, 1, 54, fp) != 54) { fclose(fp); return -1; }
    TexMap m;
    for (int i = 18; i <= 21; i++) m.y_res += header[i] * (int)std::pow(256, i - 18);
    for (int i = 22; i <= 25; i++) m.x_res += header[i] * (int)std::pow(256, i - 22);
    m.rgb.assign((size_t)m.x_res * m.y_res * 3, 0.0);
    for (int i = 0; i < m.x_res; i++)
        for (int j = 0; j < m.y_res; j++) {
            unsigned char px[3] = {0, 0, 0};
            if (fread(px, 1, 3, fp) != 3) break;
            size_t o = ((size_t)i * m.y_res + j) * 3;
            m.rgb[o] = px[2] / 255.0; m.rgb[o + 1] = px[1] / 255.0; m.rgb[o + 2] = px[0] / 255.0;
        }
    fclose(fp);
    s->maps.push_back(m);
    return (int32_t)s->maps.size() - 1;
}
void orc_texture_res(const orc_scene* s, int32_t idx, int32_t* xres, int32_t* yres) { *xres = s->maps[idx].x_res; *yres = s->maps[idx].y_res; }
void orc_texture_get(const orc_scene* s, int32_t idx, int64_t n, const double* uv, double* rgb)
{
    for (int64_t i = 0; i < n; i++) {
        Col c = tex_get(s->maps[idx], uv[2 * i], uv[2 * i + 1]);
        rgb[3 * i] = c.r; rgb[3 * i + 1] = c.g; rgb[3 * i + 2] = c.b;
    }
}

// BVH.cpp:98-106
void orc_build_bvh(orc_scene* s)
{
    s->nodes.clear();
    if (s->bvh_objs.empty()) return;
    Builder b;
    b.s = s;
    b.pbox.resize(s->prims.size());
    for (int o : s->bvh_objs) b.pbox[o] = prim_box(s->prims[o]);
    s->nodes.reserve(2 * s->bvh_objs.size() + 1);
    s->nodes.emplace_back();
    s->nodes[0].objs = s->bvh_objs;
    b.calc_box(0);
    b.split(0);
}

static void dump_node(const orc_scene* s, int n, std::vector<int32_t>& rec, std::vector<double>& boxes)
{
    const Node& nd = s->nodes[n];
    bool leaf = nd.child[0] < 0;
    rec.push_back(leaf ? 1 : 0);
    rec.push_back((int32_t)nd.objs.size());
    for (int o : nd.objs) rec.push_back(o);
    for (int i = 0; i < 3; i++) { boxes.push_back(nd.box.d[i][0]); boxes.push_back(nd.box.d[i][1]); }
    if (!leaf) { dump_node(s, nd.child[0], rec, boxes); dump_node(s, nd.child[1], rec, boxes); }
}
int64_t orc_bvh_dump(orc_scene* s, int64_t* n_boxes)
{
    s->dump_rec.clear(); s->dump_boxes.clear();
    dump_node(s, 0, s->dump_rec, s->dump_boxes);
    *n_boxes = (int64_t)s->dump_boxes.size() / 6;
    return (int64_t)s->dump_rec.size();
}
void orc_bvh_dump_copy(orc_scene* s, int32_t* rec, double* boxes)
{
    memcpy(rec, s->dump_rec.data(), s->dump_rec.size() * sizeof(int32_t));
    memcpy(boxes, s->dump_boxes.data(), s->dump_boxes.size() * sizeof(double));
}

void orc_shoot(const orc_scene* s, int64_t n, const double* org, const double* dir, double* dist, int32_t* ids)
{
    for (int64_t i = 0; i < n; i++) ids[i] = shoot(s, rd3(org + 3 * i), rd3(dir + 3 * i), dist[i]);
}
void orc_shoot_mt(const orc_scene* s, int64_t n, const double* org, const double* dir, double* dist, int32_t* ids, int32_t threads)
{
    if (threads < 1) threads = 1;
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([=]() {
            int64_t a = n * t / threads, b = n * (t + 1) / threads;
            for (int64_t i = a; i < b; i++) ids[i] = shoot(s, rd3(org + 3 * i), rd3(dir + 3 * i), dist[i]);
        });
    for (auto& x : th) x.join();
}
// ---------------------------------------------------------------- single-precision restatement
// The same traversal (BVH.cpp:5-24, 108-179) and primitive tests (Triangle.h:260-279, Sphere.h:44-90,
// Plane.h:26-39) with every quantity rounded to float and every operation done in float, on the
// SAME tree: boxes are the double boxes rounded to float, a triangle is stored as A, B - A, C - A
// (differences taken in double, then rounded), rays are the double rays rounded.  This is what a
// float build of the reference computes on float-rounded scene data; it checks the float
// instantiation of the device kernels bit for bit.
namespace {
struct PrimF { int kind; float a[3], e1[3], e2[3]; float radius; float pdistance; };
struct SceneF {
    std::vector<PrimF> prims;
    std::vector<float> box;   // 6 per node: [axis][lo, hi]
};

float box_intersect_f(const float* bx, const float* o, const float* d, float current_best)
{
    float t0 = 0.f, t1 = current_best;
    for (int i = 0; i < 3; i++) {
        float inv_ray = 1.f / d[i];
        float far_t = (bx[2 * i] - o[i]) * inv_ray;
        float near_t = (bx[2 * i + 1] - o[i]) * inv_ray;
        if (near_t > far_t) std::swap(near_t, far_t);
        t0 = near_t > t0 ? near_t : t0;
        t1 = far_t > t1 ? t1 : far_t;
        if (t0 > t1) return -1.f;
    }
    return t0;
}

float prim_intersect_f(const PrimF& p, const float* o, const float* d)
{
    if (p.kind == TRI) {
        const float *A = p.a, *e1 = p.e1, *e2 = p.e2;
        float qx = d[1] * e2[2] - d[2] * e2[1];
        float qy = d[2] * e2[0] - d[0] * e2[2];
        float qz = d[0] * e2[1] - d[1] * e2[0];
        float a = e1[0] * qx + e1[1] * qy + e1[2] * qz;
        if (std::fabs(a) < 0.00001f) return -1.f;
        float f = 1.f / a;
        float sx = o[0] - A[0], sy = o[1] - A[1], sz = o[2] - A[2];
        float u = f * (sx * qx + sy * qy + sz * qz);
        if (u < 0.f) return -1.f;
        float rx = sy * e1[2] - sz * e1[1];
        float ry = sz * e1[0] - sx * e1[2];
        float rz = sx * e1[1] - sy * e1[0];
        float v = f * (d[0] * rx + d[1] * ry + d[2] * rz);
        if (v < 0.f || u + v > 1.f) return -1.f;
        return f * (rx * e2[0] + ry * e2[1] + rz * e2[2]);
    }
    if (p.kind == SPHERE) {
        const float* c = p.a;
        float ox = o[0] + 0.1f * d[0], oy = o[1] + 0.1f * d[1], oz = o[2] + 0.1f * d[2];
        float b = (2.f * (ox - c[0]) * d[0]) + (2.f * (oy - c[1]) * d[1]) + (2.f * (oz - c[2]) * d[2]);
        float cc = (ox - c[0]) * (ox - c[0]) + (oy - c[1]) * (oy - c[1]) + (oz - c[2]) * (oz - c[2]) - p.radius * p.radius;
        float disc = b * b - 4.f * cc;
        if (disc > 0.f) {
            float root_1 = ((-1.f * b - std::sqrt(disc)) / 2.f) - 0.000001f;
            if (root_1 > 0.f) return root_1;
            return ((-1.f * b + std::sqrt(disc) - b) / 2.f) - 0.000001f;
        }
        return -1.f;
    }
    const float* n = p.a;
    float a = d[0] * n[0] + d[1] * n[1] + d[2] * n[2];
    if (a == 0.f) return -1.f;
    float px = o[0] + (-(n[0] * p.pdistance)), py = o[1] + (-(n[1] * p.pdistance)), pz = o[2] + (-(n[2] * p.pdistance));
    float b = n[0] * px + n[1] * py + n[2] * pz;
    return -1.f * b / a;
}

int shoot_node_f(const orc_scene* s, const SceneF& f, int n, const float* o, const float* d, float& distance)
{
    const Node& nd = s->nodes[n];
    if (box_intersect_f(&f.box[6 * (size_t)n], o, d, distance) < 0) return -1;
    int result = -1;
    for (int obj : nd.objs) {
        float t = prim_intersect_f(f.prims[obj], o, d);
        if (t >= 0 && t < distance) { result = obj; distance = std::min(distance, t); }
    }
    if (nd.child[0] < 0) return result;
    float td[2];
    for (int i = 0; i < 2; i++) td[i] = box_intersect_f(&f.box[6 * (size_t)nd.child[i]], o, d, distance);
    int first = td[0] < td[1] ? 0 : 1;
    int rc = shoot_node_f(s, f, nd.child[first], o, d, distance);
    if (rc >= 0) result = rc;
    rc = shoot_node_f(s, f, nd.child[1 - first], o, d, distance);
    if (rc >= 0) result = rc;
    return result;
}

SceneF make_scene_f(const orc_scene* s)
{
    SceneF f;
    f.prims.resize(s->prims.size());
    for (size_t i = 0; i < s->prims.size(); i++) {
        const Prim& p = s->prims[i];
        PrimF& q = f.prims[i];
        q.kind = p.kind;
        q.radius = q.pdistance = 0.f;
        for (int c = 0; c < 3; c++) q.a[c] = q.e1[c] = q.e2[c] = 0.f;
        if (p.kind == TRI) {
            V3 e1 = vsub(p.B, p.A), e2 = vsub(p.C, p.A);
            for (int c = 0; c < 3; c++) { q.a[c] = (float)p.A[c]; q.e1[c] = (float)e1[c]; q.e2[c] = (float)e2[c]; }
        } else if (p.kind == SPHERE) {
            for (int c = 0; c < 3; c++) q.a[c] = (float)p.center[c];
            q.radius = (float)p.radius;
        } else {
            for (int c = 0; c < 3; c++) q.a[c] = (float)p.pnormal[c];
            q.pdistance = (float)p.pdistance;
        }
    }
    f.box.resize(6 * s->nodes.size());
    for (size_t n = 0; n < s->nodes.size(); n++)
        for (int ax = 0; ax < 3; ax++) {
            f.box[6 * n + 2 * ax] = (float)s->nodes[n].box.d[ax][0];
            f.box[6 * n + 2 * ax + 1] = (float)s->nodes[n].box.d[ax][1];
        }
    return f;
}
}  // namespace

// dist is in/out like BVH::Shoot's `distance` (rounded to float on the way in, widened on the way out for hits)
void orc_shoot_f32(const orc_scene* s, int64_t n, const double* org, const double* dir, double* dist, int32_t* ids, int32_t threads)
{
    const SceneF f = make_scene_f(s);
    if (threads < 1) threads = 1;
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t]() {
            int64_t a = n * t / threads, b = n * (t + 1) / threads;
            for (int64_t i = a; i < b; i++) {
                float o[3] = {(float)org[3 * i], (float)org[3 * i + 1], (float)org[3 * i + 2]};
                float d[3] = {(float)dir[3 * i], (float)dir[3 * i + 1], (float)dir[3 * i + 2]};
                float distance = (float)dist[i];
                int result = -1;
                if (!(d[0] != d[0])) {
                    if (!s->nodes.empty() && !(box_intersect_f(&f.box[0], o, d, distance) < 0)) result = shoot_node_f(s, f, 0, o, d, distance);
                    for (int pl : s->planes) {
                        float tt = prim_intersect_f(f.prims[pl], o, d);
                        if (tt >= 0 && tt < distance) { result = pl; distance = std::min(distance, tt); }
                    }
                }
                ids[i] = result;
                if (result >= 0) dist[i] = (double)distance;
            }
        });
    for (auto& x : th) x.join();
}

void orc_shoot_stats(const orc_scene* s, int64_t n, const double* org, const double* dir, double max_dist, int64_t* counters)
{
    Counters c;
    for (int64_t i = 0; i < n; i++) { double d = max_dist; shoot(s, rd3(org + 3 * i), rd3(dir + 3 * i), d, &c); }
    counters[0] = c.slab; counters[1] = c.inner; counters[2] = c.prim; counters[3] = c.nodes;
}
// brute force with the traversal's accept rule in object-index order (property-test helper, cf. BVH.cpp:220-246)
void orc_brute_force(const orc_scene* s, int64_t n, const double* org, const double* dir, double* dist, int32_t* ids)
{
    for (int64_t i = 0; i < n; i++) {
        V3 o = rd3(org + 3 * i), d = rd3(dir + 3 * i);
        int best = -1;
        for (size_t k = 0; k < s->prims.size(); k++) {
            double t = prim_intersect(s->prims[k], o, d);
            if (t >= 0 && t < dist[i]) { dist[i] = t; best = (int)k; }
        }
        ids[i] = best;
    }
}

void orc_obj_intersect(const orc_scene* s, int32_t obj, int64_t n, const double* org, const double* dir, double* t)
{
    for (int64_t i = 0; i < n; i++) t[i] = prim_intersect(s->prims[obj], rd3(org + 3 * i), rd3(dir + 3 * i));
}
void orc_bbox_intersect(const double* box, int64_t n, const double* org, const double* dir, const double* best, double* t)
{
    Box b;
    for (int i = 0; i < 3; i++) { b.d[i][0] = box[2 * i]; b.d[i][1] = box[2 * i + 1]; }
    for (int64_t i = 0; i < n; i++) t[i] = box_intersect(b, rd3(org + 3 * i), rd3(dir + 3 * i), best[i]);
}
void orc_obj_bbox(const orc_scene* s, int32_t obj, double* box)
{
    Box b = prim_box(s->prims[obj]);
    for (int i = 0; i < 3; i++) { box[2 * i] = b.d[i][0]; box[2 * i + 1] = b.d[i][1]; }
}
void orc_obj_normal(const orc_scene* s, const int32_t* obj, int64_t n, const double* pos, double* out)
{
    for (int64_t i = 0; i < n; i++) wr3(out + 3 * i, prim_normal(s->prims[obj[i]], rd3(pos + 3 * i)));
}
void orc_obj_tangent(const orc_scene* s, const int32_t* obj, int64_t n, const double* pos, double* out)
{
    for (int64_t i = 0; i < n; i++) wr3(out + 3 * i, prim_tangent(s->prims[obj[i]], rd3(pos + 3 * i)));
}
void orc_obj_color(const orc_scene* s, const int32_t* obj, int64_t n, const double* pos, double* out)
{
    for (int64_t i = 0; i < n; i++) {
        Col c = prim_color(s, s->prims[obj[i]], rd3(pos + 3 * i));
        out[4 * i] = c.r; out[4 * i + 1] = c.g; out[4 * i + 2] = c.b; out[4 * i + 3] = c.s;
    }
}
void orc_reflect_dir(int64_t n, const double* nrm, const double* dir, double* out)
{
    for (int64_t i = 0; i < n; i++) wr3(out + 3 * i, reflect_dir(rd3(nrm + 3 * i), rd3(dir + 3 * i)));
}

void orc_seed(orc_scene* s, uint32_t seed) { s->gen.seed(seed); }

// main.cpp:697-718, single-threaded
int64_t orc_emit_photons(orc_scene* s, int64_t n_emit, const double* light_pos, const orc_params* P)
{
    std::uniform_real_distribution<> dis(-1, 1);
    Emit e{s, P};
    V3 lp = rd3(light_pos);
    for (int64_t pn = 0; pn < n_emit; pn++) {
        double x, y, z;
        do { x = dis(s->gen); y = dis(s->gen); z = dis(s->gen); } while (x * x + y * y + z * z > 1);
        V3 d = vnorm(V3(x, y, z));
        e.emit(lp, d, Col(1.0, 1.0, 1.0, 0.0), 0);
    }
    return (int64_t)s->photons[0].size();
}
int64_t orc_photon_count(const orc_scene* s, int32_t which) { return (int64_t)s->photons[which].size(); }
void orc_photons_get(const orc_scene* s, int32_t which, double* out)
{
    const auto& v = s->photons[which];
    for (size_t i = 0; i < v.size(); i++) {
        wr3(out + 10 * i, v[i].pos);
        out[10 * i + 3] = v[i].power.r; out[10 * i + 4] = v[i].power.g; out[10 * i + 5] = v[i].power.b;
        wr3(out + 10 * i + 6, v[i].dir);
        out[10 * i + 9] = v[i].power.s;
    }
}
void orc_photons_set(orc_scene* s, int32_t which, int64_t n, const double* in)
{
    auto& v = s->photons[which];
    v.clear();
    for (int64_t i = 0; i < n; i++)
        v.push_back(Photon{rd3(in + 10 * i), Col(in[10 * i + 3], in[10 * i + 4], in[10 * i + 5], in[10 * i + 9]), rd3(in + 10 * i + 6)});
}

// Kdtree.h:59-66, Kdtree.cpp:54-61; main.cpp:724-728.  The photon vector itself is reordered by
// the top-level sort in the reference (the tree sorts the caller's vector in place); indices here
// always refer to the order at the time of orc_photons_get *after* the build.
void orc_build_kdtree(orc_scene* s)
{
    for (int w = 0; w < 2; w++) {
        KTree& t = s->kd[w];
        t.nodes.clear();
        t.root = -1;
        auto& ph = s->photons[w];
        if (ph.size() <= 8) continue;  // Kdtree.cpp:13 assert; main.cpp:726 builds the volume tree only if > 3
        std::vector<int> idx(ph.size());
        for (size_t i = 0; i < idx.size(); i++) idx[i] = (int)i;
        t.root = kd_create(ph, t, idx);
        // idx now holds the root-level sorted order: apply it to the photon array and remap the tree
        std::vector<int> inv(ph.size());
        std::vector<Photon> sorted(ph.size());
        for (size_t i = 0; i < idx.size(); i++) { sorted[i] = ph[idx[i]]; inv[idx[i]] = (int)i; }
        ph.swap(sorted);
        for (auto& nd : t.nodes) {
            nd.self = inv[nd.self];
            for (auto& p : nd.left) p = inv[p];
            for (auto& p : nd.right) p = inv[p];
        }
    }
}

static void dump_kd(const KTree& t, int n, std::vector<int32_t>& rec)
{
    const KNode& nd = t.nodes[n];
    rec.push_back(nd.axis); rec.push_back(nd.self);
    rec.push_back((int32_t)nd.left.size()); rec.push_back((int32_t)nd.right.size());
    rec.push_back(nd.lb >= 0 ? 1 : 0); rec.push_back(nd.rb >= 0 ? 1 : 0);
    for (int p : nd.left) rec.push_back(p);
    for (int p : nd.right) rec.push_back(p);
    if (nd.lb >= 0) dump_kd(t, nd.lb, rec);
    if (nd.rb >= 0) dump_kd(t, nd.rb, rec);
}
int64_t orc_kd_dump(orc_scene* s)
{
    s->dump_rec.clear();
    dump_kd(s->kd[0], s->kd[0].root, s->dump_rec);
    return (int64_t)s->dump_rec.size();
}
void orc_kd_dump_copy(orc_scene* s, int32_t* rec) { memcpy(rec, s->dump_rec.data(), s->dump_rec.size() * sizeof(int32_t)); }

void orc_knn(const orc_scene* s, int32_t which, int32_t k, int64_t n, const double* centers, int32_t* out_idx,
             int32_t* out_cnt, double* out_d2)
{
    for (int64_t q = 0; q < n; q++) {
        V3 c = rd3(centers + 3 * q);
        std::vector<int> r = kd_knn(s, which, k, c, 0.28);
        out_cnt[q] = (int32_t)r.size();
        for (int j = 0; j < k; j++) {
            if (j < (int)r.size()) { out_idx[q * k + j] = r[j]; out_d2[q * k + j] = vsqd(s->photons[which][r[j]].pos, c); }
            else { out_idx[q * k + j] = -1; out_d2[q * k + j] = -1; }
        }
    }
}

void orc_get_color_at(const orc_scene* s, const orc_params* P, int64_t n, const double* pos, const double* dir,
                      const int32_t* obj, int32_t depth, double* rgb)
{
    std::mt19937 g = s->gen;
    ShadeCtx cx{s, P, 0, &g};
    for (int64_t i = 0; i < n; i++) {
        Col c = color_at(cx, rd3(pos + 3 * i), rd3(dir + 3 * i), obj[i], depth);
        rgb[3 * i] = c.r; rgb[3 * i + 1] = c.g; rgb[3 * i + 2] = c.b;
    }
}

void orc_primary_rays(int32_t w, int32_t h, const double* cam, double* org, double* dir)
{
    Cam c = make_cam(cam);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            size_t i = (size_t)y * w + x;
            wr3(org + 3 * i, c.pos);
            wr3(dir + 3 * i, cam_ray_dir(c, w, h, 1, x, y, 0));
        }
}
// one ray per (pixel, aax): dir = w*h*aadepth*3
void orc_primary_rays_aa(int32_t w, int32_t h, int32_t aadepth, const double* cam, double* org, double* dir)
{
    Cam c = make_cam(cam);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int a = 0; a < aadepth; a++) {
                size_t i = ((size_t)y * w + x) * aadepth + a;
                wr3(org + 3 * i, c.pos);
                wr3(dir + 3 * i, cam_ray_dir(c, w, h, aadepth, x, y, a));
            }
}

// main.cpp:735-861 (DOF_SAMPLE 1).  Threads take interleaved rows instead of column strips (the
// reference's strip split asserts width % threads == 0); pixels are independent, so the image is
// the same.
void orc_render(const orc_scene* s, const orc_params* P, int32_t w, int32_t h, int32_t aadepth, const double* cam,
                int32_t row_begin, int32_t row_end, int32_t threads, double* rgb, int64_t* ray_count)
{
    Cam c = make_cam(cam);
    if (threads < 1) threads = 1;
    std::vector<int64_t> counts(threads, 0);
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t]() {
            std::mt19937 g = s->gen;
            ShadeCtx cx{s, P, 0, &g};
            std::vector<double> tr(aadepth * aadepth), tg(aadepth * aadepth), tb(aadepth * aadepth);
            for (int y = row_begin + t; y < row_end; y += threads)
                for (int x = 0; x < w; x++) {
                    for (int aax = 0; aax < aadepth; aax++)
                        for (int aay = 0; aay < aadepth; aay++) {
                            int aa_index = aay * aadepth + aax;
                            V3 d = cam_ray_dir(c, w, h, aadepth, x, y, aax);
                            double dof[3] = {0, 0, 0};
                            const int n_dof = P->dof_samples > 1 ? P->dof_samples : 1;
                            // main.cpp:792-826
                            V3 focal = vadd(c.pos, vmul(d, P->focal_len));
                            V3 worldUp(0, 1, 0);
                            V3 camUp = vadd(worldUp, vneg(vmul(d, vdot(worldUp, d))));
                            V3 camSide = vcross(camUp, d);
                            std::uniform_real_distribution<> dis_rand_dof(-P->aperture, P->aperture);
                            for (int ds = 0; ds < n_dof; ds++) {
                                V3 org = c.pos, dj = d;
                                if (n_dof != 1) {
                                    double jx, jy;
                                    if (P->counter_rng) {
                                        uint32_t px = (uint32_t)((long long)y * w + x), draw = (uint32_t)(((aay * aadepth + aax) * n_dof + ds) * 2);
                                        jx = -P->aperture + 2 * P->aperture * counter_uniform(px, draw, P->rng_seed);
                                        jy = -P->aperture + 2 * P->aperture * counter_uniform(px, draw + 1u, P->rng_seed);
                                    } else {
                                        jx = dis_rand_dof(g); jy = dis_rand_dof(g);
                                    }
                                    org = vadd(vadd(c.pos, vmul(camUp, jx)), vmul(camSide, jy));
                                    dj = vnorm(vadd(focal, vneg(org)));
                                }
                                double dist = 1e9;
                                int io = shoot(s, org, dj, dist);
                                cx.rays++;
                                if (io >= 0 && dist > P->accuracy) {
                                    V3 ipos = vadd(org, vmul(d, dist));          // main.cpp:821: the un-jittered direction
                                    Col col = color_at(cx, ipos, d, io, 1);
                                    dof[0] += col.r; dof[1] += col.g; dof[2] += col.b;
                                }
                            }
                            dof[0] /= n_dof; dof[1] /= n_dof; dof[2] /= n_dof;
                            tr[aa_index] = dof[0] / 1; tg[aa_index] = dof[1] / 1; tb[aa_index] = dof[2] / 1;
                        }
                    double R = 0, G = 0, B = 0;
                    for (int i = 0; i < aadepth * aadepth; i++) R += tr[i];
                    for (int i = 0; i < aadepth * aadepth; i++) G += tg[i];
                    for (int i = 0; i < aadepth * aadepth; i++) B += tb[i];
                    size_t px = (size_t)y * w + x;
                    rgb[3 * px] = R / (aadepth * aadepth);
                    rgb[3 * px + 1] = G / (aadepth * aadepth);
                    rgb[3 * px + 2] = B / (aadepth * aadepth);
                }
            counts[t] = cx.rays;
        });
    for (auto& x : th) x.join();
    int64_t total = 0;
    for (auto v : counts) total += v;
    if (ray_count) *ray_count = total;
}

// IO.cpp:55-65: byte = (unsigned char) floor(v*255), written B,G,R
void orc_quantize(int64_t n_px, const double* rgb, uint8_t* bgr)
{
    for (int64_t i = 0; i < n_px; i++) {
        bgr[3 * i] = (unsigned char)std::floor(rgb[3 * i + 2] * 255);
        bgr[3 * i + 1] = (unsigned char)std::floor(rgb[3 * i + 1] * 255);
        bgr[3 * i + 2] = (unsigned char)std::floor(rgb[3 * i] * 255);
    }
}

// IO.cpp:5-68: 54-byte header with the reference's field placement (size field counts 4 bytes
// per pixel, biSizeImage written one byte early at offset 21, ppm at 25 and 29), no row padding.
int32_t orc_save_bmp(const char* path, int32_t w, int32_t h, int32_t dpi, const double* rgb)
{
    int k = w * h, sz = 4 * k, filesize = 54 + sz;
    int ppm = dpi * 39;
    unsigned char fh[14] = {'B', 'M', 0, 0, 0, 0, 0, 0, 0, 0, 54, 0, 0, 0};
    unsigned char ih[40] = {40, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 24, 0};
    for (int i = 0; i < 4; i++) {
        fh[2 + i] = (unsigned char)(filesize >> (8 * i));
        ih[4 + i] = (unsigned char)(w >> (8 * i));
        ih[8 + i] = (unsigned char)(h >> (8 * i));
    }
    for (int i = 0; i < 4; i++) ih[21 + i] = (unsigned char)(sz >> (8 * i));
    for (int i = 0; i < 4; i++) ih[25 + i] = (unsigned char)(ppm >> (8 * i));
    for (int i = 0; i < 4; i++) ih[29 + i] = (unsigned char)(ppm >> (8 * i));
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    fwrite(fh, 1, 14, f);
    fwrite(ih, 1, 40, f);
    std::vector<uint8_t> bgr((size_t)k * 3);
    orc_quantize(k, rgb, bgr.data());
    fwrite(bgr.data(), 1, bgr.size(), f);
    fclose(f);
    return 0;
}

}  // extern "C"
