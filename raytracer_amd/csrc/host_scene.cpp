// Host scene assembly + BVH build/flatten.  See host_scene.h.
#include "host_scene.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <thread>
#include <array>
#include <map>
#include <memory>
#include <mutex>
#include <string>

#include "rt_math.h"

namespace rth {

static void cp3(double* d, const double* s) { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; }

// ---------------------------------------------------------------- scene assembly
int Scene::add_triangle(const double* A, const double* C, const double* B, const double* rgbs, float ior)
{
    // Triangle(pointA, pointC, pointB, ...) stores A = 1st, C = 2nd, B = 3rd argument (Triangle.h:137-140)
    Object o{};
    o.sh.kind = rtf::PRIM_TRI;
    cp3(o.sh.A, A); cp3(o.sh.B, B); cp3(o.sh.C, C);
    rtm::D3 n = rtm::tri_normal(o.sh);          // NormalA/B/C start as the face normal (Triangle.h:141-143)
    double nn[3] = {n.x, n.y, n.z};
    cp3(o.sh.NA, nn); cp3(o.sh.NB, nn); cp3(o.sh.NC, nn);
    for (int i = 0; i < 4; i++) o.sh.color[i] = rgbs[i];
    o.sh.ior = ior;
    o.sh.map = -1;
    objects.push_back(o);
    bvh.built = false;
    return (int)objects.size() - 1;
}

int Scene::set_normals(int obj, const double* na, const double* nc, const double* nb)
{
    if (obj < 0 || obj >= (int)objects.size() || objects[obj].sh.kind != rtf::PRIM_TRI) return -1;
    rtf::ObjShade& s = objects[obj].sh;       // setNormals(a, c, b), Triangle.h:150-156
    cp3(s.NA, na); cp3(s.NB, nb); cp3(s.NC, nc);
    s.has_normal = 1;
    return 0;
}

int Scene::set_texture(int obj, const double* ta, const double* tc, const double* tb, int use, int map)
{
    if (obj < 0 || obj >= (int)objects.size() || objects[obj].sh.kind != rtf::PRIM_TRI) return -1;
    if (use && (map < 0 || map >= (int)textures.size())) return -1;
    rtf::ObjShade& s = objects[obj].sh;       // setTexture(A, C, B, use, map), Triangle.h:106-113
    s.tA[0] = ta[0]; s.tA[1] = ta[1]; s.tB[0] = tb[0]; s.tB[1] = tb[1]; s.tC[0] = tc[0]; s.tC[1] = tc[1];
    s.has_texture = use ? 1 : 0;
    s.map = map;
    objects[obj].has_uv = true;
    return 0;
}

int Scene::add_sphere(const double* c, double radius, const double* rgbs, float ior)
{
    Object o{};
    o.sh.kind = rtf::PRIM_SPHERE;
    cp3(o.sh.A, c);
    o.sh.B[0] = radius;
    for (int i = 0; i < 4; i++) o.sh.color[i] = rgbs[i];
    o.sh.ior = ior;
    o.sh.map = -1;
    objects.push_back(o);
    bvh.built = false;
    return (int)objects.size() - 1;
}

int Scene::add_plane(const double* n, double distance, const double* rgbs)
{
    Object o{};
    o.sh.kind = rtf::PRIM_PLANE;
    cp3(o.sh.A, n);
    o.sh.B[0] = distance;
    for (int i = 0; i < 4; i++) o.sh.color[i] = rgbs[i];
    o.sh.ior = 1;                             // Object::getRefraIdx default, Object.h:28
    o.sh.map = -1;
    objects.push_back(o);
    bvh.built = false;
    return (int)objects.size() - 1;
}

namespace {
struct Corners { double A[3], B[3], C[3], D[3], E[3], F[3]; };
// the six derived corners of IO.cpp:80-86 / :118-124
Corners corners(const double* c1, const double* c2)
{
    Corners k;
    double a[3] = {c2[0], c1[1], c1[2]}, b[3] = {c2[0], c1[1], c2[2]}, c[3] = {c1[0], c1[1], c2[2]};
    double d[3] = {c2[0], c2[1], c1[2]}, e[3] = {c1[0], c2[1], c1[2]}, f[3] = {c1[0], c2[1], c2[2]};
    cp3(k.A, a); cp3(k.B, b); cp3(k.C, c); cp3(k.D, d); cp3(k.E, e); cp3(k.F, f);
    return k;
}
}  // namespace

void Scene::add_cornell_box(const double* c1, const double* c2)
{
    // IO.cpp:108-154: left (green), far (white), right (red), top, bottom; the camera side stays open
    Corners k = corners(c1, c2);
    const double red[4] = {1, 0.25, 0.25, 0}, green[4] = {0.25, 1, 0.25, 0}, white[4] = {1.0, 1.0, 1.0, 0};
    const double* tri[10][4] = {
        {k.D, k.A, c1, green}, {c1, k.E, k.D, green}, {c2, k.B, k.A, white}, {k.A, k.D, c2, white}, {k.F, k.C, k.B, red},
        {k.B, c2, k.F, red},   {k.D, k.E, k.F, white}, {k.F, c2, k.D, white}, {c1, k.A, k.B, white}, {k.B, k.C, c1, white}};
    for (auto& t : tri) add_triangle(t[0], t[1], t[2], t[3], 20);
}

void Scene::add_cube(const double* c1, const double* c2, const double* rgbs)
{
    // IO.cpp:70-106: left, far, right, front, top, bottom
    Corners k = corners(c1, c2);
    const double* tri[12][3] = {{k.D, k.A, c1}, {c1, k.E, k.D}, {c2, k.B, k.A}, {k.A, k.D, c2}, {k.F, k.C, k.B}, {k.B, c2, k.F},
                                {k.E, c1, k.C}, {k.C, k.F, k.E}, {k.D, k.E, k.F}, {k.F, c2, k.D}, {c1, k.A, k.B}, {k.B, k.C, c1}};
    for (auto& t : tri) add_triangle(t[0], t[1], t[2], rgbs, 2);
}

// ---------------------------------------------------------------- OBJ loader (IO.h:22-125)
namespace {
// one face corner "v", "v/vt", "v//vn" or "v/vt/vn" -> zero-based indices, -1 when absent
void parse_corner(const std::string& tok, int& v, int& vt, int& vn)
{
    v = vt = vn = -1;
    const char* s = tok.c_str();
    char* end = nullptr;
    long a = strtol(s, &end, 10);
    if (end != s) v = (int)a - 1;
    if (*end != '/') return;
    s = end + 1;
    if (*s != '/') {
        long b = strtol(s, &end, 10);
        if (end != s) vt = (int)b - 1;
    } else {
        end = const_cast<char*>(s);
    }
    if (*end != '/') return;
    s = end + 1;
    long c = strtol(s, &end, 10);
    if (end != s) vn = (int)c - 1;
}
}  // namespace

namespace {
// What a mesh file says, statement by statement, in the file's order (a face may only name vertices that came before it,
// and a truncated face ends the file: IO.h:75-119).  Parsed once per file: a scene of instanced meshes loads the same
// file thousands of times with different offsets, and reading and splitting 45 KB of text was nearly all of each load.
struct ObjStatement {
    enum Kind : int { V, VN, VT, F, END } kind;
    double a[3];              // V / VN: x y z (as read, before the instance's offset); VT: u v
    int v[3], vt[3], vn[3];   // F: zero-based corners, -1 when absent
};
struct ParsedObj {
    std::vector<ObjStatement> st;
    long long size = -1, mtime_ns = -1;      // of the file when it was parsed
};
std::mutex g_obj_mu;
std::map<std::string, std::shared_ptr<const ParsedObj>> g_obj_cache;

std::shared_ptr<const ParsedObj> parse_obj(const char* path)
{
    struct stat sb;
    if (stat(path, &sb) != 0) return nullptr;
    const long long size = (long long)sb.st_size, mtime_ns = (long long)sb.st_mtim.tv_sec * 1000000000ll + sb.st_mtim.tv_nsec;
    {
        std::lock_guard<std::mutex> lock(g_obj_mu);
        auto it = g_obj_cache.find(path);
        if (it != g_obj_cache.end() && it->second->size == size && it->second->mtime_ns == mtime_ns) return it->second;
    }
    FILE* fp = fopen(path, "rb");
    if (!fp) return nullptr;
    std::string text;
    char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, fp)) > 0) text.append(buf, got);
    fclose(fp);
    // whitespace-separated words, exactly like the reference's `input >> head` stream (IO.h:75-119):
    // comment text is just a run of unknown words
    std::vector<std::string> words;
    {
        size_t i = 0, n = text.size();
        while (i < n) {
            while (i < n && isspace((unsigned char)text[i])) i++;
            size_t j = i;
            while (j < n && !isspace((unsigned char)text[j])) j++;
            if (j > i) words.emplace_back(text, i, j - i);
            i = j;
        }
    }
    auto parsed = std::make_shared<ParsedObj>();
    parsed->size = size; parsed->mtime_ns = mtime_ns;
    auto num = [&](size_t k) { return k < words.size() ? strtod(words[k].c_str(), nullptr) : 0.0; };
    for (size_t w = 0; w < words.size(); w++) {
        const std::string& head = words[w];
        ObjStatement e{};
        if (head == "v" || head == "vn") {
            e.kind = head == "v" ? ObjStatement::V : ObjStatement::VN;
            e.a[0] = num(w + 1); e.a[1] = num(w + 2); e.a[2] = num(w + 3);
            w += 3;
        } else if (head == "vt") {
            e.kind = ObjStatement::VT;
            e.a[0] = num(w + 1); e.a[1] = num(w + 2);
            w += 2;
        } else if (head == "f") {
            if (w + 3 >= words.size()) { e.kind = ObjStatement::END; parsed->st.push_back(e); break; }   // truncated face at end of file
            e.kind = ObjStatement::F;
            for (int k = 0; k < 3; k++) parse_corner(words[w + 1 + k], e.v[k], e.vt[k], e.vn[k]);
            w += 3;
        } else {
            continue;
        }
        parsed->st.push_back(e);
    }
    std::lock_guard<std::mutex> lock(g_obj_mu);
    if (g_obj_cache.size() >= 64) g_obj_cache.clear();      // (a handful of meshes in any real use; never grows without bound)
    g_obj_cache[path] = parsed;
    return parsed;
}
}  // namespace

int Scene::load_obj(const char* path, const double* rgbs, float ior, float xo, float yo, float zo, int use_texture, int map)
{
    const std::shared_ptr<const ParsedObj> parsed = parse_obj(path);
    if (!parsed) return -1;                  // the reference returns false silently (IO.h:72-73)
    std::vector<std::array<double, 3>> points, normals;
    std::vector<std::array<double, 2>> uvs;
    for (const ObjStatement& e : parsed->st) {
        if (e.kind == ObjStatement::V) {
            points.push_back({e.a[0] + xo, e.a[1] + yo, e.a[2] + zo});
        } else if (e.kind == ObjStatement::VN) {
            normals.push_back({e.a[0], e.a[1], e.a[2]});
        } else if (e.kind == ObjStatement::VT) {
            uvs.push_back({e.a[0], e.a[1]});
        } else if (e.kind == ObjStatement::END) {
            break;
        } else {
            const int *v = e.v, *vt = e.vt, *vn = e.vn;
            for (int k = 0; k < 3; k++)
                if (v[k] < 0 || v[k] >= (int)points.size()) return -2;   // reference: vector::at throws
            // new Triangle(points[v0], points[v1], points[v2], ...) = ctor order (A, C, B), IO.h:103
            int id = add_triangle(points[v[0]].data(), points[v[1]].data(), points[v[2]].data(), rgbs, ior);
            bool has_n = true, has_t = true;
            for (int k = 0; k < 3; k++) {
                has_n = has_n && vn[k] >= 0 && vn[k] < (int)normals.size();
                has_t = has_t && vt[k] >= 0 && vt[k] < (int)uvs.size();
            }
            if (has_n) set_normals(id, normals[vn[0]].data(), normals[vn[1]].data(), normals[vn[2]].data());   // IO.h:104
            if (has_t) {                                                                                         // IO.h:105
                rtf::ObjShade& s = objects[id].sh;
                s.tA[0] = uvs[vt[0]][0]; s.tA[1] = uvs[vt[0]][1];
                s.tC[0] = uvs[vt[1]][0]; s.tC[1] = uvs[vt[1]][1];
                s.tB[0] = uvs[vt[2]][0]; s.tB[1] = uvs[vt[2]][1];
                s.has_texture = (use_texture && map >= 0 && map < (int)textures.size()) ? 1 : 0;
                s.map = map;
                objects[id].has_uv = true;
            }
        }
    }
    return 0;
}

// ---------------------------------------------------------------- textures (TextureMap.h:20-61)
int Scene::add_texture_bmp(const char* path)
{
    FILE* fp = fopen(path, "rb");
    if (!fp) return -1;
    unsigned char header[54];
    if (fread(header, 1, 54, fp) != 54) { fclose(fp); return -1; }
    Texture t;
    // width (bytes 18-21) is the inner, column count y_res; height (22-25) the row count x_res
    t.y_res = header[18] | header[19] << 8 | header[20] << 16 | header[21] << 24;
    t.x_res = header[22] | header[23] << 8 | header[24] << 16 | header[25] << 24;
    if (t.x_res <= 0 || t.y_res <= 0) { fclose(fp); return -1; }
    size_t n = (size_t)t.x_res * t.y_res;
    // The header is not trusted: the texels it announces must be in the file (the reference would read past
    // the end into an uninitialised buffer, TextureMap.h:45-53; a corrupt size must not become a huge allocation).
    long data_at = ftell(fp);
    if (data_at < 0 || fseek(fp, 0, SEEK_END) != 0) { fclose(fp); return -1; }
    long file_end = ftell(fp);
    if (file_end < data_at || (unsigned long long)(file_end - data_at) < (unsigned long long)n * 3ull || fseek(fp, data_at, SEEK_SET) != 0) { fclose(fp); return -1; }
    std::vector<unsigned char> raw(n * 3, 0);
    size_t got = fread(raw.data(), 1, raw.size(), fp);   // no row padding is skipped, as in the reference
    fclose(fp);
    if (got != raw.size()) return -1;
    t.rgb.resize(n * 3);
    for (size_t i = 0; i < n; i++) {
        t.rgb[3 * i] = raw[3 * i + 2] / 255.0;       // file order is B, G, R (TextureMap.h:50-53)
        t.rgb[3 * i + 1] = raw[3 * i + 1] / 255.0;
        t.rgb[3 * i + 2] = raw[3 * i] / 255.0;
    }
    textures.push_back(std::move(t));
    return (int)textures.size() - 1;
}

int Scene::add_texture_rgb(int x_res, int y_res, const double* rgb)
{
    if (x_res <= 0 || y_res <= 0 || !rgb) return -1;
    Texture t;
    t.x_res = x_res; t.y_res = y_res;
    t.rgb.assign(rgb, rgb + (size_t)x_res * y_res * 3);
    textures.push_back(std::move(t));
    return (int)textures.size() - 1;
}

// ---------------------------------------------------------------- BVH build (BVH.cpp:26-106)
Box object_box(const rtf::ObjShade& o)
{
    Box b;
    if (o.kind == rtf::PRIM_TRI) {            // Triangle::getBBox, Triangle.h:115-125
        for (int i = 0; i < 3; i++) {
            b.lo[i] = std::min(o.A[i], std::min(o.B[i], o.C[i]));
            b.hi[i] = std::max(o.A[i], std::max(o.B[i], o.C[i]));
        }
    } else {                                  // Sphere::getBBox, Sphere.h:17-27
        for (int i = 0; i < 3; i++) { b.lo[i] = o.A[i] - o.B[0]; b.hi[i] = o.A[i] + o.B[0]; }
    }
    return b;
}

namespace {

inline Box unite(const Box& a, const Box& b)   // BBox::Union, BVH.cpp:31-40
{
    Box r;
    for (int i = 0; i < 3; i++) { r.lo[i] = std::min(a.lo[i], b.lo[i]); r.hi[i] = std::max(a.hi[i], b.hi[i]); }
    return r;
}
// BBox::Size, BVH.h:33-38 -- evaluation order kept (it decides which split wins on near-ties)
inline double extent_sum(const Box& b) { return b.hi[0] - b.lo[0] + b.hi[1] - b.lo[1] + b.hi[2] - b.lo[2]; }

}  // namespace

rtf::Prim<double> make_prim(const rtf::ObjShade& o, uint32_t id)
{
    rtf::Prim<double> p{};
    p.id = id;
    p.kind = o.kind;
    if (o.kind == rtf::PRIM_TRI) {
        for (int i = 0; i < 3; i++) { p.a[i] = o.A[i]; p.e1[i] = o.B[i] - o.A[i]; p.e2[i] = o.C[i] - o.A[i]; }
    } else {
        for (int i = 0; i < 3; i++) p.a[i] = o.A[i];
        p.e1[0] = o.B[0];
    }
    return p;
}

namespace {

struct Builder {
    const std::vector<Object>& objs;
    const std::vector<Box>& box;                        // per object
    const std::vector<std::array<double, 3>>& centre;   // BBox::Center, BVH.h:23-28
    FlatBVH& out;

    static uint32_t shift_ref(uint32_t ref, uint32_t node_off, uint32_t prim_off)
    {
        if (ref & rtf::LEAF_BIT) return ref + prim_off;     // the first-primitive field sits in the low bits
        return ref + node_off;
    }

    // append a subtree built on its own (local numbering) behind what `out` already holds
    uint32_t absorb(const FlatBVH& sub, uint32_t sub_ref)
    {
        const uint32_t node_off = (uint32_t)out.nodes.size(), prim_off = (uint32_t)out.prims.size();
        for (rtf::Node<double> nd : sub.nodes) {
            nd.c0 = shift_ref(nd.c0, node_off, prim_off);
            nd.c1 = shift_ref(nd.c1, node_off, prim_off);
            out.nodes.push_back(nd);
        }
        out.prims.insert(out.prims.end(), sub.prims.begin(), sub.prims.end());
        out.n_leaves += sub.n_leaves;
        out.depth = std::max(out.depth, sub.depth);
        return shift_ref(sub_ref, node_off, prim_off);
    }

    // returns the child reference and its box for a node holding `ids` (in the node's object order)
    uint32_t build(std::vector<int>& ids, Box& node_box, int level, int par_levels = 0)
    {
        if (level > out.depth) out.depth = level;
        if (ids.size() < 16) {                   // leaf: BVH.h:45, BVH.cpp:59
            node_box = box[ids[0]];              // calculateBBox, BVH.cpp:44-49
            for (int id : ids) node_box = unite(node_box, box[id]);
            uint32_t first = (uint32_t)out.prims.size();
            for (int id : ids) out.prims.push_back(make_prim(objs[id].sh, (uint32_t)id));
            out.n_leaves++;
            return rtf::make_leaf(first, (uint32_t)ids.size());
        }
        size_t n = ids.size();
        // axis with the widest spread of box centres, first maximum wins (BVH.cpp:61-67)
        double span[3];
        for (int ax = 0; ax < 3; ax++) {
            double mn = centre[ids[0]][ax], mx = mn;
            for (int id : ids) { mn = std::min(mn, centre[id][ax]); mx = std::max(mx, centre[id][ax]); }
            span[ax] = mx - mn;
        }
        int axis = int(std::max_element(span, span + 3) - span);
        // the same std::sort call on the same initial order with an equivalent comparator gives the
        // reference's permutation, ties included (BVH.cpp:68)
        std::sort(ids.begin(), ids.end(), [&](int a, int b) { return centre[a][axis] < centre[b][axis]; });
        // prefix / suffix boxes and the sum-of-extents cost over the middle third (BVH.cpp:69-85)
        std::vector<Box> left(n), right(n);
        left[0] = box[ids[0]];
        right[0] = box[ids[n - 1]];
        for (size_t i = 1; i < n; i++) {
            left[i] = unite(left[i - 1], box[ids[i]]);
            right[i] = unite(right[i - 1], box[ids[n - 1 - i]]);
        }
        int bad_zone = std::max(1, (int)n / 3);
        size_t split_pos = 0;
        double best = 1e9;                      // entries outside the zone keep 1e9; first minimum wins
        for (size_t i = 0; i < n; i++) {
            double q = 1e9;
            if (!((int)i < bad_zone || i > n - 1 - bad_zone)) q = extent_sum(left[i]) + extent_sum(right[n - 1 - i]);
            if (q < best) { best = q; split_pos = i; }
        }
        std::vector<Box>().swap(left);
        std::vector<Box>().swap(right);
        std::vector<int> l(ids.begin(), ids.begin() + split_pos + 1), r(ids.begin() + split_pos + 1, ids.end());
        std::vector<int>().swap(ids);
        uint32_t me = (uint32_t)out.nodes.size();
        out.nodes.emplace_back();
        Box b0, b1;
        uint32_t c0, c1;
        if (par_levels > 0 && n >= 65536) {
            // the two subtrees are independent: build them side by side, each into its own arrays, and
            // append them in the order the serial build produces (node pre-order, leaf order)
            FlatBVH lsub, rsub;
            uint32_t lref = 0, rref = 0;
            Builder lb{objs, box, centre, lsub}, rb{objs, box, centre, rsub};
            std::thread other([&] { rref = rb.build(r, b1, level + 1, par_levels - 1); });
            lref = lb.build(l, b0, level + 1, par_levels - 1);
            other.join();
            c0 = absorb(lsub, lref);
            c1 = absorb(rsub, rref);
        } else {
            c0 = build(l, b0, level + 1);
            c1 = build(r, b1, level + 1);
        }
        rtf::Node<double>& nd = out.nodes[me];
        for (int i = 0; i < 3; i++) { nd.lo0[i] = b0.lo[i]; nd.hi0[i] = b0.hi[i]; nd.lo1[i] = b1.lo[i]; nd.hi1[i] = b1.hi[i]; }
        nd.c0 = c0;
        nd.c1 = c1;
        node_box = unite(b0, b1);               // calculateBBox for an inner node, BVH.cpp:52-53
        return me;
    }
};

}  // namespace

void Scene::build_bvh()
{
    bvh = FlatBVH{};
    std::vector<int> ids;
    std::vector<Box> box(objects.size());
    std::vector<std::array<double, 3>> centre(objects.size());
    for (size_t i = 0; i < objects.size(); i++) {
        const rtf::ObjShade& o = objects[i].sh;
        if (o.kind == rtf::PRIM_PLANE) {
            bvh.planes.push_back(make_prim(o, (uint32_t)i));
            continue;
        }
        box[i] = object_box(o);
        for (int ax = 0; ax < 3; ax++) centre[i][ax] = (box[i].hi[ax] + box[i].lo[ax]) / 2;
        ids.push_back((int)i);
    }
    Builder b{objects, box, centre, bvh};
    // large scenes: the top 4 levels fork (up to 16 subtrees side by side); the tree and its layout
    // are those of the serial build
    if (!ids.empty()) bvh.root_ref = b.build(ids, bvh.root_box, 0, 4);
    bvh.built = true;
}

void Scene::bvh_dump(std::vector<int32_t>& rec, std::vector<double>& boxes) const
{
    rec.clear();
    boxes.clear();
    if (bvh.root_ref == rtf::NO_NODE) return;
    struct Item { uint32_t ref; Box box; };
    std::vector<Item> stack{{bvh.root_ref, bvh.root_box}};
    while (!stack.empty()) {
        Item it = stack.back();
        stack.pop_back();
        for (int i = 0; i < 3; i++) { boxes.push_back(it.box.lo[i]); boxes.push_back(it.box.hi[i]); }
        if (it.ref & rtf::LEAF_BIT) {
            uint32_t first = it.ref & rtf::LEAF_FIRST_MASK, cnt = (it.ref >> rtf::LEAF_COUNT_SHIFT) & 15;
            rec.push_back(1);
            rec.push_back((int32_t)cnt);
            for (uint32_t k = 0; k < cnt; k++) rec.push_back((int32_t)bvh.prims[first + k].id);
        } else {
            const rtf::Node<double>& nd = bvh.nodes[it.ref];
            rec.push_back(0);
            rec.push_back(0);
            Item a{nd.c0, {}}, c{nd.c1, {}};
            for (int i = 0; i < 3; i++) { a.box.lo[i] = nd.lo0[i]; a.box.hi[i] = nd.hi0[i]; c.box.lo[i] = nd.lo1[i]; c.box.hi[i] = nd.hi1[i]; }
            stack.push_back(c);   // pre-order: child 0 first
            stack.push_back(a);
        }
    }
}

std::vector<rtf::Node<float>> to_f32(const rtf::Node<double>* n, size_t count)
{
    std::vector<rtf::Node<float>> o(count);
    for (size_t k = 0; k < count; k++) {
        for (int i = 0; i < 3; i++) {
            o[k].lo0[i] = (float)n[k].lo0[i]; o[k].hi0[i] = (float)n[k].hi0[i];
            o[k].lo1[i] = (float)n[k].lo1[i]; o[k].hi1[i] = (float)n[k].hi1[i];
        }
        o[k].c0 = n[k].c0;
        o[k].c1 = n[k].c1;
    }
    return o;
}

std::vector<rtf::Prim<float>> to_f32(const rtf::Prim<double>* p, size_t count)
{
    std::vector<rtf::Prim<float>> o(count);
    for (size_t k = 0; k < count; k++) {
        for (int i = 0; i < 3; i++) { o[k].a[i] = (float)p[k].a[i]; o[k].e1[i] = (float)p[k].e1[i]; o[k].e2[i] = (float)p[k].e2[i]; }
        o[k].id = p[k].id;
        o[k].kind = p[k].kind;
    }
    return o;
}

// ---------------------------------------------------------------- flat image of a built scene
namespace {
inline size_t up64(size_t v) { return (v + 63) & ~(size_t)63; }

// section offsets for the counts in h (fills off_* and total_bytes)
void flat_layout(FlatImageHeader& h)
{
    size_t at = sizeof(FlatImageHeader);
    auto take = [&](size_t bytes) { size_t o = at; at = up64(at + bytes); return (uint64_t)o; };
    h.off_objects = take(h.n_objects * sizeof(rtf::ObjShade));
    h.off_has_uv = take(h.n_objects);
    h.off_nodes = take(h.n_nodes * sizeof(rtf::Node<double>));
    h.off_prims = take(h.n_prims * sizeof(rtf::Prim<double>));
    h.off_planes = take(h.n_planes * sizeof(rtf::Prim<double>));
    h.off_texdesc = take(h.n_textures * sizeof(rtf::TexDesc));
    h.off_texels = take(h.n_texel_doubles * sizeof(double));
    h.total_bytes = at;
}

FlatImageHeader flat_header(const Scene& s)
{
    FlatImageHeader h{};
    h.magic = FLAT_MAGIC;
    h.version = FLAT_VERSION;
    h.sizeof_header = sizeof(FlatImageHeader);
    h.sizeof_objshade = sizeof(rtf::ObjShade);
    h.sizeof_node = sizeof(rtf::Node<double>);
    h.sizeof_prim = sizeof(rtf::Prim<double>);
    h.n_objects = s.objects.size();
    h.n_nodes = s.bvh.nodes.size();
    h.n_prims = s.bvh.prims.size();
    h.n_planes = s.bvh.planes.size();
    h.n_textures = s.textures.size();
    for (const Texture& t : s.textures) h.n_texel_doubles += t.rgb.size();
    for (int i = 0; i < 3; i++) { h.root_lo[i] = s.bvh.root_box.lo[i]; h.root_hi[i] = s.bvh.root_box.hi[i]; }
    h.root_ref = s.bvh.root_ref;
    h.depth = s.bvh.depth;
    h.n_leaves = s.bvh.n_leaves;
    flat_layout(h);
    return h;
}
}  // namespace

size_t flat_image_bytes(const Scene& s) { return (size_t)flat_header(s).total_bytes; }

void flat_image_write(const Scene& s, void* out)
{
    const FlatImageHeader h = flat_header(s);
    char* b = (char*)out;
    memset(b, 0, (size_t)h.total_bytes);      // (padding between sections is defined: two images of one scene compare equal)
    memcpy(b, &h, sizeof h);
    rtf::ObjShade* o = (rtf::ObjShade*)(b + h.off_objects);
    uint8_t* uv = (uint8_t*)(b + h.off_has_uv);
    for (size_t i = 0; i < s.objects.size(); i++) { memcpy(&o[i], &s.objects[i].sh, sizeof(rtf::ObjShade)); uv[i] = s.objects[i].has_uv ? 1 : 0; }
    if (h.n_nodes) memcpy(b + h.off_nodes, s.bvh.nodes.data(), h.n_nodes * sizeof(rtf::Node<double>));
    if (h.n_prims) memcpy(b + h.off_prims, s.bvh.prims.data(), h.n_prims * sizeof(rtf::Prim<double>));
    if (h.n_planes) memcpy(b + h.off_planes, s.bvh.planes.data(), h.n_planes * sizeof(rtf::Prim<double>));
    rtf::TexDesc* td = (rtf::TexDesc*)(b + h.off_texdesc);
    double* tx = (double*)(b + h.off_texels);
    int64_t at = 0;
    for (size_t i = 0; i < s.textures.size(); i++) {
        td[i] = rtf::TexDesc{s.textures[i].x_res, s.textures[i].y_res, at};
        if (!s.textures[i].rgb.empty()) memcpy(tx + at, s.textures[i].rgb.data(), s.textures[i].rgb.size() * sizeof(double));
        at += (int64_t)s.textures[i].rgb.size();
    }
}

int flat_image_open(const void* buf, size_t bytes, FlatImageView* v, std::string* err)
{
    auto bad = [&](const char* why) { if (err) *err = why; return -1; };
    if (!buf || bytes < sizeof(FlatImageHeader)) return bad("flat scene image: shorter than its header");
    if ((uintptr_t)buf % 64 != 0) return bad("flat scene image: the buffer is not 64-byte aligned");
    FlatImageHeader h;
    memcpy(&h, buf, sizeof h);
    if (h.magic != FLAT_MAGIC) return bad("flat scene image: wrong magic word");
    if (h.version != FLAT_VERSION) return bad("flat scene image: unknown version");
    if (h.sizeof_header != sizeof(FlatImageHeader) || h.sizeof_objshade != sizeof(rtf::ObjShade) || h.sizeof_node != sizeof(rtf::Node<double>) ||
        h.sizeof_prim != sizeof(rtf::Prim<double>))
        return bad("flat scene image: record sizes differ from this build's");
    const uint64_t lim = 1ull << 40;
    if (h.n_objects >= (1ull << 31) || h.n_nodes >= (1ull << 31) || h.n_prims >= (1ull << rtf::LEAF_COUNT_SHIFT) || h.n_planes >= (1ull << 31) ||
        h.n_textures >= (1ull << 20) || h.n_texel_doubles >= lim)
        return bad("flat scene image: a count is out of range");
    FlatImageHeader want = h;
    flat_layout(want);
    if (want.total_bytes != h.total_bytes || want.off_objects != h.off_objects || want.off_has_uv != h.off_has_uv || want.off_nodes != h.off_nodes ||
        want.off_prims != h.off_prims || want.off_planes != h.off_planes || want.off_texdesc != h.off_texdesc || want.off_texels != h.off_texels)
        return bad("flat scene image: section offsets do not match the counts");
    if (h.total_bytes > bytes) return bad("flat scene image: truncated");
    const char* b = (const char*)buf;
    FlatImageView w;
    w.h = (const FlatImageHeader*)b;
    w.objects = (const rtf::ObjShade*)(b + h.off_objects);
    w.has_uv = (const uint8_t*)(b + h.off_has_uv);
    w.nodes = (const rtf::Node<double>*)(b + h.off_nodes);
    w.prims = (const rtf::Prim<double>*)(b + h.off_prims);
    w.planes = (const rtf::Prim<double>*)(b + h.off_planes);
    w.texdesc = (const rtf::TexDesc*)(b + h.off_texdesc);
    w.texels = (const double*)(b + h.off_texels);
    // textures, then the objects that name them
    for (uint64_t i = 0; i < h.n_textures; i++) {
        const rtf::TexDesc& t = w.texdesc[i];
        if (t.x_res <= 0 || t.y_res <= 0 || t.offset < 0 || (uint64_t)t.offset + 3ull * (uint64_t)t.x_res * (uint64_t)t.y_res > h.n_texel_doubles)
            return bad("flat scene image: a texture lies outside the texel pool");
    }
    for (uint64_t i = 0; i < h.n_objects; i++) {
        const rtf::ObjShade& o = w.objects[i];
        if (o.kind > rtf::PRIM_PLANE) return bad("flat scene image: unknown object kind");
        if (o.has_texture && (o.map < 0 || (uint64_t)o.map >= h.n_textures)) return bad("flat scene image: an object names a texture that is not there");
    }
    for (uint64_t i = 0; i < h.n_prims; i++)
        if (w.prims[i].id >= h.n_objects || w.prims[i].kind > rtf::PRIM_SPHERE) return bad("flat scene image: a primitive record names no object");
    for (uint64_t i = 0; i < h.n_planes; i++)
        if (w.planes[i].id >= h.n_objects || w.planes[i].kind != rtf::PRIM_PLANE) return bad("flat scene image: a plane record names no object");
    // the tree: leaves inside the primitive array, inner children behind their parent (pre-order, hence no cycle)
    auto leaf_ok = [&](uint32_t ref) {
        const uint64_t first = ref & rtf::LEAF_FIRST_MASK, cnt = (ref >> rtf::LEAF_COUNT_SHIFT) & 15;
        return cnt >= 1 && first + cnt <= h.n_prims;
    };
    if (h.root_ref == rtf::NO_NODE) {
        if (h.n_nodes || h.n_prims) return bad("flat scene image: nodes without a root");
    } else if (h.root_ref & rtf::LEAF_BIT) {
        if (h.n_nodes || !leaf_ok(h.root_ref)) return bad("flat scene image: bad root leaf");
        w.n_leaves = 1;
    } else {
        if (h.root_ref != 0 || h.n_nodes == 0) return bad("flat scene image: the root is not node 0");
        std::vector<int32_t> level(h.n_nodes, -1);
        level[0] = 0;
        for (uint64_t i = 0; i < h.n_nodes; i++) {
            if (level[i] < 0) return bad("flat scene image: a node no parent refers to");
            for (uint32_t c : {w.nodes[i].c0, w.nodes[i].c1}) {
                if (c & rtf::LEAF_BIT) {
                    if (!leaf_ok(c)) return bad("flat scene image: a leaf lies outside the primitive array");
                    w.n_leaves++;
                    w.depth = std::max(w.depth, level[i] + 1);
                } else {
                    if (c <= i || c >= h.n_nodes || level[c] >= 0) return bad("flat scene image: a child reference breaks the pre-order");
                    level[c] = level[i] + 1;
                    w.depth = std::max(w.depth, level[c]);
                }
            }
        }
    }
    *v = w;
    return 0;
}

void flat_image_read(const FlatImageView& v, Scene& s)
{
    const FlatImageHeader& h = *v.h;
    s = Scene{};
    s.objects.resize(h.n_objects);
    for (uint64_t i = 0; i < h.n_objects; i++) { s.objects[i].sh = v.objects[i]; s.objects[i].has_uv = v.has_uv[i] != 0; }
    s.textures.resize(h.n_textures);
    for (uint64_t i = 0; i < h.n_textures; i++) {
        Texture& t = s.textures[i];
        t.x_res = v.texdesc[i].x_res; t.y_res = v.texdesc[i].y_res;
        const double* from = v.texels + v.texdesc[i].offset;
        t.rgb.assign(from, from + (size_t)3 * t.x_res * t.y_res);
    }
    s.bvh.nodes.assign(v.nodes, v.nodes + h.n_nodes);
    s.bvh.prims.assign(v.prims, v.prims + h.n_prims);
    s.bvh.planes.assign(v.planes, v.planes + h.n_planes);
    for (int i = 0; i < 3; i++) { s.bvh.root_box.lo[i] = h.root_lo[i]; s.bvh.root_box.hi[i] = h.root_hi[i]; }
    s.bvh.root_ref = h.root_ref;
    s.bvh.depth = v.depth;
    s.bvh.n_leaves = v.n_leaves;
    s.bvh.built = true;
}

// ---------------------------------------------------------------- BMP writer (IO.cpp:5-68)
void quantize_bgr8(int64_t n_px, const double* rgb, uint8_t* bgr)
{
    for (int64_t i = 0; i < n_px; i++) {       // (unsigned char) floor(v * 255), B G R (IO.cpp:58-62)
        bgr[3 * i] = (unsigned char)floor(rgb[3 * i + 2] * 255);
        bgr[3 * i + 1] = (unsigned char)floor(rgb[3 * i + 1] * 255);
        bgr[3 * i + 2] = (unsigned char)floor(rgb[3 * i] * 255);
    }
}

int save_bmp_bgr8(const char* path, int w, int h, int dpi, const uint8_t* bgr)
{
    // The reference's header, field for field: bfSize counts 4 bytes per pixel, the image-size
    // field starts at info-header byte 21 (one early), pixels-per-metre at 25 and 29; 24 bpp rows
    // are written bottom-up without padding.
    uint32_t pixels = (uint32_t)w * (uint32_t)h;
    uint32_t img = 4u * pixels, file_size = 54u + img, ppm = (uint32_t)(dpi * 39);
    unsigned char head[54];
    memset(head, 0, sizeof head);
    head[0] = 'B'; head[1] = 'M'; head[10] = 54;
    unsigned char* info = head + 14;
    info[0] = 40; info[12] = 1; info[14] = 24;
    auto put = [](unsigned char* p, uint32_t v) { for (int i = 0; i < 4; i++) p[i] = (unsigned char)(v >> (8 * i)); };
    put(head + 2, file_size);
    put(info + 4, (uint32_t)w);
    put(info + 8, (uint32_t)h);
    put(info + 21, img);
    put(info + 25, ppm);
    put(info + 29, ppm);
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    bool ok = fwrite(head, 1, 54, f) == 54 && fwrite(bgr, 1, (size_t)pixels * 3, f) == (size_t)pixels * 3;
    fclose(f);
    return ok ? 0 : -1;
}

}  // namespace rth
