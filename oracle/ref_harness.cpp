// TEST INFRASTRUCTURE ONLY -- never linked into, loaded by or shipped with the product library.
//
// C-ABI harness around the *untouched* reference renderer.  It is compiled only where the
// reference sources are mounted (REF_DIR, default /root/reference) by oracle/Makefile into
// oracle/_ref/librt_ref.so, and is used by oracle/make_golden.py to (a) validate the CPU
// restatement in rt_oracle.cpp and (b) generate the golden vectors committed under tests/golden/.
//
// The reference is one translation unit around main.cpp (globals bvh, kdtree, photons,
// scene_objects, the TU-local std::mt19937 `gen`), so this file includes main.cpp textually with
// `main` renamed, and links BVH.cpp / IO.cpp / Kdtree.cpp unmodified.  Nothing is copied: the
// include paths point into REF_DIR.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdint>
#include <iostream>
#include <fstream>
#include <sstream>
#include <vector>
#include <string>
#include <limits>
#include <queue>
#include <mutex>
#include <thread>
#include <chrono>
#include <random>
#include <ctime>
#include <ratio>
#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include <time.h>
#include <assert.h>

// Expose BVH::root / KDTree::Root / Triangle members for topology dumps.  Access specifiers do not
// change the object layout under the Itanium ABI, so the separately compiled reference TUs agree.
#define private public
#define main reference_main
#include "main.cpp"
#undef main
#undef private

namespace {

struct TexHolder { TextureMap* map; };
std::vector<TextureMap*> g_maps;

void dump_node(const BVHNode* n, std::vector<int32_t>& rec, std::vector<double>& boxes)
{
    // pre-order: [is_leaf, count, ids...] ; bbox as 6 doubles (xmin,xmax,ymin,ymax,zmin,zmax)
    bool leaf = (n->child[0] == nullptr);
    rec.push_back(leaf ? 1 : 0);
    rec.push_back((int32_t)n->objects.size());
    for (auto* o : n->objects) {
        auto it = std::find(scene_objects.begin(), scene_objects.end(), o);
        rec.push_back((int32_t)(it - scene_objects.begin()));
    }
    for (int i = 0; i < 3; i++) { boxes.push_back(n->bbox.data[i][0]); boxes.push_back(n->bbox.data[i][1]); }
    if (!leaf) { dump_node(n->child[0], rec, boxes); dump_node(n->child[1], rec, boxes); }
}

std::vector<int32_t> g_rec;
std::vector<double> g_boxes;

int obj_index(Object* o)
{
    if (!o) return -1;
    // scene_objects is small enough for the golden sizes; cache a map for big batches
    static std::vector<Object*> cached;
    static std::vector<std::pair<Object*, int>> sorted;
    if (cached != scene_objects) {
        cached = scene_objects;
        sorted.clear();
        for (size_t i = 0; i < cached.size(); i++) sorted.push_back({cached[i], (int)i});
        std::sort(sorted.begin(), sorted.end());
    }
    auto it = std::lower_bound(sorted.begin(), sorted.end(), std::make_pair(o, -1));
    if (it == sorted.end() || it->first != o) return -2;
    return it->second;
}

}  // namespace

extern "C" {

void ref_seed(unsigned s) { gen.seed(s); }

void ref_scene_reset()
{
    scene_objects.clear();
    photons.clear();
    volumePhotons.clear();
    bvh = nullptr;
    kdtree = nullptr;
    volumeKDTree = nullptr;
}

int ref_scene_size() { return (int)scene_objects.size(); }

// ObjReader::ReadContent (IO.h:61).  tex_path may be NULL (empty map as in main.cpp:661).
int ref_scene_load_obj(const char* path, double r, double g, double b, double special, float ior,
                       float xo, float yo, float zo, int use_texture, const char* tex_path)
{
    TextureMap* map = new TextureMap();
    if (tex_path) map->TextureMapRead(tex_path);
    g_maps.push_back(map);
    ObjReader reader(path, Color(r, g, b, special), ior, xo, yo, zo, map, use_texture != 0);
    return reader.ReadContent(&scene_objects) ? 1 : 0;
}

void ref_scene_add_cornell(const double* c1, const double* c2)
{
    makeCornellBox(scene_objects, Vect(c1[0], c1[1], c1[2]), Vect(c2[0], c2[1], c2[2]));
}

void ref_scene_add_cube(const double* c1, const double* c2, const double* rgbs)
{
    makeCube(scene_objects, Vect(c1[0], c1[1], c1[2]), Vect(c2[0], c2[1], c2[2]),
             Color(rgbs[0], rgbs[1], rgbs[2], rgbs[3]));
}

// Triangle(A, C, B, color, ior) -- note the reference ctor argument order (Triangle.h:137).
void ref_scene_add_triangle(const double* A, const double* C, const double* B, const double* rgbs, float ior)
{
    scene_objects.push_back(new Triangle(Vect(A[0], A[1], A[2]), Vect(C[0], C[1], C[2]), Vect(B[0], B[1], B[2]),
                                         Color(rgbs[0], rgbs[1], rgbs[2], rgbs[3]), ior));
}

// setNormals(a, c, b) on the last added object, which must be a Triangle (Triangle.h:150).
void ref_scene_last_set_normals(const double* na, const double* nc, const double* nb)
{
    Triangle* t = dynamic_cast<Triangle*>(scene_objects.back());
    t->setNormals(Vect(na[0], na[1], na[2]), Vect(nc[0], nc[1], nc[2]), Vect(nb[0], nb[1], nb[2]));
}

// setTexture(tA, tC, tB, use, map) on the last added Triangle (Triangle.h:106); map index from ref_texture_read.
void ref_scene_last_set_texture(const double* ta, const double* tc, const double* tb, int use, int map_idx)
{
    Triangle* t = dynamic_cast<Triangle*>(scene_objects.back());
    t->setTexture(MyTexture(ta[0], ta[1]), MyTexture(tc[0], tc[1]), MyTexture(tb[0], tb[1]), use != 0, g_maps.at(map_idx));
}

void ref_scene_add_sphere(const double* c, double radius, const double* rgbs, float ior)
{
    scene_objects.push_back(new Sphere(Vect(c[0], c[1], c[2]), radius, Color(rgbs[0], rgbs[1], rgbs[2], rgbs[3]), ior));
}

void ref_build_bvh() { bvh = new BVH(scene_objects); }

// BVH::Shoot (BVH.cpp:108): dist is in/out, ids out (-1 = miss)
void ref_shoot(int64_t n, const double* org, const double* dir, double* dist, int32_t* ids)
{
    for (int64_t i = 0; i < n; i++) {
        Ray r(Vect(org[3 * i], org[3 * i + 1], org[3 * i + 2]), Vect(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]));
        Object* o = bvh->Shoot(r, dist[i]);
        ids[i] = obj_index(o);
    }
}

// findIntersection of one object for many rays (Triangle.h:260 / Sphere.h:44)
void ref_obj_intersect(int obj, int64_t n, const double* org, const double* dir, double* t)
{
    Object* o = scene_objects.at(obj);
    for (int64_t i = 0; i < n; i++) {
        Ray r(Vect(org[3 * i], org[3 * i + 1], org[3 * i + 2]), Vect(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]));
        t[i] = o->findIntersection(r);
    }
}

// BBox::Intersect (BVH.cpp:5) with box = xmin,xmax,ymin,ymax,zmin,zmax
void ref_bbox_intersect(const double* box, int64_t n, const double* org, const double* dir, const double* best, double* t)
{
    BBox b;
    for (int i = 0; i < 3; i++) { b.data[i][0] = box[2 * i]; b.data[i][1] = box[2 * i + 1]; }
    for (int64_t i = 0; i < n; i++) {
        Ray r(Vect(org[3 * i], org[3 * i + 1], org[3 * i + 2]), Vect(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]));
        t[i] = b.Intersect(r, best[i]);
    }
}

void ref_obj_bbox(int obj, double* box)
{
    BBox b = scene_objects.at(obj)->getBBox();
    for (int i = 0; i < 3; i++) { box[2 * i] = b.data[i][0]; box[2 * i + 1] = b.data[i][1]; }
}

// getNormalAt / getColor(pos) / getTangentAt (Triangle.h:29,235,242; Sphere.h:31,33,40)
void ref_obj_normal(const int32_t* obj, int64_t n, const double* pos, double* out)
{
    for (int64_t i = 0; i < n; i++) {
        Vect v = scene_objects.at(obj[i])->getNormalAt(Vect(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]));
        out[3 * i] = v.getVectX(); out[3 * i + 1] = v.getVectY(); out[3 * i + 2] = v.getVectZ();
    }
}
void ref_obj_tangent(const int32_t* obj, int64_t n, const double* pos, double* out)
{
    for (int64_t i = 0; i < n; i++) {
        Vect v = scene_objects.at(obj[i])->getTangentAt(Vect(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]));
        out[3 * i] = v.getVectX(); out[3 * i + 1] = v.getVectY(); out[3 * i + 2] = v.getVectZ();
    }
}
void ref_obj_color(const int32_t* obj, int64_t n, const double* pos, double* out)
{
    for (int64_t i = 0; i < n; i++) {
        Color c = scene_objects.at(obj[i])->getColor(Vect(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]));
        out[4 * i] = c.getColorRed(); out[4 * i + 1] = c.getColorGreen(); out[4 * i + 2] = c.getColorBlue();
        out[4 * i + 3] = c.getColorSpecial();
    }
}
float ref_obj_ior(int obj) { return scene_objects.at(obj)->getRefraIdx(); }

// TextureMap (TextureMap.h:20,132)
int ref_texture_read(const char* path)
{
    TextureMap* m = new TextureMap();
    m->x_res = m->y_res = 0;
    m->TextureMapRead(path);
    g_maps.push_back(m);
    return (int)g_maps.size() - 1;
}
void ref_texture_res(int idx, int* xres, int* yres) { *xres = g_maps.at(idx)->x_res; *yres = g_maps.at(idx)->y_res; }
void ref_texture_get(int idx, int64_t n, const double* uv, double* rgb)
{
    for (int64_t i = 0; i < n; i++) {
        Color c = g_maps.at(idx)->GetColor(uv[2 * i], uv[2 * i + 1]);
        rgb[3 * i] = c.getColorRed(); rgb[3 * i + 1] = c.getColorGreen(); rgb[3 * i + 2] = c.getColorBlue();
    }
}

// BVH topology, pre-order.  Returns number of int32 words; call ref_bvh_dump_copy afterwards.
int64_t ref_bvh_dump(int64_t* n_boxes)
{
    g_rec.clear(); g_boxes.clear();
    dump_node(&bvh->root, g_rec, g_boxes);
    *n_boxes = (int64_t)g_boxes.size() / 6;
    return (int64_t)g_rec.size();
}
void ref_bvh_dump_copy(int32_t* rec, double* boxes)
{
    memcpy(rec, g_rec.data(), g_rec.size() * sizeof(int32_t));
    memcpy(boxes, g_boxes.data(), g_boxes.size() * sizeof(double));
}

// Photon emission loop of main.cpp:697-718, run single-threaded so the std::mt19937 stream is
// deterministic.  photonEmission itself (main.cpp:396) is the reference's.
int64_t ref_emit_photons(int64_t n_emit, const double* light_pos)
{
    std::uniform_real_distribution<> dis(-1, 1);
    Color lightColor(1.0, 1.0, 1.0, 0.0);
    Vect light_position(light_pos[0], light_pos[1], light_pos[2]);
    double accuracy = 0.000001, ambientLight = 0;
    for (int64_t pn = 0; pn < n_emit; pn++) {
        double x, y, z;
        do { x = dis(gen); y = dis(gen); z = dis(gen); } while (x * x + y * y + z * z > 1);
        Vect d(x, y, z);
        d = d.normalize();
        Ray photon_ray(light_position, d);
        photonEmission(photon_ray, d, scene_objects, accuracy, ambientLight, lightColor, 0);
    }
    return (int64_t)photons.size();
}
int64_t ref_volume_photon_count() { return (int64_t)volumePhotons.size(); }

// Photon records as 10 doubles: pos(3) power rgb(3) dir(3) special-of-power(1)
void ref_photons_get(int which, double* out)
{
    auto& v = which ? volumePhotons : photons;
    for (size_t i = 0; i < v.size(); i++) {
        Photon* p = v[i];
        out[10 * i + 0] = p->position.getVectX(); out[10 * i + 1] = p->position.getVectY(); out[10 * i + 2] = p->position.getVectZ();
        out[10 * i + 3] = p->power.getColorRed(); out[10 * i + 4] = p->power.getColorGreen(); out[10 * i + 5] = p->power.getColorBlue();
        out[10 * i + 6] = p->dir.getVectX(); out[10 * i + 7] = p->dir.getVectY(); out[10 * i + 8] = p->dir.getVectZ();
        out[10 * i + 9] = p->power.getColorSpecial();
    }
}
// Replace the photon set (lets the restatement and the reference run on an identical set).
void ref_photons_set(int which, int64_t n, const double* in)
{
    auto& v = which ? volumePhotons : photons;
    v.clear();
    for (int64_t i = 0; i < n; i++) {
        v.push_back(new Photon(Vect(in[10 * i], in[10 * i + 1], in[10 * i + 2]),
                               Color(in[10 * i + 3], in[10 * i + 4], in[10 * i + 5], in[10 * i + 9]), 0, 0,
                               Vect(in[10 * i + 6], in[10 * i + 7], in[10 * i + 8])));
    }
}

void ref_build_kdtree()
{
    kdtree = new KDTree(photons);  // sorts `photons` in place at top level (Kdtree.cpp:24)
    if (volumePhotons.size() > 3) volumeKDTree = new KDTree(volumePhotons);
}

// KDTree::findKNN (Kdtree.cpp:135).  out_idx: n x k (index into the *current* order of `photons`,
// -1 padded), out_cnt: n, out_d2: n x k squared distances (nearest first).
void ref_knn(int which, int k, int64_t n, const double* centers, int32_t* out_idx, int32_t* out_cnt, double* out_d2)
{
    KDTree* t = which ? volumeKDTree : kdtree;
    auto& v = which ? volumePhotons : photons;
    std::vector<std::pair<Photon*, int>> sorted;
    for (size_t i = 0; i < v.size(); i++) sorted.push_back({v[i], (int)i});
    std::sort(sorted.begin(), sorted.end());
    for (int64_t q = 0; q < n; q++) {
        Vect c(centers[3 * q], centers[3 * q + 1], centers[3 * q + 2]);
        std::vector<Photon*> r = t->findKNN(k, c);
        out_cnt[q] = (int32_t)r.size();
        for (int j = 0; j < k; j++) {
            if (j < (int)r.size()) {
                auto it = std::lower_bound(sorted.begin(), sorted.end(), std::make_pair(r[j], -1));
                out_idx[q * k + j] = it->second;
                out_d2[q * k + j] = r[j]->position.sqrDist(c);
            } else {
                out_idx[q * k + j] = -1;
                out_d2[q * k + j] = -1;
            }
        }
    }
}

// kd-tree topology, pre-order: per node [axis, self_idx, n_left_bucket, n_right_bucket, has_left, has_right]
static void dump_kd(KDNode* n, std::vector<int32_t>& rec, const std::vector<std::pair<Photon*, int>>& sorted)
{
    auto idx = [&](Photon* p) { return std::lower_bound(sorted.begin(), sorted.end(), std::make_pair(p, -1))->second; };
    rec.push_back(n->axis);
    rec.push_back(idx(n->self));
    rec.push_back((int32_t)n->left.size());
    rec.push_back((int32_t)n->right.size());
    rec.push_back(n->left_branch ? 1 : 0);
    rec.push_back(n->right_branch ? 1 : 0);
    for (auto* p : n->left) rec.push_back(idx(p));
    for (auto* p : n->right) rec.push_back(idx(p));
    if (n->left_branch) dump_kd(n->left_branch, rec, sorted);
    if (n->right_branch) dump_kd(n->right_branch, rec, sorted);
}
int64_t ref_kd_dump()
{
    std::vector<std::pair<Photon*, int>> sorted;
    for (size_t i = 0; i < photons.size(); i++) sorted.push_back({photons[i], (int)i});
    std::sort(sorted.begin(), sorted.end());
    g_rec.clear();
    dump_kd(kdtree->Root, g_rec, sorted);
    return (int64_t)g_rec.size();
}
void ref_kd_dump_copy(int32_t* rec) { memcpy(rec, g_rec.data(), g_rec.size() * sizeof(int32_t)); }

// getColorAt (main.cpp:158) -- HEAD semantics, PHOTONMUM/PHOTONUSE/TRACING_DEPTH as compiled in.
void ref_get_color_at(int64_t n, const double* pos, const double* dir, const int32_t* obj, int depth, double* rgb)
{
    for (int64_t i = 0; i < n; i++) {
        Color c = getColorAt(Vect(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]), Vect(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]),
                             scene_objects.at(obj[i]), 0.000001, 0, depth);
        rgb[3 * i] = c.getColorRed(); rgb[3 * i + 1] = c.getColorGreen(); rgb[3 * i + 2] = c.getColorBlue();
    }
}

// computeReflectionDir (main.cpp:125)
void ref_reflect_dir(int64_t n, const double* nrm, const double* dir, double* out)
{
    for (int64_t i = 0; i < n; i++) {
        Vect v = computeReflectionDir(Vect(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]), Vect(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]));
        out[3 * i] = v.getVectX(); out[3 * i + 1] = v.getVectY(); out[3 * i + 2] = v.getVectZ();
    }
}

// saveBmp (IO.cpp:5)
void ref_save_bmp(const char* path, int w, int h, int dpi, const double* rgb)
{
    RGBType* px = new RGBType[(size_t)w * h];
    for (size_t i = 0; i < (size_t)w * h; i++) { px[i].r = rgb[3 * i]; px[i].g = rgb[3 * i + 1]; px[i].b = rgb[3 * i + 2]; }
    saveBmp(path, w, h, dpi, px);
    delete[] px;
}

// Camera rays of main.cpp:753-790 (aadepth == 1, DOF off) written with the reference's own Vect
// operations in the reference's order; cam = pos(3), look_at(3).  Pixel index = y*w + x.
void ref_primary_rays(int w, int h, const double* cam, double* org, double* dir)
{
    Vect Y(0, 1, 0);
    Vect camPos(cam[0], cam[1], cam[2]);
    Vect look_at(cam[3], cam[4], cam[5]);
    Vect diff_btw(camPos.getVectX() - look_at.getVectX(), camPos.getVectY() - look_at.getVectY(), camPos.getVectZ() - look_at.getVectZ());
    Vect camDir = diff_btw.negtive().normalize();
    Vect camRight = Y.crossProduct(camDir).normalize();
    Vect camDown = camRight.crossProduct(camDir);
    int width = w, height = h;
    double aspectRatio = (double)width / (double)height;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double xamnt, yamnt;
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
            Vect d = camDir.vectAdd(camRight.vectMult(xamnt - 0.5).vectAdd(camDown.vectMult(yamnt - 0.5))).normalize();
            size_t i = (size_t)y * w + x;
            org[3 * i] = camPos.getVectX(); org[3 * i + 1] = camPos.getVectY(); org[3 * i + 2] = camPos.getVectZ();
            dir[3 * i] = d.getVectX(); dir[3 * i + 1] = d.getVectY(); dir[3 * i + 2] = d.getVectZ();
        }
}

}  // extern "C"
