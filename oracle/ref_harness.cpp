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

}  // extern "C"

namespace {

// The shade loop with the recursion the reference evidently means: a reflection / refraction ray that
// hits something is continued on the object it HIT.  HEAD passes the object the ray LEFT (`hit_obj`,
// main.cpp:198,249) and tests `hit_obj != nullptr` instead of the new hit (main.cpp:193,245), which
// shades every mirror and glass surface black.  Everything here is the reference's own code --
// bvh->Shoot, Object::getColor / getNormalAt / getRefraIdx, computeReflectionDir, computeGlossyRay,
// Color and Vect arithmetic, and getColorAt itself for surfaces that spawn no ray (special <= 0: that
// branch of main.cpp:260-390 does not recurse, so HEAD's code already is the intended code) -- driven
// in the order of main.cpp:169-258 with those two tokens changed.
Color shade_intended(Vect pos_in, Vect hit_dir, Object* hit_obj, double accuracy, double ambientLight, int depth)
{
    Color final_color(0, 0, 0, 0);
    if (depth > TRACING_DEPTH) return final_color.clip();                       // main.cpp:161
    Vect hit_pos = pos_in.vectAdd(hit_dir.vectMult(accuracy));                  // main.cpp:162
    Color hit_color = hit_obj->getColor(hit_pos);                               // main.cpp:164
    double colorSpecial = hit_color.getColorSpecial();
    if (!(colorSpecial > 0)) return getColorAt(pos_in, hit_dir, hit_obj, accuracy, ambientLight, depth);
    Vect hit_normal = hit_obj->getNormalAt(hit_pos);                            // main.cpp:165

    Vect reflection_direction = computeReflectionDir(hit_normal, hit_dir);      // main.cpp:170
    const double frac = colorSpecial * 100.0 - floor(colorSpecial * 100.0);
    const double sample = (frac - 0.7654 < 0.0001 && frac - 0.7653 > 0) ? GLOSSY_SAMPLE : 1;   // main.cpp:177-179
    Color reflect_sum(0, 0, 0, 0);
    for (int i = 0; i < sample; i++) {                                          // main.cpp:181-206
        Vect world_dir = sample != 1 ? computeGlossyRay(reflection_direction, hit_normal, 0.2) : reflection_direction;
        Ray ray(hit_pos, world_dir);
        double intersect_distance = 1e9;
        Object* intersect_obj = bvh->Shoot(ray, intersect_distance);
        if (intersect_obj != nullptr && intersect_distance > accuracy) {        // main.cpp:193, on the new hit
            Vect next_pos = hit_pos.vectAdd(reflection_direction.vectMult(intersect_distance));
            Color c = shade_intended(next_pos, reflection_direction, intersect_obj, accuracy, ambientLight, depth + 1);   // main.cpp:198
            if (colorSpecial > 1) reflect_sum = reflect_sum.colorAdd(c.colorScalar(2 - hit_color.getColorSpecial()));
            else reflect_sum = reflect_sum.colorAdd(c);
        }
    }
    reflect_sum = reflect_sum.colorScalar(1.0 / sample);                        // main.cpp:207-208
    final_color = final_color.colorAdd(reflect_sum);

    if (colorSpecial > 1) {                                                     // main.cpp:210-258
        Color refract_sum(0, 0, 0, 0);
        for (int i = 0; i < sample; i++) {
            double refractance = hit_color.getColorSpecial() - 1;
            double n1n2 = 1.0 / hit_obj->getRefraIdx();
            double dot1 = hit_normal.dotProduct(hit_dir.negtive());
            if (dot1 < 0) {                                                     // (the flip persists into the later passes)
                hit_normal = hit_normal.negtive();
                dot1 = hit_normal.dotProduct(hit_dir.negtive());
                n1n2 = 1.0 / n1n2;
            }
            double nNL = n1n2 * dot1;
            double underSQRT = 1 - n1n2 * n1n2 * (1 - dot1 * dot1);
            if (underSQRT > 0) {
                double coeffN = nNL - sqrt(underSQRT);
                Vect refraction_direction = hit_normal.vectMult(coeffN).vectAdd((hit_dir.negtive().vectMult(n1n2)).negtive());
                Vect world_dir = sample != 1 ? computeGlossyRay(refraction_direction, hit_normal, 0.2) : refraction_direction;
                Ray ray(hit_pos, world_dir);
                double intersect_distance = 1e9;
                Object* intersect_obj = bvh->Shoot(ray, intersect_distance);
                if (intersect_obj != nullptr && intersect_distance > accuracy) {    // main.cpp:245, on the new hit
                    Vect next_pos = hit_pos.vectAdd(refraction_direction.vectMult(accuracy + intersect_distance));
                    Color c = shade_intended(next_pos, refraction_direction, intersect_obj, accuracy, ambientLight, depth + 1);   // main.cpp:249
                    refract_sum = refract_sum.colorAdd(c.colorScalar(refractance));
                }
            }
            refract_sum = refract_sum.colorScalar(1.0 / sample);                // main.cpp:255-256, inside the loop as there
            final_color = final_color.colorAdd(refract_sum);
        }
    }
    return final_color.clip();                                                  // main.cpp:391
}

// Camera basis of main.cpp:626-635 and the sample position of main.cpp:753-787 (aax serves both offsets
// when aadepth > 1), with the reference's own Vect operations.
struct RefCam { Vect pos, dir, right, down; };
RefCam ref_cam(const double* cam)
{
    Vect Y(0, 1, 0);
    Vect camPos(cam[0], cam[1], cam[2]);
    Vect look_at(cam[3], cam[4], cam[5]);
    Vect diff_btw(camPos.getVectX() - look_at.getVectX(), camPos.getVectY() - look_at.getVectY(), camPos.getVectZ() - look_at.getVectZ());
    RefCam c;
    c.pos = camPos;
    c.dir = diff_btw.negtive().normalize();
    c.right = Y.crossProduct(c.dir).normalize();
    c.down = c.right.crossProduct(c.dir);
    return c;
}
Vect ref_cam_dir(const RefCam& c, int width, int height, int aadepth, int x, int y, int aax)
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
    } else {
        if (width > height) {
            xamnt = ((x + (double)aax / ((double)aadepth - 1)) / width) * aspectRatio - (((width - height) / (double)height) / 2);
            yamnt = ((height - y) + (double)aax / ((double)aadepth - 1)) / height;
        } else if (height > width) {
            xamnt = (x + (double)aax / ((double)aadepth - 1)) / width;
            yamnt = (((height - y) + (double)aax / ((double)aadepth - 1)) / height) / aspectRatio - (((height - width) / (double)width) / 2);
        } else {
            xamnt = (x + (double)aax / ((double)aadepth - 1)) / width;
            yamnt = ((height - y) + (double)aax / ((double)aadepth - 1)) / height;
        }
    }
    return c.dir.vectAdd(c.right.vectMult(xamnt - 0.5).vectAdd(c.down.vectMult(yamnt - 0.5))).normalize();
}

}  // namespace

extern "C" {

// shade_intended for a batch of hits (same arguments as ref_get_color_at)
void ref_color_intended(int64_t n, const double* pos, const double* dir, const int32_t* obj, int depth, double* rgb)
{
    for (int64_t i = 0; i < n; i++) {
        Color c = shade_intended(Vect(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]), Vect(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]),
                                 scene_objects.at(obj[i]), 0.000001, 0, depth);
        rgb[3 * i] = c.getColorRed(); rgb[3 * i + 1] = c.getColorGreen(); rgb[3 * i + 2] = c.getColorBlue();
    }
}

// One ray per (pixel, aax): the aadepth distinct camera rays of a pixel (main.cpp:771-787); dir = w*h*aadepth*3
void ref_primary_rays_aa(int w, int h, int aadepth, const double* cam, double* org, double* dir)
{
    RefCam c = ref_cam(cam);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int a = 0; a < aadepth; a++) {
                size_t i = ((size_t)y * w + x) * aadepth + a;
                Vect d = ref_cam_dir(c, w, h, aadepth, x, y, a);
                org[3 * i] = c.pos.getVectX(); org[3 * i + 1] = c.pos.getVectY(); org[3 * i + 2] = c.pos.getVectZ();
                dir[3 * i] = d.getVectX(); dir[3 * i + 1] = d.getVectY(); dir[3 * i + 2] = d.getVectZ();
            }
}

// The pixel loop of main.cpp:737-859 (DOF_SAMPLE 1), one thread, rows outermost (the reference's threads walk
// column strips; pixels only interact through the shared generator).  intended = 0 shades with the
// reference's getColorAt, 1 with shade_intended.  rgb: h x w x 3, pixel index y*w+x as main.cpp:739.
void ref_render(int w, int h, int aadepth, const double* cam, int intended, double* rgb)
{
    RefCam c = ref_cam(cam);
    const double accuracy = 0.000001, ambientLight = 0;
    std::vector<double> tempRed(aadepth * aadepth), tempGreen(aadepth * aadepth), tempBlue(aadepth * aadepth);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            for (int aax = 0; aax < aadepth; aax++)
                for (int aay = 0; aay < aadepth; aay++) {
                    int aa_index = aay * aadepth + aax;
                    Vect cam_ray_direction = ref_cam_dir(c, w, h, aadepth, x, y, aax);
                    Ray cam_ray(c.pos, cam_ray_direction);
                    double dofColor[3] = {0, 0, 0};
                    double intersect_distance = 1e9;
                    Object* intersect_obj = bvh->Shoot(cam_ray, intersect_distance);                  // main.cpp:812-813
                    if (intersect_obj != nullptr && intersect_distance > accuracy) {                  // main.cpp:815-820
                        Vect ipos = cam_ray.getRayOrigin().vectAdd(cam_ray_direction.vectMult(intersect_distance));
                        Color col = intended ? shade_intended(ipos, cam_ray_direction, intersect_obj, accuracy, ambientLight, 1)
                                             : getColorAt(ipos, cam_ray_direction, intersect_obj, accuracy, ambientLight, 1);
                        dofColor[0] += col.getColorRed(); dofColor[1] += col.getColorGreen(); dofColor[2] += col.getColorBlue();
                    }
                    tempRed[aa_index] = dofColor[0] / DOF_SAMPLE;                                     // main.cpp:831-833
                    tempGreen[aa_index] = dofColor[1] / DOF_SAMPLE;
                    tempBlue[aa_index] = dofColor[2] / DOF_SAMPLE;
                }
            double totalRed = 0, totalGreen = 0, totalBlue = 0;                                       // main.cpp:839-859
            for (int i = 0; i < aadepth * aadepth; i++) totalRed += tempRed[i];
            for (int i = 0; i < aadepth * aadepth; i++) totalGreen += tempGreen[i];
            for (int i = 0; i < aadepth * aadepth; i++) totalBlue += tempBlue[i];
            size_t px = (size_t)y * w + x;
            rgb[3 * px] = totalRed / (aadepth * aadepth);
            rgb[3 * px + 1] = totalGreen / (aadepth * aadepth);
            rgb[3 * px + 2] = totalBlue / (aadepth * aadepth);
        }
}

// A mesh whose faces carry no normal index (`f v/vt`, teapot1.obj, or `f v`, test.obj): the reference's own reader
// dereferences a null strtok result on them (IO.h:93), so the harness reads the file the way IO.h:75-119
// does -- words from the stream, offsets added as floats (IO.h:82), Triangle(points[v1], points[v2],
// points[v3]) (IO.h:103), setTexture where the face has texture indices (IO.h:105) -- and leaves out only the
// setNormals call the file has no data for: every such triangle shades with its flat normal.
int ref_scene_load_obj_no_normals(const char* path, double r, double g, double b, double special, float ior,
                                  float xo, float yo, float zo, int use_texture, int map_idx)
{
    std::fstream input(path, std::ios::in);
    if (!input) return 0;
    std::vector<Vect> points;
    std::vector<MyTexture> textures;
    std::string head;
    while (input >> head) {
        if (head == "v") {
            double a, b2, c2;
            input >> a >> b2 >> c2;
            points.push_back(Vect(a + xo, b2 + yo, c2 + zo));
        } else if (head == "vt") {
            double a, b2;
            input >> a >> b2;
            textures.push_back(MyTexture(a, b2));
        } else if (head == "f") {
            std::string corner[3];
            input >> corner[0] >> corner[1] >> corner[2];
            if (!input) break;
            int v[3], vt[3];
            for (int k = 0; k < 3; k++) {
                v[k] = atoi(corner[k].c_str()) - 1;
                size_t slash = corner[k].find('/');
                vt[k] = (slash != std::string::npos && slash + 1 < corner[k].size() && corner[k][slash + 1] != '/') ? atoi(corner[k].c_str() + slash + 1) - 1 : -1;
            }
            Triangle* t = new Triangle(points.at(v[0]), points.at(v[1]), points.at(v[2]), Color(r, g, b, special), ior);
            if (vt[0] >= 0 && vt[1] >= 0 && vt[2] >= 0)
                t->setTexture(textures.at(vt[0]), textures.at(vt[1]), textures.at(vt[2]), use_texture != 0 && map_idx >= 0,
                              map_idx >= 0 ? g_maps.at(map_idx) : new TextureMap());
            scene_objects.push_back(t);
        }
    }
    return 1;
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
