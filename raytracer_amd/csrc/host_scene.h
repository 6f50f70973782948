// Host side of the path: scene assembly, the reference-compatible BVH builder and its flattener.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "rt_flat.h"

namespace rth {

struct Box { double lo[3], hi[3]; };

struct Texture { int x_res = 0, y_res = 0; std::vector<double> rgb; };

// One scene object in insertion order (the reference's scene_objects index, main.cpp:46).
struct Object {
    rtf::ObjShade sh;   // geometry + material as the kernels see it
    bool has_uv = false;   // texture coordinates were given (setTexture), whether or not a map is used
};

struct FlatBVH {
    std::vector<rtf::Node<double>> nodes;     // inner nodes, pre-order
    std::vector<rtf::Prim<double>> prims;     // leaf-ordered
    std::vector<rtf::Prim<double>> planes;    // side list, insertion order
    Box root_box{};
    uint32_t root_ref = rtf::NO_NODE;
    int depth = 0;                            // levels below the root (0: root is a leaf)
    int n_leaves = 0;
    bool built = false;
};

struct Scene {
    std::vector<Object> objects;
    std::vector<Texture> textures;
    FlatBVH bvh;

    int add_triangle(const double* A, const double* C, const double* B, const double* rgbs, float ior);
    int set_normals(int obj, const double* na, const double* nc, const double* nb);
    int set_texture(int obj, const double* ta, const double* tc, const double* tb, int use, int map);
    int add_sphere(const double* c, double radius, const double* rgbs, float ior);
    int add_plane(const double* n, double distance, const double* rgbs);
    void add_cornell_box(const double* c1, const double* c2);
    void add_cube(const double* c1, const double* c2, const double* rgbs);
    int load_obj(const char* path, const double* rgbs, float ior, float xo, float yo, float zo, int use_texture, int map);
    int add_texture_bmp(const char* path);
    int add_texture_rgb(int x_res, int y_res, const double* rgb);
    void build_bvh();
    void bvh_dump(std::vector<int32_t>& rec, std::vector<double>& boxes) const;
};

Box object_box(const rtf::ObjShade& o);
// traversal record of an object: A, B - A, C - A (triangle) or centre, radius (sphere) or normal, distance (plane)
rtf::Prim<double> make_prim(const rtf::ObjShade& o, uint32_t id);

// float copies of the flattened arrays
std::vector<rtf::Node<float>> to_f32(const std::vector<rtf::Node<double>>& n);
std::vector<rtf::Prim<float>> to_f32(const std::vector<rtf::Prim<double>>& p);

int save_bmp_bgr8(const char* path, int w, int h, int dpi, const uint8_t* bgr);
void quantize_bgr8(int64_t n_px, const double* rgb, uint8_t* bgr);

}  // namespace rth
