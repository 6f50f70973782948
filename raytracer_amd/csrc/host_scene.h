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
std::vector<rtf::Node<float>> to_f32(const rtf::Node<double>* n, size_t count);
std::vector<rtf::Prim<float>> to_f32(const rtf::Prim<double>* p, size_t count);
inline std::vector<rtf::Node<float>> to_f32(const std::vector<rtf::Node<double>>& n) { return to_f32(n.data(), n.size()); }
inline std::vector<rtf::Prim<float>> to_f32(const std::vector<rtf::Prim<double>>& p) { return to_f32(p.data(), p.size()); }

// ---- flat image of a built scene -----------------------------------------------------------------------------
// One position-independent block of bytes holding everything the device needs of a scene whose tree has been
// built: shading records, inner nodes, leaf-ordered primitive records, the plane list, texture descriptors and
// texels.  A process that built the tree (the reference builds it once, before its threads fan out,
// main.cpp:678,735) hands the image to the others, which upload it without parsing meshes or building anything.
// Every section starts at a multiple of 64 bytes; the image itself must sit at a 64-byte aligned address.
constexpr uint32_t FLAT_MAGIC = 0x53465452u;      // "RTFS"
constexpr uint32_t FLAT_VERSION = 1;
struct FlatImageHeader {
    uint32_t magic, version;
    uint64_t total_bytes;
    uint32_t sizeof_header, sizeof_objshade, sizeof_node, sizeof_prim;      // layout guard between builds
    uint64_t n_objects, n_nodes, n_prims, n_planes, n_textures, n_texel_doubles;
    uint64_t off_objects, off_has_uv, off_nodes, off_prims, off_planes, off_texdesc, off_texels;
    double root_lo[3], root_hi[3];
    uint32_t root_ref;
    int32_t depth, n_leaves;
    uint32_t reserved[15];
};
static_assert(sizeof(FlatImageHeader) % 64 == 0, "sections behind the header stay 64-byte aligned");

// pointers into an image that flat_image_open has checked
struct FlatImageView {
    const FlatImageHeader* h = nullptr;
    const rtf::ObjShade* objects = nullptr;
    const uint8_t* has_uv = nullptr;
    const rtf::Node<double>* nodes = nullptr;
    const rtf::Prim<double>* prims = nullptr;
    const rtf::Prim<double>* planes = nullptr;
    const rtf::TexDesc* texdesc = nullptr;
    const double* texels = nullptr;
    int depth = 0, n_leaves = 0;      // recomputed from the nodes, not taken from the header
};

size_t flat_image_bytes(const Scene& s);
void flat_image_write(const Scene& s, void* out);      // out: flat_image_bytes(s) bytes, 64-byte aligned; the tree must be built
// Checks everything a kernel would trust (sizes, section bounds, child references, primitive / object / texture
// indices; inner children follow their parent, so the tree has no cycle).  0, or -1 with the reason in *err.
int flat_image_open(const void* buf, size_t bytes, FlatImageView* v, std::string* err);
void flat_image_read(const FlatImageView& v, Scene& s);      // replaces the scene's contents

int save_bmp_bgr8(const char* path, int w, int h, int dpi, const uint8_t* bgr);
void quantize_bgr8(int64_t n_px, const double* rgb, uint8_t* bgr);

}  // namespace rth
