// Flattened scene layouts shared by the host flattener and the HIP kernels.
//
// The reference keeps a pointer tree of BVHNode {child[2], bbox, vector<Object*>} (BVH.h:43-55) and
// virtual Object instances.  Here the tree is an array of fixed-size inner-node records that hold
// BOTH child boxes (one fetch decides both children), and the leaves are ranges of a leaf-ordered
// primitive array; both exist in a double and a float instantiation.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define RT_FLAT_HD __host__ __device__
#else
#define RT_FLAT_HD
#endif

namespace rtf {

// child reference: inner-node index, or (LEAF_BIT | count << 27 | first_prim) with count <= 15
// (the reference splits every node holding >= 16 objects, BVH.h:45 / BVH.cpp:59).
constexpr uint32_t LEAF_BIT = 0x80000000u;
constexpr uint32_t NO_NODE = 0xFFFFFFFFu;   // empty scene
constexpr int LEAF_COUNT_SHIFT = 27;
constexpr uint32_t LEAF_FIRST_MASK = (1u << LEAF_COUNT_SHIFT) - 1;

inline RT_FLAT_HD uint32_t make_leaf(uint32_t first, uint32_t count) { return LEAF_BIT | (count << LEAF_COUNT_SHIFT) | first; }

// Inner node: float = 64 B, double = 128 B; 64 B aligned either way.
template <class R>
struct alignas(64) Node {
    R lo0[3], hi0[3];   // child 0 box (BBox::data[i][0] / [i][1], BVH.h:14)
    R lo1[3], hi1[3];   // child 1 box
    uint32_t c0, c1;
};

enum PrimKind : uint32_t { PRIM_TRI = 0, PRIM_SPHERE = 1, PRIM_PLANE = 2 };

// Leaf-ordered primitive: float = 48 B, double = 80 B.
//   triangle: a = A, e1 = B - A, e2 = C - A (the two edge vectors Triangle::findIntersection
//             recomputes per call, Triangle.h:263-264; same subtraction, so bit-identical)
//   sphere:   a = centre, e1[0] = radius
//   plane:    a = normal, e1[0] = distance
template <class R>
struct alignas(16) Prim {
    R a[3], e1[3], e2[3];
    uint32_t id;      // object id = index in insertion order
    uint32_t kind;
};

// Per-object shading record (double only; shading runs once per hit).
struct alignas(16) ObjShade {
    double A[3], B[3], C[3];        // Triangle.h:15 (sphere: A = centre, B[0] = radius; plane: A = normal)
    double NA[3], NB[3], NC[3];     // Triangle.h:16
    double tA[2], tB[2], tC[2];     // Triangle.h:20
    double color[4];                // r, g, b, special (Color.h:7)
    float ior;                      // refraIdx
    uint32_t kind;
    uint32_t has_normal;            // hasSetNormal
    uint32_t has_texture;
    int32_t map;
    int32_t pad[3];
};

struct TexDesc { int32_t x_res, y_res; int64_t offset; };   // offset in doubles into the texel pool (3 per texel)

template <class R>
struct SceneView {
    const Node<R>* nodes;
    const Prim<R>* prims;
    const Prim<R>* planes;
    R root_lo[3], root_hi[3];
    uint32_t root_ref;
    int32_t n_planes;
    int32_t tri_only;   // every primitive in the tree is a triangle (no per-primitive kind dispatch)
};

}  // namespace rtf
