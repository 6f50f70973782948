// BVH of the scene API (reference: BVH.h:57-74, BVH.cpp:98-179).  The constructor registers the
// objects with the host scene behind the C ABI, which builds the reference's tree (same topology,
// leaf order and boxes) and flattens it; the flattened tree is copied to HBM.  Shoot() is the
// single-ray form of the batch entry point rt_trace_closest -- correct but latency bound; the
// pixel loop should go through Renderer, which keeps rays on the device.
//
// BVH(objects, true) builds a linear BVH on the GPU instead (milliseconds for millions of
// triangles): the same closest hits except on exact ties between primitives, where the reference's
// visiting order decides; the reference's host tree is then built only if something needs it
// (host_tree(), used by the serial photon emission).
#pragma once
#include <vector>

#include "Device.h"
#include "Object.h"
#include "Ray.h"
#include "TextureMap.h"

class BVH {
    std::vector<Object*> objects;   // borrowed, as in the reference (BVH.cpp:103)
    rt_scene* scene = nullptr;
    bool host_built = false;

public:
    BVH(std::vector<Object*>& in_objects, bool device_build = false) : objects(in_objects)
    {
        rt::check(rt_scene_create(&scene), "rt_scene_create");
        SceneSink sink;
        sink.scene = scene;
        for (Object* o : objects) rt::check(o->describe(sink), "Object::describe");
        // (the scene is replicated on every GPU of rt::devices())
        if (device_build) {
            for (rt_context* ctx : rt::devices()) rt::check(rt_scene_upload_lbvh(ctx, scene, nullptr), "rt_scene_upload_lbvh");
            return;
        }
        host_tree();
        for (rt_context* ctx : rt::devices()) rt::check(rt_scene_upload(ctx, scene), "rt_scene_upload");
    }
    ~BVH() { rt_scene_destroy(scene); }
    BVH(const BVH&) = delete;
    BVH& operator=(const BVH&) = delete;

    // nearest object hit below `distance`; `distance` becomes the hit distance (unchanged on a miss)
    Object* Shoot(const Ray& ray, double& distance) const
    {
        Vect o = ray.getRayOrigin(), d = ray.getRayDirection();
        int32_t id = -1;
        rt::check(rt_trace_closest(rt::device(), RT_F64, 1, o.data(), d.data(), &distance, &id), "rt_trace_closest");
        return id >= 0 ? objects[(size_t)id] : nullptr;
    }
    // batch form: n rays (origins / directions n x 3), distances in/out, hit objects out
    void Shoot(size_t n, const double* origins, const double* directions, double* distances, std::vector<Object*>& hits) const
    {
        std::vector<int32_t> ids(n);
        rt::check(rt_trace_closest(rt::device(), RT_F64, (int64_t)n, origins, directions, distances, ids.data()), "rt_trace_closest");
        hits.resize(n);
        for (size_t i = 0; i < n; i++) hits[i] = ids[i] >= 0 ? objects[(size_t)ids[i]] : nullptr;
    }

    rt_scene* host_scene() const { return scene; }
    // the host scene with the reference's tree built (BVH.cpp:98)
    rt_scene* host_tree()
    {
        if (!host_built) rt::check(rt_scene_build_bvh(scene), "rt_scene_build_bvh");
        host_built = true;
        return scene;
    }
    const std::vector<Object*>& scene_objects() const { return objects; }
};
