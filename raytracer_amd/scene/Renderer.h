// Renderer: what replaces the two loops of the reference's main() -- photon emission
// (main.cpp:683-720) and the per-pixel ray-trace loop (main.cpp:735-863).  Emission stays on the
// host (it is RNG-order dependent); the pixel loop runs on the GPU through rt_render_rows.
#pragma once
#include <vector>

#include "BVH.h"
#include "Camera.h"
#include "Device.h"
#include "IO.h"
#include "Kdtree.h"
#include "Photon.h"

struct RenderParams : rt_render_params {
    RenderParams() { rt_default_params(this); }
};

class Renderer {
    BVH* bvh;

public:
    explicit Renderer(BVH* b) : bvh(b) {}

    // main.cpp:697-718; appends `new Photon`s (caller owns them).  on_device = false: one host thread
    // with a seeded std::mt19937, the reference's serial photon list; true: the same walk on the GPU
    // with one random stream per photon (same distribution, much faster).
    void emitPhotons(long long n_emit, Vect light_position, unsigned seed, std::vector<Photon*>& photons, std::vector<Photon*>& volumePhotons,
                     int bounce = 5, double accuracy = 0.000001, bool on_device = false)
    {
        rt_scene* s = on_device ? bvh->host_scene() : bvh->host_tree();
        rt_scene_photons_clear(s);
        if (on_device)
            rt::check(rt_scene_emit_photons_gpu(rt::device(), s, n_emit, light_position.data(), seed, bounce, accuracy, nullptr), "rt_scene_emit_photons_gpu");
        else
            rt::check(rt_scene_emit_photons(s, n_emit, light_position.data(), seed, bounce, accuracy), "rt_scene_emit_photons");
        for (int which = 0; which < 2; which++) {
            long long n = rt_scene_photon_count(s, which);
            std::vector<double> flat((size_t)n * 9);
            if (n) rt::check(rt_scene_photons_get(s, which, flat.data()), "rt_scene_photons_get");
            std::vector<Photon*>& out = which ? volumePhotons : photons;
            for (long long i = 0; i < n; i++) {
                const double* f = &flat[(size_t)i * 9];
                out.push_back(new Photon(Vect(f[0], f[1], f[2]), Color(f[3], f[4], f[5], 0), 0, 0, Vect(f[6], f[7], f[8])));
            }
        }
    }

    // main.cpp:683-727 in one call that never leaves the GPU: emission (the on_device streams above),
    // both photon lists in emission order and both kd-trees; nothing comes back to the host, so no
    // KDTree object is needed before render().  Returns the photons stored in the surface map.
    long long emitPhotonMaps(long long n_emit, Vect light_position, unsigned seed, int bounce = 5, double accuracy = 0.000001,
                             long long* volume_count = nullptr)
    {
        int64_t counts[2] = {0, 0};
        rt::check(rt_emit_photon_maps_dev(rt::device(), n_emit, light_position.data(), seed, bounce, accuracy, counts, nullptr), "rt_emit_photon_maps_dev");
        if (volume_count) *volume_count = counts[1];
        return counts[0];
    }

    // the pixel loop: pixels[y * width + x], y = 0 is the bottom BMP row (main.cpp:739,758)
    rt_stats render(const Camera& cam, const rt_render_params& params, RGBType* pixels)
    {
        rt_camera c = cam.abi();
        rt_stats st;
        rt::check(rt_render_rows(rt::device(), &c, &params, 0, params.height, &pixels[0].r, nullptr, &st), "rt_render_rows");
        return st;
    }
};
