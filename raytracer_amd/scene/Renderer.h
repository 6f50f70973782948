// Renderer: what replaces the two loops of the reference's main() -- photon emission
// (main.cpp:683-720) and the per-pixel ray-trace loop (main.cpp:735-863).  Emission stays on the
// host (it is RNG-order dependent); the pixel loop runs on the GPU through rt_render_rows.
#pragma once
#include <cstring>
#include <string>
#include <thread>
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
        // every GPU emits for itself: the streams are keyed by seed and photon number, so the maps are identical
        for (rt_context* ctx : rt::devices())
            rt::check(rt_emit_photon_maps_dev(ctx, n_emit, light_position.data(), seed, bounce, accuracy, counts, nullptr), "rt_emit_photon_maps_dev");
        if (volume_count) *volume_count = counts[1];
        return counts[0];
    }

    // the pixel loop: pixels[y * width + x], y = 0 is the bottom BMP row (main.cpp:739,758).  With several GPUs
    // (RT_DEVICES) one host thread drives each -- the reference's thread fan-out, main.cpp:689-695,735-736 -- over
    // bands of 8 rows dealt round-robin (cost is uneven over the frame), or contiguous row blocks when the height does
    // not divide into such bands.  Returns RT_OK or the rt_status of the part that failed (rt_last_error() and
    // last_error() then hold its message; `pixels` is incomplete); the reference's loop cannot fail, a GPU can.
    int render(const Camera& cam, const rt_render_params& params, RGBType* pixels, rt_stats* stats = nullptr)
    {
        rt_camera c = cam.abi();
        const std::vector<rt_context*>& ctxs = rt::devices();
        const int n = (int)ctxs.size(), w = params.width, h = params.height;
        rt_stats st;
        memset(&st, 0, sizeof st);
        error.clear();
        if (n == 1) {
            int rc = rt_render_rows(ctxs[0], &c, &params, 0, h, &pixels[0].r, nullptr, &st);
            if (rc < 0) error = rt_last_error();
            if (stats) *stats = st;
            return rc;
        }
        const int band = 8;
        const bool bands = h % (band * n) == 0;
        std::vector<rt_stats> part_st((size_t)n);
        std::vector<int> status((size_t)n, 0);
        std::vector<std::string> message((size_t)n);
        std::vector<std::vector<double>> part(bands ? (size_t)n : 0);
        std::vector<std::thread> threads;
        for (int r = 0; r < n; r++)
            threads.emplace_back([&, r]() {
                rt_render_params p = params;
                memset(&part_st[(size_t)r], 0, sizeof(rt_stats));
                if (bands) {
                    p.band_rows = band; p.n_parts = n; p.part = r;
                    part[(size_t)r].resize((size_t)w * (size_t)(h / n) * 3);
                    status[(size_t)r] = rt_render_rows(ctxs[(size_t)r], &c, &p, 0, h / n, part[(size_t)r].data(), nullptr, &part_st[(size_t)r]);
                } else {
                    const int r0 = (int)((long long)h * r / n), r1 = (int)((long long)h * (r + 1) / n);
                    if (r1 > r0) status[(size_t)r] = rt_render_rows(ctxs[(size_t)r], &c, &p, r0, r1, &pixels[(size_t)r0 * (size_t)w].r, nullptr, &part_st[(size_t)r]);
                }
                if (status[(size_t)r] < 0) message[(size_t)r] = rt_last_error();      // (the message is per thread)
            });
        for (std::thread& t : threads) t.join();
        for (int r = 0; r < n; r++)
            if (status[(size_t)r] < 0) {
                error = "GPU " + std::to_string(r) + " of " + std::to_string(n) + ": " + message[(size_t)r];
                return status[(size_t)r];
            }
        if (bands)
            for (int r = 0; r < n; r++)
                for (int yc = 0; yc < h / n; yc++) {
                    const int y = ((yc / band) * n + r) * band + yc % band;
                    memcpy(&pixels[(size_t)y * (size_t)w].r, &part[(size_t)r][(size_t)yc * (size_t)w * 3], (size_t)w * 3 * sizeof(double));
                }
        for (const rt_stats& s : part_st) {
            st.primary_rays += s.primary_rays; st.secondary_rays += s.secondary_rays; st.shadow_rays += s.shadow_rays;
            st.gather_queries += s.gather_queries; st.gather_single += s.gather_single;
            st.kernel_ms = s.kernel_ms > st.kernel_ms ? s.kernel_ms : st.kernel_ms;
            st.trace_kernel_ms = s.trace_kernel_ms > st.trace_kernel_ms ? s.trace_kernel_ms : st.trace_kernel_ms;
        }
        if (stats) *stats = st;
        return RT_OK;
    }

    // message of the last failed render() of this object
    const std::string& last_error() const { return error; }

private:
    std::string error;
};
