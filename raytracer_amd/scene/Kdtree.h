// Photon kd-tree of the scene API (reference: Kdtree.h:25-73, Kdtree.cpp:11-158).  The constructor
// flattens the photons into the GPU photon map (rt_photons_upload); findKNN returns the k nearest
// photons within PHOTONDIST = 0.28 of `center`, nearest first, as borrowed pointers.
// KDTree(photons, which, true) builds the tree on the GPU instead of the host (same split rule,
// same k-NN results; about ten times faster for a million photons).  The device holds two photon
// maps; each live KDTree owns one of them (see claim_slot).
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "Device.h"
#include "Photon.h"
#include "Vect.h"

class KDTree {
    std::vector<Photon*> photons;   // borrowed
    int which;                      // the device photon-map slot this tree owns
    bool device_build;

    // Which of the two device photon maps a live tree owns.  The reference builds `kdtree` and then
    // `volumeKDTree` with the same one-argument constructor (main.cpp:724,727): the first live tree takes the
    // surface map, the second the volume map, and a tree's slot is free again when it is destroyed.
    static bool* slot_taken()
    {
        static bool taken[2] = {false, false};
        return taken;
    }
    static int claim_slot(int wanted)
    {
        bool* taken = slot_taken();
        int slot = wanted;
        if (slot < 0) slot = !taken[0] ? 0 : (!taken[1] ? 1 : -1);
        if (slot < 0 || slot > 1 || taken[slot]) {
            fprintf(stderr, "raytracer_amd: KDTree: %s\n", slot < 0 || slot > 1 ? "a surface and a volume photon map are live already (the device holds two maps)"
                                                                                : "that photon map belongs to another live KDTree");
            abort();
        }
        taken[slot] = true;
        return slot;
    }

public:
    // which_map: -1 = the first free one of 0 = surface map (`kdtree`, main.cpp:724) and 1 = volume map
    // (`volumeKDTree`, main.cpp:727); naming a map another live tree owns is fatal.
    KDTree(std::vector<Photon*>& in_photons, int which_map = -1, bool build_on_device = false)
        : photons(in_photons), which(claim_slot(which_map)), device_build(build_on_device)
    {
        createKD(in_photons);
    }
    KDTree(const KDTree&) = delete;
    KDTree& operator=(const KDTree&) = delete;
    ~KDTree()
    {
        // the map leaves the devices with its tree: a later frame must not gather from photons no tree stands for
        for (rt_context* ctx : rt::devices()) (void)rt_photons_upload(ctx, which, 0, nullptr);
        slot_taken()[which] = false;
    }

    int map() const { return which; }

    void createKD(std::vector<Photon*>& in_photons)
    {
        photons = in_photons;
        std::vector<double> flat(photons.size() * 9);
        for (size_t i = 0; i < photons.size(); i++) {
            const Photon* p = photons[i];
            double* f = &flat[i * 9];
            for (int c = 0; c < 3; c++) { f[c] = p->position[c]; f[6 + c] = p->dir[c]; }
            f[3] = p->power.getColorRed(); f[4] = p->power.getColorGreen(); f[5] = p->power.getColorBlue();
        }
        for (rt_context* ctx : rt::devices()) {       // the photon map is replicated like the scene
            if (device_build)
                rt::check(rt_photons_upload_dev_build(ctx, which, (int64_t)photons.size(), flat.data()), "rt_photons_upload_dev_build");
            else
                rt::check(rt_photons_upload(ctx, which, (int64_t)photons.size(), flat.data()), "rt_photons_upload");
        }
    }

    std::vector<Photon*> findKNN(int k, Vect center) const
    {
        std::vector<int32_t> idx((size_t)k);
        int32_t count = 0;
        rt::check(rt_knn_query(rt::device(), which, k, 0.28, 1, center.data(), idx.data(), &count), "rt_knn_query");
        std::vector<Photon*> out;
        for (int i = 0; i < count; i++) out.push_back(photons[(size_t)idx[i]]);
        std::sort(out.begin(), out.end(), [&](const Photon* a, const Photon* b) { return a->position.sqrDist(center) < b->position.sqrDist(center); });
        return out;
    }
};
