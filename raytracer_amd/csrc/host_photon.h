// Host side of the photon map: emission (stays on the host: it is RNG-order dependent,
// main.cpp:396-598, 697-718) and the kd-tree build + flattening for the gather kernel.
#pragma once
#include <stdint.h>

#include <random>
#include <vector>

#include "host_scene.h"

namespace rth {

// Photon.h:7-28 without the unused phi/theta bytes: position, power (r,g,b), direction.
struct PhotonRec { double pos[3], power[3], dir[3]; };

struct EmitParams {
    int bounce = 5;            // BOUNCE, main.cpp:25
    double accuracy = 1e-6;    // main.cpp:614
};

// Closest hit on the host over the flattened BVH, literal recursion of BVH::Shoot (BVH.cpp:108-179).
// Used by the host-side photon emission only; never by a device entry point.
int host_shoot(const Scene& s, const double* o, const double* d, double& dist);

// Emission loop of main.cpp:697-718 run on one thread with a seeded std::mt19937 (the reference
// seeds from std::random_device and shares the generator between threads, so its photon list is
// not reproducible; a fixed seed and one thread give the same list as the reference run that way).
void emit_photons(const Scene& s, int64_t n_emit, const double* light_pos, uint32_t seed, const EmitParams& ep,
                  std::vector<PhotonRec>& surface, std::vector<PhotonRec>& volume);

// Flattened photon kd-tree.  The reference's tree (Kdtree.cpp:11-52: widest axis, sort, pivot at
// n/2, buckets of <= 8) is kept as an IMPLICIT tree over one array in in-order layout: a node is a
// range [lo, hi), its pivot sits at mid = lo + (hi - lo) / 2, the left subtree is [lo, mid), the
// right one (mid, hi); ranges of <= LEAF photons are leaves scanned by a whole wave.
struct FlatPhotonMap {
    static constexpr int LEAF = 64;
    std::vector<float> pos;        // 4 per photon, tree order: x, y, z and the split axis (as a float 0/1/2) of the node
                                   // whose pivot this photon is -- one aligned 16-byte record per photon
    std::vector<float> aux;        // 6 per photon: power rgb, dir xyz
    std::vector<uint8_t> axis;     // split axis of the node whose pivot this photon is
    std::vector<int32_t> order;    // tree position -> index in the caller's array
    std::vector<float> leaf_box;   // 6 per slot (lo xyz, hi xyz of the float positions): slot = first photon of the leaf / 32
                                   // (a leaf holds 33..64 photons unless it is the whole map, so the slots are distinct)
    int64_t n = 0;
};

void build_photon_map(const double* photons9, int64_t n, FlatPhotonMap& out, int threads = 8);

}  // namespace rth
