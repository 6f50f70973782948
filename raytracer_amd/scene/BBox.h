// Axis-aligned box of the scene API (reference: BBox in BVH.h:11-41, BVH.cpp:5-40).
#pragma once
#include <algorithm>

#include "../csrc/rt_math.h"
#include "Ray.h"
#include "Vect.h"

struct BBox {
    double data[3][2];   // x,y,z | min,max

    Vect Center() const { return Vect((data[0][1] + data[0][0]) / 2, (data[1][1] + data[1][0]) / 2, (data[2][1] + data[2][0]) / 2); }
    double Size() const { return data[0][1] - data[0][0] + data[1][1] - data[1][0] + data[2][1] - data[2][0]; }
    BBox Union(const BBox& o) const
    {
        BBox r;
        for (int i = 0; i < 3; i++) { r.data[i][0] = std::min(data[i][0], o.data[i][0]); r.data[i][1] = std::max(data[i][1], o.data[i][1]); }
        return r;
    }
    // slab test clipped to [0, current_best]; returns the entry distance (0 inside) or -1
    double Intersect(const Ray& ray, double current_best) const
    {
        double lo[3] = {data[0][0], data[1][0], data[2][0]}, hi[3] = {data[0][1], data[1][1], data[2][1]};
        Vect o = ray.getRayOrigin(), d = ray.getRayDirection();
        return rtm::slab_literal<double>(lo, hi, o.data(), d.data(), current_best);
    }
};
