// Plane of the scene API (reference: Plane.h:10-53).  In the reference Plane does not override the
// `const&` virtuals and lacks getBBox / getTangentAt, so it cannot be instantiated; here it is
// concrete.  An unbounded plane has no box: the BVH keeps planes in a side list that is tested
// after the tree (getBBox returns an infinite box and is never used for building).
#pragma once
#include <limits>

#include "Object.h"

class Plane : public Object {
    Vect normal;
    double distance;
    Color color;

public:
    Plane() : normal(1, 0, 0), distance(0), color(0.5, 0.5, 0.5, 0) {}
    Plane(Vect normalValue, double distanceValue, Color colorValue) : normal(normalValue), distance(distanceValue), color(colorValue) {}

    virtual Vect getPlaneNormal() { return normal; }
    virtual double getPlaneDistance() { return distance; }
    virtual Color getColor() { return color; }
    virtual Color getColor(const Vect&) { return color; }
    virtual Vect getNormalAt(const Vect&) { return normal; }
    virtual Vect getTangentAt(const Vect&) { return Vect(1, 0, 0); }
    virtual BBox getBBox() const
    {
        BBox r;
        double inf = std::numeric_limits<double>::infinity();
        for (int i = 0; i < 3; i++) { r.data[i][0] = -inf; r.data[i][1] = inf; }
        return r;
    }
    virtual double findIntersection(const Ray& ray)
    {
        Vect o = ray.getRayOrigin(), d = ray.getRayDirection();
        return rtm::plane_intersect<double>(normal.data(), distance, o.data(), d.data());
    }
    virtual int describe(SceneSink& sink) const { return rt_scene_add_plane(sink.scene, normal.data(), distance, color.data()); }
};
