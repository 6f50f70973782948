// Sphere of the scene API (reference: Sphere.h:9-105), including the reference's intersection
// quirks (origin pre-offset by 0.1 * direction, second root as written there) -- parity needs them.
#pragma once
#include "Object.h"

class Sphere : public Object {
    Vect center;
    double radius;
    Color color;
    float refraIdx;

public:
    Sphere() : center(0, 0, 0), radius(1.0), color(0.5, 0.5, 0.5, 0), refraIdx(1.0) {}
    Sphere(Vect centerValue, double radiusValue, Color colorValue, float refraIdxValue)
        : center(centerValue), radius(radiusValue), color(colorValue), refraIdx(refraIdxValue) {}

    virtual BBox getBBox() const
    {
        BBox r;
        for (int i = 0; i < 3; i++) { r.data[i][0] = center[i] - radius; r.data[i][1] = center[i] + radius; }
        return r;
    }
    virtual Vect getSphereCenter() { return center; }
    virtual double getSphereRadius() { return radius; }
    virtual Color getColor() { return color; }
    virtual Color getColor(const Vect&) { return color; }
    virtual Vect getNormalAt(const Vect& point) { return point.vectAdd(center.negtive()).normalize(); }
    virtual float getRefraIdx() { return refraIdx; }
    virtual Vect getTangentAt(const Vect&) { return Vect(1, 0, 0); }
    virtual double findIntersection(const Ray& ray)
    {
        Vect o = ray.getRayOrigin(), d = ray.getRayDirection();
        return rtm::sphere_intersect<double>(center.data(), radius, o.data(), d.data());
    }
    virtual int describe(SceneSink& sink) const { return rt_scene_add_sphere(sink.scene, center.data(), radius, color.data(), refraIdx); }
};
