// Object: the abstract scene-object interface main.cpp is written against (reference:
// Object.h:11-29).  The virtuals keep their reference signatures and run on the host; what the
// GPU needs travels through describe(), which registers the object with the host scene behind
// the C ABI (rt_scene_add_*), in scene_objects order.
#pragma once
#include <map>

#include "../../include/rt_amd.h"
#include "BBox.h"
#include "Color.h"
#include "Ray.h"
#include "Vect.h"

class TextureMap;

// What describe() writes into: the host scene behind the C ABI plus the texture maps already
// registered with it (a map is uploaded once, however many triangles share it).
struct SceneSink {
    rt_scene* scene = nullptr;
    std::map<const TextureMap*, int> maps;
    int map_index(const TextureMap* m);   // defined in TextureMap.h
};

class Object {
public:
    Object() {}
    virtual ~Object() {}

    virtual Color getColor() { return Color(0.0, 0.0, 0.0, 0); }
    virtual Color getColor(const Vect&) { return Color(0.0, 0.0, 0.0, 0); }
    virtual BBox getBBox() const = 0;
    virtual double findIntersection(const Ray&) { return 0; }
    virtual bool isTextured() { return false; }
    virtual Vect getNormalAt(const Vect& point) = 0;
    virtual Vect getTangentAt(const Vect& point) = 0;
    virtual float getRefraIdx() { return 1; }

    // register with the host scene; returns the object id (>= 0) or a negative rt_status
    virtual int describe(SceneSink& sink) const = 0;
};
