// Ray = origin + direction (reference: Ray.h:7-22).  No tmin/tmax: the range travels as the
// in/out `distance` argument of BVH::Shoot.
#pragma once
#include "Vect.h"

class Ray {
    Vect o, d;

public:
    Ray() : o(0, 0, 0), d(1, 0, 0) {}
    Ray(Vect origin, Vect direction) : o(origin), d(direction) {}
    Vect getRayOrigin() const { return o; }
    Vect getRayDirection() const { return d; }
};
