// Light: a point light (reference: Light.h:8-27); its position is the photon origin.
#pragma once
#include "Source.h"

class Light : public Source {
    Vect position;
    Color color;

public:
    Light() : position(0, 0, 0), color(1, 1, 1, 0) {}
    Light(Vect p, Color c) : position(p), color(c) {}
    virtual Vect getLightPosition() { return position; }
    virtual Color getLightColor() { return color; }
};
