// Source: abstract light interface (reference: Source.h:4-12).
#pragma once
#include "Color.h"
#include "Vect.h"

class Source {
public:
    Source() {}
    virtual ~Source() {}
    virtual Vect getLightPosition() { return Vect(0, 0, 0); }
    virtual Color getLightColor() { return Color(1, 1, 1, 0); }
};
