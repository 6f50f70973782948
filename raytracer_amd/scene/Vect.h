// Vect: the 3-vector of the scene API (reference: Vect.h:7-76).  Same constructor and method
// names; every operation returns by value and keeps the reference's evaluation order, so host
// arithmetic done through this class matches the reference bit for bit.
#pragma once
#include <cassert>
#include <cmath>

class Vect {
    double v[3];

public:
    Vect() : v{0, 0, 0} {}
    Vect(double i, double j, double k) : v{i, j, k} {}

    double getVectX() const { return v[0]; }
    double getVectY() const { return v[1]; }
    double getVectZ() const { return v[2]; }
    double operator[](int idx) const { assert(idx >= 0 && idx < 3); return v[idx]; }
    const double* data() const { return v; }

    double magnitude() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
    Vect normalize() const { double m = magnitude(); return Vect(v[0] / m, v[1] / m, v[2] / m); }
    Vect negtive() const { return Vect(-v[0], -v[1], -v[2]); }

    double dotProduct(const Vect& o) const { return v[0] * o.v[0] + v[1] * o.v[1] + v[2] * o.v[2]; }
    Vect crossProduct(const Vect& o) const
    {
        return Vect(v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]);
    }
    Vect vectAdd(const Vect& o) const { return Vect(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
    Vect vectMinus(const Vect& o) const { return Vect(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
    Vect vectMult(double s) const { return Vect(v[0] * s, v[1] * s, v[2] * s); }
    double sqrDist(const Vect& o) const
    {
        return (v[0] - o.v[0]) * (v[0] - o.v[0]) + (v[1] - o.v[1]) * (v[1] - o.v[1]) + (v[2] - o.v[2]) * (v[2] - o.v[2]);
    }
};
