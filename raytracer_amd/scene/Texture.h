// MyTexture: a (u, v) texture coordinate (reference: Texture.h:3-48).
#pragma once

class MyTexture {
public:
    double x, y;
    MyTexture() : x(0), y(0) {}
    MyTexture(double u, double v) : x(u), y(v) {}
    MyTexture Vect2DMulti(double k) const { return MyTexture(x * k, y * k); }
    MyTexture Vect2DAdd(const MyTexture& t) const { return MyTexture(x + t.x, y + t.y); }
    MyTexture Vect2DNegate() const { return MyTexture(x * (-1), (-1) * y); }
    double ReturnU() const { return x; }
    double ReturnV() const { return y; }
};
