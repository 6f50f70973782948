// Color: r, g, b plus `special`, the material selector (reference: Color.h:6-72):
//   special < 0 sub-surface scattering, == 0 Lambert (photon map), (0,1] mirror, (1,2] mirror + refraction.
// Binary operations keep the LEFT operand's special, as the reference does.
#pragma once

class Color {
    double c[4];

public:
    Color() : c{0.5, 0.5, 0.5, 0.0} {}
    Color(double r, double g, double b, double s) : c{r, g, b, s} {}

    double getColorRed() const { return c[0]; }
    double getColorGreen() const { return c[1]; }
    double getColorBlue() const { return c[2]; }
    double getColorSpecial() const { return c[3]; }
    double setColorRed(double x) { return c[0] = x; }
    double setColorGreen(double x) { return c[1] = x; }
    double setColorBlue(double x) { return c[2] = x; }
    double setColorSpecial(double x) { return c[3] = x; }
    const double* data() const { return c; }

    double brightness() const { return (c[0] + c[1] + c[2]) / 3; }
    Color colorScalar(double k) const { return Color(c[0] * k, c[1] * k, c[2] * k, c[3]); }
    Color colorAdd(const Color& o) const { return Color(c[0] + o.c[0], c[1] + o.c[1], c[2] + o.c[2], c[3]); }
    Color colorMultiply(const Color& o) const { return Color(c[0] * o.c[0], c[1] * o.c[1], c[2] * o.c[2], c[3]); }
    Color colorAverage(const Color& o) const { return Color((c[0] + o.c[0]) / 2, (c[1] + o.c[1]) / 2, (c[2] + o.c[2]) / 2, c[3]); }
    // clamp every channel to [0,1] in place and return the result (Color.h:39-57)
    Color clip()
    {
        for (int i = 0; i < 3; i++) {
            if (c[i] > 1) c[i] = 1;
            if (c[i] < 0) c[i] = 0;
        }
        return *this;
    }
};
