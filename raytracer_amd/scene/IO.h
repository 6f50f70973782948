// Scene I/O of the API (reference: IO.h:12-125, IO.cpp:5-154): RGBType, saveBmp, makeCube,
// makeCornellBox and ObjReader.  Header-only.  ObjReader accepts faces without texture / normal
// indices (the reference dereferences a null strtok result on them, IO.h:91-101).
#pragma once
#include <string>
#include <vector>

#include "../../include/rt_amd.h"
#include "Color.h"
#include "Object.h"
#include "Texture.h"
#include "TextureMap.h"
#include "Triangle.h"
#include "Vect.h"

struct RGBType {
    double r, g, b;
};

// byte-identical to the reference writer (54-byte header with its field placement, 24 bpp, no row padding)
inline void saveBmp(const char* filename, int w, int h, int dpi, RGBType* data)
{
    static_assert(sizeof(RGBType) == 3 * sizeof(double), "RGBType is three packed doubles");
    rt_save_bmp(filename, w, h, dpi, &data[0].r);
}

namespace rt_io_detail {
struct BoxCorners { Vect A, B, C, D, E, F; };
inline BoxCorners corners(const Vect& c1, const Vect& c2)
{
    return {Vect(c2[0], c1[1], c1[2]), Vect(c2[0], c1[1], c2[2]), Vect(c1[0], c1[1], c2[2]),
            Vect(c2[0], c2[1], c1[2]), Vect(c1[0], c2[1], c1[2]), Vect(c1[0], c2[1], c2[2])};
}
}  // namespace rt_io_detail

// twelve triangles, order: left, far, right, front, top, bottom (IO.cpp:70-106)
inline void makeCube(std::vector<Object*>& objects, Vect corner1, Vect corner2, Color color)
{
    auto k = rt_io_detail::corners(corner1, corner2);
    const Vect t[12][3] = {{k.D, k.A, corner1}, {corner1, k.E, k.D}, {corner2, k.B, k.A}, {k.A, k.D, corner2}, {k.F, k.C, k.B}, {k.B, corner2, k.F},
                           {k.E, corner1, k.C}, {k.C, k.F, k.E}, {k.D, k.E, k.F}, {k.F, corner2, k.D}, {corner1, k.A, k.B}, {k.B, k.C, corner1}};
    for (auto& v : t) objects.push_back(new Triangle(v[0], v[1], v[2], color, 2));
}

// ten triangles: green left wall, white far wall, red right wall, white top and bottom; the
// camera side stays open (IO.cpp:108-154)
inline void makeCornellBox(std::vector<Object*>& objects, Vect corner1, Vect corner2)
{
    auto k = rt_io_detail::corners(corner1, corner2);
    Color red(1, 0.25, 0.25, 0), green(0.25, 1, 0.25, 0), white(1.0, 1.0, 1.0, 0);
    struct { Vect a, c, b; Color col; } t[10] = {{k.D, k.A, corner1, green}, {corner1, k.E, k.D, green}, {corner2, k.B, k.A, white},
                                                  {k.A, k.D, corner2, white}, {k.F, k.C, k.B, red},         {k.B, corner2, k.F, red},
                                                  {k.D, k.E, k.F, white},     {k.F, corner2, k.D, white},   {corner1, k.A, k.B, white},
                                                  {k.B, k.C, corner1, white}};
    for (auto& v : t) objects.push_back(new Triangle(v.a, v.c, v.b, v.col, 20));
}

class ObjReader {
    std::string file_name;
    Color object_color;
    float refraIdxValue = 1, x_offset = 0, y_offset = 0, z_offset = 0;
    bool use_texture = false;
    TextureMap* my_map = nullptr;

public:
    ObjReader() {}
    ObjReader(std::string file, Color color, float refraIdx, float xoff, float yoff, float zoff, TextureMap* texture_map, bool use_tex)
        : file_name(file), object_color(color), refraIdxValue(refraIdx), x_offset(xoff), y_offset(yoff), z_offset(zoff), use_texture(use_tex),
          my_map(texture_map) {}
    void SetPara(std::string file, Color color, float refraIdx, float xoff, float yoff, float zoff)
    {
        file_name = file; object_color = color; refraIdxValue = refraIdx; x_offset = xoff; y_offset = yoff; z_offset = zoff;
    }

    // appends one `new Triangle` per face to *scene_objects (caller owns them, as in the reference)
    bool ReadContent(std::vector<Object*>* scene_objects)
    {
        // parse through the library's loader so host and GPU see the same triangles, then rebuild
        // them as API objects
        rt_scene* tmp = nullptr;
        if (rt_scene_create(&tmp) < 0) return false;
        int rc = rt_scene_load_obj(tmp, file_name.c_str(), object_color.data(), refraIdxValue, x_offset, y_offset, z_offset, 0, -1);
        if (rc < 0) { rt_scene_destroy(tmp); return false; }
        int n = rt_scene_size(tmp);
        for (int i = 0; i < n; i++) {
            double g[9 + 9 + 6];
            int flags = 0;
            rt_scene_get_triangle(tmp, i, g, g + 9, g + 18, &flags);
            Triangle* t = new Triangle(Vect(g[0], g[1], g[2]), Vect(g[6], g[7], g[8]), Vect(g[3], g[4], g[5]), object_color, refraIdxValue);
            if (flags & 1) t->setNormals(Vect(g[9], g[10], g[11]), Vect(g[15], g[16], g[17]), Vect(g[12], g[13], g[14]));
            if (flags & 2) t->setTexture(MyTexture(g[18], g[19]), MyTexture(g[22], g[23]), MyTexture(g[20], g[21]), use_texture, my_map);
            scene_objects->push_back(t);
        }
        rt_scene_destroy(tmp);
        return true;
    }
};
