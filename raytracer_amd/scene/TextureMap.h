// TextureMap: 24-bit BMP texture with wrap-around bilinear lookup (reference: TextureMap.h:9-188).
// u indexes rows (x_res = BMP height), v indexes columns (y_res = BMP width); texels are kept as
// row-major doubles in [0,1].  The file is closed after reading (the reference leaks the handle).
#pragma once
#include <cstdio>
#include <string>
#include <vector>

#include "../csrc/rt_math.h"
#include "Color.h"
#include "Object.h"
#include "Texture.h"

class TextureMap {
public:
    int x_res = 0, y_res = 0;
    std::vector<double> texels;   // x_res * y_res * 3
    std::string source;           // file the map was read from (empty: none)

    TextureMap() {}

    void TextureMapRead(std::string file_name)
    {
        FILE* fp = fopen(file_name.c_str(), "rb");
        if (!fp) return;                         // silent, as the reference (TextureMap.h:23-24)
        unsigned char header[54];
        if (fread(header, 1, 54, fp) != 54) { fclose(fp); return; }
        y_res = header[18] | header[19] << 8 | header[20] << 16 | header[21] << 24;
        x_res = header[22] | header[23] << 8 | header[24] << 16 | header[25] << 24;
        size_t n = (size_t)x_res * y_res;
        std::vector<unsigned char> raw(n * 3, 0);
        size_t got = fread(raw.data(), 1, raw.size(), fp);
        (void)got;
        fclose(fp);
        texels.resize(n * 3);
        for (size_t i = 0; i < n; i++) {
            texels[3 * i] = raw[3 * i + 2] / 255.0;
            texels[3 * i + 1] = raw[3 * i + 1] / 255.0;
            texels[3 * i + 2] = raw[3 * i] / 255.0;
        }
        source = file_name;
    }

    double getBilinearInterpolate(double lu, double ld, double ru, double rd, double s, double t)
    {
        return lu * (1 - s) * (1 - t) + ld * s * (1 - t) + ru * (1 - s) * t + rd * s * t;
    }

    Color GetColor(double u, double v)
    {
        double rgb[3] = {0, 0, 0};
        if (!texels.empty()) rtm::tex_lookup(texels.data(), x_res, y_res, u, v, rgb);
        return Color(rgb[0], rgb[1], rgb[2], 0);
    }
    Color GetColor(MyTexture t) { return GetColor(t.ReturnU(), t.ReturnV()); }
};

inline int SceneSink::map_index(const TextureMap* m)
{
    if (!m || m->texels.empty()) return -1;
    auto it = maps.find(m);
    if (it != maps.end()) return it->second;
    int idx = rt_scene_add_texture_rgb(scene, m->x_res, m->y_res, m->texels.data());
    maps[m] = idx;
    return idx;
}
