// Host-side code of the library under the sanitizers (no GPU): scene assembly, the forked BVH build,
// the serial photon emission with its closest-hit walk, the threaded kd build and the BMP writer.
//   host_sanitize <mesh.obj> <tex.bmp> <out.bmp> <instances>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../raytracer_amd/csrc/host_photon.h"
#include "../../raytracer_amd/csrc/host_scene.h"

int main(int argc, char** argv)
{
    if (argc < 5) return 2;
    const int inst = atoi(argv[4]);
    rth::Scene s;
    const double grey[4] = {0.5, 0.5, 0.5, 0.0}, glass[4] = {1.0, 1.0, 1.0, 1.8}, c1[3] = {1, 1, 1}, c2[3] = {-1, -1, -1};
    if (s.add_texture_bmp(argv[2]) < 0) return 6;
    for (int i = 0; i < inst; i++)      // enough triangles for the build to fork its top levels
        if (s.load_obj(argv[1], i % 2 ? glass : grey, 1.5f, 0.0f, 1.5f * (float)(i / 8), 1.5f * (float)(i % 8), 0, -1) != 0) return 3;
    const double centre[3] = {0.3, 0.3, 0.3}, mirror[4] = {1, 1, 1, 0.9};
    s.add_sphere(centre, 0.2, mirror, 1.0f);
    s.add_cornell_box(c1, c2);
    s.build_bvh();
    std::vector<int32_t> rec;
    std::vector<double> boxes;
    s.bvh_dump(rec, boxes);

    std::vector<rth::PhotonRec> surface, volume;
    rth::EmitParams ep;
    const double light[3] = {0.1, 0.2, 0.0};
    rth::emit_photons(s, 3000, light, 12345u, ep, surface, volume);
    rth::FlatPhotonMap fm;
    rth::build_photon_map(surface.empty() ? nullptr : &surface[0].pos[0], (int64_t)surface.size(), fm, 8);
    rth::build_photon_map(nullptr, 0, fm, 8);

    const double o[3] = {2.8, 0, 0}, nan_dir[3] = {0.0 / 0.0, 0, 0}, axis_dir[3] = {-1, 0, 0};
    double dist = 1e9;
    int hit_nan = rth::host_shoot(s, o, nan_dir, dist);
    dist = 1e9;
    int hit_axis = rth::host_shoot(s, o, axis_dir, dist);

    std::vector<double> rgb(16 * 8 * 3, 0.25);
    std::vector<uint8_t> bgr(rgb.size());
    rth::quantize_bgr8(16 * 8, rgb.data(), bgr.data());
    if (rth::save_bmp_bgr8(argv[3], 16, 8, 72, bgr.data()) != 0) return 4;

    // bitmap headers that lie about their size: a truncated file and one announcing 2^31-1 x 2^31-1 texels are refused
    // before anything is allocated for them; the 16 x 8 file just written reads back
    {
        std::string good = argv[3], bad = good + ".bad";
        if (s.add_texture_bmp(good.c_str()) < 0) return 7;
        unsigned char header[54] = {'B', 'M'};
        for (int huge = 0; huge < 2; huge++) {
            const int w = huge ? 0x7fffffff : 64, h = huge ? 0x7fffffff : 64;
            for (int i = 0; i < 4; i++) { header[18 + i] = (unsigned char)(w >> (8 * i)); header[22 + i] = (unsigned char)(h >> (8 * i)); }
            FILE* f = fopen(bad.c_str(), "wb");
            if (!f) return 8;
            fwrite(header, 1, 54, f);
            fwrite(bgr.data(), 1, 100, f);      // far fewer texels than announced
            fclose(f);
            if (s.add_texture_bmp(bad.c_str()) >= 0) return 9;
        }
        remove(bad.c_str());
    }
    printf("objects %zu nodes %zu depth %d photons %zu nan-hit %d axis-hit %d\n", s.objects.size(), s.bvh.nodes.size(), s.bvh.depth, surface.size(), hit_nan, hit_axis);
    return hit_nan == -1 && hit_axis >= 0 ? 0 : 5;
}
