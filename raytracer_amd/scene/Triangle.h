// Triangle of the scene API (reference: Triangle.h:11-280).  Constructor argument order is
// (A, C, B) as in the reference, likewise setNormals(a, c, b) and setTexture(tA, tC, tB, ...).
// Geometry and material live in the flat record the GPU kernels shade from, so host and device
// evaluate normals, colours and intersections through the same functions (csrc/rt_math.h).
#pragma once
#include "Object.h"
#include "Texture.h"
#include "TextureMap.h"

class Triangle : public Object {
    rtf::ObjShade sh{};
    TextureMap* my_map = nullptr;

    static Vect v3(const rtm::D3& d) { return Vect(d.x, d.y, d.z); }
    static rtm::D3 d3(const Vect& v) { return rtm::D3{v.getVectX(), v.getVectY(), v.getVectZ()}; }
    void set_points(const Vect& a, const Vect& b, const Vect& c)
    {
        for (int i = 0; i < 3; i++) { sh.A[i] = a[i]; sh.B[i] = b[i]; sh.C[i] = c[i]; }
        rtm::D3 n = rtm::tri_normal(sh);
        double nn[3] = {n.x, n.y, n.z};
        for (int i = 0; i < 3; i++) sh.NA[i] = sh.NB[i] = sh.NC[i] = nn[i];
    }

public:
    Triangle()
    {
        sh.kind = rtf::PRIM_TRI;
        set_points(Vect(1, 0, 0), Vect(0, 1, 0), Vect(0, 0, 1));
        const double grey[4] = {0.5, 0.5, 0.5, 0};
        for (int i = 0; i < 4; i++) sh.color[i] = grey[i];
        sh.ior = 1;
        sh.map = -1;
    }
    Triangle(Vect pointA, Vect pointC, Vect pointB, Color colorValue, float refraIdxValue)
    {
        sh.kind = rtf::PRIM_TRI;
        set_points(pointA, pointB, pointC);
        for (int i = 0; i < 4; i++) sh.color[i] = colorValue.data()[i];
        sh.ior = refraIdxValue;
        sh.map = -1;
    }

    virtual void setNormals(Vect normal_a, Vect normal_c, Vect normal_b)
    {
        for (int i = 0; i < 3; i++) { sh.NA[i] = normal_a[i]; sh.NB[i] = normal_b[i]; sh.NC[i] = normal_c[i]; }
        sh.has_normal = 1;
    }
    void setTexture(MyTexture texture_A, MyTexture texture_C, MyTexture texture_B, bool use_texture, TextureMap* map)
    {
        sh.tA[0] = texture_A.x; sh.tA[1] = texture_A.y;
        sh.tB[0] = texture_B.x; sh.tB[1] = texture_B.y;
        sh.tC[0] = texture_C.x; sh.tC[1] = texture_C.y;
        my_map = map;
        sh.has_texture = use_texture ? 1 : 0;
    }

    virtual bool isTextured() { return sh.has_texture != 0; }
    virtual Color getColor() { return Color(sh.color[0], sh.color[1], sh.color[2], sh.color[3]); }
    virtual Color getColor(const Vect& above_position)
    {
        if (sh.has_texture && my_map && !my_map->texels.empty()) {
            rtf::ObjShade tmp = sh;
            tmp.map = 0;
            rtf::TexDesc td{my_map->x_res, my_map->y_res, 0};
            double out[4];
            rtm::obj_color(tmp, &td, my_map->texels.data(), d3(above_position), out);
            return Color(out[0], out[1], out[2], out[3]);
        }
        return getColor();
    }
    virtual BBox getBBox() const
    {
        BBox r;
        for (int i = 0; i < 3; i++) {
            r.data[i][0] = std::min(sh.A[i], std::min(sh.B[i], sh.C[i]));
            r.data[i][1] = std::max(sh.A[i], std::max(sh.B[i], sh.C[i]));
        }
        return r;
    }
    virtual Vect getProjectPosition(const Vect& above_position) { return v3(rtm::tri_project(sh, d3(above_position))); }
    virtual Vect getTriangleNormal() { return v3(rtm::tri_normal(sh)); }
    virtual Vect getTriangleSmoothNormal(const Vect& above_position)
    {
        rtf::ObjShade tmp = sh;
        tmp.has_normal = 1;
        return v3(rtm::obj_normal(tmp, d3(above_position)));
    }
    virtual double getTriangleDistance() { return getTriangleNormal().dotProduct(Vect(sh.A[0], sh.A[1], sh.A[2])); }
    virtual Vect getNormalAt(const Vect& point) { return v3(rtm::obj_normal(sh, d3(point))); }
    virtual Vect getTangentAt(const Vect& point) { return v3(rtm::obj_tangent(sh, d3(point))); }
    virtual float getRefraIdx() { return sh.ior; }
    virtual double findIntersection(const Ray& ray)
    {
        double e1[3], e2[3];
        for (int i = 0; i < 3; i++) { e1[i] = sh.B[i] - sh.A[i]; e2[i] = sh.C[i] - sh.A[i]; }
        Vect o = ray.getRayOrigin(), d = ray.getRayDirection();
        return rtm::tri_intersect<double>(sh.A, e1, e2, o.data(), d.data());
    }

    virtual int describe(SceneSink& sink) const
    {
        int id = rt_scene_add_triangle(sink.scene, sh.A, sh.C, sh.B, sh.color, sh.ior);
        if (id < 0) return id;
        if (sh.has_normal) rt_scene_set_normals(sink.scene, id, sh.NA, sh.NC, sh.NB);
        int map = sink.map_index(my_map);
        rt_scene_set_texture(sink.scene, id, sh.tA, sh.tC, sh.tB, (sh.has_texture && map >= 0) ? 1 : 0, map);
        return id;
    }
};
