/* TEST INFRASTRUCTURE ONLY -- the product library never includes, links or loads this.
 *
 * C interface of the CPU restatement (rt_oracle.cpp) of the reference renderer's hot path:
 * closest-hit BVH traversal, primitive tests, normals/texture lookups, the recursive shade loop,
 * the photon kd-tree k-NN gather and the pixel quantiser -- all in double precision, same
 * operation order as the reference, compiled with -ffp-contract=off.  Each function names the
 * reference file:line it follows in rt_oracle.cpp.
 *
 * Parity status: PINNED.  oracle/make_golden.py checks every function below against the compiled
 * reference itself (oracle/_ref/librt_ref.so, built from the untouched sources) and commits the
 * resulting vectors under tests/golden/; tests/test_oracle_golden.py re-checks the restatement
 * against those vectors wherever the reference is not mounted.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

typedef struct orc_params {
    int32_t photon_num;      /* PHOTONMUM  (main.cpp:21)  */
    int32_t photon_use;      /* PHOTONUSE  (main.cpp:22)  */
    int32_t tracing_depth;   /* TRACING_DEPTH (main.cpp:28) */
    int32_t bounce;          /* BOUNCE (main.cpp:25) */
    int32_t head_quirk;      /* 1: recurse with hit_obj as HEAD does (main.cpp:198,249); 0: recurse on the new hit */
    int32_t use_photon_map;  /* 0: diffuse hits return black without a gather (no photon map loaded) */
    double accuracy;         /* 1e-6 (main.cpp:614) */
    double photon_dist;      /* PHOTONDIST 0.28 (Kdtree.cpp:6) */
    double focal_len;        /* FOCAL_LEN 2.6 (main.cpp:36) */
    double aperture;         /* APERTURE_SIZE 0.2 (main.cpp:37) */
    int32_t dof_samples;     /* DOF_SAMPLE 1 (main.cpp:35) */
    int32_t counter_rng;     /* 0: draws come from the scene's std::mt19937 in call order, as in the reference;
                                1: every draw is a function of (pixel or hit point, draw number, rng_seed) -- the
                                form a parallel renderer can reproduce */
    uint32_t rng_seed;
    int32_t pad_;
} orc_params;

void orc_default_params(orc_params* p);

orc_scene* orc_scene_new(void);
void orc_scene_free(orc_scene*);
int32_t orc_scene_size(const orc_scene*);

/* Scene construction -- same argument orders as the reference constructors. */
int32_t orc_add_triangle(orc_scene*, const double* A, const double* C, const double* B, const double* rgbs, float ior);
void orc_last_set_normals(orc_scene*, const double* na, const double* nc, const double* nb);
void orc_last_set_texture(orc_scene*, const double* ta, const double* tc, const double* tb, int32_t use, int32_t map_idx);
int32_t orc_add_sphere(orc_scene*, const double* c, double radius, const double* rgbs, float ior);
int32_t orc_add_plane(orc_scene*, const double* normal, double distance, const double* rgbs);
void orc_add_cornell(orc_scene*, const double* c1, const double* c2);
void orc_add_cube(orc_scene*, const double* c1, const double* c2, const double* rgbs);
int32_t orc_load_obj(orc_scene*, const char* path, const double* rgbs, float ior, float xo, float yo, float zo,
                     int32_t use_texture, int32_t map_idx);
int32_t orc_texture_read(orc_scene*, const char* path);
void orc_texture_res(const orc_scene*, int32_t idx, int32_t* xres, int32_t* yres);
void orc_texture_get(const orc_scene*, int32_t idx, int64_t n, const double* uv, double* rgb);

/* Acceleration structures */
void orc_build_bvh(orc_scene*);
int64_t orc_bvh_dump(orc_scene*, int64_t* n_boxes);
void orc_bvh_dump_copy(orc_scene*, int32_t* rec, double* boxes);
void orc_shoot(const orc_scene*, int64_t n, const double* org, const double* dir, double* dist, int32_t* ids);
void orc_shoot_mt(const orc_scene*, int64_t n, const double* org, const double* dir, double* dist, int32_t* ids, int32_t threads);
/* per-ray work counters of the traversal: slab calls, inner nodes entered, primitive tests */
/* the same traversal in single precision on float-rounded scene data (checks the float device instantiation) */
void orc_shoot_f32(const orc_scene*, int64_t n, const double* org, const double* dir, double* dist, int32_t* ids, int32_t threads);
void orc_shoot_stats(const orc_scene*, int64_t n, const double* org, const double* dir, double max_dist, int64_t* counters);
void orc_brute_force(const orc_scene*, int64_t n, const double* org, const double* dir, double* dist, int32_t* ids);

/* Primitive-level functions */
void orc_obj_intersect(const orc_scene*, int32_t obj, int64_t n, const double* org, const double* dir, double* t);
void orc_bbox_intersect(const double* box, int64_t n, const double* org, const double* dir, const double* best, double* t);
void orc_obj_bbox(const orc_scene*, int32_t obj, double* box);
void orc_obj_normal(const orc_scene*, const int32_t* obj, int64_t n, const double* pos, double* out);
void orc_obj_tangent(const orc_scene*, const int32_t* obj, int64_t n, const double* pos, double* out);
void orc_obj_color(const orc_scene*, const int32_t* obj, int64_t n, const double* pos, double* out);
void orc_reflect_dir(int64_t n, const double* nrm, const double* dir, double* out);

/* Photon map */
void orc_seed(orc_scene*, uint32_t seed);
int64_t orc_emit_photons(orc_scene*, int64_t n_emit, const double* light_pos, const orc_params*);
int64_t orc_photon_count(const orc_scene*, int32_t which);
void orc_photons_get(const orc_scene*, int32_t which, double* out);
void orc_photons_set(orc_scene*, int32_t which, int64_t n, const double* in);
void orc_build_kdtree(orc_scene*);
int64_t orc_kd_dump(orc_scene*);
void orc_kd_dump_copy(orc_scene*, int32_t* rec);
void orc_knn(const orc_scene*, int32_t which, int32_t k, int64_t n, const double* centers, int32_t* out_idx,
             int32_t* out_cnt, double* out_d2);

/* Shading */
void orc_get_color_at(const orc_scene*, const orc_params*, int64_t n, const double* pos, const double* dir,
                      const int32_t* obj, int32_t depth, double* rgb);
void orc_primary_rays(int32_t w, int32_t h, const double* cam, double* org, double* dir);
void orc_primary_rays_aa(int32_t w, int32_t h, int32_t aadepth, const double* cam, double* org, double* dir);
/* full pixel loop (main.cpp:735-861): rgb = w*h*3 doubles, ray_count = closest-hit queries issued */
void orc_render(const orc_scene*, const orc_params*, int32_t w, int32_t h, int32_t aadepth, const double* cam,
                int32_t row_begin, int32_t row_end, int32_t threads, double* rgb, int64_t* ray_count);
void orc_quantize(int64_t n_px, const double* rgb, uint8_t* bgr);
int32_t orc_save_bmp(const char* path, int32_t w, int32_t h, int32_t dpi, const double* rgb);

#ifdef __cplusplus
}
#endif
#endif
