/* rt_amd.h -- C ABI of the MI355X (gfx950) ray-trace hot path.
 *
 * The reference renderer (rossig7/RayTracer) has no FFI: its boundary is the C++ object API that
 * main.cpp is written against.  This header is the thin C layer the host-side C++ classes of
 * this project (raytracer_amd/scene/ -- same class and method names as the reference) call to reach
 * the HIP kernels; each entry point names the reference interface it replaces (file:line relative
 * to the reference tree).
 *
 * Conventions
 *   - every function returns RT_OK (0) or a negative rt_status; rt_last_error() gives the text of
 *     the last failure on the calling thread.  No exception crosses this boundary.
 *   - plain pointers and sizes only.  Host pointers unless the parameter name starts with `d_`
 *     (device memory of the context's GPU).
 *   - one rt_context per GPU, driven by one host thread at a time; distinct contexts are
 *     independent.  A context never falls back to the CPU: every entry point that computes fails
 *     with RT_ERR_NO_DEVICE if no gfx950 device is usable.
 *   - object ids are indices in insertion order (the reference's scene_objects order, main.cpp:46);
 *     a miss is id -1 (the reference returns nullptr, BVH.cpp:115,122).
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: rt_stats gained tile_order (a caller built against version 1 passes a structure that is too small) */
#define RT_ABI_VERSION 2

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,      /* bad argument (null pointer, size mismatch, unknown enum)      */
    RT_ERR_NO_DEVICE = -2,    /* no usable gfx950 device / HIP runtime error at context creation */
    RT_ERR_HIP = -3,          /* a HIP call or kernel launch failed                              */
    RT_ERR_STATE = -4,        /* call order violated (e.g. trace before scene upload)            */
    RT_ERR_IO = -5,           /* file could not be read / written                                */
    RT_ERR_LIMIT = -6,        /* a compiled-in limit was exceeded (e.g. BVH deeper than the traversal stack) */
    RT_ERR_NOMEM = -7         /* a host allocation failed                                         */
} rt_status;

/* arithmetic the kernels run in.  RT_F64 reproduces the reference bit for bit (hit ids, t);
 * RT_F32 is the FP32-VALU throughput instantiation of the same kernel source. */
typedef enum rt_precision { RT_F64 = 0, RT_F32 = 1 } rt_precision;

typedef struct rt_context rt_context;
typedef struct rt_scene rt_scene;

/* Camera exactly as main.cpp:626-635 builds it (all double): position, direction, right, down. */
typedef struct rt_camera {
    double pos[3];
    double dir[3];
    double right[3];
    double down[3];
} rt_camera;

/* Run-time form of the reference's compile-time constants (main.cpp:21-39, Kdtree.cpp:6) and of
 * the locals of main() (main.cpp:605-614).  rt_default_params() fills the reference's values. */
typedef struct rt_render_params {
    uint32_t struct_size;     /* sizeof(rt_render_params), checked                               */
    int32_t width, height;    /* main.cpp:606-607                                                */
    int32_t aadepth;          /* main.cpp:610; aadepth^2 samples, reference sample pattern       */
    int32_t tracing_depth;    /* TRACING_DEPTH, main.cpp:28                                      */
    int32_t photon_num;       /* PHOTONMUM, main.cpp:21 (normalisation of the radiance estimate) */
    int32_t photon_use;       /* PHOTONUSE, main.cpp:22 (k of the k-NN gather)                   */
    int32_t head_quirk;       /* 1: recurse with the *old* hit object as HEAD does (main.cpp:198,249) */
    int32_t use_photon_map;   /* 0: diffuse hits shade black (no gather)                         */
    int32_t precision;        /* rt_precision of traversal + shading                             */
    int32_t shadow_rays;      /* extension (not in the reference): 1 = one occlusion ray per primary hit toward light_pos, the occluder's id (-1: lit, -2: no
                               * primary hit) written to d_shadow_ids; 2 = the same ray, and with d_shadow_ids NULL an occluded hit gets bit 30 set in its d_ids entry
                               * (one word per pixel to move between GPUs).  That form is an occlusion query over the same segment, walked FROM the light
                               * (origin light_pos, direction -L, the same range) and ended at the first occluder */
    int32_t count_work;       /* 1: also count inner-node visits and primitive tests (slower instrumented kernel) */
    /* Row interleave for multi-GPU tiling: with band_rows > 0 the call covers only the bands
     * b with b % n_parts == part (band b = image rows [b*band_rows, (b+1)*band_rows)); row_begin /
     * row_end and the output buffers are then in this part's compact row space [0, height/n_parts).
     * height must be a multiple of band_rows * n_parts.  band_rows = 0: contiguous rows. */
    int32_t band_rows, n_parts, part;
    double accuracy;          /* main.cpp:614                                                    */
    double photon_dist;       /* PHOTONDIST, Kdtree.cpp:6                                        */
    double light_pos[3];      /* main.cpp:654                                                    */
    /* Depth of field, main.cpp:35-37, 792-810: dof_samples rays per sample through a jittered lens position
     * toward the focal point (1 = off, the reference's setting).  Glossy materials (special = x.xx7654,
     * main.cpp:177) take 16 jittered reflection / refraction rays whatever these fields say.  Every draw
     * is a function of (pixel or hit point, draw number, rng_seed), not of a shared generator. */
    double focal_len;         /* FOCAL_LEN 2.6                                                   */
    double aperture;          /* APERTURE_SIZE 0.2                                               */
    int32_t dof_samples;      /* DOF_SAMPLE 1                                                    */
    uint32_t rng_seed;
} rt_render_params;

/* Ray counters of one call (the Mrays/s metric counts closest-hit queries, i.e. BVH::Shoot calls). */
typedef struct rt_stats {
    uint64_t primary_rays;
    uint64_t secondary_rays;  /* reflection + refraction */
    uint64_t shadow_rays;
    uint64_t gather_queries;  /* KDTree::findKNN calls */
    uint64_t inner_visits;    /* inner BVH nodes entered (only with count_work) */
    uint64_t prim_tests;      /* findIntersection calls (only with count_work) */
    double kernel_ms;         /* device time of the call measured with HIP events on the context's stream */
    double trace_kernel_ms;   /* device time of the traversal kernel launches only */
    uint64_t gather_single;   /* of gather_queries: answered one per wave (subsurface hits, groups the grouped form handed over) */
    uint64_t tile_order;      /* rt_trace_primary*: the order this launch took the frame's tiles in -- 1: the plain sequence, 2: from the
                               * frame's two halves in turn, 3 (single precision): two tiles per wave; the library measures them on the
                               * first launches of a view that repeats and keeps the fastest (same frames either way).  0 from the
                               * other entry points */
} rt_stats;

/* ---- process / context ------------------------------------------------------------------- */
int rt_abi_version(void);
const char* rt_last_error(void);
int rt_device_count(void);
int rt_context_create(int device, rt_context** out);
void rt_context_destroy(rt_context* ctx);
/* the HIP stream (hipStream_t) all work of this context is enqueued on */
void* rt_context_stream(rt_context* ctx);
/* Waits for the context's stream.  The *_dev entry points called without `stats` do not wait for their
 * kernels; if one of those kernels ran out of traversal stack or queue room (its output is then incomplete)
 * the context remembers it and this call returns RT_ERR_LIMIT, once.  Calls that do wait report it themselves. */
int rt_context_synchronize(rt_context* ctx);
void rt_default_params(rt_render_params* p);
/* camera basis from position + look-at, main.cpp:626-635 */
void rt_camera_look_at(const double pos[3], const double look_at[3], rt_camera* out);

/* ---- host scene assembly: replaces `vector<Object*> scene_objects` + `new BVH(scene_objects)`
 *      (main.cpp:46,660-678).  Argument orders are the reference constructors'. ------------------ */
int rt_scene_create(rt_scene** out);
void rt_scene_destroy(rt_scene* s);
int rt_scene_size(const rt_scene* s);
/* Triangle(A, C, B, color, ior), Triangle.h:137; returns the object id or <0 */
int rt_scene_add_triangle(rt_scene* s, const double A[3], const double C[3], const double B[3], const double rgbs[4], float ior);
/* Triangle::setNormals(a, c, b), Triangle.h:150 */
int rt_scene_set_normals(rt_scene* s, int obj, const double na[3], const double nc[3], const double nb[3]);
/* Triangle::setTexture(tA, tC, tB, use, map), Triangle.h:106 */
int rt_scene_set_texture(rt_scene* s, int obj, const double ta[2], const double tc[2], const double tb[2], int use_texture, int map);
/* Sphere(center, radius, color, ior), Sphere.h:100 */
int rt_scene_add_sphere(rt_scene* s, const double center[3], double radius, const double rgbs[4], float ior);
/* Plane(normal, distance, color), Plane.h:48.  Planes have no bounding box: they are kept in a
 * side list tested after the BVH with the same accept rule (the reference cannot instantiate Plane). */
int rt_scene_add_plane(rt_scene* s, const double normal[3], double distance, const double rgbs[4]);
/* makeCornellBox / makeCube, IO.cpp:108 / IO.cpp:70 */
int rt_scene_add_cornell_box(rt_scene* s, const double corner1[3], const double corner2[3]);
int rt_scene_add_cube(rt_scene* s, const double corner1[3], const double corner2[3], const double rgbs[4]);
/* ObjReader(file, color, ior, xoff, yoff, zoff, map, use_texture)::ReadContent, IO.h:34,61.
 * Tolerates faces without vt / vn indices (the reference crashes on them, IO.h:91-101). */
int rt_scene_load_obj(rt_scene* s, const char* path, const double rgbs[4], float ior, float xoff, float yoff, float zoff,
                      int use_texture, int map);
/* TextureMap::TextureMapRead, TextureMap.h:20; returns the map index or <0 */
int rt_scene_add_texture_bmp(rt_scene* s, const char* path);
/* rows x cols texels, rgb doubles in [0,1], row-major as TextureMap::ColorMap[x][y] */
int rt_scene_add_texture_rgb(rt_scene* s, int x_res, int y_res, const double* rgb);
/* BVH::BVH(vector<Object*>&), BVH.cpp:98: same topology, leaf order and boxes as the reference */
int rt_scene_build_bvh(rt_scene* s);
/* topology in pre-order for tests: rec = per node [is_leaf, n, ids...], boxes = 6 doubles per node.
 * Call with NULL buffers to get the sizes. */
int rt_scene_bvh_dump(const rt_scene* s, int32_t* rec, int64_t* n_rec, double* boxes, int64_t* n_boxes);
int rt_scene_bvh_depth(const rt_scene* s);
/* read a triangle back: points = A,B,C (as stored after the (A,C,B) constructor order), normals =
 * NA,NB,NC, uvs = tA,tB,tC; flags bit 0: setNormals was called, bit 1: setTexture was called.
 * Lets ObjReader::ReadContent (IO.h:61) hand API objects to the caller from the library's parser. */
int rt_scene_get_triangle(const rt_scene* s, int obj, double points[9], double normals[9], double uvs[6], int* flags);

/* Photon emission, main.cpp:396-598 + the loop of main.cpp:697-718, on the host (it is RNG-order
 * dependent): n_emit photons from light_pos, one thread, std::mt19937 seeded with `seed` (the
 * reference seeds from std::random_device).  bounce = BOUNCE (main.cpp:25).  Needs the BVH.  The
 * stored photons stay in the scene: which = 0 surface map (main.cpp:44), 1 volume map (main.cpp:45);
 * rt_scene_photons_get copies n x 9 doubles (position, power rgb, direction; Photon.h:7-28). */
int rt_scene_emit_photons(rt_scene* s, int64_t n_emit, const double light_pos[3], uint32_t seed, int bounce, double accuracy);
/* The same emission on the GPU (one photon per lane through the closest-hit kernel's traversal):
 * every photon owns a counter-based random stream keyed by (seed, photon number), so the photon
 * list has the reference's distribution but not its serial std::mt19937 order -- the reference's
 * own multi-threaded emission (main.cpp:689-720, one generator shared by all threads without
 * synchronisation) is not reproducible either.  `s` must be the scene last uploaded to `ctx`.
 * Photons are appended to the scene's lists in (photon number, bounce) order, independent of the
 * launch.  stats (optional): secondary_rays = Shoot calls, kernel_ms. */
int rt_scene_emit_photons_gpu(rt_context* ctx, rt_scene* s, int64_t n_emit, const double light_pos[3], uint32_t seed, int bounce,
                              double accuracy, rt_stats* stats);
int64_t rt_scene_photon_count(const rt_scene* s, int which);
int rt_scene_photons_get(const rt_scene* s, int which, double* photons);
void rt_scene_photons_clear(rt_scene* s);

/* ---- device: scene + traversal ----------------------------------------------------------- */
/* copy the flattened scene (64 B-aligned node / primitive / material arrays, f64 and f32 forms) to HBM */
int rt_scene_upload(rt_context* ctx, const rt_scene* s);
/* The same upload with the tree built on the GPU instead of taking the host tree of
 * rt_scene_build_bvh (which need not have been called): a linear BVH over Morton-sorted
 * primitives, milliseconds instead of seconds for millions of triangles.  NOT the reference's
 * tree (BVH.cpp:57-96): closest hits agree with it except where the reference's own box culling
 * decides (hits within rounding of a box face) and on exact ties, so this form is checked against
 * exhaustive intersection, not against the reference's goldens.  build_ms (optional): device time
 * of the build. */
int rt_scene_upload_lbvh(rt_context* ctx, const rt_scene* s, double* build_ms);

/* A built scene as one block of bytes, and back.  The reference builds its tree once, before its render threads fan
 * out over the image (main.cpp:678, 735); with one process per GPU the process that built the tree hands the others
 * this image instead of each parsing the meshes and building the same tree again.  The image holds the per-object
 * shading records, the inner nodes, the leaf-ordered primitive records, the plane list, texture descriptors and
 * texels as plain records behind a header with a magic word, a version and the record sizes; it contains no
 * pointers and may be sent between processes of the same build as it is (photons are not part of it).
 *   rt_scene_export_flat: buf NULL -> *bytes = size needed; otherwise writes the image (buf 64-byte aligned, *bytes
 *                         its capacity on entry, the image's size on return).  The tree must have been built.
 *   rt_scene_import_flat: replaces the contents of `s` by the image's scene, tree included (rt_scene_bvh_dump,
 *                         rt_scene_upload, photon emission then work as on the scene that was exported).
 *   rt_scene_upload_flat: rt_scene_upload straight from an image, without an rt_scene in between.
 * Both readers check the image before they trust it -- sizes, section bounds, every child, primitive, object and
 * texture index, and that inner children follow their parent (no cycle) -- and return RT_ERR_INVALID with the reason
 * otherwise; buf must be 64-byte aligned. */
int rt_scene_export_flat(const rt_scene* s, void* buf, size_t* bytes);
int rt_scene_import_flat(rt_scene* s, const void* buf, size_t bytes);
int rt_scene_upload_flat(rt_context* ctx, const void* buf, size_t bytes);

/* BVH::Shoot(ray, dist) for a batch, BVH.cpp:108.  org/dir: n x 3 doubles; dist: in = max range
 * (the reference's in/out `distance`), out = hit distance (unchanged on a miss); ids: object id or -1.
 * Host buffers (copied through PCIe, not the measured path). */
int rt_trace_closest(rt_context* ctx, int precision, int64_t n, const double* org, const double* dir, double* dist, int32_t* ids);
/* same with device-resident SoA rays: d_org/d_dir point to 3*n doubles laid out x[n],y[n],z[n];
 * d_dist n doubles in/out, d_ids n int32.  Enqueued on the context stream, not synchronised. */
int rt_trace_closest_dev(rt_context* ctx, int precision, int64_t n, const double* d_org, const double* d_dir, double* d_dist, int32_t* d_ids);

/* Primary rays of the pixel loop generated on the device (main.cpp:753-790, aadepth 1) + BVH::Shoot
 * (main.cpp:812-813) for image rows [row_begin,row_end); optionally one shadow (occlusion) ray per
 * hit toward light_pos.  d_ids / d_t / d_shadow_ids are device arrays indexed by
 * (y-row_begin)*width+x; d_t and d_shadow_ids may be NULL.  Enqueued, not synchronised. */
int rt_trace_primary_dev(rt_context* ctx, const rt_camera* cam, const rt_render_params* p, int row_begin, int row_end,
                         int32_t* d_ids, double* d_t, int32_t* d_shadow_ids, rt_stats* stats);
/* host-buffer convenience form of the above (synchronises) */
int rt_trace_primary(rt_context* ctx, const rt_camera* cam, const rt_render_params* p, int row_begin, int row_end,
                     int32_t* ids, double* t, int32_t* shadow_ids, rt_stats* stats);

/* ---- device: photon map -------------------------------------------------------------------- */
/* KDTree(vector<Photon*>&), Kdtree.h:59 + Kdtree.cpp:11-61: builds the reference's kd-tree on the
 * host (same topology), flattens it and copies it to HBM.  photons: n x 9 doubles
 * (position, power rgb, direction; Photon.h:7-28).  which: 0 surface map, 1 volume map. */
int rt_photons_upload(rt_context* ctx, int which, int64_t n, const double* photons);
/* The same with the kd-tree built on the GPU (Kdtree.cpp:11-52 level by level: widest axis of each
 * range from its bounds, median by a sort of the whole array keyed by (range, coordinate)).  Same
 * split rule on the float positions the gather reads; photons of equal coordinate may fall on
 * either side of a pivot, which no k-NN result depends on. */
int rt_photons_upload_dev_build(rt_context* ctx, int which, int64_t n, const double* photons);
/* main.cpp:683-727 without leaving the GPU: emission as rt_scene_emit_photons_gpu (same random
 * streams, so the same photons), the lists put in (photon number, bounce) order, both kd-trees built
 * on the device.  The scene must have been uploaded.  counts (optional): photons in the surface and
 * volume maps.  stats (optional): secondary_rays = Shoot calls, trace_kernel_ms = emission,
 * kernel_ms = emission + ordering + tree builds. */
int rt_emit_photon_maps_dev(rt_context* ctx, int64_t n_emit, const double light_pos[3], uint32_t seed, int bounce, double accuracy,
                            int64_t counts[2], rt_stats* stats);
/* photons in map `which` of the context; rt_photon_map_get copies the n x 9 doubles of a map built on
 * the device (by the two calls above), in list order */
int64_t rt_photon_map_size(const rt_context* ctx, int which);
int rt_photon_map_get(rt_context* ctx, int which, double* photons);
/* KDTree::findKNN(k, center), Kdtree.cpp:135, for a batch: counts[n], and (optional) the squared
 * distance of the farthest returned photon r2[n] (-1 when none). */
int rt_knn_radius(rt_context* ctx, int which, int k, double max_dist, int64_t n, const double* centers, int32_t* counts, double* r2);
/* KDTree::findKNN returning the set itself: idx is n x k indices into the uploaded photon array
 * (-1 padded, unordered); counts[n]. */
int rt_knn_query(rt_context* ctx, int which, int k, double max_dist, int64_t n, const double* centers, int32_t* idx, int32_t* counts);
/* findKNN + the radiance estimate of main.cpp:260-295 for a batch of diffuse hits:
 * pos/normal n x 3, color n x 3 (hit colour), rgb out n x 3 (clipped as Color::clip) */
int rt_knn_estimate(rt_context* ctx, const rt_render_params* p, int64_t n, const double* pos, const double* normal,
                    const double* color, double* rgb, int32_t* counts);

/* The same estimate through the form the pixel path uses: requests are taken in groups of 64
 * consecutive ones (as one shading wave queues them) and answered one per lane over a photon list
 * shared by the group; requests a group cannot take fall to the per-request form above.  Same
 * k-nearest sets, sums in another order. */
int rt_knn_estimate_grouped(rt_context* ctx, const rt_render_params* p, int64_t n, const double* pos, const double* normal,
                            const double* color, double* rgb);

/* No counterpart in the reference.  rt_knn_estimate_grouped for callers that cannot allocate device memory -- code
 * under HIP graph capture, tight serving loops, harnesses that forbid it: requests and results live in the caller's
 * device buffers (d_pos / d_normal / d_color: n x 3 doubles; d_rgb: n x 3 doubles out), and every byte of working
 * memory (request queue, hand-over list, packed photon lists of 512 waves, counters) is a fixed array inside the library, about 41 MB on each GPU it
 * runs on, held until the process ends.  The call makes no hipMalloc / hipFree; it is enqueued on the context's
 * stream and not synchronised.  n <= RT_NOALLOC_MAX_REQUESTS, more is refused with RT_ERR_LIMIT.  Every call on a
 * GPU shares that GPU's array, whichever context it comes through: two calls overlapping in time on one GPU would
 * race on it, so the library puts them in order -- a call's kernels start only when those of the call before it on
 * that GPU have finished (a stream wait on an event, no host wait), and the enqueue itself is under a mutex.  Calls
 * through different contexts of one GPU are therefore safe, and never run side by side. */
#define RT_NOALLOC_MAX_REQUESTS 65536
int rt_knn_estimate_dev_noalloc(rt_context* ctx, const rt_render_params* p, int64_t n, const double* d_pos, const double* d_normal,
                                const double* d_color, double* d_rgb);

/* ---- device: full pixel path ------------------------------------------------------------- */
/* The pixel loop of main.cpp:735-861 for rows [row_begin,row_end): primary rays, BVH::Shoot,
 * getColorAt (main.cpp:158-392: reflection / refraction bounce queue + photon gather) and the
 * quantiser of saveBmp (IO.cpp:55-65).  d_rgb: (rows*width*3) doubles (may be NULL);
 * d_bgr8: (rows*width*3) bytes B,G,R (may be NULL).  Enqueued, not synchronised -- except for scenes
 * whose rays no queue size bounds (a glossy material: 16 + 16 rays per hit, main.cpp:177-257; or
 * tracing_depth > 4): those are rendered in parts sized by what the ray queues hold, each part waited
 * for, a part that overflows being split (rows, then columns, down to one 8 x 8 tile) and rendered again. */
int rt_render_rows_dev(rt_context* ctx, const rt_camera* cam, const rt_render_params* p, int row_begin, int row_end,
                       double* d_rgb, uint8_t* d_bgr8, rt_stats* stats);
int rt_render_rows(rt_context* ctx, const rt_camera* cam, const rt_render_params* p, int row_begin, int row_end,
                   double* rgb, uint8_t* bgr8, rt_stats* stats);

/* ---- device: several GPUs in one process ------------------------------------------------- */
/* Frame assembly over the GPUs of ONE process.  The reference's threads write column strips of one host frame
 * (main.cpp:735-861); here part r of n_parts (rt_render_params.band_rows / n_parts / part) is traced by ctxs[r] into
 * the compact device buffer d_parts[r] (height / n_parts rows of row_bytes), and this call copies every band to its
 * place in d_frame (height rows of row_bytes, on the device of ctxs[0]): one strided device-to-device copy per part,
 * enqueued behind that part's kernels, direct over xGMI where the devices can reach each other.  Returns when the frame
 * is complete.  (Processes with one GPU each gather over RCCL instead, see raytracer_amd/tiling.py.) */
int rt_gather_rows(rt_context* const* ctxs, const void* const* d_parts, int n_parts, int band_rows, int height, size_t row_bytes,
                   void* d_frame);

/* ---- device: one process per GPU ----------------------------------------------------------- */
/* Frame assembly over PROCESSES, one GPU each, for hosts written against this header (Python hosts use
 * torch.distributed, raytracer_amd/tiling.py; both are the join of the reference's render threads, main.cpp:862).
 * Rank r of n_ranks traces part r of the row interleave (rt_render_params.band_rows / n_parts / part) into a compact
 * device buffer; rt_gather_frame sends it to rank 0 over RCCL -- every peer over its own xGMI link, one grouped
 * send/receive, no ring -- and rank 0 puts every band in its place in d_frame.  RCCL (librccl.so.1) is loaded when the
 * first of these calls is made and by nothing else in the library; without it they return RT_ERR_STATE.
 *   rt_comm_unique_id   rank 0 makes the 128-byte id; the host hands it to the other ranks by whatever channel it has
 *   rt_comm_create      collective over the n_ranks processes; the communicator is bound to ctx's device and stream
 *   rt_gather_frame     collective; d_part: height / n_ranks rows of row_bytes (this rank's bands, compact); d_frame:
 *                       height rows of row_bytes on rank 0, ignored elsewhere; height must be a multiple of
 *                       band_rows * n_ranks.  Enqueued on ctx's stream behind the kernels that traced the part, not
 *                       synchronised; rank 0 keeps a staging buffer of (n_ranks - 1) parts between calls. */
typedef struct rt_comm rt_comm;
#define RT_COMM_ID_BYTES 128
int rt_comm_unique_id(unsigned char id[RT_COMM_ID_BYTES]);
int rt_comm_create(rt_context* ctx, const unsigned char id[RT_COMM_ID_BYTES], int rank, int n_ranks, rt_comm** out);
void rt_comm_destroy(rt_comm* comm);
int rt_gather_frame(rt_comm* comm, const void* d_part, int band_rows, int height, size_t row_bytes, void* d_frame);

/* ---- host I/O either side of the path ---------------------------------------------------- */

/* saveBmp(filename, w, h, dpi, data), IO.cpp:5: byte-identical file (header quirks included) */
int rt_save_bmp(const char* path, int w, int h, int dpi, const double* rgb);
int rt_save_bmp_bgr8(const char* path, int w, int h, int dpi, const uint8_t* bgr);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
