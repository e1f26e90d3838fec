/*
 * rtx_hip.h — C ABI of the MI355X (gfx950) path-tracing backend.
 *
 * This is the boundary a host renderer binds instead of the reference's optional GPU
 * backend  raytracer::cuda::GPUPathTracer  (engine/cuda/cuda_api.hpp:72-110).  Unlike that
 * backend it carries the *whole* scene (triangles, boxes, quad/disk/point lights, textures,
 * environment map) and keeps geometry in double precision so that the closest-hit
 * validation entry point sees the same bits as the CPU renderer.
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no exceptions cross the boundary
 *   - every function returns 0 on success and a negative rtx_status otherwise;
 *     rtx_last_error() returns a thread-local message for the last failure
 *   - the caller owns every input array (copied during rtx_upload_scene) and every output
 *     buffer; the library owns device memory and frees it in rtx_destroy
 *   - one host thread per context; calls on one context are synchronous unless stated
 *   - film layout: RGB interleaved floats, pixel (x, y) at (y*width + x)*3, y = 0 is the
 *     BOTTOM row — the convention of Image::set_pixel / GPUPathTracer::render
 *     (engine/renderer.hpp:60-69, engine/main.cpp:394-397); values are linear averaged
 *     radiance, no exposure / tone map / gamma (engine/main.cpp:399-427 applies those)
 */
#ifndef RTX_HIP_H
#define RTX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rtx_status {
  RTX_OK = 0,
  RTX_ERR_INVALID_ARGUMENT = -1,
  RTX_ERR_NO_DEVICE = -2,
  RTX_ERR_HIP = -3,
  RTX_ERR_NOT_READY = -4,      /* render before upload_scene / set_camera */
  RTX_ERR_UNSUPPORTED = -5,    /* e.g. use_mnee with the wavefront pipeline */
  RTX_ERR_OUT_OF_MEMORY = -6
} rtx_status;

/* ---- scene description: the fields of the reference's scene objects, verbatim ---- */

/* Sphere (engine/sphere.hpp:55-61) */
typedef struct rtx_sphere { double center[3]; double radius; int32_t material_id; int32_t pad_; } rtx_sphere;
/* Plane (engine/primitives.hpp:24-30); `normal` is the stored, already normalised member */
typedef struct rtx_plane { double point[3]; double normal[3]; int32_t material_id; int32_t pad_; } rtx_plane;
/* Triangle (engine/primitives.hpp:68-73) */
typedef struct rtx_triangle { double v0[3], v1[3], v2[3]; int32_t material_id; int32_t pad_; } rtx_triangle;
/* Box (engine/primitives.hpp:134-140) */
typedef struct rtx_box { double box_min[3], box_max[3]; int32_t material_id; int32_t pad_; } rtx_box;
/* PointLight (engine/scene.hpp:45-50) */
typedef struct rtx_point_light { double position[3]; double intensity[3]; } rtx_point_light;
/* QuadLight (engine/primitives.hpp:219-234): every member, including the derived ones */
typedef struct rtx_quad_light { double corner[3], edge_u[3], edge_v[3], normal[3], emission[3]; double area; } rtx_quad_light;
/* DiskLight (engine/primitives.hpp:321-344) */
typedef struct rtx_disk_light { double center[3], normal[3]; double radius; double emission[3]; double area; double tangent[3], bitangent[3]; } rtx_disk_light;

/* MaterialType (engine/material.hpp:29-33) */
enum { RTX_MAT_LAMBERTIAN = 0, RTX_MAT_METAL = 1, RTX_MAT_DIELECTRIC = 2 };
/* TextureType (engine/texture.hpp, Texture::type) */
enum { RTX_TEX_SOLID = 0, RTX_TEX_CHECKER = 1, RTX_TEX_IMAGE = 2 };

/* Material (engine/material.hpp:48-66), the members the path tracer reads */
typedef struct rtx_material {
  int32_t type;            /* RTX_MAT_* */
  int32_t has_texture;     /* Material::has_texture */
  double albedo[3];
  double emission[3];
  double fuzz;
  double refraction_index;
  int32_t tex_type;        /* RTX_TEX_* (Texture::type) */
  int32_t tex_image;       /* index into rtx_scene_desc::images, or -1 */
  double tex_color1[3];
  double tex_color2[3];
  double tex_scale;
  /* NormalMap (engine/texture.hpp:45-49; Material::normal_map / has_normal_map, engine/material.hpp:58-60), applied by
   * apply_normal_map (engine/renderer.cpp:72-94) in both render modes.  A zero-initialised record has none. */
  int32_t has_normal_map;  /* Material::has_normal_map */
  int32_t normal_image;    /* index into rtx_scene_desc::images of the tangent-space map; read when has_normal_map */
  double normal_strength;  /* NormalMap::strength */
} rtx_material;

/* Texture image data (engine/texture.hpp:196-199): 8-bit RGB, row 0 = top */
typedef struct rtx_image { const uint8_t* rgb; int32_t width, height; } rtx_image;

/* EnvironmentMap (engine/texture.hpp:352-362): float RGB texels, row 0 = top (theta = 0) */
typedef struct rtx_environment { const float* rgb; int32_t width, height; double intensity; double rotation; /* radians */ } rtx_environment;

typedef struct rtx_scene_desc {
  const rtx_sphere* spheres;           size_t n_spheres;
  const rtx_plane* planes;             size_t n_planes;
  const rtx_triangle* triangles;       size_t n_triangles;
  const rtx_box* boxes;                size_t n_boxes;
  const rtx_point_light* point_lights; size_t n_point_lights;
  const rtx_quad_light* quad_lights;   size_t n_quad_lights;
  const rtx_disk_light* disk_lights;   size_t n_disk_lights;
  const rtx_material* materials;       size_t n_materials;
  const rtx_image* images;             size_t n_images;
  const rtx_environment* environment;  /* NULL when the scene has none */
} rtx_scene_desc;

/* Camera (engine/scene.hpp:55-59): the four stored vectors */
typedef struct rtx_camera { double origin[3], lower_left_corner[3], horizontal[3], vertical[3]; } rtx_camera;

enum { RTX_PRECISION_F32 = 0, RTX_PRECISION_F64 = 1 };
enum { RTX_PIPELINE_AUTO = 0, RTX_PIPELINE_FUSED = 1, RTX_PIPELINE_WAVEFRONT = 2 };
/* RenderMode (engine/renderer.hpp:26-34), the modes this backend implements, with the reference's values:
 * Whitted = Renderer::ray_color_whitted (engine/renderer.cpp:483-569), PathTrace = ray_color_path (:580-841),
 * BDPT = BDPTIntegrator::Li (engine/bdpt.cpp:71-128; max_eye_depth = max_light_depth = max_depth as in
 * engine/renderer.cpp:304-313), PathPhoton = the caustic photon pass (engine/renderer.cpp:344-372, engine/photon_integrator.hpp:93-293)
 * followed by Renderer::ray_color_path_with_caustics (engine/renderer.cpp:844-976).  Any other value returns RTX_ERR_UNSUPPORTED. */
enum { RTX_MODE_WHITTED = 0, RTX_MODE_PATHTRACE = 1, RTX_MODE_BDPT = 2, RTX_MODE_PATHPHOTON = 6 };
/* Counter-based generator behind the samples (the reference's per-thread mt19937 has no parallel counterpart):
 * AUTO = PCG4D for single precision, PHILOX for double; PHILOX = Philox-4x32-10 (Random123), the generator the
 * double-precision parity mode and the CPU oracle share; PCG4D = the pcg4d hash, a third of the multiplies. */
enum { RTX_RNG_AUTO = 0, RTX_RNG_PHILOX = 1, RTX_RNG_PCG4D = 2 };

/* Renderer::Settings (engine/renderer.hpp:101-133), PathTrace mode members, plus what a
 * GPU backend needs in addition (seed, precision, pipeline, row sharding). */
typedef struct rtx_settings {
  int32_t width, height;
  int32_t samples_per_pixel;
  int32_t max_depth;
  int32_t use_nee, use_mis, use_mnee;      /* use_mnee (Renderer::Settings::use_mnee, engine/renderer.hpp:101-133): fused pipeline; with RTX_PRECISION_F32 the closest-hit
                                            * queries run in single precision and the manifold walk in double, with RTX_PRECISION_F64 everything is the reference's arithmetic */
  int32_t use_background_gradient;
  double background_color[3], background_top[3], background_bottom[3];
  uint32_t seed;
  int32_t precision;                        /* RTX_PRECISION_* */
  int32_t pipeline;                         /* RTX_PIPELINE_* */
  /* Row sharding: this context renders rows y with  y % shard_count == shard_index.
   * shard_count = 1 renders the whole film. */
  int32_t shard_index, shard_count;
  int32_t mode;                             /* RTX_MODE_* (Renderer::Settings::mode); Whitted runs in the fused pipeline */
  int32_t rng;                              /* RTX_RNG_* */
  double clamp_max;                         /* Renderer::Settings::clamp_max: luminance cap per BDPT strategy, <= 0 = off (engine/bdpt.cpp:114-121) */
  /* PathPhoton mode (Renderer::Settings, engine/renderer.hpp:126-132; engine/renderer.cpp:344-372) */
  uint64_t caustic_photon_count;            /* caustic photons emitted (default 50000) */
  int32_t photon_gather_count;              /* photons gathered per estimate (default 100) */
  float caustic_gather_radius;              /* gather radius, a float in the reference too (default 0.1f) */
} rtx_settings;

/* rtx_stats::flags.  RTX_STAT_BEFORE_TREE: the frame was rendered while the reference-order tree of a large scene was still being
 * built on the host (rtx_upload_scene returns before it is done, see there): exact ties in t between two bounded primitives went by
 * primitive number, and the (pixel, sample chunk) work items they occurred in were noted.  RTX_STAT_TIES_REPAIRED: there were such
 * ties, so the call waited for the tree and rendered those work items again with the reference's tie order (the other counters and
 * times of the statistics are the first pass's; seconds_total covers everything).  RTX_STAT_RENDERED_TWICE: there were more of
 * them than the list holds (65 536), and the whole frame was rendered again.  RTX_STAT_WALKS_SIDE_BY_SIDE (queue pipeline): the
 * frame was small enough (fewer than 48 Mi path slots) for the closest-hit and the any-hit walk of every turn to run side by side on
 * two streams; seconds_extend then includes time the closest-hit kernel shared the device with the other walk. */
enum { RTX_STAT_BEFORE_TREE = 1, RTX_STAT_TIES_REPAIRED = 2, RTX_STAT_RENDERED_TWICE = 4, RTX_STAT_WALKS_SIDE_BY_SIDE = 8 };
typedef struct rtx_stats {
  uint64_t samples;          /* camera paths started */
  uint64_t rays_closest;     /* closest-hit queries executed */
  uint64_t rays_shadow;      /* shadow (any-hit) queries executed */
  uint64_t segments;         /* path vertices shaded */
  double seconds_render;     /* device time of the render kernels of the call (HIP events); PathPhoton: without the photon pass */
  double seconds_main_kernel;/* device time of the dominant kernel alone (fused: the path-tracing kernel) */
  double seconds_total;      /* wall time of the call */
  uint32_t kernel_launches;
  uint32_t pipeline_used;    /* RTX_PIPELINE_FUSED or _WAVEFRONT */
  uint32_t precision_used;   /* RTX_PRECISION_* the render ran in */
  uint32_t flags;            /* RTX_STAT_*: how the call went */
  double seconds_photon_pass;/* PathPhoton: wall time of the photon pass (trace kernel, kd-tree and box build on the device); ~0 when the context reuses the
                              * map of an earlier render (same scene upload, seed, generator, caustic_photon_count, precision); else 0.  rays_closest
                              * includes the pass's queries in both cases */
  /* queue-based pipeline only (0 otherwise): the closest-hit traversal kernel of the ray queue */
  double seconds_extend;     /* device time of its launches, summed (HIP events around each) */
  uint64_t nodes_visited;    /* node records it fetched and tested */
  uint64_t prims_tested;     /* leaf records it fetched and tested */
} rtx_stats;

/* One closest-hit result: the members of hit_record (core/hit.h:20-29) that identify the
 * hit, plus which primitive won. */
enum { RTX_PRIM_SPHERE = 0, RTX_PRIM_PLANE = 1, RTX_PRIM_TRIANGLE = 2, RTX_PRIM_BOX = 3, RTX_PRIM_QUAD_LIGHT = 4, RTX_PRIM_DISK_LIGHT = 5 };
typedef struct rtx_hit {
  int32_t hit, prim_type, prim_index, material_id, front_face, pad_;
  double t, point[3], normal[3], u, v;
} rtx_hit;

typedef struct rtx_ctx rtx_ctx;

/* Replaces raytracer::cuda::test_cuda() (engine/cuda/cuda_api.hpp:17, cuda_test.cu:25-81):
 * runs a trivial kernel on the device and checks its output. */
int rtx_selftest(int device_id);

/* Replaces GPUPathTracer::GPUPathTracer() / ~GPUPathTracer() (cuda_api.hpp:74-75). */
int rtx_create(rtx_ctx** out_ctx, int device_id);
int rtx_destroy(rtx_ctx* ctx);

/* One context over several GPUs of the node, for a single-process host such as the reference's C++ caller, whose GPU branch
 * (engine/main.cpp:362-435) constructs one GPUPathTracer and so drives one GPU however many the machine has.  The context
 * behaves like a single-device one: rtx_upload_scene builds the reference-order tree once on the host and gives every device its
 * own copy of the scene (each over its own host link, in parallel); rtx_set_camera applies to all; rtx_render and
 * rtx_render_device deal the rows of the film (or of the caller's shard) to the devices in turn (rtx_multi_shard), render them
 * side by side, and gather them into the caller's buffer, which for rtx_render_device is memory of device_ids[0].  The film
 * equals the single-device film bit for bit.  rtx_trace_closest*, rtx_postprocess* and rtx_get_bvh_info run on device_ids[0];
 * rtx_render_device_noalloc is refused.  A device may be listed more than once (it then holds one scene copy and renders one
 * share per listing): that is meant for testing the path on a machine with fewer GPUs.  rtx_destroy frees all of it. */
int rtx_create_multi(rtx_ctx** out_ctx, const int* device_ids, int n_devices);

/* The share of device `device_rank` of `n_devices` in a render of shard (shard_index, shard_count), as a shard of its own:
 * *out_index of *out_count.  Packed row j of the caller's shard belongs to device j % n_devices, and is that device's packed row
 * j / n_devices. */
int rtx_multi_shard(int shard_index, int shard_count, int device_rank, int n_devices, int* out_index, int* out_count);

/* How a multi-device context brings the devices' rows into the film of rtx_render_device (device memory of device_ids[0]):
 * RTX_GATHER_PEER_COPY (default): one hipMemcpyPeerAsync per device into a staging buffer on the first device, ordered behind that
 * device's render by an event, then one strided copy there; RTX_GATHER_RCCL: grouped ncclSend / ncclRecv into the first device over
 * one communicator per device (ncclCommInitAll), RCCL being loaded with dlopen("librccl.so.1") on this call, so the library itself
 * does not depend on it.  The film is the same bit for bit.  RCCL needs every device listed once; a list with repeats is refused
 * (RTX_ERR_UNSUPPORTED), a missing library too.  Host-memory output (rtx_render) is copied by every device over its own host
 * link in both modes. */
enum { RTX_GATHER_PEER_COPY = 0, RTX_GATHER_RCCL = 1 };
int rtx_multi_set_gather(rtx_ctx* ctx, int mode);

/* Replaces GPUPathTracer::upload_scene (cuda_api.hpp:84-89; cuda_pathtracer.cu:234-318).
 * Copies the description, builds both BVHs (the reference-order binary tree of
 * engine/bvh.hpp:148-218 for validation, and the wide one the integrator traverses) and
 * the environment-map sampling tables (engine/texture.hpp:403-473).
 * Scenes of 2^18 or more bounded primitives: the call returns when the tables and the wide tree are on the device(s); the
 * reference-order tree is finished by a host thread of the library's own (it reads nothing of the caller's any more).  The
 * single-precision path tracer needs that tree only to decide exact ties in t the reference's way, so it renders at once: the work
 * items of a frame in which such ties occur are rendered again when the tree is there (rtx_stats::flags), every film is the one the
 * finished tree gives.  Whatever walks the reference-order tree itself - rtx_trace_closest*, double precision, the other render modes,
 * rtx_get_bvh_info, rtx_get_reference_bvh - waits for it, and so do rtx_render_device_async (its film is complete in stream order)
 * and rtx_render_device_noalloc (a frame ahead of the tree needs a tie list and a second pass).  A scene in which a primitive's box is flat along an axis is built the
 * ordinary way (the reference cannot reach some such primitives, which the wide tree has to know), as is every scene when the
 * environment variable RTX_EAGER_TREE is set. */
int rtx_upload_scene(rtx_ctx* ctx, const rtx_scene_desc* scene);

/* Replaces GPUPathTracer::set_camera (cuda_api.hpp:99).  Takes the vectors of an already
 * constructed Camera, so there is no ordering constraint against the settings. */
int rtx_set_camera(rtx_ctx* ctx, const rtx_camera* camera);

/* Replaces GPUPathTracer::set_settings + GPUPathTracer::render (cuda_api.hpp:94,105;
 * cuda_pathtracer.cu:320-379) and, on the CPU side, the PathTrace branch of
 * Renderer::render (engine/renderer.cpp:378-449).  Synchronous.  `out_rgb` is host memory,
 * width*height*3 floats for shard_count == 1, otherwise rows_of_shard*width*3 floats with the
 * shard's rows packed bottom-up.  `stats` may be NULL. */
int rtx_render(rtx_ctx* ctx, const rtx_settings* settings, float* out_rgb, rtx_stats* stats);

/* Same, but `d_out_rgb` is DEVICE memory on the context's device and the work is enqueued
 * on `hip_stream` (a hipStream_t, NULL = the context's own stream).  Returns after the
 * kernels have finished (stats need the device counters). */
int rtx_render_device(rtx_ctx* ctx, const rtx_settings* settings, void* d_out_rgb, void* hip_stream, rtx_stats* stats);

/* rtx_render_device in two halves, for callers that want the host back while the GPU renders (a second scene on another context,
 * a film gather of the previous frame).  It has no counterpart in the reference, whose GPUPathTracer::render is synchronous
 * (cuda_pathtracer.cu:375).  rtx_render_device_async puts the render on `hip_stream` - the caller's stream, taken as given: NULL
 * is HIP's default stream here, not the context's own, so that the caller can order its work behind the render - and
 * returns without waiting; the film is complete where the stream stands at that moment (for that reason the first such call after
 * the upload of a large scene waits for the reference-order tree, see rtx_upload_scene: it never renders ahead of it).  The fused
 * pipeline enqueues and returns at once; the queue-based pipeline reads its queue lengths back between kernels, so the call
 * returns when its last kernels are enqueued.  rtx_render_wait waits for that render and fills `stats` (may be NULL).  A context
 * holds ONE render in flight: any other render call on it first waits for the one pending.  Single-device contexts only. */
int rtx_render_device_async(rtx_ctx* ctx, const rtx_settings* settings, void* d_out_rgb, void* hip_stream);
int rtx_render_wait(rtx_ctx* ctx, rtx_stats* stats);

/* rtx_render_device for callers that cannot allocate device memory while they run: code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference.  It allocates nothing (no hipMalloc /
 * hipFree, no allocating library routine): the chunk sums and counters of the render live in a 64 MiB array that is part of the
 * library's code object on every GPU (held until the process ends), and the kernels are the ones rtx_render_device launches,
 * so the film is the same bit for bit.  Limits, each refused with a message: single precision, RTX_MODE_PATHTRACE or
 * RTX_MODE_WHITTED, the fused pipeline (RTX_PIPELINE_AUTO must resolve to it: scenes of at most 1024 bounded primitives; larger
 * scenes with pipeline = RTX_PIPELINE_FUSED), use_mnee = 0, and
 * rows_of_shard * width * chunks * 12 bytes <= 64 MiB - 256, chunks = min(samples_per_pixel, 16) (e.g. 640x360 at any sample
 * count, 1920x1080 at 2 samples per pixel).  With stats = NULL the call only enqueues kernels on `hip_stream` (NULL = the context's
 * stream) and returns - no event, no copy, no memset node: the form a caller records into a HIP graph; with stats it also
 * times the kernels with events, waits for them and fills the statistics.  Every call on one GPU uses that GPU's one
 * array: calls that overlap in time on the same GPU (other contexts, other streams) race; calls on one stream do not. */
int rtx_render_device_noalloc(rtx_ctx* ctx, const rtx_settings* settings, void* d_out_rgb, void* hip_stream, rtx_stats* stats);

/* Replaces Image::apply_tone_mapping followed by the quantisation of Image::write_png / write_ppm
 * (engine/renderer.cpp:186-198 with the operators of :102-184, then :232-258; called from main.cpp:399-431 on the
 * GPU branch and renderer.cpp after the CPU render): per channel  c = film * exposure;  tone map;  clamp to [0,1];
 * sqrt;  (int)(255.999 * v);  rows flipped so that row 0 of `out_rgb8` is the TOP of the image.
 * tone_mapper follows enum class ToneMapper (engine/renderer.hpp:39-45).  Computed in double precision in the
 * reference's operation order: the bytes equal the reference's for the same float film.  A NaN channel gives 0.
 * `film_rgb`: width*height*3 floats, row 0 = bottom (what rtx_render returns); both buffers are host memory. */
enum { RTX_TONE_NONE = 0, RTX_TONE_REINHARD = 1, RTX_TONE_REINHARD_EXTENDED = 2, RTX_TONE_ACES = 3, RTX_TONE_UNCHARTED2 = 4 };
int rtx_postprocess(rtx_ctx* ctx, const float* film_rgb, int width, int height, int tone_mapper, double exposure, uint8_t* out_rgb8);

/* Same on DEVICE buffers of the context's device, enqueued on `hip_stream` (NULL = the context's stream); returns
 * after the kernel has finished. */
int rtx_postprocess_device(rtx_ctx* ctx, const void* d_film_rgb, int width, int height, int tone_mapper, double exposure, void* d_out_rgb8,
                           void* hip_stream);

/* Closest-hit validation entry point: the GPU counterpart of Scene::hit
 * (engine/scene.hpp:574-674) in double precision on the reference-order BVH.
 * rays8: n records of 8 doubles (origin xyz, direction xyz, t_min, t_max), host memory. */
int rtx_trace_closest(rtx_ctx* ctx, const double* rays8, size_t n, rtx_hit* out_hits);

/* Same query through the wide BVH and float/double kernels the integrator uses
 * (precision = RTX_PRECISION_*); for measuring ID agreement of the fast path. */
int rtx_trace_closest_fast(rtx_ctx* ctx, const double* rays8, size_t n, int precision, rtx_hit* out_hits);

/* Number of rows a shard owns: |{ y in [0,height) : y % shard_count == shard_index }|. */
int rtx_shard_rows(int height, int shard_index, int shard_count);

/* BVH statistics of the uploaded scene (0 when there is no bounded primitive). */
typedef struct rtx_bvh_info {
  uint64_t ref_nodes, ref_prims;   /* the reference-order binary tree (validation entry point, double precision, tie ranks) */
  uint64_t wide_nodes, wide_bytes; /* node records of the eight-wide tree the single-precision integrator walks, and their total size */
  double build_seconds;            /* host time of the reference-order build */
  uint64_t leaf_bytes;             /* size of one leaf record of the eight-wide tree (one primitive) */
  uint64_t wide_leaves;            /* its leaf records */
  uint64_t wide_levels;            /* its depth in node levels */
  double wide_build_seconds;       /* time of its build on the GPU */
} rtx_bvh_info;
int rtx_get_bvh_info(rtx_ctx* ctx, rtx_bvh_info* out);

/* Validation: the reference-order tree as this library built it and as rtx_trace_closest and the double-precision mode walk it -
 * the node array and the primitive list BVH::build leaves behind (engine/bvh.hpp:148-218; a node is BVHNode, bvh.hpp:124-130:
 * its box, `left` / `right` child or `first` / `count` of a leaf; a primitive is its type, RTX_PRIM_*, and its index within that
 * type).  Copied back from the device.  `nodes` holds rtx_bvh_info::ref_nodes entries, prim_types / prim_indices ref_prims each. */
typedef struct rtx_bvh_node {
  double lo[3], hi[3];
  int32_t left, right, first, count;
} rtx_bvh_node;
int rtx_get_reference_bvh(rtx_ctx* ctx, rtx_bvh_node* nodes, size_t n_nodes, int32_t* prim_types, int32_t* prim_indices, size_t n_prims);

const char* rtx_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RTX_HIP_H */
