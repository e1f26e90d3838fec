/* TEST INFRASTRUCTURE ONLY — CPU restatement of the reference path tracer's hot path.
 *
 * Nothing in the product (ray-tracer_amd/, include/) may call into this library; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, as the checker.
 *
 * It takes the same scene/camera/settings structures as the GPU library (rtx_hip.h is only
 * a set of plain structs) so one description can be handed to both sides.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include "../include/rtx_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

enum { ORC_RNG_MT19937 = 0,  /* std::mt19937 drawn in the reference's order (engine/renderer.cpp:33-56) */
       ORC_RNG_COUNTER = 1,  /* Philox-4x32-10 keyed by (pixel, seed): one block per path vertex, the GPU's RTX_RNG_PHILOX */
       ORC_RNG_COUNTER_PCG = 2 /* the same block layout filled by the pcg4d hash, the GPU's RTX_RNG_PCG4D */ };

int  orc_create(const rtx_scene_desc* desc, orc_scene** out);
void orc_destroy(orc_scene* s);

/* Reference-order BVH (engine/bvh.hpp:148-218 via engine/scene.hpp:493-546). */
int orc_bvh_counts(const orc_scene* s, int64_t* n_nodes, int64_t* n_prims);
int orc_bvh_dump(const orc_scene* s, double* boxes6, int32_t* node_ints4, int32_t* prim_ints2);

/* Scene::hit (engine/scene.hpp:574-674) on n rays of 8 doubles each. */
int orc_trace_closest(const orc_scene* s, const double* rays8, size_t n, rtx_hit* out, int threads);

/* Renderer::render, PathTrace branch (engine/renderer.cpp:378-449) + ray_color_path
 * (engine/renderer.cpp:580-841), MNEE (engine/spectral/mnee.hpp) included when settings->use_mnee is set.  `film` receives width*height*3
 * UNCLAMPED linear doubles, pixel (x, y) at (y*width + x)*3, y = 0 at the bottom.
 * With ORC_RNG_MT19937 and threads = 1 the film equals the reference's bit for bit when
 * the reference generator is seeded with settings->seed. */
int orc_render(const orc_scene* s, const rtx_camera* cam, const rtx_settings* st, int rng_mode, int threads,
               double* film, rtx_stats* stats);

/* Scalar known-answer stream; mirrors `ref_harness kat` draw for draw. */
int orc_kat(const orc_scene* s, unsigned seed, int n, double* out, size_t cap, size_t* written);

/* Image::apply_tone_mapping + the quantisation of Image::write_png (engine/renderer.cpp:102-198, :232-258):
 * exposure, tone map (ToneMapper enum order: 0 None, 1 Reinhard, 2 ReinhardExtended, 3 ACES, 4 Uncharted2), clamp,
 * sqrt gamma, int(255.999 * v), vertical flip.  `film`: width*height*3 floats, row 0 = bottom;
 * `rgb8`: width*height*3 bytes, row 0 = TOP. */
int orc_postprocess(const float* film, int width, int height, int tone_mapper, double exposure, uint8_t* rgb8);

/* One block of the pcg4d generator for (pixel, sample, block, seed), exposed for statistical tests. */
void orc_pcg4d(uint32_t pixel, uint32_t sample, uint32_t block, uint32_t seed, uint32_t out[4]);
/* PathPhoton, ORC_RNG_MT19937 only: the seed the reference's thread-id-seeded aim generator had in the run to reproduce
 * (engine/photon_integrator.hpp:457-459; printed by oracle/_ref/ref_harness as @@AIMSEED@@). */
void orc_set_photon_aim_seed(unsigned long long seed);

/* The counter-based generator, exposed so tests can pin it (Random123 known answers). */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif
