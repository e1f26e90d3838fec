// Device-side scene layout shared by the host uploader and the gfx950 kernels.
//
// Everything is templated on the arithmetic type R: double instantiations hold exactly the
// bits the caller passed through rtx_scene_desc (validation / parity mode), float
// instantiations are the production layout.
#pragma once

#include <stdint.h>

namespace rtx {

template <class R> struct DSphere { R c[3]; R radius; int32_t mat; int32_t pad; };
template <class R> struct DPlane { R p[3]; R n[3]; int32_t mat; float pd; };   // pd = n . p, read by the flat single-precision test
template <class R> struct DTri { R v0[3], v1[3], v2[3]; int32_t mat; int32_t pad; };
template <class R> struct DBox { R lo[3], hi[3]; int32_t mat; int32_t pad; };
template <class R> struct DPoint { R pos[3], I[3]; };
template <class R> struct DQuad { R corner[3], eu[3], ev[3], n[3], Le[3]; R area; };
template <class R> struct DDisk { R c[3], n[3]; R radius; R Le[3]; R area; R tan[3], bit[3]; };

template <class R> struct DMaterial {
  int32_t type, has_texture, tex_type, tex_image;
  int32_t nmap_image, pad;   // tangent-space normal map: index into DScene::images, or -1
  R albedo[3], emission[3], fuzz, ior, c1[3], c2[3], scale, nmap_strength;
};

struct DImage { const uint8_t* rgb; int32_t w, h; };

// Reference-order binary BVH node: the members of BVHNode (engine/bvh.hpp:124-130).
struct DNodeRef { double lo[3], hi[3]; int32_t left, right, first, count; };
struct DPrimRef { int32_t type, index; };

// Single-precision traversal layout: an eight-wide tree built on the GPU (bvh_build.hip) from the Morton order of the
// primitives' centroids.  Its topology is its own (closest hits do not depend on it: boxes are conservative, and a tie in t
// is resolved by the primitives' positions in the REFERENCE's leaf order, DScene::rank); only the reference-order tree above
// decides which primitives can be hit at all (a primitive whose reference leaf box is flat along an axis is never reached
// there and is built into this tree as unhittable).
//
// One node record is 64 bytes, four 16-byte loads: the boxes of up to eight children as 8-bit offsets from the node's own origin
// in units of 2^e per axis, rounded outward (Ylitie, Karras, Laine: "Efficient Incoherent Ray Traversal on GPUs Through
// Compressed Wide BVHs", HPG 2017).  The origin is itself a point of a 2^21-step grid over the scene's bounds
// (DScene::wt_anchor + u * DScene::wt_step per axis, evaluated with one fmaf exactly as the builder evaluated it), three 21-bit
// numbers in two words.  Node and leaf records share one array of 64-byte units.  The records of a node's children form one
// block: child k's record is unit inner_base + k; children 0 .. n_inner-1 are node records, children n_inner .. n_child-1 leaf
// records.
// Unused slots hold planes (255, 0), an interval no ray enters (the traversal picks near and far planes by the ray's
// direction, it does not order them).
struct alignas(64) DNode8 {
  uint32_t org[2];        // x: bits 0-20 of org[0]; y: bits 21-31 of org[0] (low 11) and bits 0-9 of org[1] (high 10); z: bits 10-30 of org[1]
  uint32_t exps;          // byte a (0..2): biased float exponent of axis a's unit; bits 24-27: n_inner; bits 28-31: n_child - 1
  int32_t inner_base;
  uint32_t qlo[3][2];     // byte (k & 3) of qlo[a][k >> 2]: child k's lower plane on axis a
  uint32_t qhi[3][2];
};
static_assert(sizeof(DNode8) == 64, "node record");
constexpr int kNodeOriginBits = 21;
// One leaf record is one unit: ONE primitive.  ref >= 0: a triangle in place (v = its first vertex, then the edges v1 - v0 and v2 - v0), ref = the primitive's number in
// the reference's primitive list (spheres, then triangles, then boxes: engine/scene.hpp:496-533); ref == -1: nothing to test (a
// primitive the reference can never hit); ref <= -2: a sphere or a box, primitive number -2 - ref, tested through its table.  A
// record without a triangle has NaN vertices, which fail the triangle test.
struct alignas(64) DLeaf {
  float v[9];
  int32_t ref;
  int32_t pad[6];
};
static_assert(sizeof(DLeaf) == 64, "leaf record");
constexpr int kLeafMaxPrims = 1;
constexpr int kWalkDone = 0x7fffffff;
// Words of stack a walk of a tree of `levels` node levels can need: one two-word entry per level, and one to spare.
constexpr int walk_stack_slots(int levels) { return 2 * (levels + 1); }
// Deepest eight-wide tree the kernels walk: its stack (80 words per lane, 80 KB for a 256-lane workgroup) is what the launch code
// checks against the device's LDS per workgroup before every launch; a scene over that limit is refused with a message.  A soup of
// 10 M triangles builds 8 levels; cuts at Morton code bits consume at least one bit of the 63 per level before ranges of equal
// codes are halved, so only adversarial inputs get anywhere near.
constexpr int kMaxWalkLevels = 39;
// Shading record of one primitive of a flat scene, in the order the flat intersection tests them (spheres in the
// reference's leaf order, planes, quad lights, disk lights): what the winner's normal and material need, fetched
// once per hit instead of being carried through every test.  sphere: (a,b,c) = centre, w = 1 / radius; others:
// (a,b,c) = outward normal, w = 0.
struct alignas(16) DFlatRec { float a, b, c, w; int32_t mat, type, idx, pad; };

// Guides of the importance sampler's two distribution functions: entry k = the first index whose value exceeds k / kEnvGuide, so a
// number in [k / kEnvGuide, (k + 1) / kEnvGuide) has its upper bound between entries k and k + 1 and the bisection starts from there
// instead of from the whole row (the same index, in three steps instead of ten).
constexpr int kEnvGuide = 256;
template <class R> struct DEnv {
  const float* texel;        // RGB, row 0 = top
  const R* marginal;         // h+1
  const R* conditional;      // h*(w+1)
  const int32_t* marginal_guide;      // kEnvGuide + 1
  const int32_t* conditional_guide;   // h * (kEnvGuide + 1)
  int32_t w, h;
  R intensity, rotation;
  int32_t valid, importance; // EnvironmentMap::valid() / has_importance_sampling()
};

template <class R> struct DScene {
  const DSphere<R>* spheres; const DPlane<R>* planes; const DTri<R>* tris; const DBox<R>* boxes;
  const DPoint<R>* points; const DQuad<R>* quads; const DDisk<R>* disks; const DMaterial<R>* mats;
  const DImage* images; const int32_t* emissive;   // indices of emissive spheres, ascending
  const DNodeRef* nodes; const DPrimRef* prims;     // double precision: the reference-order tree, walked nearest child first (left child of node i = i + 1); the reference's primitive order
  const DNode8* nodes8;                                // single-precision traversal layout: 64-byte units, node records and leaf records (NULL in the double scene and for scenes tested linearly)
  const int32_t* rank;                                 // position of every bounded primitive (numbered as in DLeaf::ref) in the reference's leaf order; NULL while the
                                                       // reference-order tree is still being built on the host: exact ties in t then go by primitive number, and the
                                                       // (pixel, chunk) work item they occur in is noted for a second look (DRenderParams::tie_list)
  const DFlatRec* flat;                                // flat scenes only (single precision)
  float bb_lo[3], bb_scale[3];                         // the eight-wide tree's root box as a 128^3 grid: cell = (p - bb_lo) * bb_scale (ray sorting)
  float wt_anchor[3], wt_step[3];                      // the grid the node origins of the eight-wide tree lie on (DNode8::org)
  int32_t n_spheres, n_planes, n_tris, n_boxes, n_points, n_quads, n_disks, n_mats, n_images, n_emissive;
  int32_t n_nodes, n_prims, n_nodes8, n_leaves8, walk_levels;
  int32_t n_nmaps;                                     // materials with a normal map (picks the kernel instances that apply them)
  DEnv<R> env;
};

struct DSceneRef {  // validation BVH, always double
  const DNodeRef* nodes; const DPrimRef* prims; int32_t n_nodes, n_prims;
};

template <class R> struct DCamera { R origin[3], llc[3], horizontal[3], vertical[3]; };

struct DRenderParams {
  int32_t width, height, spp, max_depth;
  int32_t use_nee, use_mis, use_gradient;
  int32_t shard_index, shard_count, rows;  // rows = rows owned by this shard
  int32_t n_chunks, chunk_spp;             // samples are summed in fixed chunks (deterministic film)
  uint32_t seed;
  int32_t rng;                              // RNG_PHILOX / RNG_PCG4D (pt_device.hpp)
  double bg[3], bg_top[3], bg_bottom[3];
  double clamp_max;                         // BDPT: luminance cap per strategy, <= 0 off
  // PathPhoton: the caustic photon map of this render (pt_photon.hpp), DPhotonNode<R> / DPhotonData<R> arrays, and the
  // bounding boxes of its photons (DPhotonBox<R>): one per run of 64 photons, then one per group of 64 runs
  const void* ph_nodes; const void* ph_data; const void* ph_boxes;
  int32_t n_photons, gather_k;
  int32_t ph_bucket, ph_cand_cap;           // a kd subtree of at most ph_bucket photons is searched through its boxes, not walked
  // gather_k above kPhotonMaxGather: the candidate lists of the density estimate live in device memory instead of LDS, two lists
  // of ph_cand_cap entries (distances, then photon numbers) per wave of the grid; 0 / NULL otherwise
  void* ph_cand;
  double gather_radius;
  // Work items to render instead of all of them (NULL: all): the repair pass of a frame that was rendered before the
  // reference-order tree arrived re-renders the items in which an exact tie in t occurred.  item index i -> item_list[i].
  const uint32_t* item_list;
  uint32_t n_list;
  // Where a render without DScene::rank notes such items: tie_list[k] for the k-th tie, k = DCounters::ties++, while k < tie_cap.
  uint32_t* tie_list;
  uint32_t tie_cap;
};

// PathPhoton (pt_photon.hpp)
// Default of DRenderParams::ph_bucket: 64 groups of 64 runs of 64 photons.  Measured on the 1080p Cornell box with 50 000 photons
// (16 spp): walking down to 64 photons 896 ms, 1024: 470, 16384: 235, no walk: 156.
constexpr int kPhotonBucket = 64 * 64 * 64;
constexpr unsigned long long kPhotonMaxCount = 1ull << 26;   // caustic photons per render (photon numbers and tree positions are 32-bit; 5 GB of double-precision records)
constexpr int kPhotonMaxGather = 128;   // largest photon_gather_count whose candidate list fits the LDS of a wave; larger counts keep the list in device memory
struct DPhotonLight { int32_t sphere; uint32_t first, count; int32_t pad; double power[3]; };   // photons [first, first + count) leave this emissive sphere
template <class R> struct DPhotonNode { R p[3]; int32_t axis; };       // kd-tree in the reference's implicit layout: the median of a range sits at its middle
template <class R> struct DPhotonData { R dir[3]; R power[3]; };
template <class R> struct DPhotonBox { R lo[3]; R hi[3]; };

// ties: closest-hit queries with an exact tie in t between two bounded primitives that was decided WITHOUT the reference's leaf
// order (DScene::rank == NULL); the work items they belong to are in DRenderParams::tie_list
struct DCounters { unsigned long long samples, closest, shadow, segments, nodes_visited, prims_tested, ties; };

// One closest-hit answer in the layout of rtx_hit (include/rtx_hip.h).
struct DHitOut {
  int32_t hit, prim_type, prim_index, material_id, front_face, pad;
  double t, point[3], normal[3], u, v;
};

constexpr int kBlock = 256;          // threads per workgroup of the render / trace kernels
constexpr int kLinearMaxPrims = 12;  // at or below this many bounded primitives: no tree walk
constexpr int kStackDepth = 48;      // per-lane traversal stack entries (LDS)

// How a kernel instance intersects (see pt_device.hpp).
enum { MODE_TREE = 0, MODE_TREE_UV = 1, MODE_FLAT = 2 };
// Integrator a fused kernel instance runs: ray_color_path, ray_color_path with MNEE, ray_color_whitted.
enum { INTEG_PATH = 0, INTEG_PATH_MNEE = 1, INTEG_WHITTED = 2, INTEG_BDPT = 3, INTEG_PATHPHOTON = 4 };
constexpr int kBdptStoreFields = 18;   // values kept per connectible light-subpath vertex (pt_bdpt.hpp)
constexpr int kFlatMaxSpheres = 16;   // flat mode: sphere-only bounded primitives, at most this many
constexpr size_t kMaxLdsMaterialBytes = 32768;

enum { PRIM_SPHERE = 0, PRIM_PLANE = 1, PRIM_TRIANGLE = 2, PRIM_BOX = 3, PRIM_QUAD = 4, PRIM_DISK = 5 };
enum { MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_DIELECTRIC = 2 };
enum { TEX_SOLID = 0, TEX_CHECKER = 1, TEX_IMAGE = 2 };

}  // namespace rtx
