// TEST INFRASTRUCTURE ONLY — never shipped, never linked into the product.
//
// Driver around the *real* reference renderer.  It is compiled by
// oracle/Makefile together with the reference's own translation units
// (engine/renderer.cpp, engine/bdpt.cpp, core/*.c) from where they lie under
// /root/reference; nothing of the reference is copied.  It only builds where
// /root/reference exists and is used to (a) generate the golden fixtures under
// tests/golden/ and (b) time the reference CPU renderer.
//
// Sub-commands (all binary outputs are little-endian, doubles are IEEE-754):
//   dump   <scene.json>                       loader tables as JSON on stdout
//   bvh    <scene.json> <out.bin> [tris.bin]  BVH exactly as Scene::build_bvh makes it
//   hits   <scene.json> <rays.bin> <out.bin> [tris.bin]
//   render <scene.json> <W> <H> <spp> <depth> <seed> <nee> <mis> <mnee> <threads> <out.bin>
//   kat    <scene.json> <seed> <n> <out.bin>
//   texels <scene.json> <out_prefix>
//
// Reference entry points exercised: SceneLoader::load (engine/scene_loader.hpp:98),
// Scene::hit (engine/scene.hpp:574), BVH::build/hit (engine/bvh.hpp:148,161),
// Renderer::render (engine/renderer.cpp:263), Scene::sample_light/light_pdf/
// env_light_pdf (engine/scene.hpp:280,407,470), EnvironmentMap::sample/pdf
// (engine/texture.hpp:487,634), Material::get_albedo/reflectance
// (engine/material.hpp:71,308), vec3_refract (core/vec3.c:69).

#include <omp.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "scene_loader.hpp"  // reference header, found through -I/root/reference/engine

namespace raytracer {
// Defined (non-static) in engine/renderer.cpp:33-34.
extern thread_local std::mt19937 rng;
}  // namespace raytracer

using namespace raytracer;

namespace {

struct HitOut {
  int32_t hit, prim_type, prim_index, material_id, front_face, pad;
  double t, p[3], n[3], u, v;
};
static_assert(sizeof(HitOut) == 96, "HitOut layout");

enum { T_SPHERE = 0, T_PLANE = 1, T_TRIANGLE = 2, T_BOX = 3, T_QUAD = 4, T_DISK = 5 };

std::vector<double> read_doubles(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<double> v(n / 8);
  if (fread(v.data(), 8, v.size(), f) != v.size()) exit(2);
  fclose(f);
  return v;
}

// tris.bin: N records of 10 doubles: v0 v1 v2 (9) + material id (as double).
void append_triangles(Scene& scene, const char* path) {
  auto v = read_doubles(path);
  size_t n = v.size() / 10;
  for (size_t i = 0; i < n; ++i) {
    const double* t = &v[i * 10];
    scene.add_triangle({t[0], t[1], t[2]}, {t[3], t[4], t[5]}, {t[6], t[7], t[8]}, (int)t[9]);
  }
}

// Same primitive list, in the same order, as Scene::build_bvh (engine/scene.hpp:496-533);
// Scene keeps its own BVH private, so the harness builds a second one through the
// public BVH::build and checks below that both give identical closest hits.
BVH build_public_bvh(const Scene& s) {
  std::vector<BVHPrimitive> prims;
  for (size_t i = 0; i < s.spheres.size(); ++i) {
    BVHPrimitive p; p.type = PRIM_SPHERE; p.index = (int)i;
    vec3 rad = {s.spheres[i].radius, s.spheres[i].radius, s.spheres[i].radius};
    p.bounds.min_pt = vec3_sub(s.spheres[i].center, rad);
    p.bounds.max_pt = vec3_add(s.spheres[i].center, rad);
    prims.push_back(p);
  }
  for (size_t i = 0; i < s.triangles.size(); ++i) {
    BVHPrimitive p; p.type = PRIM_TRIANGLE; p.index = (int)i;
    p.bounds.expand(s.triangles[i].v0); p.bounds.expand(s.triangles[i].v1); p.bounds.expand(s.triangles[i].v2);
    prims.push_back(p);
  }
  for (size_t i = 0; i < s.boxes.size(); ++i) {
    BVHPrimitive p; p.type = PRIM_BOX; p.index = (int)i;
    p.bounds.min_pt = s.boxes[i].box_min; p.bounds.max_pt = s.boxes[i].box_max;
    prims.push_back(p);
  }
  BVH b;
  b.build(std::move(prims));
  return b;
}

void jnum(double d) { printf("%.17g", d); }
void jvec(vec3 v) { printf("[%.17g,%.17g,%.17g]", v.x, v.y, v.z); }

int cmd_dump(const char* path) {
  // Loader chatter goes to stdout too; the JSON document is the last line, prefixed.
  auto d = SceneLoader::load(path);
  const Scene& s = d.scene;
  printf("@@JSON@@{");
  printf("\"materials\":[");
  for (size_t i = 0; i < s.materials.size(); ++i) {
    const Material& m = s.materials[i];
    printf("%s{\"type\":%d,\"albedo\":", i ? "," : "", (int)m.type); jvec(m.albedo);
    printf(",\"emission\":"); jvec(m.emission);
    printf(",\"fuzz\":"); jnum(m.fuzz); printf(",\"ior\":"); jnum(m.refraction_index);
    printf(",\"has_texture\":%d,\"tex_type\":%d,\"tex_c1\":", (int)m.has_texture, (int)m.texture.type); jvec(m.texture.color1);
    printf(",\"tex_c2\":"); jvec(m.texture.color2); printf(",\"tex_scale\":"); jnum(m.texture.scale);
    printf(",\"tex_w\":%d,\"tex_h\":%d,\"has_normal_map\":%d", m.texture.image_width, m.texture.image_height, (int)m.has_normal_map);
    printf(",\"nmap_w\":%d,\"nmap_h\":%d,\"nmap_strength\":", m.normal_map.width, m.normal_map.height); jnum(m.normal_map.strength); printf("}");
  }
  printf("],\"spheres\":[");
  for (size_t i = 0; i < s.spheres.size(); ++i) {
    printf("%s{\"center\":", i ? "," : ""); jvec(s.spheres[i].center);
    printf(",\"radius\":"); jnum(s.spheres[i].radius); printf(",\"material\":%d}", s.spheres[i].material_id);
  }
  printf("],\"planes\":[");
  for (size_t i = 0; i < s.planes.size(); ++i) {
    printf("%s{\"point\":", i ? "," : ""); jvec(s.planes[i].point);
    printf(",\"normal\":"); jvec(s.planes[i].normal); printf(",\"material\":%d}", s.planes[i].material_id);
  }
  printf("],\"boxes\":[");
  for (size_t i = 0; i < s.boxes.size(); ++i) {
    printf("%s{\"min\":", i ? "," : ""); jvec(s.boxes[i].box_min);
    printf(",\"max\":"); jvec(s.boxes[i].box_max); printf(",\"material\":%d}", s.boxes[i].material_id);
  }
  printf("],\"triangles\":[");
  for (size_t i = 0; i < s.triangles.size(); ++i) {
    printf("%s{\"v0\":", i ? "," : ""); jvec(s.triangles[i].v0); printf(",\"v1\":"); jvec(s.triangles[i].v1);
    printf(",\"v2\":"); jvec(s.triangles[i].v2); printf(",\"material\":%d}", s.triangles[i].material_id);
  }
  printf("],\"point_lights\":[");
  for (size_t i = 0; i < s.lights.size(); ++i) {
    printf("%s{\"position\":", i ? "," : ""); jvec(s.lights[i].position);
    printf(",\"intensity\":"); jvec(s.lights[i].intensity); printf("}");
  }
  printf("],\"quad_lights\":[");
  for (size_t i = 0; i < s.quad_lights.size(); ++i) {
    const QuadLight& q = s.quad_lights[i];
    printf("%s{\"corner\":", i ? "," : ""); jvec(q.corner); printf(",\"edge_u\":"); jvec(q.edge_u);
    printf(",\"edge_v\":"); jvec(q.edge_v); printf(",\"normal\":"); jvec(q.normal);
    printf(",\"emission\":"); jvec(q.emission); printf(",\"area\":"); jnum(q.area); printf("}");
  }
  printf("],\"disk_lights\":[");
  for (size_t i = 0; i < s.disk_lights.size(); ++i) {
    const DiskLight& q = s.disk_lights[i];
    printf("%s{\"center\":", i ? "," : ""); jvec(q.center); printf(",\"normal\":"); jvec(q.normal);
    printf(",\"radius\":"); jnum(q.radius); printf(",\"emission\":"); jvec(q.emission);
    printf(",\"area\":"); jnum(q.area); printf(",\"tangent\":"); jvec(q.tangent);
    printf(",\"bitangent\":"); jvec(q.bitangent); printf("}");
  }
  printf("],\"camera\":{\"position\":"); jvec(d.camera_position); printf(",\"target\":"); jvec(d.camera_target);
  printf(",\"up\":"); jvec(d.camera_up); printf(",\"fov\":"); jnum(d.camera_fov);
  Camera cam(d.camera_position, d.camera_target, d.camera_up, d.camera_fov, (double)d.width / d.height);
  printf(",\"origin\":"); jvec(cam.origin); printf(",\"lower_left_corner\":"); jvec(cam.lower_left_corner);
  printf(",\"horizontal\":"); jvec(cam.horizontal); printf(",\"vertical\":"); jvec(cam.vertical);
  printf("},\"render\":{\"width\":%d,\"height\":%d,\"samples\":%d,\"max_depth\":%d,\"mode\":%d,\"nee\":%d,\"mis\":%d,\"tone_mapper\":%d,\"exposure\":",
         d.width, d.height, d.samples, d.max_depth, (int)d.mode, (int)d.use_nee, (int)d.use_mis, (int)d.tone_mapper);
  jnum(d.exposure);
  printf(",\"output\":\"%s\"},\"background\":{\"color\":", d.output_file.c_str()); jvec(d.background_color);
  printf(",\"top\":"); jvec(d.background_top); printf(",\"bottom\":"); jvec(d.background_bottom);
  printf(",\"gradient\":%d}", (int)d.use_background_gradient);
  if (s.environment) {
    const EnvironmentMap& e = *s.environment;
    // FNV-1a over the float texels pins the decoded image without shipping it.
    uint64_t h = 1469598103934665603ull;
    const unsigned char* b = (const unsigned char*)e.hdr_data->data();
    for (size_t i = 0; i < e.hdr_data->size() * 4; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    printf(",\"environment\":{\"width\":%d,\"height\":%d,\"intensity\":", e.width, e.height); jnum(e.intensity);
    printf(",\"rotation\":"); jnum(e.rotation); printf(",\"total_luminance\":"); jnum(e.total_luminance);
    printf(",\"texel_fnv1a\":\"%016llx\"}", (unsigned long long)h);
  }
  printf("}\n");
  return 0;
}

int cmd_bvh(const char* scene_path, const char* out, const char* tris) {
  auto d = SceneLoader::load(scene_path);
  if (tris) append_triangles(d.scene, tris);
  BVH b = build_public_bvh(d.scene);
  FILE* f = fopen(out, "wb");
  int64_t hdr[2] = {(int64_t)b.nodes.size(), (int64_t)b.primitives.size()};
  fwrite(hdr, 8, 2, f);
  for (const BVHNode& n : b.nodes) {
    double box[6] = {n.bounds.min_pt.x, n.bounds.min_pt.y, n.bounds.min_pt.z, n.bounds.max_pt.x, n.bounds.max_pt.y, n.bounds.max_pt.z};
    int32_t ii[4] = {n.left, n.right, n.prim_offset, n.prim_count};
    fwrite(box, 8, 6, f); fwrite(ii, 4, 4, f);
  }
  for (const BVHPrimitive& p : b.primitives) { int32_t ii[2] = {p.type, p.index}; fwrite(ii, 4, 2, f); }
  fclose(f);
  return 0;
}

int cmd_hits(const char* scene_path, const char* rays_path, const char* out, const char* tris) {
  auto d = SceneLoader::load(scene_path);
  Scene& s = d.scene;
  if (tris) append_triangles(s, tris);
  s.build_bvh();
  BVH pub = build_public_bvh(s);
  auto rays = read_doubles(rays_path);
  size_t n = rays.size() / 8;
  std::vector<HitOut> res(n);
  long mismatches = 0;
#pragma omp parallel for schedule(static) reduction(+ : mismatches)
  for (size_t i = 0; i < n; ++i) {
    const double* q = &rays[i * 8];
    ray r = ray_create({q[0], q[1], q[2]}, {q[3], q[4], q[5]});
    double tmin = q[6], tmax = q[7];
    // (1) the real thing
    hit_record rec = {};
    bool hit = s.hit(r, tmin, tmax, rec);
    // (2) attribution: same test order through the public pieces, remembering who won
    int ptype = -1, pidx = -1;
    hit_record rec2 = {};
    double closest = tmax;
    bool hit2 = false;
    BVH::HitFunc fn = [&](int type, int index, ray rr, double a, double b, hit_record& out_rec) -> bool {
      bool h = false;
      if (type == PRIM_SPHERE) h = s.spheres[index].hit(rr, a, b, out_rec);
      else if (type == PRIM_TRIANGLE) h = s.triangles[index].hit(rr, a, b, out_rec);
      else if (type == PRIM_BOX) h = s.boxes[index].hit(rr, a, b, out_rec);
      if (h) { ptype = type; pidx = index; }
      return h;
    };
    if (pub.hit(r, tmin, closest, rec2, fn)) { hit2 = true; closest = rec2.t; }
    hit_record tmp = {};
    for (size_t k = 0; k < s.planes.size(); ++k)
      if (s.planes[k].hit(r, tmin, closest, tmp)) { hit2 = true; closest = tmp.t; rec2 = tmp; ptype = T_PLANE; pidx = (int)k; }
    for (size_t k = 0; k < s.quad_lights.size(); ++k)
      if (s.quad_lights[k].hit(r, tmin, closest, tmp)) { hit2 = true; closest = tmp.t; rec2 = tmp; rec2.material_id = -(int)k - 1; ptype = T_QUAD; pidx = (int)k; }
    for (size_t k = 0; k < s.disk_lights.size(); ++k)
      if (s.disk_lights[k].hit(r, tmin, closest, tmp)) { hit2 = true; closest = tmp.t; rec2 = tmp; rec2.material_id = -1000 - (int)k; ptype = T_DISK; pidx = (int)k; }
    if (hit != hit2 || (hit && (memcmp(&rec.t, &rec2.t, 8) != 0 || rec.material_id != rec2.material_id ||
                                memcmp(&rec.normal, &rec2.normal, 24) != 0)))
      mismatches++;
    HitOut& o = res[i];
    memset(&o, 0, sizeof(o));
    o.hit = hit; o.prim_type = hit ? ptype : -1; o.prim_index = hit ? pidx : -1;
    if (hit) {
      o.material_id = rec.material_id; o.front_face = rec.front_face; o.t = rec.t;
      o.p[0] = rec.point.x; o.p[1] = rec.point.y; o.p[2] = rec.point.z;
      o.n[0] = rec.normal.x; o.n[1] = rec.normal.y; o.n[2] = rec.normal.z;
      o.u = rec.u; o.v = rec.v;
    }
  }
  if (mismatches) { fprintf(stderr, "attribution pass disagrees with Scene::hit on %ld rays\n", mismatches); return 3; }
  FILE* f = fopen(out, "wb");
  fwrite(res.data(), sizeof(HitOut), n, f);
  fclose(f);
  return 0;
}

int cmd_render(int argc, char** argv) {
  if (argc < 13) { fprintf(stderr, "render: bad args\n"); return 2; }
  const char* scene_path = argv[2];
  int W = atoi(argv[3]), H = atoi(argv[4]), spp = atoi(argv[5]), depth = atoi(argv[6]);
  unsigned seed = (unsigned)strtoul(argv[7], nullptr, 10);
  int nee = atoi(argv[8]), mis = atoi(argv[9]), mnee = atoi(argv[10]), threads = atoi(argv[11]);
  const char* out = argv[12];
  auto d = SceneLoader::load(scene_path);
  Renderer::Settings st;
  st.width = W; st.height = H; st.samples_per_pixel = spp; st.max_depth = depth;
  const int mode_arg = argc > 13 ? atoi(argv[13]) : 1;   // optional 12th value: RenderMode
  st.mode = mode_arg == 0 ? RenderMode::Whitted : mode_arg == 2 ? RenderMode::BDPT : mode_arg == 6 ? RenderMode::PathPhoton : RenderMode::PathTrace;
  // PathPhoton: optional caustic photon count, gather count, gather radius (Renderer::Settings defaults otherwise)
  if (argc > 14) st.caustic_photon_count = (size_t)strtoull(argv[14], nullptr, 10);
  if (argc > 15) st.photon_gather_count = atoi(argv[15]);
  if (argc > 16) st.caustic_gather_radius = (float)atof(argv[16]);
  // The photon pass aims every caustic photon with a generator seeded by the hash of the calling thread's id
  // (engine/photon_integrator.hpp:457-459); it runs on this thread, so this is the seed it will use.
  printf("@@AIMSEED@@ %llu\n", (unsigned long long)std::hash<std::thread::id>{}(std::this_thread::get_id()));
  st.use_nee = nee; st.use_mis = mis; st.use_mnee = mnee;
  st.tone_mapper = ToneMapper::None;
  // ToneMapper::None still clamps to [0,1] (engine/renderer.cpp:175-182).  Scaling by an
  // exact power of two beforehand keeps every radiance below 2^20 unclamped and loses
  // no bits; the harness undoes it on output.
  const double k = 1048576.0;
  st.exposure = 1.0 / k;
  st.background_color = d.background_color; st.background_top = d.background_top;
  st.background_bottom = d.background_bottom; st.use_background_gradient = d.use_background_gradient;
  omp_set_dynamic(0);
  omp_set_num_threads(threads);
#pragma omp parallel
  { raytracer::rng.seed(seed + (unsigned)omp_get_thread_num()); }
  Camera cam(d.camera_position, d.camera_target, d.camera_up, d.camera_fov, (double)W / H);
  Renderer renderer(st);
  auto t0 = std::chrono::steady_clock::now();
  Image img = renderer.render(d.scene, cam);
  double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  FILE* f = fopen(out, "wb");
  int32_t hdr[2] = {W, H};
  fwrite(hdr, 4, 2, f);
  for (const color& c : img.pixels) { double v[3] = {c.x * k, c.y * k, c.z * k}; fwrite(v, 8, 3, f); }
  fclose(f);
  printf("\n@@TIME@@ %.6f s  %d threads  %.6f Msamples/s\n", secs, threads, (double)W * H * spp / secs * 1e-6);
  return 0;
}

// Known-answer vectors for the scalar helpers around the integrator.  The sampled
// inputs come from the reference's own generator so a CPU restatement driven by the
// same mt19937 stream can be compared bit for bit.
int cmd_kat(const char* scene_path, unsigned seed, int n, const char* out) {
  auto d = SceneLoader::load(scene_path);
  Scene& s = d.scene;
  s.build_bvh();
  raytracer::rng.seed(seed);
  std::vector<double> o;
  for (int i = 0; i < n; ++i) {
    point3 p = {random_double(-3, 3), random_double(0, 5), random_double(-5, 2)};
    LightSample ls = s.sample_light(p);
    o.push_back(ls.valid ? 1.0 : 0.0);
    if (ls.valid) {
      o.insert(o.end(), {ls.position.x, ls.position.y, ls.position.z, ls.normal.x, ls.normal.y, ls.normal.z,
                         ls.emission.x, ls.emission.y, ls.emission.z, ls.pdf, ls.distance});
      o.push_back(s.light_pdf(p, ls.position));
    }
    vec3 dir = {random_double(-1, 1), random_double(-1, 1), random_double(-1, 1)};
    o.push_back(s.env_light_pdf(dir));
    if (s.environment) {
      color c = s.environment->sample(dir);
      o.insert(o.end(), {c.x, c.y, c.z});
    }
    double uu = random_double(), vv = random_double();
    for (const Material& m : s.materials) {
      color a = m.get_albedo(p, uu, vv);
      o.insert(o.end(), {a.x, a.y, a.z});
    }
    double cosv = random_double(), ratio = (i & 1) ? 1.5 : 1.0 / 1.5;
    o.push_back(Material::reflectance(cosv, ratio));
    vec3 nrm = vec3_normalize({random_double(-1, 1), random_double(-1, 1), random_double(-1, 1)});
    vec3 ud = vec3_normalize(dir);
    vec3 rf = vec3_refract(ud, nrm, ratio);
    vec3 rl = vec3_reflect(ud, nrm);
    o.insert(o.end(), {rf.x, rf.y, rf.z, rl.x, rl.y, rl.z});
  }
  FILE* f = fopen(out, "wb");
  fwrite(o.data(), 8, o.size(), f);
  fclose(f);
  return 0;
}

// Decoded 8-bit texels of every image texture, in material order: <prefix><material id>.rgb
// preceded by two int32 (width, height).  Lets the repo pin its own image decoding.
int cmd_texels(const char* scene_path, const char* prefix) {
  auto d = SceneLoader::load(scene_path);
  for (size_t i = 0; i < d.scene.materials.size(); ++i) {
    const NormalMap& nm = d.scene.materials[i].normal_map;
    if (d.scene.materials[i].has_normal_map && nm.valid()) {   // <prefix>n<material id>.rgb
      std::string fn = std::string(prefix) + "n" + std::to_string(i) + ".rgb";
      FILE* f = fopen(fn.c_str(), "wb");
      int32_t hdr[2] = {nm.width, nm.height};
      fwrite(hdr, 4, 2, f);
      fwrite(nm.data->data(), 1, nm.data->size(), f);
      fclose(f);
    }
    const Texture& t = d.scene.materials[i].texture;
    if (!d.scene.materials[i].has_texture || !t.image_data) continue;
    std::string fn = std::string(prefix) + std::to_string(i) + ".rgb";
    FILE* f = fopen(fn.c_str(), "wb");
    int32_t hdr[2] = {t.image_width, t.image_height};
    fwrite(hdr, 4, 2, f);
    fwrite(t.image_data->data(), 1, t.image_data->size(), f);
    fclose(f);
  }
  return 0;
}

// Post-process of a film through the reference's own code: Image::apply_tone_mapping followed by
// Image::write_ppm (engine/renderer.cpp:186-230).  <film.f32> holds width*height*3 floats, row 0 = bottom.
int cmd_post(const char* film_path, int w, int h, int mapper, double exposure, const char* ppm_path) {
  std::vector<float> film((size_t)w * h * 3);
  FILE* f = fopen(film_path, "rb");
  if (!f || fread(film.data(), 4, film.size(), f) != film.size()) { fprintf(stderr, "cannot read %s\n", film_path); return 1; }
  fclose(f);
  Image img(w, h);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const float* p = &film[((size_t)y * w + x) * 3];
      img.set_pixel(x, y, color{(double)p[0], (double)p[1], (double)p[2]});
    }
  img.apply_tone_mapping((ToneMapper)mapper, exposure);
  return img.write_ppm(ppm_path) ? 0 : 1;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: ref_harness dump|bvh|hits|render|kat|texels|post ...\n"); return 2; }
  std::string cmd = argv[1];
  if (cmd == "dump") return cmd_dump(argv[2]);
  if (cmd == "bvh" && argc >= 4) return cmd_bvh(argv[2], argv[3], argc > 4 ? argv[4] : nullptr);
  if (cmd == "hits" && argc >= 5) return cmd_hits(argv[2], argv[3], argv[4], argc > 5 ? argv[5] : nullptr);
  if (cmd == "render") return cmd_render(argc, argv);
  if (cmd == "kat" && argc >= 6) return cmd_kat(argv[2], (unsigned)strtoul(argv[3], nullptr, 10), atoi(argv[4]), argv[5]);
  if (cmd == "texels" && argc >= 4) return cmd_texels(argv[2], argv[3]);
  if (cmd == "post" && argc >= 8) return cmd_post(argv[2], atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atof(argv[6]), argv[7]);
  fprintf(stderr, "bad command\n");
  return 2;
}
