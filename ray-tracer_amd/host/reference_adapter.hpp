// Adapter from the reference's own scene objects to the C ABI structures: the code a
// maintainer adds next to  main.cpp:286-435  (the `--cuda-render` branch) to drive this backend.
//
// Written as templates over the reference's types so that this header does not include
// reference headers itself; it is instantiated with raytracer::Scene / raytracer::Camera /
// raytracer::Renderer::Settings (ref: engine/scene.hpp:120-134, :55-59; engine/renderer.hpp:101-133).
// Every member is passed through unchanged (doubles stay doubles): the GPU side sees the bits
// the CPU renderer uses.
#pragma once

#include <vector>

#include "../../include/rtx_hip.h"

namespace raytracer {
namespace hip {

struct SceneBuffers {   // owns the arrays an rtx_scene_desc points into
  std::vector<rtx_sphere> spheres;
  std::vector<rtx_plane> planes;
  std::vector<rtx_triangle> triangles;
  std::vector<rtx_box> boxes;
  std::vector<rtx_point_light> point_lights;
  std::vector<rtx_quad_light> quad_lights;
  std::vector<rtx_disk_light> disk_lights;
  std::vector<rtx_material> materials;
  std::vector<rtx_image> images;
  rtx_environment environment{};
  rtx_scene_desc desc{};
};

template <class V> inline void put3(double* dst, const V& v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }

// `scene` must outlive `out` (texture and environment texels are referenced, not copied).
template <class SceneT> inline const rtx_scene_desc& to_scene_desc(const SceneT& scene, SceneBuffers& out) {
  out = SceneBuffers();
  for (const auto& s : scene.spheres) { rtx_sphere d{}; put3(d.center, s.center); d.radius = s.radius; d.material_id = s.material_id; out.spheres.push_back(d); }
  for (const auto& p : scene.planes) { rtx_plane d{}; put3(d.point, p.point); put3(d.normal, p.normal); d.material_id = p.material_id; out.planes.push_back(d); }
  for (const auto& t : scene.triangles) { rtx_triangle d{}; put3(d.v0, t.v0); put3(d.v1, t.v1); put3(d.v2, t.v2); d.material_id = t.material_id; out.triangles.push_back(d); }
  for (const auto& b : scene.boxes) { rtx_box d{}; put3(d.box_min, b.box_min); put3(d.box_max, b.box_max); d.material_id = b.material_id; out.boxes.push_back(d); }
  for (const auto& l : scene.lights) { rtx_point_light d{}; put3(d.position, l.position); put3(d.intensity, l.intensity); out.point_lights.push_back(d); }
  for (const auto& q : scene.quad_lights) {
    rtx_quad_light d{}; put3(d.corner, q.corner); put3(d.edge_u, q.edge_u); put3(d.edge_v, q.edge_v); put3(d.normal, q.normal);
    put3(d.emission, q.emission); d.area = q.area; out.quad_lights.push_back(d);
  }
  for (const auto& q : scene.disk_lights) {
    rtx_disk_light d{}; put3(d.center, q.center); put3(d.normal, q.normal); d.radius = q.radius; put3(d.emission, q.emission);
    d.area = q.area; put3(d.tangent, q.tangent); put3(d.bitangent, q.bitangent); out.disk_lights.push_back(d);
  }
  for (const auto& m : scene.materials) {
    rtx_material d{};
    d.type = static_cast<int>(m.type);              // MaterialType: Lambertian 0, Metal 1, Dielectric 2
    d.has_texture = m.has_texture ? 1 : 0;
    put3(d.albedo, m.albedo); put3(d.emission, m.emission);
    d.fuzz = m.fuzz; d.refraction_index = m.refraction_index;
    d.tex_type = static_cast<int>(m.texture.type);  // TextureType: Solid 0, Checker 1, Image 2
    d.tex_image = -1;
    put3(d.tex_color1, m.texture.color1); put3(d.tex_color2, m.texture.color2); d.tex_scale = m.texture.scale;
    if (m.texture.image_data && m.texture.image_width > 0 && m.texture.image_height > 0 && m.texture.image_channels == 3) {
      rtx_image im{m.texture.image_data->data(), m.texture.image_width, m.texture.image_height};
      d.tex_image = static_cast<int>(out.images.size());
      out.images.push_back(im);
    }
    d.has_normal_map = 0; d.normal_image = -1; d.normal_strength = 1.0;
    if (m.has_normal_map && m.normal_map.valid()) {   // stbi_load(..., 3): always three channels (texture.hpp:61)
      rtx_image im{m.normal_map.data->data(), m.normal_map.width, m.normal_map.height};
      d.has_normal_map = 1;
      d.normal_image = static_cast<int>(out.images.size());
      d.normal_strength = m.normal_map.strength;
      out.images.push_back(im);
    }
    out.materials.push_back(d);
  }
  rtx_scene_desc& d = out.desc;
  d.spheres = out.spheres.data(); d.n_spheres = out.spheres.size();
  d.planes = out.planes.data(); d.n_planes = out.planes.size();
  d.triangles = out.triangles.data(); d.n_triangles = out.triangles.size();
  d.boxes = out.boxes.data(); d.n_boxes = out.boxes.size();
  d.point_lights = out.point_lights.data(); d.n_point_lights = out.point_lights.size();
  d.quad_lights = out.quad_lights.data(); d.n_quad_lights = out.quad_lights.size();
  d.disk_lights = out.disk_lights.data(); d.n_disk_lights = out.disk_lights.size();
  d.materials = out.materials.data(); d.n_materials = out.materials.size();
  d.images = out.images.data(); d.n_images = out.images.size();
  d.environment = nullptr;
  if (scene.environment && scene.environment->valid()) {
    const auto& e = *scene.environment;
    out.environment = rtx_environment{e.hdr_data->data(), e.width, e.height, e.intensity, e.rotation};
    d.environment = &out.environment;
  }
  return d;
}

template <class CameraT> inline rtx_camera to_camera(const CameraT& cam) {
  rtx_camera c{};
  put3(c.origin, cam.origin); put3(c.lower_left_corner, cam.lower_left_corner);
  put3(c.horizontal, cam.horizontal); put3(c.vertical, cam.vertical);
  return c;
}

// Renderer::Settings -> rtx_settings; the members only a GPU backend has get defaults
// (seed 1234 is the reference GPU backend's fixed seed, ref: engine/cuda/cuda_pathtracer.cu:354).
template <class SettingsT> inline rtx_settings to_settings(const SettingsT& s) {
  rtx_settings r{};
  r.width = s.width; r.height = s.height; r.samples_per_pixel = s.samples_per_pixel; r.max_depth = s.max_depth;
  r.use_nee = s.use_nee; r.use_mis = s.use_mis; r.use_mnee = s.use_mnee;
  r.use_background_gradient = s.use_background_gradient;
  put3(r.background_color, s.background_color); put3(r.background_top, s.background_top); put3(r.background_bottom, s.background_bottom);
  r.seed = 1234u; r.precision = RTX_PRECISION_F32; r.pipeline = RTX_PIPELINE_AUTO; r.shard_index = 0; r.shard_count = 1;
  r.mode = (int32_t)s.mode; r.clamp_max = s.clamp_max;
  r.caustic_photon_count = s.caustic_photon_count; r.photon_gather_count = s.photon_gather_count; r.caustic_gather_radius = s.caustic_gather_radius;   // RenderMode values are kept (engine/renderer.hpp:26-34)
  return r;
}

}  // namespace hip
}  // namespace raytracer
