"""Host-side mirror of the reference's scene loading path.

`load(path)` accepts the same JSON schema as ``SceneLoader::load``
(engine/scene_loader.hpp:98-472) and produces the same tables: material ids in
alphabetical key order (nlohmann's default object is a std::map), ``position``-only object
keys, material-0 fallback, the centred-quad frame construction, the derived members of
``QuadLight`` / ``DiskLight`` / ``Plane`` / ``Camera``.  All arithmetic is done with Python
floats (IEEE doubles, no fused multiply-add) in the reference's operand order, so the
resulting doubles are the reference's bit for bit; the loader-dump fixtures under
tests/golden pin that.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
import struct
import sys
from dataclasses import dataclass, field

import numpy as np

from . import abi

PI = 3.14159265358979323846

# RenderMode (engine/renderer.hpp:26-34)
MODE_WHITTED, MODE_PATHTRACE, MODE_BDPT, MODE_SPECTRAL, MODE_PLT, MODE_PHOTONMAP, MODE_PATHPHOTON = range(7)
# ToneMapper (engine/renderer.hpp:39-45)
TM_NONE, TM_REINHARD, TM_REINHARD_EXT, TM_ACES, TM_UNCHARTED2 = range(5)


# ---- core/vec3.c, on 3-tuples ------------------------------------------------------------
def v_add(a, b): return (a[0] + b[0], a[1] + b[1], a[2] + b[2])
def v_sub(a, b): return (a[0] - b[0], a[1] - b[1], a[2] - b[2])
def v_scale(a, t): return (a[0] * t, a[1] * t, a[2] * t)
def v_dot(a, b): return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]
def v_cross(a, b): return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])
def v_len(a): return math.sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2])


def v_normalize(a):
    l = v_len(a)
    return v_scale(a, 1.0 / l) if l > 0.0 else a


def _v3(x):
    return (float(x[0]), float(x[1]), float(x[2]))


def make_camera(lookfrom, lookat, vup, vfov, aspect) -> abi.Camera:
    """Camera::Camera (engine/scene.hpp:69-91)."""
    theta = vfov * PI / 180.0
    h = math.tan(theta / 2.0)
    vh = 2.0 * h
    vw = aspect * vh
    w = v_normalize(v_sub(lookfrom, lookat))
    u = v_normalize(v_cross(vup, w))
    v = v_cross(w, u)
    horizontal = v_scale(u, vw)
    vertical = v_scale(v, vh)
    llc = v_sub(v_sub(v_sub(lookfrom, v_scale(horizontal, 0.5)), v_scale(vertical, 0.5)), w)
    cam = abi.Camera()
    cam.origin[:] = lookfrom
    cam.lower_left_corner[:] = llc
    cam.horizontal[:] = horizontal
    cam.vertical[:] = vertical
    return cam


def make_quad_light(corner, eu, ev, emission) -> abi.QuadLight:
    """QuadLight::QuadLight (engine/primitives.hpp:228-234)."""
    cr = v_cross(eu, ev)
    area = v_len(cr)
    q = abi.QuadLight()
    q.corner[:] = corner
    q.edge_u[:] = eu
    q.edge_v[:] = ev
    q.normal[:] = v_scale(cr, 1.0 / area)
    q.emission[:] = emission
    q.area = area
    return q


def make_quad_light_centered(center, u_dir, v_dir, width, height, emission) -> abi.QuadLight:
    """QuadLight::centered (engine/primitives.hpp:245-252)."""
    u = v_scale(v_normalize(u_dir), width)
    v = v_scale(v_normalize(v_dir), height)
    corner = v_sub(v_sub(center, v_scale(u, 0.5)), v_scale(v, 0.5))
    return make_quad_light(corner, u, v, emission)


def make_disk_light(center, normal, radius, emission) -> abi.DiskLight:
    """DiskLight::DiskLight (engine/primitives.hpp:333-344)."""
    n = v_normalize(normal)
    d = abi.DiskLight()
    d.center[:] = center
    d.normal[:] = n
    d.radius = radius
    d.emission[:] = emission
    d.area = PI * radius * radius
    up = (0.0, 1.0, 0.0) if abs(n[1]) < 0.999 else (1.0, 0.0, 0.0)
    t = v_normalize(v_cross(up, n))
    d.tangent[:] = t
    d.bitangent[:] = v_cross(n, t)
    return d


def read_radiance_hdr(path: str) -> np.ndarray:
    """Decode a Radiance .hdr (32-bit_rle_rgbe, -Y +X) into float32 RGB [H, W, 3].

    Texel value = mantissa * 2^(e-136) in float, 0 when e == 0: the conversion the
    reference's vendored decoder applies (engine/stb_image.h:7131-7156)."""
    with open(path, "rb") as f:
        data = f.read()
    pos = 0

    def line():
        nonlocal pos
        end = data.index(b"\n", pos)
        s = data[pos:end]
        pos = end + 1
        return s

    first = line()
    if first not in (b"#?RADIANCE", b"#?RGBE"):
        raise ValueError("not a Radiance HDR file")
    ok = False
    while True:
        s = line()
        if s == b"":
            break
        if s == b"FORMAT=32-bit_rle_rgbe":
            ok = True
    if not ok:
        raise ValueError("unsupported HDR format")
    tok = line().split()
    if len(tok) != 4 or tok[0] != b"-Y" or tok[2] != b"+X":
        raise ValueError("unsupported HDR data layout")
    height, width = int(tok[1]), int(tok[3])
    rgbe = np.zeros((height, width, 4), dtype=np.uint8)
    buf = np.frombuffer(data, dtype=np.uint8)
    if width < 8 or width >= 32768:
        rgbe[:] = buf[pos:pos + height * width * 4].reshape(height, width, 4)
    else:
        for j in range(height):
            c1, c2, l1, l2 = data[pos], data[pos + 1], data[pos + 2], data[pos + 3]
            if c1 != 2 or c2 != 2 or (l1 & 0x80):
                raise ValueError("flat (non-RLE) scanlines inside an RLE-width file are not supported")
            if ((l1 << 8) | l2) != width:
                raise ValueError("corrupt HDR scanline length")
            pos += 4
            for k in range(4):
                i = 0
                while i < width:
                    count = data[pos]
                    pos += 1
                    if count > 128:
                        count -= 128
                        rgbe[j, i:i + count, k] = data[pos]
                        pos += 1
                    else:
                        rgbe[j, i:i + count, k] = buf[pos:pos + count]
                        pos += count
                    i += count
    e = rgbe[..., 3].astype(np.int32)
    scale = np.ldexp(np.float32(1.0), e - 136).astype(np.float32)
    out = rgbe[..., :3].astype(np.float32) * scale[..., None]
    out[e == 0] = 0.0
    return np.ascontiguousarray(out, dtype=np.float32)


@dataclass
class SceneData:
    """SceneLoader::SceneData (engine/scene_loader.hpp:27-64) in array form."""
    spheres: list = field(default_factory=list)
    planes: list = field(default_factory=list)
    triangles: list = field(default_factory=list)
    boxes: list = field(default_factory=list)
    point_lights: list = field(default_factory=list)
    quad_lights: list = field(default_factory=list)
    disk_lights: list = field(default_factory=list)
    materials: list = field(default_factory=list)
    material_names: list = field(default_factory=list)
    images: list = field(default_factory=list)          # uint8 [H, W, 3] arrays
    env_texels: np.ndarray | None = None               # float32 [H, W, 3]
    env_intensity: float = 1.0
    env_rotation: float = 0.0                          # radians
    triangle_array: np.ndarray | None = None           # optional bulk triangles, structured dtype of rtx_triangle

    camera_position: tuple = (0.0, 1.0, 5.0)
    camera_target: tuple = (0.0, 0.0, 0.0)
    camera_up: tuple = (0.0, 1.0, 0.0)
    camera_fov: float = 60.0

    width: int = 1920
    height: int = 1080
    samples: int = 16
    max_depth: int = 50
    mode: int = MODE_WHITTED
    use_nee: bool = True
    use_mis: bool = True
    tone_mapper: int = TM_ACES
    exposure: float = 1.0
    clamp_max: float = 10.0
    caustic_photon_count: int = 50000      # engine/scene_loader.hpp:56-61
    photon_gather_count: int = 100
    caustic_gather_radius: float = 0.1
    background_color: tuple = (0.0, 0.0, 0.0)
    background_top: tuple = (0.5, 0.7, 1.0)
    background_bottom: tuple = (1.0, 1.0, 1.0)
    use_background_gradient: bool = False
    output_file: str = "output.png"
    warnings: list = field(default_factory=list)

    _keep: list = field(default_factory=list, repr=False)

    # -- the structures handed across the C ABI -------------------------------------------
    def camera(self, width: int | None = None, height: int | None = None) -> abi.Camera:
        w = self.width if width is None else width
        h = self.height if height is None else height
        return make_camera(self.camera_position, self.camera_target, self.camera_up, self.camera_fov, float(w) / h)

    def settings(self, **over) -> abi.Settings:
        st = abi.Settings()
        st.width, st.height = self.width, self.height
        st.samples_per_pixel, st.max_depth = self.samples, self.max_depth
        st.use_nee, st.use_mis, st.use_mnee = int(self.use_nee), int(self.use_mis), 0
        st.use_background_gradient = int(self.use_background_gradient)
        st.background_color[:] = self.background_color
        st.background_top[:] = self.background_top
        st.background_bottom[:] = self.background_bottom
        st.seed = 1234  # the reference GPU backend's fixed seed (engine/cuda/cuda_pathtracer.cu:354)
        st.precision = abi.PRECISION_F32
        st.pipeline = abi.PIPELINE_AUTO
        st.shard_index, st.shard_count = 0, 1
        st.clamp_max = self.clamp_max
        st.caustic_photon_count, st.photon_gather_count, st.caustic_gather_radius = self.caustic_photon_count, self.photon_gather_count, self.caustic_gather_radius
        st.mode = abi.MODE_PATHTRACE   # the hot path; pass mode=self.mode (or abi.MODE_WHITTED) for the scene file's own mode
        for k, v in over.items():
            if not hasattr(st, k):
                raise AttributeError(k)
            setattr(st, k, v)
        return st

    def desc(self) -> abi.SceneDesc:
        d = abi.SceneDesc()
        keep = self._keep
        keep.clear()

        def arr(items, ctype):
            a = (ctype * max(1, len(items)))(*items)
            keep.append(a)
            return C.cast(a, C.POINTER(ctype)), len(items)

        d.spheres, d.n_spheres = arr(self.spheres, abi.Sphere)
        d.planes, d.n_planes = arr(self.planes, abi.Plane)
        if self.triangle_array is not None:
            ta = np.ascontiguousarray(self.triangle_array)
            assert ta.dtype.itemsize == C.sizeof(abi.Triangle)
            keep.append(ta)
            d.triangles = C.cast(ta.ctypes.data, C.POINTER(abi.Triangle))
            d.n_triangles = ta.shape[0]
        else:
            d.triangles, d.n_triangles = arr(self.triangles, abi.Triangle)
        d.boxes, d.n_boxes = arr(self.boxes, abi.Box)
        d.point_lights, d.n_point_lights = arr(self.point_lights, abi.PointLight)
        d.quad_lights, d.n_quad_lights = arr(self.quad_lights, abi.QuadLight)
        d.disk_lights, d.n_disk_lights = arr(self.disk_lights, abi.DiskLight)
        d.materials, d.n_materials = arr(self.materials, abi.Material)
        imgs = []
        for im in self.images:
            im = np.ascontiguousarray(im, dtype=np.uint8)
            keep.append(im)
            ci = abi.Image()
            ci.rgb = im.ctypes.data_as(C.POINTER(C.c_uint8))
            ci.height, ci.width = im.shape[0], im.shape[1]
            imgs.append(ci)
        d.images, d.n_images = arr(imgs, abi.Image)
        if self.env_texels is not None:
            tex = np.ascontiguousarray(self.env_texels, dtype=np.float32)
            keep.append(tex)
            env = abi.Environment()
            env.rgb = tex.ctypes.data_as(C.POINTER(C.c_float))
            env.height, env.width = tex.shape[0], tex.shape[1]
            env.intensity, env.rotation = self.env_intensity, self.env_rotation
            keep.append(env)
            d.environment = C.pointer(env)
        return d

    def bounded_primitive_count(self) -> int:
        nt = len(self.triangles) if self.triangle_array is None else int(self.triangle_array.shape[0])
        return len(self.spheres) + nt + len(self.boxes)


TRIANGLE_DTYPE = np.dtype([("v0", "<f8", 3), ("v1", "<f8", 3), ("v2", "<f8", 3), ("material_id", "<i4"), ("pad_", "<i4")])


def _default_material() -> abi.Material:
    """A default-constructed Material (engine/material.hpp:48-66) with a default Texture."""
    m = abi.Material()
    m.type = abi.MAT_LAMBERTIAN
    m.albedo[:] = (0.5, 0.5, 0.5)
    m.emission[:] = (0.0, 0.0, 0.0)
    m.has_texture = 0
    m.fuzz = 0.0
    m.refraction_index = 1.5
    m.tex_type = abi.TEX_SOLID
    m.tex_image = -1
    m.tex_color1[:] = (1.0, 1.0, 1.0)
    m.tex_color2[:] = (0.0, 0.0, 0.0)
    m.tex_scale = 1.0
    m.has_normal_map = 0
    m.normal_image = -1
    m.normal_strength = 1.0
    return m


def _parse_mode(s: str) -> int:
    """engine/scene_loader.hpp:345-359 / :429-444."""
    if s in ("pathtrace", "pathtracing", "path"):
        return MODE_PATHTRACE
    if s in ("bdpt", "bidirectional"):
        return MODE_BDPT
    if s in ("spectral", "wavelength"):
        return MODE_SPECTRAL
    if s in ("plt", "polarized"):
        return MODE_PLT
    if s in ("photon", "photonmap"):
        return MODE_PHOTONMAP
    if s in ("pathphoton", "hybrid", "path_photon"):
        return MODE_PATHPHOTON
    return MODE_WHITTED


def _load_image_rgb8(path: str):
    from PIL import Image  # decoding only; a different JPEG decoder may differ from the reference's by +-1

    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


def load(path: str, base_dir: str | None = None, image_loader=None) -> SceneData:
    """SceneLoader::load.  Relative texture / environment paths resolve against `base_dir`
    (default: the current directory, as in the reference, then the scene file's parent's
    parent, where the reference keeps ``textures/``)."""
    data = SceneData()
    try:
        with open(path, "r") as f:
            j = json.load(f)
    except OSError:
        print(f"Error: Could not open scene file: {path}", file=sys.stderr)
        return data
    except json.JSONDecodeError as e:
        print(f"Error parsing JSON: {e}", file=sys.stderr)
        return data

    search = [base_dir] if base_dir else [os.getcwd(), os.path.dirname(os.path.dirname(os.path.abspath(path)))]

    def resolve(fn):
        if os.path.isabs(fn):
            return fn
        for b in search:
            p = os.path.join(b, fn)
            if os.path.exists(p):
                return p
        return fn

    def parse_texture(tex, m: abi.Material):
        """SceneLoader::parse_texture (scene_loader.hpp:69-91) into the material record."""
        ttype = tex.get("type", "solid")
        m.tex_scale = 1.0
        m.tex_color1[:] = (1.0, 1.0, 1.0)
        m.tex_color2[:] = (0.0, 0.0, 0.0)
        if ttype == "checker":
            m.tex_type = abi.TEX_CHECKER
            m.tex_color1[:] = _v3(tex.get("color1", [1.0, 1.0, 1.0]))
            m.tex_color2[:] = _v3(tex.get("color2", [0.0, 0.0, 0.0]))
            m.tex_scale = float(tex.get("scale", 10.0))
        elif ttype == "image":
            fn = tex.get("file", "")
            img = None
            if fn:
                try:
                    img = (image_loader or _load_image_rgb8)(resolve(fn))
                except Exception as e:  # Texture::load_image: magenta on failure
                    print(f"Error: Could not load texture: {fn} ({e})", file=sys.stderr)
            if img is None:
                m.tex_type = abi.TEX_SOLID
                m.tex_color1[:] = (1.0, 0.0, 1.0)
            else:
                m.tex_type = abi.TEX_IMAGE
                m.tex_image = len(data.images)
                data.images.append(img)
        else:
            m.tex_type = abi.TEX_SOLID
            m.tex_color1[:] = _v3(tex.get("color", [0.5, 0.5, 0.5]))

    material_ids: dict[str, int] = {}
    for name in sorted(j.get("materials", {}).keys()):  # std::map iteration order
        mat = j["materials"][name]
        m = _default_material()
        mtype = mat.get("type", "lambertian")
        has_texture = "texture" in mat
        if mtype == "lambertian":
            if has_texture:
                parse_texture(mat["texture"], m)
                m.has_texture = 1
            else:
                m.albedo[:] = _v3(mat.get("color", [0.5, 0.5, 0.5]))
        elif mtype == "metal":
            fuzz = float(mat.get("fuzz", 0.0))
            m.type = abi.MAT_METAL
            m.fuzz = fuzz if fuzz < 1.0 else 1.0
            if has_texture:
                parse_texture(mat["texture"], m)
                m.has_texture = 1
            else:
                m.albedo[:] = _v3(mat.get("color", [0.8, 0.8, 0.8]))
        elif mtype in ("dielectric", "glass"):
            m.type = abi.MAT_DIELECTRIC
            m.albedo[:] = (1.0, 1.0, 1.0)
            m.refraction_index = float(mat.get("ior", 1.5))
        elif mtype == "emissive":
            c = _v3(mat.get("color", [1.0, 1.0, 1.0]))
            e = _v3(mat.get("emission", [1.0, 1.0, 1.0]))
            if "emit" in mat and isinstance(mat["emit"], list) and len(mat["emit"]) >= 3:
                e = _v3(mat["emit"])
            inten = float(mat.get("intensity", 1.0))
            m.albedo[:] = c
            m.emission[:] = (e[0] * inten, e[1] * inten, e[2] * inten)
        if "normal_map" in mat and mat["normal_map"].get("file", ""):  # scene_loader.hpp:165-172, NormalMap::load (texture.hpp:58-77)
            fn = mat["normal_map"]["file"]
            try:
                img = (image_loader or _load_image_rgb8)(resolve(fn))
                m.has_normal_map = 1
                m.normal_image = len(data.images)
                m.normal_strength = float(mat["normal_map"].get("strength", 1.0))
                data.images.append(img)
            except Exception as e:  # a map that fails to load leaves has_normal_map false
                print(f"Error: Could not load normal map: {fn} ({e})", file=sys.stderr)
        if "roughness_map" in mat and mat["roughness_map"].get("file", ""):
            data.warnings.append(f"material '{name}': roughness_map is not read by the path tracer and is ignored")
        material_ids[name] = len(data.materials)
        data.materials.append(m)
        data.material_names.append(name)

    def get_material(obj) -> int:
        if "material" in obj:
            nm = obj["material"]
            if nm in material_ids:
                return material_ids[nm]
            print(f"Warning: Unknown material '{nm}'", file=sys.stderr)
        return 0

    for obj in j.get("objects", []):
        otype = obj.get("type", "")
        mid = get_material(obj)
        if otype == "sphere":
            s = abi.Sphere()
            s.center[:] = _v3(obj.get("position", [0, 0, 0]))
            s.radius = float(obj.get("radius", 1.0))
            s.material_id = mid
            data.spheres.append(s)
        elif otype == "plane":
            p = abi.Plane()
            p.point[:] = _v3(obj.get("position", [0, 0, 0]))
            p.normal[:] = v_normalize(_v3(obj.get("normal", [0, 1, 0])))  # Plane ctor, primitives.hpp:29-30
            p.material_id = mid
            data.planes.append(p)
        elif otype == "box":
            b = abi.Box()
            if "min" in obj and "max" in obj:
                b.box_min[:] = _v3(obj["min"])
                b.box_max[:] = _v3(obj["max"])
            else:  # Box::centered, primitives.hpp:145-148
                c = _v3(obj.get("position", [0, 0, 0]))
                half = (float(obj.get("width", 1.0)) / 2.0, float(obj.get("height", 1.0)) / 2.0,
                        float(obj.get("depth", 1.0)) / 2.0)
                b.box_min[:] = v_sub(c, half)
                b.box_max[:] = v_add(c, half)
            b.material_id = mid
            data.boxes.append(b)
        elif otype == "triangle":
            t = abi.Triangle()
            t.v0[:] = _v3(obj["v0"])
            t.v1[:] = _v3(obj["v1"])
            t.v2[:] = _v3(obj["v2"])
            t.material_id = mid
            data.triangles.append(t)

    for light in j.get("lights", []):
        ltype = light.get("type", "point")
        col = _v3(light.get("color", [1, 1, 1]))
        inten = float(light.get("intensity", 1.0))
        emission = (col[0] * inten, col[1] * inten, col[2] * inten)
        if ltype == "point":
            pl = abi.PointLight()
            pl.position[:] = _v3(light.get("position", [0, 5, 0]))
            pl.intensity[:] = emission
            data.point_lights.append(pl)
        elif ltype in ("quad", "rect", "rectangle"):
            if "corner" in light and "edge_u" in light and "edge_v" in light:
                data.quad_lights.append(make_quad_light(_v3(light["corner"]), _v3(light["edge_u"]), _v3(light["edge_v"]), emission))
            else:  # scene_loader.hpp:277-293
                pos = _v3(light.get("position", [0, 3, 0]))
                width = float(light.get("width", 1.0))
                height = float(light.get("height", 1.0))
                n = v_normalize(_v3(light.get("normal", [0, -1, 0])))
                up = (0.0, 1.0, 0.0) if abs(n[1]) < 0.999 else (1.0, 0.0, 0.0)
                u_dir = v_normalize(v_cross(up, n))
                v_dir = v_cross(n, u_dir)
                data.quad_lights.append(make_quad_light_centered(pos, u_dir, v_dir, width, height, emission))
        elif ltype in ("disk", "disc"):
            data.disk_lights.append(make_disk_light(_v3(light.get("position", [0, 3, 0])), _v3(light.get("normal", [0, -1, 0])),
                                                    float(light.get("radius", 0.5)), emission))

    if "camera" in j:
        cam = j["camera"]
        if "position" in cam:
            data.camera_position = _v3(cam["position"])
        if "target" in cam:
            data.camera_target = _v3(cam["target"])
        if "up" in cam:
            data.camera_up = _v3(cam["up"])
        data.camera_fov = float(cam.get("fov", 60.0))

    if "render" in j:
        r = j["render"]
        data.width = int(r.get("width", 2560))
        data.height = int(r.get("height", 1440))
        data.samples = int(r.get("samples", 16))
        data.max_depth = int(r.get("max_depth", 50))
        data.output_file = r.get("output", "output.png")
        data.use_nee = bool(r.get("nee", True))
        data.use_mis = bool(r.get("mis", True))
        data.exposure = float(r.get("exposure", 1.0))
        data.clamp_max = float(r.get("clamp_max", 10.0))
        data.mode = _parse_mode(r.get("mode", "whitted"))
        tm = r.get("tonemapper", "aces")
        data.tone_mapper = {"none": TM_NONE, "reinhard": TM_REINHARD, "uncharted2": TM_UNCHARTED2}.get(tm, TM_ACES)

    if "photon_mapping" in j:   # engine/scene_loader.hpp:369-378 (the members PathPhoton reads)
        pm = j["photon_mapping"]
        data.caustic_photon_count = int(pm.get("caustic_photon_count", 50000))
        data.photon_gather_count = int(pm.get("gather_count", 100))
        data.caustic_gather_radius = float(np.float32(pm.get("caustic_radius", 0.1)))

    if "background" in j:
        bg = j["background"]
        if isinstance(bg, list) and len(bg) >= 3:
            data.background_color = _v3(bg)
            data.use_background_gradient = False
        elif isinstance(bg, dict):
            if "color" in bg:
                data.background_color = _v3(bg["color"])
                data.use_background_gradient = False
            if "gradient" in bg:
                g = bg["gradient"]
                if "top" in g:
                    data.background_top = _v3(g["top"])
                if "bottom" in g:
                    data.background_bottom = _v3(g["bottom"])
                data.use_background_gradient = True

    if "environment" in j:
        env = j["environment"]
        fn = env.get("file", "")
        if fn:
            try:
                data.env_texels = read_radiance_hdr(resolve(fn))
                data.env_intensity = float(env.get("intensity", 1.0))
                data.env_rotation = float(env.get("rotation", 0.0)) * PI / 180.0
            except (OSError, ValueError) as e:
                print(f"Error: Could not load environment map: {fn} ({e})", file=sys.stderr)

    if "render_mode" in j:
        data.mode = _parse_mode(j["render_mode"])
    if "image" in j:
        img = j["image"]
        data.width = int(img.get("width", data.width))
        data.height = int(img.get("height", data.height))
        data.samples = int(img.get("samples_per_pixel", img.get("samples", data.samples)))
        data.max_depth = int(img.get("max_depth", data.max_depth))
    return data
