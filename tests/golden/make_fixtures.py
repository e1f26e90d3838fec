"""Regenerate tests/golden/ from the REAL reference renderer.

Runs only where /root/reference exists (the build container): it drives
oracle/_ref/ref_harness, i.e. the reference's own SceneLoader, Scene::hit, BVH::build and
Renderer::render compiled in place by oracle/Makefile.  The outputs are data only:

  loader/<scene>.json   tables SceneLoader::load produced (materials in id order, primitives,
                        lights with derived members, Camera vectors, render settings)
  hits/<scene>.npz      ray sets + Scene::hit results (t, primitive, material, normal, point, u, v)
  bvh/<name>.npz        node / primitive arrays of the BVH exactly as Scene::build_bvh makes it
  films/<name>.npz      seeded single-thread Renderer::render films (std::mt19937 seed given)
  converged/<scene>.npz higher-sample films + the measured Monte-Carlo noise floor
  kat/<scene>.npy       scalar known-answer stream (light sampling, pdfs, env lookup, textures, ...)

Usage:  python tests/golden/make_fixtures.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import rtx_pkg  # noqa: E402

pkg = rtx_pkg.load()
REF = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
REFROOT = "/root/reference"
SCENES = os.path.join(HERE, "scenes")

HIT_DTYPE = pkg.renderer.HIT_DTYPE


def run(args, **kw):
    # cwd = tests/golden so that "textures/..." inside scene files resolves against the copies
    # of the reference's assets kept there (the reference resolves them against its cwd)
    return subprocess.run([REF] + [str(a) for a in args], cwd=HERE, check=True, capture_output=True, text=True, **kw)


def scene_path(name):
    return os.path.join(SCENES, name + ".json")


def dump_loader(name):
    out = run(["dump", scene_path(name)]).stdout
    line = [l for l in out.splitlines() if l.startswith("@@JSON@@")][-1][len("@@JSON@@"):]
    doc = json.loads(line)
    os.makedirs(os.path.join(HERE, "loader"), exist_ok=True)
    with open(os.path.join(HERE, "loader", name + ".json"), "w") as f:
        json.dump(doc, f, indent=0, separators=(",", ":"))
        f.write("\n")
    return doc


def camera_rays(doc, W, H):
    cam = doc["camera"]
    # Camera vectors were computed for the scene's own aspect; rebuild for W x H the way the
    # test does (through the package's Camera mirror, itself pinned by loader/<scene>.json).
    c = pkg.scene_loader.make_camera(tuple(cam["position"]), tuple(cam["target"]), tuple(cam["up"]), cam["fov"], float(W) / H)
    o = np.array(c.origin[:]); llc = np.array(c.lower_left_corner[:]); hz = np.array(c.horizontal[:]); vt = np.array(c.vertical[:])
    xs, ys = np.meshgrid(np.arange(W), np.arange(H))
    s = ((xs + 0.5) / W).ravel(); t = ((ys + 0.5) / H).ravel()
    d = ((llc[None] + s[:, None] * hz[None]) + t[:, None] * vt[None]) - o[None]
    n = W * H
    return np.concatenate([np.tile(o, (n, 1)), d, np.full((n, 1), 0.001), np.full((n, 1), 1e9)], axis=1)


def random_rays(n, seed, centre, half):
    rng = np.random.default_rng(seed)
    ro = rng.uniform(-1, 1, (n, 3)) * np.asarray(half) + np.asarray(centre)
    rd = rng.normal(size=(n, 3))
    k = n // 10
    rd[0 * k:1 * k, 0] = 0.0           # zero components: the 1/0 slab paths
    rd[1 * k:2 * k, 1] = 0.0
    rd[2 * k:3 * k, 2] = -0.0
    rd[3 * k:4 * k, 1:] = 0.0          # axis-parallel rays
    rd[4 * k:5 * k] *= 1e-3            # short, unnormalised directions
    tmax = rng.choice([1e9, 4.0, 0.75], (n, 1))
    tmin = rng.choice([0.001, 0.0, 0.25], (n, 1))
    return np.concatenate([ro, rd, tmin, tmax], axis=1)


def ref_hits(name, rays, tris=None):
    with tempfile.TemporaryDirectory() as td:
        rp, op = os.path.join(td, "rays.bin"), os.path.join(td, "hits.bin")
        np.ascontiguousarray(rays, dtype=np.float64).tofile(rp)
        args = ["hits", scene_path(name), rp, op]
        if tris is not None:
            tp = os.path.join(td, "tris.bin")
            tris_to_bin(tris).tofile(tp)
            args.append(tp)
        run(args)
        return np.fromfile(op, dtype=HIT_DTYPE)


def tris_to_bin(tri):
    out = np.zeros((tri.shape[0], 10), dtype=np.float64)
    out[:, 0:3] = tri["v0"]; out[:, 3:6] = tri["v1"]; out[:, 6:9] = tri["v2"]; out[:, 9] = tri["material_id"]
    return out


def ref_bvh(name, tris=None):
    with tempfile.TemporaryDirectory() as td:
        op = os.path.join(td, "bvh.bin")
        args = ["bvh", scene_path(name), op]
        if tris is not None:
            tp = os.path.join(td, "tris.bin")
            tris_to_bin(tris).tofile(tp)
            args.append(tp)
        run(args)
        raw = open(op, "rb").read()
    nn, npr = np.frombuffer(raw[:16], dtype=np.int64)
    node_dt = np.dtype([("box", "<f8", 6), ("ints", "<i4", 4)])
    nodes = np.frombuffer(raw[16:16 + nn * node_dt.itemsize], dtype=node_dt)
    prims = np.frombuffer(raw[16 + nn * node_dt.itemsize:], dtype=np.int32).reshape(npr, 2)
    return nodes["box"].copy(), nodes["ints"].copy(), prims.copy()


def ref_film(name, W, H, spp, depth, seed, nee=1, mis=1, mnee=0, threads=1, mode=1, extra=(), info=None):
    with tempfile.TemporaryDirectory() as td:
        op = os.path.join(td, "film.bin")
        out = run(["render", scene_path(name), W, H, spp, depth, seed, nee, mis, mnee, threads, op, mode] + list(extra)).stdout
        if info is not None:
            info["aim_seed"] = int([l for l in out.splitlines() if l.startswith("@@AIMSEED@@")][-1].split()[1])
        raw = np.fromfile(op, dtype=np.uint8)
    secs = float([l for l in out.splitlines() if l.startswith("@@TIME@@")][-1].split()[1])
    return np.frombuffer(raw[8:].tobytes(), dtype=np.float64).reshape(H, W, 3).copy(), secs


def post_test_film(W=48, H=20, seed=5):
    """A float film that exercises every branch of the post-process: zeros, tiny, mid, > 1, huge, negative."""
    rng = np.random.default_rng(seed)
    film = np.exp(rng.uniform(-12.0, 4.0, size=(H, W, 3))).astype(np.float32)
    film[0, :8] = [0.0, 1.0, 0.5]
    film[1, :8] = [-0.25, 1e-8, 1e6]
    film[2, :8] = np.linspace(0.0, 2.0, 24, dtype=np.float32).reshape(8, 3)
    return film


def ref_post(film, mapper, exposure):
    """uint8 [H, W, 3], row 0 = top, through Image::apply_tone_mapping + Image::write_ppm of the reference."""
    H, W, _ = film.shape
    with tempfile.TemporaryDirectory() as td:
        fp, op = os.path.join(td, "film.f32"), os.path.join(td, "out.ppm")
        film.astype(np.float32).tofile(fp)
        run(["post", fp, W, H, mapper, repr(float(exposure)), op])
        tok = open(op).read().split()
    assert tok[0] == "P3" and int(tok[1]) == W and int(tok[2]) == H and tok[3] == "255"
    return np.array(tok[4:], dtype=np.int64).reshape(H, W, 3).astype(np.uint8)


def make_post_fixtures():
    os.makedirs(os.path.join(HERE, "post"), exist_ok=True)
    film = post_test_film()
    out = {"film": film}
    for mapper in range(5):
        for k, exposure in enumerate((1.0, 0.35, 2.5)):
            out[f"m{mapper}_e{k}"] = ref_post(film, mapper, exposure)
    out["exposures"] = np.array([1.0, 0.35, 2.5])
    np.savez_compressed(os.path.join(HERE, "post", "post.npz"), **out)
    print("post fixtures", len(out) - 2)


MNEE_FILMS = [("cornell_box", 24, 14, 3, 12, 91, 1, 1), ("cornell_box", 24, 14, 3, 12, 91, 1, 0), ("cornell_box_quad", 20, 20, 3, 12, 92, 1, 1),
              ("caustics_test", 20, 20, 3, 12, 93, 1, 1), ("caustics_test", 20, 20, 3, 12, 93, 1, 0),
              # kitchen_sink with MIS on: the quad-light look-ahead reads materials[-1] (see the films above) and here the
              # garbage is large; disk_room: the straight probe of MNEE indexes materials[-1000] and the reference crashes
              ("kitchen_sink", 24, 14, 3, 10, 94, 1, 0)]


# Scenes without quad / disk lights: a Whitted ray that lands on one indexes materials[negative] in the reference
# (undefined behaviour, engine/renderer.cpp:493).  earth_demo is left out because its JPEG texture decodes differently here.
WHITTED_FILMS = [("demo", 32, 18, 4, 50, 101), ("kitchen_points", 32, 18, 4, 12, 103), ("textured", 24, 14, 3, 10, 104), ("cornell_box", 24, 14, 3, 8, 105)]


def make_whitted_fixtures():
    """Seeded single-thread films in RenderMode::Whitted (Renderer::ray_color_whitted, engine/renderer.cpp:483-569)."""
    for (n, W, H, spp, depth, seed) in WHITTED_FILMS:
        film, _ = ref_film(n, W, H, spp, depth, seed, mode=0)
        tag = f"{n}_{W}x{H}_s{spp}_d{depth}_seed{seed}_whitted"
        np.savez_compressed(os.path.join(HERE, "films", tag + ".npz"), film=film, scene=n, W=W, H=H, spp=spp, depth=depth, seed=seed, nee=1, mis=1, mode=0)
        print("film", tag, float(film.mean()))


def make_mnee_fixtures():
    """Seeded single-thread films with use_mnee = true (the CLI default of the reference)."""
    for (n, W, H, spp, depth, seed, nee, mis) in MNEE_FILMS:
        film, _ = ref_film(n, W, H, spp, depth, seed, nee, mis, mnee=1)
        tag = f"{n}_{W}x{H}_s{spp}_d{depth}_seed{seed}_nee{nee}_mis{mis}_mnee1"
        np.savez_compressed(os.path.join(HERE, "films", tag + ".npz"), film=film, scene=n, W=W, H=H, spp=spp, depth=depth, seed=seed, nee=nee, mis=mis, mnee=1)
        print("film", tag, float(film.mean()))


NORMAL_MAP_FILMS = [("normal_mapped", 32, 18, 4, 12, 111, 1, 1, 0, 1), ("normal_mapped", 24, 14, 3, 12, 112, 1, 0, 0, 1), ("normal_mapped", 24, 14, 3, 12, 113, 0, 0, 0, 1),
                    ("normal_mapped", 20, 12, 3, 10, 114, 1, 1, 1, 1), ("normal_mapped", 32, 18, 4, 12, 115, 1, 1, 0, 0)]


def make_normal_map_fixtures():
    """The normal-mapped scene (apply_normal_map, engine/renderer.cpp:72-94): loader tables, decoded map texels,
    closest hits and seeded films in both render modes, with MNEE off and on."""
    n = "normal_mapped"
    doc = dump_loader(n)
    with tempfile.TemporaryDirectory() as td:
        run(["texels", scene_path(n), os.path.join(td, "t_")])
        maps = {}
        for fn in sorted(os.listdir(td)):
            raw = np.fromfile(os.path.join(td, fn), dtype=np.uint8)
            w, h = np.frombuffer(raw[:8].tobytes(), dtype=np.int32)
            maps[fn[2:-4]] = raw[8:].reshape(h, w, 3).copy()
    np.savez_compressed(os.path.join(HERE, "loader", n + "_texels.npz"), **maps)
    rays = np.concatenate([camera_rays(doc, 48, 27), random_rays(3000, 19, (0, 1.5, -2), (4, 2, 4))], axis=0)
    hits = ref_hits(n, rays)
    np.savez_compressed(os.path.join(HERE, "hits", n + ".npz"), rays=rays, hits=hits)
    print("hits", n, len(rays), int(hits["hit"].sum()), "maps", sorted(maps))
    for (n, W, H, spp, depth, seed, nee, mis, mnee, mode) in NORMAL_MAP_FILMS:
        film, _ = ref_film(n, W, H, spp, depth, seed, nee, mis, mnee=mnee, mode=mode)
        tag = f"{n}_{W}x{H}_s{spp}_d{depth}_seed{seed}_nee{nee}_mis{mis}" + ("_mnee1" if mnee else "") + ("_whitted" if mode == 0 else "")
        extra = {"mode": 0} if mode == 0 else {}
        np.savez_compressed(os.path.join(HERE, "films", tag + ".npz"), film=film, scene=n, W=W, H=H, spp=spp, depth=depth, seed=seed, nee=nee, mis=mis, mnee=mnee, **extra)
        print("film", tag, float(film.mean()))


# RenderMode::BDPT (BDPTIntegrator::Li, engine/bdpt.cpp): every light type, rough / mirror metals, glass, textures
BDPT_FILMS = [("caustics_test", 24, 24, 4, 16, 121, 1), ("cornell_box", 24, 14, 4, 8, 122, 1), ("cornell_box", 24, 14, 3, 8, 123, 0), ("cornell_box_quad", 20, 20, 4, 12, 124, 1),
              ("kitchen_sink", 32, 18, 4, 10, 125, 1), ("disk_room", 24, 14, 4, 8, 126, 1), ("textured", 24, 14, 3, 6, 127, 1), ("kitchen_points", 24, 14, 3, 6, 128, 1)]


def make_bdpt_fixtures():
    for (n, W, H, spp, depth, seed, mis) in BDPT_FILMS:
        film, _ = ref_film(n, W, H, spp, depth, seed, 1, mis, mode=2)
        tag = f"{n}_{W}x{H}_s{spp}_d{depth}_seed{seed}_mis{mis}_bdpt"
        np.savez_compressed(os.path.join(HERE, "films", tag + ".npz"), film=film, scene=n, W=W, H=H, spp=spp, depth=depth, seed=seed, nee=1, mis=mis, mode=2)
        print("film", tag, float(film.mean()))


# RenderMode::PathPhoton (engine/renderer.cpp:344-372, :844-976).  Scenes whose lights are emissive spheres (only those emit
# photons) and that have no quad / disk light (a photon landing on one indexes materials[negative] in the reference).
# The reference aims its photons with a generator seeded by the hash of the thread id: the seed of each run is kept.
PATHPHOTON_FILMS = [("cornell_box", 24, 14, 3, 10, 131, 1, 4000, 40, 0.3), ("cornell_box", 20, 12, 2, 8, 132, 0, 1500, 10, 0.1),
                    ("kitchen_points", 24, 14, 3, 8, 133, 1, 3000, 25, 0.25), ("textured", 24, 14, 3, 8, 134, 1, 3000, 100, 0.5),
                    ("normal_mapped", 24, 14, 3, 8, 135, 1, 3000, 30, 0.3)]


def make_pathphoton_fixtures():
    for (n, W, H, spp, depth, seed, nee, photons, k, radius) in PATHPHOTON_FILMS:
        info = {}
        film, _ = ref_film(n, W, H, spp, depth, seed, nee, 1, mode=6, extra=[photons, k, repr(radius)], info=info)
        tag = f"{n}_{W}x{H}_s{spp}_d{depth}_seed{seed}_nee{nee}_pathphoton"
        np.savez_compressed(os.path.join(HERE, "films", tag + ".npz"), film=film, scene=n, W=W, H=H, spp=spp, depth=depth, seed=seed, nee=nee, mis=1, mode=6,
                            caustic_photon_count=photons, photon_gather_count=k, caustic_gather_radius=np.float32(radius), aim_seed=np.uint64(info["aim_seed"]))
        print("film", tag, float(film.mean()), info)


def make_pathphoton_converged():
    """PathPhoton films of the real reference at 4N spp plus two N-spp films of other seeds.  Every run traces its own photon
    map (other aim seed), so the noise floor holds the map's variance too."""
    for n, depth, photons, k, radius in (("cornell_box", 12, 3000, 30, 0.3),):
        W, H, N = 64, 36, 128
        extra = [photons, k, repr(radius)]
        hi, secs = ref_film(n, W, H, 4 * N, depth, 1001, threads=8, mode=6, extra=extra)
        lo_a, _ = ref_film(n, W, H, N, depth, 2002, threads=8, mode=6, extra=extra)
        lo_b, _ = ref_film(n, W, H, N, depth, 3003, threads=8, mode=6, extra=extra)
        floor = max(rel_mse(lo_a, hi), rel_mse(lo_b, hi))
        np.savez_compressed(os.path.join(HERE, "converged", n + "_pathphoton.npz"), film=hi.astype(np.float32), W=W, H=H, N=N, depth=depth, mode=6,
                            caustic_photon_count=photons, photon_gather_count=k, caustic_gather_radius=np.float32(radius),
                            noise_floor_relmse=floor, ref_msamples_per_s_8threads=W * H * 4 * N / secs * 1e-6)
        print("converged pathphoton", n, "floor", floor, "ref Msamples/s", W * H * 4 * N / secs * 1e-6)


def make_bdpt_converged():
    """BDPT films of the real reference at 4N spp plus two N-spp films of other seeds (the Monte-Carlo noise floor)."""
    for n, depth in (("caustics_test", 16), ("cornell_box_quad", 12)):
        W, H, N = 64, 36, 256
        hi, secs = ref_film(n, W, H, 4 * N, depth, 1001, threads=8, mode=2)
        lo_a, _ = ref_film(n, W, H, N, depth, 2002, threads=8, mode=2)
        lo_b, _ = ref_film(n, W, H, N, depth, 3003, threads=8, mode=2)
        floor = max(rel_mse(lo_a, hi), rel_mse(lo_b, hi))
        np.savez_compressed(os.path.join(HERE, "converged", n + "_bdpt.npz"), film=hi.astype(np.float32), W=W, H=H, N=N, depth=depth, mode=2,
                            noise_floor_relmse=floor, ref_msamples_per_s_8threads=W * H * 4 * N / secs * 1e-6)
        print("converged bdpt", n, "floor", floor, "ref Msamples/s", W * H * 4 * N / secs * 1e-6)


def make_c5_fixture(n_triangles=10_000_000, n_random=1_000_000):
    """hits/c5.npz: the synthetic BVH stress configuration (stress_base.json + synthetic.triangle_soup(10 M, seed 42)) through
    the real reference's Scene::hit: the digest (synthetic.hit_table_digest) of the closest-hit table of 1 M incoherent rays
    and of the 1920x1080 pixel-centre rays, plus the first 10 000 records of each verbatim.  The triangles and rays are pure
    functions of their seeds, so the fixture holds no geometry."""
    sd = pkg.scene_loader.load(scene_path("stress_base"))
    soup = pkg.synthetic.triangle_soup(n_triangles, seed=42)
    cam = sd.camera(1920, 1080)
    sets = {"random": pkg.synthetic.stress_rays(n_random, seed=4242), "pixels": pkg.synthetic.pixel_centre_rays(cam, 1920, 1080)}
    out = dict(n_triangles=n_triangles, soup_seed=42, n_random=n_random, ray_seed=4242, width=1920, height=1080,
               camera=np.array([cam.origin[:], cam.lower_left_corner[:], cam.horizontal[:], cam.vertical[:]]))
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        tp = os.path.join(td, "tris.bin")
        tris_to_bin(soup).tofile(tp)
        for name, rays in sets.items():
            rp, op = os.path.join(td, "rays.bin"), os.path.join(td, "hits.bin")
            np.ascontiguousarray(rays, dtype=np.float64).tofile(rp)
            run(["hits", scene_path("stress_base"), rp, op, tp])
            hits = np.fromfile(op, dtype=HIT_DTYPE)
            assert len(hits) == len(rays)
            out[name + "_digest"] = pkg.synthetic.hit_table_digest(hits)
            out[name + "_first"] = hits[:10000].copy()
            out[name + "_n_hit"] = int(hits["hit"].sum())
            out[name + "_n_tri"] = int((hits["prim_type"] == 2).sum())
            print("c5", name, len(rays), "hit", out[name + "_n_hit"], "triangles", out[name + "_n_tri"], out[name + "_digest"])
    np.savez_compressed(os.path.join(HERE, "hits", "c5.npz"), **out)


def rel_mse(x, r):
    x = np.clip(x, 0.0, 1.0); r = np.clip(r, 0.0, 1.0)   # the film Renderer::render returns is clamped
    return float(np.mean((x - r) ** 2 / (r ** 2 + 1e-2)))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "c5":   # one fixture only (it takes minutes: the reference builds its 10 M-triangle tree on one thread)
        make_c5_fixture()
        return
    if not os.path.exists(REF):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "oracle"), "ref"])
    for sub in ("loader", "hits", "bvh", "films", "converged", "kat"):
        os.makedirs(os.path.join(HERE, sub), exist_ok=True)
    if "--mnee-only" in sys.argv:
        make_mnee_fixtures()
        return
    if "--whitted-only" in sys.argv:
        make_whitted_fixtures()
        return
    if "--pathphoton-only" in sys.argv:
        make_pathphoton_fixtures()
        make_pathphoton_converged()
        return
    if "--pathphoton-converged-only" in sys.argv:
        make_pathphoton_converged()
        return
    if "--bdpt-only" in sys.argv:
        make_bdpt_fixtures()
        make_bdpt_converged()
        return
    if "--normal-only" in sys.argv:
        make_normal_map_fixtures()
        return
    make_post_fixtures()
    if "--post-only" in sys.argv:
        return

    names = ["cornell_box", "cornell_box_quad", "caustics_test", "environment_demo", "earth_demo", "demo", "kitchen_sink", "textured", "disk_room"]
    docs = {n: dump_loader(n) for n in names}
    print("loader dumps:", len(docs))

    # --- closest hits -------------------------------------------------------------------
    boxes = {"cornell_box": ((0, 2.5, -2.0), (2.4, 2.4, 3.5)), "cornell_box_quad": ((0, 2.5, -2.0), (2.4, 2.4, 3.5)),
             "environment_demo": ((0, 0.5, 1.0), (4, 2, 4)), "demo": ((0, 1, 0), (4, 2, 4)), "kitchen_sink": ((0, 1.5, -2), (4, 2, 4)), "textured": ((0, 1.5, -2), (4, 2, 4)), "disk_room": ((0, 1.5, -1.5), (3, 1.5, 3))}
    for n, (c, h) in boxes.items():
        rays = np.concatenate([camera_rays(docs[n], 48, 27), random_rays(3000, 17, c, h)], axis=0)
        hits = ref_hits(n, rays)
        np.savez_compressed(os.path.join(HERE, "hits", n + ".npz"), rays=rays, hits=hits)
        print("hits", n, len(rays), int(hits["hit"].sum()))
    # triangle soup appended to the quad-light Cornell box: exercises the tree walk
    soup = pkg.synthetic.triangle_soup(4000, seed=7, extent=2.2, size=0.25, n_materials=5)
    soup["v0"] += (0, 2.5, -2.5); soup["v1"] += (0, 2.5, -2.5); soup["v2"] += (0, 2.5, -2.5)
    rays = np.concatenate([camera_rays(docs["cornell_box_quad"], 64, 64), random_rays(6000, 23, (0, 2.5, -2.5), (2.4, 2.4, 2.4))], axis=0)
    hits = ref_hits("cornell_box_quad", rays, soup)
    np.savez_compressed(os.path.join(HERE, "hits", "soup.npz"), rays=rays, hits=hits, soup_n=4000, soup_seed=7)
    b, ii, pr = ref_bvh("cornell_box_quad", soup)
    np.savez_compressed(os.path.join(HERE, "bvh", "soup.npz"), boxes=b, ints=ii, prims=pr)
    b, ii, pr = ref_bvh("kitchen_sink")
    np.savez_compressed(os.path.join(HERE, "bvh", "kitchen_sink.npz"), boxes=b, ints=ii, prims=pr)
    print("soup hits", int(hits["hit"].sum()), "bvh nodes", len(ii))

    # --- seeded single-thread films (bit-reproducible) ----------------------------------
    films = [("cornell_box", 32, 18, 4, 500, 77, 1, 1), ("cornell_box", 24, 14, 4, 40, 78, 1, 0), ("cornell_box", 24, 14, 4, 40, 79, 0, 0),
             ("cornell_box_quad", 24, 24, 4, 50, 80, 1, 1), ("caustics_test", 24, 24, 4, 16, 81, 1, 1),
             ("environment_demo", 32, 18, 4, 50, 82, 1, 1), ("environment_demo", 24, 14, 3, 50, 85, 1, 0),
             ("demo", 32, 18, 4, 50, 84, 1, 1),
             ("textured", 32, 18, 4, 12, 89, 1, 1), ("kitchen_sink", 40, 24, 6, 24, 86, 1, 1), ("kitchen_sink", 24, 14, 4, 24, 87, 1, 0), ("kitchen_sink", 24, 14, 4, 24, 88, 0, 1),
             # disk lights: with NEE and MIS both on the reference indexes materials[-1000] in its look-ahead
             # (engine/renderer.cpp:803) and crashes, so its disk-light behaviour can only be pinned with one of them off
             ("disk_room", 32, 18, 6, 12, 90, 1, 0), ("disk_room", 24, 14, 4, 12, 91, 0, 1)]
    for (n, W, H, spp, depth, seed, nee, mis) in films:
        film, _ = ref_film(n, W, H, spp, depth, seed, nee, mis)
        tag = f"{n}_{W}x{H}_s{spp}_d{depth}_seed{seed}_nee{nee}_mis{mis}"
        np.savez_compressed(os.path.join(HERE, "films", tag + ".npz"), film=film, scene=n, W=W, H=H, spp=spp, depth=depth, seed=seed, nee=nee, mis=mis)
        print("film", tag, float(film.mean()))

    make_mnee_fixtures()
    make_whitted_fixtures()
    make_normal_map_fixtures()
    make_bdpt_fixtures()
    make_pathphoton_fixtures()

    # --- converged references + noise floor ----------------------------------------------
    for n, depth in (("cornell_box", 500), ("cornell_box_quad", 50), ("caustics_test", 16), ("environment_demo", 50), ("kitchen_sink", 24)):
        W, H, N = 64, 36, 256
        hi, secs = ref_film(n, W, H, 4 * N, depth, 1001, threads=8)
        lo_a, _ = ref_film(n, W, H, N, depth, 2002, threads=8)
        lo_b, _ = ref_film(n, W, H, N, depth, 3003, threads=8)
        floor = max(rel_mse(lo_a, hi), rel_mse(lo_b, hi))
        np.savez_compressed(os.path.join(HERE, "converged", n + ".npz"), film=hi.astype(np.float32), W=W, H=H, N=N, depth=depth,
                            noise_floor_relmse=floor, mean_lo_a=lo_a.mean(axis=(0, 1)), mean_lo_b=lo_b.mean(axis=(0, 1)),
                            ref_msamples_per_s_8threads=W * H * 4 * N / secs * 1e-6)
        print("converged", n, "floor", floor, "ref Msamples/s", W * H * 4 * N / secs * 1e-6)

    make_bdpt_converged()
    make_pathphoton_converged()

    # --- scalar known answers ---------------------------------------------------------------
    for n in ("kitchen_sink", "environment_demo", "cornell_box", "textured", "disk_room"):
        with tempfile.TemporaryDirectory() as td:
            op = os.path.join(td, "kat.bin")
            run(["kat", scene_path(n), 4242, 48, op])
            np.save(os.path.join(HERE, "kat", n + ".npy"), np.fromfile(op, dtype=np.float64))
    print("done")


if __name__ == "__main__":
    main()
