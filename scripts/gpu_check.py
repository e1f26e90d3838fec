"""Developer check on a GPU box: parity of the HIP path against the CPU oracle + a timing."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import rtx_pkg
pkg = rtx_pkg.load()
import oracle_py as O
abi = pkg.abi

def rays_for(sd, W, H, n_rand, seed):
    cam = sd.camera(W, H)
    o = np.array(cam.origin[:]); llc = np.array(cam.lower_left_corner[:]); hz = np.array(cam.horizontal[:]); vt = np.array(cam.vertical[:])
    xs, ys = np.meshgrid(np.arange(W), np.arange(H))
    s = ((xs + 0.5) / W).ravel(); t = ((ys + 0.5) / H).ravel()
    d = (llc[None] + s[:, None] * hz[None]) + t[:, None] * vt[None] - o[None]
    prim = np.concatenate([np.tile(o, (W * H, 1)), d, np.full((W * H, 1), 0.001), np.full((W * H, 1), 1e9)], axis=1)
    rng = np.random.default_rng(seed)
    ro = rng.uniform(-2.4, 2.4, (n_rand, 3)) + np.array([0, 2.5, -2.5])
    rd = rng.normal(size=(n_rand, 3))
    rd[: n_rand // 8, 0] = 0.0     # axis-parallel components
    rd[n_rand // 8: n_rand // 4, 1] = 0.0
    rnd = np.concatenate([ro, rd, np.full((n_rand, 1), 0.001), rng.choice([1e9, 3.0, 0.5], (n_rand, 1))], axis=1)
    return np.concatenate([prim, rnd], axis=0)

def main():
    assert pkg.renderer.test_hip(0)
    for name in ["cornell_box", "cornell_box_quad", "caustics_test", "environment_demo", "demo"]:
        sd = pkg.scene_loader.load(os.path.join(ROOT, "tests/golden/scenes", name + ".json"))
        osc = O.OracleScene(sd)
        rays = rays_for(sd, 64, 36, 20000, 5)
        exp = osc.trace_closest(rays)
        with pkg.renderer.HipPathTracer(0) as pt:
            pt.upload_scene(sd)
            got = pt.trace_closest(rays)
            fast64 = pt.trace_closest(rays, abi.PRECISION_F64)
            fast32 = pt.trace_closest(rays, abi.PRECISION_F32)
            def cmp(a, tag):
                same = (a["hit"] == exp["hit"]) & (a["prim_type"] == exp["prim_type"]) & (a["prim_index"] == exp["prim_index"])
                tbits = (a["t"].view(np.uint64) == exp["t"].view(np.uint64))
                nb = (a["normal"].view(np.uint64) == exp["normal"].view(np.uint64)).all(axis=1)
                pb = (a["point"].view(np.uint64) == exp["point"].view(np.uint64)).all(axis=1)
                uv = max(np.abs(a["u"] - exp["u"]).max(), np.abs(a["v"] - exp["v"]).max())
                print(f"  {tag}: id mismatches {np.sum(~same)}, t-bit mismatches {np.sum(~tbits)}, normal {np.sum(~nb)}, point {np.sum(~pb)}, uv maxabs {uv:.2e}, mat {np.sum(a['material_id'] != exp['material_id'])}")
            print(name, "rays", len(rays), "hits", exp["hit"].sum())
            cmp(got, "ref-order f64"); cmp(fast64, "fast f64"); cmp(fast32, "fast f32")
            W, H, spp = 64, 36, 16
            over = dict(width=W, height=H, samples_per_pixel=spp, max_depth=min(sd.max_depth, 50), seed=11)
            ref, ost = osc.render(sd.camera(W, H), sd.settings(**over), O.RNG_COUNTER, threads=16)
            pt.set_camera(sd.camera(W, H))
            pt.set_settings(sd.settings(precision=abi.PRECISION_F64, **over))
            f64 = pt.render().astype(np.float64); s64 = pt.last_stats
            pt.set_settings(sd.settings(precision=abi.PRECISION_F32, **over))
            f32 = pt.render().astype(np.float64); s32 = pt.last_stats
            d = np.abs(f64 - ref)
            print(f"  render f64: max abs {d.max():.3e}, frac>1e-4rel {np.mean(d > 1e-4 * (1 + ref)):.4f}, mean gpu {f64.mean():.6f} oracle {ref.mean():.6f}")
            print(f"  render f32: mean {f32.mean():.6f}; rays closest {s32.rays_closest} shadow {s32.rays_shadow} (oracle {ost.rays_closest}/{ost.rays_shadow}) f64 {s64.rays_closest}/{s64.rays_shadow}")
    # timing: Cornell 1080p, a few spp settings
    sd = pkg.scene_loader.load(os.path.join(ROOT, "tests/golden/scenes/cornell_box.json"))
    with pkg.renderer.HipPathTracer(0) as pt:
        pt.upload_scene(sd)
        pt.set_camera(sd.camera(1920, 1080))
        for spp in (16, 64, 256):
            pt.set_settings(sd.settings(width=1920, height=1080, samples_per_pixel=spp, max_depth=500))
            pt.render(); st = pt.last_stats
            rays = st.rays_closest + st.rays_shadow
            print(f"C2 1080p spp={spp}: {st.seconds_render:.3f}s  {st.samples / st.seconds_render / 1e6:.1f} Msamples/s  {rays / st.seconds_render / 1e9:.2f} Grays/s  seg/sample {st.segments / st.samples:.2f}", flush=True)

if __name__ == "__main__":
    main()
