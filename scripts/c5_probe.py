"""BVH stress scene probe: build time, closest-hit parity sample vs the CPU oracle, render throughput."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import rtx_pkg
pkg = rtx_pkg.load(); abi = pkg.abi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
check = int(sys.argv[3]) if len(sys.argv) > 3 else 0
t0 = time.time(); sd = pkg.synthetic.stress_scene(n); print(f"scene gen {time.time()-t0:.2f}s", flush=True)
with pkg.renderer.HipPathTracer(0) as pt:
    t0 = time.time(); pt.upload_scene(sd); up = time.time() - t0
    info = pt.bvh_info()
    print(f"n_tris {n}: upload {up:.2f}s (host BVH build {info.build_seconds:.2f}s), nodes {info.ref_nodes}, node bytes {info.wide_bytes/1e6:.1f} MB", flush=True)
    W, H = 1920, 1080
    pt.set_camera(sd.camera(W, H))
    if check:
        import oracle_py as O
        rng = np.random.default_rng(3)
        m = check
        ro = rng.uniform(-11, 11, (m, 3)); rd = rng.normal(size=(m, 3))
        rays = np.concatenate([ro, rd, np.full((m, 1), 0.001), np.full((m, 1), 1e9)], axis=1)
        t0 = time.time(); osc = O.OracleScene(sd); print(f"oracle build {time.time()-t0:.2f}s", flush=True)
        exp = osc.trace_closest(rays, threads=16)
        got = pt.trace_closest(rays)
        f64 = pt.trace_closest(rays, abi.PRECISION_F64)
        f32 = pt.trace_closest(rays, abi.PRECISION_F32)
        for tag, a in (("ref-order f64", got), ("walk f64", f64), ("walk f32", f32)):
            same = (a["hit"] == exp["hit"]) & (a["prim_type"] == exp["prim_type"]) & (a["prim_index"] == exp["prim_index"])
            tb = a["t"].view(np.uint64) == exp["t"].view(np.uint64)
            print(f"  {tag}: id mismatches {np.sum(~same)} / {m}, t-bit mismatches {np.sum(~tb)}, tri hits {np.sum(exp['prim_type']==2)}", flush=True)
    for pipe, s in ((abi.PIPELINE_FUSED, 1), (abi.PIPELINE_FUSED, spp), (abi.PIPELINE_WAVEFRONT, 1), (abi.PIPELINE_WAVEFRONT, spp), (abi.PIPELINE_WAVEFRONT, 4 * spp)):
        pt.set_settings(sd.settings(width=W, height=H, samples_per_pixel=s, max_depth=8, pipeline=pipe))
        f = pt.render(); st = pt.last_stats
        rays_n = st.rays_closest + st.rays_shadow
        print(f"  {'fused' if pipe == abi.PIPELINE_FUSED else 'wavefront'} mean {float(f.mean()):.5f} launches {st.kernel_launches} render {W}x{H} spp={s}: {st.seconds_main_kernel*1e3:.1f} ms  {st.samples/st.seconds_main_kernel/1e6:.1f} Msamples/s  {rays_n/st.seconds_main_kernel/1e9:.3f} Grays/s  seg/sample {st.segments/st.samples:.2f}", flush=True)
