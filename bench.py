#!/usr/bin/env python3
"""Headline benchmark: path tracing with NEE + MIS on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c5] [--multi process|single]

One "step" is one complete render of the configuration's workload:

  c2 (default, the configuration the headline metric is quoted on)
      scenes/cornell_box.json at 1920x1080, 1024 spp, max_depth 500, NEE + MIS, manifold NEE off
  c3  scenes/cornell_box_quad.json (depth 50) and scenes/caustics_test.json (depth 16, as a path tracer), 800x800, 128 spp each
  c5  the synthetic BVH stress scene: 10 M triangles, 1920x1080, 64 spp, max_depth 8

Run for c2 without --config, the line also carries c3 and c5, three steps each, under "configs".

N > 1: the image's rows are interleaved over the GPUs and the film is gathered on the first one inside the timed region; total
work is fixed, so scaling is "strong".  Two hosts for it:
  --multi process (default)  one process per GPU over torch.distributed (backend nccl = RCCL).  Started plainly
                             (`python bench.py --gpus 8`, no WORLD_SIZE in the environment) this process starts the N ranks
                             itself, as a child `python -m torch.distributed.run`, relays rank 0's line and exits with the child's
                             status; started by torch.distributed.run it is one of the ranks.
  --multi single             one process, one context over the N GPUs (rtx_create_multi), no torch.distributed.

Rank 0 prints ONE JSON line.  `value` = samples of all GPUs / max-over-ranks time.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
SCENES = os.path.join(ROOT, "tests", "golden", "scenes")

HBM_PEAK_GBPS = 8000.0           # MI355X HBM3E peak (spec)
# Vector-ALU issue ceiling: 256 CUs x 4 SIMDs at 2.4 GHz.  A SIMD takes four cycles per wave64 instruction and can issue two
# such instructions (of different waves) in one four-cycle slot: the hardware counts those slots (SQ_ACTIVE_INST_VALU2), and a
# stream of v_mov_b32 runs at 2.6 cycles per instruction (scripts/valu_peak.hip, profiles/r02_valu_peak.jsonl).  The ceiling is
# therefore two instructions per SIMD per four cycles; streams of one full-rate arithmetic opcode sustain 0.44 of it (v_fma_f32:
# 543 Gwave-inst/s, clocks included), v_rcp/v_sqrt/v_sin half of that.
N_SIMDS, CLOCK_HZ, VALU_CYCLES_PER_INST = 1024, 2.4e9, 2.0
VALU_STREAM_RATES = {"v_fma_f32": 542.9, "v_mov_b32": 955.6, "v_rcp_f32": 299.6}   # Gwave-inst/s at 8 waves per SIMD, launches of >= 100 ms
# What a queue-based path tracer would move per unit of work (path state written + read 96 B + hit record 32 B + two queue
# indices 16 B per path segment; 88 B per shadow ray; 24 B of film accumulation per sample): reported as an equivalent
# rate only; the fused kernel keeps that state in registers.
BYTES_PER_SEGMENT, BYTES_PER_SHADOW_RAY, BYTES_PER_SAMPLE = 144, 88, 24

CONFIGS = {
    "c2": [("cornell_box", 1920, 1080, 1024, 500)],
    "c3": [("cornell_box_quad", 800, 800, 128, 50), ("caustics_test", 800, 800, 128, 16)],
    "c5": [("stress10m", 1920, 1080, 64, 8)],
}
EXTRA_CONFIGS = ("c3", "c5")     # timed after the headline when it is c2
EXTRA_STEPS, EXTRA_WARMUP = 3, 1


def sha256_of(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--multi", choices=["process", "single"], default="process", help="N > 1: one process per GPU (torch.distributed) or one process over all GPUs (rtx_create_multi)")
    ap.add_argument("--device-ids", default="", help="--multi single: comma-separated device list instead of 0..N-1 (a device may repeat: rehearsal on fewer GPUs)")
    ap.add_argument("--gather", choices=["peer", "rccl"], default="peer", help="--multi single: how the film reaches the first device")
    ap.add_argument("--spp", type=int, default=0, help="override the configuration's samples per pixel (the line then says so)")
    ap.add_argument("--pipeline", choices=["auto", "fused", "wavefront"], default="auto")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0, help="0 disables the CPU baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the double-precision and id-agreement legs and the c3 / c5 lines (profiling runs)")
    ap.add_argument("--no-configs", action="store_true", help="skip the c3 / c5 lines only")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="--multi process on a machine with fewer GPUs: every rank renders on GPU 0 and the ranks meet over gloo (films staged through host memory); exercises the rank code, the timings mean nothing")
    ap.add_argument("--dry-launch", action="store_true", help="launcher rehearsal without a GPU: the ranks meet over gloo, render nothing, rank 0 prints a stub line")
    return ap.parse_args(argv)


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpus():
    import torch
    return torch.cuda.device_count()   # counts devices without initialising one


def launch_ranks(args, argv):
    """`python bench.py --gpus N` started plainly: run the N ranks as a child torch.distributed.run.  The parent makes no GPU call
    (a process that has touched the GPU must not be replaced, and needs not be: the child is a child)."""
    if not args.dry_launch:
        have = visible_gpus()
        if have < (1 if args.rehearse_on_one_gpu else args.gpus):
            print(f"bench.py: --gpus {args.gpus} asked for, {have} GPU(s) visible on this node: nothing was started", file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or args.gpus) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode   # stdout is inherited: rank 0's line is this process's line


def dry_rank(args):
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "rank_sum": float(t.item()), "steps": args.steps, "warmup": args.warmup,
                          "config": {"workload": "none (launcher rehearsal)", "name": args.config}}), flush=True)
    dist.destroy_process_group()
    return 0


class Runner:
    """One configuration's scenes uploaded, and the step that renders them all and leaves the whole film(s) on the first GPU."""

    def __init__(self, pkg, torch, dist, args, cfg, world, rank, local_rank, single_ids):
        abi = pkg.abi
        self.pkg, self.torch, self.dist, self.cfg = pkg, torch, dist, cfg
        self.world, self.rank, self.single_ids = world, rank, single_ids
        self.dev = f"cuda:{single_ids[0] if single_ids else local_rank}"
        self.stream = torch.cuda.current_stream().cuda_stream
        pipeline = {"auto": abi.PIPELINE_AUTO, "fused": abi.PIPELINE_FUSED, "wavefront": abi.PIPELINE_WAVEFRONT}[args.pipeline]
        self.jobs, self.upload_s, self.stats = [], 0.0, []
        shards = world if not single_ids else 1           # a multi-device context shards inside the library
        for (name, W, H, spp, depth) in CONFIGS[cfg]:
            spp = args.spp or spp
            sd = pkg.synthetic.stress_scene(10_000_000) if name == "stress10m" else pkg.scene_loader.load(os.path.join(SCENES, name + ".json"))
            st = sd.settings(width=W, height=H, samples_per_pixel=spp, max_depth=depth, use_nee=1, use_mis=1, seed=1234, mode=abi.MODE_PATHTRACE,
                             precision=abi.PRECISION_F32, pipeline=pipeline, shard_index=rank if shards > 1 else 0, shard_count=shards)
            pt = pkg.renderer.HipPathTracer(device_ids=single_ids) if single_ids else pkg.renderer.HipPathTracer(local_rank)
            if single_ids and args.gather == "rccl":
                pt.set_gather(abi.GATHER_RCCL)
            pt.set_settings(st)
            t0 = time.perf_counter()
            pt.upload_scene(sd)                       # inputs are resident in HBM before the timed region
            self.upload_s += time.perf_counter() - t0
            pt.set_camera(sd.camera(W, H))
            self.jobs.append(dict(name=name, W=W, H=H, spp=spp, depth=depth, sd=sd, pt=pt, settings=st))
        # Process mode: every rank's rows of every scene lie in ONE buffer (parallel.ShardedFilms), so a step gathers once.
        import importlib
        parallel = importlib.import_module("raytracer_amd.parallel")
        self.films = parallel.ShardedFilms([(j["H"], j["W"]) for j in self.jobs], rank if shards > 1 else 0, shards, self.dev, via_host=args.rehearse_on_one_gpu)
        for j, local in zip(self.jobs, self.films.local):
            j["local"] = local
        self.gather_events = []

    def step(self):
        """-> the films on the first GPU (None on the other ranks)."""
        torch = self.torch
        if self.single_ids or self.world == 1:
            for j in self.jobs:
                self.stats.append(j["pt"].render_device(j["local"].data_ptr(), self.stream))
            return [j["local"] for j in self.jobs]
        for j in self.jobs:                                   # every scene's kernels go on the stream first,
            j["pt"].render_device_async(j["local"].data_ptr(), self.stream)
        if self.rank == 0:                                    # then ONE gather of all of them,
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            films = self.films.gather()
            e1.record()
            self.gather_events.append((e0, e1))
        else:
            films = self.films.gather()
        for j in self.jobs:                                   # and only then the host waits
            self.stats.append(j["pt"].render_wait())
        return films

    def settle_background(self):
        """rtx_upload_scene returns before the reference-order tree of a large scene is built (a host thread finishes it): wait for
        it here, ahead of the timed region, so that the build's threads do not compete with the launch loop being timed."""
        self.tree_s = sum(float(j["pt"].bvh_info().build_seconds) for j in self.jobs)

    def close(self):
        for j in self.jobs:
            j["pt"].close()


def measure(runner, torch, dist, world, steps, warmup):
    """W untimed steps, then exactly K steps between barrier + synchronize; -> (films, seconds = max over ranks, samples, rays)."""
    def fence():
        if world > 1 and not runner.single_ids:
            dist.barrier()
        torch.cuda.synchronize()

    films = None
    for _ in range(warmup):
        films = runner.step()
    runner.settle_background()
    runner.stats.clear()
    runner.gather_events.clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        films = runner.step()
    fence()
    elapsed = time.perf_counter() - t0
    where = "cpu" if runner.films.via_host else runner.dev   # (gloo takes host tensors only)
    tt = torch.tensor([elapsed], dtype=torch.float64, device=where)
    tot = torch.tensor([float(sum(s.samples for s in runner.stats)), float(sum(s.rays_closest + s.rays_shadow for s in runner.stats))], dtype=torch.float64,
                       device=where)
    if world > 1 and not runner.single_ids:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    return films, float(tt.item()), float(tot[0].item()), float(tot[1].item())


def config_line(args, abi, runner, films, elapsed, total_samples, total_rays, steps, world, parallelism):
    jobs = runner.jobs
    expect = sum(j["W"] * j["H"] * j["spp"] for j in jobs) * steps
    assert films is not None and all(tuple(f.shape) == (j["H"], j["W"], 3) for f, j in zip(films, jobs))
    assert total_samples == float(expect), "work was skipped inside the timed region"
    workload = " + ".join(f"{j['name']} {j['W']}x{j['H']} {j['spp']}spp max_depth={j['depth']}" for j in jobs) + " NEE+MIS mnee=off"
    last = jobs[0]["pt"].last_stats
    n_jobs = len(jobs)
    mine = runner.stats                               # rank 0's (or the whole context's) statistics, scene by scene
    kernel_s = sum(s.seconds_render for s in mine) / steps
    out = {
        "value": total_samples / elapsed / 1e6, "unit": "Msamples/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps,
        "config": {"workload": workload, "name": runner.cfg, "parallelism": parallelism,
                   "pipeline": {abi.PIPELINE_FUSED: "fused", abi.PIPELINE_WAVEFRONT: "wavefront"}.get(last.pipeline_used, "?"),
                   "seed": 1234, "scene_upload_and_bvh_s": runner.upload_s,
                   "scene_upload_note": "rtx_upload_scene calls: tables and the eight-wide tree on the device; the reference-order tree of a scene of 2^18 or more primitives is finished in the background",
                   "reference_order_tree_s": getattr(runner, "tree_s", None)},
        "mrays_per_s": total_rays / elapsed / 1e6,
        "mrays_per_s_per_gpu": total_rays / elapsed / 1e6 / world,
        "msamples_per_s_per_gpu": total_samples / elapsed / 1e6 / world,
        "film_mean_clamped": [float(f.clamp(0, 1).mean().item()) for f in films],
        # what a step spends outside this GPU's render kernels: launches, waiting for the slowest GPU, the film gather, synchronisation
        "kernel_ms_per_step_this_gpu": kernel_s * 1e3,
        "non_kernel_ms_per_step": (elapsed / steps - kernel_s) * 1e3,
        "roofline": roofline(runner.cfg, abi, jobs, mine, world, n_jobs),
    }
    if runner.gather_events:
        # rank 0's view: waiting for the slowest rank + the transfers + re-interleaving the rows
        g_ms = sum(a.elapsed_time(b) for a, b in runner.gather_events) / len(runner.gather_events)
        into_root = runner.films.bytes_into_root()
        out["film_gather"] = {"ms_per_step_incl_wait": g_ms, "bytes_into_rank0_per_step": into_root, "gathers_per_step": 1,
                              "GBps_lower_bound": into_root / (g_ms * 1e-3) / 1e9 if g_ms > 0 else None}
    elif world > 1:
        out["film_gather"] = {"inside": "rtx_render_device of the multi-device context (rtx_create_multi)", "mode": args.gather,
                              "bytes_into_device0_per_step": sum((j["H"] - len(range(0, j["H"], world))) * j["W"] * 3 * 4 for j in jobs)}
    return out


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    in_rank = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and args.multi == "process" and not in_rank:
        sys.exit(launch_ranks(args, argv))
    if args.dry_launch:
        sys.exit(dry_rank(args) if in_rank else print(json.dumps({"dry_launch": True, "n_gpus": 1})) or 0)

    single_ids = None
    if args.multi == "single" and (args.gpus > 1 or args.device_ids):
        single_ids = [int(v) for v in args.device_ids.split(",")] if args.device_ids else list(range(args.gpus))
        if len(single_ids) != args.gpus:
            raise SystemExit(f"--device-ids lists {len(single_ids)} devices, --gpus says {args.gpus}")
    if not in_rank:
        need = (max(single_ids) + 1) if single_ids else 1
        have = visible_gpus()
        if have < need:
            print(f"bench.py: {need} GPU(s) needed, {have} visible on this node: nothing was started", file=sys.stderr)
            sys.exit(2)

    import numpy as np
    import torch
    import torch.distributed as dist
    import rtx_pkg
    pkg = rtx_pkg.load()
    abi = pkg.abi

    world = int(os.environ.get("WORLD_SIZE", "1")) if not single_ids else args.gpus
    rank = int(os.environ.get("RANK", "0")) if not single_ids else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if not single_ids else single_ids[0]
    if args.rehearse_on_one_gpu:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} disagrees with WORLD_SIZE {world}")
    torch.cuda.set_device(local_rank)
    if world > 1 and not single_ids:
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    parallelism = ("single GPU" if world == 1 else
                   f"row-interleaved x{world}, one process over all GPUs (rtx_create_multi, devices {single_ids}), film gather by {'RCCL send/recv' if args.gather == 'rccl' else 'peer copies'}" if single_ids else
                   f"row-interleaved x{world}, one process per GPU, one film gather per step over RCCL" + (" (REHEARSAL: all ranks on GPU 0, gloo through host memory)" if args.rehearse_on_one_gpu else ""))

    runner = Runner(pkg, torch, dist, args, args.config, world, rank, local_rank, single_ids)
    films, elapsed, total_samples, total_rays = measure(runner, torch, dist, world, args.steps, args.warmup)
    out = None
    if rank == 0:
        head = config_line(args, abi, runner, films, elapsed, total_samples, total_rays, args.steps, world, parallelism)
        out = {"metric": "Msamples/s (NEE+MIS path tracing; Mrays/s alongside)", "value": head.pop("value"), "unit": head.pop("unit"),
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head.pop("ms_per_step"),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic (scenes built from the reference's scene descriptions / generated in code; no image assets)"}
        head.pop("steps")
        out.update(head)
        if world == 1 and not args.no_extras:
            out["f64"] = f64_leg(pkg, runner.jobs[0])
            out["id_mismatch_f32"] = id_mismatch_leg(pkg, np, runner.jobs[0])
    runner.close()
    del runner, films

    # The other configurations, three steps each, in the same line (the headline above is untouched by them).
    if args.config == "c2" and not args.no_extras and not args.no_configs:
        extras = {}
        for cfg in EXTRA_CONFIGS:
            r2 = Runner(pkg, torch, dist, args, cfg, world, rank, local_rank, single_ids)
            f2, e2, s2, y2 = measure(r2, torch, dist, world, EXTRA_STEPS, EXTRA_WARMUP)
            if rank == 0:
                line = config_line(args, abi, r2, f2, e2, s2, y2, EXTRA_STEPS, world, parallelism)
                line["warmup"] = EXTRA_WARMUP
                if cfg == "c5" and world == 1:
                    line["id_mismatch_f32"] = id_mismatch_leg(pkg, np, r2.jobs[0])
                extras[cfg] = line
            r2.close()
            del r2, f2
        if rank == 0:
            out["configs"] = extras

    if rank == 0:
        if world == 1 and args.cpu_baseline_seconds > 0:
            job0 = dict(name=CONFIGS[args.config][0][0], depth=CONFIGS[args.config][0][4])
            out["cpu_baseline"] = cpu_baseline(pkg, job0, args)
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            out["cpu_baseline"]["gpu_over_cpu_note"] = "Msamples/s of the full-size GPU workload over Msamples/s of the CPU sample (reduced resolution, this job's share of the host's cores); a reported ratio, not vs_baseline"
        print(json.dumps(out), flush=True)
    if world > 1 and not single_ids:
        dist.destroy_process_group()


def roofline(cfg, abi, jobs, stats, world, n_jobs):
    """The dominant kernel of rank 0's launches against the ceiling that bounds it.

    Fused pipeline (c2, c3): the scene sits in scalar registers and the path in vector registers, HBM sees next to nothing:
    the bound is vector-ALU issue.  achieved = VALU wave-instructions per launch / the kernel time measured live (HIP events
    on the launch stream, rtx_stats.seconds_main_kernel).  The instruction count is a property of (kernel binary, workload):
    every lane's work items are fixed by its index, so it is taken from the committed rocprofv3 --pmc profile of this very
    workload - and only if that profile names the shared library that has just been timed (SHA-256), else it is null.
    Queue pipeline (c5): the closest-hit kernel walks every queued ray through the eight-wide tree, one 64-byte record per step,
    scattered over a tree far larger than the caches.  Its profile shows vector-ALU issue saturated (VALUBusy above 100 %) at the
    same instruction rate the fused kernel reaches, with the memory side at well under half its peak: the bound is the same
    VALU ceiling, achieved = VALU wave-instructions of the kernel's launches (committed profile of this library build and
    workload) / the kernel's live time (HIP events around every launch).  The memory side is reported beside it under "hbm":
    achieved = records fetched (counted live by the kernel) x 64 bytes / the same time, against the 8 TB/s peak; traffic = what
    reached the L2's memory side in the profile, FETCH_SIZE as tallied + WRITE_SIZE (the counter tallies 64 bytes per request,
    exact for scattered 64-byte records, half the truth for streaming reads: scripts/gather_calib.sh); the difference to the
    record bytes is what the caches served."""
    first = [s for i, s in enumerate(stats) if i % n_jobs == 0]   # scene 0 of every step
    n = len(first)
    seg = sum(s.segments for s in first) / n
    shd = sum(s.rays_shadow for s in first) / n
    smp = sum(s.samples for s in first) / n
    kern_s = sum(s.seconds_main_kernel for s in first) / n
    j0 = jobs[0]
    workload = [j0["name"], j0["W"], j0["H"], j0["spp"], j0["depth"], world]
    prof = None
    try:
        allp = json.load(open(os.path.join(ROOT, "profiles", "bench_profiles.json")))
        cand = allp.get(cfg)
        if cand and cand.get("workload") == workload and cand.get("lib_sha256") == sha256_of(abi.LIB_PATH):
            prof = cand
    except (OSError, ValueError):
        pass
    fused = first[0].pipeline_used == abi.PIPELINE_FUSED
    r = {"kernel_ms": kern_s * 1e3, "profile": "profiles/bench_profiles.json" if prof else None,
         "profile_note": None if prof else "no committed PMC profile names this library build and workload: counter-derived figures are null"}
    if fused:
        peak = N_SIMDS * CLOCK_HZ / VALU_CYCLES_PER_INST / 1e9
        r.update({"bound": "valu", "unit": "Gwave-inst/s", "peak": peak, "kernel": "render_fused_kernel<float>",
                  "peak_definition": "two wave64 instructions per SIMD per 4 cycles (dual issue) x 1024 SIMDs x 2.4 GHz",
                  "single_issue_rate": peak / 2, "measured_stream_rates": VALU_STREAM_RATES,
                  "achieved": None, "frac": None, "lanes_active": None, "valu_busy": None, "traffic": None, "hbm_frac": None})
        if prof:
            ach = prof["SQ_INSTS_VALU"] / kern_s / 1e9
            r.update({"achieved": ach, "frac": ach / peak, "lanes_active": prof.get("lanes_active"), "valu_busy": prof.get("valu_busy"),
                      "traffic": prof.get("hbm_bytes_per_launch"),
                      "hbm_frac": (prof["hbm_bytes_per_launch"] / kern_s / 1e9 / HBM_PEAK_GBPS) if prof.get("hbm_bytes_per_launch") else None,
                      "valu_insts_per_segment": prof["SQ_INSTS_VALU"] * 64.0 / seg, "kernel_symbol": prof.get("kernel"),
                      "kernel_ms_rocprof": prof.get("kernel_ms_rocprof")})
        r["wavefront_equivalent_GBps"] = (BYTES_PER_SEGMENT * seg + BYTES_PER_SHADOW_RAY * shd + BYTES_PER_SAMPLE * smp) / kern_s / 1e9
    else:
        ext_s = sum(s.seconds_extend for s in first) / n
        nodes = sum(s.nodes_visited for s in first) / n
        prims = sum(s.prims_tested for s in first) / n
        rays = sum(s.rays_closest for s in first) / n
        info = j0["pt"].bvh_info()
        rec_b = info.wide_bytes / max(1, info.wide_nodes)
        peak = N_SIMDS * CLOCK_HZ / VALU_CYCLES_PER_INST / 1e9
        r.update({"bound": "valu", "unit": "Gwave-inst/s", "peak": peak, "kernel": "wf_extend (closest-hit walk of the ray queue through the eight-wide tree)",
                  "peak_definition": "two wave64 instructions per SIMD per 4 cycles (dual issue) x 1024 SIMDs x 2.4 GHz",
                  "single_issue_rate": peak / 2, "measured_stream_rates": VALU_STREAM_RATES,
                  "kernel_ms": ext_s * 1e3, "frame_ms": kern_s * 1e3, "achieved": None, "frac": None, "traffic": None, "lanes_active": None, "valu_busy": None,
                  # small launches (shards of a frame) run the any-hit walk of a turn beside the closest-hit walk: kernel_ms then includes shared time
                  "walks_side_by_side": bool(first[0].flags & abi.STAT_WALKS_SIDE_BY_SIDE)})
        if ext_s > 0 and nodes > 0:
            alg = nodes * rec_b + prims * info.leaf_bytes
            hbm = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBPS, "achieved": alg / ext_s / 1e9, "frac": alg / ext_s / 1e9 / HBM_PEAK_GBPS,
                   "achieved_definition": "node and leaf records fetched (counted live by the kernel) x 64 bytes / the kernel's live time",
                   "record_bytes_per_launch_set": alg, "traffic": None}
            r.update({"node_records_per_ray": nodes / rays, "leaf_records_per_ray": prims / rays, "node_record_bytes": rec_b, "leaf_record_bytes": info.leaf_bytes,
                      "grays_per_s_closest_hit": rays / ext_s / 1e9, "hbm": hbm})
            if prof:
                ach = prof["SQ_INSTS_VALU"] / ext_s / 1e9
                r.update({"achieved": ach, "frac": ach / peak, "lanes_active": prof.get("lanes_active"), "valu_busy": prof.get("valu_busy"),
                          "valu_wave_insts_per_ray": prof["SQ_INSTS_VALU"] / rays, "l2_hit_rate": prof.get("l2_hit_rate"),
                          "kernel_ms_rocprof": prof.get("kernel_ms_rocprof"), "kernel_symbol": prof.get("kernel")})
                if prof.get("hbm_bytes_per_launch"):
                    t = prof["hbm_bytes_per_launch"]
                    hbm.update({"traffic": t, "traffic_GBps": t / ext_s / 1e9, "traffic_frac_of_peak": t / ext_s / 1e9 / HBM_PEAK_GBPS,
                                "share_of_record_bytes_served_by_caches": 1.0 - t / alg, "traffic_rule": prof.get("hbm_bytes_rule")})
                    r["traffic"] = t
    return r


def f64_leg(pkg, job):
    """The same workload in double precision (the reference's arithmetic type), at fewer samples per pixel."""
    abi = pkg.abi
    pt, sd = job["pt"], job["sd"]
    spp = max(1, min(job["spp"], 32))
    pt.set_settings(sd.settings(width=job["W"], height=job["H"], samples_per_pixel=spp, max_depth=job["depth"], use_nee=1, use_mis=1, seed=1234,
                                mode=abi.MODE_PATHTRACE, precision=abi.PRECISION_F64))
    best = None
    for _ in range(2):
        s = pt.render_device(job["local"].data_ptr())
        best = s if best is None or s.seconds_render < best.seconds_render else best
    pt.set_settings(job["settings"])
    return {"spp": spp, "ms_per_step": best.seconds_render * 1e3, "msamples_per_s": best.samples / best.seconds_render / 1e6,
            "mrays_per_s": (best.rays_closest + best.rays_shadow) / best.seconds_render / 1e6,
            "pipeline": {abi.PIPELINE_FUSED: "fused", abi.PIPELINE_WAVEFRONT: "wavefront"}.get(best.pipeline_used, "?")}


def id_mismatch_leg(pkg, np, job):
    """Share of rays for which the single-precision production walk names another primitive than the double-precision
    reference-order kernel (which the tests pin to the real reference bit for bit)."""
    pt = job["pt"]
    fx = os.path.join(ROOT, "tests", "golden", "hits", job["name"] + ".npz")
    if job["name"] == "stress10m":
        rays, what = pkg.synthetic.stress_rays(1_000_000, seed=4242), "1 M seeded incoherent rays"
    elif os.path.exists(fx):
        rays, what = np.load(fx)["rays"], "the fixture's pixel-centre and random rays"
    else:
        return None
    ref = pt.trace_closest(rays)
    f32 = pt.trace_closest(rays, pkg.abi.PRECISION_F32)
    agree = (f32["hit"] == ref["hit"]) & (f32["prim_type"] == ref["prim_type"]) & (f32["prim_index"] == ref["prim_index"])
    return {"rate": 1.0 - float(agree.mean()), "rays": int(len(rays)), "ray_set": what}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(pkg, job, args):
    """CPU renderer timed on this host's cores on a bounded sample of the same workload: the REAL reference
    (oracle/_ref/ref_harness = its Renderer::render compiled in place, use_mnee = false) when that binary is present and the
    scene is one of its files, else the CPU oracle (a restatement of it).  Threads = the CPU share of the job: OMP_NUM_THREADS
    where the launcher sets it (a one-GPU box hands 16 of the host's cores to a job and asks for pools of that size), else the
    CPUs this process may run on."""
    import re
    import tempfile
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or allowed
    host = {"host_nproc": os.cpu_count(), "cpus_allowed": allowed, "omp_num_threads_env": os.environ.get("OMP_NUM_THREADS"), "cpu_model": cpu_model()}
    name, depth = job["name"], job["depth"]
    scene_file = os.path.join(SCENES, name + ".json")
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    W, H = 320, 180
    if os.path.exists(ref) and os.path.exists(scene_file):
        def run(spp):
            with tempfile.TemporaryDirectory() as td:
                o = subprocess.run([ref, "render", scene_file, str(W), str(H), str(spp), str(depth), "1", "1", "1", "0", str(threads),
                                    os.path.join(td, "f.bin")], cwd=os.path.join(ROOT, "tests", "golden"), capture_output=True, text=True, check=True).stdout
            return float(re.search(r"@@TIME@@ ([0-9.]+) s", o).group(1))
        try:
            t = run(4)
            spp = int(max(8, min(4096, args.cpu_baseline_seconds / max(t, 1e-3) * 4)))
            secs = run(spp)
            return {"value": W * H * spp / secs / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "reference", **host,
                    "sample": f"reference Renderer::render (use_mnee=false) on {name}.json {W}x{H} {spp}spp max_depth={depth} NEE+MIS, {secs:.1f} s"}
        except (subprocess.CalledProcessError, OSError, AttributeError) as e:
            print(f"reference baseline failed ({e}); falling back to the port", file=sys.stderr)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    sd = pkg.synthetic.stress_scene(10_000_000) if name == "stress10m" else pkg.scene_loader.load(scene_file)
    osc = O.OracleScene(sd)
    cam = sd.camera(W, H)
    probe_spp = 4
    _, ps = osc.render(cam, sd.settings(width=W, height=H, samples_per_pixel=probe_spp, max_depth=depth, seed=1), O.RNG_COUNTER, threads)
    rate = W * H * probe_spp / max(ps.seconds_render, 1e-6)
    spp = int(max(8, min(2048, args.cpu_baseline_seconds * rate / (W * H))))
    _, s = osc.render(cam, sd.settings(width=W, height=H, samples_per_pixel=spp, max_depth=depth, seed=2), O.RNG_COUNTER, threads)
    return {"value": s.samples / s.seconds_render / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port", **host,
            "sample": f"{name} {W}x{H} {spp}spp max_depth={depth} NEE+MIS (same scene and estimator, reduced resolution), {s.seconds_render:.1f} s",
            "mrays_per_s_reference_equivalent": (s.rays_closest + s.rays_shadow) / s.seconds_render / 1e6}


if __name__ == "__main__":
    main()
