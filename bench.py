#!/usr/bin/env python3
"""bench.py -- Mrays/s of the ray/scene hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json metric: "Mrays/s ... Sponza 1080p path trace; ms/frame
@1spp", configs[3]): sponza-standin (procedural atrium, 65,536 triangles --
the real sponza.obj is not in the reference tree), procedural angular-map
environment (stand-in for uffizi_probe.hdr), 1920x1080, path traced, 4
bounces, 16 spp, light (0.5,4,1)/1000 W (main.cpp:24).  One step = one 16-spp
frame per GPU: with N GPUs a step takes 16*N samples of the frame, image tiles
are interleaved over the ranks (scene replicated per GPU), every rank renders
its tiles for those samples and the tiles are gathered to rank 0 over RCCL once
per step.  Per-GPU work per step is therefore fixed ("weak").  The 16 samples
of a pixel are traced together (33 M paths resident in HBM), nothing is cached
between samples: every sample regenerates and re-traces its primary rays as
the reference does.  ms/frame @1spp = ms_per_step / 16 (config.ms_per_frame_1spp).
A ray = one closest-hit or any-hit traversal (primary, extension and shadow
rays each count 1).

The scene arena, accumulation and frame buffers are resident in HBM before the
timed region.  torch is plumbing only: device buffers, the stream, RCCL.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

W, H, BOUNCES, SEED, SPP = 1920, 1080, 4, 1, 16
TILE_W, TILE_H = 64, 8
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes(rays, node_visits, tri_tests):
    """32 B ray in + 16 B hit out, 32 B per node visit, 36 B per triangle test"""
    return 48 * rays + 32 * node_visits + 36 * tri_tests


def cpu_baseline(arrays, env, lights, eye, lookat):
    """the oracle (CPU restatement, pinned against the reference's recorded outputs) timed on this host:
    1 thread, bit-trail traversal (what the reference's Scene instantiates, cs488.h:2008), same scene and
    settings, on a bounded sample of the frame's rows"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_binding as ob
    import scene_fixtures as sf
    orc = sf.oracle_scene_from_mesh_arrays(*arrays, env=env, lights=lights)
    cam = ob.make_camera(W, H, eye=eye, lookat=lookat)
    rows, rays, t0 = 0, 0, time.perf_counter()
    # every 24th row first (a representative cut of the frame), keep adding rows for ~12 s
    order = [r for phase in range(24) for r in range(phase, H, 24)]
    rgb = None
    for r in order:
        rgb = orc.pathtrace(cam, spp=1, max_bounces=BOUNCES, rng_mode=1, seed=SEED, variant=ob.BITTRAIL, rows=(r, r + 1))
        rays += orc.last_rays
        rows += 1
        if rows >= 45 and time.perf_counter() - t0 > 12.0:
            break
    dt = time.perf_counter() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
            "sample": f"{rows} of {H} rows (interleaved every 24th) of the same 1920x1080 1-spp 4-bounce frame, "
                      f"{rays} rays in {dt:.1f} s, bit-trail traversal, 1 thread"}


def _cpu_worker(job):
    """one process of the all-cores leg: rows start, start + stride, ... of the same frame for about `budget` seconds"""
    detail, start, stride, budget = job
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_binding as ob
    import scene_fixtures as sf
    orc = sf.oracle_scene_from_mesh_arrays(*sf.sponza_standin(detail), env=sf.procedural_env(1024), lights=sf.SPONZA_LIGHT)
    cam = ob.make_camera(W, H, eye=sf.SPONZA_EYE, lookat=sf.SPONZA_LOOKAT)
    rays, rows, t0 = 0, 0, time.perf_counter()
    for sample in range(64):  # its rows of sample 0 of the frame, then of sample 1, ... until the budget is spent
        for r in range(start, H, stride):
            orc.pathtrace(cam, spp=1, sample_offset=sample, max_bounces=BOUNCES, rng_mode=1, seed=SEED, variant=ob.BITTRAIL,
                          rows=(r, r + 1))
            rays += orc.last_rays
            rows += 1
            if time.perf_counter() - t0 > budget:
                return rays, rows, time.perf_counter() - t0
    return rays, rows, time.perf_counter() - t0


def cpu_baseline_all_cores(detail, budget=10.0):
    """the same oracle, scanline-parallel over the host cores this job may use (the counter RNG makes
    rows independent).  Child processes of this script (`--cpu-rows`), none of which touches the GPU; each is
    waited for with a time limit and killed by its own pid if it overstays"""
    import subprocess
    cores = max(1, min(16, os.cpu_count() or 1))  # the GPU box gives a one-GPU job 16 CPUs
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-rows", str(detail), str(k), str(cores), str(budget)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for k in range(cores)]
    res = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=budget + 90.0)
            res.append(json.loads(out.strip().splitlines()[-1]))
        except Exception:
            pr.kill()
            pr.wait()
    if not res:
        raise RuntimeError("no CPU worker finished")
    rays = sum(r[0] for r in res)
    dt = max(r[2] for r in res)
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": len(res),
            "sample": f"{sum(r[1] for r in res)} rows interleaved over {len(res)} processes, {rays} rays in {dt:.1f} s"}


def main():
    if len(sys.argv) == 6 and sys.argv[1] == "--cpu-rows":  # a worker of cpu_baseline_all_cores: no torch, no GPU
        print(json.dumps(_cpu_worker((int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])))), flush=True)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--with-primary", action="store_true",
                    help="also time configs[0] (cornellbox 512x512 primary rays only); off by default so that the kernel "
                         "averages of a rocprofv3 run of this command are those of the headline workload")
    ap.add_argument("--spp", type=int, default=SPP, help="samples per GPU per step (default 16 = BASELINE configs[3]; other "
                                                         "values are for studying launch-size effects, not the headline)")
    ap.add_argument("--bvh", choices=("reference", "all-axes"), default="reference",
                    help="all-axes: the three-axis SAH build (SURVEY 8(f).1), NOT the reference's tree and not the headline")
    ap.add_argument("--detail", type=int, default=1, help="sponza-standin tessellation (1: 65,536 tris, 2: 256,768)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import scene_fixtures as sf
    pkg = importlib.import_module("path-tracer_amd")
    B = pkg.binding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    # ---- scene: host-side load/build, then one arena upload ----
    arrays = sf.sponza_standin(args.detail)
    env = sf.procedural_env(1024)
    scene = sf.product_scene_from_mesh_arrays(*arrays, env=env, lights=sf.SPONZA_LIGHT, all_axes=args.bvh == "all-axes")
    stream = torch.cuda.current_stream(dev)
    ctx = B.Context(scene, device=local_rank, stream=stream.cuda_stream)
    cam = B.make_camera(W, H, sf.SPONZA_EYE, sf.SPONZA_LOOKAT)
    accum = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
    rgb = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)

    gatherer = pkg.tiles.TileGather(W, H, rank, world, dev, TILE_W, TILE_H)  # owned pixels -> rank 0, one RCCL gather per step

    spp_per_step = args.spp * world  # one 16-spp frame-equivalent of paths per GPU per step
    state = {"sample": 0}

    def step():
        p = B.RenderParams(B.MODE_PATH, spp_per_step, state["sample"], BOUNCES, SEED, rank, world, TILE_W, TILE_H)
        ctx.render_device(cam, p, accum.data_ptr(), rgb.data_ptr())
        state["sample"] += spp_per_step
        gatherer.gather(rgb)  # no-op on one GPU

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    ctx.reset_stats()
    ctx.set_profiling(True)  # HIP events around every kernel launch, on the stream the kernels run on
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    st = ctx.stats()
    ctx.set_profiling(False)

    # per-kernel visit counts of exactly the same frames (deterministic), outside the timed region
    accum2 = torch.zeros_like(accum)
    rgb2 = torch.zeros_like(rgb)
    ctx.reset_stats()
    ctx.set_counting(True)
    first = state["sample"] - args.steps * spp_per_step
    for k in range(args.steps):
        p = B.RenderParams(B.MODE_PATH, spp_per_step, first + k * spp_per_step, BOUNCES, SEED, rank, world, TILE_W, TILE_H)
        ctx.render_device(cam, p, accum2.data_ptr(), rgb2.data_ptr())
    ctx.synchronize()
    cnt = ctx.stats()
    ctx.set_counting(False)
    assert cnt["rays_closest"] == st["rays_closest"] and cnt["rays_occluded"] == st["rays_occluded"]

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    rays = torch.tensor([float(st["rays_closest"] + st["rays_occluded"])], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed_max, total_rays = float(t.item()), float(rays.item())

    if rank == 0:
        def kernel_line(rays_k, visits, tests, ms, launches):
            byts = algorithmic_bytes(rays_k, visits, tests)
            return {"launches": int(launches), "avg_launch_ms": ms / max(launches, 1),
                    "algorithmic_bytes_per_launch": byts / max(launches, 1), "rays": int(rays_k),
                    "node_visits_per_ray": visits / max(rays_k, 1), "tri_tests_per_ray": tests / max(rays_k, 1),
                    "achieved_GBs": byts / max(ms, 1e-9) / 1e6, "algorithmic_bytes_per_ray": byts / max(rays_k, 1)}
        kc = kernel_line(st["rays_closest"], cnt["node_visits_closest"], cnt["tri_tests_closest"], st["ms_closest"],
                         st["launches_closest"])
        ko = kernel_line(st["rays_occluded"], cnt["node_visits_occluded"], cnt["tri_tests_occluded"], st["ms_occluded"],
                         st["launches_occluded"])
        # the kernel behind the two traversal entries: the order-free 4-wide traversal when the scene qualifies
        # (each launch is followed by a literal-walk launch over the handful of rays it handed back), else the walk
        tv = os.environ.get("MIPT_TRAVERSAL")  # A/B runs: walk = literal threaded walk, bittrail = StacklessBVH2 literal
        wide = scene.wide_nodes > 0 and tv not in ("walk", "bittrail")
        names = (("k_trace_wide_closest", "k_trace_wide_occluded") if wide else
                 ("k_trace_bittrail_closest", "k_trace_bittrail_occluded") if tv == "bittrail" else
                 ("k_trace_closest", "k_trace_occluded"))
        dom, dom_name = (kc, names[0]) if st["ms_closest"] >= st["ms_occluded"] else (ko, names[1])
        traffic, valu_frac, l2_hit = None, None, None
        tf = os.path.join(ROOT, "profiles", "traffic.json")  # written from rocprofv3 --pmc passes (see profiles/README.md)
        if os.path.exists(tf):
            try:
                rec = json.load(open(tf)).get(dom_name, {})
                traffic, valu_frac, l2_hit = rec.get("hbm_bytes_per_launch"), rec.get("valu_active_frac"), rec.get("l2_hit_rate")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s, sponza-standin 1920x1080 path trace, 16 spp per frame (ms/frame @1spp = ms_per_step/16)",
            "value": total_rays / elapsed_max / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"sponza-standin ({len(arrays[3])} tris, procedural; sponza.obj absent) 1920x1080 "
                                   f"path trace, {BOUNCES} bounces, {args.spp} spp per GPU per step, procedural env "
                                   f"(uffizi_probe.hdr absent), tiles {TILE_W}x{TILE_H} interleaved over {world} GPU(s), "
                                   f"RCCL gather to rank 0" if world > 1 else
                                   f"sponza-standin ({len(arrays[3])} tris, procedural; sponza.obj absent) 1920x1080 "
                                   f"path trace, {BOUNCES} bounces, {args.spp} spp per step, procedural env (uffizi_probe.hdr absent)",
                       "bvh": args.bvh, "resolution": [W, H], "bounces": BOUNCES, "spp_per_step": spp_per_step, "seed": SEED,
                       "rays_per_step": total_rays / args.steps,
                       "ms_per_frame_1spp": elapsed_max / args.steps * 1e3 / spp_per_step * world},
            "roofline": {"bound": "hbm", "achieved": dom["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": dom["achieved_GBs"] / HBM_PEAK_GBS, "traffic": traffic, "kernel": dom_name,
                         "per_kernel": {"k_trace_closest": kc, "k_trace_occluded": ko, "kernel_names": list(names),
                                        "other_ms_total": st["ms_other"], "other_launches": int(st["launches_other"])},
                         "valu_active_frac": valu_frac, "l2_hit_rate": l2_hit,
                         "note": "achieved = algorithmic bytes (48 B/ray + 32 B/node visit + 36 B/triangle test) per "
                                 "launch / average launch duration from HIP events in the timed region (rank 0); node "
                                 "visits = box tests (4 per visit of a 4-wide node + the exact test of each candidate "
                                 "leaf). The traversal re-reads a few MB of nodes from LDS/L1/L2, so algorithmic bytes "
                                 "exceed HBM traffic (`traffic`, PMC) about tenfold and frac can exceed 1: the kernel is "
                                 "bound by vector-ALU issue (valu_active_frac, PMC), not by HBM"},
        }
        if world == 1 and args.with_primary:
            # primary-only rate of configs[0] (cornellbox 512x512, closest-hit ids), reported beside the
            # headline; 262,144 rays per frame cannot fill the chip, this is a latency figure
            try:
                cb = sf.product_scene_from_mesh_arrays(*sf.load_scene_arrays("cornellbox"))
                cctx = B.Context(cb, device=local_rank, stream=stream.cuda_stream)
                ccam = B.default_camera(512, 512)
                cbuf = torch.zeros(512 * 512 * 3, dtype=torch.float32, device=dev)
                pp = B.RenderParams(B.MODE_PRIMARY, 1, 0, 0, SEED, 0, 1, TILE_W, TILE_H)
                for _ in range(3):
                    cctx.render_device(cam=ccam, params=pp, accum_ptr=0, rgb_ptr=cbuf.data_ptr())
                torch.cuda.synchronize(dev)
                tp = time.perf_counter()
                for _ in range(50):
                    cctx.render_device(cam=ccam, params=pp, accum_ptr=0, rgb_ptr=cbuf.data_ptr())
                torch.cuda.synchronize(dev)
                out["config"]["primary_only_cornellbox_512_mrays_s"] = 50 * 512 * 512 / (time.perf_counter() - tp) / 1e6
            except Exception as e:  # informational only
                out["config"]["primary_only_cornellbox_512_mrays_s"] = None
                out["config"]["primary_only_error"] = str(e)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(arrays, env, sf.SPONZA_LIGHT, sf.SPONZA_EYE, sf.SPONZA_LOOKAT)
            try:  # the same oracle on all the cores of this job, reported beside the 1-thread figure
                out["cpu_baseline"]["all_cores"] = cpu_baseline_all_cores(args.detail)
            except Exception as e:
                out["cpu_baseline"]["all_cores"] = {"error": str(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
