#!/usr/bin/env python3
"""bench.py -- Mrays/s of the ray/scene hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 either way: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or typed as above, in which case this process spawns the N
ranks itself (before it touches torch or the GPU), waits for them and relays rank 0's JSON line.

Workload (BASELINE.json metric: "Mrays/s ... Sponza 1080p path trace; ms/frame @1spp", configs[3]): sponza-standin
(procedural atrium, 65,536 triangles -- the real sponza.obj is not in the reference tree), procedural angular-map
environment (stand-in for uffizi_probe.hdr), 1920x1080, path traced, 4 bounces, 16 spp, light (0.5,4,1)/1000 W
(main.cpp:24).  One step = ONE 16-spp frame: with N GPUs the frame's tiles are interleaved over the ranks (scene
replicated per GPU), every rank renders its tiles for the 16 samples and the tiles are gathered to rank 0 over RCCL
once per step -- the frame is divided, so this is STRONG scaling, as configs[3] is written ("tile-split over 2/4/8").
For N > 1 the weak-scaling figure (16*N samples per step: per-GPU work fixed) is measured in a second timed loop and
reported beside it (config.weak_scaling).  The 16 samples of a pixel are traced together (33 M paths resident in HBM
on one GPU), nothing is cached between samples: every sample regenerates and re-traces its primary rays as the
reference does.  ms/frame @1spp = ms_per_step / 16 (config.ms_per_frame_1spp).  A ray = one closest-hit or any-hit
traversal (primary, extension and shadow rays each count 1).

The scene arena, accumulation and frame buffers are resident in HBM before the timed region.  torch is plumbing only:
device buffers, the stream, RCCL.
"""
import argparse
import ctypes as C
import glob
import hashlib
import importlib
import json
import os
import socket
import subprocess
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
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector peak (256 CUs x 128 FMA lanes x 2 flop x 2.4 GHz)


def algorithmic_bytes(rays, node_visits, tri_tests):
    """32 B ray in + 16 B hit out, 32 B per node visit, 36 B per triangle test"""
    return 48 * rays + 32 * node_visits + 36 * tri_tests


def algorithmic_flops(node_visits, tri_tests):
    """27 flop per box test (6 sub, 6 div, 15 cmp/sel), 95 per triangle test"""
    return 27 * node_visits + 95 * tri_tests


def kernel_source_hash():
    """identity of the kernel sources a PMC summary belongs to: sha256 over the device code, the API that launches it
    and the build flags.  tools/pmc_summary.py stores it in profiles/traffic.json; counters recorded for other sources
    are not reported."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "path-tracer_amd", "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "device", "*.h*")) + glob.glob(os.path.join(csrc, "pt", "*.hpp")) +
                   [os.path.join(csrc, "api", "api_device.hip"), os.path.join(csrc, "Makefile")])
    for f in files:
        h.update(os.path.relpath(f, csrc).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """`python bench.py --gpus N` typed without a launcher: start the N ranks as children of this process -- which has
    not imported torch or touched the GPU -- with the environment torch.distributed.run would give them, relay rank 0's
    output, exit with the worst exit code.  Children are killed by their own pids if one of them fails or overstays."""
    import tempfile
    port = _free_port()
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    rc = 0
    deadline = time.time() + 1500.0
    try:
        while True:
            codes = [pr.poll() for pr in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:  # a rank that died leaves the others waiting in a rendezvous or a collective: end them
                rc = bad[0]
                print(f"bench.py: rank {codes.index(bad[0])} exited with status {bad[0]}; stopping the other ranks", file=sys.stderr)
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                rc = 124
                print("bench.py: ranks still running after 1500 s; stopping them", file=sys.stderr)
                break
            time.sleep(0.1)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
                pr.wait()
    out0.seek(0)
    sys.stdout.write(out0.read())
    sys.stdout.flush()
    return rc


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
        return 0
    if len(sys.argv) >= 3 and sys.argv[1] == "--native-node-child":  # a process of its own: the native one-process node
        return native_node_child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--native-node", action="store_true",
                    help="after the headline measurement, rank 0 also times the NATIVE node (mipt_node_render: one process, a host "
                         "thread + stream + context per GPU, RCCL gather of the owned tiles) on the same N GPUs and the same frame, "
                         "in a child process under a time limit; reported as config.native_node (on by default for N > 1)")
    ap.add_argument("--no-native-node", action="store_true")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--with-primary", action="store_true",
                    help="also time configs[0] (cornellbox 512x512 primary rays only); off by default so that the kernel "
                         "averages of a rocprofv3 run of this command are those of the headline workload")
    ap.add_argument("--spp", type=int, default=SPP, help="samples per step (default 16 = BASELINE configs[3]; other "
                                                         "values are for studying launch-size effects, not the headline)")
    ap.add_argument("--bvh", choices=("reference", "all-axes"), default="reference",
                    help="all-axes: the three-axis SAH build (SURVEY 8(f).1), NOT the reference's tree and not the headline")
    ap.add_argument("--detail", type=int, default=1, help="sponza-standin tessellation (1: 65,536 tris, 2: 256,768)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="serialise shadow-ray and extension-ray launches on one stream in the timed loop too (for rocprofv3 "
                         "--kernel-trace runs: overlapped kernels' begin-to-end times include waiting for CUs)")
    ap.add_argument("--sync-gather", action="store_true",
                    help="N > 1: gather every step's tiles on the current stream (the next step waits for it) instead of on a side stream")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group and run the per-step gather, barriers and reductions even with one "
                         "rank: the N > 1 code path on a one-GPU box (tests/test_gpu_node.py)")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the N-rank plumbing (launcher, rendezvous over gloo, tile gather, barrier, "
                         "max-over-ranks timing, rank 0's JSON line) with the render replaced by filling the owned pixels: "
                         "no GPU, no measurement (tests/test_bench_launcher.py)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus)  # nothing below has run in this process: no torch import, no GPU call

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    import scene_fixtures as sf
    pkg = importlib.import_module("path-tracer_amd")

    if args.dry_run:
        return dry_run(args, rank, world, torch, dist, pkg)

    B = pkg.binding
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist  # the collectives run
    # stdout carries ONE line, the JSON: libraries that print there (RCCL's version banner when the first communicator
    # is made) are sent to stderr for the rest of the run; the line itself goes to the kept descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if multi:
        if "MASTER_ADDR" not in os.environ:  # --force-dist typed without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=dev)

    # ---- scene: host-side load/build, then one arena upload ----
    arrays = sf.sponza_standin(args.detail)
    env = sf.procedural_env(1024)
    # N ranks build ONE hierarchy: rank 0 builds it (on the host's cores) into a keyed file, the others load that file
    # (mipt_scene_create_cached; the key covers the tree, the leaf order, the positions and the light)
    wide_cache = None
    if world > 1:
        import tempfile
        wide_cache = os.path.join(tempfile.gettempdir(), f"mipt_bench_{os.environ.get('MASTER_PORT', '0')}_{args.detail}_{args.bvh}.wide")
    def make_scene():
        return sf.product_scene_from_mesh_arrays(*arrays, env=env, lights=sf.SPONZA_LIGHT, all_axes=args.bvh == "all-axes", wide_cache=wide_cache)
    t_scene = time.perf_counter()
    if world > 1:
        scene = make_scene() if rank == 0 else None
        dist.barrier()
        if rank != 0:
            scene = make_scene()
    else:
        scene = make_scene()
    t_scene = time.perf_counter() - t_scene
    stream = torch.cuda.current_stream(dev)
    ctx = B.Context(scene, device=local_rank, stream=stream.cuda_stream)
    cam = B.make_camera(W, H, sf.SPONZA_EYE, sf.SPONZA_LOOKAT)
    accum = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
    rgb = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)

    gatherer = pkg.tiles.TileGather(W, H, rank, world, dev, TILE_W, TILE_H, always=args.force_dist)  # owned pixels -> rank 0, one RCCL gather per step

    def sync():
        gatherer.finish()
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed_loop(spp_per_step, profile=False):
        """W warm-up steps, then exactly K timed steps of spp_per_step samples of the frame (this rank's tiles), each
        followed by the gather; -> (seconds, stats of the timed steps, index of the first timed sample).
        profile: HIP events around every kernel launch, kernels serialised on one stream (the per-kernel pass)"""
        state = {"sample": 0}

        def step():
            p = B.RenderParams(B.MODE_PATH, spp_per_step, state["sample"], BOUNCES, SEED, rank, world, TILE_W, TILE_H)
            ctx.render_device(cam, p, accum.data_ptr(), rgb.data_ptr())
            state["sample"] += spp_per_step
            # no-op on one GPU; else the owned tiles travel to rank 0 on a side stream beside the next step's kernels
            # (--sync-gather: on the current stream, every step waiting for its gather as in round 2)
            (gatherer.gather if args.sync_gather else gatherer.gather_async)(rgb)

        accum.zero_()
        ctx.set_overlap(not profile and not args.no_overlap)
        for _ in range(args.warmup):
            step()
        sync()
        ctx.reset_stats()
        ctx.set_profiling(profile)  # on the stream the kernels run on
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        elapsed = time.perf_counter() - t0
        st = ctx.stats()
        ctx.set_profiling(False)
        ctx.set_overlap(not args.no_overlap)
        return elapsed, st, state["sample"] - args.steps * spp_per_step

    def reduce_over_ranks(elapsed, st):
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        rays = torch.tensor([float(st["rays_closest"] + st["rays_occluded"])], dtype=torch.float64, device=dev)
        if multi:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        return float(t.item()), float(rays.item())

    # headline: one args.spp-sample frame per step, divided over the ranks (strong scaling; N = 1: the whole frame)
    spp_per_step = args.spp
    elapsed, st0, first = timed_loop(spp_per_step)
    elapsed_max, total_rays = reduce_over_ranks(elapsed, st0)
    # per-kernel durations for the roofline: the same K steps once more (same frames: the sample index restarts), with
    # HIP events around every launch and the shadow-ray launches serialised behind the closest-hit ones -- in the timed
    # loop above they overlap on two streams, where a launch's begin-to-end time includes waiting for CUs
    serial_elapsed, st, _ = timed_loop(spp_per_step, profile=True)
    assert st["rays_closest"] == st0["rays_closest"] and st["rays_occluded"] == st0["rays_occluded"]

    # per-kernel visit counts of exactly the same frames (deterministic), outside the timed region
    accum2 = torch.zeros_like(accum)
    rgb2 = torch.zeros_like(rgb)
    ctx.reset_stats()
    ctx.set_counting(True)
    for k in range(args.steps):
        p = B.RenderParams(B.MODE_PATH, spp_per_step, first + k * spp_per_step, BOUNCES, SEED, rank, world, TILE_W, TILE_H)
        ctx.render_device(cam, p, accum2.data_ptr(), rgb2.data_ptr())
    ctx.synchronize()
    cnt = ctx.stats()
    assert cnt["rays_closest"] == st["rays_closest"] and cnt["rays_occluded"] == st["rays_occluded"]
    # the same frames once more through the LITERAL threaded walk (k_trace, the reference's own tree and order): its node
    # visits and triangle tests are the reference's per-ray work, a numerator that does not move when the product's own
    # hierarchy does
    ctx.reset_stats()
    ctx.set_traversal(B.TRAVERSAL_THREADED)
    for k in range(args.steps):
        p = B.RenderParams(B.MODE_PATH, spp_per_step, first + k * spp_per_step, BOUNCES, SEED, rank, world, TILE_W, TILE_H)
        ctx.render_device(cam, p, accum2.data_ptr(), rgb2.data_ptr())
    ctx.synchronize()
    ref_cnt = ctx.stats()
    ctx.set_traversal(B.TRAVERSAL_DEFAULT)
    ctx.set_counting(False)
    assert ref_cnt["rays_closest"] == st["rays_closest"] and ref_cnt["rays_occluded"] == st["rays_occluded"]
    del accum2, rgb2

    weak = None
    if multi:  # per-GPU work fixed: N frames' worth of samples per step
        w_elapsed, w_st, _ = timed_loop(args.spp * world)
        w_max, w_rays = reduce_over_ranks(w_elapsed, w_st)
        weak = {"value": w_rays / w_max / 1e6, "unit": "Mrays/s", "ms_per_step": w_max / args.steps * 1e3,
                "spp_per_step": args.spp * world, "scaling": "weak"}

    native = None
    if (args.native_node or world > 1) and not args.no_native_node:
        # the other ranks keep their contexts but are idle (they wait at the barrier below) while rank 0's child runs
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        if rank == 0:
            native = time_native_node(world, args.steps, args.warmup, args.detail)
        if multi:
            dist.barrier()
    if rank == 0:
        def kernel_line(rays_k, visits, tests, ms, launches):
            byts = algorithmic_bytes(rays_k, visits, tests)
            flops = algorithmic_flops(visits, tests)
            return {"launches": int(launches), "avg_launch_ms": ms / max(launches, 1),
                    "algorithmic_bytes_per_launch": byts / max(launches, 1), "rays": int(rays_k),
                    "node_visits_per_ray": visits / max(rays_k, 1), "tri_tests_per_ray": tests / max(rays_k, 1),
                    "achieved_GBs": byts / max(ms, 1e-9) / 1e6, "algorithmic_bytes_per_ray": byts / max(rays_k, 1),
                    "algorithmic_flops_per_ray": flops / max(rays_k, 1), "achieved_TFLOPs": flops / max(ms, 1e-9) / 1e9,
                    "compulsory_bytes_per_launch": 52.0 * rays_k / max(launches, 1)}
        kc = kernel_line(st["rays_closest"], cnt["node_visits_closest"], cnt["tri_tests_closest"], st["ms_closest"],
                         st["launches_closest"])
        ko = kernel_line(st["rays_occluded"], cnt["node_visits_occluded"], cnt["tri_tests_occluded"], st["ms_occluded"],
                         st["launches_occluded"])
        kc_ref = kernel_line(st["rays_closest"], ref_cnt["node_visits_closest"], ref_cnt["tri_tests_closest"], st["ms_closest"],
                             st["launches_closest"])
        ko_ref = kernel_line(st["rays_occluded"], ref_cnt["node_visits_occluded"], ref_cnt["tri_tests_occluded"], st["ms_occluded"],
                             st["launches_occluded"])
        # the kernel behind the two traversal entries: the order-free 4-wide traversal when the scene qualifies
        # (each launch is followed by a literal-walk launch over the handful of rays it handed back), else the walk
        tv = os.environ.get("MIPT_TRAVERSAL")  # A/B runs: walk = literal threaded walk, bittrail = StacklessBVH2 literal
        wide = scene.wide_nodes > 0 and tv not in ("walk", "bittrail")
        names = (("k_trace_wide_closest", "k_trace_wide_occluded") if wide else
                 ("k_trace_bittrail_closest", "k_trace_bittrail_occluded") if tv == "bittrail" else
                 ("k_trace_closest", "k_trace_occluded"))
        closest_dominates = st["ms_closest"] >= st["ms_occluded"]
        dom, dom_name = (kc, names[0]) if closest_dominates else (ko, names[1])
        dom_ref = kc_ref if closest_dominates else ko_ref
        # PMC counters of the dominant kernel (separate rocprofv3 --pmc passes, tools/pmc_passes.sh): reported only
        # when they were recorded for THESE kernel sources and this launch shape
        pmc, pmc_note = {}, None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                allrec = json.load(open(tf))
                if allrec.get("kernel_source_hash") != kernel_source_hash():
                    pmc_note = "profiles/traffic.json was recorded for other kernel sources: counters withheld"
                elif world != 1 or args.spp != SPP or args.detail != 1 or args.bvh != "reference":
                    pmc_note = "profiles/traffic.json was recorded for the default one-GPU workload: counters withheld"
                else:
                    pmc = allrec.get(dom_name, {})
            except Exception as e:
                pmc_note = f"profiles/traffic.json unreadable: {e}"
        traffic = pmc.get("hbm_bytes_per_launch")
        # the path-mode shading kernel: HBM-bound; bytes per launch from the same PMC passes, duration from this run's events
        shade_pmc = allrec.get("k_shade_path", {}) if pmc else {}
        shade_launches = int(st["launches_shade"])
        shade_ms = st["ms_shade"] / max(shade_launches, 1)
        shade_bytes = shade_pmc.get("hbm_bytes_per_launch")
        k_shade = {"launches": shade_launches, "avg_launch_ms": shade_ms, "bound": "hbm",
                   "hbm_bytes_per_launch": shade_bytes,
                   "achieved_TBs": (shade_bytes / (shade_ms * 1e-3) / 1e12) if shade_bytes and shade_ms > 0 else None,
                   "frac_of_hbm_peak": (shade_bytes / (shade_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if shade_bytes and shade_ms > 0 else None,
                   "bytes_per_path_bounce": (shade_bytes * shade_launches / max(st["rays_closest"], 1)) if shade_bytes else None,
                   "share_of_step": st["ms_shade"] / max(serial_elapsed * 1e3, 1e-9)}
        ms_frame = elapsed_max / args.steps * 1e3 / spp_per_step
        part = f"tiles {TILE_W}x{TILE_H} interleaved over {world} GPUs, one RCCL gather to rank 0 per step" if world > 1 else "one GPU"
        out = {
            "metric": "Mrays/s, sponza-standin 1920x1080 path trace, 16 spp per frame (ms/frame @1spp = ms_per_step/16)",
            "value": total_rays / elapsed_max / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"sponza-standin ({len(arrays[3])} tris, procedural; sponza.obj absent) 1920x1080 path "
                                   f"trace, {BOUNCES} bounces, ONE {args.spp}-spp frame per step, procedural env "
                                   f"(uffizi_probe.hdr absent), {part}",
                       "bvh": args.bvh, "resolution": [W, H], "bounces": BOUNCES, "spp_per_step": spp_per_step, "seed": SEED,
                       "rays_per_step": total_rays / args.steps, "ms_per_frame_1spp": ms_frame,
                       "scene_create_s": t_scene,
                       "ms_per_step_serialised_with_events": serial_elapsed / args.steps * 1e3,
                       "weak_scaling": weak, "native_node": native,
                       "gather": None if not multi else ("current stream, every step waits for it" if args.sync_gather else
                                                         "side stream beside the next step's kernels (TileGather.gather_async)")},
            "roofline": {"bound": "valu", "achieved": dom_ref["achieved_TFLOPs"], "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": dom_ref["achieved_TFLOPs"] / FP32_VALU_PEAK_TFLOPS, "traffic": traffic, "kernel": dom_name,
                         "flops_per_ray_reference_walk": dom_ref["algorithmic_flops_per_ray"],
                         "achieved_own_tree": dom["achieved_TFLOPs"], "frac_own_tree": dom["achieved_TFLOPs"] / FP32_VALU_PEAK_TFLOPS,
                         "flops_per_ray_own_tree": dom["algorithmic_flops_per_ray"],
                         "algorithmic_GBs": dom["achieved_GBs"],
                         "traffic_over_compulsory": (traffic / dom["compulsory_bytes_per_launch"]) if traffic else None,
                         "valu_issue_occupancy": pmc.get("valu_issue_occupancy"),
                         "useful_lane_frac": pmc.get("useful_lane_frac"),
                         "lanes_per_valu_inst": pmc.get("lanes_per_valu_inst"), "l2_hit_rate": pmc.get("l2_hit_rate"),
                         "pmc_note": pmc_note,
                         "per_kernel": {"k_trace_closest": kc, "k_trace_occluded": ko, "kernel_names": list(names),
                                        "k_shade_path": k_shade,
                                        "reference_walk_counts": {"closest": {"node_visits_per_ray": kc_ref["node_visits_per_ray"],
                                                                              "tri_tests_per_ray": kc_ref["tri_tests_per_ray"]},
                                                                  "occluded": {"node_visits_per_ray": ko_ref["node_visits_per_ray"],
                                                                               "tri_tests_per_ray": ko_ref["tri_tests_per_ray"]}},
                                        "other_ms_total": st["ms_other"], "other_launches": int(st["launches_other"])},
                         "note": "per-kernel durations come from a second pass over the same K steps (this rank) with HIP events "
                                 "around every launch, one launch per traversal and all on one stream (config."
                                 "ms_per_step_serialised_with_events); the timed loop traces a bounce's extension and shadow rays "
                                 "in ONE launch (k_trace_wide_both). achieved / frac = the REFERENCE's work per ray -- node visits "
                                 "and triangle tests of the literal threaded walk (StacklessBVH::intersect, counted by the COUNT "
                                 "instantiation of k_trace on the same frames, outside the timed region) priced as SURVEY 8(d) "
                                 "prices it, 27 flop per box test + 95 per triangle test -- x rays per launch / the average "
                                 "duration of the launch that actually answers those rays (k_trace_wide), against the fp32 "
                                 "vector peak: a figure only time can move. achieved_own_tree / frac_own_tree = the same with "
                                 "the product's own counts (eight slot tests per visit of an eight-slot node + the exact test "
                                 "of each candidate leaf): it rises when the product tests MORE boxes. The traversal is bound by "
                                 "vector-ALU issue and by the latency of its own dependent loads, not by HBM; -ffp-contract=off "
                                 "(bit-exactness) forbids FMA in the triangle test. algorithmic_GBs = SURVEY 8(d)'s byte formula "
                                 "(48 B/ray + 32 B/box test + 36 B/triangle test, own-tree counts) / the same duration: those "
                                 "bytes are served by LDS / L1 / L2, not HBM, and may exceed the HBM peak. traffic = HBM-side "
                                 "bytes per launch from the PMC passes (2 x FETCH_SIZE + WRITE_SIZE), "
                                 "traffic_over_compulsory = traffic / (52 B x rays). valu_issue_occupancy = VALU "
                                 "instructions x measured issue cost per instruction class / (SIMDs x kernel cycles); "
                                 "useful_lane_frac = that x lanes_per_valu_inst / 64 (profiles/README.md). per_kernel."
                                 "k_shade_path: the HBM-bound shading kernel, PMC bytes per launch / its average launch "
                                 "duration from this run's events against the 8 TB/s HBM3E peak"},
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
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def native_node_child(ngpus, steps, warmup, detail):
    """the same frame through mipt_node_* (api/api_node.hip): one process, ngpus GPUs; prints one JSON line"""
    import scene_fixtures as sf
    B = importlib.import_module("path-tracer_amd").binding
    scene = sf.product_scene_from_mesh_arrays(*sf.sponza_standin(detail), env=sf.procedural_env(1024), lights=sf.SPONZA_LIGHT)
    node = B.Node(scene, ngpus=ngpus)
    cam = B.make_camera(W, H, sf.SPONZA_EYE, sf.SPONZA_LOOKAT)
    for _ in range(warmup):
        node.render(cam, mode=B.MODE_PATH, spp=SPP, max_bounces=BOUNCES, seed=SEED, tile_w=TILE_W, tile_h=TILE_H, readback=False)
    node.synchronize()
    r0 = node.stats()
    t0 = time.perf_counter()
    for _ in range(steps):  # one fresh 16-spp frame per step, gathered on GPU 0, nothing copied to the host
        node.render(cam, mode=B.MODE_PATH, spp=SPP, max_bounces=BOUNCES, seed=SEED, tile_w=TILE_W, tile_h=TILE_H, readback=False)
    node.synchronize()
    dt = time.perf_counter() - t0
    r1 = node.stats()
    rays = (r1["rays_closest"] + r1["rays_occluded"]) - (r0["rays_closest"] + r0["rays_occluded"])
    print(json.dumps({"value": rays / dt / 1e6, "unit": "Mrays/s", "ms_per_step": dt / steps * 1e3, "n_gpus": ngpus, "steps": steps,
                      "uses_rccl": node.uses_rccl, "scaling": "strong",
                      "what": "mipt_node_render: one process, a host thread + stream + context per GPU, owned tiles gathered to GPU 0 "
                              "over RCCL (grouped send / receive of exactly the owned bytes), frame left on GPU 0"}), flush=True)
    return 0


def time_native_node(ngpus, steps, warmup, detail, limit=120.0):
    """runs native_node_child in a process of its own (a fault or a hang there must not take the headline with it)"""
    try:
        pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--native-node-child", str(ngpus), str(steps), str(warmup), str(detail)],
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=limit)
        if pr.returncode != 0:
            return {"error": f"child exited with status {pr.returncode}: {pr.stderr.strip().splitlines()[-1] if pr.stderr.strip() else ''}"}
        return json.loads(pr.stdout.strip().splitlines()[-1])
    except subprocess.TimeoutExpired:
        return {"error": f"child still running after {limit:.0f} s; killed"}
    except Exception as e:
        return {"error": str(e)}


def dry_run(args, rank, world, torch, dist, pkg):
    """the N-rank plumbing on CPU (gloo): every rank fills the pixels it owns as its GPU would, the tiles are gathered to
    rank 0 once per step, steps are bracketed by barriers, the time is the max over ranks, rank 0 checks the assembled
    frame and prints the JSON line (value 0: nothing is measured here)"""
    w, h = 320, 200
    if os.environ.get("MIPT_BENCH_TEST_DIE_RANK") == str(rank):  # tests: a rank that dies before the rendezvous
        print(f"rank {rank}: dying on request", file=sys.stderr)
        return 3
    if world > 1:
        dist.init_process_group("gloo")
    g = pkg.tiles.TileGather(w, h, rank, world, torch.device("cpu"), TILE_W, TILE_H)
    want = torch.arange(w * h * 3, dtype=torch.float32) * 0.25 + 3.0
    ok = True
    t0 = None
    for k in range(args.warmup + args.steps):
        if k == args.warmup:
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
        rgb = torch.zeros(w * h * 3)
        rgb.view(-1, 3)[g.idx[rank]] = (want * (k + 1)).view(-1, 3)[g.idx[rank]]
        g.gather(rgb)
        if rank == 0:
            ok = ok and bool(torch.equal(rgb, want * (k + 1)))
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - (t0 or time.perf_counter())], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "dry run of the N-rank plumbing (no GPU, nothing measured)", "value": 0.0, "unit": "Mrays/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(t.item()) / max(args.steps, 1) * 1e3,
                          "dry_run": True, "frames_assembled_correctly": ok}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
