#!/usr/bin/env python3
"""Device-resident timings of BASELINE.json's five configs on one GPU (the parity of the same configs is
tests/test_gpu_baseline_configs.py; the headline is bench.py = configs[3]).  Scenes from tests/golden (the
reference's .obj files are not on the GPU box), sponza-standin and the procedural probe as everywhere."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import scene_fixtures as sf
B = importlib.import_module("path-tracer_amd").binding
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
env = sf.procedural_env(1024)
CONFIGS = [  # name, scene, w, h, mode, spp per step, bounces, steps
    ("1 cornellbox 512x512 primary", "cornellbox", 512, 512, B.MODE_PRIMARY, 1, 0, 200),
    ("2 bunny 1024x1024 direct + env", "bunny", 1024, 1024, B.MODE_DIRECT, 1, 6, 50),
    ("3 teapot-glass 1080p path 64 spp 8 bounces", "teapot-glass", 1920, 1080, B.MODE_PATH, 64, 8, 2),
    ("4 sponza-standin 1080p path 16 spp 4 bounces", "sponza", 1920, 1080, B.MODE_PATH, 16, 4, 5),
    ("5 sponza-standin 4K progressive path, 16 of 1024 spp per step, 4 bounces", "sponza", 3840, 2160, B.MODE_PATH, 16, 4, 2),
    # stand-in sensitivity: the headline's Mrays/s must not be a property of one 65k-triangle atrium
    ("4b sponza-standin detail 2 (256,768 tris) 1080p path 16 spp 4 bounces", "sponza2", 1920, 1080, B.MODE_PATH, 16, 4, 5),
    ("4c bunny (69,451 tris, open scene) 1080p path 16 spp 4 bounces", "bunny", 1920, 1080, B.MODE_PATH, 16, 4, 5),
    ("4d testObj (11,950 tris, deep tree) 1080p path 16 spp 4 bounces", "testObj", 1920, 1080, B.MODE_PATH, 16, 4, 5),
]
scenes = {}
for label, name, w, h, mode, spp, bounces, steps in CONFIGS:
    if name not in scenes:
        arrays = sf.sponza_standin(2 if name == "sponza2" else 1) if name.startswith("sponza") else sf.load_scene_arrays(name)
        tc = time.perf_counter()  # scene creation: validation, wide hierarchy on the host's cores, packing (no cache file)
        sc = sf.product_scene_from_mesh_arrays(*arrays, env=env, lights=sf.SPONZA_LIGHT if name.startswith("sponza") else sf.DEFAULT_LIGHT)
        scenes[name] = (sc, B.Context(sc, device=0, stream=stream), time.perf_counter() - tc)
    sc, ctx, create_s = scenes[name]
    cam = B.make_camera(w, h, sf.SPONZA_EYE, sf.SPONZA_LOOKAT) if name.startswith("sponza") else B.default_camera(w, h)
    accum = torch.zeros(h * w * 3, dtype=torch.float32, device=dev)
    rgb = torch.zeros(h * w * 3, dtype=torch.float32, device=dev)
    sample = 0
    for k in range(steps + 1):
        if k == 1:
            torch.cuda.synchronize(); ctx.reset_stats(); t0 = time.perf_counter()
        ctx.render_device(cam, B.RenderParams(mode, spp, sample, bounces, 1, 0, 1, 64, 8), accum.data_ptr(), rgb.data_ptr())
        sample += spp
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    st = ctx.stats()
    rays = (st["rays_closest"] + st["rays_occluded"]) / steps
    # box / triangle tests per ray of one more step with the counting kernels (outside the timing)
    ctx.reset_stats(); ctx.set_counting(True)
    ctx.render_device(cam, B.RenderParams(mode, spp, 0, bounces, 1, 0, 1, 64, 8), accum.data_ptr(), rgb.data_ptr())
    torch.cuda.synchronize()
    c = ctx.stats(); ctx.set_counting(False)
    print(json.dumps({"config": label, "ms_per_step": round(dt * 1e3, 3), "Mrays_per_step": round(rays / 1e6, 2),
                      "Mrays_per_s": round(rays / dt / 1e6, 1), "ms_per_frame_1spp": round(dt * 1e3 / spp, 3),
                      "wide_nodes": sc.wide_nodes, "scene_create_s": round(create_s, 3),
                      "box_tests_per_closest_ray": round(c["node_visits_closest"] / max(c["rays_closest"], 1), 2),
                      "tri_tests_per_closest_ray": round(c["tri_tests_closest"] / max(c["rays_closest"], 1), 2),
                      "box_tests_per_shadow_ray": round(c["node_visits_occluded"] / max(c["rays_occluded"], 1), 2),
                      "tri_tests_per_shadow_ray": round(c["tri_tests_occluded"] / max(c["rays_occluded"], 1), 2)}), flush=True)
    del accum, rgb
