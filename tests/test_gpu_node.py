"""mipt_node_* / mipt_render_node (include/mipt.h): one frame over the GPUs of a node, through the C-ABI.
On the one-GPU test box: a 1-GPU node exercises the real RCCL path (communicator, ncclGather, pack / unpack kernels);
nodes whose ranks share GPU 0 exercise N > 1 (thread per rank, tile ownership, the gather's layout, unpack on rank 0)
with device-to-device copies in place of RCCL.  Bar: the node's frame equals the one-context frame bit for bit (the
per-pixel counter RNG is keyed on the global pixel index)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def setup(B, sf):
    arrays = sf.load_scene_arrays("cornellbox-glass")
    scene = sf.product_scene_from_mesh_arrays(*arrays, env=sf.procedural_env(64))
    ctx = B.Context(scene, device=0)
    cam = B.default_camera(200, 150)  # neither a multiple of the 64 x 8 tile: edge tiles are clipped
    ref_path = ctx.render(cam, mode=B.MODE_PATH, spp=5, max_bounces=4, seed=3)
    ref_direct = ctx.render(cam, mode=B.MODE_DIRECT)
    return scene, ctx, cam, ref_path, ref_direct


def test_one_gpu_node_uses_rccl_and_equals_mipt_render(B, setup):
    scene, ctx, cam, ref_path, ref_direct = setup
    node = B.Node(scene, ngpus=1)
    assert node.num_gpus == 1 and node.uses_rccl
    img = node.render(cam, mode=B.MODE_PATH, spp=5, max_bounces=4, seed=3)
    assert np.array_equal(bits(img), bits(ref_path))
    assert np.array_equal(bits(node.render(cam, mode=B.MODE_DIRECT)), bits(ref_direct))
    # progressive across calls: 2 + 3 samples == 5 samples; a call at sample 0 starts a new frame
    node.render(cam, mode=B.MODE_PATH, spp=2, sample_offset=0, max_bounces=4, seed=3)
    img = node.render(cam, mode=B.MODE_PATH, spp=3, sample_offset=2, max_bounces=4, seed=3)
    assert np.array_equal(bits(img), bits(ref_path))
    # another resolution and tiling on the same node: the frame memory is laid out again
    cam2 = B.default_camera(96, 64)
    assert np.array_equal(bits(node.render(cam2, mode=B.MODE_DIRECT, tile_w=16, tile_h=16)), bits(ctx.render(cam2, mode=B.MODE_DIRECT)))
    assert node.render(cam, mode=B.MODE_DIRECT, readback=False) is None
    st = node.stats()
    assert st["rays_closest"] > 0 and st["last_frame_ms"] > 0
    # the convenience entry point of SURVEY 8(b)
    img = B.render_node(scene, cam, 1, mode=B.MODE_PATH, spp=5, max_bounces=4, seed=3)
    assert np.array_equal(bits(img), bits(ref_path))


@pytest.mark.parametrize("n", [2, 3, 8])
def test_ranks_sharing_a_gpu_assemble_the_one_gpu_frame(B, setup, n):
    scene, ctx, cam, ref_path, ref_direct = setup
    node = B.Node(scene, devices=[0] * n)
    assert node.num_gpus == n and not node.uses_rccl
    for _ in range(2):  # twice: the send / receive buffers are reused
        img = node.render(cam, mode=B.MODE_PATH, spp=5, max_bounces=4, seed=3)
        assert np.array_equal(bits(img), bits(ref_path)), f"{n} ranks: {(bits(img) != bits(ref_path)).any(axis=2).sum()} pixels differ"
    assert np.array_equal(bits(node.render(cam, mode=B.MODE_DIRECT)), bits(ref_direct))
    # the ranks together traced exactly the rays of the one-context frame
    ctx.reset_stats()
    ctx.render(cam, mode=B.MODE_DIRECT)
    one = ctx.stats()
    before = node.stats()
    node.render(cam, mode=B.MODE_DIRECT)
    after = node.stats()
    assert after["rays_closest"] - before["rays_closest"] == one["rays_closest"]
    assert after["rays_occluded"] - before["rays_occluded"] == one["rays_occluded"]


def test_node_errors(B, setup):
    scene, ctx, cam, _, _ = setup
    with pytest.raises(B.MiptError, match="does not exist"):
        B.Node(scene, devices=[0, 99])
    with pytest.raises(B.MiptError):
        B.Node(scene, ngpus=0)
    node = B.Node(scene, devices=[0, 0])
    with pytest.raises(B.MiptError, match="multiples of 8"):
        node.render(cam, mode=B.MODE_DIRECT, tile_w=12, tile_h=8)
    with pytest.raises(B.MiptError):   # every rank refuses the same bad mode before the gather: no rank is left waiting
        node.render(cam, mode=7)
    assert np.array_equal(bits(node.render(cam, mode=B.MODE_DIRECT)), bits(setup[4]))  # and the node still works


def test_grouped_send_receive_path_in_a_fresh_process(B, setup, tmp_path):
    """ncclGather is an RCCL extension; a build of the library without it gathers with one group of ncclSend / ncclRecv.
    MIPT_NODE_NO_NCCLGATHER=1 forces that path (decided when RCCL is first loaded, hence a fresh process): the 1-GPU
    node's frame must again equal the one-context frame bit for bit."""
    import os, subprocess, sys
    script = tmp_path / "p2p.py"
    script.write_text(
        "import importlib, sys, numpy as np\n"
        f"sys.path[:0] = [{os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r}, {os.path.dirname(os.path.abspath(__file__))!r}]\n"
        "import scene_fixtures as sf\n"
        "B = importlib.import_module('path-tracer_amd').binding\n"
        "scene = sf.product_scene_from_mesh_arrays(*sf.load_scene_arrays('cornellbox-glass'), env=sf.procedural_env(64))\n"
        "cam = B.default_camera(200, 150)\n"
        "ref = B.Context(scene, device=0).render(cam, mode=B.MODE_PATH, spp=3, max_bounces=4, seed=3)\n"
        "node = B.Node(scene, ngpus=1)\n"
        "assert node.uses_rccl\n"
        "img = node.render(cam, mode=B.MODE_PATH, spp=3, max_bounces=4, seed=3)\n"
        "assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))\n"
        "print('p2p ok')\n")
    env = dict(os.environ, MIPT_NODE_NO_NCCLGATHER="1")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=200)
    assert r.returncode == 0 and "p2p ok" in r.stdout, r.stderr[-2000:]


def test_bench_rccl_path_with_one_rank(tmp_path):
    """bench.py's N > 1 code path -- RCCL process group bound to the device, per-step tile gather on device tensors,
    barriers, MAX / SUM reductions, the weak-scaling loop -- run with ONE rank (--force-dist), which is all a one-GPU box
    allows; the headline must equal what the plain run reports for the same frames (same rays, a sane rate)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--force-dist", "--native-node"], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), ("stdout must carry the JSON line and nothing else (RCCL's banner goes to stderr)", r.stdout[:400])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["value"] > 1000.0
    assert "side stream" in out["config"]["gather"]  # the per-step gather runs beside the next step's kernels (TileGather.gather_async)
    weak = out["config"]["weak_scaling"]
    assert weak is not None and weak["spp_per_step"] == 16 and abs(weak["value"] / out["value"] - 1.0) < 0.1
    # ... and beside it the NATIVE node (mipt_node_render in a child process: one process, thread per GPU, RCCL), the
    # same frame: same rays per step, a rate in the same range
    nat = out["config"]["native_node"]
    assert nat is not None and "error" not in nat, nat
    assert nat["uses_rccl"] and nat["n_gpus"] == 1 and abs(nat["value"] / out["value"] - 1.0) < 0.15, nat


def test_tile_gather_on_a_side_stream_leaves_the_frame_as_the_synchronous_one(tmp_path):
    """TileGather.gather_async (bench.py's per-step gather for N > 1: pack behind an event, collective and unpack on a side
    stream, the current stream waiting for the pack only) in a one-rank RCCL world, several steps in a row with work
    enqueued behind each: the frame ends up as gather() leaves it.  In a process of its own (a process group per process)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "gather_async.py"
    script.write_text(
        "import importlib, os, sys, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {root!r})\n"
        "tiles = importlib.import_module('path-tracer_amd').tiles\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29577', RANK='0', WORLD_SIZE='1')\n"
        "dev = torch.device('cuda', 0); torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', device_id=dev)\n"
        "w, h = 640, 360\n"
        "g = tiles.TileGather(w, h, 0, 1, dev, always=True)\n"
        "a = torch.arange(w * h * 3, dtype=torch.float32, device=dev)\n"
        "b = a.clone()\n"
        "for k in range(4):\n"
        "    a += 1.0; b += 1.0\n"
        "    g.gather_async(a)\n"
        "    g.gather(b)\n"
        "g.finish(); torch.cuda.synchronize()\n"
        "assert torch.equal(a, b) and float(a[0]) == 4.0\n"
        "dist.destroy_process_group()\n"
        "print('gather ok')\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=200)
    assert r.returncode == 0 and "gather ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_calls_without_a_host_buffer_return_enqueued_and_the_last_call_fetches_the_frame(B, setup, devices):
    """a progressive sequence as configs[4] runs it: every call but the last leaves the frame on GPU 0 (the call returns
    with its gather enqueued on the transfer stream, the next call's kernels run beside it, the two send slots are used
    in turn); the last call fetches it.  1 + 1 + 1 + 2 samples == the one-context 5-sample frame bit for bit, with the
    RCCL transport (one GPU) and with ranks sharing the GPU; then a fresh frame on the same node."""
    scene, ctx, cam, ref_path, ref_direct = setup
    node = B.Node(scene, devices=devices)
    for rep in range(2):
        for k in range(3):
            assert node.render(cam, mode=B.MODE_PATH, spp=1, sample_offset=k, max_bounces=4, seed=3, readback=False) is None
        img = node.render(cam, mode=B.MODE_PATH, spp=2, sample_offset=3, max_bounces=4, seed=3)
        assert np.array_equal(bits(img), bits(ref_path)), (devices, rep)
    node.render(cam, mode=B.MODE_DIRECT, readback=False)
    node.synchronize()
    assert np.array_equal(bits(node.render(cam, mode=B.MODE_DIRECT)), bits(ref_direct))


def test_more_than_one_physical_gpu_or_a_stated_skip(B, setup):
    """what a one-GPU box cannot run says so: on a machine with several GPUs this builds the node over ALL of them (one
    RCCL communicator over distinct devices, grouped send / receive over xGMI) and compares its frames with the one-GPU
    frames bit for bit -- the first multi-GPU run reports what ran instead of silently having run nothing"""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"{n} GPU visible: ncclCommInitAll over distinct devices, the RCCL transfer across xGMI and WORLD_SIZE > 1 on "
                    "the nccl backend have not run on this machine (the same code runs here with one GPU and with ranks sharing it)")
    scene, ctx, cam, ref_path, ref_direct = setup
    node = B.Node(scene, ngpus=n)
    assert node.num_gpus == n and node.uses_rccl
    for _ in range(2):
        assert np.array_equal(bits(node.render(cam, mode=B.MODE_PATH, spp=5, max_bounces=4, seed=3)), bits(ref_path))
    for k in range(3):
        node.render(cam, mode=B.MODE_PATH, spp=1, sample_offset=k, max_bounces=4, seed=3, readback=False)
    assert np.array_equal(bits(node.render(cam, mode=B.MODE_PATH, spp=2, sample_offset=3, max_bounces=4, seed=3)), bits(ref_path))
    assert np.array_equal(bits(node.render(cam, mode=B.MODE_DIRECT)), bits(ref_direct))
