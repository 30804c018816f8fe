"""Pins the CPU oracle (oracle/pt_oracle.c) against the outputs of the
UNMODIFIED reference recorded in SURVEY.md 8(c) (default camera, 512x384,
SAH on).  The reference holds no tests or golden vectors of its own and
cannot be built in this image (GLEW/GLFW headers absent, stand-ins not
allowed), so these recorded outputs are what parity is anchored on.
Scenes come from tests/golden/scene_*.npz (the build's own loader run on the
reference's media files; generator: tests/golden/make_fixtures.py)."""
import numpy as np
import pytest

# scene -> (tris, nodes, leaves, depth, max tris/leaf, primary hits of 196608, sum t, FNV-1a-32 of hit t bits,
#           threaded node visits/ray, tri tests/ray, bit-trail box tests/ray, tri tests/ray,
#           1-spp path-traced mean (R+G+B) per pixel with a constant-1 64x64 env, light (3,3,3)/1000 W, fresh PCG32)
PINS = {
    "cornellbox": (36, 63, 32, 15, 2, 105025, 162229.843437, 0x650D0DE6, 16.54, 1.59, 17.47, 1.22, 1.537802),
    "bunny": (69451, 125629, 62815, 21, 3, 61066, 78886.185961, 0x0449E289, 34.30, 1.99, 20.81, 0.82, 2.544141),
    "teapot-glass": (15704, 30957, 15479, 21, 2, 22466, 29858.619651, 0xE14F072F, 15.85, 1.42, 9.57, 0.47, 3.000000),
    "testObj": (11950, 21763, 10882, 30, 6, 196608, None, None, 66.92, 7.18, 47.94, 3.38, 1.025928),
}


@pytest.fixture(scope="module")
def scenes(sf):
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = sf.oracle_scene_from_mesh_arrays(*sf.load_scene_arrays(name),
                                                           env=np.ones((64, 64, 3), np.float32))
        return cache[name]
    return get


@pytest.mark.parametrize("name", list(PINS))
def test_bvh_and_primary_hits_match_reference_pins(ob, scenes, name):
    tris, nodes, leaves, depth, maxleaf, hits, sum_t, fnv, nv, nt, bv, bt, _ = PINS[name]
    s = scenes(name)
    st = s.stats()
    assert (s.ntris, st["nodes"], st["leaves"], st["depth"], st["max_leaf"]) == (tris, nodes, leaves, depth, maxleaf)
    cam = ob.make_camera(512, 384)
    o, d = s.eye_rays(cam)
    ref_bits = None
    for variant, (ev, et) in ((ob.THREADED, (nv, nt)), (ob.BITTRAIL, (bv, bt)), (ob.RECURSIVE, (None, None))):
        tri, t, bary = s.intersect(o, d, variant=variant)
        hit = tri >= 0
        assert int(hit.sum()) == hits
        if sum_t is not None:
            assert abs(float(t[hit].astype(np.float64).sum()) - sum_t) < 5e-6
            assert ob.fnv1a32_of_bits(t[hit].view(np.uint32)) == fnv
        if ev is not None:
            assert round(s.last_counts[0] / t.size, 2) == ev and round(s.last_counts[1] / t.size, 2) == et
        # the three traversals agree bitwise (SURVEY 8(c) "cross-variant agreement")
        if ref_bits is None:
            ref_bits = t.view(np.uint32).copy()
        else:
            assert np.array_equal(t.view(np.uint32), ref_bits)


@pytest.mark.parametrize("name", ["cornellbox", "teapot-glass"])
def test_bruteforce_agrees(ob, scenes, name):
    """bruteforceIntersect (cs488.h:931-947) == BVH traversals, as the survey measured on these two scenes"""
    s = scenes(name)
    cam = ob.make_camera(128, 96)
    o, d = s.eye_rays(cam)
    a = s.intersect(o, d, variant=ob.THREADED)
    b = s.intersect(o, d, variant=ob.BRUTEFORCE)
    assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


@pytest.mark.parametrize("name", list(PINS))
def test_pathtrace_mean_radiance_matches_reference_pin(ob, scenes, name):
    """hop 1 of the image parity chain: global PCG32 stream, serial pixel order, no bounce cap, bit-trail
    traversal (what Scene instantiates) -> the reference's own 1-spp mean radiance to 6 decimals"""
    s = scenes(name)
    ob.pcg32_reset()
    img = s.pathtrace(ob.make_camera(512, 384), spp=1, max_bounces=-1, rng_mode=0, variant=ob.BITTRAIL)
    assert abs(img.astype(np.float64).sum() / (512 * 384) - PINS[name][-1]) < 1.5e-6


def test_counter_rng_statistically_matches_pcg32(ob, scenes):
    """hop 2 preparation: the counter RNG the GPU uses gives the same expectation as the reference's stream"""
    s = scenes("cornellbox")
    cam = ob.make_camera(128, 96)
    ob.pcg32_reset()
    spp = 16
    a = s.pathtrace(cam, spp=spp, max_bounces=-1, rng_mode=0)
    b = s.pathtrace(cam, spp=spp, max_bounces=-1, rng_mode=1, seed=11)
    pa, pb = a.sum(axis=2).reshape(-1).astype(np.float64), b.sum(axis=2).reshape(-1).astype(np.float64)
    sigma = np.sqrt((pa.var() + pb.var()) / pa.size)  # std of the difference of the two image means (upper bound)
    assert abs(pa.mean() - pb.mean()) <= 3 * sigma


def test_pcg32_known_answers(ob):
    """PCG32::pcg32_fast (cs488.h:134-147) from its seed: first outputs computed by hand from the recurrence"""
    state, mult, out = 0xCAFEF00DD15EA5E5, 6364136223846793005, []
    for _ in range(4):
        x = state
        count = x >> 61
        state = (x * mult) & 0xFFFFFFFFFFFFFFFF
        x ^= x >> 22
        out.append(np.float32(np.float64((x >> (22 + count)) & 0xFFFFFFFF) / 4294967296.0))
    ob.pcg32_reset()
    got = [np.float32(ob.lib().orc_pcg32_rand()) for _ in range(4)]
    assert got == out


def test_golden_fixtures_are_the_oracles_output(ob, sf, scenes):
    """committed golden records == a fresh oracle run (guards the fixtures against drift)"""
    for name in ("cornellbox", "teapot-glass"):
        g = sf.load_golden(name)
        s = scenes(name)
        o, d = s.eye_rays(ob.make_camera(int(g["width"]), int(g["height"])))
        tri, t, bary = s.intersect(o, d)
        assert ob.fnv1a32_of_bits(tri.view(np.uint32)) == int(g["fnv_tri"])
        st = int(g["stride"])
        assert np.array_equal(tri[::st], g["sub_tri"]) and np.array_equal(t[::st].view(np.uint32), g["sub_t"].view(np.uint32))
