"""Pins against the two reference sources that compile UNMODIFIED in the build container (oracle/Makefile `ref`):
linalg.h (the vector primitives whose operation order defines bit-exactness) and stb_image.h (what
Image::load, cs488.h:204-222, and loadTexture, cs488.h:975-988, decode with).  tests/golden/ref_linalg.npz and
ref_stb.npz hold inputs and the reference's outputs (tests/golden/make_ref_vectors.py); they travel, the reference does
not.  What this pins: the oracle's linalg restatement, the product's pt/vec.hpp on the host AND on the GPU, and the
product's own PNG / Radiance decoders.  What it cannot pin: anything in cs488.h itself (traversal, integrators,
getBackground) -- that header needs GLEW / GLFW, which the image lacks."""
import hashlib
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MEDIA = "/root/reference/media"
OPS = ("dot", "cross", "normalize", "det3", "mul", "length2", "normalize2", "div")


def same_bits(x, want_bits):
    """bit-for-bit, except that any NaN equals any NaN (sign and payload of a generated NaN differ between x86 and gfx950;
    the reference's code only ever compares NaNs, it never looks at their bits)"""
    got = np.ascontiguousarray(x, np.float32).reshape(-1).view(np.uint32)
    want = np.asarray(want_bits, np.uint32).reshape(-1)
    nan_g = (got & 0x7fffffff) > 0x7f800000
    nan_w = (want & 0x7fffffff) > 0x7f800000
    return bool(np.all((got == want) | (nan_g & nan_w)))


@pytest.fixture(scope="module")
def lin():
    return np.load(os.path.join(GOLDEN, "ref_linalg.npz"))


@pytest.fixture(scope="module")
def stb():
    return np.load(os.path.join(GOLDEN, "ref_stb.npz"))


def test_oracle_linalg_equals_the_references_linalg_h(ob, lin):
    for op, name in enumerate(OPS):
        assert same_bits(ob.vec_ops(op, lin["a"], lin["b"], lin["c"], lin["d"]), lin["ref_" + name]), name


def test_product_host_linalg_equals_the_references_linalg_h(B, lin):
    for op, name in enumerate(OPS):
        assert same_bits(B.vec_ops(op, lin["a"], lin["b"], lin["c"], lin["d"]), lin["ref_" + name]), name


@pytest.mark.gpu
def test_product_device_linalg_equals_the_references_linalg_h(B, sf, lin):
    """the same header compiled for gfx950 (-ffp-contract=off, correctly rounded divide and sqrt): the device's dot,
    cross, normalize, determinant, mul and division return the reference's bits, NaN inputs, infinities and denormals
    included"""
    ctx = B.Context(sf.product_scene_from_mesh_arrays(*sf.load_scene_arrays("cube")), device=0)
    for op, name in enumerate(OPS):
        assert same_bits(ctx.vec_ops(op, lin["a"], lin["b"], lin["c"], lin["d"]), lin["ref_" + name]), name


def test_live_reference_build_agrees_on_fresh_operands(B, ob):
    """where oracle/_ref is built (the build container): fresh random operands through the reference's header, the
    oracle and the product's host build"""
    L = ob.ref_linalg()
    if L is None:
        pytest.skip("oracle/_ref/libref_linalg.so is not on this machine (needs /root/reference); the committed vectors cover it")
    rng = np.random.default_rng(int.from_bytes(os.urandom(4), "little"))
    n = 200_000
    a, b, c, d = [(rng.normal(size=(n, 3)) * np.exp2(rng.integers(-30, 30, size=(n, 1)))).astype(np.float32) for _ in range(4)]
    for op, name in enumerate(OPS):
        ref = ob.vec_ops(op, a, b, c, d, fn=L.ref_vec_ops)
        assert same_bits(ob.vec_ops(op, a, b, c, d), ref.view(np.uint32)), name
        assert same_bits(B.vec_ops(op, a, b, c, d), ref.view(np.uint32)), name


def _flipped(file_order_bits):
    """Image::load's loop (cs488.h:213-219): the file's top row becomes j = height - 1"""
    return np.asarray(file_order_bits)[::-1]


def test_png_decoder_equals_the_references_stb_image(B, sf, stb, tmp_path):
    p = tmp_path / "default.png"
    p.write_bytes(stb["default_png_file"].tobytes())
    img = B.Image.load(str(p)).pixels()  # float path: stbi_loadf's pow(v / 255, 2.2) on the decoded bytes
    assert img.shape == (128, 128, 3)
    assert same_bits(img, _flipped(stb["default_png_f32"]))
    # 8-bit path (loadTexture): the texels the product's loader made from the reference's media, committed as fixtures
    tex = sf.load_scene_arrays("teapot-glass")[5][0]
    assert np.array_equal(tex, stb["default_png_u8"])
    tex2 = sf.load_scene_arrays("textured")[5][0]
    assert tuple(tex2.shape) == tuple(stb["test_png_shape"])
    assert hashlib.sha256(np.ascontiguousarray(tex2).tobytes()).digest() == stb["test_png_u8_sha256"].tobytes()
    q = os.path.join(MEDIA, "test.png")
    if os.path.exists(q):  # the 1.3 MB file itself is only in the build container
        big = B.Image.load(q).pixels()
        assert hashlib.sha256(np.ascontiguousarray(big[::-1]).tobytes()).digest() == stb["test_png_f32_sha256"].tobytes()
        assert same_bits(big[::-1][::97], stb["test_png_f32_rows"])


def test_radiance_decoder_equals_the_references_stb_image(B, stb, tmp_path):
    for tag in ("flat", "rle"):
        p = tmp_path / f"{tag}.hdr"
        p.write_bytes(stb[f"hdr_{tag}_file"].tobytes())
        img = B.Image.load(str(p)).pixels()
        assert img.shape == (24, 40, 3)
        assert same_bits(img, _flipped(stb[f"hdr_{tag}_f32"])), tag


def test_live_reference_stb_agrees_on_a_written_environment(B, ob, sf, tmp_path):
    """where oracle/_ref is built: the procedural environment written by the product's .hdr writer, decoded by the
    reference's stb_image.h and by the product"""
    S = ob.ref_stb()
    if S is None:
        pytest.skip("oracle/_ref/libref_stb.so is not on this machine (needs /root/reference); the committed vectors cover it")
    p = str(tmp_path / "env.hdr")
    B.save_hdr(p, sf.procedural_env(256))
    ref = ob.ref_stb_loadf(S, p)
    assert same_bits(B.Image.load(p).pixels(), ref[::-1].view(np.uint32))
