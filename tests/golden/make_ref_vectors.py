"""Generates tests/golden/ref_linalg.npz and tests/golden/ref_stb.npz from the two reference sources that compile
UNMODIFIED in the build container (oracle/Makefile `ref`: /root/reference/cpu/src/linalg.h and stb_image.h, read by
include path into oracle/_ref/).  Run in the build container, where /root/reference exists:

    make -C oracle ref && python tests/golden/make_ref_vectors.py

ref_linalg.npz   inputs a, b, c, d (n x 3 float32: random, widely scaled, nearly cancelling, and special values) and, per
                 op, what the reference's linalg.h returns (dot, cross, normalize, determinant(float3x3), mul(float3x3,
                 float3), length2, normalize(float2), float3 / float3) -- bit patterns as uint32.
ref_stb.npz      what the reference's stb_image.h decodes: default.png (the file's 113 bytes, stbi_load bytes, stbi_loadf
                 floats), test.png (sha256 of stbi_load bytes and stbi_loadf floats; the 1.3 MB file stays in the
                 reference's media directory; its texels are already in scene_textured.npz), and two Radiance files
                 made here (flat and run-length encoded scanlines; file bytes + stbi_loadf floats).

Data only: inputs and the reference's outputs.  Nothing of the reference's source text.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import oracle_binding as ob  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
MEDIA = "/root/reference/media"
OPS = ("dot", "cross", "normalize", "det3", "mul", "length2", "normalize2", "div")


def linalg_inputs(n=6144, seed=20261111):
    rng = np.random.default_rng(seed)
    cols = []
    for k in range(4):
        x = rng.normal(size=(n, 3))
        x[n // 4: n // 2] *= np.exp2(rng.integers(-60, 60, size=(n // 4, 1)))        # widely scaled rows
        x[n // 2: 5 * n // 8] *= np.exp2(rng.integers(-20, 20, size=(n // 8, 3)))     # ... and components
        cols.append(x.astype(np.float32))
    a, b, c, d = cols
    # nearly cancelling operands: b close to a, c close to a combination of a and b (determinants near 0)
    s = slice(5 * n // 8, 3 * n // 4)
    b[s] = a[s] * np.float32(1 + 2.0 ** -20) + (rng.normal(size=a[s].shape) * 1e-6).astype(np.float32)
    c[s] = (a[s] * np.float32(0.25) + b[s] * np.float32(0.75)) + (rng.normal(size=a[s].shape) * 1e-7).astype(np.float32)
    # special values: zeros of both signs, denormals, huge, tiny, infinities, NaN, exact small integers
    special = np.array([0.0, -0.0, 1.0, -1.0, 2.0, 3.0, 0.5, 1e-45, -1e-45, 1e-39, 1.17549435e-38, 3.4028235e38, -3.4028235e38,
                        1e19, 1e-19, np.inf, -np.inf, np.nan, 0.1, 1.0 / 3.0], np.float32)
    s = slice(3 * n // 4, 7 * n // 8)
    m = s.stop - s.start
    for arr in (a, b, c, d):
        arr[s] = special[rng.integers(0, len(special), size=(m, 3))]
    # axis-aligned unit vectors and the camera's own vectors (cs488.h:81-83)
    s = slice(7 * n // 8, n)
    m = s.stop - s.start
    unit = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1], [0, 0, 1.5], [0, 0, -1.5]], np.float32)
    for arr in (a, b, c, d):
        arr[s] = unit[rng.integers(0, len(unit), size=m)]
    return a, b, c, d


def write_hdr(path, rgbe, rle):
    """a Radiance file from (h, w, 4) uint8 RGBE pixels, top row first; rle: new-style run-length encoded scanlines"""
    h, w, _ = rgbe.shape
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w))
        if not rle:
            f.write(rgbe.tobytes())
            return
        for j in range(h):
            f.write(bytes([2, 2, w >> 8, w & 255]))
            for ch in range(4):
                row = rgbe[j, :, ch]
                i = 0
                while i < w:
                    run = 1
                    while i + run < w and run < 127 and row[i + run] == row[i]:
                        run += 1
                    if run >= 3:
                        f.write(bytes([128 + run, int(row[i])]))
                        i += run
                    else:
                        k = i
                        while k < w and k - i < 128:
                            r2 = 1
                            while k + r2 < w and r2 < 3 and row[k + r2] == row[k]:
                                r2 += 1
                            if r2 >= 3:
                                break
                            k += 1
                        f.write(bytes([k - i]) + row[i:k].tobytes())
                        i = k


def main():
    L = ob.ref_linalg()
    S = ob.ref_stb()
    if L is None or S is None:
        raise SystemExit("oracle/_ref is not built: run `make -C oracle ref` where /root/reference exists")
    a, b, c, d = linalg_inputs()
    out = dict(a=a, b=b, c=c, d=d)
    for op, name in enumerate(OPS):
        out["ref_" + name] = ob.vec_ops(op, a, b, c, d, fn=L.ref_vec_ops).view(np.uint32)
    np.savez_compressed(os.path.join(OUT, "ref_linalg.npz"), **out)
    print("ref_linalg.npz:", a.shape[0], "elements x", len(OPS), "ops,", os.path.getsize(os.path.join(OUT, "ref_linalg.npz")), "bytes")

    g = {}
    p = os.path.join(MEDIA, "default.png")
    g["default_png_file"] = np.frombuffer(open(p, "rb").read(), np.uint8)
    g["default_png_u8"] = ob.ref_stb_load8(S, p)
    g["default_png_f32"] = ob.ref_stb_loadf(S, p).view(np.uint32)
    p = os.path.join(MEDIA, "test.png")
    u8, f32 = ob.ref_stb_load8(S, p), ob.ref_stb_loadf(S, p)
    g["test_png_shape"] = np.array(u8.shape, np.int32)
    g["test_png_u8_sha256"] = np.frombuffer(hashlib.sha256(u8.tobytes()).digest(), np.uint8)
    g["test_png_f32_sha256"] = np.frombuffer(hashlib.sha256(f32.tobytes()).digest(), np.uint8)
    g["test_png_f32_rows"] = f32[::97].view(np.uint32)  # every 97th row in full
    # Radiance files: RGBE pixels with every exponent class (0 = black, small, large), flat and run-length encoded
    rng = np.random.default_rng(7)
    h, w = 24, 40
    rgbe = rng.integers(0, 256, size=(h, w, 4)).astype(np.uint8)
    rgbe[:, :, 3] = rng.integers(100, 160, size=(h, w))
    rgbe[0, :8, 3] = 0                     # e == 0: black, whatever the mantissas say
    rgbe[1, :, :] = rgbe[1, :1, :]         # a constant row: one long run per channel
    rgbe[2, 10:30, 0] = 77                 # a run inside literal data
    rgbe[3, :, 3] = np.arange(w) + 1       # tiny exponents
    rgbe[4, :, 3] = 255 - np.arange(w)     # huge exponents
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        for tag, rle in (("flat", False), ("rle", True)):
            q = os.path.join(tmp, tag + ".hdr")
            write_hdr(q, rgbe, rle)
            g[f"hdr_{tag}_file"] = np.frombuffer(open(q, "rb").read(), np.uint8)
            dec = ob.ref_stb_loadf(S, q)
            assert dec is not None and dec.shape == (h, w, 3), tag
            g[f"hdr_{tag}_f32"] = dec.view(np.uint32)
        assert np.array_equal(g["hdr_flat_f32"], g["hdr_rle_f32"])
    np.savez_compressed(os.path.join(OUT, "ref_stb.npz"), **g)
    print("ref_stb.npz:", os.path.getsize(os.path.join(OUT, "ref_stb.npz")), "bytes; test.png", tuple(u8.shape))


if __name__ == "__main__":
    main()
