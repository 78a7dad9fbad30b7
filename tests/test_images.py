"""HDR image readers / writers of the host library (sky texture in, framebuffer out).

The reference's own skyboxes (Assets/SkyBoxes/*.exr) are not in the checkout, so the files
read here are written by this test with an independent, minimal Python encoder of the two
formats; values read back must equal what was encoded.
"""
import struct
import zlib

import numpy as np
import pytest

from raytracing_amd import host as H


def _exr_filter(raw: bytes) -> bytes:
    a = np.frombuffer(raw, dtype=np.uint8)
    t = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)
    d = t.copy()
    d[1:] = (t[1:] - t[:-1] + 128 + 256) % 256
    return d.astype(np.uint8).tobytes()


def _exr_rle(data: bytes) -> bytes:
    out = bytearray()
    i, n = 0, len(data)
    while i < n:
        j = i
        while j + 1 < n and data[j + 1] == data[i] and j - i < 126:
            j += 1
        run = j - i + 1
        if run >= 3:
            out += struct.pack("b", run - 1) + data[i:i + 1]
            i += run
        else:
            k = i
            while k < n and k - i < 127 and not (k + 2 < n and data[k] == data[k + 1] == data[k + 2]):
                k += 1
            out += struct.pack("b", -(k - i)) + data[i:k]
            i = k
    return bytes(out)


def write_exr(path, img, pixel_type, compression, channels="ABGR"):
    """img: (h, w, len(channels)) float32 in file row order (top first), channel order as named."""
    h, w, nc = img.shape
    order = sorted(range(nc), key=lambda i: channels[i])
    hdr = struct.pack("<ii", 20000630, 2)

    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data

    chl = b"".join(channels[i].encode() + b"\0" + struct.pack("<iBBBBii", pixel_type, 0, 0, 0, 0, 1, 1) for i in order) + b"\0"
    hdr += attr("channels", "chlist", chl)
    hdr += attr("compression", "compression", struct.pack("<B", compression))
    hdr += attr("dataWindow", "box2i", struct.pack("<iiii", 0, 0, w - 1, h - 1))
    hdr += attr("displayWindow", "box2i", struct.pack("<iiii", 0, 0, w - 1, h - 1))
    hdr += attr("lineOrder", "lineOrder", b"\0")
    hdr += attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    hdr += attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0))
    hdr += attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    hdr += b"\0"
    lines = 16 if compression == 3 else 1
    chunks = []
    for y0 in range(0, h, lines):
        raw = bytearray()
        for y in range(y0, min(h, y0 + lines)):
            for i in order:
                col = img[y, :, i]
                raw += (col.astype(np.float16) if pixel_type == 1 else col.astype(np.float32)).tobytes()
        raw = bytes(raw)
        if compression == 0:
            data = raw
        elif compression == 1:
            data = _exr_rle(_exr_filter(raw))
        else:
            data = zlib.compress(_exr_filter(raw))
        if len(data) >= len(raw):
            data = raw
        chunks.append(struct.pack("<ii", y0, len(data)) + data)
    table_at = len(hdr)
    off = table_at + 8 * len(chunks)
    table = b""
    for c in chunks:
        table += struct.pack("<Q", off)
        off += len(c)
    with open(path, "wb") as f:
        f.write(hdr + table + b"".join(chunks))


def _picture(h, w, seed):
    rng = np.random.default_rng(seed)
    img = rng.random((h, w, 4), dtype=np.float32) * 4.0
    img[2:5, 3:9] = 0.25            # flat areas (runs for RLE)
    img[..., 3] = 1.0
    return img


@pytest.mark.parametrize("compression", [0, 1, 2, 3])
@pytest.mark.parametrize("pixel_type", [1, 2])
def test_exr_read(tmp_path, compression, pixel_type):
    img = _picture(37, 23, compression * 2 + pixel_type)          # 37 rows: a partial last ZIP block
    abgr = img[..., [3, 2, 1, 0]]
    p = tmp_path / "sky.exr"
    write_exr(p, abgr, pixel_type, compression)
    want = img.astype(np.float16).astype(np.float32) if pixel_type == 1 else img
    got = H.load_image(p, flip_y=False)
    assert got.shape == (37, 23, 4)
    assert np.array_equal(got, want)
    assert np.array_equal(H.load_image(p, flip_y=True), want[::-1])


def test_exr_rgb_without_alpha_and_luminance_only(tmp_path):
    img = _picture(9, 12, 5)
    p = tmp_path / "rgb.exr"
    write_exr(p, img[..., [2, 1, 0]], 2, 3, channels="BGR")
    got = H.load_image(p, flip_y=False)
    assert np.array_equal(got[..., :3], img[..., :3]) and np.all(got[..., 3] == 1.0)
    p = tmp_path / "y.exr"
    write_exr(p, img[..., :1], 2, 0, channels="Y")
    got = H.load_image(p, flip_y=False)
    for c in range(3):
        assert np.array_equal(got[..., c], img[..., 0])


def test_exr_unsupported_and_corrupt_files_are_errors(tmp_path):
    img = _picture(4, 4, 1)
    p = tmp_path / "piz.exr"
    write_exr(p, img[..., [3, 2, 1, 0]], 2, 4)          # PIZ id in the header
    with pytest.raises(H.HostError, match="PIZ"):
        H.load_image(p)
    p = tmp_path / "cut.exr"
    write_exr(p, img[..., [3, 2, 1, 0]], 2, 0)
    data = p.read_bytes()
    p.write_bytes(data[:-20])
    with pytest.raises(H.HostError):
        H.load_image(p)
    p = tmp_path / "junk.exr"
    p.write_bytes(b"not an image")
    with pytest.raises(H.HostError):
        H.load_image(p)
    with pytest.raises(H.HostError):
        H.load_image(tmp_path / "missing.exr")


def _rgbe(rgb):
    m = rgb.max(axis=-1)
    out = np.zeros(rgb.shape[:-1] + (4,), dtype=np.uint8)
    nz = m >= 1e-32
    frac, e = np.frexp(m[nz])
    scale = frac * 256.0 / m[nz]
    out[nz, :3] = (rgb[nz] * scale[:, None]).astype(np.uint8)
    out[nz, 3] = (e + 128).astype(np.uint8)
    return out


def _rgbe_decode(q):
    f = np.where(q[..., 3] > 0, np.ldexp(1.0, q[..., 3].astype(np.int32) - 136), 0.0).astype(np.float32)
    return q[..., :3].astype(np.float32) * f[..., None]


def _hdr_rle_scanline(q):
    w = q.shape[0]
    out = bytearray([2, 2, w >> 8, w & 255])
    for c in range(4):
        col = bytes(q[:, c])
        i = 0
        while i < w:
            j = i
            while j + 1 < w and col[j + 1] == col[i] and j - i < 126:
                j += 1
            run = j - i + 1
            if run >= 4:
                out += bytes([128 + run, col[i]])
                i += run
            else:
                k = min(w, i + 100)
                out += bytes([k - i]) + col[i:k]
                i = k
    return bytes(out)


@pytest.mark.parametrize("rle", [False, True])
def test_radiance_hdr_read(tmp_path, rle):
    img = _picture(11, 40, 7)[..., :3]
    img[0, 0] = 0.0
    q = _rgbe(img)
    p = tmp_path / "sky.hdr"
    body = b"".join(_hdr_rle_scanline(q[y]) if rle else q[y].tobytes() for y in range(q.shape[0]))
    p.write_bytes(b"#?RADIANCE\n# made by a test\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n-Y 11 +X 40\n" + body)
    got = H.load_image(p, flip_y=False)
    assert got.shape == (11, 40, 4)
    assert np.array_equal(got[..., :3], _rgbe_decode(q))
    assert np.all(got[..., 3] == 1.0)
    # RGBE keeps 8 bits of the largest component
    assert np.allclose(got[..., :3], img, rtol=0, atol=img.max() / 128)


def test_writers_round_trip(tmp_path):
    img = _picture(33, 17, 11)
    img[..., 3] = np.linspace(0, 1, 17, dtype=np.float32)[None, :]
    for flip in (False, True):
        p = tmp_path / f"out{int(flip)}.exr"
        H.save_image(p, img, flip_y=flip)
        assert np.array_equal(H.load_image(p, flip_y=flip), img)
        assert np.array_equal(H.load_image(p, flip_y=not flip), img[::-1])
    p = tmp_path / "out.hdr"
    H.save_image(p, img, flip_y=False)
    assert np.array_equal(H.load_image(p, flip_y=False)[..., :3], _rgbe_decode(_rgbe(img[..., :3])))
    p = tmp_path / "out.ppm"
    H.save_image(p, img, flip_y=False)
    data = p.read_bytes()
    head = b"P6\n17 33\n255\n"
    assert data.startswith(head)
    px = np.frombuffer(data[len(head):], dtype=np.uint8).reshape(33, 17, 3)
    want = np.floor(np.clip(img[..., :3], 0, 1) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)
    assert np.array_equal(px, want)
    with pytest.raises(H.HostError):
        H.save_image(tmp_path / "out.png", img)


def test_exr_hostile_offsets_and_windows_are_errors(tmp_path):
    """A chunk offset near 2^64 (which would wrap the bounds check), one just past the end of the file, and
    a data window whose extent overflows 32 bits must all be refused, not read out of bounds."""
    img = _picture(4, 4, 2)
    good = tmp_path / "good.exr"
    write_exr(good, img[..., [3, 2, 1, 0]], 2, 0)
    data = bytearray(good.read_bytes())
    H.load_image(good)
    # the offset table is the 4 x 8 bytes in front of the first chunk: find it through the first offset value
    first = None
    for at in range(0, len(data) - 8):
        if struct.unpack_from("<Q", data, at)[0] == at + 32:
            first = at
            break
    assert first is not None
    for bad in (2 ** 64 - 4, 2 ** 64 - 1, len(data), len(data) - 2, 2 ** 40):
        d = bytearray(data)
        struct.pack_into("<Q", d, first + 8, bad)
        p = tmp_path / "bad.exr"
        p.write_bytes(bytes(d))
        with pytest.raises(H.HostError):
            H.load_image(p)
    # data window (x0, y0, x1, y1) = (-2^31, 0, 2^31 - 1, 3): width overflows int
    at = bytes(data).index(b"dataWindow\0box2i\0") + len(b"dataWindow\0box2i\0") + 4
    d = bytearray(data)
    struct.pack_into("<iiii", d, at, -2 ** 31, 0, 2 ** 31 - 1, 3)
    p = tmp_path / "wide.exr"
    p.write_bytes(bytes(d))
    with pytest.raises(H.HostError, match="data window"):
        H.load_image(p)
