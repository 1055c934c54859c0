"""PNG -> ABGR texture words (host/TextureIO.h through libfyprt_host.so), the contract of the reference's
Texture::Texture (Texture.cu:8-40, stb_image forced to 4 channels).  Pins: round trips through an independent encoder for
every colour type / bit depth / filter type, hand-written expected words, rejection of what is not supported — and, when
the reference tree is present (this container only, never on the GPU box), its own 2048x2048 texture assets decode and agree
with an independent zlib + numpy unfilter."""
import struct
import zlib
from pathlib import Path

import numpy as np
import pytest

from fypraytracer_amd import texture

REF_ASSETS = Path("/root/reference/FYPRayTracer/Assets/3D Models/Test")


def _abgr(r, g, b, a):
    return (np.uint32(a) << 24) | (np.uint32(b) << 16) | (np.uint32(g) << 8) | np.uint32(r)


def test_known_words(tmp_path):
    rgba = np.array([[[255, 0, 0, 255], [0, 255, 0, 128]], [[0, 0, 255, 0], [16, 32, 48, 64]]], np.uint8)
    (tmp_path / "a.png").write_bytes(texture.encode_png(rgba, 6))
    px = texture.load_png(tmp_path / "a.png")
    assert px.tolist() == [[0xFF0000FF, 0x8000FF00], [0x00FF0000, 0x40302010]]      # A<<24 | B<<16 | G<<8 | R, row 0 first


@pytest.mark.parametrize("ctype,channels", [(0, 1), (2, 3), (4, 2), (6, 4)])
@pytest.mark.parametrize("depth", [8, 16])
def test_round_trip_all_filters(tmp_path, ctype, channels, depth):
    rng = np.random.default_rng(ctype * 10 + depth)
    h, w = 13, 17
    hi = 65535 if depth == 16 else 255
    s = rng.integers(0, hi + 1, (h, w, channels)).astype(np.uint16 if depth == 16 else np.uint8)
    (tmp_path / "t.png").write_bytes(texture.encode_png(s, ctype, bit_depth=depth, filters=[0, 1, 2, 3, 4]))
    px = texture.load_png(tmp_path / "t.png")
    s8 = (s >> 8).astype(np.uint32) if depth == 16 else s.astype(np.uint32)     # 16-bit: the high byte (stb's conversion)
    if ctype == 0:
        want = _abgr(s8[..., 0], s8[..., 0], s8[..., 0], 255)
    elif ctype == 2:
        want = _abgr(s8[..., 0], s8[..., 1], s8[..., 2], 255)
    elif ctype == 4:
        want = _abgr(s8[..., 0], s8[..., 0], s8[..., 0], s8[..., 1])
    else:
        want = _abgr(s8[..., 0], s8[..., 1], s8[..., 2], s8[..., 3])
    assert np.array_equal(px, want)


def test_palette_with_transparency(tmp_path):
    pal = np.array([[10, 20, 30], [40, 50, 60], [70, 80, 90]], np.uint8)
    idx = np.array([[0, 1, 2], [2, 1, 0]], np.uint8)[..., None]
    (tmp_path / "p.png").write_bytes(texture.encode_png(idx, 3, palette=pal, trns=[0, 128], filters=[1, 4]))
    px = texture.load_png(tmp_path / "p.png")
    alpha = np.array([0, 128, 255])
    want = _abgr(pal[idx[..., 0], 0], pal[idx[..., 0], 1], pal[idx[..., 0], 2], alpha[idx[..., 0]])
    assert np.array_equal(px, want)


def test_rejects_bad_input(tmp_path):
    (tmp_path / "x.png").write_bytes(b"not a png at all")
    with pytest.raises(ValueError, match="not a PNG"):
        texture.load_png(tmp_path / "x.png")
    good = bytearray(texture.encode_png(np.zeros((4, 4, 4), np.uint8), 6))
    good[40] ^= 0xFF                                                        # corrupt the IDAT payload -> CRC mismatch
    (tmp_path / "c.png").write_bytes(bytes(good))
    with pytest.raises(ValueError, match="CRC"):
        texture.load_png(tmp_path / "c.png")
    ihdr = struct.pack(">IIBBBBB", 2, 2, 8, 6, 0, 0, 1)                     # interlaced
    png = b"\x89PNG\r\n\x1a\n" + texture._chunk(b"IHDR", ihdr) + texture._chunk(b"IDAT", zlib.compress(b"\0" * 18)) + texture._chunk(b"IEND", b"")
    (tmp_path / "i.png").write_bytes(png)
    with pytest.raises(ValueError, match="interlaced"):
        texture.load_png(tmp_path / "i.png")
    with pytest.raises(ValueError, match="cannot"):
        texture.load_png(tmp_path / "missing.png")


def _independent_decode(path):
    """zlib + numpy unfilter (Sub / Up vectorised, Average / Paeth per byte column) — a second implementation to compare with."""
    raw = Path(path).read_bytes()
    pos, idat, hdr = 8, b"", None
    while pos < len(raw):
        n, kind = struct.unpack(">I4s", raw[pos:pos + 8])
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", raw[pos + 8:pos + 21])
        elif kind == b"IDAT":
            idat += raw[pos + 8:pos + 8 + n]
        pos += 12 + n
    w, h, depth, ctype = hdr[:4]
    ch = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
    data = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * ch)
    out = np.zeros((h, w * ch), np.uint8)
    prev = np.zeros(w * ch, np.int32)
    for y in range(h):
        ft, f = int(data[y, 0]), data[y, 1:].astype(np.int32)
        if ft == 0:
            cur = f
        elif ft == 2:
            cur = (f + prev) & 0xFF
        elif ft == 1:
            cur = (np.cumsum(f.reshape(w, ch), axis=0) & 0xFF).reshape(-1)
        else:
            cur = np.zeros(w * ch, np.int32)
            left = np.zeros(ch, np.int32); upleft = np.zeros(ch, np.int32)
            fr, pr = f.reshape(w, ch), prev.reshape(w, ch)
            cr = cur.reshape(w, ch)
            for x in range(w):
                up = pr[x]
                if ft == 3:
                    pred = (left + up) >> 1
                else:
                    p = left + up - upleft
                    pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - upleft)
                    pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, upleft))
                cr[x] = (fr[x] + pred) & 0xFF
                left, upleft = cr[x], up
        out[y] = cur
        prev = cur
    px = out.reshape(h, w, ch).astype(np.uint32)
    if ctype == 2:
        return _abgr(px[..., 0], px[..., 1], px[..., 2], 255)
    return _abgr(px[..., 0], px[..., 1], px[..., 2], px[..., 3])


@pytest.mark.skipif(not REF_ASSETS.exists(), reason="reference assets are only present in the build container")
@pytest.mark.parametrize("name", ["bananaDiffuse.png", "toasterBaseColor.png"])
def test_reference_texture_assets_decode(name):
    px = texture.load_png(REF_ASSETS / name)
    assert px.shape == (2048, 2048)
    rows = slice(0, 24)                                                     # the per-byte Python unfilter is slow: compare the first rows
    want = _independent_decode_rows(REF_ASSETS / name, 24)
    assert np.array_equal(px[rows], want)


def _independent_decode_rows(path, nrows):
    import io
    raw = Path(path).read_bytes()
    pos, idat, hdr = 8, b"", None
    while pos < len(raw):
        n, kind = struct.unpack(">I4s", raw[pos:pos + 8])
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", raw[pos + 8:pos + 21])
        elif kind == b"IDAT":
            idat += raw[pos + 8:pos + 8 + n]
        pos += 12 + n
    w, h, depth, ctype = hdr[:4]
    ch = {2: 3, 6: 4}[ctype]
    need = nrows * (1 + w * ch)
    part = zlib.decompressobj().decompress(idat, need)
    small = texture.encode_png(np.zeros((1, 1, ch), np.uint8), ctype)       # reuse the chunk writer for a cropped file
    ihdr = struct.pack(">IIBBBBB", w, nrows, 8, ctype, 0, 0, 0)
    tmp = b"\x89PNG\r\n\x1a\n" + texture._chunk(b"IHDR", ihdr) + texture._chunk(b"IDAT", zlib.compress(part[:need])) + texture._chunk(b"IEND", b"")
    p = Path("/tmp/_crop.png"); p.write_bytes(tmp)
    return _independent_decode(p)
