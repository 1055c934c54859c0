"""Texture ingest: Python face of host/TextureIO.h — Texture::Texture(std::string&) of the reference
(Texture.cu:8-40: 4-channel load, A<<24 | B<<16 | G<<8 | R, rows top to bottom).  `load_png` goes through the C++ reader
(libfyprt_host.so); `encode_png` is a minimal writer (filter 0, zlib) used to make fixtures and to round-trip in tests."""
import ctypes as C
import struct
import zlib
from pathlib import Path

import numpy as np

_LIB_PATH = Path(__file__).resolve().parent / "host" / "libfyprt_host.so"
_lib = None


def _host_lib():
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise RuntimeError(f"{_LIB_PATH} is missing: run `python __graft_entry__.py build` (host/build.sh)")
        _lib = C.CDLL(str(_LIB_PATH))
        _lib.fyprt_host_load_png.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p, C.c_char_p, C.c_size_t]
    return _lib


def load_png(path):
    """-> (height, width) uint32 array of ABGR words, the layout fyprt_texture / Scene.textures expect."""
    lib, w, h, err = _host_lib(), C.c_uint32(), C.c_uint32(), C.create_string_buffer(256)
    p = str(path).encode()
    if lib.fyprt_host_load_png(p, C.byref(w), C.byref(h), None, err, 256):
        raise ValueError(f"cannot load {path}: {err.value.decode()}")
    px = np.empty((h.value, w.value), dtype=np.uint32)
    if lib.fyprt_host_load_png(p, C.byref(w), C.byref(h), px.ctypes.data, err, 256):
        raise ValueError(f"cannot load {path}: {err.value.decode()}")
    return px


def _chunk(kind, data):
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data))


def encode_png(samples, color_type, bit_depth=8, palette=None, trns=None, filters=None):
    """samples: (H, W, channels) array of uint8 / uint16 samples (channels = 1, 3, 1, 2, 4 for colour type 0, 2, 3, 4, 6).
    `filters`: optional per-row filter types (0-4) to apply when writing; default 0."""
    a = np.asarray(samples)
    h, w = a.shape[:2]
    raw = a.astype(">u2" if bit_depth == 16 else np.uint8).reshape(h, -1).view(np.uint8).reshape(h, -1)
    bpp = raw.shape[1] // w
    out = bytearray()
    prev = np.zeros(raw.shape[1], dtype=np.int32)
    for y in range(h):
        ft = 0 if filters is None else int(filters[y % len(filters)])
        cur = raw[y].astype(np.int32)
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        upleft = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if ft == 0:
            f = cur
        elif ft == 1:
            f = cur - left
        elif ft == 2:
            f = cur - prev
        elif ft == 3:
            f = cur - ((left + prev) >> 1)
        else:
            p = left + prev - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
            f = cur - pred
        out.append(ft)
        out += (f & 0xFF).astype(np.uint8).tobytes()
        prev = cur
    png = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bit_depth, color_type, 0, 0, 0))
    if palette is not None:
        png += _chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    if trns is not None:
        png += _chunk(b"tRNS", np.asarray(trns, np.uint8).tobytes())
    data = zlib.compress(bytes(out), 6)
    half = len(data) // 2
    png += _chunk(b"IDAT", data[:half]) + _chunk(b"IDAT", data[half:]) + _chunk(b"IEND", b"")      # two IDAT chunks on purpose
    return png
