"""Benchmark / quality logging of the reference's app layer — Python mirror of host/MisUtils.h:
MisUtils::{SaveABGRToBMP, ComputeMSE, ComputePSNR, GetTimestampedFilename} (Utility/MisUtils.cpp:13-157) and the record
name MainLayer::SaveRenderImage / SaveBenchmarkResults build (WalnutApp.cpp:787-875)."""
import struct
import time

import numpy as np

TECHNIQUE_NAMES = ("BRUTE_FORCE", "UNIFORM_SAMPLING", "COSINE_WEIGHTED_SAMPLING", "GGX_SAMPLING", "BRDF_SAMPLING",
                   "LIGHT_SOURCE_SAMPLING", "NEE", "RESTIR_DI", "RESTIR_GI")


def save_abgr_to_bmp(path, abgr):
    """24-bit bottom-up BMP; render row 0 (NDC y = -1) is the first BMP row (MisUtils.cpp:13-95)."""
    abgr = np.ascontiguousarray(abgr, dtype=np.uint32)
    h, w = abgr.shape
    row = (w * 3 + 3) // 4 * 4
    px = np.zeros((h, row), np.uint8)
    for k, shift in enumerate((16, 8, 0)):                     # B, G, R
        px[:, k:w * 3:3] = (abgr >> shift) & 0xFF
    head = struct.pack("<2sIHHI", b"BM", 54 + row * h, 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, 0, 0, 0, 0, 0)
    with open(path, "wb") as f:
        f.write(head)
        f.write(px.tobytes())


def load_bmp_to_abgr(path):
    """Rows in file order (bottom row first) — the layout ComputeMSE's flip of its first argument assumes."""
    raw = open(path, "rb").read()
    if raw[:2] != b"BM":
        raise ValueError("not a BMP")
    off, = struct.unpack_from("<I", raw, 10)
    w, h, _, bpp, comp = struct.unpack_from("<iiHHI", raw, 18)
    if bpp != 24 or comp != 0 or w <= 0 or h == 0:
        raise ValueError("only uncompressed 24-bit BMP")
    H = abs(h)
    row = (w * 3 + 3) // 4 * 4
    px = np.frombuffer(raw, np.uint8, row * H, off).reshape(H, row)[:, :w * 3].reshape(H, w, 3).astype(np.uint32)
    img = 0xFF000000 | (px[..., 0] << 16) | (px[..., 1] << 8) | px[..., 2]
    return (img[::-1] if h < 0 else img).astype(np.uint32)


def compute_mse(orig, noisy):
    """RGB mean squared error; `orig` is read vertically flipped (MisUtils.cpp:118-147)."""
    o = np.asarray(orig, np.uint32)[::-1]
    n = np.asarray(noisy, np.uint32)
    tot = 0
    for shift in (0, 8, 16):
        d = ((o >> shift) & 0xFF).astype(np.int64) - ((n >> shift) & 0xFF).astype(np.int64)
        tot += int((d * d).sum())
    return tot / float(o.size * 3)


def compute_psnr(mse):
    return float("inf") if mse == 0.0 else 10.0 * np.log10(255.0 * 255.0 / mse)


def _f(x):
    return "%f" % float(np.float32(x))                         # std::to_string(float)


def benchmark_record_name(settings, average_frame_time_ms, render_time_ms, mse=None, psnr=None):
    s = settings
    n = "RenderedImages/output_%s(ms)_%s(min)s_%s" % (_f(average_frame_time_ms), _f(np.float32(render_time_ms) / np.float32(60000.0)),
                                                     TECHNIQUE_NAMES[s.technique])
    if s.technique not in (7, 8):
        n += "_%dsample(s)_%drayBounces(s)" % (s.sample_count, s.light_bounces)
    else:
        n += "_%dcandidate(s)" % s.light_candidate_count
        if s.use_temporal_reuse:
            n += "_temporalHistoryLimit(%d)" % s.temporal_history_limit
        if s.use_spatial_reuse:
            n += "_NeighbourCount(%d)_NeighbourRadius(%d)" % (s.spatial_neighbor_num, s.spatial_neighbor_radius)
        if s.technique == 8:
            n += "_%drayBounces(s)" % s.light_bounces
    if mse is not None:
        n += "_MSE(%s)_PSNR(%s)" % (_f(mse), _f(psnr if psnr is not None else compute_psnr(mse)))
    return n


def timestamped_filename(base, extension=".bmp"):
    return base + "_" + time.strftime("%Y-%m-%d_%H-%M-%S") + extension
