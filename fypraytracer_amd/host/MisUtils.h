// Benchmark / quality logging of the reference's app layer, header-only, host code:
//   MisUtils::{SaveABGRToBMP, GetTimestampedFilename, ComputeMSE, ComputePSNR}   Utility/MisUtils.cpp:13-157
//   the record (file-name) format of MainLayer::SaveRenderImage / SaveBenchmarkResults   WalnutApp.cpp:787-875
// Same names, argument meaning and results; LoadBMPToABGR is this build's reader for the reference image the app
// picks through a file dialog (stb_image there; rows are kept in FILE order, bottom row first, which is what
// ComputeMSE's flip of its first argument assumes, MisUtils.cpp:128).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <fstream>
#include <string>
#include <vector>
#include "Renderer.h"

namespace fyprt_host {
namespace MisUtils {

inline bool SaveABGRToBMP(const std::string& filename, const uint32_t* abgrPixels, int width, int height) {
    const uint32_t row = ((uint32_t)width * 3u + 3u) / 4u * 4u, size = 54u + row * (uint32_t)height;
    std::vector<uint8_t> f(size, 0);
    const uint32_t off = 54, dib = 40, w = (uint32_t)width, h = (uint32_t)height; const uint16_t planes = 1, bpp = 24;
    f[0] = 'B'; f[1] = 'M';
    std::memcpy(&f[2], &size, 4); std::memcpy(&f[10], &off, 4); std::memcpy(&f[14], &dib, 4);
    std::memcpy(&f[18], &w, 4); std::memcpy(&f[22], &h, 4); std::memcpy(&f[26], &planes, 2); std::memcpy(&f[28], &bpp, 2);
    for (uint32_t y = 0; y < h; ++y)                       // render row 0 (NDC y = -1) is the first, i.e. bottom, BMP row
        for (uint32_t x = 0; x < w; ++x) {
            const uint32_t p = abgrPixels[(size_t)y * w + x]; uint8_t* q = &f[54 + (size_t)y * row + (size_t)x * 3];
            q[0] = (uint8_t)(p >> 16); q[1] = (uint8_t)(p >> 8); q[2] = (uint8_t)p;
        }
    std::ofstream file(filename, std::ios::binary);
    if (!file) return false;
    file.write((const char*)f.data(), (std::streamsize)f.size());
    return (bool)file;
}

// 24-bit uncompressed BMP -> ABGR words, alpha 255; rows in file order when the height field is positive.
inline bool LoadBMPToABGR(const std::string& filename, std::vector<uint32_t>& out, uint32_t& width, uint32_t& height) {
    std::ifstream file(filename, std::ios::binary);
    if (!file) return false;
    std::vector<uint8_t> f((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());
    if (f.size() < 54 || f[0] != 'B' || f[1] != 'M') return false;
    uint32_t off; int32_t w, h; uint16_t bpp; uint32_t comp;
    std::memcpy(&off, &f[10], 4); std::memcpy(&w, &f[18], 4); std::memcpy(&h, &f[22], 4); std::memcpy(&bpp, &f[28], 2); std::memcpy(&comp, &f[30], 4);
    if (bpp != 24 || comp != 0 || w <= 0 || h == 0) return false;
    const bool topDown = h < 0; const uint32_t H = (uint32_t)(topDown ? -h : h), W = (uint32_t)w, row = (W * 3u + 3u) / 4u * 4u;
    if ((size_t)off + (size_t)row * H > f.size()) return false;
    out.resize((size_t)W * H); width = W; height = H;
    for (uint32_t y = 0; y < H; ++y)
        for (uint32_t x = 0; x < W; ++x) {
            const uint8_t* q = &f[off + (size_t)y * row + (size_t)x * 3];
            out[(size_t)(topDown ? H - 1 - y : y) * W + x] = 0xFF000000u | ((uint32_t)q[0] << 16) | ((uint32_t)q[1] << 8) | q[2];
        }
    return true;
}

inline std::string GetTimestampedFilename(const std::string& baseName, const std::string& extension = ".bmp") {
    const std::time_t now = std::time(nullptr);
    std::tm tm{}; localtime_r(&now, &tm);
    char buf[32]; std::strftime(buf, sizeof buf, "%Y-%m-%d_%H-%M-%S", &tm);
    return baseName + "_" + buf + extension;
}

// RGB mean squared error; `orig` is read vertically flipped (MisUtils.cpp:128: the reference image comes from a file).
inline double ComputeMSE(const uint32_t* orig, const uint32_t* noisy, uint32_t width, uint32_t height) {
    double mse = 0.0;
    for (uint32_t x = 0; x < width; ++x)                   // the reference's loop order: the double sum is order dependent
        for (uint32_t y = 0; y < height; ++y) {
            const uint32_t p0 = orig[x + (size_t)(height - 1 - y) * width], p1 = noisy[x + (size_t)y * width];
            for (int s = 0; s < 24; s += 8) { const int d = (int)((p0 >> s) & 0xFF) - (int)((p1 >> s) & 0xFF); mse += d * d; }
        }
    return mse / double(uint64_t(width) * uint64_t(height) * 3);
}
inline double ComputePSNR(double mse) { return mse == 0.0 ? (double)INFINITY : 10.0 * std::log10((255.0 * 255.0) / mse); }

inline const char* SamplingTechniqueName(int t) {          // WalnutApp.cpp samplingTechniqueNames, enum order of SamplingTechniqueEnum.h:6-14
    static const char* n[9] = {"BRUTE_FORCE", "UNIFORM_SAMPLING", "COSINE_WEIGHTED_SAMPLING", "GGX_SAMPLING", "BRDF_SAMPLING",
                               "LIGHT_SOURCE_SAMPLING", "NEE", "RESTIR_DI", "RESTIR_GI"};
    return (t >= 0 && t < 9) ? n[t] : "UNKNOWN";
}

// The record name both save paths build (WalnutApp.cpp:787-823, :833-875) before the time stamp is appended;
// std::to_string formatting (%f for floats).  `mse`/`psnr` are appended iff `withQuality` (they pass through float there).
inline std::string BenchmarkRecordName(const RenderingSettings& s, float averageFrameTimeMs, float renderTimeMs, bool withQuality = false,
                                       double mse = 0.0, double psnr = 0.0) {
    std::string n = "RenderedImages/output";
    n += "_" + std::to_string(averageFrameTimeMs) + "(ms)";
    n += "_" + std::to_string(renderTimeMs / 60000.0f) + "(min)s";
    const int t = (int)s.currentSamplingTechnique;
    n += std::string("_") + SamplingTechniqueName(t);
    if (t != RESTIR_DI && t != RESTIR_GI) {
        n += "_" + std::to_string(s.sampleCount) + "sample(s)";
        n += "_" + std::to_string(s.lightBounces) + "rayBounces(s)";
    } else {
        n += "_" + std::to_string(s.lightCandidateCount) + "candidate(s)";
        if (s.useTemporalReuse) n += "_temporalHistoryLimit(" + std::to_string(s.temporalHistoryLimit) + ")";
        if (s.useSpatialReuse) {
            n += "_NeighbourCount(" + std::to_string(s.spatialNeighborNum) + ")";
            n += "_NeighbourRadius(" + std::to_string(s.spatialNeighborRadius) + ")";
        }
        if (t == RESTIR_GI) n += "_" + std::to_string(s.lightBounces) + "rayBounces(s)";
    }
    if (withQuality) {
        n += "_MSE(" + std::to_string((float)mse) + ")";
        n += "_PSNR(" + std::to_string((float)psnr) + ")";
    }
    return n;
}

}  // namespace MisUtils
}  // namespace fyprt_host
