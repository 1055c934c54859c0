// Texture ingest of the reference's host layer: Texture::Texture(std::string&) (Texture.cu:8-40) loads an
// image with stb_image forced to 4 channels and packs each pixel as A<<24 | B<<16 | G<<8 | R, rows top to bottom, no flip.
// This is a self-contained PNG reader with that contract (zlib for inflate; stb_image is not in the image):
// non-interlaced PNG, bit depth 8 or 16 (16 keeps the high byte, as stb's 8-bit conversion does), colour types
// grey / RGB / palette (+ tRNS alpha) / grey+alpha / RGBA; grey is replicated to RGB, missing alpha is 255.
#pragma once
#include <zlib.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace fyprt_host {

struct TextureImage { uint32_t width = 0, height = 0; std::vector<uint32_t> pixels; std::string fileName; std::string error; };

namespace detail {
inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline int paeth(int a, int b, int c) { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
}  // namespace detail

inline bool DecodePNG(const uint8_t* file, size_t size, TextureImage& out) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    out = TextureImage();
    if (size < 8 || std::memcmp(file, sig, 8) != 0) { out.error = "not a PNG"; return false; }
    uint32_t W = 0, H = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    for (size_t p = 8; p + 12 <= size;) {
        const uint32_t len = detail::be32(file + p); const uint8_t* type = file + p + 4; const uint8_t* data = file + p + 8;
        if (p + 12 + (size_t)len > size) { out.error = "truncated chunk"; return false; }
        if (crc32(crc32(0L, Z_NULL, 0), type, len + 4) != detail::be32(data + len)) { out.error = "chunk CRC mismatch"; return false; }
        if (!std::memcmp(type, "IHDR", 4) && len == 13) { W = detail::be32(data); H = detail::be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12]; }
        else if (!std::memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!std::memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        p += 12 + (size_t)len;
    }
    if (W == 0 || H == 0 || W > (1u << 15) || H > (1u << 15)) { out.error = "bad IHDR"; return false; }
    if (interlace != 0) { out.error = "interlaced PNG is not supported"; return false; }
    if (!(depth == 8 || depth == 16) || (ctype == 3 && depth != 8)) { out.error = "unsupported bit depth"; return false; }
    int channels;
    switch (ctype) { case 0: channels = 1; break; case 2: channels = 3; break; case 3: channels = 1; break; case 4: channels = 2; break; case 6: channels = 4; break;
                     default: out.error = "bad colour type"; return false; }
    const size_t bpp = (size_t)channels * (depth / 8), stride = (size_t)W * bpp;
    std::vector<uint8_t> raw((stride + 1) * H);
    uLongf rawLen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size()) { out.error = "inflate failed"; return false; }
    // undo the per-row filters in place (PNG spec 9.2): 0 None, 1 Sub, 2 Up, 3 Average, 4 Paeth
    std::vector<uint8_t> zero(stride, 0);
    for (uint32_t y = 0; y < H; ++y) {
        uint8_t* row = raw.data() + (size_t)y * (stride + 1) + 1; const uint8_t ft = row[-1];
        const uint8_t* up = y ? row - (stride + 1) : zero.data();
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? row[i - bpp] : 0, b = up[i], c = i >= bpp ? up[i - bpp] : 0;
            int v = row[i];
            switch (ft) { case 0: break; case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) >> 1; break; case 4: v += detail::paeth(a, b, c); break;
                          default: out.error = "bad filter type"; return false; }
            row[i] = (uint8_t)v;
        }
    }
    out.width = W; out.height = H; out.pixels.resize((size_t)W * H);
    const size_t step = depth / 8;                                  // 16-bit samples: big-endian, the high byte comes first
    for (uint32_t y = 0; y < H; ++y) {
        const uint8_t* row = raw.data() + (size_t)y * (stride + 1) + 1;
        for (uint32_t x = 0; x < W; ++x) {
            const uint8_t* s = row + (size_t)x * bpp; uint32_t R, G, B, A = 255;
            switch (ctype) {
                case 0: R = G = B = s[0]; break;
                case 2: R = s[0]; G = s[step]; B = s[2 * step]; break;
                case 3: {
                    const size_t k = s[0];
                    if (3 * k + 2 < plte.size()) { R = plte[3 * k]; G = plte[3 * k + 1]; B = plte[3 * k + 2]; } else { R = G = B = 0; }
                    if (k < trns.size()) A = trns[k];
                    break;
                }
                case 4: R = G = B = s[0]; A = s[step]; break;
                default: R = s[0]; G = s[step]; B = s[2 * step]; A = s[3 * step]; break;
            }
            out.pixels[(size_t)y * W + x] = (A << 24) | (B << 16) | (G << 8) | R;      // Texture.cu:30
        }
    }
    return true;
}

// Texture::Texture(std::string& imageFilePath): false + `error` (and a message on stderr, as the reference prints) on failure
inline bool LoadTexturePNG(const std::string& path, TextureImage& out) {
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) { out = TextureImage(); out.error = "cannot open " + path; std::fprintf(stderr, "Failed to load image: %s\n", path.c_str()); return false; }
    std::vector<uint8_t> buf; uint8_t tmp[1 << 16]; size_t n;
    while ((n = std::fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    std::fclose(f);
    const bool ok = DecodePNG(buf.data(), buf.size(), out);
    if (!ok) { std::fprintf(stderr, "Failed to load image: %s (%s)\n", path.c_str(), out.error.c_str()); return false; }
    const size_t slash = path.find_last_of("/\\");
    out.fileName = slash == std::string::npos ? path : path.substr(slash + 1);      // Texture.cu:37-39
    return true;
}

}  // namespace fyprt_host
