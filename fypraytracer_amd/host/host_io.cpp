// C-callable face of the host-side asset readers (TextureIO.h) for the Python mirror (fypraytracer_amd/texture.py):
// libfyprt_host.so, host code only — no GPU, no HIP.
#include "TextureIO.h"
extern "C" {
// Two calls: with pixels == nullptr it decodes, caches and reports the size; with a buffer of width*height words it copies.
int fyprt_host_load_png(const char* path, uint32_t* width, uint32_t* height, uint32_t* pixels, char* err, size_t errLen) {
    static thread_local fyprt_host::TextureImage img; static thread_local std::string cached;
    if (!pixels || cached != path) {
        cached.clear();
        if (!fyprt_host::LoadTexturePNG(path, img)) { if (err && errLen) { std::snprintf(err, errLen, "%s", img.error.c_str()); } return 1; }
        cached = path;
    }
    if (width) *width = img.width;
    if (height) *height = img.height;
    if (pixels) std::memcpy(pixels, img.pixels.data(), img.pixels.size() * 4);
    return 0;
}
}
