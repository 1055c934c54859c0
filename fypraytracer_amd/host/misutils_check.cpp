// Small command-line face of MisUtils.h for tests/test_misutils.py (host-only, no GPU):
//   misutils_check save W H out.bmp      deterministic ABGR pattern p(x,y) = pcg-free integer mix, written with SaveABGRToBMP
//   misutils_check mse ref.bmp img.bmp   loads both (file row order), prints ComputeMSE(ref, img) and ComputePSNR
//   misutils_check name TECH             prints BenchmarkRecordName for default settings with that technique (+ quality fields)
#include <cstdlib>
#include "MisUtils.h"
using namespace fyprt_host;
int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const std::string cmd = argv[1];
    if (cmd == "save" && argc == 5) {
        const uint32_t W = std::atoi(argv[2]), H = std::atoi(argv[3]);
        std::vector<uint32_t> px((size_t)W * H);
        for (uint32_t y = 0; y < H; ++y) for (uint32_t x = 0; x < W; ++x) px[(size_t)y * W + x] = 0xFF000000u | (((x * 7u + y * 13u) & 0xFF) << 16) | (((x ^ y) & 0xFF) << 8) | ((x * y) & 0xFF);
        return MisUtils::SaveABGRToBMP(argv[4], px.data(), (int)W, (int)H) ? 0 : 1;
    }
    if (cmd == "mse" && argc == 4) {
        std::vector<uint32_t> a, b; uint32_t w, h, w2, h2;
        if (!MisUtils::LoadBMPToABGR(argv[2], a, w, h) || !MisUtils::LoadBMPToABGR(argv[3], b, w2, h2) || w != w2 || h != h2) return 1;
        const double mse = MisUtils::ComputeMSE(a.data(), b.data(), w, h);
        std::printf("%.17g %.17g\n", mse, MisUtils::ComputePSNR(mse));
        return 0;
    }
    if (cmd == "name" && argc == 3) {
        RenderingSettings s; s.currentSamplingTechnique = (SamplingTechniqueEnum)std::atoi(argv[2]);
        std::printf("%s\n%s\n", MisUtils::BenchmarkRecordName(s, 1.3371f, 66.855f).c_str(),
                    MisUtils::BenchmarkRecordName(s, 1.3371f, 66.855f, true, 12.3456789, MisUtils::ComputePSNR(12.3456789)).c_str());
        return 0;
    }
    return 2;
}
