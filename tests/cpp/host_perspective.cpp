// host_perspective.cpp — the C++ host layer's perspective find_ransac (cybervision_amd/csrc/host/cvhip_host.hpp:
// FundamentalMatrix::new(Perspective, max_dimension).find_ransac, fundamentalmatrix.rs:72-147, 231-257) on a real
// GPU, on a match list read from a file.  Writes F and the inlier list for tests/test_host_cpp_gpu.py.
//
// usage: host_perspective matches.bin N max_dimension seed outdir      (matches: 4 x u32 per match)
#include <cstdio>
#include <fstream>

#include "../../cybervision_amd/csrc/host/cvhip_host.hpp"

using namespace cvhip_host;

int main(int argc, char **argv)
{
    if (argc != 6) {
        std::fprintf(stderr, "usage: %s matches.bin N max_dimension seed outdir\n", argv[0]);
        return 2;
    }
    try {
        const size_t n = std::stoul(argv[2]);
        const double max_dimension = std::stod(argv[3]);
        const uint64_t seed = std::stoull(argv[4]);
        const std::string out = argv[5];
        std::vector<uint32_t> flat(4 * n);
        std::ifstream f(argv[1], std::ios::binary);
        f.read(reinterpret_cast<char *>(flat.data()), (std::streamsize)(flat.size() * sizeof(uint32_t)));
        if (!f) throw std::runtime_error("cannot read the match list");
        std::vector<PointMatch> matches(n);
        for (size_t i = 0; i < n; i++) {
            matches[i].first = {flat[4 * i], flat[4 * i + 1]};
            matches[i].second = {flat[4 * i + 2], flat[4 * i + 3]};
        }
        GpuDevice dev = create_gpu_context(HardwareMode::Gpu);
        const FundamentalMatrix fm(ProjectionMode::Perspective, max_dimension);
        const FundamentalMatrixResult fr = fm.find_ransac(dev, matches, seed);
        std::ofstream ff(out + "/f_persp.bin", std::ios::binary);
        ff.write(reinterpret_cast<const char *>(fr.f.data()), (std::streamsize)(9 * sizeof(double)));
        std::vector<uint32_t> inl;
        for (const auto &m : fr.inliers) {
            inl.push_back((uint32_t)m.first.x);
            inl.push_back((uint32_t)m.first.y);
            inl.push_back((uint32_t)m.second.x);
            inl.push_back((uint32_t)m.second.y);
        }
        std::ofstream fi(out + "/inliers_persp.bin", std::ios::binary);
        fi.write(reinterpret_cast<const char *>(inl.data()), (std::streamsize)(inl.size() * sizeof(uint32_t)));
        std::printf("{\"matches\": %zu, \"inliers\": %zu}\n", n, fr.inliers.size());
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "host_perspective: %s\n", e.what());
        return 1;
    }
}
