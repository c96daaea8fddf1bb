// host_pipeline.cpp — exercises the C++ host layer (cybervision_amd/csrc/host/cvhip_host.hpp) end to end
// on a real GPU: ORB on both images -> KeypointMatching -> FundamentalMatrix::find_ransac (affine) ->
// PointCorrelations over a box pyramid.  Writes raw results that tests/test_host_cpp_gpu.py compares
// with the ctypes path (bit-exact) and with the known geometry.
//
// usage: host_pipeline img1.raw img2.raw W H outdir
#include <cstdio>
#include <fstream>
#include <iostream>

#include "../../cybervision_amd/csrc/host/cvhip_host.hpp"

using namespace cvhip_host;

static Grid<uint8_t> read_raw(const char *path, size_t w, size_t h)
{
    Grid<uint8_t> g(w, h, 0);
    std::ifstream f(path, std::ios::binary);
    f.read(reinterpret_cast<char *>(g.data()), (std::streamsize)(w * h));
    if (!f) throw std::runtime_error(std::string("cannot read ") + path);
    return g;
}
static Grid<uint8_t> box_downsample(const Grid<uint8_t> &s)
{
    Grid<uint8_t> d(s.width() / 2, s.height() / 2, 0);
    for (size_t y = 0; y < d.height(); y++)
        for (size_t x = 0; x < d.width(); x++) {
            const unsigned v = s.val(2 * x, 2 * y) + s.val(2 * x + 1, 2 * y) + s.val(2 * x, 2 * y + 1) + s.val(2 * x + 1, 2 * y + 1);
            d.val_mut(x, y) = (uint8_t)((v + 2) >> 2);
        }
    return d;
}
template <typename T> static void write_vec(const std::string &path, const std::vector<T> &v)
{
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char *>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
}

int main(int argc, char **argv)
{
    if (argc != 6) {
        std::fprintf(stderr, "usage: %s img1.raw img2.raw W H outdir\n", argv[0]);
        return 2;
    }
    try {
        const size_t w = std::stoul(argv[3]), h = std::stoul(argv[4]);
        const std::string out = argv[5];
        const Grid<uint8_t> img1 = read_raw(argv[1], w, h), img2 = read_raw(argv[2], w, h);
        GpuDevice dev = create_gpu_context(HardwareMode::Gpu);

        // sparse stage: reconstruction.rs:400-526 for one pyramid level
        const auto kp1 = orb::extract_points(dev, img1), kp2 = orb::extract_points(dev, img2);
        std::vector<uint32_t> kp_flat;
        for (const auto &k : kp1) {
            kp_flat.push_back((uint32_t)k.first.x);
            kp_flat.push_back((uint32_t)k.first.y);
            kp_flat.insert(kp_flat.end(), k.second.begin(), k.second.end());
        }
        write_vec(out + "/kp1.bin", kp_flat);
        const KeypointMatching matching(dev, kp1, kp2, ProjectionMode::Affine);
        std::vector<uint32_t> m_flat;
        for (const auto &m : matching.matches) {
            m_flat.push_back((uint32_t)m.first.x);
            m_flat.push_back((uint32_t)m.first.y);
            m_flat.push_back((uint32_t)m.second.x);
            m_flat.push_back((uint32_t)m.second.y);
        }
        write_vec(out + "/matches.bin", m_flat);
        const FundamentalMatrix fm(ProjectionMode::Affine, (double)std::max(w, h));
        const FundamentalMatrixResult fr = fm.find_ransac(dev, matching.matches, 12345);
        write_vec(out + "/f.bin", std::vector<double>(fr.f.begin(), fr.f.end()));

        // dense stage: reconstruction.rs:554-588 with the known horizontal geometry
        const size_t steps = PointCorrelations::optimal_scale_steps({(uint32_t)w, (uint32_t)h});
        std::vector<Grid<uint8_t>> p1{img1}, p2{img2};
        for (size_t i = 0; i < steps; i++) {
            p1.push_back(box_downsample(p1.back()));
            p2.push_back(box_downsample(p2.back()));
        }
        const std::array<double, 9> f_h = {0, 0, 0, 0, 0, 1, 0, -1, 0};
        const auto grid = correlate_dense(dev, p1, p2, f_h, ProjectionMode::Affine);
        std::vector<int32_t> xy(2 * w * h, -1);
        std::vector<float> corr(w * h, 0.0f);
        size_t valid = 0;
        for (size_t i = 0; i < w * h; i++)
            if (grid.data()[i]) {
                xy[2 * i] = (int32_t)grid.data()[i]->first.x;
                xy[2 * i + 1] = (int32_t)grid.data()[i]->first.y;
                corr[i] = grid.data()[i]->second;
                valid++;
            }
        write_vec(out + "/dense_xy.bin", xy);
        write_vec(out + "/dense_corr.bin", corr);
        std::printf("{\"device\": \"%s\", \"keypoints1\": %zu, \"keypoints2\": %zu, \"matches\": %zu, \"inliers\": %zu, "
                    "\"dense_valid\": %zu, \"levels\": %zu}\n",
                    dev.name().c_str(), kp1.size(), kp2.size(), matching.matches.size(), fr.inliers.size(), valid,
                    steps + 1);
        // error behaviour: too few matches for the perspective model (207) surface as RansacError with the
        // reference's text, a bad image as GpuError
        bool threw = false;
        try {
            std::vector<PointMatch> few(matching.matches.begin(), matching.matches.begin() + 50);
            FundamentalMatrix(ProjectionMode::Perspective, 1000.0).find_ransac(dev, few, 1);
        } catch (const RansacError &e) {
            threw = std::string(e.what()).find("Not enough matches") != std::string::npos;
        }
        if (!threw) throw std::runtime_error("perspective find_ransac did not report 'Not enough matches'");
        threw = false;
        try {
            PointCorrelations bad(dev, {4, 4}, {4, 4}, f_h, ProjectionMode::Affine);
        } catch (const GpuError &e) {
            threw = e.code == CVHIP_ERR_INVALID;
        }
        if (!threw) throw std::runtime_error("tiny image did not raise GpuError");
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "host_pipeline: %s\n", e.what());
        return 1;
    }
}
