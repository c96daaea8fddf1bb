// cvhip_host.hpp — C++ host layer above the C ABI (include/cvhip.h).
//
// The reference's host code is Rust; its toolchain is absent here, so this header is the compiled
// host side that mirrors the reference's interface for the accelerated path — same names,
// argument meaning and error behaviour — and is what a C++ consumer (or a cxx-bridge) would use:
//
//   create_gpu_context(HardwareMode)                       src/correlation/mod.rs:145-147
//   PointCorrelations::{new, correlate_images, complete,   src/correlation/mod.rs:150-245, 542-550
//                       optimal_scale_steps, get_selected_hardware}, .correlated_points
//   orb::extract_points, orb::optimal_scale_steps          src/orb.rs:50-84, 407-415
//   KeypointMatching::new(...).matches                     src/pointmatching.rs:29-77
//   FundamentalMatrix::{new, find_ransac}                  src/fundamentalmatrix.rs:72-147
//
// Header-only, no dependencies beyond the C ABI and the C++17 standard library.  Nothing here does
// arithmetic on the dense/ORB/matcher path: it only calls libcvhip.so.  RANSAC keeps the reference's
// split: hypotheses are sampled and fitted on the host (as fundamentalmatrix.rs:155-190, 260-286
// do), every batch is scored on the GPU (cvhip_ransac_score), the best is chosen with the reference's
// ordering (:623-649).  Errors surface as exceptions carrying cvhip_last_error(), the analogue of
// Result<_, GpuError>; a failed GpuDevice construction is what makes the reference fall back to its
// CPU path (mod.rs:170-173) — here it simply throws, there is no CPU fallback in this library.
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <optional>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/cvhip.h"

namespace cvhip_host {

struct GpuError : std::runtime_error { // GpuError (correlation/gpu/vulkan.rs:1204-1272)
    int code;
    GpuError(int c, const std::string &where)
        : std::runtime_error(where + ": " + cvhip_last_error()), code(c) {}
};
inline void check(int rc, const char *where)
{
    if (rc != CVHIP_OK) throw GpuError(rc, where);
}

enum class HardwareMode { Gpu, GpuLowPower, Cpu }; // correlation/mod.rs:49-54
enum class ProjectionMode { Affine, Perspective }; // correlation/mod.rs:43-47

template <typename T> struct Point2D { // data.rs:8-19
    T x, y;
};

template <typename T> class Grid { // data.rs:22-64: row-major, index = width*y + x
  public:
    Grid() = default;
    Grid(size_t width, size_t height, T v) : width_(width), height_(height), data_(width * height, v) {}
    size_t width() const { return width_; }
    size_t height() const { return height_; }
    const T &val(size_t x, size_t y) const { return data_.at(width_ * y + x); }
    T &val_mut(size_t x, size_t y) { return data_.at(width_ * y + x); }
    const T *data() const { return data_.data(); }
    T *data() { return data_.data(); }

  private:
    size_t width_ = 0, height_ = 0;
    std::vector<T> data_;
};

using Match = std::pair<Point2D<uint32_t>, float>; // correlation/mod.rs:33

class GpuDevice { // GpuDevice = DefaultDeviceContext (correlation/mod.rs:35)
  public:
    explicit GpuDevice(HardwareMode mode, int ordinal = -1)
    {
        if (mode == HardwareMode::Cpu) throw std::invalid_argument("HardwareMode::Cpu has no GPU device");
        check(cvhip_device_create(mode == HardwareMode::GpuLowPower ? 1 : 0, ordinal, &dev_), "cvhip_device_create");
    }
    ~GpuDevice() { cvhip_device_destroy(dev_); }
    GpuDevice(const GpuDevice &) = delete;
    GpuDevice &operator=(const GpuDevice &) = delete;
    cvhip_device *handle() const { return dev_; }
    std::string name() const { return cvhip_device_name(dev_); }

  private:
    cvhip_device *dev_ = nullptr;
};
inline GpuDevice create_gpu_context(HardwareMode mode) { return GpuDevice(mode); } // mod.rs:145-147 (guaranteed elision)

class PointCorrelations { // correlation/mod.rs:63-245, GPU branch
  public:
    Grid<std::optional<Match>> correlated_points;

    PointCorrelations(GpuDevice &dev, std::pair<uint32_t, uint32_t> img1_dimensions,
                      std::pair<uint32_t, uint32_t> img2_dimensions, const std::array<double, 9> &fundamental_matrix,
                      ProjectionMode projection_mode)
        : dims1_(img1_dimensions), selected_hardware_("GPU " + dev.name())
    {
        check(cvhip_ctx_create(dev.handle(), img1_dimensions.first, img1_dimensions.second, img2_dimensions.first,
                               img2_dimensions.second, projection_mode == ProjectionMode::Perspective ? 1 : 0,
                               fundamental_matrix.data(), &ctx_),
              "cvhip_ctx_create");
        // complete() always lands in host memory: the last level goes out in row bands, each crossing PCIe under the
        // search of the next (same grid)
        check(cvhip_ctx_set_result_bands(ctx_, 0), "cvhip_ctx_set_result_bands"); // (0: the library's choice by size)
    }
    ~PointCorrelations() { cvhip_ctx_destroy(ctx_); }
    PointCorrelations(const PointCorrelations &) = delete;
    PointCorrelations &operator=(const PointCorrelations &) = delete;

    const std::string &get_selected_hardware() const { return selected_hardware_; }

    // correlate_images (mod.rs:217-245): forward, reverse, cross-check x2, first_pass = false
    void correlate_images(const Grid<uint8_t> &img1, const Grid<uint8_t> &img2, float scale)
    {
        check(cvhip_correlate_level(ctx_, img1.data(), (uint32_t)img1.width(), (uint32_t)img1.height(), img2.data(),
                                    (uint32_t)img2.width(), (uint32_t)img2.height(), scale, first_pass_ ? 1 : 0,
                                    nullptr, nullptr),
              "cvhip_correlate_level");
        first_pass_ = false;
    }

    // complete (mod.rs:208-215): pulls the forward grid into `correlated_points`
    void complete()
    {
        const size_t w = dims1_.first, h = dims1_.second;
        std::vector<uint32_t> cells(w * h); // y2 << 16 | x2, all ones = None: 8 bytes per cell over PCIe with the score
        std::vector<float> corr(w * h);
        check(cvhip_complete_packed(ctx_, 0, cells.data(), corr.data()), "cvhip_complete_packed");
        correlated_points = Grid<std::optional<Match>>(w, h, std::nullopt);
        for (size_t i = 0; i < w * h; i++)
            if (cells[i] != 0xFFFFFFFFu) correlated_points.data()[i] = Match{{cells[i] & 0xFFFFu, cells[i] >> 16}, corr[i]};
    }

    static size_t optimal_scale_steps(std::pair<uint32_t, uint32_t> dimensions) // mod.rs:542-550
    {
        const size_t min_dimension = std::min(dimensions.first, dimensions.second);
        if (min_dimension <= 64) return 0;
        return (size_t)std::floor(std::log2((double)min_dimension / 64.0));
    }

  private:
    cvhip_ctx *ctx_ = nullptr;
    std::pair<uint32_t, uint32_t> dims1_;
    bool first_pass_ = true;
    std::string selected_hardware_;
};

// The dense stage of ImageReconstruction::correlate_dense (reconstruction.rs:554-588) over prebuilt
// pyramids: pyr[k] is the 1/2^k image; levels run coarse to fine.
inline Grid<std::optional<Match>> correlate_dense(GpuDevice &dev, const std::vector<Grid<uint8_t>> &pyr1,
                                                  const std::vector<Grid<uint8_t>> &pyr2,
                                                  const std::array<double, 9> &f, ProjectionMode mode)
{
    PointCorrelations pc(dev, {(uint32_t)pyr1[0].width(), (uint32_t)pyr1[0].height()},
                         {(uint32_t)pyr2[0].width(), (uint32_t)pyr2[0].height()}, f, mode);
    const size_t steps = pyr1.size() - 1;
    for (size_t i = 0; i <= steps; i++) {
        const size_t k = steps - i;
        pc.correlate_images(pyr1[k], pyr2[k], 1.0f / (float)(1u << k));
    }
    pc.complete();
    return std::move(pc.correlated_points);
}

namespace orb {
using Keypoint = std::pair<Point2D<size_t>, std::array<uint32_t, 8>>; // orb.rs:9
constexpr uint32_t MAX_KEYPOINTS = 10000;                            // orb.rs:41

// orb.rs:43-48: `Option<&PL>` becomes a nullable pointer; report_status is called on the calling thread
struct ProgressListener {
    virtual void report_status(float pos) const = 0;
    virtual ~ProgressListener() = default;
};
inline std::vector<Keypoint> extract_points(GpuDevice &dev, const Grid<uint8_t> &img,
                                            const ProgressListener *progress_listener = nullptr) // orb.rs:50-84
{
    std::vector<uint32_t> xy(2 * MAX_KEYPOINTS), desc(8 * MAX_KEYPOINTS);
    uint32_t n = 0;
    const auto thunk = [](void *user, float pos) { static_cast<const ProgressListener *>(user)->report_status(pos); };
    check(cvhip_orb_extract(dev.handle(), img.data(), (uint32_t)img.width(), (uint32_t)img.height(), MAX_KEYPOINTS,
                            xy.data(), desc.data(), &n, progress_listener ? +thunk : nullptr,
                            const_cast<ProgressListener *>(progress_listener)),
          "cvhip_orb_extract");
    std::vector<Keypoint> out(n);
    for (uint32_t i = 0; i < n; i++) {
        out[i].first = {xy[2 * i], xy[2 * i + 1]};
        std::copy_n(&desc[8 * i], 8, out[i].second.begin());
    }
    return out;
}
inline size_t optimal_scale_steps(std::pair<uint32_t, uint32_t> dimensions) // orb.rs:407-415
{
    const size_t min_dimension = std::min(dimensions.first, dimensions.second);
    if (min_dimension <= 256) return 0;
    return (size_t)std::floor(std::log2((double)min_dimension / 256.0));
}
} // namespace orb

using PointMatch = std::pair<Point2D<size_t>, Point2D<size_t>>; // fundamentalmatrix.rs:33, pointmatching.rs

struct KeypointMatching { // pointmatching.rs:17-77
    std::vector<PointMatch> matches;
    KeypointMatching(GpuDevice &dev, const std::vector<orb::Keypoint> &points1,
                     const std::vector<orb::Keypoint> &points2, ProjectionMode mode)
    {
        const uint32_t threshold = mode == ProjectionMode::Affine ? 32 : 48; // pointmatching.rs:8-9
        auto flatten = [](const std::vector<orb::Keypoint> &kp, std::vector<uint32_t> &xy, std::vector<uint32_t> &d) {
            xy.resize(2 * kp.size());
            d.resize(8 * kp.size());
            for (size_t i = 0; i < kp.size(); i++) {
                xy[2 * i] = (uint32_t)kp[i].first.x;
                xy[2 * i + 1] = (uint32_t)kp[i].first.y;
                std::copy(kp[i].second.begin(), kp[i].second.end(), &d[8 * i]);
            }
        };
        std::vector<uint32_t> xy1, d1, xy2, d2;
        flatten(points1, xy1, d1);
        flatten(points2, xy2, d2);
        std::vector<uint32_t> out(4 * std::max<size_t>(points1.size(), 1));
        uint32_t n = 0;
        check(cvhip_match_points(dev.handle(), xy1.data(), d1.data(), (uint32_t)points1.size(), xy2.data(), d2.data(),
                                 (uint32_t)points2.size(), threshold, out.data(), nullptr, &n),
              "cvhip_match_points");
        matches.resize(n);
        for (uint32_t i = 0; i < n; i++)
            matches[i] = {{out[4 * i], out[4 * i + 1]}, {out[4 * i + 2], out[4 * i + 3]}};
    }
};

struct RansacError : std::runtime_error { // fundamentalmatrix.rs:665-682
    using std::runtime_error::runtime_error;
};

struct FundamentalMatrixResult { // fundamentalmatrix.rs:57-61
    std::array<double, 9> f;     // row-major
    std::vector<PointMatch> inliers;
};

// FundamentalMatrix (fundamentalmatrix.rs:63-257) for the affine model.  Sampling and the 4-point fit
// run on the host exactly as in the reference (rejection sampling from the top 5000 matches, >= 10 px
// apart; mean-centred 4x4 smallest right-singular vector); all hypotheses of one check interval are
// scored on the GPU in one call.  The perspective model (7-point, fundamentalmatrix.rs:289-389) runs
// entirely on the device (cvhip_ransac_perspective); its final LM refit (:391-426, 515-621) is
// cvhip_optimize_perspective_f.
class FundamentalMatrix {
  public:
    FundamentalMatrix(ProjectionMode projection, double max_dimension) : projection_(projection), max_dimension_(max_dimension)
    {
        // fundamentalmatrix.rs:16-30, 72-101
        ransac_k_ = 1000000;
        if (projection == ProjectionMode::Affine) {
            ransac_n_ = 4;
            ransac_t_ = 0.1;
            ransac_d_ = 10;
            ransac_d_early_exit_ = 1000;
        } else {
            ransac_n_ = 7;
            ransac_t_ = 10.0 / 1000.0 * max_dimension;
            ransac_d_ = 200;
            ransac_d_early_exit_ = 50000;
        }
    }

    FundamentalMatrixResult find_ransac(GpuDevice &dev, const std::vector<PointMatch> &point_matches,
                                        uint64_t seed = std::random_device{}()) const
    {
        if (point_matches.size() < ransac_d_ + ransac_n_) throw RansacError("Not enough matches");
        std::vector<uint32_t> flat(4 * point_matches.size());
        for (size_t i = 0; i < point_matches.size(); i++) {
            flat[4 * i] = (uint32_t)point_matches[i].first.x;
            flat[4 * i + 1] = (uint32_t)point_matches[i].first.y;
            flat[4 * i + 2] = (uint32_t)point_matches[i].second.x;
            flat[4 * i + 3] = (uint32_t)point_matches[i].second.y;
        }
        if (projection_ != ProjectionMode::Affine) {
            // Perspective: sampling, the 7-point model and its checks, scoring and best-pick all run on the device
            // (cvhip_ransac_perspective); optimize_result (:231-257) then refits the winner on its inliers with the
            // reference's own LM loop (cvhip_optimize_perspective_f, host arithmetic) and re-selects the inliers.
            FundamentalMatrixResult res;
            std::vector<uint8_t> mask(point_matches.size());
            uint32_t cnt = 0;
            const int rc = cvhip_ransac_perspective(dev.handle(), flat.data(), (uint32_t)point_matches.size(), max_dimension_,
                                                    seed, 0, res.f.data(), &cnt, mask.data());
            if (rc == CVHIP_ERR_NO_MODEL) throw RansacError(cvhip_last_error());
            check(rc, "cvhip_ransac_perspective");
            std::vector<uint32_t> inl;
            inl.reserve(4 * (size_t)cnt);
            for (size_t i = 0; i < point_matches.size(); i++)
                if (mask[i]) inl.insert(inl.end(), flat.begin() + 4 * i, flat.begin() + 4 * i + 4);
            std::array<double, 9> refit{};
            int refined = 0;
            check(cvhip_optimize_perspective_f(res.f.data(), inl.data(), (uint32_t)(inl.size() / 4), refit.data(), &refined),
                  "cvhip_optimize_perspective_f");
            res.f = refit; // .unwrap_or(res.f): the call returns F itself when the refit is rejected
            for (size_t i = 0; i < point_matches.size(); i++)
                if (fits(res.f, point_matches[i])) res.inliers.push_back(point_matches[i]); // :248-254
            return res;
        }
        std::mt19937_64 rng(seed); // the reference seeds SmallRng from the OS: runs are not reproducible there
        const size_t check_interval = 50000, outer = ransac_k_ / check_interval;
        bool have = false;
        Best best{};
        std::vector<double> fs;
        std::vector<uint32_t> counts;
        std::vector<double> errs;
        for (size_t round = 0; round < outer; round++) {
            fs.clear();
            for (size_t it = 0; it < check_interval; it++) { // ransac_iteration, :177-190
                std::array<PointMatch, 4> sample;
                choose_inliers(point_matches, rng, sample);
                std::array<double, 9> f;
                if (!calculate_model_affine(sample, f)) continue;
                bool finite = true, sample_fits = true;
                for (double v : f) finite = finite && std::isfinite(v);
                if (!finite) continue; // validate_f, :197-199
                for (const auto &m : sample) sample_fits = sample_fits && fits(f, m); // :206-209
                if (!sample_fits) continue;
                fs.insert(fs.end(), f.begin(), f.end());
            }
            const uint32_t H = (uint32_t)(fs.size() / 9);
            counts.assign(H, 0);
            errs.assign(H, 0.0);
            check(cvhip_ransac_score(dev.handle(), fs.data(), H, flat.data(), (uint32_t)point_matches.size(), ransac_t_,
                                     counts.data(), errs.data()),
                  "cvhip_ransac_score");
            for (uint32_t h = 0; h < H; h++) {
                if (counts[h] < ransac_d_ + ransac_n_) continue; // :218-220
                Best cand;
                std::copy_n(&fs[9 * (size_t)h], 9, cand.f.begin());
                cand.matches_count = counts[h];
                cand.best_error = errs[h] / (double)counts[h];
                if (!have || better(cand, best)) {
                    best = cand;
                    have = true;
                }
            }
            if (have && best.matches_count > ransac_d_early_exit_) break; // :135-141
        }
        if (!have) throw RansacError("No reliable matches found");
        FundamentalMatrixResult res; // optimize_result, affine branch (:231-239)
        res.f = best.f;
        for (const auto &m : point_matches)
            if (fits(best.f, m)) res.inliers.push_back(m);
        return res;
    }

    // reprojection_error (fundamentalmatrix.rs:461-471), nalgebra evaluation order
    static double reprojection_error(const std::array<double, 9> &F, const PointMatch &m)
    {
        const double p1x = (double)m.first.x, p1y = (double)m.first.y, p2x = (double)m.second.x, p2y = (double)m.second.y;
        const double r0 = (p2x * F[0] + p2y * F[3]) + F[6], r1 = (p2x * F[1] + p2y * F[4]) + F[7],
                     r2 = (p2x * F[2] + p2y * F[5]) + F[8];
        double n = r0 * p1x;
        n = r1 * p1y + n;
        n = r2 + n;
        double a0 = F[0] * p1x;
        a0 = F[1] * p1y + a0;
        a0 = F[2] + a0;
        double a1 = F[3] * p1x;
        a1 = F[4] * p1y + a1;
        a1 = F[5] + a1;
        const double b0 = (F[0] * p2x + F[3] * p2y) + F[6], b1 = (F[1] * p2x + F[4] * p2y) + F[7];
        return (n * n) / (a0 * a0 + a1 * a1 + b0 * b0 + b1 * b1);
    }

  private:
    struct Best {
        std::array<double, 9> f;
        size_t matches_count;
        double best_error;
    };
    // Ord for RansacIterationResult (fundamentalmatrix.rs:623-649): more matches, then finite and lower error
    static bool better(const Best &a, const Best &b)
    {
        if (a.matches_count != b.matches_count) return a.matches_count > b.matches_count;
        const bool af = std::isfinite(a.best_error), bf = std::isfinite(b.best_error);
        if (af != bf) return af;
        if (!af) return false;
        return a.best_error < b.best_error;
    }
    bool fits(const std::array<double, 9> &f, const PointMatch &m) const // fits_model, :452-458
    {
        const double err = reprojection_error(f, m);
        return std::isfinite(err) && !(std::fabs(err) > ransac_t_);
    }
    static size_t dist(size_t x, size_t y) { return x > y ? x - y : y - x; }
    template <typename Rng>
    static void choose_inliers(const std::vector<PointMatch> &pm, Rng &rng, std::array<PointMatch, 4> &out) // :155-175
    {
        const size_t limit = std::min<size_t>(pm.size(), 5000);
        std::uniform_int_distribution<size_t> pick(0, limit - 1);
        size_t have = 0;
        while (have < 4) {
            const PointMatch &next = pm[pick(rng)];
            bool close = false;
            for (size_t i = 0; i < have; i++) {
                const PointMatch &c = out[i];
                close = close || dist(next.first.x, c.first.x) < 10 || dist(next.first.y, c.first.y) < 10 ||
                        dist(next.second.x, c.second.x) < 10 || dist(next.second.y, c.second.y) < 10;
            }
            if (!close) out[have++] = next;
        }
    }
    // calculate_model_affine (fundamentalmatrix.rs:260-286): rows (x2, y2, x1, y1), mean-centred; the
    // right-singular vector of the smallest singular value via Jacobi eigen-decomposition of A^T A.
    static bool calculate_model_affine(const std::array<PointMatch, 4> &s, std::array<double, 9> &f)
    {
        double a[4][4], mean[4] = {0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            a[i][0] = (double)s[i].second.x;
            a[i][1] = (double)s[i].second.y;
            a[i][2] = (double)s[i].first.x;
            a[i][3] = (double)s[i].first.y;
            for (int j = 0; j < 4; j++) mean[j] += a[i][j] / 4.0;
        }
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) a[i][j] -= mean[j];
        double m[4][4], v[4][4];
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                m[i][j] = 0;
                for (int k = 0; k < 4; k++) m[i][j] += a[k][i] * a[k][j];
                v[i][j] = i == j ? 1.0 : 0.0;
            }
        for (int sweep = 0; sweep < 60; sweep++) { // cyclic Jacobi
            double off = 0;
            for (int p = 0; p < 4; p++)
                for (int q = p + 1; q < 4; q++) off += m[p][q] * m[p][q];
            if (off < 1e-300) break;
            for (int p = 0; p < 4; p++)
                for (int q = p + 1; q < 4; q++) {
                    if (std::fabs(m[p][q]) < 1e-300) continue;
                    const double theta = (m[q][q] - m[p][p]) / (2.0 * m[p][q]);
                    const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                    const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                    for (int k = 0; k < 4; k++) {
                        const double mkp = m[k][p], mkq = m[k][q];
                        m[k][p] = c * mkp - sn * mkq;
                        m[k][q] = sn * mkp + c * mkq;
                    }
                    for (int k = 0; k < 4; k++) {
                        const double mpk = m[p][k], mqk = m[q][k];
                        m[p][k] = c * mpk - sn * mqk;
                        m[q][k] = sn * mpk + c * mqk;
                    }
                    for (int k = 0; k < 4; k++) {
                        const double vkp = v[k][p], vkq = v[k][q];
                        v[k][p] = c * vkp - sn * vkq;
                        v[k][q] = sn * vkp + c * vkq;
                    }
                }
        }
        int order[4] = {0, 1, 2, 3};
        std::sort(order, order + 4, [&](int x, int y) { return m[x][x] > m[y][y]; });
        // singular values = sqrt(eigenvalues); reject rank-deficient samples: s[1] < 1e-3 (:272-275)
        if (std::sqrt(std::max(m[order[1]][order[1]], 0.0)) < 0.001) return false;
        const int last = order[3];
        const double vt[4] = {v[0][last], v[1][last], v[2][last], v[3][last]};
        const double e = vt[0] * mean[0] + vt[1] * mean[1] + vt[2] * mean[2] + vt[3] * mean[3];
        const double raw[9] = {0, 0, vt[0], 0, 0, vt[1], vt[2], vt[3], -e};
        for (int i = 0; i < 9; i++) f[i] = raw[i] / raw[8];
        return true;
    }

    ProjectionMode projection_;
    double max_dimension_;
    size_t ransac_k_, ransac_n_, ransac_d_, ransac_d_early_exit_;
    double ransac_t_;
};

} // namespace cvhip_host
