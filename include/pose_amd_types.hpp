// Value types and data-format readers of the C++ host facade (no device work, no library calls):
//   Mat<R,C>                          stand-in for the fixed-size Eigen matrices of the reference's signatures
//   SiftMatch, SIFT_keypoint, point3D, view, feature_3D_info
//                                     (query_processor.hpp:20-31, SIFT_keypoint.hpp:35-55, parse_bundler.hpp:52-174)
//   SIFT_loader::load_Lowe_features   (SIFT_loader.cpp:110-154)   Lowe .key text files
//   bundler_camera, parse_bundler     (bundler_camera.hpp:30-60, parse_bundler.cpp:80-171)  Bundler bundle.out
// include/pose_amd.hpp builds the solver and processor classes on these.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <new>
#include <set>
#include <string>
#include <vector>

namespace pose_estimation_amd {

// Column-major and contiguous like Eigen::Matrix<double,R,C>, so an Eigen build can construct it
// from `m.data()`.
template <int R, int C>
struct Mat {
    double v[R * C];
    Mat() { std::memset(v, 0, sizeof v); }
    explicit Mat(const double *colmajor) { std::memcpy(v, colmajor, sizeof v); }
    double &operator()(int r, int c) { return v[c * R + r]; }
    double operator()(int r, int c) const { return v[c * R + r]; }
    double *data() { return v; }
    const double *data() const { return v; }
};
using Matrix3d = Mat<3, 3>;
using Vector3d = Mat<3, 1>;
using Vector2d = Mat<2, 1>;

// ---- correspondence records -----------------------------------------------------------------------
struct SiftMatch {
    unsigned int lindex = 0;    // query keypoint
    unsigned int mindex = 0;    // 3D point
    float distance = 0.0f;
    std::set<unsigned int> camset;
    SiftMatch() {}
    SiftMatch(unsigned int lindex_p, unsigned int mindex_p, float distance_p) : lindex(lindex_p), mindex(mindex_p), distance(distance_p) {}
};
struct SIFT_keypoint { float x = 0, y = 0, scale = 0, orientation = 0; };
struct point3D { float x = 0, y = 0, z = 0; };
// parse_bundler.hpp:108-122: one observation of a 3-D point
struct view {
    uint32_t camera = 0;        // camera that sees the point
    uint32_t key = 0;           // index of the feature in that camera's .key file
    float x = 0, y = 0, scale = 0, orientation = 0;
};
// parse_bundler.hpp:125-175: a 3-D point, its views and their SIFT descriptors (128 bytes per view,
// the descriptor of view_list[i] at descriptors[128 * i])
struct feature_3D_info {
    point3D point;
    std::vector<view> view_list;
    std::vector<unsigned char> descriptors;
};

// ---- SIFT_loader.cpp:110-154 ----------------------------------------------------------------------
// Format: "<count> <descriptor size = 128>", then per keypoint "y x scale orientation" followed by
// 128 descriptor values.  Coordinates have their origin in the upper left corner of the image.
// Returns false (and leaves the loader empty) on a missing, truncated or implausible file: a
// keypoint takes at least 2 * (4 + 128) characters, so a count beyond size / 264 is a corrupt header.
class SIFT_loader {
public:
    bool load_Lowe_features(const char *filename)
    {
        keypoints_.clear();
        descriptors_.clear();
        std::ifstream in(filename, std::ios::in);
        if (!in.is_open()) return false;
        in.seekg(0, std::ios::end);
        const std::streamoff bytes = in.tellg();
        in.seekg(0, std::ios::beg);
        uint32_t count = 0, size_descriptor = 0;
        in >> count >> size_descriptor;
        if (!in || size_descriptor != 128 || bytes < 0 || (uint64_t)count > (uint64_t)bytes / 264 + 1) return false;
        try {
            keypoints_.assign(count, SIFT_keypoint());
            descriptors_.assign((size_t)count * 128, 0);
        } catch (const std::bad_alloc &) {
            keypoints_.clear(); descriptors_.clear();
            return false;
        }
        double x, y, scale, orientation;
        unsigned int e;
        for (uint32_t i = 0; i < count && in; ++i) {
            in >> y >> x >> scale >> orientation;          // the file stores y before x
            SIFT_keypoint k;
            k.x = (float)x; k.y = (float)y; k.scale = (float)scale; k.orientation = (float)orientation;
            keypoints_[i] = k;
            for (int j = 0; j < 128 && in; ++j) { in >> e; descriptors_[(size_t)i * 128 + j] = (unsigned char)e; }
        }
        if (in.fail()) { keypoints_.clear(); descriptors_.clear(); return false; }
        return true;
    }
    std::vector<SIFT_keypoint> &get_keypoints() { return keypoints_; }
    std::vector<unsigned char> &get_descriptors() { return descriptors_; }   // 128 per keypoint
    uint32_t get_nb_features() const { return (uint32_t)keypoints_.size(); }

private:
    std::vector<SIFT_keypoint> keypoints_;
    std::vector<unsigned char> descriptors_;
};

// ---- bundler_camera.hpp:30-60, parse_bundler.cpp:80-171 ---------------------------------------------
struct bundler_camera {
    Matrix3d rotation;
    Vector3d translation;
    double focal_length = 0, kappa_1 = 0, kappa_2 = 0;
    int width = 0, height = 0;
};

class parse_bundler {
public:
    // bundle.out v0.3: "<cameras> <points>", per camera f k1 k2, R (3 rows), t; per point position,
    // colour, view list "<k> (camera key x y) * k".  Counts are checked against the file size before
    // anything is allocated (a camera takes >= 30 characters, a point >= 14, a view >= 8).
    bool parse_data(const char *bundle_file)
    {
        cameras_.clear();
        points_.clear();
        std::ifstream in(bundle_file, std::ios::in);
        if (!in.is_open()) return false;
        in.seekg(0, std::ios::end);
        const uint64_t bytes = (uint64_t)std::max<std::streamoff>(in.tellg(), 0);
        in.seekg(0, std::ios::beg);
        std::string header;
        std::getline(in, header);                            // "# Bundle file v0.3"
        uint32_t ncam = 0, npts = 0;
        in >> ncam >> npts;
        if (!in || (uint64_t)ncam * 30 + (uint64_t)npts * 14 > bytes + 64) return false;
        try {
            cameras_.assign(ncam, bundler_camera());
            for (bundler_camera &c : cameras_) {
                in >> c.focal_length >> c.kappa_1 >> c.kappa_2;
                for (int r = 0; r < 3; ++r)
                    for (int k = 0; k < 3; ++k) in >> c.rotation(r, k);
                for (int r = 0; r < 3; ++r) in >> c.translation(r, 0);
                if (!in) break;
            }
            points_.assign(npts, feature_3D_info());
            for (uint32_t i = 0; i < npts && in; ++i) {
                int r, g, b;
                uint32_t k = 0;
                in >> points_[i].point.x >> points_[i].point.y >> points_[i].point.z >> r >> g >> b >> k;
                if (!in || (uint64_t)k * 8 > bytes) { in.setstate(std::ios::failbit); break; }
                points_[i].view_list.resize(k);
                for (view &v : points_[i].view_list) in >> v.camera >> v.key >> v.x >> v.y;
            }
        } catch (const std::bad_alloc &) {
            in.setstate(std::ios::failbit);
        }
        if (in.fail()) { cameras_.clear(); points_.clear(); return false; }
        return true;
    }
    // parse_bundler.cpp:225-290: the descriptors of every point's views, taken from the Lowe .key file
    // of the view's camera (key_files[camera]; the view list names the keypoint inside that file).
    // Fills feature_3D_info::descriptors, 128 bytes per view in view-list order.  The views are
    // grouped by camera once, so every key file is read once and only its own views are visited.
    bool load_descriptors(const std::vector<std::string> &key_files)
    {
        if (key_files.size() < cameras_.size()) return false;
        struct ref { uint32_t point, slot, key; };
        std::vector<std::vector<ref>> by_camera(cameras_.size());
        for (size_t i = 0; i < points_.size(); ++i) {
            points_[i].descriptors.assign(points_[i].view_list.size() * 128, 0);
            for (size_t k = 0; k < points_[i].view_list.size(); ++k) {
                const view &v = points_[i].view_list[k];
                if ((size_t)v.camera >= cameras_.size()) return false;
                by_camera[v.camera].push_back({(uint32_t)i, (uint32_t)k, v.key});
            }
        }
        for (size_t c = 0; c < cameras_.size(); ++c) {
            if (by_camera[c].empty()) continue;
            SIFT_loader keys;
            if (!keys.load_Lowe_features(key_files[c].c_str())) return false;
            const std::vector<unsigned char> &d = keys.get_descriptors();
            for (const ref &r : by_camera[c]) {
                if ((size_t)r.key >= keys.get_nb_features()) return false;
                std::copy(d.begin() + (size_t)r.key * 128, d.begin() + ((size_t)r.key + 1) * 128, points_[r.point].descriptors.begin() + (size_t)r.slot * 128);
            }
        }
        return true;
    }
    std::vector<bundler_camera> &get_cameras() { return cameras_; }
    std::vector<feature_3D_info> &get_feature_infos() { return points_; }
    // the cameras that see point i: what SiftMatch::camset holds (advanced.cpp:272-285)
    std::set<unsigned int> camset(size_t i) const
    {
        std::set<unsigned int> s;
        for (const view &v : points_[i].view_list) s.insert(v.camera);
        return s;
    }

private:
    std::vector<bundler_camera> cameras_;
    std::vector<feature_3D_info> points_;
};

} // namespace pose_estimation_amd
