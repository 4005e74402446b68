// Host-side callers and data formats on either side of the RANSAC path (no device work here):
//   SIFT_loader::load_Lowe_features   (SIFT_loader.cpp:110-154)   Lowe .key text files
//   parse_bundler::parse_data         (parse_bundler.cpp:80-171)  Bundler bundle.out: cameras + points
//   LoadGoldenQuery                   (import_export.cpp:57-105)  Bundler camera -> Query pose convention
//   Benchmark                         (benchmark.cpp:19-67)       Process every query, compare, write the CSV
// The matching stage (FLANN) that the reference runs inside Process is not part of this library,
// so the benchmark's query source hands over the good_matches lists together with the queries.
#pragma once
#include <chrono>
#include <cmath>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "pose_amd.hpp"

namespace pose_estimation_amd {

// ---- SIFT_loader.cpp:110-154 ---------------------------------------------------------------------
// Format: "<count> <descriptor size = 128>", then per keypoint "y x scale orientation" followed by
// 128 descriptor values.  Coordinates have their origin in the upper left corner of the image.
class SIFT_loader {
public:
    bool load_Lowe_features(const char *filename)
    {
        std::ifstream in(filename, std::ios::in);
        if (!in.is_open()) return false;
        uint32_t count = 0, size_descriptor = 0;
        in >> count >> size_descriptor;
        if (!in || size_descriptor != 128) return false;
        keypoints_.assign(count, SIFT_keypoint());
        descriptors_.assign((size_t)count * 128, 0);
        double x, y, scale, orientation;
        unsigned int e;
        for (uint32_t i = 0; i < count; ++i) {
            in >> y >> x >> scale >> orientation;          // the file stores y before x
            SIFT_keypoint k;
            k.x = (float)x; k.y = (float)y; k.scale = (float)scale; k.orientation = (float)orientation;
            keypoints_[i] = k;
            for (int j = 0; j < 128; ++j) { in >> e; descriptors_[(size_t)i * 128 + j] = (unsigned char)e; }
        }
        return !in.fail();
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

struct bundler_view { uint32_t camera = 0, key = 0; float x = 0, y = 0; };

class parse_bundler {
public:
    // bundle.out v0.3: "<cameras> <points>", per camera f k1 k2, R (3 rows), t; per point position,
    // colour, view list "<k> (camera key x y) * k".
    bool parse_data(const char *bundle_file)
    {
        std::ifstream in(bundle_file, std::ios::in);
        if (!in.is_open()) return false;
        std::string header;
        std::getline(in, header);                            // "# Bundle file v0.3"
        uint32_t ncam = 0, npts = 0;
        in >> ncam >> npts;
        if (!in) return false;
        cameras_.assign(ncam, bundler_camera());
        for (bundler_camera &c : cameras_) {
            in >> c.focal_length >> c.kappa_1 >> c.kappa_2;
            for (int r = 0; r < 3; ++r)
                for (int k = 0; k < 3; ++k) in >> c.rotation(r, k);
            for (int r = 0; r < 3; ++r) in >> c.translation(r, 0);
        }
        points_.assign(npts, feature_3D_info());
        views_.assign(npts, std::vector<bundler_view>());
        for (uint32_t i = 0; i < npts; ++i) {
            int r, g, b;
            uint32_t k = 0;
            in >> points_[i].point.x >> points_[i].point.y >> points_[i].point.z >> r >> g >> b >> k;
            views_[i].resize(k);
            for (bundler_view &v : views_[i]) in >> v.camera >> v.key >> v.x >> v.y;
        }
        return !in.fail();
    }
    // parse_bundler.cpp:225-290: the descriptors of every point's views, taken from the Lowe .key file
    // of the view's camera (key_files[camera]; the view list names the keypoint inside that file).
    // Fills feature_3D_info::descriptors, 128 bytes per view in view-list order.
    bool load_descriptors(const std::vector<std::string> &key_files)
    {
        if (key_files.size() < cameras_.size()) return false;
        for (size_t i = 0; i < points_.size(); ++i) points_[i].descriptors.assign(views_[i].size() * 128, 0);
        for (size_t c = 0; c < cameras_.size(); ++c) {
            bool needed = false;
            for (size_t i = 0; i < points_.size() && !needed; ++i)
                for (const bundler_view &v : views_[i]) needed = needed || v.camera == c;
            if (!needed) continue;
            SIFT_loader keys;
            if (!keys.load_Lowe_features(key_files[c].c_str())) return false;
            const std::vector<unsigned char> &d = keys.get_descriptors();
            for (size_t i = 0; i < points_.size(); ++i)
                for (size_t k = 0; k < views_[i].size(); ++k) {
                    const bundler_view &v = views_[i][k];
                    if (v.camera != c) continue;
                    if ((size_t)v.key >= keys.get_nb_features()) return false;
                    std::copy(d.begin() + (size_t)v.key * 128, d.begin() + ((size_t)v.key + 1) * 128, points_[i].descriptors.begin() + k * 128);
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
        for (const bundler_view &v : views_[i]) s.insert(v.camera);
        return s;
    }

private:
    std::vector<bundler_camera> cameras_;
    std::vector<feature_3D_info> points_;
    std::vector<std::vector<bundler_view>> views_;
};

// ---- import_export.cpp:57-105 -------------------------------------------------------------------------
// The query_camera_id-th line of list.orig.txt that starts with "query/" names the camera of the
// golden reconstruction; its Bundler pose (looking down -Z) becomes position = -R^T t,
// rotation = (diag(-1, 1, -1) R)^T.
inline Query GoldenQueryFromCamera(const bundler_camera &camera)
{
    Query q(camera.width, camera.height, camera.focal_length);
    const Matrix3d &R = camera.rotation;
    Vector3d pos;
    Matrix3d rot;
    const double flip[3] = {-1.0, 1.0, -1.0};
    for (int r = 0; r < 3; ++r) {
        pos(r, 0) = -(R(0, r) * camera.translation(0, 0) + R(1, r) * camera.translation(1, 0) + R(2, r) * camera.translation(2, 0));
        for (int c = 0; c < 3; ++c) rot(r, c) = flip[c] * R(c, r);          // ((xzflip R)^T)(r, c) = (xzflip R)(c, r)
    }
    q.set_camera_position(pos);
    q.set_camera_rotation(rot);
    return q;
}

inline Query LoadGoldenQuery(parse_bundler &pb, int query_camera_id, const std::string &list_path)
{
    std::ifstream queryfile(list_path + "list.orig.txt");
    std::string line;
    int counter_query = 0, counter_total = 0, total_camera_id = 0;
    while (std::getline(queryfile, line)) {
        const size_t slash = line.find('/');
        if (slash != std::string::npos && line.substr(0, slash) == "query") ++counter_query;
        ++counter_total;
        if (counter_query == query_camera_id + 1) { total_camera_id = counter_total; break; }
    }
    if (total_camera_id <= 0 || (size_t)total_camera_id > pb.get_cameras().size())
        throw std::runtime_error("LoadGoldenQuery: query camera not found in " + list_path + "list.orig.txt");
    return GoldenQueryFromCamera(pb.get_cameras()[(size_t)total_camera_id - 1]);
}

// ---- benchmark.cpp:19-67 ------------------------------------------------------------------------------
// The reference's QueryLoader returns queries whose matches Process computes itself; here the
// source also returns the good_matches of each query (the matching stage is outside this library).
struct QuerySource {
    std::vector<Query> queries;
    std::vector<std::vector<SiftMatch>> good_matches;
    std::vector<feature_3D_info> sifts;
    int GetQueryCount() const { return (int)queries.size(); }
};

struct BenchmarkRow { int index; double seconds, position_error, rotation_error, focal_length_error; };

// golden[i]: the pose to compare query i with (LoadGoldenQuery / GoldenQueryFromCamera).  Writes one
// line "index,seconds,position error,rotation error (Frobenius),focal length error" per query with
// 12 significant digits, flushed per query, and prints the reference's summary block to `log`
// (nullptr: silent).  Query i uses the random stream (qp->seed, i).
inline std::vector<BenchmarkRow> Benchmark(QuerySource &ql, std::vector<Query> &golden, QueryProcessor *qp,
                                           const std::string &out_path = "out/benchmark.txt", std::ostream *log = &std::cout)
{
    std::ofstream out_file(out_path);
    if (!out_file.is_open()) throw std::runtime_error("Benchmark: cannot open " + out_path);
    std::vector<BenchmarkRow> rows;
    for (int i = 0; i < ql.GetQueryCount(); ++i) {
        Query query = ql.queries[(size_t)i];                 // a copy, as GetQuery(i, false) returns
        qp->query_id = (uint32_t)i;
        const auto start = std::chrono::high_resolution_clock::now();
        qp->Process(query, ql.good_matches[(size_t)i], ql.sifts);
        const auto end = std::chrono::high_resolution_clock::now();
        const double elapsed_seconds = std::chrono::duration<double>(end - start).count();
        const Query &g = golden[(size_t)i];
        double pe = 0, re = 0;
        const Vector3d gp = g.camera_position(), qpz = query.camera_position();
        const Matrix3d gr = g.camera_rotation(), qr = query.camera_rotation();
        for (int k = 0; k < 3; ++k) pe += (gp(k, 0) - qpz(k, 0)) * (gp(k, 0) - qpz(k, 0));
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) re += (gr(r, c) - qr(r, c)) * (gr(r, c) - qr(r, c));
        const BenchmarkRow row = {i + 1, elapsed_seconds, std::sqrt(pe), std::sqrt(re), g.focal_length() - query.focal_length()};
        if (log) {
            *log << "=================================================" << std::endl;
            *log << "Index: " << row.index << std::endl;
            *log << "Image size: " << query.image_width() << " x " << query.image_height() << std::endl;
            *log << "Computation time: " << row.seconds << " s" << std::endl;
            *log << "Position error: " << row.position_error << std::endl;
            *log << "Rotation error: " << row.rotation_error << std::endl;
            *log << "Focal length error: " << row.focal_length_error << std::endl;
            *log << "=================================================" << std::endl;
        }
        out_file << row.index << "," << std::setprecision(12) << row.seconds << "," << std::setprecision(12) << row.position_error << ","
                 << std::setprecision(12) << row.rotation_error << "," << std::setprecision(12) << row.focal_length_error << std::endl;
        out_file.flush();
        rows.push_back(row);
    }
    return rows;
}

} // namespace pose_estimation_amd
