// Host-side callers on either side of the RANSAC path:
//   LoadGoldenQuery                   (import_export.cpp:57-105)  Bundler camera -> Query pose convention
//   Benchmark                         (benchmark.cpp:19-67)       Process every query, compare, write the CSV
// The readers (Lowe .key files, Bundler bundle.out) are in pose_amd_types.hpp, Query / QueryLoader and
// the processors in pose_amd.hpp.  Benchmark exists with the reference's signature (the processor
// matches the queries' descriptors itself) and for callers that bring their own good_matches lists.
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
// For callers with a matching stage of their own: queries together with their good_matches lists.
struct QuerySource {
    std::vector<Query> queries;
    std::vector<std::vector<SiftMatch>> good_matches;
    std::vector<feature_3D_info> sifts;
    int GetQueryCount() const { return (int)queries.size(); }
};

struct BenchmarkRow { int index; double seconds, position_error, rotation_error, focal_length_error; };

namespace detail {
inline BenchmarkRow benchmark_row(int i, double seconds, const Query &g, const Query &query)
{
    double pe = 0, re = 0;
    const Vector3d gp = g.camera_position(), qpz = query.camera_position();
    const Matrix3d gr = g.camera_rotation(), qr = query.camera_rotation();
    for (int k = 0; k < 3; ++k) pe += (gp(k, 0) - qpz(k, 0)) * (gp(k, 0) - qpz(k, 0));
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) re += (gr(r, c) - qr(r, c)) * (gr(r, c) - qr(r, c));
    return {i + 1, seconds, std::sqrt(pe), std::sqrt(re), g.focal_length() - query.focal_length()};
}
inline void benchmark_report(const BenchmarkRow &row, const Query &query, std::ofstream &out_file, std::ostream *log)
{
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
}
} // namespace detail

// benchmark.hpp:17, benchmark.cpp:19-67 with the reference's arguments: for every query of the loader,
// the golden pose from `golden_pb` (LoadGoldenQuery on <list_path>list.orig.txt), a copy of the query
// through qp->Process(query) (timed), and one line "index,seconds,position error,rotation error
// (Frobenius),focal length error" with 12 significant digits, flushed per query; the reference's
// summary block goes to `log` (nullptr: silent).  The reference hard-codes the two paths given here
// as defaults.  Query i uses the random stream (qp->seed, i).
inline std::vector<BenchmarkRow> Benchmark(QueryLoader &ql, parse_bundler &golden_pb, QueryProcessor *qp,
                                           const std::string &list_path = "data/Dubrovnik6K/bundle/",
                                           const std::string &out_path = "out/benchmark.txt", std::ostream *log = &std::cout)
{
    std::ofstream out_file(out_path);
    if (!out_file.is_open()) throw std::runtime_error("Benchmark: cannot open " + out_path);
    std::vector<BenchmarkRow> rows;
    for (int i = 0; i < ql.GetQueryCount(); ++i) {
        const Query golden = LoadGoldenQuery(golden_pb, i, list_path);
        Query query = ql.GetQuery(i, false);                 // a copy
        qp->query_id = (uint32_t)i;
        const auto start = std::chrono::high_resolution_clock::now();
        qp->Process(query);
        const auto end = std::chrono::high_resolution_clock::now();
        const BenchmarkRow row = detail::benchmark_row(i, std::chrono::duration<double>(end - start).count(), golden, query);
        detail::benchmark_report(row, query, out_file, log);
        rows.push_back(row);
    }
    return rows;
}

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
        const BenchmarkRow row = detail::benchmark_row(i, elapsed_seconds, golden[(size_t)i], query);
        detail::benchmark_report(row, query, out_file, log);
        rows.push_back(row);
    }
    return rows;
}

} // namespace pose_estimation_amd
