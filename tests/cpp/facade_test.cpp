// Exercises the C++ host facade (include/pose_amd.hpp) the way the reference's own test
// drivers exercise its classes (p3p_testing.cpp:40-65, p4p_testing.cpp:49-110), then runs the
// two processors on a synthetic query with a planted pose.  Exit status 0 = all checks passed.
#include <cmath>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <iomanip>
#include <random>
#include <set>
#include <string>

#include "pose_amd.hpp"
#include "pose_amd_io.hpp"
#include "../../oracle/pose_oracle.h"      // the CPU oracle: the checker of this test, nothing more

using namespace pose_estimation_amd;

static int failures = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

static void test_p3p()
{
    // p3p_testing.cpp:40-53
    const double pts[9] = {0.0, 0.0, 1.0, 10.5, 10.5, 10.5, -10.5, -9.5, -9.5};                // one point per row
    const double fts[9] = {-0.049875, -0.049875, 0.049875, 0.049875, -0.049875, -0.049875, 0.99751, 0.99751, 0.99751};
    Mat<3, 3> worldPoints, featureVectors;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { worldPoints(r, c) = pts[3 * r + c]; featureVectors(r, c) = fts[3 * r + c]; }
    const double golden[48] = {
        1.7739, 0.9917, 0.0064, -0.1281, 0.5665, 1.0000, 0.0003, -0.0066, 0.1272, 0.9993, -0.0019, 0.0372, -0.4751, 0.9952,
        -0.0049, 0.0977, 0.6782, 0.1283, -0.0356, 0.9911, 0.5227, 0.0066, -0.0059, 1.0000, 0.7136, -0.0366, 0.1435, 0.9890,
        0.6218, -0.0977, -0.1019, 0.9900, -9.6385, 0.0018, -0.9994, -0.0361, -9.9416, 0.0003, -1.0000, -0.0059, -11.4235,
        -0.0072, -0.9896, 0.1434, -8.9839, 0.0051, -0.9948, -0.1019};
    P3p p3p;
    Mat<3, 16> sol;
    CHECK(p3p.computePoses(featureVectors, worldPoints, sol) == 0);
    double err = 0;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 16; ++c) { const double d = sol(r, c) - golden[16 * r + c]; err += d * d; }
    CHECK(std::sqrt(err) < 0.15);                  // p3p_testing.cpp:65
    Mat<3, 3> collinear;
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) collinear(r, c) = c;
    CHECK(p3p.computePoses(featureVectors, collinear, sol) == -1);
}

static void test_p4pf()
{
    // p4p_testing.cpp:49-70
    const double M[12] = {-3.33639834336120, -23.35549638285873, -13.18941519576778, 6.43164913914748,
                          0.65948286155096, -4.90376715918747, 1.17103701629876, 0.14580433383203,
                          -8.46658219501120, -3.99876939947909, -3.02248927651177, -22.16086539862748};
    const double m[8] = {0.11009888473695, 0.39776592879400, 0.28752996253253, -0.05017617729940,
                         0.03882059658299, -0.17303453640632, -0.05791310109713, 0.19297848817239};
    Mat<3, 4> M3D;
    Mat<2, 4> m2D;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) M3D(r, c) = M[4 * r + c];
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 4; ++c) m2D(r, c) = m[4 * r + c];
    P4pf p4pf;
    std::vector<double> f;
    std::vector<Matrix3d> R;
    std::vector<Vector3d> t;
    f.push_back(-1.0);                              // results are appended, never overwritten
    R.push_back(Matrix3d());
    t.push_back(Vector3d());
    CHECK(p4pf.P4Pf_m(m2D, M3D, &f, &R, &t) == 0);
    CHECK(f.size() >= 2 && f.size() == R.size() && f.size() == t.size() && f[0] == -1.0);
    double best = 1e9;
    size_t bi = 0;
    for (size_t i = 1; i < f.size(); ++i)
        if (std::fabs(f[i] - 1.5) < best) { best = std::fabs(f[i] - 1.5); bi = i; }
    CHECK(best < 1e-9);
    const double tgt[3] = {2.85817409844554, -2.17296255562889, 77.54246130780075};
    double terr = 0;
    for (int k = 0; k < 3; ++k) terr += (t[bi](k, 0) - tgt[k]) * (t[bi](k, 0) - tgt[k]);
    CHECK(std::sqrt(terr) < 1e-7);
    CHECK(std::fabs(R[bi](0, 0) + 0.94382954756954) < 1e-9 && std::fabs(R[bi](1, 2) + 0.56276540799790) < 1e-9);
    // degenerate: two identical world points (P4pf.cpp:343-346)
    for (int r = 0; r < 3; ++r) M3D(r, 1) = M3D(r, 0);
    std::vector<double> f2;
    std::vector<Matrix3d> R2;
    std::vector<Vector3d> t2;
    CHECK(p4pf.P4Pf_m(m2D, M3D, &f2, &R2, &t2) == -1 && f2.empty());
}

struct Scene {
    Query query;
    std::vector<SiftMatch> matches;
    std::vector<feature_3D_info> sifts;
    double Ctrue[3];
    Scene(int w, int h, double focal) : query(w, h, focal) {}
};

// n matches, the first `inl` consistent with a camera at Ctrue looking down +Z (Rc = I)
static Scene make_scene(int n, int inl, bool known_focal, unsigned seed)
{
    const int w = 1600, h = 1200;
    const double f = 1200.0;
    Scene s(w, h, known_focal ? f : 0.0);
    s.Ctrue[0] = 3.0; s.Ctrue[1] = -2.0; s.Ctrue[2] = 1.0;
    std::mt19937 rng(seed);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    for (int i = 0; i < n; ++i) {
        const double z = 10.0 + 95.0 * (u(rng) + 1.0);
        const double px = 0.45 * w * u(rng), py = 0.45 * h * u(rng);        // centred image coordinates, x right / y up
        feature_3D_info p3;
        p3.point.x = (float)(s.Ctrue[0] + px * z / f); p3.point.y = (float)(s.Ctrue[1] + py * z / f); p3.point.z = (float)(s.Ctrue[2] + z);
        s.sifts.push_back(p3);
        SIFT_keypoint k;
        double ox = px, oy = py;
        if (i >= inl) { ox = 0.5 * w * u(rng); oy = 0.5 * h * u(rng); }
        // obs = -(k - w/2)  (query_processor.cpp:111-112)
        k.x = (float)(w / 2.0 - ox); k.y = (float)(h / 2.0 - oy);
        s.query.sift_keypoints().push_back(k);
        SiftMatch m;
        m.lindex = (unsigned)i; m.mindex = (unsigned)i; m.distance = 0.1f;
        // camera sets: inliers share cameras 1..6, outliers see two random ones
        if (i < inl) { for (unsigned c = 1; c <= 6; ++c) if ((i + c) % 3 != 0 || i % 7 == 0) m.camset.insert(c); if (i % 7 == 0) m.camset.insert(7); }
        else { m.camset.insert(100 + (unsigned)(rng() % 500)); m.camset.insert(700 + (unsigned)(rng() % 500)); }
        s.matches.push_back(m);
    }
    return s;
}

static void test_processors()
{
    for (int known = 0; known < 2; ++known) {
        Scene s = make_scene(600, 240, known != 0, 5u + known);
        QueryProcessorBasic basic;
        basic.seed = 42;
        CHECK(basic.Process(s.query, s.matches, s.sifts));
        CHECK(basic.result.found == 1 && basic.result.inliers >= 12 && basic.result.steps < 10000);
        CHECK(basic.samples.size() == (size_t)basic.result.steps * 4);
        CHECK(basic.fitted_matches.size() == basic.result.inliers);
        const Vector3d C = s.query.camera_position();
        const double d = std::sqrt((C(0, 0) - 3.0) * (C(0, 0) - 3.0) + (C(1, 0) + 2.0) * (C(1, 0) + 2.0) + (C(2, 0) - 1.0) * (C(2, 0) - 1.0));
        CHECK(d < 3.0);
        CHECK(std::fabs(s.query.focal_length() - 1200.0) < 60.0);
        // the kept pose, re-tested through TestHypothesis, gives back the same inliers
        double q = 0;
        std::vector<SiftMatch> again = basic.TestHypothesis(s.query, s.matches, s.query.sift_keypoints(), s.sifts, 0.5, q);
        CHECK(again.size() == basic.fitted_matches.size() && q > 0.0 && q <= 1.0);

        Scene a = make_scene(600, 240, known != 0, 9u + known);
        QueryProcessorAdvanced adv;
        adv.seed = 43;
        CHECK(adv.Process(a.query, a.matches, a.sifts));
        CHECK(adv.result.found == 1 && adv.result.steps == 100 && adv.samples.size() == 400);
        CHECK(adv.result.inliers >= 100);           // co-occurrence sampling finds the planted cluster
    }
}

// ProcessBatch must give every query the pose Process gives it with the same stream
static void test_batch()
{
    std::vector<Scene> scenes;
    for (int i = 0; i < 6; ++i) scenes.push_back(make_scene(500 + 40 * i, 150 + 20 * i, (i & 1) != 0, 20u + i));
    std::vector<feature_3D_info> sifts;                 // one shared 3D point list, as in the reference
    std::vector<Query *> queries;
    std::vector<std::vector<SiftMatch> *> matches;
    for (Scene &s : scenes) {
        const unsigned base = (unsigned)sifts.size();
        sifts.insert(sifts.end(), s.sifts.begin(), s.sifts.end());
        for (SiftMatch &m : s.matches) m.mindex += base;
        queries.push_back(&s.query);
        matches.push_back(&s.matches);
    }
    QueryProcessorBasic basic;
    basic.seed = 77; basic.query_id = 5;
    std::vector<pose_amd_result> results;
    CHECK(basic.ProcessBatch(queries, matches, sifts, results));
    CHECK(results.size() == scenes.size());
    for (size_t i = 0; i < scenes.size(); ++i) {
        Scene single = make_scene(500 + 40 * (int)i, 150 + 20 * (int)i, (i & 1) != 0, 20u + (unsigned)i);
        QueryProcessorBasic one;
        one.seed = 77; one.query_id = 5 + (uint32_t)i;
        CHECK(one.Process(single.query, single.matches, single.sifts));
        CHECK(results[i].found == one.result.found && results[i].steps == one.result.steps && results[i].inliers == one.result.inliers);
        CHECK(results[i].best_hyp == one.result.best_hyp);
        const Vector3d a = scenes[i].query.camera_position(), b = single.query.camera_position();
        CHECK(a(0, 0) == b(0, 0) && a(1, 0) == b(1, 0) && a(2, 0) == b(2, 0));
        CHECK(scenes[i].query.focal_length() == single.query.focal_length());
    }
    // the same batch spread over a device list (device 0 twice: two contexts, two host threads)
    std::vector<Scene> again;
    for (int i = 0; i < 6; ++i) again.push_back(make_scene(500 + 40 * i, 150 + 20 * i, (i & 1) != 0, 20u + i));
    std::vector<Query *> q2;
    std::vector<std::vector<SiftMatch> *> m2;
    unsigned base = 0;
    for (Scene &s : again) {
        for (SiftMatch &m : s.matches) m.mindex += base;
        base += (unsigned)s.sifts.size();
        q2.push_back(&s.query);
        m2.push_back(&s.matches);
    }
    QueryProcessorBasic multi;
    multi.seed = 77; multi.query_id = 5;
    multi.devices = {0, 0};
    std::vector<pose_amd_result> r2;
    CHECK(multi.ProcessBatch(q2, m2, sifts, r2));
    for (size_t i = 0; i < scenes.size(); ++i) {
        CHECK(r2[i].found == results[i].found && r2[i].steps == results[i].steps && r2[i].best_hyp == results[i].best_hyp && r2[i].inliers == results[i].inliers);
        CHECK(again[i].query.camera_position()(0, 0) == scenes[i].query.camera_position()(0, 0));
    }
    pose_amd_multi_release();
}

// SoA view of a scene's matches for the oracle
struct OracleQuery {
    std::vector<float> kx, ky, X, Y, Z;
    orc_query q;
    OracleQuery(Scene &s, size_t n)
    {
        for (size_t i = 0; i < n; ++i) {
            const SiftMatch &m = s.matches[i];
            kx.push_back(s.query.sift_keypoints()[m.lindex].x); ky.push_back(s.query.sift_keypoints()[m.lindex].y);
            X.push_back(s.sifts[m.mindex].point.x); Y.push_back(s.sifts[m.mindex].point.y); Z.push_back(s.sifts[m.mindex].point.z);
        }
        q.n = (int)n; q.kx = kx.data(); q.ky = ky.data(); q.X = X.data(); q.Y = Y.data(); q.Z = Z.data();
        q.width = s.query.image_width(); q.height = s.query.image_height(); q.focal = 0.0;
    }
};

// QueryProcessor::Project and the fitpoints Process leaves in the query (query_processor.cpp:46-86,
// basic.cpp:296-307) against the oracle's projection: same operations, same order, equal bits.
static void test_project()
{
    Scene s = make_scene(700, 280, false, 31u);
    QueryProcessorBasic basic;
    basic.seed = 11;
    CHECK(basic.Process(s.query, s.matches, s.sifts));
    CHECK(s.query.fitpoints_2d().size() == basic.fitted_matches.size() && s.query.fitpoints_3d().size() == basic.fitted_matches.size());
    OracleQuery oq(s, s.matches.size());
    const Mat<3, 4> Pm = s.query.proj_matrix();
    double P[12];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) P[4 * r + c] = Pm(r, c);
    std::vector<double> e2(2 * s.matches.size()), e3(3 * s.matches.size());
    orc_project(P, oq.q.n, oq.q.X, oq.q.Y, oq.q.Z, e2.data(), e3.data());
    std::vector<Vector2d> p2;
    std::vector<Vector3d> p3;
    CHECK(basic.Project(s.query, p2, p3, s.matches, s.query.sift_keypoints(), s.sifts));
    CHECK(p2.size() == s.matches.size());
    bool same = true;
    for (size_t i = 0; i < s.matches.size(); ++i)
        same = same && p2[i](0, 0) == e2[2 * i] && p2[i](1, 0) == e2[2 * i + 1] && p3[i](0, 0) == e3[3 * i] && p3[i](2, 0) == e3[3 * i + 2];
    CHECK(same);
    // the stored fitpoints are the projections of the fitted matches, in match order
    size_t k = 0;
    bool fit_ok = true;
    for (size_t i = 0; i < s.matches.size() && k < basic.fitted_matches.size(); ++i)
        if (s.matches[i].lindex == basic.fitted_matches[k].lindex) {
            fit_ok = fit_ok && s.query.fitpoints_2d()[k](0, 0) == e2[2 * i] && s.query.fitpoints_3d()[k](1, 0) == e3[3 * i + 1];
            ++k;
        }
    CHECK(fit_ok && k == basic.fitted_matches.size());
}

// The advanced processor's backmatching leg (advanced.cpp:583-606) against the oracle: 100 steps
// that cannot reach 12 inliers, a stand-in Backmatching that appends 100 matches, 100 more steps
// with the first leg's quality as the bar.
struct BackmatchStub { Scene *scene; Scene *extra; };
static bool append_extra(std::vector<SiftMatch> &good_matches, void *user)
{
    BackmatchStub *b = (BackmatchStub *)user;
    const unsigned kbase = (unsigned)b->scene->query.sift_keypoints().size(), pbase = (unsigned)b->scene->sifts.size();
    for (size_t i = 0; i < 100; ++i) {
        b->scene->query.sift_keypoints().push_back(b->extra->query.sift_keypoints()[i]);
        b->scene->sifts.push_back(b->extra->sifts[i]);
        SiftMatch m = b->extra->matches[i];
        m.lindex = kbase + (unsigned)i; m.mindex = pbase + (unsigned)i;
        good_matches.push_back(m);
    }
    return true;
}

static void test_backmatching()
{
    for (int variant = 0; variant < 2; ++variant) {
        Scene s = make_scene(400, 9, false, 41u + variant);          // 9 inliers: the first leg stays below 12
        Scene extra = make_scene(100, variant == 0 ? 90 : 0, false, 43u + variant);   // same planted camera; all-outlier variant
        s.matches[0].camset = {1, 2, 3, 4, 5, 6};                    // somebody must be seen by >= 5 cameras
        BackmatchStub stub{&s, &extra};
        QueryProcessorAdvanced adv;
        adv.seed = 47; adv.query_id = 2;
        adv.backmatching = append_extra; adv.backmatching_user = &stub;
        CHECK(adv.Process(s.query, s.matches, s.sifts));
        CHECK(adv.backmatched && adv.result.steps == 200 && adv.samples.size() == 800 && s.matches.size() == 500);
        OracleQuery q1(s, 400), q2(s, 500);
        orc_ransac_params prm = {1, 0.5, 100, 10, 12, 0};
        orc_ransac_result o;
        std::vector<uint8_t> mask(500);
        int leg = -1;
        const int second = orc_ransac_backmatch(&q1.q, &q2.q, &prm, adv.samples.data(), 100, adv.samples.data() + 400, 100, &o, mask.data(), &leg);
        CHECK(second == 1 && o.steps == 200);
        CHECK(adv.best_leg == leg && adv.result.found == o.found);
        CHECK(adv.result.best_hyp == o.best_step && adv.result.inliers == o.hyposet && adv.result.area_fit == o.area_fit &&
              adv.result.area_all == o.area_all);
        if (variant == 0) CHECK(leg == 1 && o.hyposet >= 50);
        CHECK(adv.fitted_matches.size() == o.hyposet && s.query.fitpoints_2d().size() == o.hyposet);
        if (adv.result.best_sol == o.best_sol) {
            const Vector3d C = s.query.camera_position();
            double d = 0, nrm = 0;
            for (int k = 0; k < 3; ++k) { d += (C(k, 0) - o.C[k]) * (C(k, 0) - o.C[k]); nrm += o.C[k] * o.C[k]; }
            CHECK(std::sqrt(d) <= 1e-6 * std::sqrt(nrm) + 1e-9);     // loose here: the gated 1e-8 comparison is in the Python suite
        }
    }
}

// benchmark.cpp:19-67 through the C++ Benchmark: golden poses are the planted ones
static void test_benchmark()
{
    QuerySource src;
    std::vector<Query> golden;
    for (int i = 0; i < 3; ++i) {
        Scene s = make_scene(500, 200, i == 1, 70u + i);
        const unsigned base = (unsigned)src.sifts.size();
        src.sifts.insert(src.sifts.end(), s.sifts.begin(), s.sifts.end());
        for (SiftMatch &m : s.matches) m.mindex += base;
        src.queries.push_back(s.query);
        src.good_matches.push_back(s.matches);
        bundler_camera cam;                                       // Bundler convention: looks down -Z
        cam.focal_length = 1200.0; cam.width = 1600; cam.height = 1200;
        cam.rotation(0, 0) = -1.0; cam.rotation(1, 1) = 1.0; cam.rotation(2, 2) = -1.0;      // diag(-1,1,-1) * I
        const double C[3] = {3.0, -2.0, 1.0};
        for (int r = 0; r < 3; ++r) cam.translation(r, 0) = -cam.rotation(r, r) * C[r];   // t = -R C
        golden.push_back(GoldenQueryFromCamera(cam));
        CHECK(std::fabs(golden.back().camera_position()(0, 0) - 3.0) < 1e-12 && golden.back().camera_rotation()(0, 0) == 1.0 &&
              golden.back().camera_rotation()(2, 2) == 1.0);
    }
    QueryProcessorBasic basic;
    basic.seed = 5;
    const std::string path = "/tmp/pose_amd_benchmark_test.txt";
    std::vector<BenchmarkRow> rows = Benchmark(src, golden, &basic, path, nullptr);
    CHECK(rows.size() == 3);
    std::ifstream in(path);
    std::string line;
    int nlines = 0;
    while (std::getline(in, line)) {
        int idx = 0;
        double sec = 0, pe = 0, re = 0, fe = 0;
        CHECK(std::sscanf(line.c_str(), "%d,%lf,%lf,%lf,%lf", &idx, &sec, &pe, &re, &fe) == 5);
        CHECK(idx == nlines + 1 && sec > 0 && pe < 3.0 && re < 0.1 && std::fabs(fe) < 60.0);
        CHECK(std::fabs(pe - rows[(size_t)nlines].position_error) <= 1e-11 * std::max(1.0, pe));
        ++nlines;
    }
    CHECK(nlines == 3);
    CHECK(src.queries[0].camera_position()(0, 0) == 0.0);           // Benchmark works on copies, like GetQuery(i, false)
}

// query_processor_basic.cpp:22-96: averaged descriptors, two nearest, ratio test -- against plain loops
static void test_matching()
{
    std::mt19937 rng(5);
    std::uniform_int_distribution<int> byte(0, 90), views(1, 4), jitter(-4, 4);
    std::vector<feature_3D_info> sifts(300);
    for (auto &s : sifts) {
        std::vector<int> proto(128);
        for (int &v : proto) v = byte(rng);
        const int nv = views(rng);
        for (int k = 0; k < nv; ++k)
            for (int j = 0; j < 128; ++j) s.descriptors.push_back((unsigned char)std::max(0, std::min(255, proto[j] + jitter(rng))));
    }
    sifts[77].descriptors = sifts[12].descriptors;                       // an exact tie: the lower index comes first
    std::vector<std::vector<unsigned char>> store(60, std::vector<unsigned char>(128));
    std::vector<unsigned char *> desc;
    for (size_t i = 0; i < store.size(); ++i) {
        for (int j = 0; j < 128; ++j)
            store[i][j] = i % 2 ? (unsigned char)byte(rng) : (unsigned char)std::max(0, std::min(255, (int)sifts[(i * 5) % 300].descriptors[j] + jitter(rng)));
        desc.push_back(store[i].data());
    }
    DescriptorMatcher matcher(sifts);
    std::vector<SiftMatch> good = matcher.Match(desc, 0.7f);
    // the definition, in the reference's arithmetic for the averages and double sums for the distances
    std::vector<std::vector<float>> feat(sifts.size(), std::vector<float>(128, 0.f));
    for (size_t i = 0; i < sifts.size(); ++i) {
        const size_t nv = sifts[i].descriptors.size() / 128;
        for (size_t k = 0; k < nv; ++k) for (int j = 0; j < 128; ++j) feat[i][j] += (float)sifts[i].descriptors[j + 128 * k];
        for (int j = 0; j < 128; ++j) feat[i][j] /= (float)nv;
    }
    size_t g = 0, expected_good = 0;
    for (size_t i = 0; i < desc.size(); ++i) {
        double d1 = 1e300, d2 = 1e300;
        size_t i1 = 0;
        for (size_t p = 0; p < feat.size(); ++p) {
            double d = 0;
            for (int j = 0; j < 128; ++j) { const double t = (double)desc[i][j] - (double)feat[p][j]; d += t * t; }
            if (d < d1) { d2 = d1; d1 = d; i1 = p; } else if (d < d2) d2 = d;
        }
        const bool is_good = (float)d1 / (float)d2 < 0.7f;
        expected_good += is_good;
        if (is_good) {
            CHECK(g < good.size() && good[g].lindex == i && good[g].mindex == i1);
            if (g < good.size()) CHECK(std::fabs(good[g].distance - (float)d1) <= 1e-6f * (float)d1);
            ++g;
        }
    }
    CHECK(g == good.size() && expected_good >= 20 && expected_good <= 40);
}

// The reference's own call sequence (pose_estimation.cpp:72-98, 139, 158): a Bundler file and Lowe
// .key files on disk -> parse_bundler -> QueryProcessor(parse_bundler&) -> QueryLoader ->
// qp->Process(query) with nothing else handed over, and Benchmark(ql, golden_pb, qp).  The pose must
// be the one the pre-matched entry gives on DescriptorMatcher::Match's list.
static void write_key(const std::string &path, const std::vector<SIFT_keypoint> &keys, const std::vector<unsigned char> &desc)
{
    std::ofstream f(path);
    f << keys.size() << " 128\n";
    for (size_t i = 0; i < keys.size(); ++i) {
        f << keys[i].y << " " << keys[i].x << " 1.5 0.25\n";            // Lowe: y before x
        for (int j = 0; j < 128; ++j) f << " " << (int)desc[i * 128 + j] << ((j % 20 == 19) ? "\n" : "");
        f << "\n";
    }
}

static void test_process_query_only(const std::string &dir)
{
    const int w = 1600, h = 1200, ncam = 12, npts = 1200, ninl = 240, nwrong = 100, nrandom = 60;
    const double f = 1200.0, C[3] = {3.0, -2.0, 1.0};
    std::mt19937 rng(2024);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    std::uniform_int_distribution<int> byte(0, 90), jitter(-3, 3);
    auto noisy = [&](const std::vector<int> &proto, std::vector<unsigned char> &out) {
        for (int j = 0; j < 128; ++j) out.push_back((unsigned char)std::max(0, std::min(255, proto[(size_t)j] + jitter(rng))));
    };
    // model: points in front of the planted camera, 2-6 views each, one .key file per camera
    std::vector<std::vector<int>> proto((size_t)npts, std::vector<int>(128));
    std::vector<std::vector<SIFT_keypoint>> cam_keys((size_t)ncam);
    std::vector<std::vector<unsigned char>> cam_desc((size_t)ncam);
    std::ofstream bundle(dir + "/model.out");
    bundle << "# Bundle file v0.3\n" << ncam + 1 << " " << npts << "\n";
    for (int c = 0; c < ncam; ++c) bundle << "1000 0 0\n1 0 0\n0 1 0\n0 0 1\n0 0 0\n";
    // the query's own camera as the reconstruction knows it (Bundler looks down -Z): R = diag(-1,1,-1), t = -R C
    bundle << std::setprecision(17) << f << " 0 0\n-1 0 0\n0 1 0\n0 0 -1\n" << C[0] << " " << -C[1] << " " << C[2] << "\n";
    std::vector<float> px((size_t)npts), py((size_t)npts);
    for (int i = 0; i < npts; ++i) {
        for (int &v : proto[(size_t)i]) v = byte(rng);
        const double z = 10.0 + 95.0 * (u(rng) + 1.0);
        const double ix = 0.45 * w * u(rng), iy = 0.45 * h * u(rng);
        const float X = (float)(C[0] + ix * z / f), Y = (float)(C[1] + iy * z / f), Z = (float)(C[2] + z);
        px[(size_t)i] = (float)(w / 2.0 - ix); py[(size_t)i] = (float)(h / 2.0 - iy);        // keypoint of an exact observation
        const int nv = i % 9 == 0 ? 6 : 2 + (int)(rng() % 3);
        bundle << std::setprecision(9) << X << " " << Y << " " << Z << "\n128 128 128\n" << nv;
        for (int k = 0; k < nv; ++k) {
            const int c = (i + 5 * k) % ncam;                                             // distinct cameras for nv <= 6
            bundle << " " << c << " " << cam_keys[(size_t)c].size() << " 0 0";
            cam_keys[(size_t)c].push_back(SIFT_keypoint());
            noisy(proto[(size_t)i], cam_desc[(size_t)c]);
        }
        bundle << "\n";
    }
    bundle.close();
    std::vector<std::string> key_files;
    for (int c = 0; c < ncam; ++c) {
        key_files.push_back(dir + "/cam" + std::to_string(c) + ".key");
        write_key(key_files.back(), cam_keys[(size_t)c], cam_desc[(size_t)c]);
    }
    key_files.push_back(dir + "/no_such_file.key");                                       // nobody's view names the query camera
    // the query: inliers observe their point, wrong matches carry a model descriptor at a random place, the rest is noise
    std::vector<SIFT_keypoint> qkeys;
    std::vector<unsigned char> qdesc;
    for (int i = 0; i < ninl + nwrong + nrandom; ++i) {
        SIFT_keypoint k;
        if (i < ninl) { k.x = px[(size_t)(3 * i)]; k.y = py[(size_t)(3 * i)]; noisy(proto[(size_t)(3 * i)], qdesc); }
        else {
            k.x = (float)(w / 2.0 + 0.5 * w * u(rng)); k.y = (float)(h / 2.0 + 0.5 * h * u(rng));
            if (i < ninl + nwrong) noisy(proto[(size_t)(3 * (i - ninl) + 1)], qdesc);
            else for (int j = 0; j < 128; ++j) qdesc.push_back((unsigned char)byte(rng));
        }
        qkeys.push_back(k);
    }
    write_key(dir + "/q0.key", qkeys, qdesc);
    { std::ofstream l(dir + "/list.query.fused.txt"); l << "query/q0.jpg " << w << " " << h << "\n"; }          // no focal length: P4Pf
    {
        std::ofstream l(dir + "/list.orig.txt");
        for (int c = 0; c < ncam; ++c) l << "db/c" << c << ".jpg 0 1000\n";
        l << "query/q0.jpg 0 " << f << "\n";
    }

    parse_bundler pb;
    CHECK(pb.parse_data((dir + "/model.out").c_str()));
    CHECK(pb.load_descriptors(key_files));
    CHECK(pb.get_feature_infos().size() == (size_t)npts && pb.get_feature_infos()[0].view_list.size() == 6);
    QueryLoader ql(dir + "/", dir + "/");
    CHECK(ql.GetQueryCount() == 1);
    for (int advanced = 0; advanced < 2; ++advanced) {
        QueryProcessorBasic basic(pb);
        QueryProcessorAdvanced adv(pb);
        QueryProcessor *qp = advanced ? (QueryProcessor *)&adv : (QueryProcessor *)&basic;      // the reference's `QueryProcessor *qp`
        qp->seed = 31;
        Query query = ql.GetQuery(0, false);
        CHECK(query.image_width() == w && query.focal_length() == 0.0 && query.sift_descriptors().size() == qkeys.size());
        CHECK(qp->Process(query));                                                      // pose_estimation.cpp:139
        CHECK(qp->result.found == 1 && qp->result.inliers >= (advanced ? 100u : 12u));
        // the match list: every inlier and every wrong match passes the ratio test, noise does not
        CHECK(qp->good_matches.size() >= (size_t)(ninl + nwrong) && qp->good_matches.size() <= (size_t)(ninl + nwrong + 5));
        CHECK(qp->good_matches[0].lindex == 0 && qp->good_matches[0].mindex == 0 && qp->good_matches[1].mindex == 3);
        if (advanced) CHECK(qp->good_matches[0].camset.size() == 6 && qp->good_matches[1].camset == pb.camset(3));
        else CHECK(qp->good_matches[0].camset.empty());
        const Vector3d Cq = query.camera_position();
        CHECK(std::sqrt((Cq(0, 0) - C[0]) * (Cq(0, 0) - C[0]) + (Cq(1, 0) - C[1]) * (Cq(1, 0) - C[1]) + (Cq(2, 0) - C[2]) * (Cq(2, 0) - C[2])) < 3.0);
        // the pre-matched entry on the matcher's own list: the identical pose
        DescriptorMatcher matcher(pb.get_feature_infos());
        Query again = ql.GetQuery(0, false);
        std::vector<SiftMatch> good = matcher.Match(again.sift_descriptors(), advanced ? 0.9f : 0.7f);
        if (advanced) for (SiftMatch &m : good) m.camset = pb.camset(m.mindex);
        CHECK(good.size() == qp->good_matches.size());
        QueryProcessorBasic pre_basic;
        QueryProcessorAdvanced pre_adv;
        QueryProcessor *pre = advanced ? (QueryProcessor *)&pre_adv : (QueryProcessor *)&pre_basic;
        pre->seed = 31;
        CHECK(pre->Process(again, good, pb.get_feature_infos()));
        CHECK(pre->result.steps == qp->result.steps && pre->result.best_hyp == qp->result.best_hyp && pre->result.inliers == qp->result.inliers);
        const Vector3d Ca = again.camera_position();
        CHECK(Ca(0, 0) == Cq(0, 0) && Ca(1, 0) == Cq(1, 0) && Ca(2, 0) == Cq(2, 0) && again.focal_length() == query.focal_length());
        bool threw = false;
        try { pre->Process(again); } catch (const std::logic_error &) { threw = true; }
        CHECK(threw);                                                                   // no model: Process(Query&) cannot match
        // benchmark.hpp:17 with the reference's arguments; the golden pose is the planted one
        const std::string out = dir + "/benchmark.txt";
        std::vector<BenchmarkRow> rows = Benchmark(ql, pb, qp, dir + "/", out, nullptr);
        CHECK(rows.size() == 1 && rows[0].index == 1 && rows[0].position_error < 3.0 && rows[0].rotation_error < 0.1 && std::fabs(rows[0].focal_length_error) < 60.0);
        std::ifstream in(out);
        std::string line;
        CHECK((bool)std::getline(in, line) && std::count(line.begin(), line.end(), ',') == 4);
    }
}

// BundleAdjust with local_optimisation: the default leaves Process exactly as the reference's no-op
// does; switched on, 64 more steps sample among the fitted matches only, and the kept pose can only
// get better by the processor's own criterion.
static void test_local_optimisation()
{
    for (int known = 0; known < 2; ++known) {
        Scene a = make_scene(900, 300, known != 0, 90u + known), b = make_scene(900, 300, known != 0, 90u + known);
        QueryProcessorBasic plain, lo;
        plain.seed = lo.seed = 17;
        lo.local_optimisation = true;
        CHECK(plain.Process(a.query, a.matches, a.sifts) && lo.Process(b.query, b.matches, b.sifts));
        CHECK(plain.lo_samples.empty() && !plain.lo_improved);
        CHECK(lo.samples == plain.samples && lo.result.steps == plain.result.steps);          // the loop itself is untouched
        CHECK(lo.lo_samples.size() == 4 * (size_t)lo.lo_steps);
        // every tuple index names a match the loop's pose had fitted
        std::set<unsigned> fitted;
        for (const SiftMatch &m : plain.fitted_matches) fitted.insert(m.lindex);
        bool inside = true;
        for (size_t i = 0; i < lo.lo_samples.size(); ++i)
            if ((i % 4) < (known ? 3u : 4u)) inside = inside && fitted.count(b.matches[lo.lo_samples[i]].lindex) == 1;
        CHECK(inside);
        const float q0 = (float)plain.result.area_fit / (float)plain.result.area_all, q1 = (float)lo.result.area_fit / (float)lo.result.area_all;
        CHECK(q1 >= q0 && (lo.lo_improved == (q1 > q0)));
        CHECK(lo.fitted_matches.size() == lo.result.inliers && b.query.fitpoints_2d().size() == lo.result.inliers);
        if (!lo.lo_improved) CHECK(lo.result.best_hyp == plain.result.best_hyp && b.query.camera_position()(0, 0) == a.query.camera_position()(0, 0));
        else CHECK(lo.result.inliers >= 12);
        std::printf("local optimisation (%s): quality %.4f -> %.4f, inliers %u -> %u\n", known ? "P3P" : "P4Pf", q0, q1, plain.result.inliers, lo.result.inliers);
    }
}

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    try {
        test_p3p();
        test_p4pf();
        test_processors();
        test_batch();
        test_project();
        test_backmatching();
        test_benchmark();
        test_matching();
        test_process_query_only(dir);
        test_local_optimisation();
    } catch (const std::exception &e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 2;
    }
    std::printf(failures ? "%d check(s) failed\n" : "facade ok\n", failures);
    return failures ? 1 : 0;
}
