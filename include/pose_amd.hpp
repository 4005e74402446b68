// C++ host facade over the C ABI of libpose_amd (include/pose_amd.h).
//
// It keeps the class interface the reference's host code is written against, so that the
// reference's call sites compile unchanged once its solver/processor headers are replaced:
//   P3p::computePoses / solveQuartic                     (P3p.hpp:59-60)
//   P4pf::P4Pf_m / P4Pf                                  (P4pf.hpp:39-41)
//   QueryProcessor::Project, TestHypothesis, QueryProcessorBasic/Advanced::Process
//                                                        (query_processor.hpp:33-73)
//   QueryLoader / Query (list file, .key files)          (query_loader.hpp:30-91)
// The reference passes Eigen matrices by value; Eigen is not a dependency here, so the small
// fixed-size type Mat<R,C> of pose_amd_types.hpp is used instead.
//
// Error behaviour follows the reference: the solvers return 0 / -1 and never throw; Process
// always returns true.  A missing GPU or a failed HIP call is not a reference condition: it
// throws std::runtime_error carrying pose_amd_last_error().  There is no CPU fallback.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "pose_amd.h"
#include "pose_amd_types.hpp"

namespace pose_estimation_amd {

// One context per device, created on first use (the reference creates one solver object per
// Process call, basic.cpp:100-101; contexts are heavier, so they are shared per thread).
class Device {
public:
    explicit Device(int index = 0)
    {
        if (pose_amd_create(index, &ctx_) != POSE_AMD_OK) throw std::runtime_error(std::string("pose_amd_create: ") + pose_amd_last_error(nullptr));
    }
    ~Device() { if (ctx_) pose_amd_destroy(ctx_); }
    Device(const Device &) = delete;
    Device &operator=(const Device &) = delete;
    pose_amd_ctx *ctx() const { return ctx_; }
    void check(int rc, const char *what) const
    {
        if (rc != POSE_AMD_OK) throw std::runtime_error(std::string(what) + ": " + pose_amd_last_error(ctx_));
    }
    static Device &current()
    {
        static thread_local Device dev(0);
        return dev;
    }

private:
    pose_amd_ctx *ctx_ = nullptr;
};

// ---- P3p.hpp:54-61 ----------------------------------------------------------------------------
class P3p {
public:
    explicit P3p(Device *dev = nullptr) : dev_(dev ? dev : &Device::current()) {}
    // featureVectors, worldPoints: one vector per column.  solutions: 4 blocks [C | R].
    int computePoses(Mat<3, 3> featureVectors, Mat<3, 3> worldPoints, Mat<3, 16> &solutions)
    {
        int status = 0;
        double sol[48];
        // the ABI takes one vector after the other = the columns of a column-major matrix
        dev_->check(pose_amd_p3p_batch(dev_->ctx(), 1, featureVectors.data(), worldPoints.data(), sol, &status), "p3p_batch");
        if (status != 0) return -1;
        for (int i = 0; i < 4; ++i) {
            const double *s = sol + 12 * i;      // C[3], Rc[9] row-major
            for (int r = 0; r < 3; ++r) {
                solutions(r, 4 * i) = s[r];
                for (int c = 0; c < 3; ++c) solutions(r, 4 * i + 1 + c) = s[3 + 3 * r + c];
            }
        }
        return 0;
    }
    int solveQuartic(Mat<5, 1> factors, Mat<4, 1> &realRoots)
    {
        dev_->check(pose_amd_quartic_batch(dev_->ctx(), 1, factors.data(), realRoots.data()), "quartic_batch");
        return 0;
    }

private:
    Device *dev_;
};

// ---- P4pf.hpp:28-59 ---------------------------------------------------------------------------
class P4pf {
public:
    explicit P4pf(Device *dev = nullptr) : dev_(dev ? dev : &Device::current()) {}
    // m2D: 2x4 image points (centred), M3D: 3x4 world points; results are appended.
    int P4Pf_m(Mat<2, 4> m2D, Mat<3, 4> M3D, std::vector<double> *focalLengths_m, std::vector<Matrix3d> *rotationMatrices,
               std::vector<Vector3d> *translationResult)
    {
        int nsol = 0;
        double f[POSE_AMD_P4PF_MAXSOL], R[9 * POSE_AMD_P4PF_MAXSOL], t[3 * POSE_AMD_P4PF_MAXSOL];
        dev_->check(pose_amd_p4pf_batch(dev_->ctx(), 1, m2D.data(), M3D.data(), &nsol, f, R, t), "p4pf_batch");
        if (nsol < 0) return -1;
        for (int i = 0; i < nsol; ++i) {
            focalLengths_m->push_back(f[i]);
            Matrix3d Ri;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) Ri(r, c) = R[9 * i + 3 * r + c];
            rotationMatrices->push_back(Ri);
            translationResult->push_back(Vector3d(t + 3 * i));
        }
        return 0;
    }
    // The reference's Gauss-Jordan variant solves the same system (P4pf.cpp:558-854).
    int P4Pf(Mat<2, 4> m2D, Mat<3, 4> M3D, std::vector<double> *f, std::vector<Matrix3d> *R, std::vector<Vector3d> *t)
    {
        return P4Pf_m(m2D, M3D, f, R, t);
    }

private:
    Device *dev_;
};

// The matching step of QueryProcessorBasic (query_processor_basic.cpp:22-53 and :60-96): one float
// descriptor per 3-D point -- the mean of the descriptors of its views -- and, per query
// descriptor, the two nearest of them by squared L2 distance with Lowe's ratio test.  The search is
// exact where the reference's FLANN kd-tree search is approximate.
class DescriptorMatcher {
public:
    explicit DescriptorMatcher(const std::vector<feature_3D_info> &sifts, Device *dev = nullptr) : dev_(dev ? dev : &Device::current())
    {
        std::vector<uint32_t> off(sifts.size() + 1, 0);
        for (size_t i = 0; i < sifts.size(); ++i) off[i + 1] = off[i] + (uint32_t)(sifts[i].descriptors.size() / 128);
        std::vector<unsigned char> all((size_t)off.back() * 128 + 1);
        for (size_t i = 0; i < sifts.size(); ++i)
            std::copy(sifts[i].descriptors.begin(), sifts[i].descriptors.begin() + (size_t)(off[i + 1] - off[i]) * 128, all.begin() + (size_t)off[i] * 128);
        dev_->check(pose_amd_knn_set_model_views(dev_->ctx(), (int)sifts.size(), off.data(), all.data(), nullptr), "knn_set_model_views");
    }
    // descriptors: 128 bytes per query keypoint (Query::sift_descriptors()); returns good_matches
    // (lindex = keypoint, mindex = nearest 3-D point, distance = squared L2 distance to it)
    std::vector<SiftMatch> Match(const std::vector<unsigned char *> &descriptors, float ratio = 0.7f) const
    {
        const size_t n = descriptors.size();
        std::vector<unsigned char> q(n * 128 + 1);
        for (size_t i = 0; i < n; ++i) std::copy(descriptors[i], descriptors[i] + 128, q.begin() + i * 128);
        std::vector<int> idx(2 * n + 1);
        std::vector<float> dist(2 * n + 1);
        std::vector<unsigned char> good(n + 1);
        dev_->check(pose_amd_knn_match(dev_->ctx(), (int)n, q.data(), ratio, idx.data(), dist.data(), good.data(), nullptr), "knn_match");
        std::vector<SiftMatch> out;
        for (size_t i = 0; i < n; ++i) {
            if (!good[i]) continue;
            SiftMatch m;
            m.lindex = (unsigned)i; m.mindex = (unsigned)idx[2 * i]; m.distance = dist[2 * i];
            out.push_back(m);
        }
        return out;
    }

private:
    Device *dev_;
};

// query_loader.hpp:30-75.  The image itself (cv::Mat, LoadImage) is not part of the path.
class QueryLoader;
class Query {
public:
    Query(int width, int height, double focal_length = 0.0) : focal_(focal_length), w_(width), h_(height)
    {
        rot_(0, 0) = rot_(1, 1) = rot_(2, 2) = 1.0;
    }
    // query_loader.cpp:84-88: a query of a list, its features are read by LoadSIFT
    Query(QueryLoader *query_loader, std::string query_image_name, double focal_length)
        : focal_(focal_length), w_(0), h_(0), loader_(query_loader), name_(std::move(query_image_name))
    {
        rot_(0, 0) = rot_(1, 1) = rot_(2, 2) = 1.0;
    }
    double focal_length() const { return focal_; }
    Vector3d camera_position() const { return pos_; }
    Matrix3d camera_rotation() const { return rot_; }
    Mat<3, 4> proj_matrix() const { return proj_; }
    void set_camera_position(Vector3d p) { pos_ = p; }
    void set_camera_rotation(Matrix3d r) { rot_ = r; }
    void set_focal_length(double f) { focal_ = f; }
    int image_width() const { return w_; }
    int image_height() const { return h_; }
    void set_image_width(int w) { w_ = w; }
    void set_image_height(int h) { h_ = h; }
    const std::string &image_name() const { return name_; }
    std::vector<SIFT_keypoint> &sift_keypoints() { return keys_; }
    // One pointer per keypoint to its 128 descriptor bytes (query_loader.hpp:47).  The bytes are owned
    // by the query and shared by its copies, so the pointers of a copy stay valid (the reference
    // copies queries freely: basic.cpp:59, benchmark.cpp:28).
    std::vector<unsigned char *> &sift_descriptors() { return desc_; }
    // Takes over keypoints and their descriptors (128 bytes per keypoint).
    void set_sift(const std::vector<SIFT_keypoint> &keypoints, const std::vector<unsigned char> &descriptors)
    {
        keys_ = keypoints;
        bytes_ = std::make_shared<std::vector<unsigned char>>(descriptors);
        bytes_->resize(keys_.size() * 128);
        desc_.resize(keys_.size());
        for (size_t i = 0; i < keys_.size(); ++i) desc_[i] = bytes_->data() + i * 128;
    }
    // query_loader.cpp:153-177: reads <query_path><name>.key (the reference first tries to gunzip it
    // from the database directory through system(); this loader reads only).  Defined behind QueryLoader.
    inline bool LoadSIFT();
    // projections / world points of the fitted matches, filled by Process (query_loader.hpp:76-79)
    std::vector<Vector2d> &fitpoints_2d() { return fit2d_; }
    std::vector<Vector3d> &fitpoints_3d() { return fit3d_; }
    void set_fitpoints_2d(const std::vector<Vector2d> &p) { fit2d_ = p; }
    void set_fitpoints_3d(const std::vector<Vector3d> &p) { fit3d_ = p; }
    // query_loader.cpp:222-254
    bool ComputeMatrices(Device *dev = nullptr)
    {
        Device *d = dev ? dev : &Device::current();
        double Rc[9], P[12];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) Rc[3 * r + c] = rot_(r, c);
        d->check(pose_amd_compute_matrices_batch(d->ctx(), 1, &focal_, pos_.data(), Rc, P), "compute_matrices_batch");
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) proj_(r, c) = P[4 * r + c];
        return true;
    }

private:
    double focal_;
    Vector3d pos_;
    Matrix3d rot_;
    Mat<3, 4> proj_;
    std::vector<SIFT_keypoint> keys_;
    std::vector<unsigned char *> desc_;
    std::shared_ptr<std::vector<unsigned char>> bytes_;
    std::vector<Vector2d> fit2d_;
    std::vector<Vector3d> fit3d_;
    int w_, h_;
    QueryLoader *loader_ = nullptr;
    std::string name_;
};

// query_loader.hpp:77-91, query_loader.cpp:28-82: the list "<database_path>list.query.fused.txt" with
// one query per line, "query/<name>.jpg <width> <height> [<focal length>]"; the features of query
// <name> are read from "<query_path><name>.key" when the query is asked for.
class QueryLoader {
public:
    QueryLoader(std::string database_path, std::string query_path) : database_path_(std::move(database_path)), query_path_(std::move(query_path))
    {
        std::ifstream queryfile(database_path_ + "list.query.fused.txt");
        std::string line;
        while (std::getline(queryfile, line)) {
            std::vector<std::string> f;
            size_t pos = 0;
            while (pos <= line.size()) {
                const size_t e = line.find(' ', pos);
                f.push_back(line.substr(pos, e == std::string::npos ? std::string::npos : e - pos));
                if (e == std::string::npos) break;
                pos = e + 1;
            }
            if (f.size() < 3) continue;
            const double focal = f.size() == 4 ? std::atof(f[3].c_str()) : 0.0;
            // the name between the first '/' and the first '.' behind it (CutString, pose_utils.cpp:36-56)
            const size_t slash = f[0].find('/');
            const size_t dot = slash == std::string::npos ? std::string::npos : f[0].find('.', slash + 1);
            if (slash == std::string::npos || dot == std::string::npos || dot == slash + 1) continue;
            Query query(this, f[0].substr(slash + 1, dot - slash - 1), focal);
            query.set_image_width(std::atoi(f[1].c_str()));
            query.set_image_height(std::atoi(f[2].c_str()));
            queries_.push_back(query);
        }
    }
    std::string database_path() const { return database_path_; }
    std::string query_path() const { return query_path_; }
    int GetQueryCount() const { return (int)queries_.size(); }
    // the features are read when a query is first asked for; load_image is accepted and ignored
    Query &GetQuery(int index, bool /*load_image*/)
    {
        queries_[(size_t)index].LoadSIFT();
        return queries_[(size_t)index];
    }

private:
    std::string database_path_, query_path_;
    std::vector<Query> queries_;
};

inline bool Query::LoadSIFT()
{
    if (!loader_) return false;
    SIFT_loader siftl;
    if (!siftl.load_Lowe_features((loader_->query_path() + name_ + ".key").c_str())) return false;
    set_sift(siftl.get_keypoints(), siftl.get_descriptors());
    return true;
}

// ---- query_processor.hpp:33-73 ------------------------------------------------------------------
// Two ways in.  A processor built on a parse_bundler, as in the reference (query_processor.hpp:35),
// averages the model's descriptors and holds them on the device; its Process(Query&) is the
// reference's: descriptors -> two nearest -> ratio test -> RANSAC.  A processor built without a
// model takes the `good_matches` list the reference has at basic.cpp:105 / advanced.cpp:296 together
// with the 3-D points they refer to (the entry for callers with a matching stage of their own).
class QueryProcessor {
public:
    explicit QueryProcessor(Device *dev = nullptr) : dev_(dev ? dev : &Device::current()) {}
    // query_processor_basic.cpp:22-55 / query_processor_advanced.cpp:24-56 (both build the same index)
    explicit QueryProcessor(parse_bundler &parsebundler, Device *dev = nullptr)
        : dev_(dev ? dev : &Device::current()), parsebundler_(&parsebundler), sifts_(&parsebundler.get_feature_infos()),
          matcher_(new DescriptorMatcher(parsebundler.get_feature_infos(), dev_))
    {
    }
    virtual ~QueryProcessor() {}
    // query_processor.hpp:36.  Throws std::logic_error on a processor built without a model.
    virtual bool Process(Query &query) = 0;
    virtual bool Process(Query &query, std::vector<SiftMatch> &good_matches, std::vector<feature_3D_info> &sifts) = 0;
    // good_matches of the last Process(Query&)
    std::vector<SiftMatch> good_matches;

    // Reprojection of `matches` with query.proj_matrix() (query_processor.cpp:46-86).
    bool Project(Query &query, std::vector<Vector2d> &fitpoints_2d, std::vector<Vector3d> &fitpoints_3d, std::vector<SiftMatch> &matches,
                 std::vector<SIFT_keypoint> &sift_keypoints, std::vector<feature_3D_info> &sifts)
    {
        const size_t n = matches.size();
        fitpoints_2d.resize(n);
        fitpoints_3d.resize(n);
        if (n == 0) return true;
        solver_focal_ = query.focal_length();
        upload(query, matches, sift_keypoints, sifts, false);
        const Mat<3, 4> Pm = query.proj_matrix();
        double P[12];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) P[4 * r + c] = Pm(r, c);
        std::vector<double> p2(2 * n), p3(3 * n);
        dev_->check(pose_amd_project_batch(dev_->ctx(), 1, P, p2.data(), p3.data()), "project_batch");
        for (size_t i = 0; i < n; ++i) { fitpoints_2d[i] = Vector2d(p2.data() + 2 * i); fitpoints_3d[i] = Vector3d(p3.data() + 3 * i); }
        return true;
    }

    // query_processor.cpp:36-44: the reference's refinement step is an empty function that returns
    // true (its Levenberg-Marquardt is a TODO there); Process calls it between ComputeMatrices and
    // the final Project (basic.cpp:299, advanced.cpp:613).  By default it does the same here.
    // With `local_optimisation` set it does what the report's outlook names as the next step, a local
    // optimisation on the fitted matches: `lo_steps` more RANSAC steps whose minimal samples are drawn
    // from the fitted matches only, every pose scored against ALL matches with the processor's own
    // criterion; the kept pose, fitted_matches and the fit points change only if a pose is strictly
    // better.  The reference has nothing to compare this with (parity unpinned); it only acts
    // inside Process, which knows the full match list.
    bool local_optimisation = false;
    uint32_t lo_steps = 64;
    std::vector<uint32_t> lo_samples;       // the tuples of the last local optimisation (indices into good_matches)
    bool lo_improved = false;
    bool BundleAdjust(Query &query, std::vector<SiftMatch> &fitted, std::vector<SIFT_keypoint> &sift_keypoints, std::vector<feature_3D_info> &sifts)
    {
        lo_samples.clear();
        lo_improved = false;
        const int m = solver_focal_ == 0.0 ? 4 : 3;
        if (!local_optimisation || !cur_matches_ || !result.found || fitted.size() < 8 || lo_steps == 0) return true;
        // positions of the fitted matches in the full list (fitted keeps the list's order)
        std::vector<uint32_t> pos;
        {
            size_t k = 0;
            for (size_t i = 0; i < cur_matches_->size() && k < fitted.size(); ++i)
                if ((*cur_matches_)[i].lindex == fitted[k].lindex && (*cur_matches_)[i].mindex == fitted[k].mindex) { pos.push_back((uint32_t)i); ++k; }
            if (k != fitted.size()) return true;
        }
        // tuples among the fitted matches from the counter-based generator (a stream range of its own)
        upload(query, fitted, sift_keypoints, sifts, false);
        std::vector<uint32_t> t((size_t)lo_steps * 4);
        dev_->check(pose_amd_sample(dev_->ctx(), POSE_AMD_SAMPLE_BASIC, seed, query_id, 0x40000000u, lo_steps, t.data()), "sample");
        for (size_t i = 0; i < t.size(); ++i) t[i] = (i % 4) < (size_t)m ? pos[t[i]] : 0u;
        lo_samples = t;
        upload(query, *cur_matches_, sift_keypoints, sifts, false);
        dev_->check(pose_amd_set_samples(dev_->ctx(), t.data(), lo_steps), "set_samples");
        const uint32_t steps_before = result.steps;
        lo_improved = decide(query, *cur_matches_, 0);
        result.steps = steps_before;                  // the loop's own step count stays what the reference counts
        return true;
    }

    // Inliers of query.proj_matrix() among `matches` and the coverage quality (query_processor.cpp:88-177).
    std::vector<SiftMatch> TestHypothesis(Query &query, std::vector<SiftMatch> &matches, std::vector<SIFT_keypoint> &sift_keypoints,
                                          std::vector<feature_3D_info> &sifts, double match_eps, double &fit_quality)
    {
        solver_focal_ = query.focal_length();
        upload(query, matches, sift_keypoints, sifts, false);
        const int n = (int)matches.size();
        const Mat<3, 4> Pm = query.proj_matrix();
        double P[12];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) P[4 * r + c] = Pm(r, c);
        int count = 0;
        std::vector<uint64_t> words((size_t)(n + 63) / 64);
        dev_->check(pose_amd_score_batch(dev_->ctx(), 1, P, match_eps, &count, words.data()), "score_batch");
        int64_t area = 0, area_all = 0;
        dev_->check(pose_amd_coverage_batch(dev_->ctx(), 1, P, match_eps, &area, &area_all), "coverage_batch");
        fit_quality = (double)((float)area / (float)area_all);
        std::vector<SiftMatch> fitted;
        for (int i = 0; i < n; ++i)
            if ((words[(size_t)i >> 6] >> (i & 63)) & 1u) fitted.push_back(matches[(size_t)i]);
        return fitted;
    }

    // what the last Process consumed / produced, for parity checks on identical samples
    std::vector<uint32_t> samples;          // 4 indices per step
    pose_amd_result result{};
    std::vector<SiftMatch> fitted_matches;

protected:
    void upload(Query &query, std::vector<SiftMatch> &matches, std::vector<SIFT_keypoint> &keys, std::vector<feature_3D_info> &sifts,
                bool camsets)
    {
        const size_t n = matches.size();
        std::vector<float> kx(n), ky(n), X(n), Y(n), Z(n);
        std::vector<uint32_t> lindex(n), off(n + 1, 0), ids;
        for (size_t i = 0; i < n; ++i) {
            const SiftMatch &m = matches[i];
            kx[i] = keys[m.lindex].x; ky[i] = keys[m.lindex].y;
            const point3D &p = sifts[m.mindex].point;
            X[i] = p.x; Y[i] = p.y; Z[i] = p.z;
            lindex[i] = m.lindex;
            if (camsets) ids.insert(ids.end(), m.camset.begin(), m.camset.end());
            off[i + 1] = (uint32_t)ids.size();
        }
        dev_->check(pose_amd_set_query(dev_->ctx(), (int)n, kx.data(), ky.data(), X.data(), Y.data(), Z.data(), lindex.data(),
                                       query.image_width(), query.image_height(), solver_focal_),
                    "set_query");
        if (camsets) dev_->check(pose_amd_set_camsets(dev_->ctx(), off.data(), ids.data()), "set_camsets");
    }

    // sample `steps` tuples, run the loop, store the kept pose in `query`
    bool run(Query &query, std::vector<SiftMatch> &matches, int sample_mode, uint32_t hyp_begin, uint32_t steps, int early_exit)
    {
        const size_t old = samples.size();
        samples.resize(old + (size_t)steps * 4);
        dev_->check(pose_amd_sample(dev_->ctx(), sample_mode, seed, query_id, hyp_begin, steps, samples.data() + old), "sample");
        return decide(query, matches, early_exit);
    }

    // the loop over the tuples the context holds; a pose that beats the kept one replaces it
    bool decide(Query &query, std::vector<SiftMatch> &matches, int early_exit)
    {
        pose_amd_params prm;
        std::memset(&prm, 0, sizeof prm);
        prm.select_mode = select_mode; prm.eps = inlier_eps; prm.early_exit = early_exit;
        prm.inlier_divisor = inlier_divisor; prm.inlier_absolute = inlier_absolute;
        pose_amd_result r;
        dev_->check(pose_amd_ransac_run(dev_->ctx(), &prm, &r, nullptr, nullptr), "ransac_run");
        const bool better = r.found && (!result.found || (select_mode == POSE_AMD_SELECT_COVERAGE
                                                              ? (float)r.area_fit / (float)r.area_all > (float)result.area_fit / (float)result.area_all
                                                              : r.inliers > result.inliers));
        r.steps += result.steps;
        if (!better) { result.steps = r.steps; return false; }
        result = r;
        // fitted matches and their final projection (basic.cpp:296-307) in one call
        const size_t n = matches.size();
        std::vector<uint8_t> mask(n);
        std::vector<double> p2(2 * n), p3(3 * n);
        int nfit = 0;
        dev_->check(pose_amd_best_fit(dev_->ctx(), mask.data(), p2.data(), p3.data(), &nfit), "best_fit");
        fitted_matches.clear();
        for (size_t i = 0; i < n; ++i)
            if (mask[i]) fitted_matches.push_back(matches[i]);
        fit2d_.resize((size_t)nfit); fit3d_.resize((size_t)nfit);
        for (int i = 0; i < nfit; ++i) { fit2d_[(size_t)i] = Vector2d(p2.data() + 2 * i); fit3d_[(size_t)i] = Vector3d(p3.data() + 3 * i); }
        if (solver_focal_ == 0.0) query.set_focal_length(r.focal);      // unknown focal: the P4Pf estimate
        query.set_camera_position(Vector3d(r.C));
        Matrix3d Rc;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) Rc(a, b) = r.Rc[3 * a + b];
        query.set_camera_rotation(Rc);
        return true;
    }

public:
    // RANSAC parameters (basic.cpp:107-119) and the stream of the counter-based generator that
    // replaces the time-seeded mt19937_64 (pose_utils.cpp:58-84)
    double inlier_eps = 0.5;
    uint32_t inlier_divisor = 10, inlier_absolute = 12;
    int select_mode = POSE_AMD_SELECT_COVERAGE;
    uint64_t seed = 0;
    uint32_t query_id = 0;

protected:
    // what the reference does after its loop (basic.cpp:296-307): final matrices, reprojected fitted matches
    void finish(Query &query, std::vector<SiftMatch> &matches, std::vector<feature_3D_info> &sifts)
    {
        query.ComputeMatrices(dev_);
        cur_matches_ = &matches;
        std::vector<SiftMatch> fitted = fitted_matches;          // BundleAdjust may replace fitted_matches
        BundleAdjust(query, fitted, query.sift_keypoints(), sifts);
        cur_matches_ = nullptr;
        if (lo_improved) query.ComputeMatrices(dev_);
        query.set_fitpoints_2d(fit2d_);
        query.set_fitpoints_3d(fit3d_);
    }
    std::vector<SiftMatch> *cur_matches_ = nullptr;          // the full match list while Process runs
    // basic.cpp:60-96 / advanced.cpp:242-285: two nearest model descriptors of every query descriptor,
    // Lowe's ratio test on the squared distances; with_camsets fills SiftMatch::camset from the view lists
    void match(Query &query, float ratio, bool with_camsets)
    {
        if (!matcher_) throw std::logic_error("QueryProcessor::Process(Query&): the processor was built without a parse_bundler");
        good_matches = matcher_->Match(query.sift_descriptors(), ratio);
        if (with_camsets)
            for (SiftMatch &m : good_matches)
                for (const view &v : (*sifts_)[m.mindex].view_list) m.camset.insert(v.camera);
    }
    std::vector<Vector2d> fit2d_;
    std::vector<Vector3d> fit3d_;
    Device *dev_;
    double solver_focal_ = 0.0;             // focal length the query came with: 0 selects P4Pf (basic.cpp:154)
    parse_bundler *parsebundler_ = nullptr;
    std::vector<feature_3D_info> *sifts_ = nullptr;
    std::shared_ptr<DescriptorMatcher> matcher_;
};

class QueryProcessorBasic : public QueryProcessor {
public:
    using QueryProcessor::QueryProcessor;
    uint32_t max_ransac_steps = 10000;
    std::vector<int> devices;    // ProcessBatch: GPUs to spread the queries over (empty: this processor's device only)
    float ratio = 0.7f;          // basic.cpp:87
    // query_processor_basic.cpp:57-310
    bool Process(Query &query) override
    {
        match(query, ratio, false);
        return Process(query, good_matches, *sifts_);
    }
    // query_processor_basic.cpp:105-293
    bool Process(Query &query, std::vector<SiftMatch> &good_matches, std::vector<feature_3D_info> &sifts) override
    {
        samples.clear(); fitted_matches.clear(); fit2d_.clear(); fit3d_.clear(); std::memset(&result, 0, sizeof result);
        solver_focal_ = query.focal_length();
        upload(query, good_matches, query.sift_keypoints(), sifts, false);
        run(query, good_matches, POSE_AMD_SAMPLE_BASIC, 0, max_ransac_steps, 1);
        samples.resize((size_t)result.steps * 4);
        finish(query, good_matches, sifts);
        return true;
    }

    // The loop over queries of benchmark.cpp:24-36 in one call (pose_amd_ransac_batch): query i
    // uses the random stream (seed, query_id + i).  Poses, focal lengths and projection matrices
    // are written back into the queries like Process does; results[i] holds the loop statistics.
    // fitted_matches are not produced here: TestHypothesis on the final pose returns them.
    bool ProcessBatch(std::vector<Query *> &queries, std::vector<std::vector<SiftMatch> *> &good_matches,
                      std::vector<feature_3D_info> &sifts, std::vector<pose_amd_result> &results)
    {
        const size_t nq = queries.size();
        std::vector<std::vector<float>> store(nq * 5);
        std::vector<std::vector<uint32_t>> lidx(nq);
        std::vector<pose_amd_query> desc(nq);
        for (size_t i = 0; i < nq; ++i) {
            Query &q = *queries[i];
            const std::vector<SiftMatch> &ms = *good_matches[i];
            const size_t n = ms.size();
            for (int a = 0; a < 5; ++a) store[5 * i + a].resize(n);
            lidx[i].resize(n);
            for (size_t k = 0; k < n; ++k) {
                const SIFT_keypoint &kp = q.sift_keypoints()[ms[k].lindex];
                const point3D &p = sifts[ms[k].mindex].point;
                store[5 * i][k] = kp.x; store[5 * i + 1][k] = kp.y;
                store[5 * i + 2][k] = p.x; store[5 * i + 3][k] = p.y; store[5 * i + 4][k] = p.z;
                lidx[i][k] = ms[k].lindex;
            }
            pose_amd_query &d = desc[i];
            std::memset(&d, 0, sizeof d);
            d.n = (int)n;
            d.kx = store[5 * i].data(); d.ky = store[5 * i + 1].data();
            d.X = store[5 * i + 2].data(); d.Y = store[5 * i + 3].data(); d.Z = store[5 * i + 4].data();
            d.lindex = lidx[i].data();
            d.width = q.image_width(); d.height = q.image_height(); d.focal = q.focal_length();
            d.query_id = query_id + (uint32_t)i;
        }
        pose_amd_params prm;
        std::memset(&prm, 0, sizeof prm);
        prm.select_mode = select_mode; prm.eps = inlier_eps; prm.early_exit = 1;
        prm.inlier_divisor = inlier_divisor; prm.inlier_absolute = inlier_absolute;
        results.assign(nq, pose_amd_result{});
        if (devices.empty()) {
            dev_->check(pose_amd_ransac_batch(dev_->ctx(), (int)nq, desc.data(), POSE_AMD_SAMPLE_BASIC, seed, max_ransac_steps, &prm,
                                              results.data()),
                        "ransac_batch");
        } else if (pose_amd_ransac_batch_multi(devices.data(), (int)devices.size(), (int)nq, desc.data(), POSE_AMD_SAMPLE_BASIC, seed,
                                               max_ransac_steps, &prm, results.data()) != POSE_AMD_OK) {
            throw std::runtime_error(std::string("ransac_batch_multi: ") + pose_amd_multi_last_error());
        }
        for (size_t i = 0; i < nq; ++i) {
            const pose_amd_result &r = results[i];
            Query &q = *queries[i];
            if (r.found) {
                if (q.focal_length() == 0.0) q.set_focal_length(r.focal);
                q.set_camera_position(Vector3d(r.C));
                Matrix3d Rc;
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) Rc(a, b) = r.Rc[3 * a + b];
                q.set_camera_rotation(Rc);
            }
            q.ComputeMatrices(dev_);
        }
        return true;
    }
};

class QueryProcessorAdvanced : public QueryProcessor {
public:
    using QueryProcessor::QueryProcessor;
    uint32_t max_ransac_steps = 100;
    // Stand-in for QueryProcessorAdvanced::Backmatching (advanced.cpp:75-237, a FLANN search
    // outside this library): may append matches; return false to skip.
    bool (*backmatching)(std::vector<SiftMatch> &good_matches, void *user) = nullptr;
    void *backmatching_user = nullptr;
    int best_leg = 0;            // which leg kept the pose (0: before, 1: after backmatching); result.best_hyp counts within it
    bool backmatched = false;    // the second leg ran
    float ratio = 0.9f;          // advanced.cpp:269
    // query_processor_advanced.cpp:239-624
    bool Process(Query &query) override
    {
        match(query, ratio, true);
        return Process(query, good_matches, *sifts_);
    }
    // query_processor_advanced.cpp:296-606
    bool Process(Query &query, std::vector<SiftMatch> &good_matches, std::vector<feature_3D_info> &sifts) override
    {
        samples.clear(); fitted_matches.clear(); fit2d_.clear(); fit3d_.clear(); std::memset(&result, 0, sizeof result);
        solver_focal_ = query.focal_length();
        best_leg = 0; backmatched = false;
        upload(query, good_matches, query.sift_keypoints(), sifts, true);
        run(query, good_matches, POSE_AMD_SAMPLE_ADVANCED, 0, max_ransac_steps, 0);
        // the size of good_matches before backmatching decides (advanced.cpp:587-589)
        const uint32_t thresh = std::min<uint32_t>((uint32_t)good_matches.size() / inlier_divisor, inlier_absolute);
        if (backmatching && (result.found ? result.inliers : 0u) < thresh && backmatching(good_matches, backmatching_user)) {
            backmatched = true;
            upload(query, good_matches, query.sift_keypoints(), sifts, true);
            // steps restart, the kept pose and its quality carry over (advanced.cpp:597-602)
            if (run(query, good_matches, POSE_AMD_SAMPLE_ADVANCED, max_ransac_steps, max_ransac_steps, 0)) best_leg = 1;
        }
        finish(query, good_matches, sifts);
        return true;
    }
};

} // namespace pose_estimation_amd
