// Host-only checks of the data-format readers and the golden-pose convention
// (include/pose_amd_io.hpp): Lowe .key files (SIFT_loader.cpp:110-154), Bundler bundle.out
// (parse_bundler.cpp:80-171), LoadGoldenQuery (import_export.cpp:57-105).  Needs no GPU.
#include <cmath>
#include <cstdio>
#include <fstream>

#include "pose_amd_io.hpp"

using namespace pose_estimation_amd;

static int failures = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    {   // two keypoints in Lowe's layout: "y x scale orientation", then 128 values in rows of 20
        std::ofstream f(dir + "/t.key");
        f << "2 128\n";
        for (int k = 0; k < 2; ++k) {
            f << (k ? "310.25 17.5 2.5 -1.25\n" : "12.5 400.75 1.25 0.5\n");
            for (int j = 0; j < 128; ++j) f << " " << ((j * 7 + k * 3) % 256) << ((j % 20 == 19) ? "\n" : "");
            f << "\n";
        }
    }
    SIFT_loader sl;
    CHECK(sl.load_Lowe_features((dir + "/t.key").c_str()));
    CHECK(sl.get_nb_features() == 2);
    CHECK(sl.get_keypoints()[0].x == 400.75f && sl.get_keypoints()[0].y == 12.5f);        // x and y are swapped in the file
    CHECK(sl.get_keypoints()[1].x == 17.5f && sl.get_keypoints()[1].y == 310.25f && sl.get_keypoints()[1].scale == 2.5f &&
          sl.get_keypoints()[1].orientation == -1.25f);
    CHECK(sl.get_descriptors()[5] == 35 && sl.get_descriptors()[128 + 127] == (127 * 7 + 3) % 256);
    { std::ofstream f(dir + "/bad.key"); f << "1 64\n1 2 3 4\n"; }
    CHECK(!sl.load_Lowe_features((dir + "/bad.key").c_str()));                            // descriptor size must be 128
    CHECK(!sl.load_Lowe_features((dir + "/missing.key").c_str()));

    {   // Bundler v0.3: 2 cameras, 2 points
        std::ofstream f(dir + "/bundle.out");
        f << "# Bundle file v0.3\n2 2\n"
          << "1000.5 0.01 -0.02\n1 0 0\n0 1 0\n0 0 1\n0.5 -1.5 2.5\n"
          << "800 0 0\n0 -1 0\n1 0 0\n0 0 1\n1 2 3\n"
          << "1.5 2.5 -3.5\n255 128 0\n2 0 17 10.5 -20.25 1 4 0.5 0.25\n"
          << "-1 -2 -3\n1 2 3\n1 1 9 7.5 8.5\n";
        std::ofstream l(dir + "/list.orig.txt");
        l << "db/0001.jpg 0 1000.5\nquery/0002.jpg 0 800\n";
    }
    parse_bundler pb;
    CHECK(pb.parse_data((dir + "/bundle.out").c_str()));
    CHECK(pb.get_cameras().size() == 2 && pb.get_feature_infos().size() == 2);
    CHECK(pb.get_cameras()[0].focal_length == 1000.5 && pb.get_cameras()[0].kappa_2 == -0.02 && pb.get_cameras()[0].translation(2, 0) == 2.5);
    CHECK(pb.get_cameras()[1].rotation(0, 1) == -1.0 && pb.get_cameras()[1].rotation(1, 0) == 1.0);
    CHECK(pb.get_feature_infos()[0].point.z == -3.5f && pb.get_feature_infos()[1].point.x == -1.0f);
    CHECK(pb.camset(0).size() == 2 && pb.camset(0).count(1) == 1 && pb.camset(1).size() == 1);
    {   // descriptors of the views: camera 0 sees point 0 as its keypoint 17, camera 1 sees point 0 as 4 and point 1 as 9
        for (int c = 0; c < 2; ++c) {
            std::ofstream f(dir + (c ? "/c1.key" : "/c0.key"));
            f << "20 128\n";
            for (int k = 0; k < 20; ++k) {
                f << "1 2 3 4\n";
                for (int j = 0; j < 128; ++j) f << " " << ((j + 10 * k + 100 * c) % 256);
                f << "\n";
            }
        }
        CHECK(pb.load_descriptors({dir + "/c0.key", dir + "/c1.key"}));
        const std::vector<unsigned char> &d0 = pb.get_feature_infos()[0].descriptors, &d1 = pb.get_feature_infos()[1].descriptors;
        CHECK(d0.size() == 256 && d1.size() == 128);
        CHECK(d0[0] == (0 + 170) % 256 && d0[128 + 5] == (5 + 40 + 100) % 256 && d1[127] == (127 + 90 + 100) % 256);
        CHECK(!pb.load_descriptors({dir + "/c0.key"}));                        // a key file per camera
    }
    // query 0 = the first "query/" line = camera 2: R = [0 -1 0; 1 0 0; 0 0 1], t = (1, 2, 3)
    Query g = LoadGoldenQuery(pb, 0, dir + "/");
    CHECK(g.focal_length() == 800.0);
    const Vector3d C = g.camera_position();               // -R^T t = (-2, 1, -3)
    CHECK(C(0, 0) == -2.0 && C(1, 0) == 1.0 && C(2, 0) == -3.0);
    const Matrix3d Rc = g.camera_rotation();              // (diag(-1,1,-1) R)^T = [0 1 0; 1 0 0; 0 0 -1]
    CHECK(Rc(0, 0) == 0.0 && Rc(0, 1) == 1.0 && Rc(1, 0) == 1.0 && Rc(2, 2) == -1.0 && Rc(0, 2) == 0.0);
    bool threw = false;
    try { LoadGoldenQuery(pb, 1, dir + "/"); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);                                         // only one query line in the list
    {   // query_loader.cpp:28-82, 153-177: the list file and the lazy .key read; copies share the descriptor bytes
        std::ofstream l(dir + "/list.query.fused.txt");
        l << "query/t.jpg 486 648 659.19732\nquery/missing.jpg 1024 768\nnot a query line\n";
    }
    QueryLoader ql(dir + "/", dir + "/");
    CHECK(ql.GetQueryCount() == 2);
    {
        Query copy = ql.GetQuery(0, false);
        CHECK(copy.image_name() == "t" && copy.image_width() == 486 && copy.image_height() == 648 && copy.focal_length() == 659.19732);
        CHECK(copy.sift_keypoints().size() == 2 && copy.sift_descriptors().size() == 2);
        Query second = copy;
        ql.GetQuery(0, false);                                 // reloads the loader's own query; the copies keep their bytes
        CHECK(second.sift_descriptors()[1][127] == (127 * 7 + 3) % 256 && second.sift_descriptors()[0][5] == 35);
        CHECK(second.sift_keypoints()[0].x == 400.75f);
        Query &none = ql.GetQuery(1, false);                    // no .key file: no features, unknown focal length
        CHECK(none.focal_length() == 0.0 && none.sift_keypoints().empty() && none.image_width() == 1024);
    }
    {   // truncated and implausible files are refused without large allocations
        { std::ofstream f(dir + "/trunc.key"); f << "2 128\n1 2 3 4\n 1 2 3\n"; }
        CHECK(!sl.load_Lowe_features((dir + "/trunc.key").c_str()) && sl.get_nb_features() == 0);
        { std::ofstream f(dir + "/huge.key"); f << "4000000000 128\n1 2 3 4\n"; }
        CHECK(!sl.load_Lowe_features((dir + "/huge.key").c_str()));
        { std::ofstream f(dir + "/huge.out"); f << "# Bundle file v0.3\n4000000000 4000000000\n"; }
        parse_bundler bad;
        CHECK(!bad.parse_data((dir + "/huge.out").c_str()) && bad.get_cameras().empty());
        { std::ofstream f(dir + "/views.out"); f << "# Bundle file v0.3\n0 1\n1 2 3\n0 0 0\n4000000000 0 0 0 0\n"; }
        CHECK(!bad.parse_data((dir + "/views.out").c_str()));
    }
    std::printf(failures ? "%d check(s) failed\n" : "io ok\n", failures);
    return failures ? 1 : 0;
}
