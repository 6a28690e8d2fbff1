// test_report.cpp -- the host-only parts of pointcloudcomparator_amd/host/report.hpp (no GPU): descriptor files,
// the section strings of results.txt, centroid arithmetic and the nearest-free-centroid rule.  Prints "report ok".
#include <cstdio>
#include <fstream>
#include <sstream>
#include "report.hpp"

using namespace pcc;
#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static std::string slurp(const std::string& p) {
    std::ifstream f(p.c_str());
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    // descriptor file round trip; clusters that are not listed stay empty, unknown clusters are ignored
    {
        std::ofstream f((dir + "/d.txt").c_str());
        f << "pcc_descriptors 1\ncluster 2 2\n";
        for (int k = 0; k < 2; ++k) { for (int j = 0; j < 32; ++j) f << (k * 100 + j) * 0.5 << " "; f << "\n"; }
        f << "cluster 7 1\n";
        for (int j = 0; j < 32; ++j) f << j << " ";
        f << "\n";
    }
    std::vector<report::DescPtr> des;
    REQUIRE(report::loadDescriptors(dir + "/d.txt", 4, des));
    REQUIRE(des.size() == 4 && des[0]->empty() && des[1]->empty() && des[3]->empty() && des[2]->size() == 2);
    REQUIRE(des[2]->points[1].histogram[3] == 51.5f);
    { std::ofstream f((dir + "/bad.txt").c_str()); f << "pcc_descriptors 1\ncluster 0 2\n1 2 3\n"; }
    REQUIRE(!report::loadDescriptors(dir + "/bad.txt", 4, des));
    REQUIRE(!report::loadDescriptors(dir + "/missing.txt", 4, des));
    // centroid: float accumulators in point order
    PointCloud<PointXYZRGB> c;
    float sx = 0.f;
    for (int i = 0; i < 1000; ++i) { PointXYZRGB p; p.x = 0.1f * i; p.y = 1.f; p.z = -2.f; c.push_back(p); sx += p.x; }
    float cen[3];
    report::centroidOf(c, cen);
    REQUIRE(cen[0] == sx / 1000 && cen[1] == 1.f && cen[2] == -2.f);
    // nearest centroid not yet taken: strict <, lowest index on ties, -1 when all are taken
    std::vector<std::vector<float> > others = {{1, 0, 0}, {0, 1, 0}, {3, 0, 0}};
    const float origin[3] = {0, 0, 0};
    std::set<int> taken;
    REQUIRE(report::nearestFreeCentroid(origin, others, taken) == 0);
    taken.insert(0);
    REQUIRE(report::nearestFreeCentroid(origin, others, taken) == 1);
    taken.insert(1); taken.insert(2);
    REQUIRE(report::nearestFreeCentroid(origin, others, taken) == -1);
    // section strings
    {
        report::Writer w(dir + "/r.txt");
        w.header("a.ply", "b.ply");
        w.counts(10, 20, 1, 2);
        w.sectionTitle("Information of clusters of PCL2:");
        w.clusterBegin(2, 0, 216);
        const float cc[3] = {1.5f, -0.25f, 1e-5f};
        w.clusterEnd(7, cc);
        w.compare("points", 5, 3);
        w.compare("descriptors", 3, 5);
        w.compare("points", 4, 4);
        w.matchRule();
        report::Scores s;
        s.points1 = 432; s.points2 = 432; s.des1 = 18; s.des2 = 19; s.matches = 2;
        REQUIRE(report::scoreSections(w, s, 4) == 2);
        w.close();
    }
    const std::string t = slurp(dir + "/r.txt");
    const std::string want =
        "Results of comparison between a.ply and b.ply\n" + std::string(80, '-') + "\n\n"
        "Number of points of PCL 1: 10\nNumber of points of PCL 2: 20\n" + std::string(40, '+') + "\n"
        "Number of clusters of PCL 1: 1\nNumber of clusters of PCL 2: 2\n"
        "\n" + std::string(36, '-') + "\nInformation of clusters of PCL2:\n" + std::string(36, '-') + "\n"
        "PCL2 cluster 0:\n\tNumber of points: 216\n\tNumber of descriptors: 7\n\tCoordinates of centroid: [1.5,-0.25,1e-05]\n"
        "\t\tSegment of PCL 1 has more points: 5 over: 3\n\t\tSegment of PCL 2 has more descriptors: 5 over: 3\n"
        "\t\tBoth segments have the same number of points: 4\n"
        "      " + std::string(58, '+') + "\t\n"
        "\n" + std::string(28, '-') + "\n\npoints score pcl1: 432\npoints score pcl2: 432\n\ndescriptors score pcl1: 18\n"
        "descriptors score pcl2: 19\n\ncolor elements score pcl1: 0\ncolor elements score pcl2: 0\n\n" + std::string(28, '-') + "\n\n";
    REQUIRE(t.compare(0, want.size(), want) == 0);
    REQUIRE(t.find("Ratio of similarity over the 2 matches: 0.649123\n") != std::string::npos);
    REQUIRE(t.find("Ratio of general similarity of pcl 1 over pcl 2: 0.324561\n") != std::string::npos);
    std::printf("report ok\n");
    return 0;
}
