// report.hpp -- the comparison report of the reference (results.txt + the lines it prints), written against
// its output-format contract: section strings, field order and the integer arithmetic behind the numbers
// (reference src/comparator.cpp:1139-1141 header, :1199-1205 counts, :1220-1255 / :1257-1290 per-cluster
// sections, :1292-1384 cluster matching, :1386-1509 match sections, :1518 total, :1550-1567 noise,
// :1570-1615 scores and ratios, :1617-1635 verdict).
//
// What feeds it is the nearest-neighbour path of this repository: cluster sizes from the segmentation,
// matchRIFTFeaturesKnn() (one pcc_match_knn per candidate pair) for the correspondences.  The descriptors
// themselves (SIFT keypoints + RIFT histograms) are NOT computed here -- SURVEY.md section 2 keeps that pipeline
// out of scope -- they are read from files a caller provides (descriptors_io below); without them every
// cluster has 0 descriptors, nothing can match and no verdict is given.  The colour-based element count of a match
// (color_growing_segmentation on both clusters, :1466-1500) runs pcl::RegionGrowingRGB's 100-neighbour search on the GPU
// (include/pcc/region_growing_rgb.hpp) and writes the reference's three lines.
#pragma once
#include <cmath>
#include <fstream>
#include <iostream>
#include <set>
#include <sstream>
#include <string>
#include <vector>
#include "pcc/comparator_nn.hpp"
#include "pcc/region_growing_rgb.hpp"

namespace pcc {
namespace report {

typedef PointCloud<PointXYZRGB>::Ptr CloudPtr;
typedef PointCloud<RIFT32>::Ptr DescPtr;

// ---- descriptor files ------------------------------------------------------------------------------
// Text, one file per cloud:   pcc_descriptors 1
//                             cluster <index> <count>
//                             <count> lines of 32 floats (one RIFT32 histogram each)
//                             cluster ...
// Clusters that are not listed have no descriptors.  Returns false when the file cannot be parsed.
inline bool loadDescriptors(const std::string& path, size_t n_clusters, std::vector<DescPtr>& out) {
    out.assign(n_clusters, DescPtr());
    for (DescPtr& d : out) d.reset(new PointCloud<RIFT32>);
    std::ifstream f(path.c_str());
    std::string word;
    int version = 0;
    if (!f || !(f >> word >> version) || word != "pcc_descriptors" || version != 1) return false;
    while (f >> word) {
        size_t index = 0, count = 0;
        if (word != "cluster" || !(f >> index >> count)) return false;
        PointCloud<RIFT32> tmp;
        for (size_t k = 0; k < count; ++k) {
            RIFT32 h;
            for (float& v : h.histogram)
                if (!(f >> v)) return false;
            tmp.push_back(h);
        }
        if (index < n_clusters) *out[index] = tmp;  // descriptors of clusters this run did not produce are ignored
    }
    return true;
}

// centroid the way the reference sums it: float accumulators in point order, one division at the end
inline void centroidOf(const PointCloud<PointXYZRGB>& c, float out[3]) {
    out[0] = out[1] = out[2] = 0.f;
    for (const PointXYZRGB& p : c.points) { out[0] += p.x; out[1] += p.y; out[2] += p.z; }
    const size_t n = c.points.size();
    for (int a = 0; a < 3; ++a) out[a] = out[a] / n;
}

// closestCentroid (reference :1069-1087): nearest centroid not yet taken, float distance through double pow/sqrt,
// strict <, start value 1e12f; -1 when every centroid is taken.  (stays on the host.)
inline int nearestFreeCentroid(const float c[3], const std::vector<std::vector<float> >& others, const std::set<int>& taken) {
    int arg = -1;
    float best = 1000000000000.f;
    for (size_t j = 0; j < others.size(); ++j) {
        if (taken.count((int)j)) continue;
        const float d2 = std::pow(c[0] - others[j][0], 2) + std::pow(c[1] - others[j][1], 2) + std::pow(c[2] - others[j][2], 2);
        const float d = std::sqrt(d2);
        if (d < best) { best = d; arg = (int)j; }
    }
    return arg;
}

struct Scores {
    double points1 = 0, points2 = 0, des1 = 0, des2 = 0, colour1 = 0, colour2 = 0, matches = 0;
};

class Writer {
public:
    explicit Writer(const std::string& path) { f_.open(path.c_str()); }  // like the reference, a failure to open is not an error
    std::ofstream& file() { return f_; }

    void header(const std::string& a, const std::string& b) {
        f_ << "Results of comparison between " << a << " and " << b << "\n" << std::string(80, '-') << "\n\n";
    }
    void counts(size_t n1, size_t n2, size_t c1, size_t c2) {
        f_ << "Number of points of PCL 1: " << n1 << "\n" << "Number of points of PCL 2: " << n2 << "\n"
           << std::string(40, '+') << "\n"
           << "Number of clusters of PCL 1: " << c1 << "\n" << "Number of clusters of PCL 2: " << c2 << "\n";
    }
    void sectionTitle(const std::string& title) {
        f_ << "\n" << std::string(36, '-') << "\n" << title << "\n" << std::string(36, '-') << "\n";
    }
    void clusterBegin(int which, size_t j, size_t points) {
        f_ << "PCL" << which << " cluster " << j << ":\n" << "\tNumber of points: " << points << "\n";
    }
    void clusterEnd(size_t descriptors, const float c[3]) {
        f_ << "\tNumber of descriptors: " << descriptors << "\n"
           << "\tCoordinates of centroid: [" << c[0] << "," << c[1] << "," << c[2] << "]\n";
    }
    // "more X: a over: b" / "same X: a" (the three-way comparisons of the match sections)
    void compare(const char* what_plural, size_t a, size_t b) {
        if (a > b) f_ << "\t\tSegment of PCL 1 has more " << what_plural << ": " << a << " over: " << b << "\n";
        else if (a < b) f_ << "\t\tSegment of PCL 2 has more " << what_plural << ": " << b << " over: " << a << "\n";
        else f_ << "\t\tBoth segments have the same number of " << what_plural << ": " << a << "\n";
    }
    // the colour lines of a match (reference :1472-1500): PCL 1's count comes first in all three, "over" without a colon
    void compareColour(size_t a, size_t b) {
        if (a > b) f_ << "\t\tSegment of PCL 1 has more elements based on color differences: " << a << " over " << b << "\n";
        else if (a < b) f_ << "\t\tSegment of PCL 2 has more elements based on color differences: " << a << " over " << b << "\n";
        else f_ << "\t\tSegment of PCL 1 and segment of PCL 2 have the same number of elements based on color differences: " << a << "\n";
    }
    void matchRule() { f_ << "      " << std::string(58, '+') << "\t\n"; }
    void close() { f_.close(); }

private:
    std::ofstream f_;
};

// Sections "Information of clusters of PCL2 / PCL 1", the cluster matching between them and "Information of
// matches ...".  Returns the scores; matches[i] = cluster of PCL 2 matched to cluster i of PCL 1, or -1.
inline Scores clusterSections(Writer& w, const std::vector<CloudPtr>& clusters1, const std::vector<CloudPtr>& clusters2,
                              const std::vector<DescPtr>& des1, const std::vector<DescPtr>& des2, std::vector<int>& matches) {
    std::ofstream& f = w.file();
    std::vector<std::vector<float> > centroids2;
    w.sectionTitle("Information of clusters of PCL2:");
    for (size_t j = 0; j < clusters2.size(); ++j) {
        float c[3];
        centroidOf(*clusters2[j], c);
        w.clusterBegin(2, j, clusters2[j]->points.size());
        w.clusterEnd(des2[j]->points.size(), c);
        centroids2.push_back(std::vector<float>(c, c + 3));
    }
    w.sectionTitle("Information of clusters of PCL 1:");
    matches.assign(clusters1.size(), -1);
    for (size_t i = 0; i < clusters1.size(); ++i) {
        float c[3];
        centroidOf(*clusters1[i], c);
        w.clusterBegin(1, i, clusters1[i]->points.size());
        w.clusterEnd(des1[i]->points.size(), c);
        // the three nearest clusters of PCL 2 (by centroid) are tried in order of distance
        std::set<int> taken;
        int candidate[3];
        for (int k = 0; k < 3; ++k) {
            candidate[k] = nearestFreeCentroid(c, centroids2, taken);
            if (k < 2) taken.insert(candidate[k]);
        }
        size_t best_corr = 0;
        const size_t n1 = des1[i]->points.size();
        for (int k = 0; k < 3; ++k) {
            const int j = candidate[k];
            if (j == -1) { std::cout << "No closer centroid found" << std::endl; continue; }
            const size_t n2 = des2[j]->points.size();
            if (n1 <= 3 || n2 <= 3) continue;  // "!empty() and size() > 3" on both sides
            // size ratio: size_t / size_t, kept only when it is exactly 1 (0.5 < coef < 2 on an integer quotient)
            const size_t coef = clusters2[j]->points.size() / clusters1[i]->points.size();
            if (coef != 1) continue;
            std::cout << "Des pcl 1: " << n1 << std::endl << "Des pcl 2: " << n2 << std::endl;
            const size_t corr = matchRIFTFeaturesKnn(des1[i], des2[j]).size();  // matches + the dummy first element
            // percentage and acceptance use integer division by the LARGER descriptor count
            const bool first_larger = n1 > n2;
            const size_t denom = first_larger ? n1 : n2;
            std::cout << "Percentage of RIFT correspondences of clusters " << i << " and " << j << " is: " << corr / denom * 100
                      << std::endl;
            if (corr / denom > 0.5 && corr > best_corr) {
                best_corr = corr;
                matches[i] = j;
                if (first_larger) std::cout << "Match accepted" << std::endl;  // (the reference prints it in this branch only)
            }
        }
    }
    w.sectionTitle("Information of matches of clusters of PCL 1 and PCL 2:");
    Scores s;
    for (size_t i = 0; i < matches.size(); ++i) {
        if (matches[i] != -1) {
            const size_t j = (size_t)matches[i];
            ++s.matches;
            f << "\tMatched cluster " << i << " of PCL 1 with cluster " << j << " of PCL 2:\n";
            const size_t p1 = clusters1[i]->points.size(), p2 = clusters2[j]->points.size();
            const size_t d1 = des1[i]->points.size(), d2 = des2[j]->points.size();
            w.compare("points", p1, p2);
            s.points1 += p1; s.points2 += p2;
            w.compare("descriptors", d1, d2);
            s.des1 += d1; s.des2 += d2;
            // colour based segmentation of both clusters: which one has more elements of different colours
            const size_t c1 = color_growing_segmentation<PointXYZRGB>(clusters1[i]).size();
            const size_t c2 = color_growing_segmentation<PointXYZRGB>(clusters2[j]).size();
            w.compareColour(c1, c2);
            s.colour1 += c1; s.colour2 += c2;
        } else {
            std::cout << "No match" << std::endl;
            f << "\t\tCluster " << i << " of PCL 1 has no match in PCL 2\n";
        }
        w.matchRule();
    }
    std::cout << std::endl;
    f << "Total number of matches found: " << s.matches << "\n\n";
    return s;
}

// score block, ratios and the verdict: 1 = the first cloud has more information, 2 = the second, 0 = the same
inline int scoreSections(Writer& w, const Scores& s, size_t n_clusters2) {
    std::ofstream& f = w.file();
    const char* names[3] = {"points", "descriptors", "color elements"};
    const double v1[3] = {s.points1, s.des1, s.colour1}, v2[3] = {s.points2, s.des2, s.colour2};
    std::cout << std::endl << std::string(28, '-') << std::endl;
    for (int k = 0; k < 3; ++k)
        std::cout << names[k] << " score pcl1: " << v1[k] << std::endl << names[k] << " score pcl2: " << v2[k] << std::endl;
    f << "\n" << std::string(28, '-') << "\n\n";
    for (int k = 0; k < 3; ++k)
        f << names[k] << " score pcl1: " << v1[k] << "\n" << names[k] << " score pcl2: " << v2[k] << (k < 2 ? "\n\n" : "\n");
    f << "\n" << std::string(28, '-') << "\n\n";
    double ratio = 0;
    int wins1 = 0, wins2 = 0;
    for (int k = 0; k < 3; ++k) {
        ratio += v2[k] != 0 ? v1[k] / v2[k] : 0;
        if (v1[k] > v2[k]) ++wins1;
        else if (v1[k] < v2[k]) ++wins2;
    }
    ratio /= 3;
    const double global = ratio * (s.matches / n_clusters2);
    f << "Ratio of similarity over the " << s.matches << " matches: " << ratio << "\n"
      << "Ratio of general similarity of pcl 1 over pcl 2: " << global << "\n";
    std::cout << std::string(28, '-') << std::endl
              << "Ratio of similarity over the " << s.matches << " matches: " << ratio << std::endl
              << "Ratio of general similarity of pcl 1 over pcl 2: " << global << std::endl;
    return wins1 > wins2 ? 1 : (wins1 < wins2 ? 2 : 0);
}

}  // namespace report
}  // namespace pcc
