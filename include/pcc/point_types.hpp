// point_types.hpp -- POD point types and containers with the layouts the reference's PCL types
// have at the search boundary: pcl::PointXYZ (16 B), pcl::PointXYZRGB
// (32 B, 16-byte aligned, x y z at 0/4/8, rgb at 16), pcl::Histogram<N> (RIFT32 =
// pcl::Histogram<32>, reference src/comparator.cpp:9), pcl::PointIndices, pcl::Correspondence.
// Only what the hot path touches; no PCL dependency.
#pragma once
#include <cstdint>
#include <cmath>
#include <memory>
#include <vector>

namespace pcc {

struct alignas(16) PointXYZ {
    float x = 0, y = 0, z = 0, pad_ = 1.0f;
};
struct alignas(16) PointXYZRGB {
    float x = 0, y = 0, z = 0, pad_ = 1.0f;
    union {
        float rgb;
        struct { std::uint8_t b, g, r, a; };
        std::uint32_t rgba;
    };
    float pad2_[3] = {0, 0, 0};
    PointXYZRGB() : rgba(0) {}
};
template <int N>
struct Histogram {
    float histogram[N];
};
// pcl::Normal: normal[3] + pad, curvature + pad (32 bytes)
struct alignas(16) Normal {
    float normal_x = 0, normal_y = 0, normal_z = 0, pad_ = 0;
    float curvature = 0;
    float pad2_[3] = {0, 0, 0};
};
static_assert(sizeof(Normal) == 32, "PCL layout");
static_assert(sizeof(PointXYZ) == 16 && sizeof(PointXYZRGB) == 32 && sizeof(Histogram<32>) == 128, "PCL layouts");

struct PointIndices {
    std::vector<int> indices;
};
struct ModelCoefficients {
    std::vector<float> values;
};
struct Correspondence {
    int index_query = 0;
    int index_match = -1;
    float distance = 0;  // squared distance, as PCL stores it
};

// PCL's DefaultPointRepresentation: the first min(sizeof(T)/4, 3) floats are the search
// coordinates -- that is offset 0 for every type above.
template <class PointT>
inline const float* coords(const PointT& p) { return reinterpret_cast<const float*>(&p); }
template <class PointT>
inline bool isFinite(const PointT& p) {
    const float* c = coords(p);
    return std::isfinite(c[0]) && std::isfinite(c[1]) && std::isfinite(c[2]);
}

template <class PointT>
struct PointCloud {
    typedef std::shared_ptr<PointCloud<PointT>> Ptr;
    typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
    std::vector<PointT> points;
    std::uint32_t width = 0, height = 1;
    bool is_dense = true;
    std::size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
    PointT& at(std::size_t i) { return points.at(i); }
    const PointT& at(std::size_t i) const { return points.at(i); }
    PointT& operator[](std::size_t i) { return points[i]; }
    const PointT& operator[](std::size_t i) const { return points[i]; }
    void push_back(const PointT& p) { points.push_back(p); width = (std::uint32_t)points.size(); }
};

}  // namespace pcc
