// ply_io.hpp -- minimal PLY reader/writer for the comparator CLI (what pcl::io::loadPLYFile gives
// the reference at src/comparator.cpp:1119,1130): vertex element with x, y, z (float or double)
// and optional colour (red/green/blue uchar, or a packed float/uint "rgb"/"rgba").  Formats:
// ascii 1.0 and binary_little_endian 1.0.  Other vertex properties are skipped by size; elements
// after the vertices (faces) are ignored.  Returns -1 on failure like loadPLYFile.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "pcc/point_types.hpp"

namespace pcc {
namespace io {

namespace detail {
inline int type_size(const std::string& t) {
    if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
    if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
    if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
    if (t == "double" || t == "float64") return 8;
    return 0;
}
inline bool is_float_type(const std::string& t) { return t == "float" || t == "float32" || t == "double" || t == "float64"; }
inline double read_scalar(const char* p, const std::string& t) {
    if (t == "float" || t == "float32") { float v; std::memcpy(&v, p, 4); return v; }
    if (t == "double" || t == "float64") { double v; std::memcpy(&v, p, 8); return v; }
    if (t == "uchar" || t == "uint8") return (unsigned char)p[0];
    if (t == "char" || t == "int8") return (signed char)p[0];
    if (t == "ushort" || t == "uint16") { std::uint16_t v; std::memcpy(&v, p, 2); return v; }
    if (t == "short" || t == "int16") { std::int16_t v; std::memcpy(&v, p, 2); return v; }
    if (t == "uint" || t == "uint32") { std::uint32_t v; std::memcpy(&v, p, 4); return v; }
    if (t == "int" || t == "int32") { std::int32_t v; std::memcpy(&v, p, 4); return v; }
    return 0.0;
}
struct Prop { std::string type, name; int offset = 0; };
}  // namespace detail

inline int loadPLYFile(const std::string& path, PointCloud<PointXYZRGB>& cloud) {
    std::ifstream f(path.c_str(), std::ios::binary);
    if (!f) return -1;
    std::string line;
    if (!std::getline(f, line) || line.substr(0, 3) != "ply") return -1;
    bool ascii = false, ble = false, in_vertex = false, header_done = false;
    size_t nvert = 0;
    std::vector<detail::Prop> props;
    int stride = 0;
    bool vertex_first = true, seen_element = false;
    while (std::getline(f, line)) {
        if (!line.empty() && line[line.size() - 1] == '\r') line.erase(line.size() - 1);
        std::istringstream ls(line);
        std::string tok;
        ls >> tok;
        if (tok == "format") {
            std::string fmt; ls >> fmt;
            ascii = fmt == "ascii";
            ble = fmt == "binary_little_endian";
            if (!ascii && !ble) return -1;
        } else if (tok == "element") {
            std::string name; size_t cnt = 0;
            if (!(ls >> name >> cnt)) return -1;  // a count that does not parse is a broken header, not a size
            in_vertex = name == "vertex";
            if (in_vertex) { nvert = cnt; vertex_first = !seen_element; }
            seen_element = true;
        } else if (tok == "property" && in_vertex) {
            detail::Prop p; ls >> p.type;
            if (p.type == "list") return -1;  // no list properties on vertices
            ls >> p.name;
            p.offset = stride;
            int sz = detail::type_size(p.type);
            if (!sz) return -1;
            stride += sz;
            props.push_back(p);
        } else if (tok == "end_header") { header_done = true; break; }
    }
    if (!header_done || !vertex_first) return -1;  // vertices must be the first element
    int ix = -1, iy = -1, iz = -1, ir = -1, ig = -1, ib = -1, irgb = -1;
    for (size_t i = 0; i < props.size(); ++i) {
        const std::string& n = props[i].name;
        if (n == "x") ix = (int)i; else if (n == "y") iy = (int)i; else if (n == "z") iz = (int)i;
        else if (n == "red" || n == "r") ir = (int)i; else if (n == "green" || n == "g") ig = (int)i;
        else if (n == "blue" || n == "b") ib = (int)i; else if (n == "rgb" || n == "rgba") irgb = (int)i;
    }
    if (ix < 0 || iy < 0 || iz < 0) return -1;
    {   // the header is not trusted: the vertex count is bounded by what the rest of the file can hold
        // (binary: whole records; ascii: at least "0 0 0\n" per vertex), so a corrupt count fails here
        // with -1 like loadPLYFile does, instead of throwing from a giant allocation
        const std::streampos here = f.tellg();
        f.seekg(0, std::ios::end);
        const std::streampos end = f.tellg();
        f.seekg(here);
        if (here < 0 || end < here) return -1;
        const size_t rest = (size_t)(end - here);
        const size_t min_rec = ascii ? 2 * props.size() : (size_t)(stride > 0 ? stride : 1);
        if (nvert > rest / (min_rec ? min_rec : 1) + 1) return -1;
    }
    cloud.points.assign(nvert, PointXYZRGB());
    std::vector<double> vals(props.size());
    std::vector<char> rec(stride > 0 ? stride : 1);
    for (size_t v = 0; v < nvert; ++v) {
        std::uint32_t packed_rgb = 0;
        bool have_packed = false;
        if (ascii) {
            if (!std::getline(f, line)) return -1;
            std::istringstream ls(line);
            for (size_t i = 0; i < props.size(); ++i) {
                if ((int)i == irgb && !detail::is_float_type(props[i].type)) {
                    unsigned long u; if (!(ls >> u)) return -1;
                    packed_rgb = (std::uint32_t)u; have_packed = true; vals[i] = 0;
                } else {
                    std::string t; if (!(ls >> t)) return -1;
                    vals[i] = std::strtod(t.c_str(), nullptr);  // accepts nan / inf
                    if ((int)i == irgb) { float fv = (float)vals[i]; std::memcpy(&packed_rgb, &fv, 4); have_packed = true; }
                }
            }
        } else {
            f.read(rec.data(), stride);
            if (!f) return -1;
            for (size_t i = 0; i < props.size(); ++i) {
                vals[i] = detail::read_scalar(rec.data() + props[i].offset, props[i].type);
                if ((int)i == irgb) { std::memcpy(&packed_rgb, rec.data() + props[i].offset, 4); have_packed = true; }
            }
        }
        PointXYZRGB& p = cloud.points[v];
        p.x = (float)vals[ix]; p.y = (float)vals[iy]; p.z = (float)vals[iz];
        if (have_packed) p.rgba = packed_rgb;
        else if (ir >= 0 && ig >= 0 && ib >= 0) { p.r = (std::uint8_t)vals[ir]; p.g = (std::uint8_t)vals[ig]; p.b = (std::uint8_t)vals[ib]; p.a = 255; }
    }
    cloud.width = (std::uint32_t)nvert;
    cloud.height = 1;
    cloud.is_dense = false;
    return 0;
}

inline int savePLYFileBinary(const std::string& path, const PointCloud<PointXYZRGB>& cloud) {
    std::ofstream f(path.c_str(), std::ios::binary);
    if (!f) return -1;
    f << "ply\nformat binary_little_endian 1.0\nelement vertex " << cloud.size()
      << "\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n";
    for (const PointXYZRGB& p : cloud.points) {
        f.write(reinterpret_cast<const char*>(&p.x), 12);
        const char c[3] = {(char)p.r, (char)p.g, (char)p.b};
        f.write(c, 3);
    }
    return f ? 0 : -1;
}

// pcl::removeNaNFromPointCloud (src/comparator.cpp:1144-1148): drop points with a non-finite coordinate
inline void removeNaNFromPointCloud(PointCloud<PointXYZRGB>& cloud, std::vector<int>& index) {
    index.clear();
    size_t j = 0;
    for (size_t i = 0; i < cloud.points.size(); ++i) {
        if (!isFinite(cloud.points[i])) continue;
        cloud.points[j++] = cloud.points[i];
        index.push_back((int)i);
    }
    cloud.points.resize(j);
    cloud.width = (std::uint32_t)j;
    cloud.height = 1;
    cloud.is_dense = true;
}

}  // namespace io
}  // namespace pcc
