// mesh.cpp -- see mesh.hpp.  The PLY text follows what ply-rs 0.1.3 (un-vendored, Cargo.lock) emits for the header
// the reference builds in meshers/mesh.rs:41-96: parity of the byte stream is UNPINNED, the element and property
// lists, their order and types are the reference's.
#include "mesh.hpp"

#include <hip/hip_runtime_api.h>

#include <charconv>
#include <cmath>
#include <cstring>

namespace sdfviewer {

std::optional<Meshers> mesher_from_name(const std::string& k) {
    if (k == "marching-cubes") return Meshers::MarchingCubes;
    if (k == "linear-hashed-marching-cubes") return Meshers::LinearHashedMarchingCubes;
    if (k == "dual-contouring-minimize-qef") return Meshers::DualContouringMinimizeQEF;
    if (k == "dual-contouring-particle-based-minimization") return Meshers::DualContouringParticleBasedMinimization;
    return std::nullopt;
}

std::optional<Mesh> mesh_sdf(Meshers mesher, const SDFSurface& sdf, const MesherConfig& cfg, std::string* err) {
    auto fail = [&](const std::string& m) -> std::optional<Mesh> {
        if (err) *err = m;
        return std::nullopt;
    };
    const auto dev = sdf.device_sdf();
    if (!dev) return fail("this SDF has no device form: it cannot be meshed on the GPU");
    if (mesher != Meshers::MarchingCubes) return fail("Unsupported algorithm");  // isosurface.rs:49
    const BoundingBox bb = sdf.bounding_box();
    const float lo[3] = {bb[0].x, bb[0].y, bb[0].z}, hi[3] = {bb[1].x, bb[1].y, bb[1].z};
    sdfv_mesh m{};
    if (sdfv_mesh_extract(&dev->params, dev->sdf_id, lo, hi, (uint32_t)cfg.max_voxels_per_axis,
                          SDFV_MESHER_MARCHING_CUBES, &m, nullptr) != SDFV_OK)
        return fail(sdfv_last_error());
    Mesh out;
    out.vertices.resize(m.n_vertices);
    out.indices.resize(m.n_indices);
    hipError_t e = hipSuccess;
    if (m.n_vertices) e = hipMemcpy(out.vertices.data(), m.vertices, m.n_vertices * sizeof(Vertex), hipMemcpyDeviceToHost);
    if (e == hipSuccess && m.n_indices) e = hipMemcpy(out.indices.data(), m.indices, m.n_indices * 4, hipMemcpyDeviceToHost);
    sdfv_mesh_free(&m);
    if (e != hipSuccess) return fail(std::string("copying the mesh to the host: ") + hipGetErrorString(e));
    return out;
}

int Mesh::postproc(const SDFSurface& sdf) {
    const auto dev = sdf.device_sdf();
    if (!dev) return SDFV_ERR_INVALID_ARGUMENT;
    return sdfv_mesh_postproc_host(&dev->params, dev->sdf_id, reinterpret_cast<sdfv_vertex*>(vertices.data()),
                                   vertices.size());
}

std::string format_f32(float v) {
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    // shortest round-trip digits (scientific form), then laid out positionally with zero padding like Rust's Display
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf), std::fabs(v), std::chars_format::scientific);
    std::string sci(buf, r.ptr);  // d[.ddd]e[+-]xx
    const size_t e = sci.find('e');
    std::string digits;
    for (char c : sci.substr(0, e))
        if (c != '.') digits += c;
    const int exp10 = std::stoi(sci.substr(e + 1));
    std::string out = std::signbit(v) ? "-" : "";
    const int n = (int)digits.size();
    if (exp10 >= n - 1) {
        out += digits + std::string((size_t)(exp10 - (n - 1)), '0');
    } else if (exp10 >= 0) {
        out += digits.substr(0, (size_t)exp10 + 1) + "." + digits.substr((size_t)exp10 + 1);
    } else {
        out += "0." + std::string((size_t)(-exp10 - 1), '0') + digits;
    }
    return out;
}

uint8_t ply_color_u8(float c) {
    const float v = c * 255.9999f;
    if (!(v > 0.0f)) return 0;  // negatives and NaN: Rust's saturating `as u8`
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}

size_t Mesh::serialize_ply(std::ostream& out, const std::string& version_info) const {
    std::string s;
    s.reserve(96 * vertices.size() + 16 * indices.size() + 512);
    s += "ply\nformat ascii 1.0\n";
    s += "comment Created with " + version_info + "\n";  // mesh.rs:45
    s += "element vertex " + std::to_string(vertices.size()) + "\n";
    for (const char* p : {"x", "y", "z", "nx", "ny", "nz"}) s += std::string("property float ") + p + "\n";
    for (const char* p : {"red", "green", "blue"}) s += std::string("property uchar ") + p + "\n";
    for (const char* p : {"metallic", "roughness", "occlusion"}) s += std::string("property float ") + p + "\n";
    s += "element face " + std::to_string(indices.size() / 3) + "\n";
    s += "property list uchar int vertex_index\nend_header\n";
    for (const Vertex& v : vertices) {
        const float f6[6] = {v.position.x, v.position.y, v.position.z, v.normal.x, v.normal.y, v.normal.z};
        for (float f : f6) s += format_f32(f) + " ";
        s += std::to_string(ply_color_u8(v.color.x)) + " " + std::to_string(ply_color_u8(v.color.y)) + " " +
             std::to_string(ply_color_u8(v.color.z)) + " ";
        s += format_f32(v.metallic) + " " + format_f32(v.roughness) + " " + format_f32(v.occlusion) + "\n";
    }
    for (size_t t = 0; t + 2 < indices.size(); t += 3) {  // chunks_exact(3), mesh.rs:116
        s += "3 " + std::to_string((int32_t)indices[t]) + " " + std::to_string((int32_t)indices[t + 1]) + " " +
             std::to_string((int32_t)indices[t + 2]) + "\n";
    }
    out.write(s.data(), (std::streamsize)s.size());
    return s.size();
}

}  // namespace sdfviewer
