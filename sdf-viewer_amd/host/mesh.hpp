// mesh.hpp -- host mirror of the reference's mesh export (src/sdf/meshers/):
//   struct Mesh / Vertex            meshers/mesh.rs:8-17,133-155
//   Mesh::postproc                  meshers/mesh.rs:22-33     -> sdfv_mesh_postproc (device)
//   Mesh::serialize_ply             meshers/mesh.rs:37-129    (ASCII PLY through the un-vendored ply-rs crate)
//   Config, Meshers, Mesher::mesh   meshers/mod.rs:92-149     -> sdfv_mesh_extract (device)
// Only SDFs with a device form (SDFSurface::device_sdf) can be meshed: there is no CPU path.
#pragma once

#include <cstdint>
#include <optional>
#include <ostream>
#include <string>
#include <vector>

#include "sdf_surface.hpp"

namespace sdfviewer {

// meshers/mesh.rs:133-143 -- layout-compatible with sdfv_vertex (48 bytes)
struct Vertex {
    Vec3 position, normal, color;
    float metallic = 0.0f, roughness = 0.0f, occlusion = 0.0f;
};
static_assert(sizeof(Vertex) == sizeof(sdfv_vertex), "Vertex must stay layout-compatible with sdfv_vertex");

// meshers/mod.rs:92-107: `-v, --max-voxels-per-axis`, default 64
struct MesherConfig {
    size_t max_voxels_per_axis = 64;
};

// meshers/mod.rs:114-134; the clap subcommand names are the kebab-case variants
enum class Meshers { MarchingCubes, LinearHashedMarchingCubes, DualContouringMinimizeQEF, DualContouringParticleBasedMinimization };
std::optional<Meshers> mesher_from_name(const std::string& kebab);

struct Mesh {
    std::vector<Vertex> vertices;
    std::vector<uint32_t> indices;

    // Retrieves the materials for each vertex from the SDF; fills the normals the mesher left unset (mesh.rs:20-33).
    // Returns 0 or the sdfv status (sdfv_last_error() has the text).
    int postproc(const SDFSurface& sdf);
    // ASCII PLY with the reference's element/property list (mesh.rs:47-96); returns the bytes written.
    size_t serialize_ply(std::ostream& out, const std::string& version_info) const;
};

// Mesher::mesh (meshers/mod.rs:136-149).  Only MarchingCubes has a device implementation; the others report
// "Unsupported algorithm" (isosurface.rs:49).  nullopt on error, text in *err.
std::optional<Mesh> mesh_sdf(Meshers mesher, const SDFSurface& sdf, const MesherConfig& cfg, std::string* err);

// f32 Display as Rust prints it (shortest digits that round-trip, never an exponent): what ply-rs writes for floats.
std::string format_f32(float v);
// (c * 255.9999) as u8, mesh.rs:106-108
uint8_t ply_color_u8(float c);

}  // namespace sdfviewer
