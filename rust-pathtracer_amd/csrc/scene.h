// scene.h -- host-side scene description: triangle soup + per-triangle materials.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace pt {

// Per-triangle material in the reference's HitData order (optix_wrapper.cpp:162-170).
struct Material {
  float base[3];
  float metallic;
  float roughness;
  float ior;
  float alpha;
  float transmission;
  float emission[3];
};
static_assert(sizeof(Material) == 44, "Material must match HitData (44 bytes)");

struct HostScene {
  std::vector<float> vertices;     // 3 per vertex
  std::vector<uint32_t> indices;   // 3 per triangle
  std::vector<float> normals;      // 3 per triangle, geometric, normalised
  std::vector<Material> materials; // per triangle
  // rectangle light used for next-event estimation
  float light_pos[3], light_emit[3], light_normal[3], light_half[2];
  int num_tris() const { return (int)(indices.size() / 3); }
  void compute_normals();
};

// The reference's built-in scene (optix_wrapper.cpp:379-489, light 842-845).
HostScene make_cornell();
// Cornell room plus a seeded grid of subdivided icospheres (about n_objects * 20 * 4^subdiv
// triangles) with materials drawn from the three wall colours.
HostScene make_synthetic(uint32_t seed, int n_objects, int subdiv);
// Caller geometry inside the reference's light set-up; materials11 may be null (white diffuse).
HostScene make_from_triangles(const float* vertices, int n_vertices, const uint32_t* indices, int n_triangles,
                              const float* materials11);

// Wavefront OBJ (+ MTL) in and out (scene_obj.cpp).  A loaded scene keeps the reference's rectangle light
// for next-event estimation until the caller sets another one.  On failure `err` says why.
bool load_obj(const char* path, HostScene& out, std::string& err);
bool save_obj(const char* path, const HostScene& scene, std::string& err);

}  // namespace pt
