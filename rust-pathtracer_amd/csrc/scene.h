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

  // A scene made of instances (no counterpart in the reference, whose pipeline allows single-level instancing,
  // optix_wrapper.cpp:565-566, and never uses it): the arrays above then hold the meshes one after another in
  // their own object space, and the picture is the set of instances.
  struct MeshRange { uint32_t first_tri, num_tris; };
  struct Instance {
    uint32_t mesh;
    float xform[12];      // object -> world, three rows of (x y z | translation)
    int has_material;     // the instance replaces the mesh's per-triangle materials by one of its own
    Material material;
  };
  std::vector<MeshRange> meshes;
  std::vector<Instance> instances;
  bool instanced() const { return !instances.empty(); }
  // every instance written out in world space as plain triangles (what the instanced scene must look like)
  HostScene flattened() const;
};

// The reference's built-in scene (optix_wrapper.cpp:379-489, light 842-845).
HostScene make_cornell();
// Cornell room plus a seeded grid of subdivided icospheres (about n_objects * 20 * 4^subdiv
// triangles) with materials drawn from the three wall colours.
HostScene make_synthetic(uint32_t seed, int n_objects, int subdiv);
// The same picture as instances: the room is one mesh placed once, the spheres are one unit icosphere placed
// n_objects times with a scale, a translation and a material each.
HostScene make_synthetic_instanced(uint32_t seed, int n_objects, int subdiv);
// Caller geometry inside the reference's light set-up; materials11 may be null (white diffuse).
HostScene make_from_triangles(const float* vertices, int n_vertices, const uint32_t* indices, int n_triangles,
                              const float* materials11);

// Instanced scenes (scene_instances.cpp): the inverse of an affine object -> world matrix (false if singular), and
// the two-level tree in the arrays of a flat one (layout: pt_bvh.h, TraceStateT<true>).
struct BvhBuildResult;
bool invert_affine(const float m[12], double inv[12]);
BvhBuildResult build_instanced(const HostScene& hs);

// Wavefront OBJ (+ MTL) in and out (scene_obj.cpp).  A loaded scene keeps the reference's rectangle light
// for next-event estimation until the caller sets another one.  On failure `err` says why.
bool load_obj(const char* path, HostScene& out, std::string& err);
bool save_obj(const char* path, const HostScene& scene, std::string& err);

}  // namespace pt
