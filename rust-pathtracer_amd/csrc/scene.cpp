// scene.cpp -- built-in Cornell box and the synthetic large scene.
#include "scene.h"

#include <cmath>
#include <cstring>
#include <map>
#include <utility>

namespace pt {
namespace {

struct V3 {
  float x, y, z;
};

// optix_wrapper.cpp:132-145
V3 host_normalize(V3 v) {
  float len = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
  if (len <= 0.f) return V3{0.f, 0.f, 0.f};
  float inv = 1.0f / len;
  return V3{v.x * inv, v.y * inv, v.z * inv};
}
V3 host_cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// optix_wrapper.cpp:370-377
void add_quad(HostScene& s, V3 a, V3 b, V3 c, V3 d) {
  const uint32_t base = (uint32_t)(s.vertices.size() / 3);
  for (V3 v : {a, b, c, d}) {
    s.vertices.push_back(v.x); s.vertices.push_back(v.y); s.vertices.push_back(v.z);
  }
  for (uint32_t i : {base + 0, base + 1, base + 2, base + 0, base + 2, base + 3}) s.indices.push_back(i);
}
Material make_material(V3 base, float roughness, float transmission, V3 emit) {
  Material m;
  m.base[0] = base.x; m.base[1] = base.y; m.base[2] = base.z;
  m.metallic = 0.f; m.roughness = roughness; m.ior = 1.5f; m.alpha = 1.f; m.transmission = transmission;
  m.emission[0] = emit.x; m.emission[1] = emit.y; m.emission[2] = emit.z;
  return m;
}
// optix_wrapper.cpp:392-400: a quad is two triangles with the same material
void push_material(HostScene& s, V3 base, float roughness, float transmission, V3 emit) {
  const Material m = make_material(base, roughness, transmission, emit);
  s.materials.push_back(m);
  s.materials.push_back(m);
}
void set_light(HostScene& s) {
  // optix_wrapper.cpp:842-845
  s.light_pos[0] = 0.f; s.light_pos[1] = 0.f; s.light_pos[2] = 1.99f;
  s.light_emit[0] = s.light_emit[1] = s.light_emit[2] = 15.f;
  s.light_normal[0] = 0.f; s.light_normal[1] = 0.f; s.light_normal[2] = -1.f;
  s.light_half[0] = 0.3f; s.light_half[1] = 0.3f;
}

const V3 kRed{0.49f, 0.056f, 0.016f}, kGreen{0.272f, 0.733f, 0.088f}, kWhite{0.789f, 0.801f, 0.800f}, kBlack{0, 0, 0};

void add_room(HostScene& s) {
  const V3 emit{15.f, 15.f, 15.f};
  const V3 A{-1, -1, 0}, B{1, -1, 0}, C{1, 1, 0}, D{-1, 1, 0};
  const V3 Au{-1, -1, 2}, Bu{1, -1, 2}, Cu{1, 1, 2}, Du{-1, 1, 2};
  add_quad(s, A, B, C, D);     push_material(s, kWhite, 0.25f, 0.f, kBlack);  // floor
  add_quad(s, Du, Cu, Bu, Au); push_material(s, kWhite, 0.25f, 0.f, kBlack);  // ceiling
  add_quad(s, A, Au, Bu, B);   push_material(s, kWhite, 0.25f, 0.f, kBlack);  // back wall
  add_quad(s, D, Du, Au, A);   push_material(s, kGreen, 0.25f, 0.f, kBlack);  // x = -1
  add_quad(s, B, Bu, Cu, C);   push_material(s, kRed, 0.25f, 0.f, kBlack);    // x = +1
  const V3 L0{-0.3f, -0.3f, 1.999f}, L1{0.3f, -0.3f, 1.999f}, L2{0.3f, 0.3f, 1.999f}, L3{-0.3f, 0.3f, 1.999f};
  add_quad(s, L3, L2, L1, L0); push_material(s, kWhite, 0.25f, 0.f, emit);    // emissive panel
}

}  // namespace

void set_default_light(HostScene& s) { set_light(s); }

void HostScene::compute_normals() {
  // optix_wrapper.cpp:466-476
  const size_t nt = indices.size() / 3;
  normals.resize(3 * nt);
  for (size_t i = 0; i < nt; ++i) {
    const float* a = &vertices[3 * (size_t)indices[3 * i + 0]];
    const float* b = &vertices[3 * (size_t)indices[3 * i + 1]];
    const float* c = &vertices[3 * (size_t)indices[3 * i + 2]];
    const V3 ab{b[0] - a[0], b[1] - a[1], b[2] - a[2]}, ac{c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    const V3 n = host_normalize(host_cross(ab, ac));
    normals[3 * i] = n.x; normals[3 * i + 1] = n.y; normals[3 * i + 2] = n.z;
  }
}

HostScene make_cornell() {
  HostScene s;
  add_room(s);
  // glass box, optix_wrapper.cpp:441-464
  const float z0 = 0.001f, z1 = 0.6f;
  const V3 p0{0.076f, -0.5f, z0}, p1{0.5f, -0.924f, z0}, p2{0.924f, -0.5f, z0}, p3{0.5f, -0.076f, z0};
  const V3 q0{0.076f, -0.5f, z1}, q1{0.5f, -0.924f, z1}, q2{0.924f, -0.5f, z1}, q3{0.5f, -0.076f, z1};
  auto add_rect = [&](V3 a, V3 b, V3 c, V3 d) {
    add_quad(s, a, b, c, d);
    push_material(s, kWhite, 0.f, 1.f, kBlack);
  };
  add_rect(p3, p2, p1, p0);
  add_rect(q0, q1, q2, q3);
  add_rect(p0, p1, q1, q0);
  add_rect(p1, p2, q2, q1);
  add_rect(p2, p3, q3, q2);
  add_rect(p3, p0, q0, q3);
  s.compute_normals();
  set_light(s);
  return s;
}

namespace {
struct Rng32 {
  uint32_t s;
  uint32_t next() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
  float unit() { return (float)(next() >> 8) * (1.0f / 16777216.0f); }
};

// unit icosphere, `subdiv` rounds of 4-way subdivision
void make_icosphere(int subdiv, std::vector<V3>& verts, std::vector<uint32_t>& tris) {
  const float t = (1.0f + std::sqrt(5.0f)) * 0.5f;
  const V3 base[12] = {{-1, t, 0}, {1, t, 0}, {-1, -t, 0}, {1, -t, 0}, {0, -1, t}, {0, 1, t},
                       {0, -1, -t}, {0, 1, -t}, {t, 0, -1}, {t, 0, 1}, {-t, 0, -1}, {-t, 0, 1}};
  verts.clear();
  for (const V3& v : base) verts.push_back(host_normalize(v));
  tris = {0, 11, 5, 0, 5, 1, 0, 1, 7, 0, 7, 10, 0, 10, 11, 1, 5, 9, 5, 11, 4, 11, 10, 2, 10, 7, 6, 7, 1, 8,
          3, 9, 4, 3, 4, 2, 3, 2, 6, 3, 6, 8, 3, 8, 9, 4, 9, 5, 2, 4, 11, 6, 2, 10, 8, 6, 7, 9, 8, 1};
  for (int r = 0; r < subdiv; ++r) {
    std::map<std::pair<uint32_t, uint32_t>, uint32_t> mid;
    auto midpoint = [&](uint32_t a, uint32_t b) {
      const auto key = a < b ? std::make_pair(a, b) : std::make_pair(b, a);
      auto it = mid.find(key);
      if (it != mid.end()) return it->second;
      const V3 m = host_normalize(V3{verts[a].x + verts[b].x, verts[a].y + verts[b].y, verts[a].z + verts[b].z});
      verts.push_back(m);
      const uint32_t id = (uint32_t)verts.size() - 1;
      mid.emplace(key, id);
      return id;
    };
    std::vector<uint32_t> next;
    next.reserve(tris.size() * 4);
    for (size_t i = 0; i < tris.size(); i += 3) {
      const uint32_t a = tris[i], b = tris[i + 1], c = tris[i + 2];
      const uint32_t ab = midpoint(a, b), bc = midpoint(b, c), ca = midpoint(c, a);
      for (uint32_t v : {a, ab, ca, b, bc, ab, c, ca, bc, ab, bc, ca}) next.push_back(v);
    }
    tris.swap(next);
  }
}
}  // namespace

HostScene make_synthetic(uint32_t seed, int n_objects, int subdiv) {
  HostScene s;
  add_room(s);
  std::vector<V3> sv;
  std::vector<uint32_t> st;
  make_icosphere(subdiv < 0 ? 0 : subdiv, sv, st);
  int g = 1;
  while (g * g * g < n_objects) ++g;
  Rng32 rng{seed ? seed : 1u};
  const float cell_xy = 1.8f / (float)g, cell_z = 1.5f / (float)g;
  const V3 palette[3] = {kWhite, kRed, kGreen};
  int placed = 0;
  for (int iz = 0; iz < g && placed < n_objects; ++iz)
    for (int iy = 0; iy < g && placed < n_objects; ++iy)
      for (int ix = 0; ix < g && placed < n_objects; ++ix, ++placed) {
        const float radius = (0.25f + 0.15f * rng.unit()) * std::fmin(cell_xy, cell_z);
        const V3 c{-0.9f + ((float)ix + 0.35f + 0.3f * rng.unit()) * cell_xy,
                   -0.9f + ((float)iy + 0.35f + 0.3f * rng.unit()) * cell_xy,
                   0.05f + ((float)iz + 0.35f + 0.3f * rng.unit()) * cell_z};
        const Material m = make_material(palette[rng.next() % 3u], 0.25f, 0.f, kBlack);
        const uint32_t base = (uint32_t)(s.vertices.size() / 3);
        for (const V3& v : sv) {
          s.vertices.push_back(c.x + radius * v.x);
          s.vertices.push_back(c.y + radius * v.y);
          s.vertices.push_back(c.z + radius * v.z);
        }
        for (uint32_t i : st) s.indices.push_back(base + i);
        for (size_t i = 0; i < st.size() / 3; ++i) s.materials.push_back(m);
      }
  s.compute_normals();
  set_light(s);
  return s;
}

HostScene make_synthetic_instanced(uint32_t seed, int n_objects, int subdiv) {
  HostScene s;
  add_room(s);
  HostScene::MeshRange room{0u, (uint32_t)s.num_tris()};
  std::vector<V3> sv;
  std::vector<uint32_t> st;
  make_icosphere(subdiv < 0 ? 0 : subdiv, sv, st);
  const uint32_t base = (uint32_t)(s.vertices.size() / 3);
  for (const V3& v : sv) { s.vertices.push_back(v.x); s.vertices.push_back(v.y); s.vertices.push_back(v.z); }
  for (uint32_t i : st) s.indices.push_back(base + i);
  for (size_t i = 0; i < st.size() / 3; ++i) s.materials.push_back(make_material(kWhite, 0.25f, 0.f, kBlack));
  HostScene::MeshRange ball{room.num_tris, (uint32_t)(st.size() / 3)};
  s.meshes = {room, ball};
  HostScene::Instance whole{};
  whole.mesh = 0;
  whole.xform[0] = whole.xform[5] = whole.xform[10] = 1.0f;
  s.instances.push_back(whole);
  // the placement, size and colour draws of make_synthetic, in its order
  int g = 1;
  while (g * g * g < n_objects) ++g;
  Rng32 rng{seed ? seed : 1u};
  const float cell_xy = 1.8f / (float)g, cell_z = 1.5f / (float)g;
  const V3 palette[3] = {kWhite, kRed, kGreen};
  int placed = 0;
  for (int iz = 0; iz < g && placed < n_objects; ++iz)
    for (int iy = 0; iy < g && placed < n_objects; ++iy)
      for (int ix = 0; ix < g && placed < n_objects; ++ix, ++placed) {
        const float radius = (0.25f + 0.15f * rng.unit()) * std::fmin(cell_xy, cell_z);
        const V3 c{-0.9f + ((float)ix + 0.35f + 0.3f * rng.unit()) * cell_xy,
                   -0.9f + ((float)iy + 0.35f + 0.3f * rng.unit()) * cell_xy,
                   0.05f + ((float)iz + 0.35f + 0.3f * rng.unit()) * cell_z};
        HostScene::Instance in{};
        in.mesh = 1;
        in.xform[0] = in.xform[5] = in.xform[10] = radius;
        in.xform[3] = c.x; in.xform[7] = c.y; in.xform[11] = c.z;
        in.has_material = 1;
        in.material = make_material(palette[rng.next() % 3u], 0.25f, 0.f, kBlack);
        s.instances.push_back(in);
      }
  s.compute_normals();
  set_light(s);
  return s;
}

HostScene HostScene::flattened() const {
  if (!instanced()) return *this;
  HostScene f;
  for (const Instance& in : instances) {
    const MeshRange& m = meshes[in.mesh];
    for (uint32_t t = 0; t < m.num_tris; ++t) {
      for (int k = 0; k < 3; ++k) {
        const float* v = &vertices[3 * (size_t)indices[3 * (size_t)(m.first_tri + t) + k]];
        f.indices.push_back((uint32_t)(f.vertices.size() / 3));
        for (int r = 0; r < 3; ++r)
          f.vertices.push_back(in.xform[4 * r] * v[0] + in.xform[4 * r + 1] * v[1] + in.xform[4 * r + 2] * v[2] + in.xform[4 * r + 3]);
      }
      f.materials.push_back(in.has_material ? in.material : materials[m.first_tri + t]);
    }
  }
  f.compute_normals();
  for (int k = 0; k < 3; ++k) { f.light_pos[k] = light_pos[k]; f.light_emit[k] = light_emit[k]; f.light_normal[k] = light_normal[k]; }
  f.light_half[0] = light_half[0];
  f.light_half[1] = light_half[1];
  return f;
}

HostScene make_from_triangles(const float* vertices, int n_vertices, const uint32_t* indices, int n_triangles,
                              const float* materials11) {
  HostScene s;
  s.vertices.assign(vertices, vertices + 3 * (size_t)n_vertices);
  s.indices.assign(indices, indices + 3 * (size_t)n_triangles);
  s.materials.resize((size_t)n_triangles);
  for (int i = 0; i < n_triangles; ++i) {
    if (materials11)
      memcpy(&s.materials[(size_t)i], materials11 + 11 * (size_t)i, sizeof(Material));
    else
      s.materials[(size_t)i] = make_material(kWhite, 0.25f, 0.f, kBlack);
  }
  s.compute_normals();
  set_light(s);
  return s;
}

}  // namespace pt
