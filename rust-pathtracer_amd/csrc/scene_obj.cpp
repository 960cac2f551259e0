// scene_obj.cpp -- Wavefront OBJ / MTL in and out: scene input beyond the built-in box.
//
// The reference has no scene input at all: its geometry is the hard-coded quad list of
// optix_wrapper.cpp:379-489 handed to one GAS build input (492-544).  What is read here ends up in the
// same place, a triangle list with one HitData-shaped material per triangle (optix_wrapper.cpp:162-170).
//
// Read: `v`, `f` (any polygon, fanned from its first corner; v, v/vt, v//vn, v/vt/vn; negative indices),
// `mtllib`, `usemtl`; from the material library `newmtl`, `Kd` (base colour), `Ke` (emission), `Ni` (index of
// refraction), `Pr` (roughness; without it `Ns` gives sqrt(2 / (Ns + 2))), `d` / `Tr` (opacity: d < 1 makes
// the surface a dielectric with transmission 1 - d).  Everything else is skipped.  Quads are fanned into
// two consecutive triangles, which is the order in which the BVH builder recognises parallelograms.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>

#include "scene.h"

namespace pt {
namespace {

Material default_material() {
  Material m;
  m.base[0] = 0.789f; m.base[1] = 0.801f; m.base[2] = 0.800f;  // the reference's white (optix_wrapper.cpp:386)
  m.metallic = 0.f; m.roughness = 0.25f; m.ior = 1.5f; m.alpha = 1.f; m.transmission = 0.f;
  m.emission[0] = m.emission[1] = m.emission[2] = 0.f;
  return m;
}

std::string directory_of(const std::string& path) {
  const size_t k = path.find_last_of("/\\");
  return k == std::string::npos ? std::string() : path.substr(0, k + 1);
}

// first whitespace-separated word of `line` from `pos`; advances pos
std::string next_word(const std::string& line, size_t& pos) {
  while (pos < line.size() && (line[pos] == ' ' || line[pos] == '\t')) ++pos;
  const size_t a = pos;
  while (pos < line.size() && line[pos] != ' ' && line[pos] != '\t' && line[pos] != '\r' && line[pos] != '\n') ++pos;
  return line.substr(a, pos - a);
}

bool read_line(FILE* f, std::string& line) {
  line.clear();
  char buf[4096];
  while (fgets(buf, sizeof(buf), f)) {
    line += buf;
    if (!line.empty() && line.back() == '\n') break;
  }
  while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
  return !line.empty() || !feof(f);
}

void load_mtl(const std::string& path, std::map<std::string, Material>& out) {
  FILE* f = fopen(path.c_str(), "r");
  if (!f) {
    fprintf(stderr, "[pt_hip] material library %s not found: default material used\n", path.c_str());
    return;
  }
  std::string line, current;
  bool have_pr = false;
  while (read_line(f, line)) {
    size_t pos = 0;
    const std::string key = next_word(line, pos);
    if (key.empty() || key[0] == '#') continue;
    if (key == "newmtl") {
      current = next_word(line, pos);
      out[current] = default_material();
      have_pr = false;
      continue;
    }
    if (current.empty()) continue;
    Material& m = out[current];
    float v[3] = {0.f, 0.f, 0.f};
    const int n = sscanf(line.c_str() + pos, "%f %f %f", &v[0], &v[1], &v[2]);
    if (n < 1) continue;
    if (n < 3) v[1] = v[2] = v[0];
    if (key == "Kd") { m.base[0] = v[0]; m.base[1] = v[1]; m.base[2] = v[2]; }
    else if (key == "Ke") { m.emission[0] = v[0]; m.emission[1] = v[1]; m.emission[2] = v[2]; }
    else if (key == "Ni") m.ior = v[0];
    else if (key == "Pr") { m.roughness = v[0]; have_pr = true; }
    else if (key == "Ns" && !have_pr) m.roughness = std::sqrt(2.0f / (std::fmax(v[0], 0.f) + 2.0f));
    else if (key == "d") m.transmission = v[0] < 1.0f ? 1.0f - v[0] : 0.f;
    else if (key == "Tr") m.transmission = v[0] > 0.0f ? v[0] : 0.f;
  }
  fclose(f);
}

}  // namespace

void set_default_light(HostScene& s);  // scene.cpp

bool load_obj(const char* path, HostScene& out, std::string& err) {
  FILE* f = fopen(path, "r");
  if (!f) {
    err = std::string("cannot open ") + path;
    return false;
  }
  HostScene s;
  std::map<std::string, Material> library;
  Material current = default_material();
  std::string line;
  std::vector<long> corner;
  long line_no = 0;
  while (read_line(f, line)) {
    ++line_no;
    size_t pos = 0;
    const std::string key = next_word(line, pos);
    if (key.empty() || key[0] == '#') continue;
    if (key == "v") {
      float x, y, z;
      if (sscanf(line.c_str() + pos, "%f %f %f", &x, &y, &z) != 3) {
        err = "line " + std::to_string(line_no) + ": malformed vertex";
        fclose(f);
        return false;
      }
      s.vertices.push_back(x); s.vertices.push_back(y); s.vertices.push_back(z);
    } else if (key == "f") {
      corner.clear();
      const long nv = (long)(s.vertices.size() / 3);
      for (;;) {
        const std::string w = next_word(line, pos);
        if (w.empty()) break;
        long idx = strtol(w.c_str(), nullptr, 10);  // stops at the first '/'
        if (idx < 0) idx = nv + idx + 1;            // relative to the vertices read so far
        if (idx < 1 || idx > nv) {
          err = "line " + std::to_string(line_no) + ": vertex index out of range";
          fclose(f);
          return false;
        }
        corner.push_back(idx - 1);
      }
      if (corner.size() < 3) {
        err = "line " + std::to_string(line_no) + ": face with fewer than three corners";
        fclose(f);
        return false;
      }
      for (size_t k = 1; k + 1 < corner.size(); ++k) {
        s.indices.push_back((uint32_t)corner[0]);
        s.indices.push_back((uint32_t)corner[k]);
        s.indices.push_back((uint32_t)corner[k + 1]);
        s.materials.push_back(current);
      }
    } else if (key == "mtllib") {
      load_mtl(directory_of(path) + next_word(line, pos), library);
    } else if (key == "usemtl") {
      const std::string name = next_word(line, pos);
      const auto it = library.find(name);
      current = it != library.end() ? it->second : default_material();
    }
  }
  fclose(f);
  s.compute_normals();
  set_default_light(s);
  out = std::move(s);
  return true;
}

// Writes the scene as path (OBJ) and path with its extension replaced by .mtl: one material per run of
// triangles with equal material, floats with nine significant digits so that reading the files back gives
// the same bits.
bool save_obj(const char* path, const HostScene& s, std::string& err) {
  std::string mtl_path(path);
  const size_t dot = mtl_path.find_last_of('.');
  const size_t slash = mtl_path.find_last_of("/\\");
  if (dot != std::string::npos && (slash == std::string::npos || dot > slash)) mtl_path.resize(dot);
  mtl_path += ".mtl";
  FILE* fo = fopen(path, "w");
  FILE* fm = fo ? fopen(mtl_path.c_str(), "w") : nullptr;
  if (!fo || !fm) {
    if (fo) fclose(fo);
    err = std::string("cannot write ") + path;
    return false;
  }
  const size_t name_at = mtl_path.find_last_of("/\\");
  fprintf(fo, "mtllib %s\n", name_at == std::string::npos ? mtl_path.c_str() : mtl_path.c_str() + name_at + 1);
  for (size_t v = 0; v + 2 < s.vertices.size(); v += 3)
    fprintf(fo, "v %.9g %.9g %.9g\n", s.vertices[v], s.vertices[v + 1], s.vertices[v + 2]);
  int n_materials = 0;
  const int nt = s.num_tris();
  for (int t = 0; t < nt; ++t) {
    const Material& m = s.materials[(size_t)t];
    if (t == 0 || memcmp(&m, &s.materials[(size_t)t - 1], sizeof(Material)) != 0) {
      fprintf(fm, "newmtl m%d\nKd %.9g %.9g %.9g\nKe %.9g %.9g %.9g\nNi %.9g\nPr %.9g\nTr %.9g\n\n", n_materials, m.base[0],
              m.base[1], m.base[2], m.emission[0], m.emission[1], m.emission[2], m.ior, m.roughness, m.transmission);
      fprintf(fo, "usemtl m%d\n", n_materials);
      ++n_materials;
    }
    fprintf(fo, "f %u %u %u\n", s.indices[3 * (size_t)t] + 1u, s.indices[3 * (size_t)t + 1] + 1u, s.indices[3 * (size_t)t + 2] + 1u);
  }
  fclose(fo);
  fclose(fm);
  return true;
}

}  // namespace pt
