// exr.cpp -- minimal OpenEXR scanline writer/reader (a17): RGB, 32-bit float, uncompressed.
//
// The reference writes pt.exr / pt_bayer.exr through the `exr` crate's write_rgb_file
// (src/main.rs:420-435): linear RGB f32, value * exposure.  The file contract is "a valid
// OpenEXR image with float R, G, B channels"; this writer emits the simplest such file.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pt_api.h"

namespace {

void put_bytes(std::vector<uint8_t>& b, const void* p, size_t n) {
  const uint8_t* c = static_cast<const uint8_t*>(p);
  b.insert(b.end(), c, c + n);
}
void put_str(std::vector<uint8_t>& b, const char* s) { put_bytes(b, s, strlen(s) + 1); }
void put_i32(std::vector<uint8_t>& b, int32_t v) { put_bytes(b, &v, 4); }
void put_f32(std::vector<uint8_t>& b, float v) { put_bytes(b, &v, 4); }
void put_attr(std::vector<uint8_t>& b, const char* name, const char* type, const std::vector<uint8_t>& value) {
  put_str(b, name);
  put_str(b, type);
  put_i32(b, (int32_t)value.size());
  put_bytes(b, value.data(), value.size());
}

}  // namespace

extern "C" int pt_write_exr(const char* path, int w, int h, const float* rgb, float exposure) {
  if (!path || !rgb || w <= 0 || h <= 0) return -1;
  std::vector<uint8_t> hdr;
  const uint8_t magic[8] = {0x76, 0x2f, 0x31, 0x01, 0x02, 0x00, 0x00, 0x00};
  put_bytes(hdr, magic, 8);
  {
    std::vector<uint8_t> ch;
    for (const char* name : {"B", "G", "R"}) {  // channel list is sorted by name
      put_str(ch, name);
      put_i32(ch, 2);  // FLOAT
      const uint8_t lin_res[4] = {0, 0, 0, 0};
      put_bytes(ch, lin_res, 4);
      put_i32(ch, 1);
      put_i32(ch, 1);
    }
    ch.push_back(0);
    put_attr(hdr, "channels", "chlist", ch);
  }
  put_attr(hdr, "compression", "compression", std::vector<uint8_t>{0});
  {
    std::vector<uint8_t> box;
    put_i32(box, 0); put_i32(box, 0); put_i32(box, w - 1); put_i32(box, h - 1);
    put_attr(hdr, "dataWindow", "box2i", box);
    put_attr(hdr, "displayWindow", "box2i", box);
  }
  put_attr(hdr, "lineOrder", "lineOrder", std::vector<uint8_t>{0});
  {
    std::vector<uint8_t> v;
    put_f32(v, 1.0f);
    put_attr(hdr, "pixelAspectRatio", "float", v);
  }
  {
    std::vector<uint8_t> v;
    put_f32(v, 0.0f); put_f32(v, 0.0f);
    put_attr(hdr, "screenWindowCenter", "v2f", v);
  }
  {
    std::vector<uint8_t> v;
    put_f32(v, 1.0f);
    put_attr(hdr, "screenWindowWidth", "float", v);
  }
  hdr.push_back(0);

  FILE* f = fopen(path, "wb");
  if (!f) return -1;
  const size_t line_bytes = (size_t)w * 3 * sizeof(float);
  const uint64_t table_pos = hdr.size();
  const uint64_t first_block = table_pos + (uint64_t)h * 8u;
  bool ok = fwrite(hdr.data(), 1, hdr.size(), f) == hdr.size();
  for (int y = 0; y < h && ok; ++y) {
    const uint64_t off = first_block + (uint64_t)y * (8u + line_bytes);
    ok = fwrite(&off, 8, 1, f) == 1;
  }
  std::vector<float> line((size_t)w * 3);
  for (int y = 0; y < h && ok; ++y) {
    const float* src = rgb + (size_t)y * (size_t)w * 3;
    for (int x = 0; x < w; ++x) {
      line[(size_t)x] = src[3 * x + 2] * exposure;             // B
      line[(size_t)w + x] = src[3 * x + 1] * exposure;         // G
      line[2 * (size_t)w + x] = src[3 * x + 0] * exposure;     // R
    }
    const int32_t yy = y, nbytes = (int32_t)line_bytes;
    ok = fwrite(&yy, 4, 1, f) == 1 && fwrite(&nbytes, 4, 1, f) == 1 && fwrite(line.data(), 1, line_bytes, f) == line_bytes;
  }
  ok = (fclose(f) == 0) && ok;
  return ok ? 0 : -1;
}

// Reads files of the shape pt_write_exr produces (uncompressed scanlines, float B/G/R).
extern "C" int pt_read_exr(const char* path, int* w_out, int* h_out, float* rgb, size_t capacity_floats) {
  FILE* f = fopen(path, "rb");
  if (!f) return -1;
  std::vector<uint8_t> buf;
  {
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n < 16) { fclose(f); return -1; }
    buf.resize((size_t)n);
    if (fread(buf.data(), 1, buf.size(), f) != buf.size()) { fclose(f); return -1; }
    fclose(f);
  }
  size_t pos = 0;
  auto need = [&](size_t n) { return pos + n <= buf.size(); };
  if (!(buf[0] == 0x76 && buf[1] == 0x2f && buf[2] == 0x31 && buf[3] == 0x01)) return -2;
  if (buf[4] != 2 || (buf[5] & 0x1e)) return -2;  // tiled / deep / multipart not supported
  pos = 8;
  int w = 0, h = 0, xmin = 0, ymin = 0;
  int compression = -1;
  std::vector<std::string> channels;
  // a NUL-terminated string starting at p, entirely inside [p, limit): its length, or -1
  auto cstr_len = [&](size_t p, size_t limit) -> long {
    for (size_t q = p; q < limit && q < buf.size(); ++q)
      if (buf[q] == 0) return (long)(q - p);
    return -1;
  };
  for (;;) {
    if (!need(1)) return -2;
    if (buf[pos] == 0) { pos++; break; }
    long len = cstr_len(pos, buf.size());
    if (len < 0) return -2;
    std::string name(reinterpret_cast<const char*>(&buf[pos]), (size_t)len);
    pos += (size_t)len + 1;
    len = cstr_len(pos, buf.size());
    if (len < 0) return -2;
    std::string type(reinterpret_cast<const char*>(&buf[pos]), (size_t)len);
    pos += (size_t)len + 1;
    if (!need(4)) return -2;
    int32_t size;
    memcpy(&size, &buf[pos], 4);
    pos += 4;
    if (size < 0 || !need((size_t)size)) return -2;
    const size_t end = pos + (size_t)size;
    if (name == "channels") {
      size_t p = pos;
      while (p < end && buf[p] != 0) {
        const long cl = cstr_len(p, end);
        if (cl < 0 || p + (size_t)cl + 1 + 16 > end) return -2;  // name, then 16 bytes: type, linear, sampling
        std::string cn(reinterpret_cast<const char*>(&buf[p]), (size_t)cl);
        p += (size_t)cl + 1;
        int32_t ptype;
        memcpy(&ptype, &buf[p], 4);
        if (ptype != 2) return -3;
        p += 16;
        channels.push_back(cn);
      }
    } else if (name == "compression") {
      if (size < 1) return -2;
      compression = buf[pos];
    } else if (name == "dataWindow") {
      if (size < 16) return -2;
      int32_t b[4];
      memcpy(b, &buf[pos], 16);
      xmin = b[0]; ymin = b[1];
      const long long ww = (long long)b[2] - b[0] + 1, hh = (long long)b[3] - b[1] + 1;
      if (ww <= 0 || hh <= 0 || ww > (1 << 20) || hh > (1 << 20)) return -3;
      w = (int)ww;
      h = (int)hh;
    }
    pos = end;
  }
  if (compression != 0 || w <= 0 || h <= 0 || channels.size() != 3) return -3;
  int ci[3] = {-1, -1, -1};  // position of R, G, B in the stored order
  for (size_t i = 0; i < 3; ++i) {
    if (channels[i] == "R") ci[0] = (int)i;
    if (channels[i] == "G") ci[1] = (int)i;
    if (channels[i] == "B") ci[2] = (int)i;
  }
  if (ci[0] < 0 || ci[1] < 0 || ci[2] < 0) return -3;
  if (w_out) *w_out = w;
  if (h_out) *h_out = h;
  if (!rgb) return 0;
  if (capacity_floats < (size_t)w * (size_t)h * 3) return -4;
  const size_t line_bytes = (size_t)w * 3 * sizeof(float);
  for (int y = 0; y < h; ++y) {
    if (!need(8)) return -2;
    uint64_t off;
    memcpy(&off, &buf[pos], 8);
    pos += 8;
    if (buf.size() < 8 + line_bytes || off > buf.size() - 8 - line_bytes) return -2;  // no wrap-around for a huge offset
    int32_t yy;
    memcpy(&yy, &buf[off], 4);
    const int row = yy - ymin;
    if (row < 0 || row >= h) return -2;
    const float* src = reinterpret_cast<const float*>(&buf[off + 8]);
    float* dst = rgb + (size_t)row * (size_t)w * 3;
    for (int x = 0; x < w; ++x)
      for (int k = 0; k < 3; ++k) {
        float v;
        memcpy(&v, reinterpret_cast<const uint8_t*>(src) + ((size_t)ci[k] * (size_t)w + (size_t)x) * 4, 4);
        dst[3 * x + k] = v;
      }
  }
  (void)xmin;
  return 0;
}
