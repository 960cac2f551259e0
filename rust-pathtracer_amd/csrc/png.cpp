// png.cpp -- PNG writer for display copies of a render: 8 or 16 bits per channel, sRGB transfer.
//
// The reference keeps a tone-mapped pt.png beside its EXR output and still lists an image encoder among its
// dependencies (Cargo.toml:8-13), but the code that wrote the file is gone from src/main.rs; what remains is
// the EXR path (src/main.rs:420-435), which this mirrors in argument order: linear RGB f32, value * exposure.
// Here the scaled value is clamped to [0, 1], passed through the sRGB transfer function (IEC 61966-2-1) and
// rounded to the nearest code.  Rows use the "Up" filter and one zlib stream.
#include <zlib.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../../include/pt_api.h"

namespace {

void put_be32(std::vector<uint8_t>& b, uint32_t v) {
  for (int s = 24; s >= 0; s -= 8) b.push_back((uint8_t)(v >> s));
}
void put_chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data) {
  put_be32(out, (uint32_t)data.size());
  const size_t at = out.size();
  out.insert(out.end(), type, type + 4);
  out.insert(out.end(), data.begin(), data.end());
  put_be32(out, (uint32_t)crc32(0L, out.data() + at, (uInt)(out.size() - at)));
}
float srgb_encode(float v) {
  if (!(v > 0.0f)) return 0.0f;  // negative and NaN to black
  if (v >= 1.0f) return 1.0f;
  return v <= 0.0031308f ? 12.92f * v : 1.055f * std::pow(v, 1.0f / 2.4f) - 0.055f;
}

}  // namespace

extern "C" int pt_write_png(const char* path, int w, int h, const float* rgb, float exposure, int bits) {
  if (!path || !rgb || w <= 0 || h <= 0 || (bits != 8 && bits != 16)) return -1;
  const size_t bpc = (size_t)bits / 8, row = (size_t)w * 3 * bpc;
  if ((uint64_t)(row + 1) * (uint64_t)h > 0x7fffffffull) return -2;  // one zlib call holds the image
  const float top = bits == 8 ? 255.0f : 65535.0f;
  std::vector<uint8_t> raw((row + 1) * (size_t)h), prev(row, 0), cur(row);
  for (int y = 0; y < h; ++y) {
    const float* src = rgb + (size_t)y * w * 3;
    for (size_t i = 0; i < (size_t)w * 3; ++i) {
      const uint32_t code = (uint32_t)std::lrint(srgb_encode(src[i] * exposure) * top);
      if (bits == 8) cur[i] = (uint8_t)code;
      else { cur[2 * i] = (uint8_t)(code >> 8); cur[2 * i + 1] = (uint8_t)code; }
    }
    uint8_t* dst = raw.data() + (size_t)y * (row + 1);
    dst[0] = 2;  // filter type Up
    for (size_t i = 0; i < row; ++i) dst[1 + i] = (uint8_t)(cur[i] - prev[i]);
    prev.swap(cur);
  }
  uLongf zlen = compressBound((uLong)raw.size());
  std::vector<uint8_t> z(zlen);
  if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return -3;
  z.resize(zlen);

  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  std::vector<uint8_t> ihdr;
  put_be32(ihdr, (uint32_t)w);
  put_be32(ihdr, (uint32_t)h);
  ihdr.push_back((uint8_t)bits);
  ihdr.push_back(2);  // colour type: RGB
  ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);  // deflate, adaptive filtering, no interlace
  put_chunk(out, "IHDR", ihdr);
  put_chunk(out, "sRGB", std::vector<uint8_t>{0});  // rendering intent: perceptual
  put_chunk(out, "IDAT", z);
  put_chunk(out, "IEND", {});
  FILE* f = fopen(path, "wb");
  if (!f) return -4;
  const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
  return (fclose(f) == 0 && ok) ? 0 : -4;
}
