// pt_demosaic_tile.h -- the gradient demosaic (src/main.rs:190-342) on a tile that lies wholly inside the image,
// four sites (one 2x2 cell of the colour filter array) at a time.
//
// Away from the image border no coordinate is ever clamped, so the colour of every neighbour of a site follows
// from the site's position in its 2x2 cell and the pattern, and the reference's six passes reduce -- value for
// value, same operations in the same order -- to three steps over planes that share one layout:
//   green_cell   mosaic            -> green plane            (passes 0-1: green kept at green sites, gradient-weighted
//                                                             mean of the horizontal and the vertical estimate elsewhere)
//   chroma_cell  mosaic, green     -> red-green, blue-green  (passes 2-3 and the colour differences of pass 4: at a
//                                                             green site from the two neighbours of each colour, at a
//                                                             red/blue site from its four diagonal neighbours, which
//                                                             still hold their mosaic values when the reference reads them)
//   combine_cell green, rg, bg     -> rgb                    (passes 4-6)
// Terms the reference adds that are +0 here (|0 - 0| between two sites of another colour) are left out: adding +0
// to a sum of absolute values does not change it.
// The functions work on plain float arrays, so the kernel runs them on LDS and a host build of this header (the
// CPU test harness, see pt_types.h) on host memory, both without FMA contraction and with correctly rounded division.
#pragma once
#include "pt_types.h"

namespace pt {
namespace dm {

constexpr int kTileW = 60, kTileH = 28;            // output tile
constexpr int kApron = 4;                          // mosaic rows/columns needed around it
constexpr int kStride = kTileW + 2 * kApron;       // 68: row length of every plane
constexpr int kRows = kTileH + 2 * kApron;         // 36
constexpr int kPlane = kStride * kRows;            // floats per plane
constexpr int kCellsW = (kTileW + 4) / 2;          // 32 x 16 cells cover the tile and two rings around it:
constexpr int kCellsH = (kTileH + 4) / 2;          //   where green and the colour differences are needed
constexpr int kOutCellsW = kTileW / 2, kOutCellsH = kTileH / 2;

enum { kR = 0, kG = 1, kB = 2 };
// colour at (dx, dy) of a 2x2 cell whose corner has even coordinates (src/main.rs:157-173)
constexpr int cfa_cell(int pattern, int dx, int dy) {
  constexpr int t[4][4] = {{kR, kG, kG, kB}, {kB, kG, kG, kR}, {kG, kR, kB, kG}, {kG, kB, kR, kG}};
  return t[pattern][dy * 2 + dx];
}

// Tiles of the image and which of them lie inside: tile (bx, by) with 1 <= bx < ix1 and 1 <= by < iy1 has its
// apron inside the image.
struct TileGrid {
  int gw, gh, ix1, iy1, n_edge;
};
inline TileGrid tile_grid(int w, int h) {
  TileGrid t;
  t.gw = (w + kTileW - 1) / kTileW;
  t.gh = (h + kTileH - 1) / kTileH;
  t.ix1 = w >= 2 * kTileW + kApron ? (w - kTileW - kApron) / kTileW + 1 : 1;
  t.iy1 = h >= 2 * kTileH + kApron ? (h - kTileH - kApron) / kTileH + 1 : 1;
  if (t.ix1 <= 1 || t.iy1 <= 1) t.ix1 = t.iy1 = 1;
  t.n_edge = t.gw * t.gh - (t.ix1 - 1) * (t.iy1 - 1);
  return t;
}

struct alignas(8) Pair {
  float x, y;
};
PT_HD Pair ld2(const float* p) { return *reinterpret_cast<const Pair*>(p); }
PT_HD void st2(float* p, float a, float b) { *reinterpret_cast<Pair*>(p) = Pair{a, b}; }

// src/main.rs:205-237 at a red or blue site: c0 the site, l2 r2 u2 d2 the same colour two sites away, gl gr gu gd
// its green neighbours
PT_HD float green_at(float c0, float l2, float r2, float u2, float d2, float gl, float gr, float gu, float gd) {
  const float grad_h = fabsf(gl - gr) + fabsf(2.0f * c0 - l2 - r2) * 0.5f;
  const float grad_v = fabsf(gu - gd) + fabsf(2.0f * c0 - u2 - d2) * 0.5f;
  const float gh = 0.5f * (gl + gr) + 0.25f * (2.0f * c0 - l2 - r2);
  const float gv = 0.5f * (gu + gd) + 0.25f * (2.0f * c0 - u2 - d2);
  const float wh = 1.0f / (1e-6f + grad_h * grad_h);
  const float wv = 1.0f / (1e-6f + grad_v * grad_v);
  return (wh * gh + wv * gv) / (wh + wv);
}

// i0: index of the cell's corner (even row, even column) in the planes
template <int P>
PT_HD void green_cell(const float* raw, float* g, int i0) {
  constexpr int S = kStride;
  const float* r = raw + i0;
  const Pair a0 = ld2(r - 2), a1 = ld2(r), a2 = ld2(r + 2);              // row 0, columns -2 .. 3
  const Pair b0 = ld2(r + S - 2), b1 = ld2(r + S), b2 = ld2(r + S + 2);  // row 1
  const Pair u2 = ld2(r - 2 * S), u1 = ld2(r - S), d2 = ld2(r + 2 * S), d3 = ld2(r + 3 * S);  // columns 0, 1 of rows -2, -1, 2, 3
  if constexpr (cfa_cell(P, 0, 0) != kG) {  // red/blue at (0, 0) and (1, 1)
    const float ga = green_at(a1.x, a0.x, a2.x, u2.x, d2.x, a0.y, a1.y, u1.x, b1.x);
    const float gb = green_at(b1.y, b0.y, b2.y, u1.y, d3.y, b1.x, b2.x, a1.y, d2.y);
    st2(g + i0, ga, a1.y);
    st2(g + i0 + S, b1.x, gb);
  } else {  // red/blue at (1, 0) and (0, 1)
    const float ga = green_at(a1.y, a0.y, a2.y, u2.y, d2.y, a1.x, a2.x, u1.y, b1.y);
    const float gb = green_at(b1.x, b0.x, b2.x, u1.x, d3.x, b0.y, b1.y, a1.x, d2.x);
    st2(g + i0, a1.x, ga);
    st2(g + i0 + S, gb, b1.y);
  }
}

// One site of a cell; rw and gg hold mosaic and green of rows and columns -1 .. 2 around the cell's corner
// (index [1 + dy][1 + dx]).  src/main.rs:238-303, 306-309.
template <int P, int DX, int DY>
PT_HD void chroma_site(const float (&rw)[4][4], const float (&gg)[4][4], float& rg, float& bg) {
  constexpr int c = cfa_cell(P, DX, DY);
  constexpr int X = DX + 1, Y = DY + 1;
  const float g0 = gg[Y][X];
  if constexpr (c == kG) {
    const float hor = 0.5f * (rw[Y][X - 1] + rw[Y][X + 1]) - 0.5f * (gg[Y][X - 1] + gg[Y][X + 1]) + g0;
    const float ver = 0.5f * (rw[Y - 1][X] + rw[Y + 1][X]) - 0.5f * (gg[Y - 1][X] + gg[Y + 1][X]) + g0;
    constexpr bool red_in_row = cfa_cell(P, 1 - DX, DY) == kR;
    rg = (red_in_row ? hor : ver) - g0;
    bg = (red_in_row ? ver : hor) - g0;
  } else {
    const float est = 0.25f * ((rw[Y - 1][X - 1] - gg[Y - 1][X - 1]) + (rw[Y - 1][X + 1] - gg[Y - 1][X + 1]) +
                               (rw[Y + 1][X - 1] - gg[Y + 1][X - 1]) + (rw[Y + 1][X + 1] - gg[Y + 1][X + 1])) + g0;
    const float own = rw[Y][X] - g0;
    if constexpr (c == kR) { rg = own; bg = est - g0; }
    else { bg = own; rg = est - g0; }
  }
}

template <int P>
PT_HD void chroma_cell(const float* raw, const float* g, float* rg, float* bg, int i0) {
  constexpr int S = kStride;
  float rw[4][4], gg[4][4];
#pragma unroll
  for (int dy = -1; dy <= 2; ++dy) {
    const Pair a = ld2(raw + i0 + dy * S - 2), b = ld2(raw + i0 + dy * S), c = ld2(raw + i0 + dy * S + 2);
    rw[dy + 1][0] = a.y; rw[dy + 1][1] = b.x; rw[dy + 1][2] = b.y; rw[dy + 1][3] = c.x;
    const Pair d = ld2(g + i0 + dy * S - 2), e = ld2(g + i0 + dy * S), f = ld2(g + i0 + dy * S + 2);
    gg[dy + 1][0] = d.y; gg[dy + 1][1] = e.x; gg[dy + 1][2] = e.y; gg[dy + 1][3] = f.x;
  }
  float r00, b00, r10, b10, r01, b01, r11, b11;
  chroma_site<P, 0, 0>(rw, gg, r00, b00);
  chroma_site<P, 1, 0>(rw, gg, r10, b10);
  chroma_site<P, 0, 1>(rw, gg, r01, b01);
  chroma_site<P, 1, 1>(rw, gg, r11, b11);
  st2(rg + i0, r00, r10);
  st2(rg + i0 + S, r01, r11);
  st2(bg + i0, b00, b10);
  st2(bg + i0 + S, b01, b11);
}

// src/main.rs:304-341 for one pixel: green and the two colour differences at its four neighbours
PT_HD void combine_px(float g0, float gl, float gr, float gu, float gd, float rl, float rr, float ru, float rd, float bl,
                      float br, float bu, float bd, float* o) {
  const float gh = fabsf(gl - gr);
  const float gv = fabsf(gu - gd);
  const float w_h = 1.0f / (1e-6f + gh);
  const float w_v = 1.0f / (1e-6f + gv);
  const float rgh = 0.5f * (rl + rr);
  const float rgv = 0.5f * (ru + rd);
  const float bgh = 0.5f * (bl + br);
  const float bgv = 0.5f * (bu + bd);
  const float rg2 = (w_h * rgh + w_v * rgv) / (w_h + w_v);
  const float bg2 = (w_h * bgh + w_v * bgv) / (w_h + w_v);
  o[0] = fmaxf(g0 + rg2, 0.0f);
  o[1] = fmaxf(g0, 0.0f);
  o[2] = fmaxf(g0 + bg2, 0.0f);
}

// the cell's four pixels: row0 = rgb of (0, 0) and (1, 0), row1 = rgb of (0, 1) and (1, 1)
PT_HD void combine_cell(const float* g, const float* rg, const float* bg, int i0, float* row0, float* row1) {
  constexpr int S = kStride;
  float pg[4][4], pr[4][4], pb[4][4];  // rows and columns -1 .. 2 (corners unused)
#pragma unroll
  for (int dy = -1; dy <= 2; ++dy) {
    const bool mid = dy == 0 || dy == 1;
    const Pair e = ld2(g + i0 + dy * S), r = ld2(rg + i0 + dy * S), b = ld2(bg + i0 + dy * S);
    pg[dy + 1][1] = e.x; pg[dy + 1][2] = e.y;
    pr[dy + 1][1] = r.x; pr[dy + 1][2] = r.y;
    pb[dy + 1][1] = b.x; pb[dy + 1][2] = b.y;
    if (mid) {
      const Pair e0 = ld2(g + i0 + dy * S - 2), e2 = ld2(g + i0 + dy * S + 2);
      const Pair r0 = ld2(rg + i0 + dy * S - 2), r2 = ld2(rg + i0 + dy * S + 2);
      const Pair b0 = ld2(bg + i0 + dy * S - 2), b2 = ld2(bg + i0 + dy * S + 2);
      pg[dy + 1][0] = e0.y; pg[dy + 1][3] = e2.x;
      pr[dy + 1][0] = r0.y; pr[dy + 1][3] = r2.x;
      pb[dy + 1][0] = b0.y; pb[dy + 1][3] = b2.x;
    }
  }
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int X = dx + 1, Y = dy + 1;
      combine_px(pg[Y][X], pg[Y][X - 1], pg[Y][X + 1], pg[Y - 1][X], pg[Y + 1][X], pr[Y][X - 1], pr[Y][X + 1], pr[Y - 1][X],
                 pr[Y + 1][X], pb[Y][X - 1], pb[Y][X + 1], pb[Y - 1][X], pb[Y + 1][X], (dy ? row1 : row0) + 3 * dx);
    }
}

}  // namespace dm
}  // namespace pt
