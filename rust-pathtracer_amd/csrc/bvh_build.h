// bvh_build.h -- host-side SAH builder for the compressed 8-wide BVH.
//
// Takes the place of optixAccelBuild on the driver (optix_wrapper.cpp:492-544): one
// triangle soup in, one traversable structure out.
#pragma once
#include <cstdint>
#include <vector>

#include "pt_types.h"

namespace pt {

struct BvhBuildResult {
  std::vector<BvhNode> nodes;  // node 0 is the root
  std::vector<q4> tris;        // 4 x q4 per leaf item in leaf order (traversal records, see pt_bvh.h)
  float vmag = 0.f;            // largest |coordinate| of any referenced vertex
  // per primitive: the node whose leaf slots hold it, and the bits (offsets in that node's triangle
  // range) of every triangle of that node lying in the same plane, itself included
  std::vector<uint32_t> home_node, home_mask;
  // per primitive: the order in which its corners are recorded for traversal and shading, 2 bits each
  // (bits 0-1: the input corner that comes first, ...); 0x24 = as given.  Barycentrics of a hit refer to
  // this order.
  std::vector<uint8_t> corner_order;
  // Small trees only (at most kWideMaxNodes nodes, else empty): the tree as wide nodes for kernels that keep
  // it in LDS (layout: pt_bvh.h), kWideNodeWords u4 per node under the packed tree's node indices, and where
  // each primitive lives in that tree ({node, item bits} per primitive, as home_node / home_mask).
  std::vector<u4> wide_nodes;
  std::vector<uint32_t> home_wide;
  int wide_depth = 0;          // levels of internal nodes of the wide tree
  uint32_t wide_groups = 2;    // the most slot groups (of four slots) any wide node uses
  int depth = 0;               // levels of internal nodes on the longest root-to-leaf path
  int num_tris = 0;
  int num_items = 0;           // leaf items: single triangles, or two triangles forming a parallelogram
  // Instanced scenes only (renderer.cpp: build_instanced): 6 q4 per instance for shading -- rows of the world ->
  // object matrix, then the instance's own material if it has one (pt_types.h: SceneView::inst_records)
  std::vector<q4> inst_records;
};

constexpr size_t kWideMaxNodes = 96;

// Whether consecutive triangles that form a parallelogram share one leaf item (default: yes, unless the
// environment says PT_BVH_PAIRS=0); on < 0 returns to that default.  Applies to the next build.
void bvh_set_pairing(int on);

// vertices: 3 floats each; indices: 3 per triangle.
BvhBuildResult build_bvh8(const float* vertices, const uint32_t* indices, int num_tris);
// The same tree over caller-defined boxes (6 floats each: lo xyz, hi xyz), for the top level of an instanced scene.
// Every box is one leaf item; its record in `tris` holds the box index in the w of its second word and zeros
// elsewhere, for the caller to fill in.  No pairing, no origin-plane table, no wide form.
BvhBuildResult build_bvh8_boxes(const float* boxes, int num_boxes);

}  // namespace pt
