// Host-side BVH construction for librtmi (replaces Scene::BuildBSP, Render2/scene.cpp:463-693).
//
// The reference builds one auto-partitioning BSP per object, splitting triangles into fragments
// with an unseeded RNG.  Any valid bounding-volume hierarchy over the ORIGINAL triangles gives the
// same closest hit, so we build one binned-SAH BVH over all objects' triangles (design nod to the
// reference's dead abvh.cpp: AABB hierarchy + surface-area cost, Render2/abvh.cpp:242-304).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace rtmi {

struct BuildNode {
	float lo[3], hi[3];
	int32_t left = -1, right = -1; // children (index into nodes) or -1 for a leaf
	uint32_t first = 0, count = 0; // leaf: range in `order`
};

struct Bvh2 {
	std::vector<BuildNode> nodes; // nodes[0] = root (if any triangle)
	std::vector<uint32_t> order;  // leaf ranges index this permutation of triangle ids
	uint32_t depth = 0;           // edges on the longest root-to-leaf path
	uint32_t n_leaves = 0;
	uint32_t max_leaf = 0;
	float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
};

// xyz: n × 9 doubles.  Boxes are the triangles' exact bounds rounded OUTWARD to float.
// max_depth bounds the traversal stack: once the remaining depth budget only just fits a balanced
// tree, splits fall back to the object median.
Bvh2 build_bvh2(const double *xyz, uint64_t n, uint32_t max_leaf, uint32_t max_depth, unsigned n_threads);

// The 4-wide hierarchy of the ray queries, in the device's node format (rtmi_device.hpp: WNode, twelve 32-bit words):
//   w[0..2] origin (float bits)   w[3] ex | ey<<8 | ez<<16 (signed bytes) | n_inner<<24 | n_children<<28
//   w[4..9] lo.x, hi.x, lo.y, hi.y, lo.z, hi.z: one byte per child      w[10] index of inner child 0
//   w[11]   ~(first << 3), first = position of the triangle of child 0 if it were a leaf: child k's leaf reference is w[11] − 8k
struct WideNode { uint32_t w[12]; };
// Experiment (profiles/ab_r03.md, not kept): the form-factor kernel's rays walk a sibling's subtree through the 4-wide nodes
// instead of the binary ones.  1 builds Wide4::cover, carries it through the scene and the path tables, and switches the walk.
#ifndef RTMI_FF_WIDE_WALK
#define RTMI_FF_WIDE_WALK 0
#endif

struct Wide4 {
	std::vector<WideNode> nodes;    // nodes[0] = root; empty when the binary root is a leaf
	std::vector<uint32_t> leaf_pos; // per binary build node: position of a leaf's triangle in the device's triangle array
	std::vector<uint32_t> build_of; // per wide node: the binary inner node it stands for
	std::vector<uint32_t> cover;    // per binary build node: which part of the wide hierarchy holds exactly its subtree — the wide node
	                                // (bits 0..26) and, for a binary node the collapse opened, those of that node's children (mask in
	                                // bits 27..30; 0 = the whole node); 0xffffffff for leaves.  Who walks a binary node's subtree
	                                // through the wide nodes starts there (the form-factor kernel's siblings).
	uint32_t stack_need = 0;        // entries the deepest root-to-leaf path can leave on the traversal stack
	uint32_t n_top = 0;             // nodes[0 .. n_top) are the root, its inner children and their inner children (<= 21)
};
// Collapses a binary hierarchy whose leaves hold ONE triangle each.  Returns 0, or -1 with a message.
int build_wide(const Bvh2 &bvh, Wide4 &out, std::string &err, unsigned n_threads = 1);
// Host-side check of every invariant the kernels rely on (tests): each child's box, decoded as the kernels decode it,
// contains the child's fp32 box; references are in range; every triangle position is used exactly once; stack_need
// bounds every path.  Returns the number of violations.
uint64_t check_wide(const Bvh2 &bvh, const Wide4 &wide);

} // namespace rtmi
