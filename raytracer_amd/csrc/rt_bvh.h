// rt_bvh.h — host-side BVH2 builder for the device traversal kernels.
//
// Replaces BvhTree.FromSegment_SAH (Ararem.RayTracer.Core/Acceleration/BvhTree.cs:37-136).  The reference's
// tree is a pointer graph whose topology is implementation-defined (inconsistent comparer :181-186, SAH sweep
// over the unsorted segment :84) and degenerates to a list when an infinite AABB is present; only the closest
// hit it returns is meaningful.  This builder makes a binned-SAH tree over the objects with finite boxes, one
// object per leaf, each node holding both children's boxes (64 B, one cache line per visit).
#pragma once
#include <cstdint>
#include <vector>

namespace rtbvh {

struct Box { float mn[3], mx[3]; };

struct HostNode {            // mirrors rtd::Node
    float q[12];             // min.x(A,B) min.y(A,B) min.z(A,B) max.x(A,B) max.y(A,B) max.z(A,B)
    int32_t childA, childB;  // >= 0 inner, < 0 leaf (~object), CHILD_EMPTY none
    int32_t pad0, pad1;
};
static_assert(sizeof(HostNode) == 64, "node must be one 64-B line");

constexpr int32_t CHILD_EMPTY = (int32_t)0x80000000;

struct BuildResult {
    std::vector<HostNode> nodes;   // nodes[0] is the root (empty when there are no bounded objects)
    uint32_t max_depth = 0;        // deepest inner-node level; the traversal stack needs max_depth entries
};

// object_ids[i] is the scene object index that boxes[i] belongs to.
BuildResult build(const std::vector<Box>& boxes, const std::vector<uint32_t>& object_ids);

} // namespace rtbvh
