// octree.h -- host quadtree keypoint distribution (see octree.cpp).
#pragma once
#include <vector>

namespace morb {

// x, y: integral candidate coordinates relative to the (16,16) border origin; resp: FAST scores.
// width/height: maxBorderX-minBorderX, maxBorderY-minBorderY.  N: the level's feature quota.
// selected: indices into the candidate arrays, in the reference's output (list) order.
void distribute_octree(const int* x, const int* y, const int* resp, int n, int width, int height, int N,
                       std::vector<int>& selected);

}  // namespace morb
