// Internal declarations shared by the host-side (CPU) sources.
#pragma once

#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "rtamd_host.h"

namespace rtamd_host {

int bvhBuild(rtamd_point_and_index* points, const rtamd_aabb* aabbs, int n, rtamd_node* nodes, int capacity,
             int* nodeCount);
int bvhBuildJobs(rtamd_point_and_index* points, const rtamd_aabb* aabbs, int n, rtamd_node* nodes, int capacity,
                 int* branchCount);
int bvhBuildSah(const rtamd_aabb* aabbs, int n, rtamd_node* nodes, int capacity, int* nodeCount);

void setLastError(const std::string& msg);

}  // namespace rtamd_host
