// Internal declarations shared by the host-side (CPU) sources.
#pragma once

#include <cmath>
#include <cstdint>
#include <memory>
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
const std::string& lastError();

// One PIZ-compressed OpenEXR chunk -> the bytes of the same chunk stored uncompressed (row after row, channel
// after channel).  chanWords[k]: 16-bit words per pixel of channel k (HALF 1, FLOAT / UINT 2).  (exr_piz.cpp)
bool exrPizDecode(const uint8_t* src, size_t n, const std::vector<int>& chanWords, int width, int rows,
                  std::vector<uint8_t>& out, std::string& why);

struct Blas {                         // BottomLevelAccelerateStructure, RayTracableComponent.cs:10-20
    std::vector<rtamd_node> nodes;
    std::vector<float> vertices;      // 3 per vertex
    std::vector<float> normals;       // 3 per vertex
    std::vector<rtamd_int3> triangles;
};

struct Entity {
    int blas;
    float l2w[16];
    float albedo[4], specular[4], emission[4];
};

}  // namespace rtamd_host

struct rtamd_host_scene {
    std::vector<std::unique_ptr<rtamd_host::Blas>> blobs;
    std::vector<rtamd_host::Entity> entities;
    // outputs of the last update
    std::vector<rtamd_node> tlas;
    std::vector<int32_t> blasOffsets, vtxOffsets, triOffsets;
    std::vector<rtamd_node> blasNodes;
    std::vector<rtamd_vertex> vertices;
    std::vector<rtamd_int3> triangles;
    std::vector<rtamd_object_property> objects;
    int nBlas = 0;
    int branchCount = 0;
    int tlasTree = 0;     // 0: the reference's three-job median build, 1: binned SAH
};
