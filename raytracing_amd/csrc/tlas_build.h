// Device-side construction of the production kernel's trees: the top-level tree (below) and, for callers that
// want a mesh's tree in milliseconds rather than the best one (RTAMD_BLAS_BUILD_DEVICE), a bottom-level tree.
//
// The reference rebuilds its TLAS on the CPU every frame (BuildAccelerateStructureSystem.cs:56-66,
// TopLevelAccelerateStructure.cs:14-40) and uploads it; the production kernel does not walk that
// tree but one of its own over the same instance boxes (kernel_fast.h).  For scenes with many
// instances the host-side SAH rebuild of that private tree is the slow part of a per-frame refresh,
// so it can be built on the GPU instead: Morton codes of the box centroids, one radix sort, a
// binary radix tree (Karras 2012), a bottom-up box pass, and a collapse into 4-wide nodes laid out
// breadth first.  The tree only decides in which order instances are entered; which instance a
// ray hits does not depend on it.
#pragma once

#include "fast_layout.h"
#include "rtamd.h"
#include "wide_bvh.h"

namespace rt {

struct TlasDeviceBuilder;

TlasDeviceBuilder* tlasDeviceCreate();
void tlasDeviceDestroy(TlasDeviceBuilder* b);

// Builds the tree over the leaf-slot boxes of inst[0..n) into nodesOut[0..reserve) (root = node 0,
// leaves encode the instance's position in `inst`).  Needs n >= 1 and reserve >= max(n - 1, 1).
// On return the stream has been synchronised and *nNodes / *depth describe the tree.
hipError_t tlasDeviceBuild(TlasDeviceBuilder* b, const FInst* inst, int n, WNode* nodesOut, int reserve,
                           hipStream_t stream, int* nNodes, int* depth);

// The bottom-level counterpart of buildWideTree (wide_bvh.h) for leaves of two: the tree over boxes[0..n) (host
// memory) in the same form — nodes breadth first from the root, leaves as ~(start << 3 | count) into `order`.  The
// reference bakes its bottom-level trees at edit time (Assets/Editor/RayTracing/Authoring/RayTraceableBaker.cs:75-149);
// this builder is for meshes that change between frames.  Morton order, binary radix tree over the triangles (two
// sibling triangles make one leaf), collapse: the tree decides only the order of the tests, the leaf-slot boxes and
// triangles decide the hits.
// The stream has been synchronised on return.
hipError_t blasDeviceBuild(TlasDeviceBuilder* b, const BuildBox* boxes, int n, WideTree& out, hipStream_t stream);

// The leaf array of the bottom-level trees from the triangles' records by rank (host memory): record j of `out`
// (device memory, nRecords of them, an even number) is ranked[srcOf[j]], or a blank that no ray enters when
// srcOf[j] < 0; pairs of records are stored interleaved, component by component (see FastSceneHost::buildBottom).
// The stream has been synchronised on return.
hipError_t gatherLeafRecords(TlasDeviceBuilder* b, const FTri* ranked, size_t nRanked, const int32_t* srcOf, size_t nRecords, FTri* out,
                             hipStream_t stream);

// values[0..n) (host memory, no NaN) sorted ascending by the GPU's radix sort, in place.
hipError_t sortFloatsOnDevice(TlasDeviceBuilder* b, float* values, size_t n, hipStream_t stream);

// Fills inst[r] / shade[r] for r in [0, n) from the caller's TLAS and object array as they lie in
// device memory: instance r is the one in leaf slot (slotOf[r] & 3) of TLAS node (slotOf[r] >> 2);
// blasRoot maps a BLAS index to the root of its tree in the production kernel's node array.
hipError_t tlasPackInstances(const rtamd_node* tlas, const rtamd_object_property* objects, const int32_t* slotOf,
                             const int32_t* blasRoot, const int32_t* blasDemoted /* per BLAS, may be null */, int n, FInst* inst, FShade* shade,
                             hipStream_t stream);

}  // namespace rt
