// sdflib::ExactOctreeSdf — API-compatible with the reference class (include/SdfLib/ExactOctreeSdf.h:16-134) for the hot
// path: constructor, getDistance (scalar and batched), getters.  Everything runs on the MI355X through libsdfhip; unlike
// the reference (mutable scratch, ExactOctreeSdf.h:178) queries are re-entrant.
#ifndef SDFLIB_EXACT_OCTREE_SDF_H
#define SDFLIB_EXACT_OCTREE_SDF_H
#include <utility>
#include <vector>
#include "SdfFunction.h"
#include "utils/TriangleUtils.h"

namespace sdflib {
class ExactOctreeSdf : public SdfFunction {
public:
    struct OctreeNode {
        static constexpr uint32_t IS_LEAF_MASK = 1u << 31;
        static constexpr uint32_t CHILDREN_INDEX_MASK = ~IS_LEAF_MASK;
        uint32_t childrenIndex;
        uint32_t trianglesArrayIndex;
        bool isLeaf() const { return childrenIndex & IS_LEAF_MASK; }
        uint32_t getChildrenIndex() const { return childrenIndex & CHILDREN_INDEX_MASK; }
    };
    ExactOctreeSdf() {}
    ExactOctreeSdf(const Mesh& mesh, BoundingBox box, uint32_t maxDepth, uint32_t startDepth = 1, uint32_t minTrianglesPerNode = 128, uint32_t numThreads = 1) {
        (void)numThreads;   // the GPU build always produces the single-thread (correct) array
        const BoundingBox& mb = mesh.getBoundingBox();     // only a computed box (file loader / computeBoundingBox) enables seam welding
        const float mbox[6] = {mb.min.x, mb.min.y, mb.min.z, mb.max.x, mb.max.y, mb.max.z};
        const float bmin[3] = {box.min.x, box.min.y, box.min.z}, bmax[3] = {box.max.x, box.max.y, box.max.z};
        sdfhip_multi* multi = detail::defaultMulti();
        if (multi && (1u << (3 * startDepth)) >= (uint32_t)sdfhip_multi_size(multi)) {
            // SDFLIB_DEVICES: start cells sharded over the devices, the three arrays reassembled on each of them (same arrays as one device builds)
            const size_t n = (size_t)sdfhip_multi_size(multi);
            std::vector<sdfhip_exact*> trees(n, nullptr); std::vector<sdfhip_mesh*> meshes(n, nullptr);
            detail::check(sdfhip_multi_exact_build(multi, reinterpret_cast<const float*>(mesh.getVertices().data()), (uint32_t)mesh.getVertices().size(), mesh.getIndices().data(),
                                                   (uint32_t)(mesh.getIndices().size() / 3), (mb.min.x <= mb.max.x) ? mbox : nullptr, bmin, bmax, maxDepth, startDepth, minTrianglesPerNode,
                                                   meshes.data(), trees.data()));
            mTree = trees[0]; mMesh = meshes[0];
            mReplicas.assign(trees.begin() + 1, trees.end()); mReplicaMeshes.assign(meshes.begin() + 1, meshes.end());
        } else {
            sdfhip_ctx* ctx = detail::defaultContext();
            detail::check(sdfhip_mesh_create_ex(ctx, reinterpret_cast<const float*>(mesh.getVertices().data()), (uint32_t)mesh.getVertices().size(),
                                                mesh.getIndices().data(), (uint32_t)(mesh.getIndices().size() / 3),
                                                (mb.min.x <= mb.max.x) ? mbox : nullptr, &mMesh));
            detail::check(sdfhip_exact_build(ctx, mMesh, bmin, bmax, maxDepth, startDepth, minTrianglesPerNode, &mTree));
        }
        detail::check(sdfhip_exact_get_info(mTree, &mInfo));
        mBox = BoundingBox(glm::vec3(mInfo.box_min[0], mInfo.box_min[1], mInfo.box_min[2]), glm::vec3(mInfo.box_max[0], mInfo.box_max[1], mInfo.box_max[2]));
    }
    ~ExactOctreeSdf() override { release(); }
    // Copies are deep, as the reference's implicitly generated ones are (include/SdfLib/ExactOctreeSdf.h:91-93 holds std::vectors): the four
    // arrays are read back and a device tree of its own, owning its TriangleData, is made from them (default context, no replicas).
    ExactOctreeSdf(const ExactOctreeSdf& o) : SdfFunction(o) { copyFrom(o); }
    ExactOctreeSdf& operator=(const ExactOctreeSdf& o) { if (this != &o) { release(); copyFrom(o); } return *this; }
    ExactOctreeSdf(ExactOctreeSdf&& o) noexcept { *this = std::move(o); }
    ExactOctreeSdf& operator=(ExactOctreeSdf&& o) noexcept {
        if (this != &o) {
            release();
            mTree = o.mTree; o.mTree = nullptr; mMesh = o.mMesh; o.mMesh = nullptr; mInfo = o.mInfo; mBox = o.mBox;
            mReplicas = std::move(o.mReplicas); o.mReplicas.clear(); mReplicaMeshes = std::move(o.mReplicaMeshes); o.mReplicaMeshes.clear();
        }
        return *this;
    }

    glm::ivec3 getStartGridSize() const { return glm::ivec3(mInfo.start_grid_size, mInfo.start_grid_size, mInfo.start_grid_size); }
    const BoundingBox& getGridBoundingBox() const { return mBox; }
    BoundingBox getSampleArea() const override { return mBox; }
    uint32_t getMaxTrianglesInLeafs() const { return mInfo.max_triangles_in_leafs; }
    uint32_t getMinTrianglesInLeafs() const { return mInfo.min_triangles_in_leafs; }
    uint32_t getOctreeMaxDepth() const { return mInfo.max_depth; }
    SdfFormat getFormat() const override { return SdfFormat::EXACT_OCTREE; }
    // host copy of the node array, fetched on demand
    std::vector<OctreeNode> getOctreeData() const {
        std::vector<OctreeNode> nodes(mInfo.num_nodes); std::vector<uint8_t> has(mInfo.num_nodes), masks(mInfo.num_mask_bytes + 1); std::vector<uint32_t> sets(mInfo.num_set_words + 1);
        detail::check(sdfhip_exact_download(mTree, reinterpret_cast<uint32_t*>(nodes.data()), has.data(), sets.data(), masks.data()));
        return nodes;
    }
    // host copy of the per-triangle records (ExactOctreeSdf.h:132)
    std::vector<TriangleUtils::TriangleData> getTrianglesData() const {
        std::vector<TriangleUtils::TriangleData> td(mInfo.num_triangles);
        detail::check(sdfhip_exact_triangle_data(mTree, reinterpret_cast<float*>(td.data())));
        return td;
    }
    float getDistance(glm::vec3 sample) const override { float d; getDistances(&sample, 1, &d); return d; }
    float getDistance(glm::vec3 sample, glm::vec3& outGradient) const override { float d; getDistances(&sample, 1, &d, &outGradient); return d; }
    void getDistances(const glm::vec3* samples, size_t n, float* outDistances, glm::vec3* outGradients = nullptr) const override {
        if (mReplicas.empty() || n < (1u << 16)) {
            detail::check(sdfhip_exact_query(mTree, reinterpret_cast<const float*>(samples), n, outDistances, reinterpret_cast<float*>(outGradients), nullptr, SDFHIP_HOST));
            return;
        }
        detail::splitOver(1 + mReplicas.size(), n, [&](size_t k, size_t b, size_t e) {      // one replica per device of the multi-GPU build
            sdfhip_exact* t = k == 0 ? mTree : mReplicas[k - 1];
            detail::check(sdfhip_exact_query(t, reinterpret_cast<const float*>(samples + b), e - b, outDistances + b, outGradients ? reinterpret_cast<float*>(outGradients + b) : nullptr, nullptr, SDFHIP_HOST));
        });
    }
    size_t getNumDeviceReplicas() const { return 1 + mReplicas.size(); }
    sdfhip_exact* handle() const { return mTree; }

    // archive body of include/SdfLib/ExactOctreeSdf.h:138-165: mBox, mStartGridSize, mStartDepth, mMinTrianglesInLeafs,
    // mMaxTrianglesInLeafs, mMaxTrianglesEncodedInLeafs, mBitEncodingStartDepth, mBitsPerIndex, mMaxDepth, mOctreeData,
    // mTrianglesSets, mTrianglesMasks, mTrianglesData
    bool readPayload(std::istream& is) {
        float box[6]; sdfhip_exact_info info{};
        if (!detail::get(is, box) || !detail::get(is, info.start_grid_size) || !detail::get(is, info.start_depth) || !detail::get(is, info.min_triangles_in_leafs) ||
            !detail::get(is, info.max_triangles_in_leafs) || !detail::get(is, info.max_triangles_encoded_in_leafs) || !detail::get(is, info.bit_encoding_start_depth) ||
            !detail::get(is, info.bits_per_index) || !detail::get(is, info.max_depth)) return false;
        std::vector<OctreeNode> nodes; std::vector<uint32_t> sets; std::vector<uint8_t> masks; std::vector<TriangleUtils::TriangleData> td;
        if (!detail::getVec(is, nodes) || !detail::getVec(is, sets) || !detail::getVec(is, masks) || !detail::getVec(is, td)) return false;
        for (int a = 0; a < 3; a++) { info.box_min[a] = box[a]; info.box_max[a] = box[3 + a]; }
        info.num_nodes = nodes.size(); info.num_set_words = sets.size(); info.num_mask_bytes = masks.size(); info.num_triangles = td.size();
        sets.push_back(0); masks.push_back(0);          // keep data() non-null for empty vectors
        sdfhip_exact* t = nullptr;
        detail::check(sdfhip_exact_from_data(detail::defaultContext(), &info, reinterpret_cast<const uint32_t*>(nodes.data()), sets.data(), masks.data(),
                                             reinterpret_cast<const float*>(td.data()), &t));
        release();
        mTree = t;
        detail::check(sdfhip_exact_get_info(mTree, &mInfo));
        mBox = BoundingBox(glm::vec3(box[0], box[1], box[2]), glm::vec3(box[3], box[4], box[5]));
        return true;
    }

protected:
    void writePayload(std::ostream& os) const override {
        std::vector<OctreeNode> nodes(mInfo.num_nodes); std::vector<uint8_t> has(mInfo.num_nodes), masks(mInfo.num_mask_bytes + 1); std::vector<uint32_t> sets(mInfo.num_set_words + 1);
        detail::check(sdfhip_exact_download(mTree, reinterpret_cast<uint32_t*>(nodes.data()), has.data(), sets.data(), masks.data()));
        const std::vector<TriangleUtils::TriangleData> td = getTrianglesData();
        const float box[6] = {mBox.min.x, mBox.min.y, mBox.min.z, mBox.max.x, mBox.max.y, mBox.max.z};
        detail::put(os, box); detail::put(os, (int32_t)mInfo.start_grid_size); detail::put(os, mInfo.start_depth); detail::put(os, mInfo.min_triangles_in_leafs);
        detail::put(os, mInfo.max_triangles_in_leafs); detail::put(os, mInfo.max_triangles_encoded_in_leafs); detail::put(os, mInfo.bit_encoding_start_depth);
        detail::put(os, mInfo.bits_per_index); detail::put(os, mInfo.max_depth);
        detail::putVec(os, nodes.data(), (uint64_t)mInfo.num_nodes); detail::putVec(os, sets.data(), (uint64_t)mInfo.num_set_words);
        detail::putVec(os, masks.data(), (uint64_t)mInfo.num_mask_bytes); detail::putVec(os, td.data(), (uint64_t)td.size());
    }

private:
    void copyFrom(const ExactOctreeSdf& o) {
        mInfo = o.mInfo; mBox = o.mBox; mTree = nullptr; mMesh = nullptr; mReplicas.clear(); mReplicaMeshes.clear();
        if (!o.mTree) return;
        std::vector<OctreeNode> nodes(mInfo.num_nodes); std::vector<uint8_t> has(mInfo.num_nodes), masks(mInfo.num_mask_bytes + 1); std::vector<uint32_t> sets(mInfo.num_set_words + 1);
        detail::check(sdfhip_exact_download(o.mTree, reinterpret_cast<uint32_t*>(nodes.data()), has.data(), sets.data(), masks.data()));
        const std::vector<TriangleUtils::TriangleData> td = o.getTrianglesData();
        detail::check(sdfhip_exact_from_data(detail::defaultContext(), &mInfo, reinterpret_cast<const uint32_t*>(nodes.data()), sets.data(), masks.data(),
                                             reinterpret_cast<const float*>(td.data()), &mTree));
        if (mInfo.start_grid_cell_size > 0.f) detail::check(sdfhip_exact_set_start_grid_cell_size(mTree, mInfo.start_grid_cell_size));      // the build's, not the stored box's
        detail::check(sdfhip_exact_get_info(mTree, &mInfo));
    }
    void release() {
        if (mTree) sdfhip_exact_destroy(mTree);
        for (sdfhip_exact* t : mReplicas) if (t) sdfhip_exact_destroy(t);
        if (mMesh) sdfhip_mesh_destroy(mMesh);
        for (sdfhip_mesh* m : mReplicaMeshes) if (m) sdfhip_mesh_destroy(m);
        mTree = nullptr; mMesh = nullptr; mReplicas.clear(); mReplicaMeshes.clear();
    }
    sdfhip_mesh* mMesh = nullptr;
    sdfhip_exact* mTree = nullptr;
    std::vector<sdfhip_exact*> mReplicas; std::vector<sdfhip_mesh*> mReplicaMeshes;      // SDFLIB_DEVICES: the same tree on the other devices
    sdfhip_exact_info mInfo{};
    BoundingBox mBox;
};
}  // namespace sdflib
#include "SdfLoad.h"
#endif
