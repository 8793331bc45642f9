// sdflib::OctreeSdf — API-compatible with the reference class (include/SdfLib/OctreeSdf.h:20-300) for the hot path:
// constructors, getDistance (scalar and batched), getters, OctreeNode layout.  Construction and batched queries run on
// the MI355X through the libsdfhip C ABI; the scalar getDistance(vec3) reads the host copy of the node array that
// getOctreeData() exposes (same array the reference's viewers upload), evaluated in the order of the linked library's flavour
// (libsdfhip.so: the reference's literal order; libsdfhip_enoki.so: its SDFLIB_USE_ENOKI=ON order).
#ifndef SDFLIB_OCTREE_SDF_H
#define SDFLIB_OCTREE_SDF_H
#include <array>
#include <cmath>
#include <cstring>
#include <optional>
#include <utility>
#include <string>
#include <vector>
#include "SdfFunction.h"

namespace sdflib {
class OctreeSdf : public SdfFunction {
public:
    enum InitAlgorithm { UNIFORM, NO_CONTINUITY, CONTINUITY };
    struct OctreeNode {
        static constexpr uint32_t IS_LEAF_MASK = 1u << 31;
        static constexpr uint32_t MARK_MASK = 1u << 30;
        static constexpr uint32_t CHILDREN_INDEX_MASK = ~(IS_LEAF_MASK | MARK_MASK);
        union { uint32_t childrenIndex; float value; };
        bool isLeaf() const { return childrenIndex & IS_LEAF_MASK; }
        uint32_t getChildrenIndex() const { return childrenIndex & CHILDREN_INDEX_MASK; }
    };
    enum TerminationRule { NONE, TRAPEZOIDAL_RULE, SIMPSONS_RULE, BY_DISTANCE_RULE };
    class TerminationRuleParams {
    public:
        std::array<float, 2> params;
        static TerminationRuleParams setNoneRuleParams() { return TerminationRuleParams(); }
        static TerminationRuleParams setTrapezoidalRuleParams(float e) { return TerminationRuleParams{{e, 0.f}}; }
        static TerminationRuleParams setSimpsonRuleParams(float e) { return TerminationRuleParams{{e, 0.f}}; }
        static TerminationRuleParams setByDistanceRuleParams(float e, float decay) { return TerminationRuleParams{{e, decay}}; }
        float& operator[](int p) { return params[p]; }
    };
    static std::optional<TerminationRule> stringToTerminationRule(std::string text) {
        if (text == "none" || text == "NONE") return TerminationRule::NONE;
        if (text == "trapezoidal_rule" || text == "TRAPEZOIDAL_RULE") return TerminationRule::TRAPEZOIDAL_RULE;
        if (text == "simpsons_rule" || text == "SIMPSONS_RULE") return TerminationRule::SIMPSONS_RULE;
        if (text == "by_distance_rule" || text == "BY_DISTANCE_RULE") return TerminationRule::BY_DISTANCE_RULE;
        return std::optional<TerminationRule>();
    }

    OctreeSdf() {}
    OctreeSdf(const Mesh& mesh, BoundingBox box, uint32_t depth, uint32_t startDepth, float maxError = 1e-3,
              InitAlgorithm initAlgorithm = InitAlgorithm::NO_CONTINUITY, uint32_t numThreads = 1) {
        build(mesh, box, depth, startDepth, TerminationRule::TRAPEZOIDAL_RULE, TerminationRuleParams::setTrapezoidalRuleParams(maxError), initAlgorithm, numThreads);
    }
    OctreeSdf(const Mesh& mesh, BoundingBox box, uint32_t depth, uint32_t startDepth, TerminationRule rule, TerminationRuleParams params,
              InitAlgorithm initAlgorithm, uint32_t numThreads = 1) {
        build(mesh, box, depth, startDepth, rule, params, initAlgorithm, numThreads);
    }
    ~OctreeSdf() override { release(); }
    // Copies are deep, as the reference's implicitly generated ones are (include/SdfLib/OctreeSdf.h:146-172 holds a std::vector): the host
    // array is copied and a device tree of its own is made from it (on the default context; a copy has no multi-device replicas).
    OctreeSdf(const OctreeSdf& o) : SdfFunction(o) { copyFrom(o); }
    OctreeSdf& operator=(const OctreeSdf& o) { if (this != &o) { release(); copyFrom(o); } return *this; }
    OctreeSdf(OctreeSdf&& o) noexcept { *this = std::move(o); }
    OctreeSdf& operator=(OctreeSdf&& o) noexcept {
        if (this != &o) {
            release();
            mReplicas = std::move(o.mReplicas); o.mReplicas.clear();
            mTree = o.mTree; o.mTree = nullptr; mBox = o.mBox; mValueRange = o.mValueRange; mMinBorderValue = o.mMinBorderValue;
            mStartGridSize = o.mStartGridSize; mStartGridXY = o.mStartGridXY; mStartGridCellSize = o.mStartGridCellSize; mMaxDepth = o.mMaxDepth;
            mOctreeData = std::move(o.mOctreeData);
        }
        return *this;
    }

    float getOctreeValueRange() const { return mValueRange; }
    float getOctreeMinBorderValue() const { return mMinBorderValue; }
    glm::ivec3 getStartGridSize() const { return glm::ivec3(mStartGridSize, mStartGridSize, mStartGridSize); }
    const BoundingBox& getGridBoundingBox() const { return mBox; }
    BoundingBox getSampleArea() const override { return mBox; }
    uint32_t getOctreeMaxDepth() const { return mMaxDepth; }
    const std::vector<OctreeNode>& getOctreeData() const { return mOctreeData; }
    std::vector<OctreeNode>& getOctreeData() { return mOctreeData; }
    SdfFunction::SdfFormat getFormat() const override { return SdfFunction::SdfFormat::OCTREE; }

    // scalar queries: host walk of the node array (src/sdf/OctreeSdf.cpp:93-152 semantics)
    float getDistance(glm::vec3 sample) const override { glm::vec3 g; return eval(sample, nullptr, g); }
    float getDistance(glm::vec3 sample, glm::vec3& outGradient) const override { return eval(sample, &outGradient, outGradient); }
    // batched queries on the GPU
    void getDistances(const glm::vec3* samples, size_t n, float* outDistances, glm::vec3* outGradients = nullptr) const override {
        if (mReplicas.empty() || n < (1u << 18)) {
            detail::check(sdfhip_octree_query(mTree, reinterpret_cast<const float*>(samples), n, outDistances, reinterpret_cast<float*>(outGradients), SDFHIP_HOST, SDFHIP_EVAL_EXACT));
            return;
        }
        // the tree is replicated on every device of the multi-GPU build: contiguous shares of the batch, one host thread per device
        detail::splitOver(1 + mReplicas.size(), n, [&](size_t k, size_t b, size_t e) {
            sdfhip_octree* t = k == 0 ? mTree : mReplicas[k - 1];
            detail::check(sdfhip_octree_query(t, reinterpret_cast<const float*>(samples + b), e - b, outDistances + b, outGradients ? reinterpret_cast<float*>(outGradients + b) : nullptr, SDFHIP_HOST, SDFHIP_EVAL_EXACT));
        });
    }
    size_t getNumDeviceReplicas() const { return 1 + mReplicas.size(); }
    sdfhip_octree* handle() const { return mTree; }

    // src/sdf/OctreeSdf.cpp:231-277: leaves per depth weighted by their volume fraction (8^-d of a start cell... of the root)
    void getDepthDensity(std::vector<float>& depthsDensity) {
        depthsDensity.resize(mMaxDepth + 1);
        std::vector<uint32_t> leavesPerDepth(depthsDensity.size(), 0);
        const uint32_t startDepth = (uint32_t)std::lround(std::log2((float)mStartGridSize));
        std::vector<std::pair<uint32_t, uint32_t>> stack;       // (node index, depth); explicit stack instead of std::function recursion
        for (uint32_t i = 0; i < (uint32_t)(mStartGridXY * mStartGridSize); i++) stack.emplace_back(i, startDepth);
        while (!stack.empty()) {
            const auto [idx, depth] = stack.back(); stack.pop_back();
            OctreeNode& node = mOctreeData[idx];
            node.childrenIndex &= ~OctreeNode::MARK_MASK;
            if (node.isLeaf()) { if (depth < leavesPerDepth.size()) leavesPerDepth[depth]++; }
            else for (uint32_t c = 0; c < 8; c++) stack.emplace_back(node.getChildrenIndex() + c, depth + 1);
        }
        float size = 1.0f;
        for (size_t d = 0; d < depthsDensity.size(); d++) { depthsDensity[d] = size * (float)leavesPerDepth[d]; size *= 0.125f; }
    }

    // the archive body of include/SdfLib/OctreeSdf.h:222-238 (save: mBox, mStartGridSize, mMaxDepth, mValueRange, mMinBorderValue, mOctreeData)
    bool readPayload(std::istream& is) {
        float box[6]; int32_t grid = 0; uint32_t depth = 0; float vr = 0.f, mb = 0.f; std::vector<uint32_t> words;
        if (!detail::get(is, box) || !detail::get(is, grid) || !detail::get(is, depth) || !detail::get(is, vr) || !detail::get(is, mb) || !detail::getVec(is, words)) return false;
        if (grid < 1 || words.size() < (uint64_t)grid * grid * grid) return false;
        sdfhip_octree* t = nullptr;
        detail::check(sdfhip_octree_from_data(detail::defaultContext(), words.data(), words.size(), SDFHIP_HOST, box, box + 3, grid, depth, vr, mb, &t));
        release();
        mTree = t;
        mBox = BoundingBox(glm::vec3(box[0], box[1], box[2]), glm::vec3(box[3], box[4], box[5]));
        mValueRange = vr; mMinBorderValue = mb; mStartGridSize = grid; mStartGridXY = grid * grid; mMaxDepth = depth;
        mStartGridCellSize = (mBox.max.x - mBox.min.x) / static_cast<float>(mStartGridSize);      // load(): OctreeSdf.h:234-236
        mOctreeData.resize(words.size());
        std::memcpy(mOctreeData.data(), words.data(), words.size() * sizeof(uint32_t));
        return true;
    }

protected:
    void writePayload(std::ostream& os) const override {
        const float box[6] = {mBox.min.x, mBox.min.y, mBox.min.z, mBox.max.x, mBox.max.y, mBox.max.z};
        detail::put(os, box); detail::put(os, (int32_t)mStartGridSize); detail::put(os, (uint32_t)mMaxDepth); detail::put(os, mValueRange); detail::put(os, mMinBorderValue);
        detail::putVec(os, reinterpret_cast<const uint32_t*>(mOctreeData.data()), (uint64_t)mOctreeData.size());
    }

private:
    void build(const Mesh& mesh, BoundingBox box, uint32_t depth, uint32_t startDepth, TerminationRule rule, TerminationRuleParams params,
               InitAlgorithm alg, uint32_t numThreads) {
        const BoundingBox& mb = mesh.getBoundingBox();     // only a computed box (file loader / computeBoundingBox) enables seam welding
        const float mbox[6] = {mb.min.x, mb.min.y, mb.min.z, mb.max.x, mb.max.y, mb.max.z};
        sdfhip_octree_params p{};
        p.box_min[0] = box.min.x; p.box_min[1] = box.min.y; p.box_min[2] = box.min.z;
        p.box_max[0] = box.max.x; p.box_max[1] = box.max.y; p.box_max[2] = box.max.z;
        p.depth = depth; p.start_depth = startDepth; p.rule = (int32_t)rule; p.rule_params[0] = params[0]; p.rule_params[1] = params[1];
        p.algorithm = (int32_t)alg; p.layout = numThreads < 2 ? SDFHIP_LAYOUT_GLOBAL_DFS : SDFHIP_LAYOUT_SUBTREES; p.fit_mode = SDFHIP_FIT_EXACT;
        if (sdfhip_multi* multi = detail::defaultMulti()) {
            // SDFLIB_DEVICES: sharded over the devices (the reference's numThreads >= 2 array layout), reassembled on each of them
            p.layout = SDFHIP_LAYOUT_SUBTREES;
            std::vector<sdfhip_octree*> trees((size_t)sdfhip_multi_size(multi), nullptr);
            detail::check(sdfhip_multi_octree_build(multi, reinterpret_cast<const float*>(mesh.getVertices().data()), (uint32_t)mesh.getVertices().size(), mesh.getIndices().data(),
                                                    (uint32_t)(mesh.getIndices().size() / 3), (mb.min.x <= mb.max.x) ? mbox : nullptr, &p, nullptr, trees.data()));
            mTree = trees[0]; mReplicas.assign(trees.begin() + 1, trees.end());
        } else {
            sdfhip_ctx* ctx = detail::defaultContext();
            sdfhip_mesh* m = nullptr;
            // an OctreeSdf always needs the sphere BVH: it is planned on host threads while the device prepares the TriangleData
            detail::check(sdfhip_mesh_create_opt(ctx, reinterpret_cast<const float*>(mesh.getVertices().data()), (uint32_t)mesh.getVertices().size(),
                                                 mesh.getIndices().data(), (uint32_t)(mesh.getIndices().size() / 3),
                                                 (mb.min.x <= mb.max.x) ? mbox : nullptr, SDFHIP_MESH_PLAN_BVH_EARLY, &m));
            int rc = sdfhip_octree_build(ctx, m, &p, &mTree);
            sdfhip_mesh_destroy(m);
            detail::check(rc);
        }
        sdfhip_octree_info info;
        detail::check(sdfhip_octree_get_info(mTree, &info));
        mBox = BoundingBox(glm::vec3(info.box_min[0], info.box_min[1], info.box_min[2]), glm::vec3(info.box_max[0], info.box_max[1], info.box_max[2]));
        mValueRange = info.value_range; mMinBorderValue = info.min_border_value; mStartGridSize = info.start_grid_size;
        mStartGridXY = mStartGridSize * mStartGridSize; mMaxDepth = info.max_depth;
        mStartGridCellSize = info.start_grid_cell_size;      // built: largest extent of the INPUT box / grid size (OctreeSdf.cpp:43-52), not the stored box's
        mOctreeData.resize(info.num_words);
        detail::check(sdfhip_octree_download(mTree, reinterpret_cast<uint32_t*>(mOctreeData.data()), SDFHIP_HOST));
    }
    void copyFrom(const OctreeSdf& o) {
        mBox = o.mBox; mValueRange = o.mValueRange; mMinBorderValue = o.mMinBorderValue; mStartGridSize = o.mStartGridSize; mStartGridXY = o.mStartGridXY;
        mStartGridCellSize = o.mStartGridCellSize; mMaxDepth = o.mMaxDepth; mOctreeData = o.mOctreeData;
        mTree = nullptr; mReplicas.clear();
        if (!o.mTree || mOctreeData.empty()) return;
        const float box[6] = {mBox.min.x, mBox.min.y, mBox.min.z, mBox.max.x, mBox.max.y, mBox.max.z};
        detail::check(sdfhip_octree_from_data(detail::defaultContext(), reinterpret_cast<const uint32_t*>(mOctreeData.data()), mOctreeData.size(), SDFHIP_HOST, box, box + 3,
                                              mStartGridSize, mMaxDepth, mValueRange, mMinBorderValue, &mTree));
        detail::check(sdfhip_octree_set_start_grid_cell_size(mTree, mStartGridCellSize));      // a built tree's cell size is the build's, not the stored box's (OctreeSdf.cpp:43-52)
    }
    void release() {
        if (mTree) sdfhip_octree_destroy(mTree);
        for (sdfhip_octree* t : mReplicas) if (t) sdfhip_octree_destroy(t);
        mTree = nullptr; mReplicas.clear();
    }
    static float fractf(float x) { return x - std::floor(x); }
    float eval(glm::vec3 p, glm::vec3* grad, glm::vec3& g) const {
        float f[3] = {(p.x - mBox.min.x) / mStartGridCellSize, (p.y - mBox.min.y) / mStartGridCellSize, (p.z - mBox.min.z) / mStartGridCellSize};
        int ip[3];
        for (int a = 0; a < 3; a++) { const float fl = std::floor(f[a]); ip[a] = (int)fl; f[a] = f[a] - fl; }
        if (ip[0] < 0 || ip[0] >= mStartGridSize || ip[1] < 0 || ip[1] >= mStartGridSize || ip[2] < 0 || ip[2] >= mStartGridSize) {
            // outside the start grid: same value as the batched kernel (box distance + min border value)
            float d; glm::vec3 gg;
            getDistances(&p, 1, &d, grad ? &gg : nullptr);
            if (grad) g = gg;
            return d;
        }
        const OctreeNode* node = &mOctreeData[ip[2] * mStartGridXY + ip[1] * mStartGridSize + ip[0]];
        while (!node->isLeaf()) {
            const uint32_t c = ((f[2] >= 0.5f) ? 4u : 0u) + ((f[1] >= 0.5f) ? 2u : 0u) + ((f[0] >= 0.5f) ? 1u : 0u);
            node = &mOctreeData[node->getChildrenIndex() + c];
            for (int a = 0; a < 3; a++) f[a] = fractf(2.0f * f[a]);
        }
        const float* c = &mOctreeData[node->getChildrenIndex()].value;
        auto term = [&](float t, int i, int j, int k) { for (int a = 0; a < i; a++) t = t * f[0]; for (int a = 0; a < j; a++) t = t * f[1]; for (int a = 0; a < k; a++) t = t * f[2]; return t; };
        if (grad) {
            float gr[3] = {0, 0, 0}; bool first[3] = {true, true, true};
            for (int n = 0; n < 64; n++) {
                const int e[3] = {n & 3, (n >> 2) & 3, n >> 4};
                for (int a = 0; a < 3; a++) {
                    if (e[a] == 0) continue;
                    const float t = term((float)e[a] * c[n], e[0] - (a == 0), e[1] - (a == 1), e[2] - (a == 2));
                    if (first[a]) { gr[a] = t; first[a] = false; } else gr[a] = gr[a] + t;
                }
            }
            const float inv = 1.0f / std::sqrt(gr[0] * gr[0] + gr[1] * gr[1] + gr[2] * gr[2]);
            g = glm::vec3(gr[0] * inv, gr[1] * inv, gr[2] * inv);
        }
        // the value in the order of the library this program is linked with (the reference's compile-time choice SDFLIB_USE_ENOKI):
        static const bool enokiOrder = sdfhip_interpolation_flavour() == 1;
        if (enokiOrder) {          // InterpolationMethods.h:383-430: 4-wide power vectors, enoki::dot = (a0 b0 + a1 b1) + (a2 b2 + a3 b3)
            float x[4][4];
            x[0][0] = 1.0f; x[0][1] = f[0]; x[0][2] = f[0] * f[0]; x[0][3] = f[0] * f[0] * f[0];
            for (int j = 1; j < 4; j++) for (int i = 0; i < 4; i++) x[j][i] = f[1] * x[j - 1][i];
            float sum = 0.0f;
            for (int k = 0; k < 4; k++) {
                if (k > 0) for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) x[j][i] = f[2] * x[j][i];
                float d[4];
                for (int j = 0; j < 4; j++) { const float* v = c + 16 * k + 4 * j; d[j] = (x[j][0] * v[0] + x[j][1] * v[1]) + (x[j][2] * v[2] + x[j][3] * v[3]); }
                const float slab = d[0] + d[1] + d[2] + d[3];
                sum = (k == 0) ? slab : sum + slab;
            }
            return sum;
        }
        float acc = 0.0f;
        for (int n = 0; n < 64; n++) acc = acc + term(c[n], n & 3, (n >> 2) & 3, n >> 4);          // :432-439, the literal order
        return acc;
    }

    sdfhip_octree* mTree = nullptr;
    std::vector<sdfhip_octree*> mReplicas;      // SDFLIB_DEVICES: the same tree on the other devices (batched queries are split over them)
    BoundingBox mBox;
    float mValueRange = 0.f, mMinBorderValue = 0.f;
    int mStartGridSize = 0, mStartGridXY = 0;
    float mStartGridCellSize = 0.f;
    uint32_t mMaxDepth = 0;
    std::vector<OctreeNode> mOctreeData;
};
}  // namespace sdflib
#include "SdfLoad.h"
#endif
