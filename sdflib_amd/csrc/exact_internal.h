// ExactOctreeSdf device object shared by the build and query translation units.  PRODUCT code.
#pragma once
#include "sdfhip_internal.h"
#include "dev_math.h"
#include "dev_tricubic.h"   // stencil tables (child corner sources, mid-point positions)
#include <memory>
#include <mutex>

namespace sdfhip {

// One breadth-first level of the Exact construction.
struct ExLevel {
    uint32_t depth = 0, n = 0; float half = 0.f;
    DevBuf<float> center; DevBuf<uint32_t> coord;
    DevBuf<uint32_t> cornerTri;             // 8 n : nearest triangle of every corner
    DevBuf<uint32_t> pOff, pLen;            // n : the PARENT's list (input of the culling), offsets into the previous level's list
    DevBuf<uint32_t> list, listOff, listLen; uint32_t listTotal = 0;    // this level's culled lists, packed
    DevBuf<uint32_t> flag, inner, childBase; uint32_t numInner = 0;
    DevBuf<uint32_t> midTri;                // 19 n
    DevBuf<uint32_t> ulist, uLen;           // merge step: sorted union of the children's lists (same offsets as `list`)
    DevBuf<uint32_t> maskBytes, maskOff; uint32_t maskTotal = 0; DevBuf<uint8_t> masks;   // 8 masks per merged node
    DevBuf<uint32_t> aNode, aSet, aMask;    // sizes of the subtree in the three output arrays
    DevBuf<uint32_t> pos, blk, setPos, maskPos, maskIdx;
};

}  // namespace sdfhip

// scratch of the leaf-sorted batched query, kept with the tree (grow-only) so that a call costs no hipMalloc / hipFree
struct sdfhip_exact_scratch {
    sdfhip::DevBuf<uint32_t> key, keyS, qi, qiS; sdfhip::DevBuf<unsigned char> tmp; std::mutex lock;
};

struct sdfhip_exact {
    sdfhip_exact_scratch scratch;
    sdfhip_ctx* ctx = nullptr;
    sdfhip_mesh* mesh = nullptr;           // TriangleData lives in the mesh (it must outlive the tree) ...
    sdfhip::DevBuf<float> ownTri;          // ... or in this buffer for trees created by sdfhip_exact_from_data
    sdfhip::DevBuf<float> ownFrames;
    const float* tri() const { return ownTri.p ? ownTri.p : mesh->dTri.p; }
    const float* frames() const { return ownFrames.p ? ownFrames.p : mesh->dFrames.p; }
    sdfhip_exact_info info{};
    float cellSize = 0.f;
    // per node that a query can end in: {set index, first mask offset, second mask offset, entries of the set} — what the walk from the
    // start grid finds on its way (exact_query.hip, ensureLeafCtx): made once, on the first batched query, 16 bytes per node
    sdfhip::DevBuf<uint32_t> leafCtx; bool leafCtxReady = false; std::mutex leafCtxLock;
    // the survivors of every such node's set, decoded once (exact_query.hip, ensureLeafLists): triangle ids in list order, and per node
    // {offset, count}; listsState: 0 = not made yet, 1 = ready, 2 = not kept (larger than SDFHIP_EXACT_LISTS_MB: the decoding kernel answers)
    sdfhip::DevBuf<uint32_t> leafLists, leafList; int listsState = 0; uint64_t listEntries = 0;
    sdfhip::DevBuf<uint32_t> nodes;        // 2 words per node
    sdfhip::DevBuf<uint8_t> hasTri;
    sdfhip::DevBuf<uint32_t> sets;
    sdfhip::DevBuf<uint8_t> masks;
    std::vector<std::unique_ptr<sdfhip::ExLevel>> levels;
    bool built = false;
    // host copies for the scalar / few-point entry (exact_query.hip, ensureHostCopy): fetched on first use
    std::vector<uint32_t> hNodes, hSets; std::vector<uint8_t> hMasks; std::vector<float> hTri; bool hostReady = false; std::mutex hostLock;
    // emission plan: per start cell (local order) the offset of its body / sets / masks relative to the first one emitted
    std::vector<uint32_t> relB, relS, relM; uint64_t bodyNodes = 0; uint32_t sod = 0;
    // sharded build (sdfhip_exact_build_shard): the start cells of this shard (z-major ids, ascending); levels stay alive until emit
    bool isShard = false; std::vector<uint32_t> shardCells;
};
