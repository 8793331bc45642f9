// OctreeSdf device object shared by the build and query translation units.  PRODUCT code.
#pragma once
#include "sdfhip_internal.h"
#include <memory>

namespace sdfhip {

constexpr uint32_t LEAF_BIT = 1u << 31;
constexpr uint32_t MARK_BIT = 1u << 30;
constexpr uint32_t INDEX_MASK = ~(LEAF_BIT | MARK_BIT);

// One breadth-first level of the construction (structure of arrays in HBM).
struct BuildLevel {
    uint32_t depth = 0;
    uint32_t n = 0;                 // nodes at this level
    float half = 0.f;               // half edge length of a node (same for the whole level)
    DevBuf<float> center;           // 3 n
    DevBuf<uint32_t> coord;         // n : x | y << 10 | z << 20 (integer cell coordinates at this depth)
    DevBuf<float> corner;           // 32 n : 8 corners x [f, fx, fy, fz]   (mixed derivatives are 0 in NO_CONTINUITY)
    DevBuf<float> mid;              // 76 n : 19 mid-points x [f, fx, fy, fz]
    DevBuf<uint32_t> flag;          // n : 1 = leaf
    DevBuf<uint32_t> inner;         // n : 1 = inner (scan input)
    DevBuf<uint32_t> childBase;     // n : index of child 0 in the next level (inner nodes)
    DevBuf<float> coeff;            // 64 n : coefficients of leaves
    DevBuf<uint32_t> alloc;         // n : words of the node's block + all descendants' blocks
    DevBuf<uint32_t> pos, blk;      // n : absolute position of the node word / of its block
    uint32_t numInner = 0, numLeaves = 0;
    bool presampled = false;        // created and sampled up front (levels down to the start depth exist a priori)
};

// leaf-driven lattice evaluation (plan: octree_query.hip; column kernels: octree_lattice.hip)
constexpr int kLatMaxLevels = 14;           // 3 x 13 bits of local cell coordinates in a sort key
struct LatCols { uint32_t mx[kLatMaxLevels], my[kLatMaxLevels]; int levels; uint32_t G; };
int latticeColumnsLaunch(hipStream_t st, bool exact, const float* coef, const float* F, const uint16_t* ranges, const uint4* desc, const uint32_t* sortedLeaf, uint32_t waves,
                         const LatCols& C, uint32_t nx, uint32_t ny, float* d, float* g);

}  // namespace sdfhip

struct sdfhip_octree {
    sdfhip_ctx* ctx = nullptr;
    sdfhip_octree_info info{};
    sdfhip_octree_params params{};
    sdfhip::DevBuf<uint32_t> data;          // full node array (when available): the reference's mOctreeData layout, what download / emit return
    bool dataPinned = false;                // sdfhip_octree_device_words handed the array's address out: it is not released automatically any more
    bool hasData = false;                   // an assembled array EXISTS; it need not be resident: a compacted tree (sdfhip_octree_compact, or
                                            // automatically above SDFHIP_COMPACT_ABOVE_MB) keeps the query layout only and rebuilds it on demand
    // Query-side layout — what the builders emit (round 5: a built tree is born with it and has NO resident array; the array is rebuilt
    // from it when download / device_words / a .bin file asks), or what the first query derives from `data` for a tree that arrived as an
    // array (octree_query.hip, ensureQueryLayout): the node words alone, packed
    // breadth-first (a few MB: the dependent loads of the walk stay in L2), and the leaves' coefficients as 256-byte-ALIGNED blocks
    // (a block is exactly two 128-byte lines; in `data` a block starts at any multiple of 4 bytes and straddles three).
    sdfhip::DevBuf<uint32_t> qTopo;         // [G^3 start cells][level 1 blocks]...: inner word = index of the 8-word child block, leaf word = LEAF_BIT | block id
    sdfhip::DevBuf<uint32_t> qOrig;         // the nodes' words as `data` holds them, qTopo order: with qCoef, everything `data` holds
    sdfhip::DevBuf<float> qCoef;            // 64 floats per leaf, block id order
    uint64_t qNodes = 0, qLeaves = 0;
    bool qReady = false;
    std::mutex qLock;
    std::vector<uint32_t> hTopo; std::vector<float> hCoef; bool hReady = false;      // host copies for the scalar entry (octree_query.hip)
    // Leaf-driven lattice evaluation (octree_query.hip, ensureLatticePlan): per level of the packed layout its node count and the
    // number of its first leaf block; every leaf's integer cell coordinates; and the plan of the last lattice asked for.
    std::vector<uint32_t> qLevelNodes, qLevelLeafBase;      // [levels], [levels + 1]
    sdfhip::DevBuf<uint32_t> qLeafCell;     // 2 words per leaf: x | y << 16, z   (cell coordinates at the leaf's own level)
    bool cellsReady = false;
    struct LatticeClass { uint32_t level, leafBase, leafCount, mx, my, mz; };
    struct LatticePlan {
        float origin[3] = {0, 0, 0}, step[3] = {0, 0, 0};
        uint32_t n[3] = {0, 0, 0};
        bool valid = false, leafDriven = false, outside = false;
        sdfhip::DevBuf<float> F;            // nx + ny + nz : start-cell coordinate of every lattice index per axis (the point walk's first value)
        sdfhip::DevBuf<uint16_t> ranges;    // 6 per leaf : x0, x1, y0, y1, z0, z1 (lattice indices inside the leaf, half open)
        std::vector<LatticeClass> classes;
        sdfhip::DevBuf<uint32_t> groupWaveBase, groupLeafBase, sortedLeaf, waveDesc;      // the wave list (octree_query.hip, k_lat_waves); waveDesc = uint4 per wave
        uint32_t waves = 0;
    } lattice;
    // construction state kept between build_shard and emit_shard
    std::vector<std::unique_ptr<sdfhip::BuildLevel>> levels;   // index = depth - startOctreeDepth
    uint32_t startOctreeDepth = 0;
    bool built = false;
    float cellSize = 0.f;
};

namespace sdfhip {
// octree_query.hip: the node array of a tree whose resident copy may have been released in favour of the query layout
int octreeMaterialize(sdfhip_octree* tree);                                   // makes tree->data resident again
int octreeDownload(sdfhip_octree* tree, uint32_t* out_words, int where);      // without keeping it resident
int octreeLayoutFromArray(sdfhip_octree* tree, const uint32_t* data);         // the query layout of the tree whose reference array is `data` (device)
}
