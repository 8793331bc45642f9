// The sharded OctreeSdf build driven from C++ through the C ABI alone (no Python, no torch): what a C++ host does on every rank,
// with the all-gather played by host memory — W "ranks" are built one after the other on the same device.
//   build_shard(cells of rank r) -> body sizes -> prefix sums -> emit_shard(absolute offset) -> [all-gather] -> from_data
// On a multi-GPU node the two host buffers emit_shard fills are device buffers (where = SDFHIP_DEVICE) handed to ncclAllGather.
// Built and run by tests/test_cpp_api.py; prints the number of words that differ from the single-device build.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "sdfhip.h"

#define CHECK(x) do { if ((x) != SDFHIP_OK) { std::fprintf(stderr, "%s failed: %s\n", #x, sdfhip_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: %s vertices.bin indices.bin box6.bin [ranks]\n", argv[0]); return 2; }
    auto readAll = [](const char* path) { std::vector<char> b; FILE* f = std::fopen(path, "rb"); if (!f) std::exit(3); std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET); b.resize(n); if (std::fread(b.data(), 1, n, f) != (size_t)n) std::exit(3); std::fclose(f); return b; };
    std::vector<char> vb = readAll(argv[1]), ib = readAll(argv[2]), bb = readAll(argv[3]);
    const int W = argc > 4 ? std::atoi(argv[4]) : 3;
    const float* box = reinterpret_cast<const float*>(bb.data());

    sdfhip_ctx* ctx = nullptr; sdfhip_mesh* mesh = nullptr;
    CHECK(sdfhip_ctx_create(0, nullptr, SDFHIP_STREAM_PRIVATE, &ctx));
    CHECK(sdfhip_mesh_create(ctx, reinterpret_cast<const float*>(vb.data()), (uint32_t)(vb.size() / 12), reinterpret_cast<const uint32_t*>(ib.data()), (uint32_t)(ib.size() / 12), &mesh));

    sdfhip_octree_params p; std::memset(&p, 0, sizeof p);
    for (int a = 0; a < 3; a++) { p.box_min[a] = box[a]; p.box_max[a] = box[3 + a]; }
    p.depth = 6; p.start_depth = 3; p.rule = SDFHIP_RULE_TRAPEZOIDAL; p.rule_params[0] = 1e-3f;
    p.algorithm = SDFHIP_ALG_NO_CONTINUITY; p.layout = SDFHIP_LAYOUT_SUBTREES; p.fit_mode = SDFHIP_FIT_EXACT;
    const uint32_t numCells = 1u << (3 * p.start_depth);

    // single-device build: the expected array
    sdfhip_octree* single = nullptr; sdfhip_octree_info si;
    CHECK(sdfhip_octree_build(ctx, mesh, &p, &single));
    CHECK(sdfhip_octree_get_info(single, &si));
    std::vector<uint32_t> expect(si.num_words);
    CHECK(sdfhip_octree_download(single, expect.data(), SDFHIP_HOST));

    // every rank: its shard
    std::vector<sdfhip_octree*> shard(W, nullptr); std::vector<sdfhip_octree_info> info(W);
    for (int r = 0; r < W; r++) {
        sdfhip_octree_params q = p;
        q.cell_begin = (uint32_t)((uint64_t)numCells * r / W); q.cell_end = (uint32_t)((uint64_t)numCells * (r + 1) / W);
        CHECK(sdfhip_octree_build_shard(ctx, mesh, &q, &shard[r]));
        CHECK(sdfhip_octree_get_info(shard[r], &info[r]));
    }
    // "all-gather" of the body sizes -> absolute offsets; emit; "all-gather" of grid slices and bodies into the full array
    uint64_t total = numCells;
    for (int r = 0; r < W; r++) total += info[r].body_words;
    std::vector<uint32_t> full(total);
    uint64_t off = numCells; float valueRange = 0.f, minBorder = INFINITY;
    for (int r = 0; r < W; r++) {
        CHECK(sdfhip_octree_emit_shard(shard[r], off, full.data() + info[r].cell_begin, full.data() + off, SDFHIP_HOST));
        off += info[r].body_words;
        valueRange = info[r].value_range > valueRange ? info[r].value_range : valueRange;          // max / min all-reduce
        minBorder = info[r].min_border_value < minBorder ? info[r].min_border_value : minBorder;
    }
    sdfhip_octree* tree = nullptr;
    CHECK(sdfhip_octree_from_data(ctx, full.data(), total, SDFHIP_HOST, si.box_min, si.box_max, si.start_grid_size, si.max_depth, valueRange, minBorder, &tree));
    CHECK(sdfhip_octree_set_start_grid_cell_size(tree, info[0].start_grid_cell_size));      // a BUILT tree's cell size, not a loaded one's (sdfhip.h)

    size_t mism = (total != expect.size()) ? (size_t)-1 : 0;
    if (!mism) for (size_t i = 0; i < total; i++) mism += full[i] != expect[i];
    std::printf("ranks %d words %llu sharded-vs-single mismatches %zu value_range_equal %d min_border_equal %d\n", W, (unsigned long long)total, mism,
                (int)(valueRange == si.value_range), (int)(minBorder == si.min_border_value));
    // the reassembled tree answers queries like the single one
    const float q3[3] = {0.3f * (box[0] + box[3]), 0.21f, -0.17f}; float d1 = 0, d2 = 0;
    CHECK(sdfhip_octree_query(tree, q3, 1, &d1, nullptr, SDFHIP_HOST, SDFHIP_EVAL_EXACT));
    CHECK(sdfhip_octree_query(single, q3, 1, &d2, nullptr, SDFHIP_HOST, SDFHIP_EVAL_EXACT));
    std::printf("query_equal %d\n", (int)(d1 == d2));
    for (int r = 0; r < W; r++) sdfhip_octree_destroy(shard[r]);
    sdfhip_octree_destroy(tree); sdfhip_octree_destroy(single); sdfhip_mesh_destroy(mesh); sdfhip_ctx_destroy(ctx);
    return 0;
}
