// The in-process multi-GPU build of the C ABI (sdfhip_multi_*): N contexts, shards on host threads, in-place all-gather-v over RCCL
// (distinct devices) or device-to-device copies (the same device listed several times: one-GPU boxes).  Every replica of every
// structure must equal the single-device build word for word.  Built and run by tests/test_cpp_api.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "sdfhip.h"

#define CHECK(x) do { if ((x) != SDFHIP_OK) { std::fprintf(stderr, "%s failed: %s\n", #x, sdfhip_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: %s vertices.bin indices.bin box6.bin dev[,dev...]\n", argv[0]); return 2; }
    auto readAll = [](const char* path) { std::vector<char> b; FILE* f = std::fopen(path, "rb"); if (!f) std::exit(3); std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET); b.resize(n); if (std::fread(b.data(), 1, n, f) != (size_t)n) std::exit(3); std::fclose(f); return b; };
    std::vector<char> vb = readAll(argv[1]), ib = readAll(argv[2]), bb = readAll(argv[3]);
    const float* xyz = reinterpret_cast<const float*>(vb.data()); const uint32_t nv = (uint32_t)(vb.size() / 12);
    const uint32_t* idx = reinterpret_cast<const uint32_t*>(ib.data()); const uint32_t nt = (uint32_t)(ib.size() / 12);
    const float* box = reinterpret_cast<const float*>(bb.data());
    std::vector<int> devs;
    for (char* tok = std::strtok(argv[4], ","); tok; tok = std::strtok(nullptr, ",")) devs.push_back(std::atoi(tok));
    const int W = (int)devs.size();

    sdfhip_multi* M = nullptr;
    CHECK(sdfhip_multi_create(devs.data(), W, &M));
    std::printf("ranks %d transport %s\n", sdfhip_multi_size(M), sdfhip_multi_transport(M));

    // the single-device references
    sdfhip_ctx* ctx = nullptr; sdfhip_mesh* mesh = nullptr;
    CHECK(sdfhip_ctx_create(devs[0], nullptr, SDFHIP_STREAM_PRIVATE, &ctx));
    CHECK(sdfhip_mesh_create(ctx, xyz, nv, idx, nt, &mesh));
    sdfhip_octree_params p; std::memset(&p, 0, sizeof p);
    for (int a = 0; a < 3; a++) { p.box_min[a] = box[a]; p.box_max[a] = box[3 + a]; }
    p.depth = 6; p.start_depth = 3; p.rule = SDFHIP_RULE_TRAPEZOIDAL; p.rule_params[0] = 1e-3f; p.layout = SDFHIP_LAYOUT_SUBTREES; p.fit_mode = SDFHIP_FIT_EXACT;
    const float q3[3] = {0.3f * (box[0] + box[3]), 0.21f, -0.17f};

    for (int alg : {SDFHIP_ALG_NO_CONTINUITY, SDFHIP_ALG_CONTINUITY}) {
        p.algorithm = alg;
        sdfhip_octree* single = nullptr; sdfhip_octree_info si;
        CHECK(sdfhip_octree_build(ctx, mesh, &p, &single));
        CHECK(sdfhip_octree_get_info(single, &si));
        std::vector<uint32_t> expect(si.num_words), got(si.num_words);
        CHECK(sdfhip_octree_download(single, expect.data(), SDFHIP_HOST));
        float d0 = 0; CHECK(sdfhip_octree_query(single, q3, 1, &d0, nullptr, SDFHIP_HOST, SDFHIP_EVAL_EXACT));
        std::vector<sdfhip_octree*> trees(W, nullptr);
        CHECK(sdfhip_multi_octree_build(M, xyz, nv, idx, nt, nullptr, &p, nullptr, trees.data()));
        sdfhip_multi_stats st; CHECK(sdfhip_multi_get_stats(M, &st));
        size_t mism = 0; int same = 1;
        for (int r = 0; r < W; r++) {
            sdfhip_octree_info ti; CHECK(sdfhip_octree_get_info(trees[r], &ti));
            if (ti.num_words != si.num_words) { mism = (size_t)-1; break; }
            CHECK(sdfhip_octree_download(trees[r], got.data(), SDFHIP_HOST));
            for (size_t i = 0; i < got.size(); i++) mism += got[i] != expect[i];
            float d1 = 0; CHECK(sdfhip_octree_query(trees[r], q3, 1, &d1, nullptr, SDFHIP_HOST, SDFHIP_EVAL_EXACT));
            same = same && d1 == d0 && ti.value_range == si.value_range && ti.min_border_value == si.min_border_value && (alg == SDFHIP_ALG_CONTINUITY || ti.num_leaves == si.num_leaves);
            sdfhip_octree_destroy(trees[r]);
        }
        std::printf("octree algorithm %d words %llu replicas-vs-single mismatches %zu scalars_equal %d bytes_exchanged %llu uses_rccl %d\n", alg, (unsigned long long)si.num_words, mism, same,
                    (unsigned long long)st.bytes_exchanged, st.uses_rccl);
        sdfhip_octree_destroy(single);
    }
    {   // ExactOctreeSdf
        sdfhip_exact* single = nullptr; sdfhip_exact_info si;
        CHECK(sdfhip_exact_build(ctx, mesh, box, box + 3, 6, 3, 32, &single));
        CHECK(sdfhip_exact_get_info(single, &si));
        std::vector<uint32_t> n0(2 * si.num_nodes), s0(si.num_set_words + 1), n1(2 * si.num_nodes), s1(si.num_set_words + 1);
        std::vector<uint8_t> h0(si.num_nodes), m0(si.num_mask_bytes + 1), h1(si.num_nodes), m1(si.num_mask_bytes + 1);
        CHECK(sdfhip_exact_download(single, n0.data(), h0.data(), s0.data(), m0.data()));
        float d0 = 0; uint32_t t0 = 0; CHECK(sdfhip_exact_query(single, q3, 1, &d0, nullptr, &t0, SDFHIP_HOST));
        std::vector<sdfhip_exact*> trees(W, nullptr); std::vector<sdfhip_mesh*> meshes(W, nullptr);
        CHECK(sdfhip_multi_exact_build(M, xyz, nv, idx, nt, nullptr, box, box + 3, 6, 3, 32, meshes.data(), trees.data()));
        size_t mism = 0; int same = 1;
        for (int r = 0; r < W; r++) {
            sdfhip_exact_info ti; CHECK(sdfhip_exact_get_info(trees[r], &ti));
            if (ti.num_nodes != si.num_nodes || ti.num_set_words != si.num_set_words || ti.num_mask_bytes != si.num_mask_bytes) {
                std::printf("exact replica %d sizes %llu/%llu/%llu, single %llu/%llu/%llu\n", r, (unsigned long long)ti.num_nodes, (unsigned long long)ti.num_set_words, (unsigned long long)ti.num_mask_bytes,
                            (unsigned long long)si.num_nodes, (unsigned long long)si.num_set_words, (unsigned long long)si.num_mask_bytes);
                mism = (size_t)-1; break;
            }
            CHECK(sdfhip_exact_download(trees[r], n1.data(), h1.data(), s1.data(), m1.data()));
            for (size_t i = 0; i < si.num_nodes; i++) mism += (n1[2 * i] != n0[2 * i]) + (h1[i] != h0[i]) + (h0[i] && n1[2 * i + 1] != n0[2 * i + 1]);
            for (size_t i = 0; i < si.num_set_words; i++) mism += s1[i] != s0[i];
            for (size_t i = 0; i < si.num_mask_bytes; i++) mism += m1[i] != m0[i];
            float d1 = 0; uint32_t t1 = 0; CHECK(sdfhip_exact_query(trees[r], q3, 1, &d1, nullptr, &t1, SDFHIP_HOST));
            same = same && d1 == d0 && t1 == t0 && ti.max_triangles_in_leafs == si.max_triangles_in_leafs && ti.cull_tests >= si.cull_tests;
        }
        for (int r = 0; r < W; r++) sdfhip_exact_destroy(trees[r]);
        for (int r = 0; r < W; r++) sdfhip_mesh_destroy(meshes[r]);
        std::printf("exact nodes %llu replicas-vs-single mismatches %zu scalars_equal %d\n", (unsigned long long)si.num_nodes, mism, same);
        sdfhip_exact_destroy(single);
    }
    sdfhip_mesh_destroy(mesh); sdfhip_ctx_destroy(ctx); sdfhip_multi_destroy(M);
    return 0;
}
