// Multi-GPU construction inside ONE process, for C / C++ callers: one context per device, the build sharded by start-grid cell
// exactly as sdflib_amd/distributed.py does it with one process per GPU, the node arrays reassembled over RCCL.  PRODUCT code.
//
// Reference decomposition reproduced: the OpenMP loop over start cells + merge / rebase of OctreeSdf
// (src/sdf/OctreeSdfDepthFirst.h:433-503) and of ExactOctreeSdf (include/SdfLib/ExactOctreeSdfDepthFirst.h:534-622).
//   1. every device gets the mesh (TriangleData is computed per device: 5 ms at 1.3 M triangles) and builds the sphere BVH itself
//      (the device builder, identical trees); with the host planner the tree is planned ONCE and installed on every device;
//   2. the start cells are cut into contiguous ranges balanced by a vertex-occupancy estimate; device r builds its range
//      (sdfhip_octree_build_shard / sdfhip_exact_build_shard) on its own host thread;
//   3. prefix sums of the shards' sizes give every shard its ABSOLUTE offsets; each device emits its part straight into its copy of
//      the full array(s) at those offsets;
//   4. all-gather-v: inside one RCCL group every shard's segment is broadcast in place from its owner (ncclBroadcast over xGMI; no
//      padding, no staging).  Devices that appear twice in the list (tests on a one-GPU box: RCCL refuses duplicate devices) use
//      device-to-device copies instead — the only difference between the two transports;
//   5. every device wraps its copy (sdfhip_octree_from_data / sdfhip_exact_from_parts): N identical trees, queries are then split
//      over them by the caller.  CONTINUITY trees are not separable by cell: device 0 builds, the array is broadcast.
// RCCL is bound at run time (dlopen of librccl.so, or the copy torch already loaded): libsdfhip has no link-time dependency on it.
#include "sdfhip_internal.h"
#include "octree_internal.h"
#include "exact_internal.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <thread>
#include <condition_variable>
#include <algorithm>
#include <cmath>
#include <cstring>

namespace sdfhip {

struct Rccl {
    void* handle = nullptr; bool ok = false;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load() {
        if (ok) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (handle) break;
        }
        if (!handle) return false;
        auto sym = [&](const char* n) { return dlsym(handle, n); };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll"); CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart"); GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Broadcast = (decltype(Broadcast))sym("ncclBroadcast"); GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
        ok = CommInitAll && CommDestroy && GroupStart && GroupEnd && Broadcast && AllReduce && GetErrorString;
        return ok;
    }
};
static Rccl& rccl() { static Rccl r; return r; }

}  // namespace sdfhip

struct sdfhip_multi {
    std::vector<int> devices;
    std::vector<sdfhip_ctx*> ctx;
    std::vector<ncclComm_t> comm;          // empty with the copy transport
    bool useRccl = false;
    uint64_t bytesExchanged = 0;           // of the last build: bytes every device received
    double lastExchangeSeconds = 0, lastShardSeconds = 0, lastBvhSeconds = 0;
};

namespace sdfhip {

#define SDF_NCCL(expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) { setError("%s failed: %s", #expr, rccl().GetErrorString(_r)); return SDFHIP_E_HIP; } } while (0)

// run fn(rank) on one host thread per device; the first non-zero status wins
template <typename F> static int perRank(int n, F fn) {
    std::vector<int> rc(n, SDFHIP_OK); std::vector<std::string> msg(n);
    std::vector<std::thread> th;
    for (int r = 1; r < n; r++) th.emplace_back([&, r]() { rc[r] = fn(r); if (rc[r] != SDFHIP_OK) msg[r] = sdfhip_last_error(); });
    rc[0] = fn(0); if (rc[0] != SDFHIP_OK) msg[0] = sdfhip_last_error();
    for (std::thread& t : th) t.join();
    for (int r = 0; r < n; r++) if (rc[r] != SDFHIP_OK) { setError("device %d: %s", r, msg[r].c_str()); return rc[r]; }
    return SDFHIP_OK;
}

// Segments [off, off + count) of per-device arrays buf[r] (same layout on every device), each owned by one rank: after the call every
// device holds every segment.  elemSize in bytes.
struct Segment { int owner; uint64_t off, count; };
static int allGatherV(sdfhip_multi* M, const std::vector<void*>& buf, const std::vector<Segment>& segs, size_t elemSize) {
    const int n = (int)M->ctx.size();
    if (n == 1) return SDFHIP_OK;
    for (int r = 0; r < n; r++) { SDF_HIP_CHECK(hipSetDevice(M->devices[r])); SDF_HIP_CHECK(hipStreamSynchronize(M->ctx[r]->stream)); }     // the emitted parts are complete
    if (M->useRccl) {
        SDF_NCCL(rccl().GroupStart());
        for (const Segment& s : segs) {
            if (s.count == 0) continue;
            for (int r = 0; r < n; r++) {
                char* p = (char*)buf[r] + s.off * elemSize;
                ncclResult_t e = rccl().Broadcast(p, p, s.count * elemSize, ncclUint8, s.owner, M->comm[r], M->ctx[r]->stream);
                if (e != ncclSuccess) { (void)rccl().GroupEnd(); setError("ncclBroadcast failed: %s", rccl().GetErrorString(e)); return SDFHIP_E_HIP; }
            }
        }
        SDF_NCCL(rccl().GroupEnd());
    } else {
        for (const Segment& s : segs) {
            if (s.count == 0) continue;
            for (int r = 0; r < n; r++) {
                if (r == s.owner) continue;
                SDF_HIP_CHECK(hipSetDevice(M->devices[r]));
                SDF_HIP_CHECK(hipMemcpyAsync((char*)buf[r] + s.off * elemSize, (const char*)buf[s.owner] + s.off * elemSize, s.count * elemSize, hipMemcpyDeviceToDevice, M->ctx[r]->stream));
            }
        }
    }
    for (int r = 0; r < n; r++) { SDF_HIP_CHECK(hipSetDevice(M->devices[r])); SDF_HIP_CHECK(hipStreamSynchronize(M->ctx[r]->stream)); }
    for (const Segment& s : segs) M->bytesExchanged += s.count * elemSize;
    return SDFHIP_OK;
}

// ---- CONTINUITY builds on several devices of one process ---------------------------------------------------------------------------------
// The breadth-first builder is not separable by start cell (its second iteration couples neighbouring cells, its post-pass is serial host
// code), so every device builds the WHOLE tree and what is shared out is what costs the time: the BVH traversals of every deduplicated
// sample batch (sdfhip_exchange, sdfhip.h: rank r traverses the 128-sample blocks b with b % world == r, one sum all-reduce of the id
// buffer completes the batch everywhere).  sdflib_amd/distributed.py does this across processes; here the ranks are host threads:
// the all-reduce is ncclAllReduce on the devices' communicators, or — one physical device listed several times, tests — staged copies
// and adds between barriers of the threads.
__global__ void k_add_u32(uint32_t* __restrict__ acc, const uint32_t* __restrict__ other, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] += other[i];
}
struct ThreadBarrier {                     // all `n` threads, or none: a rank that fails releases the others with an error
    std::mutex m; std::condition_variable cv; int n = 0, waiting = 0; uint64_t phase = 0; bool failed = false;
    bool arrive() {
        std::unique_lock<std::mutex> g(m);
        if (failed) return false;
        const uint64_t ph = phase;
        if (++waiting == n) { waiting = 0; phase++; cv.notify_all(); return true; }
        cv.wait(g, [&] { return phase != ph || failed; });
        return phase != ph;          // decided per PHASE: a barrier that completed lets every waiter through, also one that wakes after a later failure (its peers are already in the collective)
    }
    void fail() { std::lock_guard<std::mutex> g(m); failed = true; cv.notify_all(); }
};
struct InProcessExchange {
    sdfhip_multi* M = nullptr; int rank = 0; ThreadBarrier* bar = nullptr; std::vector<InProcessExchange>* all = nullptr;
    DevBuf<uint32_t> buf, stage, acc; uint64_t bytes = 0; double seconds = 0;
    static uint32_t* acquire(void* user, uint64_t count) {
        InProcessExchange* X = (InProcessExchange*)user;
        if (hipSetDevice(X->M->devices[X->rank]) != hipSuccess || X->buf.reserve(count ? count : 1) != SDFHIP_OK) return nullptr;
        if (hipMemsetAsync(X->buf.p, 0, 4 * count, X->M->ctx[X->rank]->stream) != hipSuccess) return nullptr;
        return X->buf.p;
    }
    static int allReduceSum(void* user, uint64_t count) {
        InProcessExchange* X = (InProcessExchange*)user;
        sdfhip_multi* M = X->M; const int n = (int)M->ctx.size(), r = X->rank;
        hipStream_t st = M->ctx[r]->stream;
        const double t0 = nowSeconds();
        X->bytes += 4 * count;
        if (M->useRccl) {
            // all ranks agree on the host to enter the collective, or none does: a rank whose build failed would leave its peers in it for ever
            if (!X->bar->arrive()) { setError("another device's CONTINUITY build failed"); return SDFHIP_E_HIP; }
            SDF_NCCL(rccl().AllReduce(X->buf.p, X->buf.p, count, ncclUint32, ncclSum, M->comm[r], st));
            X->seconds += nowSeconds() - t0;
            return SDFHIP_OK;
        }
        // copy transport: everybody's buffer complete -> sum of all into a private accumulator -> everybody done reading -> back into the buffer
        SDF_HIP_CHECK(hipStreamSynchronize(st));
        if (!X->bar->arrive()) { setError("another device's CONTINUITY build failed"); return SDFHIP_E_HIP; }
        SDF_TRY(X->acc.reserve(count ? count : 1)); SDF_TRY(X->stage.reserve(count ? count : 1));
        SDF_HIP_CHECK(hipMemcpyAsync(X->acc.p, X->buf.p, 4 * count, hipMemcpyDeviceToDevice, st));
        for (int k = 0; k < n; k++) {
            if (k == r) continue;
            SDF_HIP_CHECK(hipMemcpyAsync(X->stage.p, (*X->all)[k].buf.p, 4 * count, hipMemcpyDeviceToDevice, st));
            k_add_u32<<<gridFor(count, 256), 256, 0, st>>>(X->acc.p, X->stage.p, count);
        }
        SDF_HIP_CHECK(hipStreamSynchronize(st));
        if (!X->bar->arrive()) { setError("another device's CONTINUITY build failed"); return SDFHIP_E_HIP; }
        SDF_HIP_CHECK(hipMemcpyAsync(X->buf.p, X->acc.p, 4 * count, hipMemcpyDeviceToDevice, st));
        X->seconds += nowSeconds() - t0;
        return SDFHIP_OK;
    }
};

// work estimate per start cell: vertices in the cell and its 26 neighbours (surface cells subdivide), as distributed.py::cell_weights
static std::vector<double> cellWeights(const float* xyz, uint32_t nv, const float box_min[3], const float box_max[3], uint32_t startDepth) {
    const uint32_t G = 1u << startDepth;
    float size = 0.f, c[3];
    for (int a = 0; a < 3; a++) { size = std::max(size, box_max[a] - box_min[a]); c[a] = box_min[a] + 0.5f * (box_max[a] - box_min[a]); }
    std::vector<double> occ((size_t)G * G * G, 0.0), acc((size_t)G * G * G, 1.0);
    for (uint32_t v = 0; v < nv; v++) {
        int ijk[3];
        for (int a = 0; a < 3; a++) { const float f = (xyz[3 * (size_t)v + a] - (c[a] - 0.5f * size)) / (size / (float)G); ijk[a] = std::min((int)G - 1, std::max(0, (int)f)); }
        occ[((size_t)ijk[2] * G + ijk[1]) * G + ijk[0]] += 1.0;
    }
    for (int z = 0; z < (int)G; z++) for (int y = 0; y < (int)G; y++) for (int x = 0; x < (int)G; x++) {
        double s = 0;
        for (int dz = -1; dz <= 1; dz++) for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
            const int X = x + dx, Y = y + dy, Z = z + dz;
            if (X >= 0 && X < (int)G && Y >= 0 && Y < (int)G && Z >= 0 && Z < (int)G) s += occ[((size_t)Z * G + Y) * G + X];
        }
        acc[((size_t)z * G + y) * G + x] += s;
    }
    return acc;
}
// contiguous ranges covering [0, n), balanced by weight, at least one cell each
static std::vector<std::pair<uint32_t, uint32_t>> partition(const std::vector<double>& w, int world) {
    const uint32_t n = (uint32_t)w.size();
    std::vector<double> cum(n); double t = 0; for (uint32_t i = 0; i < n; i++) { t += w[i] + 1e-9; cum[i] = t; }
    std::vector<uint32_t> cuts{0};
    // SDFHIP_MULTI_CUTS="c1,c2,..." (world - 1 ascending cell indices) replaces the balanced cuts: tests use it to reassemble
    // deliberately lopsided shards.  Ignored unless it is a valid cut list for this build.
    if (const char* e = getenv("SDFHIP_MULTI_CUTS")) {
        std::vector<uint32_t> given;
        for (const char* q = e; *q;) { char* end; const unsigned long c = strtoul(q, &end, 10); if (end == q) break; given.push_back((uint32_t)c); q = *end == ',' ? end + 1 : end; }
        bool ok = (int)given.size() == world - 1;
        for (size_t i = 0; ok && i < given.size(); i++) ok = given[i] > (i ? given[i - 1] : 0u) && given[i] < n;
        if (ok) {
            std::vector<std::pair<uint32_t, uint32_t>> out;
            given.insert(given.begin(), 0u); given.push_back(n);
            for (int r = 0; r < world; r++) out.emplace_back(given[r], given[r + 1]);
            return out;
        }
    }
    for (int r = 1; r < world; r++) {
        const double target = t * r / world;
        uint32_t c = (uint32_t)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin()) + 1;
        c = std::max(c, cuts.back() + 1); c = std::min(c, n - (uint32_t)(world - r));
        cuts.push_back(c);
    }
    cuts.push_back(n);
    std::vector<std::pair<uint32_t, uint32_t>> out;
    for (int r = 0; r < world; r++) out.emplace_back(cuts[r], cuts[r + 1]);
    return out;
}

// the mesh and the sphere BVH on every device.  The device builder (default) reproduces the reference's tree bit for bit, so every
// device builds its own at the same time (16 ms at 1.31 M triangles) instead of waiting for device 0 and 105 MB of copies; logical
// devices sharing one GPU, and the host planner (which wants all the cores once), build ONE tree and install it everywhere.
static int meshesEverywhere(sdfhip_multi* M, const float* xyz, uint32_t nv, const uint32_t* idx, uint32_t nt, const float* bbox6, std::vector<sdfhip_mesh*>& mesh) {
    const int n = (int)M->ctx.size();
    mesh.assign(n, nullptr);
    SDF_TRY(perRank(n, [&](int r) { return sdfhip_mesh_create_ex(M->ctx[r], xyz, nv, idx, nt, bbox6, &mesh[r]); }));
    const double t0 = nowSeconds();
    bool distinct = true;
    for (int a = 0; a < n; a++) for (int b = a + 1; b < n; b++) distinct = distinct && M->devices[a] != M->devices[b];
    if (n > 1 && distinct && bvhBuildOnDevice()) SDF_TRY(perRank(n, [&](int r) { return sdfhip_mesh_build_bvh(mesh[r], nullptr); }));
    else {
        SDF_TRY(sdfhip_mesh_build_bvh(mesh[0], nullptr));
        if (n > 1) {
            const size_t nn = nt > 1 ? nt - 1 : 1;
            std::vector<double> sph(8 * nn); std::vector<int32_t> kids(2 * nn);
            SDF_TRY(sdfhip_mesh_bvh_export(mesh[0], sph.data(), kids.data(), SDFHIP_HOST));
            SDF_TRY(perRank(n, [&](int r) { return r == 0 ? SDFHIP_OK : sdfhip_mesh_bvh_import(mesh[r], sph.data(), kids.data(), SDFHIP_HOST); }));
        }
    }
    M->lastBvhSeconds = nowSeconds() - t0;
    return SDFHIP_OK;
}

}  // namespace sdfhip

using namespace sdfhip;

extern "C" {

int sdfhip_multi_create(const int* device_ids, int n, sdfhip_multi** out) {
    SDF_API_BEGIN
    SDF_REQUIRE(device_ids && out && n >= 1 && n <= 64, "bad argument");
    std::unique_ptr<sdfhip_multi> M(new sdfhip_multi());
    M->devices.assign(device_ids, device_ids + n);
    bool distinct = true;
    for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) distinct = distinct && device_ids[i] != device_ids[j];
    for (int r = 0; r < n; r++) {
        sdfhip_ctx* c = nullptr;
        const int rc = sdfhip_ctx_create(device_ids[r], nullptr, SDFHIP_STREAM_PRIVATE, &c);
        if (rc != SDFHIP_OK) { for (sdfhip_ctx* q : M->ctx) sdfhip_ctx_destroy(q); return rc; }
        M->ctx.push_back(c);
    }
    const bool wantCopy = getenv("SDFHIP_MULTI_TRANSPORT") && !strcmp(getenv("SDFHIP_MULTI_TRANSPORT"), "copy");
    if (distinct && !wantCopy) {
        if (!rccl().load()) { for (sdfhip_ctx* q : M->ctx) sdfhip_ctx_destroy(q); setError("librccl.so could not be loaded (needed for %d distinct devices)", n); return SDFHIP_E_UNSUPPORTED; }
        M->comm.resize(n);
        ncclResult_t e = rccl().CommInitAll(M->comm.data(), n, device_ids);
        if (e != ncclSuccess) { for (sdfhip_ctx* q : M->ctx) sdfhip_ctx_destroy(q); setError("ncclCommInitAll failed: %s", rccl().GetErrorString(e)); return SDFHIP_E_HIP; }
        M->useRccl = true;
    }
    *out = M.release();
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_multi_destroy(sdfhip_multi* M) {
    if (!M) return SDFHIP_OK;
    for (ncclComm_t c : M->comm) (void)rccl().CommDestroy(c);
    for (sdfhip_ctx* c : M->ctx) sdfhip_ctx_destroy(c);
    delete M;
    return SDFHIP_OK;
}

int sdfhip_multi_size(sdfhip_multi* M) { return M ? (int)M->ctx.size() : 0; }
sdfhip_ctx* sdfhip_multi_ctx(sdfhip_multi* M, int rank) { return (M && rank >= 0 && rank < (int)M->ctx.size()) ? M->ctx[rank] : nullptr; }
const char* sdfhip_multi_transport(sdfhip_multi* M) { return (M && M->useRccl) ? "rccl" : "copy"; }

int sdfhip_multi_get_stats(sdfhip_multi* M, sdfhip_multi_stats* out) {
    SDF_API_BEGIN
    SDF_REQUIRE(M && out, "NULL argument");
    out->ranks = (int32_t)M->ctx.size(); out->uses_rccl = M->useRccl ? 1 : 0; out->bytes_exchanged = M->bytesExchanged;
    out->seconds_bvh = M->lastBvhSeconds; out->seconds_shards = M->lastShardSeconds; out->seconds_exchange = M->lastExchangeSeconds;
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_multi_octree_build(sdfhip_multi* M, const float* xyz, uint32_t nv, const uint32_t* idx, uint32_t nt, const float* bbox6, const sdfhip_octree_params* params,
                              sdfhip_mesh** out_meshes, sdfhip_octree** out_trees) {
    SDF_API_BEGIN
    SDF_REQUIRE(M && xyz && idx && params && out_trees, "NULL argument");
    const int n = (int)M->ctx.size();
    M->bytesExchanged = 0; M->lastExchangeSeconds = 0; M->lastShardSeconds = 0;
    std::vector<sdfhip_mesh*> mesh;
    SDF_TRY(meshesEverywhere(M, xyz, nv, idx, nt, bbox6, mesh));
    auto cleanup = [&](int rc) { for (sdfhip_mesh* m : mesh) if (m) sdfhip_mesh_destroy(m); return rc; };
    std::vector<sdfhip_octree*> tree(n, nullptr);
    const uint32_t numCells = 1u << (3 * params->start_depth);
    const bool sharded = params->algorithm == SDFHIP_ALG_NO_CONTINUITY && n > 1 && numCells >= (uint32_t)n;
    int rc = SDFHIP_OK;
    if (!sharded && n > 1 && params->algorithm == SDFHIP_ALG_CONTINUITY) {
        // every device builds the whole tree from identical inputs, the traversals are shared out (InProcessExchange above): the trees
        // are bit-identical, nothing is broadcast afterwards
        const double t0 = nowSeconds();
        ThreadBarrier bar; bar.n = n;
        std::vector<InProcessExchange> X(n);
        for (int r = 0; r < n; r++) { X[r].M = M; X[r].rank = r; X[r].bar = &bar; X[r].all = &X; }
        rc = perRank(n, [&](int r) {
            // every error exit releases the peers waiting at the barrier (a rank that never arrives would hold them for ever)
            struct FailGuard { ThreadBarrier& b; bool armed = true; ~FailGuard() { if (armed) b.fail(); } } guard{bar};
            SDF_HIP_CHECK(hipSetDevice(M->devices[r]));
            sdfhip_exchange x{&X[r], &InProcessExchange::acquire, &InProcessExchange::allReduceSum, r, n};
            SDF_TRY(sdfhip_ctx_set_exchange(M->ctx[r], &x));
            const int brc = sdfhip_octree_build(M->ctx[r], mesh[r], params, &tree[r]);
            (void)sdfhip_ctx_set_exchange(M->ctx[r], nullptr);
            if (brc == SDFHIP_OK) guard.armed = false;
            return brc;
        });
        for (int r = 0; r < n; r++) { (void)hipSetDevice(M->devices[r]); X[r].buf.release(); X[r].stage.release(); X[r].acc.release(); }
        M->lastShardSeconds = nowSeconds() - t0;
        M->lastExchangeSeconds = X[0].seconds; M->bytesExchanged = X[0].bytes;
        if (rc != SDFHIP_OK) { for (int r = 0; r < n; r++) if (tree[r]) { sdfhip_octree_destroy(tree[r]); tree[r] = nullptr; } return cleanup(rc); }
    } else if (!sharded) {
        // a single device (or too few start cells to shard): device 0 builds, the array is broadcast
        const double t0 = nowSeconds();
        rc = sdfhip_octree_build(M->ctx[0], mesh[0], params, &tree[0]);
        if (rc != SDFHIP_OK) return cleanup(rc);
        M->lastShardSeconds = nowSeconds() - t0;
        if (n > 1) {
            const double t1 = nowSeconds();
            sdfhip_octree_info i0; sdfhip_octree_get_info(tree[0], &i0);
            std::vector<DevBuf<uint32_t>> full(n); std::vector<void*> buf(n);
            buf[0] = const_cast<uint32_t*>(sdfhip_octree_device_words(tree[0]));
            if (!buf[0]) return cleanup(SDFHIP_E_HIP);
            for (int r = 1; r < n && rc == SDFHIP_OK; r++) { if (hipSetDevice(M->devices[r]) != hipSuccess) rc = SDFHIP_E_HIP; else { rc = full[r].reserve(i0.num_words); buf[r] = full[r].p; } }
            if (rc == SDFHIP_OK) rc = allGatherV(M, buf, {Segment{0, 0, i0.num_words}}, 4);
            for (int r = 1; r < n && rc == SDFHIP_OK; r++) {
                rc = sdfhip_octree_from_data(M->ctx[r], full[r].p, i0.num_words, SDFHIP_DEVICE, i0.box_min, i0.box_max, i0.start_grid_size, i0.max_depth, i0.value_range, i0.min_border_value, &tree[r]);
                if (rc == SDFHIP_OK) rc = sdfhip_octree_set_start_grid_cell_size(tree[r], i0.start_grid_cell_size);
            }
            for (int r = 1; r < n; r++) { (void)hipSetDevice(M->devices[r]); full[r].release(); }
            M->lastExchangeSeconds = nowSeconds() - t1;
        }
    } else {
        float bmin[3], bmax[3]; memcpy(bmin, params->box_min, 12); memcpy(bmax, params->box_max, 12);
        const auto ranges = partition(cellWeights(xyz, nv, bmin, bmax, params->start_depth), n);
        std::vector<sdfhip_octree*> shard(n, nullptr); std::vector<sdfhip_octree_info> info(n);
        const double t0 = nowSeconds();
        rc = perRank(n, [&](int r) {
            sdfhip_octree_params q = *params; q.layout = SDFHIP_LAYOUT_SUBTREES; q.cell_begin = ranges[r].first; q.cell_end = ranges[r].second;
            SDF_TRY(sdfhip_octree_build_shard(M->ctx[r], mesh[r], &q, &shard[r]));
            return sdfhip_octree_get_info(shard[r], &info[r]);
        });
        M->lastShardSeconds = nowSeconds() - t0;
        const double t1 = nowSeconds();
        uint64_t total = numCells; std::vector<uint64_t> off(n);
        for (int r = 0; r < n && rc == SDFHIP_OK; r++) { off[r] = total; total += info[r].body_words; }
        if (rc == SDFHIP_OK && total > (uint64_t)INDEX_MASK) { setError("assembled octree needs %llu words: beyond the 30-bit node index", (unsigned long long)total); rc = SDFHIP_E_TOO_LARGE; }
        std::vector<DevBuf<uint32_t>> full(n); std::vector<void*> buf(n);
        if (rc == SDFHIP_OK) rc = perRank(n, [&](int r) {
            SDF_HIP_CHECK(hipSetDevice(M->devices[r]));
            SDF_TRY(full[r].reserve(total)); buf[r] = full[r].p;
            return sdfhip_octree_emit_shard(shard[r], off[r], full[r].p + ranges[r].first, full[r].p + off[r], SDFHIP_DEVICE);      // absolute indices: no rebase pass afterwards
        });
        if (rc == SDFHIP_OK) {
            std::vector<Segment> segs;
            for (int r = 0; r < n; r++) { segs.push_back(Segment{r, ranges[r].first, ranges[r].second - ranges[r].first}); segs.push_back(Segment{r, off[r], info[r].body_words}); }
            rc = allGatherV(M, buf, segs, 4);
        }
        float valueRange = 0.f, minBorder = INFINITY;
        for (int r = 0; r < n; r++) { valueRange = std::max(valueRange, info[r].value_range); minBorder = std::min(minBorder, info[r].min_border_value); }
        if (rc == SDFHIP_OK) rc = perRank(n, [&](int r) {
            SDF_TRY(sdfhip_octree_from_data(M->ctx[r], full[r].p, total, SDFHIP_DEVICE, info[0].box_min, info[0].box_max, info[0].start_grid_size, info[0].max_depth, valueRange, minBorder, &tree[r]));
            SDF_TRY(sdfhip_octree_set_start_grid_cell_size(tree[r], info[0].start_grid_cell_size));
            // statistics of the whole job on every replica
            sdfhip_octree_info& I = tree[r]->info;
            for (int s = 0; s < n; s++) {
                I.num_leaves += info[s].num_leaves; I.num_nodes += info[s].num_nodes; I.num_samples += info[s].num_samples; I.num_traversals += info[s].num_traversals;
                I.num_nearest_fallbacks += info[s].num_nearest_fallbacks; I.near_expansions += info[s].near_expansions; I.near_triangle_tests += info[s].near_triangle_tests;
                I.seconds_near_candidates = std::max(I.seconds_near_candidates, info[s].seconds_near_candidates); I.seconds_near_search = std::max(I.seconds_near_search, info[s].seconds_near_search);
                for (int d = 0; d < 16; d++) I.leaves_per_depth[d] += info[s].leaves_per_depth[d];
            }
            return SDFHIP_OK;
        });
        for (int r = 0; r < n; r++) { (void)hipSetDevice(M->devices[r]); full[r].release(); if (shard[r]) sdfhip_octree_destroy(shard[r]); }
        M->lastExchangeSeconds = nowSeconds() - t1;
    }
    if (rc != SDFHIP_OK) { for (sdfhip_octree* t : tree) if (t) sdfhip_octree_destroy(t); return cleanup(rc); }
    for (int r = 0; r < n; r++) { out_trees[r] = tree[r]; if (out_meshes) out_meshes[r] = mesh[r]; else sdfhip_mesh_destroy(mesh[r]); }
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_multi_exact_build(sdfhip_multi* M, const float* xyz, uint32_t nv, const uint32_t* idx, uint32_t nt, const float* bbox6, const float box_min[3], const float box_max[3],
                             uint32_t max_depth, uint32_t start_depth, uint32_t min_triangles_per_node, sdfhip_mesh** out_meshes, sdfhip_exact** out_trees) {
    SDF_API_BEGIN
    SDF_REQUIRE(M && xyz && idx && box_min && box_max && out_meshes && out_trees, "NULL argument (the trees read their mesh's TriangleData: out_meshes is required)");
    const int n = (int)M->ctx.size();
    M->bytesExchanged = 0; M->lastExchangeSeconds = 0; M->lastShardSeconds = 0;
    std::vector<sdfhip_mesh*> mesh;
    SDF_TRY(meshesEverywhere(M, xyz, nv, idx, nt, bbox6, mesh));
    std::vector<sdfhip_exact*> tree(n, nullptr);
    auto fail = [&](int rc) { for (sdfhip_exact* t : tree) if (t) sdfhip_exact_destroy(t); for (sdfhip_mesh* m : mesh) if (m) sdfhip_mesh_destroy(m); return rc; };
    const uint32_t numCells = 1u << (3 * start_depth);
    if (n == 1 || numCells < (uint32_t)n) {
        if (n > 1) { setError("%d devices cannot share %u start cells: use a larger start depth", n, numCells); return fail(SDFHIP_E_INVALID); }
        const double t0 = nowSeconds();
        int rc = sdfhip_exact_build(M->ctx[0], mesh[0], box_min, box_max, max_depth, start_depth, min_triangles_per_node, &tree[0]);
        if (rc != SDFHIP_OK) return fail(rc);
        M->lastShardSeconds = nowSeconds() - t0;
        out_trees[0] = tree[0]; out_meshes[0] = mesh[0];
        return SDFHIP_OK;
    }
    // cells in the reference's EMISSION order (children 7..0 at every level): concatenating the ranks' bodies / sets / masks then reproduces
    // the single-thread layout; the weights are permuted accordingly
    const uint32_t G = 1u << start_depth;
    std::vector<uint32_t> emissionRank(numCells), cellAtRank(numCells);
    for (uint32_t z = 0; z < G; z++) for (uint32_t y = 0; y < G; y++) for (uint32_t x = 0; x < G; x++) {
        uint32_t rk = 0;
        for (uint32_t level = 0; level < start_depth; level++) {
            const uint32_t sh = start_depth - 1 - level;
            const uint32_t c = ((x >> sh) & 1u) | (((y >> sh) & 1u) << 1) | (((z >> sh) & 1u) << 2);
            rk = rk * 8u + (7u - c);
        }
        emissionRank[(z * G + y) * G + x] = rk; cellAtRank[rk] = (z * G + y) * G + x;
    }
    const std::vector<double> wCell = cellWeights(xyz, nv, box_min, box_max, start_depth);
    std::vector<double> wRank(numCells);
    for (uint32_t k = 0; k < numCells; k++) wRank[k] = wCell[cellAtRank[k]];
    const auto ranges = partition(wRank, n);
    std::vector<sdfhip_exact*> shard(n, nullptr); std::vector<sdfhip_exact_info> info(n);
    const double t0 = nowSeconds();
    int rc = perRank(n, [&](int r) {
        SDF_TRY(sdfhip_exact_build_shard(M->ctx[r], mesh[r], box_min, box_max, max_depth, start_depth, min_triangles_per_node, ranges[r].first, ranges[r].second, &shard[r]));
        return sdfhip_exact_get_info(shard[r], &info[r]);
    });
    M->lastShardSeconds = nowSeconds() - t0;
    const double t1 = nowSeconds();
    uint64_t nNodes = numCells, nSets = 0, nMasks = 0;
    std::vector<uint64_t> offN(n), offS(n), offM(n);
    for (int r = 0; r < n && rc == SDFHIP_OK; r++) { offN[r] = nNodes; offS[r] = nSets; offM[r] = nMasks; nNodes += info[r].num_nodes; nSets += info[r].num_set_words; nMasks += info[r].num_mask_bytes; }
    std::vector<DevBuf<uint32_t>> nodes(n), sets(n); std::vector<DevBuf<uint8_t>> has(n), masks(n);
    std::vector<std::vector<uint32_t>> gridNodes(n), cells(n); std::vector<std::vector<uint8_t>> gridHas(n);
    if (rc == SDFHIP_OK) rc = perRank(n, [&](int r) {
        SDF_HIP_CHECK(hipSetDevice(M->devices[r]));
        SDF_TRY(nodes[r].reserve(2 * nNodes)); SDF_TRY(has[r].reserve(nNodes)); SDF_TRY(sets[r].reserve(nSets + 1)); SDF_TRY(masks[r].reserve(nMasks + 1));
        const uint32_t nc = ranges[r].second - ranges[r].first;
        cells[r].resize(nc); SDF_TRY(sdfhip_exact_shard_cells(shard[r], cells[r].data()));
        // the grid slots are few (2 words per cell): through the host; the bodies go straight to their absolute positions on the device
        DevBuf<uint32_t> dGrid; DevBuf<uint8_t> dGridHas;
        SDF_TRY(dGrid.reserve(2ull * nc)); SDF_TRY(dGridHas.reserve(nc));
        SDF_TRY(sdfhip_exact_emit_shard(shard[r], offN[r], offS[r], offM[r], dGrid.p, dGridHas.p, nodes[r].p + 2 * offN[r], has[r].p + offN[r], sets[r].p + offS[r], masks[r].p + offM[r], SDFHIP_DEVICE));
        gridNodes[r].resize(2ull * nc); gridHas[r].resize(nc);
        SDF_HIP_CHECK(hipMemcpy(gridNodes[r].data(), dGrid.p, 8ull * nc, hipMemcpyDeviceToHost));
        SDF_HIP_CHECK(hipMemcpy(gridHas[r].data(), dGridHas.p, nc, hipMemcpyDeviceToHost));
        return SDFHIP_OK;
    });
    if (rc == SDFHIP_OK) {
        std::vector<uint32_t> grid(2ull * numCells, 0u); std::vector<uint8_t> ghas(numCells, 0);
        for (int r = 0; r < n; r++) for (size_t k = 0; k < cells[r].size(); k++) { const uint32_t c = cells[r][k]; grid[2ull * c] = gridNodes[r][2 * k]; grid[2ull * c + 1] = gridNodes[r][2 * k + 1]; ghas[c] = gridHas[r][k]; }
        rc = perRank(n, [&](int r) {
            SDF_HIP_CHECK(hipSetDevice(M->devices[r]));
            SDF_HIP_CHECK(hipMemcpy(nodes[r].p, grid.data(), 8ull * numCells, hipMemcpyHostToDevice));
            SDF_HIP_CHECK(hipMemcpy(has[r].p, ghas.data(), numCells, hipMemcpyHostToDevice));
            return SDFHIP_OK;
        });
    }
    if (rc == SDFHIP_OK) {
        std::vector<void*> bN(n), bH(n), bS(n), bM(n);
        std::vector<Segment> sN, sH, sS, sM;
        for (int r = 0; r < n; r++) {
            bN[r] = nodes[r].p; bH[r] = has[r].p; bS[r] = sets[r].p; bM[r] = masks[r].p;
            sN.push_back(Segment{r, 2 * offN[r], 2 * info[r].num_nodes}); sH.push_back(Segment{r, offN[r], info[r].num_nodes});
            sS.push_back(Segment{r, offS[r], info[r].num_set_words}); sM.push_back(Segment{r, offM[r], info[r].num_mask_bytes});
        }
        rc = allGatherV(M, bN, sN, 4);
        if (rc == SDFHIP_OK) rc = allGatherV(M, bH, sH, 1);
        if (rc == SDFHIP_OK) rc = allGatherV(M, bS, sS, 4);
        if (rc == SDFHIP_OK) rc = allGatherV(M, bM, sM, 1);
    }
    if (rc == SDFHIP_OK) {
        sdfhip_exact_info full = info[0];
        full.num_nodes = nNodes; full.num_set_words = nSets; full.num_mask_bytes = nMasks; full.cull_tests = 0;
        for (int r = 0; r < n; r++) {
            full.max_triangles_in_leafs = std::max(full.max_triangles_in_leafs, info[r].max_triangles_in_leafs);
            full.max_triangles_encoded_in_leafs = std::max(full.max_triangles_encoded_in_leafs, info[r].max_triangles_encoded_in_leafs);
            full.cull_tests += info[r].cull_tests;
        }
        rc = perRank(n, [&](int r) { return sdfhip_exact_from_parts(M->ctx[r], mesh[r], &full, nodes[r].p, has[r].p, sets[r].p, masks[r].p, SDFHIP_DEVICE, &tree[r]); });
    }
    for (int r = 0; r < n; r++) { (void)hipSetDevice(M->devices[r]); nodes[r].release(); has[r].release(); sets[r].release(); masks[r].release(); if (shard[r]) sdfhip_exact_destroy(shard[r]); }
    M->lastExchangeSeconds = nowSeconds() - t1;
    if (rc != SDFHIP_OK) return fail(rc);
    for (int r = 0; r < n; r++) { out_trees[r] = tree[r]; out_meshes[r] = mesh[r]; }
    return SDFHIP_OK;
    SDF_API_END
}

}  // extern "C"

// (sdfhip_ctx_create: the runtime loads a translation unit's code object on the first use of one of its kernels — milliseconds that would
// otherwise land in the first build or the first query of a process)
namespace sdfhip { void loadKernelsMulti() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_add_u32)); (void)hipGetLastError(); } }
