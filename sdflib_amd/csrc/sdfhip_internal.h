// Internal host-side plumbing of libsdfhip (context, error handling, device buffers).  PRODUCT code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdarg.h>
#include <stdio.h>
#include <string>
#include <new>
#include <exception>
#include <vector>
#include <chrono>
#include <mutex>
#include "../../include/sdfhip.h"

namespace sdfhip {

void setError(const char* fmt, ...);

#define SDF_HIP_CHECK(expr)                                                                        \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            sdfhip::setError("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return SDFHIP_E_HIP;                                                                   \
        }                                                                                          \
    } while (0)

// Every int-returning entry point runs between these two: no C++ exception crosses the C boundary (include/sdfhip.h).
#define SDF_API_BEGIN try {
#define SDF_API_END                                                                                   \
    } catch (const std::bad_alloc&) { sdfhip::setError("out of host memory"); return SDFHIP_E_HOST; }  \
    catch (const std::exception& e) { sdfhip::setError("host-side failure: %s", e.what()); return SDFHIP_E_HOST; } \
    catch (...) { sdfhip::setError("host-side failure (unknown exception)"); return SDFHIP_E_HOST; }

#define SDF_REQUIRE(cond, msg)                                                  \
    do {                                                                        \
        if (!(cond)) { sdfhip::setError("invalid argument: %s", msg); return SDFHIP_E_INVALID; } \
    } while (0)

#define SDF_TRY(expr) do { int _r = (expr); if (_r != SDFHIP_OK) return _r; } while (0)

static inline double nowSeconds() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// allocation cost accounting (diagnostics under SDFHIP_TIMING)
inline double& g_allocSeconds() { static thread_local double v = 0; return v; }
inline long& g_allocCalls() { static thread_local long v = 0; return v; }

// Stream-ordered allocation.  A build creates and drops a few hundred device buffers (hipMalloc / hipFree cost ~40 us each
// and hipFree synchronises the device: 10 ms of a 60 ms build in the first measurements).  Inside an AllocScope the buffers of
// the calling thread come from the device's default memory pool on the given stream instead (hipMallocAsync / hipFreeAsync: no
// synchronisation, freed blocks are reused by later requests on the stream).  Outside a scope DevBuf falls back to hipMalloc /
// hipFree, which is also how long-lived buffers (trees, meshes) are released later.
struct AllocState { hipStream_t stream = nullptr; bool active = false; };
inline AllocState& tlsAlloc() { static thread_local AllocState s; return s; }
struct AllocScope {
    AllocState prev;
    explicit AllocScope(hipStream_t s) { prev = tlsAlloc(); tlsAlloc().stream = s; tlsAlloc().active = !getenv("SDFHIP_NO_POOL"); }
    ~AllocScope() { tlsAlloc() = prev; }
    AllocScope(const AllocScope&) = delete;
    AllocScope& operator=(const AllocScope&) = delete;
};

// LARGE transient blocks (>= kBigBlock) do not come from HIP's stream-ordered pool but from this cache of plain allocations, one free list
// per (device, stream): a block released by its owner goes back to the list of the stream it was used on and is handed to a later
// request on the SAME stream, so reuse is ordered by the stream exactly like hipMallocAsync / hipFreeAsync, without their cost or a
// device synchronisation.  Why not the HIP pool for these: with the system runtime of this image (C / C++ hosts; Python binds torch's
// older runtime) a build that requested a large block from the pool in mid-flight — 167 MB of culling scratch, 31 MB of candidate
// lists — intermittently produced wrong arrays or faulted in a LATER kernel, deterministically gone with plain allocations; small
// blocks (hundreds per build) stay with the pool.  Cached blocks are freed when their context is destroyed (trimStream).
struct BigBlockCache {
    struct Block { void* p; size_t bytes; bool fromPool; };
    struct Key { int device; hipStream_t stream; bool operator<(const Key& o) const { return device != o.device ? device < o.device : stream < o.stream; } };
    // per (device, stream): the large blocks in one list (few, matched by "fits within 2x"), the small ones in power-of-two size classes
    struct Lists { std::vector<Block> big; std::vector<Block> small[24]; };
    std::mutex m; std::vector<std::pair<Key, Lists>> lists;
    static BigBlockCache& get() { static BigBlockCache c; return c; }
    Lists& listOf(Key k) { for (auto& e : lists) if (!(e.first < k) && !(k < e.first)) return e.second; lists.emplace_back(k, Lists()); return lists.back().second; }
    hipError_t alloc(void** out, size_t bytes, hipStream_t st, size_t* got) {
        int dev = 0; (void)hipGetDevice(&dev);
        {
            std::lock_guard<std::mutex> g(m);
            std::vector<Block>& L = listOf(Key{dev, st}).big;
            size_t best = (size_t)-1;
            for (size_t i = 0; i < L.size(); i++) if (L[i].bytes >= bytes && L[i].bytes <= 2 * bytes && (best == (size_t)-1 || L[i].bytes < L[best].bytes)) best = i;
            if (best != (size_t)-1) { *out = L[best].p; *got = L[best].bytes; L[best] = L.back(); L.pop_back(); return hipSuccess; }
        }
        const size_t rounded = (bytes + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
        *got = rounded;
        return hipMalloc(out, rounded);
    }
    // SMALL transient blocks (< kBigBlock; a build asks for ~250 of them): the first request of a size class goes to HIP's stream-ordered
    // pool, a released block waits in its class for the next request on the same stream — a build after the first one makes no
    // allocation call at all (250 x ~14 us = 3.5 ms of a 26 ms build before).
    static int classOf(size_t bytes) { int c = 8; while (((size_t)1 << c) < bytes) c++; return c - 8; }       // 256 B .. 2 GB
    hipError_t allocSmall(void** out, size_t bytes, hipStream_t st, size_t* got) {
        int dev = 0; (void)hipGetDevice(&dev);
        const int c = classOf(bytes);
        *got = (size_t)1 << (c + 8);
        {
            std::lock_guard<std::mutex> g(m);
            std::vector<Block>& L = listOf(Key{dev, st}).small[c];
            if (!L.empty()) { *out = L.back().p; L.pop_back(); return hipSuccess; }
        }
        return hipMallocAsync(out, *got, st);
    }
    void release(void* p, size_t bytes, int dev, hipStream_t st, bool fromPool) {
        std::lock_guard<std::mutex> g(m);
        Lists& L = listOf(Key{dev, st});
        if (fromPool) L.small[classOf(bytes)].push_back(Block{p, bytes, true}); else L.big.push_back(Block{p, bytes, false});
    }
    void trimStream(int dev, hipStream_t st) {
        Lists drop;
        { std::lock_guard<std::mutex> g(m); std::swap(drop, listOf(Key{dev, st})); }
        for (const Block& b : drop.big) (void)hipFree(b.p);
        for (auto& L : drop.small) for (const Block& b : L) if (hipFreeAsync(b.p, st) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(b.p); }
    }
};
constexpr size_t kBigBlock = 4u << 20;

// diagnostic (SDFHIP_ALLOC_CHECK): registry of the live device blocks; a new block overlapping a live one is reported
struct AllocRegistry {
    std::mutex m; std::vector<std::pair<uintptr_t, size_t>> live;
    static AllocRegistry& get() { static AllocRegistry r; return r; }
    static bool on() { static const bool v = getenv("SDFHIP_ALLOC_CHECK") != nullptr; return v; }
    void add(void* p, size_t bytes) {
        std::lock_guard<std::mutex> g(m);
        const uintptr_t b = (uintptr_t)p, e = b + bytes;
        for (auto& r : live) if (b < r.first + r.second && r.first < e) fprintf(stderr, "[sdfhip] ALLOC OVERLAP: new [%p, +%zu) overlaps live [%p, +%zu)\n", p, bytes, (void*)r.first, r.second);
        live.emplace_back(b, bytes);
    }
    void remove(void* p) {
        std::lock_guard<std::mutex> g(m);
        for (size_t i = 0; i < live.size(); i++) if (live[i].first == (uintptr_t)p) { live[i] = live.back(); live.pop_back(); return; }
        fprintf(stderr, "[sdfhip] ALLOC CHECK: freeing unknown block %p\n", p);
    }
};

// Device allocation owned by one object; freed in the destructor.
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    bool pooled = false;
    hipStream_t poolStream = nullptr;      // the stream a pooled block was allocated on (and is given back on)
    size_t cachedBytes = 0; int cachedDevice = 0;      // > 0: a block of BigBlockCache (its real size)
    bool cachedSmall = false;                           // ... of its small size classes (memory of the HIP pool)
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n), pooled(o.pooled), poolStream(o.poolStream), cachedBytes(o.cachedBytes), cachedDevice(o.cachedDevice), cachedSmall(o.cachedSmall) { o.p = nullptr; o.n = 0; o.cachedBytes = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        release(); p = o.p; n = o.n; pooled = o.pooled; poolStream = o.poolStream; cachedBytes = o.cachedBytes; cachedDevice = o.cachedDevice; cachedSmall = o.cachedSmall; o.p = nullptr; o.n = 0; o.cachedBytes = 0;
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (!p) return;
        if (AllocRegistry::on()) AllocRegistry::get().remove(p);
        const double t0 = nowSeconds();
        // a block of the stream-ordered pool goes back through hipFreeAsync on the stream it came from, also when its owner (a tree, a mesh)
        // is destroyed long after the build; a cached big block returns to its stream's free list
        if (cachedBytes) { BigBlockCache::get().release(p, cachedBytes, cachedDevice, poolStream, cachedSmall); cachedBytes = 0; }
        else if (pooled) { if (hipFreeAsync(p, poolStream) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); } }
        else (void)hipFree(p);
        g_allocSeconds() += nowSeconds() - t0; g_allocCalls()++;
        p = nullptr; n = 0;
    }
    // grows only; contents are NOT preserved
    int reserve(size_t count) {
        if (count <= n && p) return SDFHIP_OK;
        release();
        if (count == 0) count = 1;
        const double t0 = nowSeconds();
        hipError_t e;
        if (tlsAlloc().active && count * sizeof(T) >= kBigBlock) {
            e = BigBlockCache::get().alloc((void**)&p, count * sizeof(T), tlsAlloc().stream, &cachedBytes);
            pooled = true; poolStream = tlsAlloc().stream; (void)hipGetDevice(&cachedDevice); cachedSmall = false;
            if (e != hipSuccess) cachedBytes = 0;
        }
        else if (tlsAlloc().active) {
            static const bool noSmallCache = getenv("SDFHIP_NO_SMALL_CACHE") != nullptr;
            if (noSmallCache) { e = hipMallocAsync((void**)&p, count * sizeof(T), tlsAlloc().stream); cachedBytes = 0; }
            else { e = BigBlockCache::get().allocSmall((void**)&p, count * sizeof(T), tlsAlloc().stream, &cachedBytes); cachedSmall = true; (void)hipGetDevice(&cachedDevice); if (e != hipSuccess) cachedBytes = 0; }
            pooled = true; poolStream = tlsAlloc().stream;
        }
        else { e = hipMalloc((void**)&p, count * sizeof(T)); pooled = false; }
        g_allocSeconds() += nowSeconds() - t0; g_allocCalls()++;
        if (e != hipSuccess) { p = nullptr; setError("device allocation of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e)); return SDFHIP_E_HIP; }
        n = count;
        if (AllocRegistry::on()) AllocRegistry::get().add(p, count * sizeof(T));
        // diagnostic: SDFHIP_POISON_ALLOC fills every new buffer with 0xCD so that code relying on fresh (zeroed) memory shows up in the tests
        // (a stream-ordered pool hands back blocks with their old contents)
        static const bool poison = getenv("SDFHIP_POISON_ALLOC") != nullptr;
        static const int fill = poison ? (int)strtol(getenv("SDFHIP_POISON_ALLOC"), nullptr, 0) : 0;       // SDFHIP_POISON_ALLOC=0xCD (or 0: zero fill)
        if (poison) { if (pooled) (void)hipMemsetAsync(p, fill, count * sizeof(T), tlsAlloc().stream); else { (void)hipMemset(p, fill, count * sizeof(T)); (void)hipDeviceSynchronize(); } }
        return SDFHIP_OK;
    }
};

static inline unsigned gridFor(uint64_t work, unsigned block) { return (unsigned)((work + block - 1) / block); }

// Point queries are answered in chunks of at most this many points (grid dimensions and the sort's element count are 32-bit;
// host arrays are staged chunk by chunk).  SDFHIP_QUERY_CHUNK overrides it (tests run the chunk loop on small inputs).
static inline uint64_t queryChunk(bool hostBuffers) {
    static const uint64_t forced = getenv("SDFHIP_QUERY_CHUNK") ? strtoull(getenv("SDFHIP_QUERY_CHUNK"), nullptr, 10) : 0;
    if (forced) return forced;
    return hostBuffers ? (1ull << 27) : (1ull << 30);
}

}  // namespace sdfhip

// Device-side staging of the host-pointer entry points (points in, distances / gradients / ids out), kept with the context
// (grow-only, up to kStageKeepBytes per buffer) so that small calls — the C++ classes' scalar getDistance among them — cost no
// hipMalloc / hipFree.  Guarded by a try-lock: a second host thread simply allocates privately.
struct sdfhip_stage {
    sdfhip::DevBuf<float> pts, dist, grad; sdfhip::DevBuf<uint32_t> ids; std::mutex lock;
    static constexpr size_t kStageKeepBytes = 256u << 20;
};

// Scratch of the two-phase nearest-triangle search (dev_bvh_fast.h), kept with the context: plain device allocations that grow and
// are reused by every build (they are the largest transient buffers of a build: 64 B per query).  Used under the context's buildLock.
struct sdfhip_near_scratch {
    sdfhip::DevBuf<uint32_t> cand, fbList, fbCount, longList; sdfhip::DevBuf<uint8_t> candCount; sdfhip::DevBuf<float> candLo;   // candLo: the candidates' lower bounds (k_near_candidates)
      // fbCount[0]: this batch's fallback list length, [1]: total since reset, [2..9]: work counters, [10]: long list length
    bool counterReady = false;
};

struct sdfhip_ctx {
    sdfhip_stage stage;
    sdfhip_near_scratch nearScratch;
    int device = 0;
    hipStream_t stream = nullptr;
    bool ownsStream = false;
    hipStream_t copyStream = nullptr;      // results flow back on this one while the next piece of a large host-pointer query goes up (created on first use)
    std::mutex copyStreamLock;
    hipDeviceProp_t prop;
    // Builds (mesh preparation, BVH, octrees) on one context run one at a time: they share the stream-ordered allocator's scope and,
    // with an exchange installed, must stay in collective order.  Queries take no part in this and run concurrently.
    std::recursive_mutex buildLock;
    sdfhip_exchange exchange{};           // world >= 1: the CONTINUITY build shares out its traversals (sdfhip.h)
};

struct sdfhip_mesh {
    sdfhip_ctx* ctx = nullptr;
    uint32_t numVertices = 0, numTriangles = 0;
    std::vector<float> hVerts;            // host copies (BVH planner input)
    std::vector<uint32_t> hIdx;
    sdfhip::DevBuf<float> dVerts;         // 3 floats per vertex
    sdfhip::DevBuf<uint32_t> dIdx;        // 3 per triangle
    sdfhip::DevBuf<float> dTri;           // 37 floats per triangle (TriangleData)
    sdfhip::DevBuf<float> dFrames;        // 20 floats per triangle: packed frame (dev_math.h loadFramePacked)
    // bounding-sphere BVH (fp64), see dev_bvh.h: 8 doubles + one int2 per inner node, 12 floats per triangle
    sdfhip::DevBuf<double> dBvhSph;
    sdfhip::DevBuf<int> dBvhKids;
    sdfhip::DevBuf<float> dBvhSph32;      // fp32 copy of the spheres (8 floats per inner node), see dev_bvh.h
    float bvhCoordScale = 0.f;            // max |coordinate| of the mesh: bounds the fp32 rounding of the sphere centres
    sdfhip::DevBuf<float> dTriVerts;
    sdfhip::DevBuf<uint32_t> dTriRank;    // leaf-order position of every triangle, see dev_bvh.h
    sdfhip::DevBuf<float> dBvhWide;       // 16 floats per inner node (used at even depths): 4-wide quantised nodes, see dev_bvh.h
    uint64_t numBvhNodes = 0;             // inner nodes (= numTriangles - 1)
    bool hasBvh = false;
    uint32_t unmatchedEdges = 0;          // edges owned by a single triangle (open / non-manifold mesh)
    uint32_t weldedEdges = 0;             // of those, half-edges re-paired by the seam welding
};

int sdfhip_mesh_ensure_bvh(sdfhip_mesh* mesh);
namespace sdfhip { int packFrames(hipStream_t st, const float* td, uint32_t numTriangles, float* frames); }
