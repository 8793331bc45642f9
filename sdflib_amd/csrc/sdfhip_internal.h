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
#include <thread>
#include "../../include/sdfhip.h"
#include "../../include/sdfhip_test.h"

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

// Transient device memory.  A build creates and drops a few hundred device buffers (hipMalloc / hipFree cost ~40 us each and hipFree
// synchronises the device: 10 ms of a 60 ms build in the first measurements).  Inside an AllocScope the buffers of the calling thread
// come from BlockCache (below) instead: blocks are handed out and taken back in the order of ONE stream, so a block released by its
// owner can go to the next request on that stream at once — the reuse semantics of hipMallocAsync / hipFreeAsync without a runtime call.
// Outside a scope DevBuf uses hipMalloc / hipFree, which is also how long-lived buffers (trees, meshes) are released later.
//
// Why not HIP's own stream-ordered pool (hipMallocAsync): it corrupts memory with this image's runtime.  Rounds 1-2 saw builds that took
// large blocks from it produce, once in 5-30 runs, a wrong array or a fault in a later kernel, and worked around it for large blocks
// only.  Round 3 reproduced it WITHOUT this library: tools/pool_repro/pool_repro.hip (100 lines of plain HIP: blocks of one stream
// filled by a kernel, checked by a kernel, freed with hipFreeAsync, the next iteration's blocks allocated with hipMallocAsync) finds
// hundreds of millions of wrong words from the second iteration on — most of them ZERO, in blocks that are still owned, also with the
// stream idle at every hipFreeAsync, with the release threshold at its default or at its maximum, for blocks of 64 KB as for 170 MB —
// and none with hipMalloc / hipFree in the same program.  The pool hands out address ranges whose earlier contents (or mapping) it
// still manipulates.  No entry point of the library calls hipMallocAsync / hipFreeAsync any more.
struct AllocState { hipStream_t stream = nullptr; bool active = false; };
inline AllocState& tlsAlloc() { static thread_local AllocState s; return s; }
struct AllocScope {
    AllocState prev;
    explicit AllocScope(hipStream_t s) { prev = tlsAlloc(); tlsAlloc().stream = s; tlsAlloc().active = !getenv("SDFHIP_NO_POOL"); }
    inline ~AllocScope();           // restores the previous state and applies the cache's high-water mark (below)
    AllocScope(const AllocScope&) = delete;
    AllocScope& operator=(const AllocScope&) = delete;
};

// Per (device, stream): LARGE blocks (>= kBigBlock) are plain allocations kept in one list and matched by "fits within 2x"; SMALL blocks
// (a build asks for ~250 of them) are carved out of 64 MB slabs in power-of-two size classes, a released block waits in its class — a
// context's first build costs a handful of hipMalloc calls, later ones none (250 runtime calls of ~14 us each before).
// How long cached memory lives: (i) a HIGH-WATER MARK — when an API call that allocated through the cache returns (~AllocScope) and the
// (device, stream) list holds more than its share of the DEVICE's mark (SDFHIP_CACHE_KEEP_MB, default 1/32 of that device's memory, minus what the
// device's other streams hold: keepFor), the largest blocks, then idle slabs, are freed
// until they do not: a process that built one huge tree does not sit on that build's peak scratch for ever; (ii) sdfhip_ctx_trim(ctx,
// keep_bytes) on request; (iii) when the last context of a (device, stream) is destroyed (trimStream) — blocks released after that are
// freed at once; (iv) when an allocation fails, everything cached on the device is freed and the request retried once (trimDevice).
struct BigBlockCache {
    struct Block { void* p; size_t bytes; };
    struct Slab { char* base; size_t bytes, used; long live; };       // live = blocks handed out and not yet released
    struct Key { int device; hipStream_t stream; bool operator<(const Key& o) const { return device != o.device ? device < o.device : stream < o.stream; } };
    struct Lists { std::vector<Block> big; std::vector<Block> small[24]; std::vector<Slab> slabs; };
    std::mutex m; std::vector<std::pair<Key, Lists>> lists;
    std::vector<std::pair<Key, int>> refs;            // live contexts per (device, stream)
    static constexpr size_t kSlabBytes = 64u << 20;
    static BigBlockCache& get() { static BigBlockCache c; return c; }
    // Default mark: 1/32 of the device's memory (9 GB of an MI355X's 288 GB).  The transient blocks of the largest builds the tests and
    // the bench run (ExactOctreeSdf depth 7: 5.7 GB of cull scratch; OctreeSdf depth 9: 1.6 GB) stay below it, so a rebuild asks the runtime
    // for nothing: multi-GB hipMalloc calls are usually lazy (1-3 ms for 5.7 GB) but were seen to take 0.2-0.3 s each on some boxes
    // (profiles/r05z_bench_n1.json's exact leg, the round-4 "one rebuild in four"); with 512 MB every large build paid them again.
    // The mark is per DEVICE (1/32 of that device's memory, or SDFHIP_CACHE_KEEP_MB), shared by all (device, stream) lists of the device:
    // what one stream's list may keep is the mark minus what the device's other streams hold (keepFor).
    static size_t deviceMark(int dev) {
        static std::mutex mm; static size_t mark[64]; static bool known[64];
        if (const char* e = getenv("SDFHIP_CACHE_KEEP_MB")) return (size_t)strtoull(e, nullptr, 10) << 20;
        std::lock_guard<std::mutex> g(mm);
        if (dev < 0 || dev >= 64) return (size_t)512 << 20;
        if (!known[dev]) {
            int cur = 0; (void)hipGetDevice(&cur);
            size_t freeB = 0, totalB = 0;
            const bool ok = hipSetDevice(dev) == hipSuccess && hipMemGetInfo(&freeB, &totalB) == hipSuccess && totalB != 0;
            (void)hipSetDevice(cur); (void)hipGetLastError();
            mark[dev] = ok ? totalB / 32 : (size_t)512 << 20; known[dev] = true;
        }
        return mark[dev];
    }
    static size_t keepBytes() { int dev = 0; (void)hipGetDevice(&dev); return deviceMark(dev); }      // the current device's mark
    size_t keepFor(int dev, hipStream_t st) {
        const size_t mark = deviceMark(dev);
        std::lock_guard<std::mutex> g(m);
        size_t others = 0;
        for (auto& e : lists) if (e.first.device == dev && e.first.stream != st) others += heldBytes(e.second);
        return others >= mark ? 0 : mark - others;
    }
    Lists& listOf(Key k) { for (auto& e : lists) if (!(e.first < k) && !(k < e.first)) return e.second; lists.emplace_back(k, Lists()); return lists.back().second; }
    void addRef(int dev, hipStream_t st) { std::lock_guard<std::mutex> g(m); for (auto& r : refs) if (r.first.device == dev && r.first.stream == st) { r.second++; return; } refs.emplace_back(Key{dev, st}, 1); }
    int dropRef(int dev, hipStream_t st) {           // returns the contexts left on that stream
        std::lock_guard<std::mutex> g(m);
        for (size_t i = 0; i < refs.size(); i++) if (refs[i].first.device == dev && refs[i].first.stream == st) { const int left = --refs[i].second; if (left <= 0) { refs[i] = refs.back(); refs.pop_back(); } return left; }
        return 0;
    }
    static size_t heldBytes(const Lists& L) { size_t t = 0; for (const Block& b : L.big) t += b.bytes; for (const Slab& s : L.slabs) t += s.bytes; return t; }
    size_t cachedBytes(int dev, hipStream_t st) {     // device memory the cache holds for (device, stream) that no live buffer uses
        std::lock_guard<std::mutex> g(m);
        for (auto& e : lists) if (e.first.device == dev && e.first.stream == st) return heldBytes(e.second) - liveSmallBytes(e.second);
        return 0;
    }
    static size_t liveSmallBytes(const Lists& L) {    // bytes of slab memory currently handed out = carved - waiting in the classes
        size_t carved = 0, waiting = 0;
        for (const Slab& s : L.slabs) carved += s.used;
        for (auto& S : L.small) for (const Block& b : S) waiting += b.bytes;
        return carved - waiting;
    }
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
        hipError_t e = hipMalloc(out, rounded);
        if (e != hipSuccess) { trimDevice(dev); e = hipMalloc(out, rounded); }       // stale cached blocks of other sizes must not make a build that fits fail
        return e;
    }
    static int classOf(size_t bytes) { int c = 8; while (((size_t)1 << c) < bytes) c++; return c - 8; }       // 256 B .. 2 GB
    hipError_t allocSmall(void** out, size_t bytes, hipStream_t st, size_t* got) {
        int dev = 0; (void)hipGetDevice(&dev);
        const int c = classOf(bytes);
        const size_t sz = (size_t)1 << (c + 8);
        *got = sz;
        for (int attempt = 0; attempt < 2; attempt++) {
            {
                std::lock_guard<std::mutex> g(m);
                Lists& L = listOf(Key{dev, st});
                if (!L.small[c].empty()) { *out = L.small[c].back().p; L.small[c].pop_back(); slabOf(L, *out)->live++; return hipSuccess; }
                for (size_t i = L.slabs.size(); i-- > 0;) { Slab& s = L.slabs[i]; if (s.used + sz <= s.bytes) { *out = s.base + s.used; s.used += sz; s.live++; return hipSuccess; } }
            }
            void* base = nullptr;
            const size_t slabBytes = sz > kSlabBytes ? sz : kSlabBytes;
            hipError_t e = hipMalloc(&base, slabBytes);
            if (e != hipSuccess) { trimDevice(dev); e = hipMalloc(&base, slabBytes); }
            if (e != hipSuccess) return e;
            std::lock_guard<std::mutex> g(m);
            listOf(Key{dev, st}).slabs.push_back(Slab{(char*)base, slabBytes, 0, 0});
        }
        return hipErrorOutOfMemory;
    }
    static Slab* slabOf(Lists& L, void* p) { for (Slab& s : L.slabs) if ((char*)p >= s.base && (char*)p < s.base + s.bytes) return &s; return nullptr; }
    void release(void* p, size_t bytes, int dev, hipStream_t st, bool small) {
        {
            std::lock_guard<std::mutex> g(m);
            bool live = false;
            for (auto& r : refs) if (r.first.device == dev && r.first.stream == st) { live = true; break; }
            Lists& L = listOf(Key{dev, st});
            if (small) {             // a slab block always goes back to its slab's class list: the slab is freed as a whole once it is idle
                Slab* s = slabOf(L, p);
                if (s) { s->live--; L.small[classOf(bytes)].push_back(Block{p, bytes}); if (live) return; }
            } else if (live) { L.big.push_back(Block{p, bytes}); return; }
        }
        // the stream's last context is gone (an object outlived it): nothing would ever reuse the block
        if (small) freeIdleSlabs(dev, st); else (void)hipFree(p);
    }
    // removes slabs without live blocks (and their waiting blocks) from L; returns them for hipFree outside the lock
    static void takeIdleSlabs(Lists& L, std::vector<void*>& drop, size_t* total, size_t keep) {
        for (size_t i = 0; i < L.slabs.size();) {
            if (L.slabs[i].live > 0 || (total && *total <= keep)) { i++; continue; }
            const Slab s = L.slabs[i];
            for (auto& S : L.small) { size_t w = 0; for (size_t k = 0; k < S.size(); k++) if (!((char*)S[k].p >= s.base && (char*)S[k].p < s.base + s.bytes)) S[w++] = S[k]; S.resize(w); }
            drop.push_back(s.base); if (total) *total -= s.bytes;
            L.slabs[i] = L.slabs.back(); L.slabs.pop_back();
        }
    }
    void freeIdleSlabs(int dev, hipStream_t st) {
        std::vector<void*> drop;
        { std::lock_guard<std::mutex> g(m); takeIdleSlabs(listOf(Key{dev, st}), drop, nullptr, 0); }
        for (void* p : drop) (void)hipFree(p);
    }
    // Frees cached memory of (device, stream) — the largest big blocks first, then idle slabs — until at most `keep` bytes stay.  hipFree
    // waits for the device, so a block still referenced by queued work of that stream is safe to drop here.
    void trimTo(int dev, hipStream_t st, size_t keep) {
        std::vector<void*> drop;
        {
            std::lock_guard<std::mutex> g(m);
            Lists& L = listOf(Key{dev, st});
            size_t total = heldBytes(L);
            while (total > keep && !L.big.empty()) {
                size_t bi = 0;
                for (size_t i = 1; i < L.big.size(); i++) if (L.big[i].bytes > L.big[bi].bytes) bi = i;
                drop.push_back(L.big[bi].p); total -= L.big[bi].bytes; L.big[bi] = L.big.back(); L.big.pop_back();
            }
            takeIdleSlabs(L, drop, &total, keep);
        }
        for (void* p : drop) (void)hipFree(p);
    }
    void trimStream(int dev, hipStream_t st) { trimTo(dev, st, 0); }
    void trimDevice(int dev) {                       // everything idle on the device, all streams (an allocation failed)
        std::vector<void*> drop;
        {
            std::lock_guard<std::mutex> g(m);
            for (auto& e : lists) if (e.first.device == dev) { for (const Block& b : e.second.big) drop.push_back(b.p); e.second.big.clear(); takeIdleSlabs(e.second, drop, nullptr, 0); }
        }
        (void)hipGetLastError();
        for (void* p : drop) (void)hipFree(p);
    }
};
constexpr size_t kBigBlock = 4u << 20;
inline AllocScope::~AllocScope() {
    const AllocState mine = tlsAlloc();
    tlsAlloc() = prev;
    if (mine.active && !prev.active) {               // the outermost scope of an API call
        int dev = 0;
        size_t keep = 0;
        if (hipGetDevice(&dev) == hipSuccess && BigBlockCache::get().cachedBytes(dev, mine.stream) > (keep = BigBlockCache::get().keepFor(dev, mine.stream))) {
            const double t0 = nowSeconds(); const size_t before = BigBlockCache::get().cachedBytes(dev, mine.stream);
            BigBlockCache::get().trimTo(dev, mine.stream, keep);
            if (getenv("SDFHIP_TIMING")) fprintf(stderr, "[sdfhip] block cache: %zu MB above the mark given back to the device in %.1f ms\n", (before - BigBlockCache::get().cachedBytes(dev, mine.stream)) >> 20, 1e3 * (nowSeconds() - t0));
        }
    }
}

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
        // a cached block returns to the free list of the stream it was handed out on, also when its owner (a tree, a mesh) is destroyed
        // long after the build
        if (cachedBytes) { BigBlockCache::get().release(p, cachedBytes, cachedDevice, poolStream, cachedSmall); cachedBytes = 0; }
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
            e = BigBlockCache::get().allocSmall((void**)&p, count * sizeof(T), tlsAlloc().stream, &cachedBytes); cachedSmall = true; (void)hipGetDevice(&cachedDevice);
            if (e != hipSuccess) cachedBytes = 0;
            pooled = true; poolStream = tlsAlloc().stream;
        }
        else {
            e = hipMalloc((void**)&p, count * sizeof(T)); pooled = false;
            if (e != hipSuccess) { int dev = 0; (void)hipGetDevice(&dev); BigBlockCache::get().trimDevice(dev); e = hipMalloc((void**)&p, count * sizeof(T)); }
        }
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

// A few words from the device for the host, behind whatever is queued on `st`: a one-thread kernel stores them and then a sequence
// number into host-mapped pinned memory (one mailbox per host thread and device), the host spins on the sequence number.  Measured on the
// box (tools/sync_probe): 10 us behind a small kernel against 23 us for hipMemcpyAsync into a pageable word + hipStreamSynchronize — and
// the builders read a count back several times per level (ctx_mesh.hip).  count <= 8; b (optional): out[i] = a[i] + b[i].
int readBackWords(hipStream_t st, const uint32_t* a, const uint32_t* b, int count, uint32_t* out);

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
    sdfhip::DevBuf<uint32_t> cand, fbList, fbCount, longList, best; sdfhip::DevBuf<uint8_t> candCount; sdfhip::DevBuf<float> candLo, candU2;   // candU2: the final upper bound of every query's squared distance; candLo: the candidates' lower bounds (k_near_candidates)
      // fbCount[0]: this batch's fallback list length, [1]: total since reset, [2..9]: work counters (leaders), [10]: long list length, [12..19]: work counters (followers); best: the leaders' triangles
      // [20..21], [22..23]: u64 totals of wide-node expansions and triangle tests since the counters were reset
    bool counterReady = false;
    // device time of the search's launches within a build: event triples (before the candidate kernel, after it, after the fallback) per batch
    static constexpr int kMaxTimed = 24;
    hipEvent_t ev[3 * kMaxTimed] = {}; int evMade = 0, evUsed = 0;
    ~sdfhip_near_scratch() { for (int i = 0; i < evMade; i++) (void)hipEventDestroy(ev[i]); }
    size_t bytes() const { return 4 * (cand.n + fbList.n + fbCount.n + longList.n + candLo.n + best.n + candU2.n) + candCount.n; }
    void release() { cand.release(); fbList.release(); fbCount.release(); longList.release(); best.release(); candCount.release(); candLo.release(); candU2.release(); counterReady = false; }
};

struct sdfhip_ctx {
    sdfhip_stage stage;
    sdfhip_near_scratch nearScratch;
    int device = 0;
    hipStream_t stream = nullptr;
    bool ownsStream = false;
    size_t contCaps[3] = {0, 0, 0};                   // what the last CONTINUITY build of this context grew its node array / node pool / post-pass list to: the next one starts there (builds are serialised by buildLock)
    std::mutex sideLock;                              // guards the lazy creation of the three streams below (queries of several host threads may run concurrently)
    hipStream_t upSide[2] = {nullptr, nullptr};       // ... and further uploader threads' streams
    hipStream_t downStream = nullptr;                 // host-pointer query batches: results go back on this stream while the next piece is uploaded and evaluated on `stream` (octree_query.hip)
    hipStream_t bvhSide[2] = {nullptr, nullptr};      // the BVH build's centre sums run on these behind each level's sort (created on first use; builds are serialised by buildLock)
    hipDeviceProp_t prop;
    // Builds (mesh preparation, BVH, octrees) on one context run one at a time: they share the stream-ordered allocator's scope and,
    // with an exchange installed, must stay in collective order.  Queries take no part in this and run concurrently.
    std::recursive_mutex buildLock;
    sdfhip_exchange exchange{};           // world >= 1: the CONTINUITY build shares out its traversals (sdfhip.h)
};

// Declared by a builder right after it takes the context's buildLock: when the build returns, the nearest search's candidate lists
// (128 B per sample of the largest batch: gigabytes after a depth-9 build) are released if they exceed the caches' high-water mark.
struct NearScratchMark {
    sdfhip_ctx* ctx;
    explicit NearScratchMark(sdfhip_ctx* c) : ctx(c) {}
    ~NearScratchMark() {
        if (ctx->nearScratch.bytes() > sdfhip::BigBlockCache::keepBytes()) { (void)hipStreamSynchronize(ctx->stream); ctx->nearScratch.release(); }
    }
};

// A BVH plan started by sdfhip_mesh_create_opt (SDFHIP_MESH_PLAN_BVH_EARLY) on a host thread of its own, so that the planner runs under
// the mesh preparation; sdfhip_mesh_build_bvh takes it over (bvh.hip).  Joined and dropped with the mesh if nobody did.
struct EarlyBvhPlan {
    std::thread th; void* plan = nullptr; void (*drop)(void*) = nullptr;
    ~EarlyBvhPlan() { if (th.joinable()) th.join(); if (plan && drop) drop(plan); }
};

struct sdfhip_mesh {
    sdfhip_ctx* ctx = nullptr;
    uint32_t numVertices = 0, numTriangles = 0;
    std::vector<float> hVerts;            // host copies (BVH planner input)
    std::vector<uint32_t> hIdx;
    EarlyBvhPlan early;
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
namespace sdfhip { void startEarlyBvhPlan(sdfhip_mesh* mesh); }      // bvh.hip: plans the BVH on a thread of its own (mesh->early)


int sdfhip_mesh_ensure_bvh(sdfhip_mesh* mesh);
namespace sdfhip { bool bvhBuildOnDevice(); }      // bvh.hip: the tree is built by the device builder (default) rather than planned on the host
namespace sdfhip { int packFrames(hipStream_t st, const float* td, uint32_t numTriangles, float* frames); }
