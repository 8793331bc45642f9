// Bounding-sphere BVH: host planner (fp64, identical tree to the reference's) + device nearest-triangle kernels.
// PRODUCT code — independent of oracle/.
//
// Host planner reproduces tmd::TriangleMeshDistance::_build_tree (reference libs/InteractiveComputerGraphics/
// InteractiveComputerGraphics/TriangleMeshDistance.h:421-490): median split of the triangle range after a
// std::sort by the FIRST vertex's coordinate on the widest AABB axis; node centre = mean of the range's vertices
// accumulated in range order; radius = max distance to them; leaves hold one triangle.  The reference's array
// is filled in DFS pre-order, so a subtree over n triangles occupies exactly 2n-1 consecutive slots: node ids are
// known up front and the two halves of a range can be planned by different host threads without changing a bit
// of the result.  The sort works on {key, triangle} pairs instead of the reference's 80-byte structs: std::sort's
// permutation depends only on comparison outcomes, which are the same.
//
// The keys are full of ties (triangles sharing their first vertex), std::sort is not stable, and the permutation it leaves
// decides which triangles fall on which side of every median — so the planner has to end with libstdc++'s permutation,
// not just with a sorted range.  IntroSortLike below restates that algorithm (introsort: median-of-3 to the front,
// unguarded Hoare partition, recurse right / loop left, 16-element threshold, 2*floor(log2 n) depth limit with heap sort
// fallback, final insertion sort) so that it can run on several threads: the independent sub-ranges a partition leaves behind are
// tasks of the planner's pool, and the partition of a large range is itself computed in parallel (parallelPartition) — same
// comparisons on the same data, same permutation.  What bounds the planner after that is its total work, ~0.8 us of CPU time per
// triangle (20 levels of gather, sum, AABB, radius and an n log n sort per node): on a host that grants the process 16 CPUs, 0.07 s
// for 1.31 M triangles however many threads are used.
// sdfhip_test_sort_matches_std() (tests/test_abi.py) compares it with std::sort on tie-heavy inputs.
#include "sdfhip_internal.h"
#include "dev_bvh_fast.h"
#include <algorithm>
#include <limits>
#include <thread>
#include <cmath>
#include <cstring>
#include <atomic>
#include <mutex>
#include <memory>
#include <condition_variable>
#include <deque>
#include <functional>
#include <pthread.h>
#include <chrono>
#include <sys/mman.h>
#include <stdlib.h>

namespace sdfhip {

// ---- helper threads of the planner ---------------------------------------------------------------------------------------
// The planner's parallelism is nested (a partition inside a sort inside a node inside a subtree) and fine grained (phases of 0.1 - 3
// ms).  Starting a std::thread for every piece was measured to stop scaling: thread creation takes the process's memory-map lock,
// and with hundreds of creations in flight the time of a node no longer shrank with its size (12 ms at every one of the top
// levels of a 1.3 M-triangle tree).  All of it therefore runs as tasks of ONE process-wide pool (created on first use, never torn
// down; an idle worker looks for work for a few microseconds — the next phase is usually that close — and then sleeps on a
// condition variable).  A thread that waits for a group of tasks executes queued tasks meanwhile, so waits nested to any depth
// cannot starve each other however few workers there are.  Three classes of tasks — SHORT (a chunk of a data-parallel phase), MEDIUM
// (a sub-range of a sort, a centre sum), LONG (a subtree) — and a waiter only helps with classes up to that of what it waits for:
// a phase that picked up somebody's subtree would stall everything queued behind that phase for the length of the subtree.
// (a fork()ed child inherits the pool object but none of its threads: it runs everything inline)
static std::atomic<bool> g_plannerPoolForked{false};
class PlannerPool {
public:
    struct Group { std::atomic<int> left{0}; };
    enum Class { SHORT = 0, MEDIUM = 1, LONG = 2 };
private:
    struct Task { std::function<void()> fn; Group* group; };
    std::mutex m; std::condition_variable cv; std::deque<Task> q[3];
    std::atomic<int> pending[3]; std::atomic<int> sleepers{0};
    int workers = 0;
    bool tryRun(int upTo) {
        bool any = false;
        for (int c = 0; c <= upTo; c++) any = any || pending[c].load(std::memory_order_acquire) > 0;
        if (!any) return false;
        Task t; bool got = false;
        {
            std::lock_guard<std::mutex> g(m);
            for (int c = 0; c <= upTo && !got; c++)
                if (!q[c].empty()) {
                    if (c == SHORT) { t = std::move(q[c].front()); q[c].pop_front(); } else { t = std::move(q[c].back()); q[c].pop_back(); }
                    pending[c].fetch_sub(1, std::memory_order_acq_rel); got = true;
                }
        }
        if (!got) return false;
        t.fn();
        if (t.group->left.fetch_sub(1, std::memory_order_acq_rel) == 1 && sleepers.load(std::memory_order_acquire) > 0) {
            std::lock_guard<std::mutex> g(m);           // a waiter of this group may be asleep (wait() checks its predicate under m)
            cv.notify_all();
        }
        return true;
    }
    void worker() {
        for (;;) {
            bool ran = false;
            for (int spin = 0; spin < 2000 && !ran; spin++) { ran = tryRun(LONG); if (!ran) __builtin_ia32_pause(); }
            if (ran) continue;
            std::unique_lock<std::mutex> g(m);
            sleepers.fetch_add(1);
            cv.wait(g, [&] { return !q[0].empty() || !q[1].empty() || !q[2].empty(); });
            sleepers.fetch_sub(1);
        }
    }
    PlannerPool() {
        for (int c = 0; c < 3; c++) pending[c].store(0);
        unsigned hc = std::thread::hardware_concurrency();
        workers = (int)(hc ? hc : 1u) - 1; if (workers > 127) workers = 127; if (workers < 0) workers = 0;
        // a container's CPU quota (cgroup v2 cpu.max = "<quota> <period>"): more runnable threads than about twice the quota only get the
        // group throttled for the rest of the period (measured on a 256-thread host with a quota of 16: 31 workers 0.07 s, 127 workers 0.07-0.15 s)
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            long long quota = 0, period = 0;
            if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) { const int cap = (int)(2 * quota / period) - 1; if (cap >= 1 && cap < workers) workers = cap; }
            fclose(f);
        }
        if (getenv("SDFHIP_BVH_POOL_THREADS")) workers = std::max(0, atoi(getenv("SDFHIP_BVH_POOL_THREADS")));
        for (int i = 0; i < workers; i++) std::thread([this] { worker(); }).detach();
        pthread_atfork(nullptr, nullptr, [] { g_plannerPoolForked.store(true); });
    }
public:
    static PlannerPool& get() { static PlannerPool* p = new PlannerPool(); return *p; }
    void spawn(Group& grp, std::function<void()> fn, Class c) {
        if (workers == 0 || g_plannerPoolForked.load(std::memory_order_relaxed)) { fn(); return; }
        grp.left.fetch_add(1, std::memory_order_acq_rel);
        { std::lock_guard<std::mutex> g(m); q[c].push_back(Task{std::move(fn), &grp}); pending[c].fetch_add(1, std::memory_order_acq_rel); }
        if (sleepers.load(std::memory_order_acquire) > 0) cv.notify_one();
    }
    // returns when the group's tasks are done; meanwhile runs queued tasks of classes <= help (never less than the group's own class)
    void wait(Group& grp, Class help) {
        int idle = 0;
        while (grp.left.load(std::memory_order_acquire) > 0) {
            if (tryRun(help)) { idle = 0; continue; }
            __builtin_ia32_pause();
            if (++idle < 2000) continue;
            // nothing to help with: sleep until the group is done or work appears (a container's CPU quota is shared with the workers —
            // a subtree's parent spinning for the length of the subtree was measured to get the whole process throttled)
            std::unique_lock<std::mutex> g(m);
            sleepers.fetch_add(1);
            cv.wait(g, [&] {
                if (grp.left.load(std::memory_order_acquire) <= 0) return true;
                for (int c = 0; c <= (int)help; c++) if (!q[c].empty()) return true;
                return false;
            });
            sleepers.fetch_sub(1);
            idle = 0;
        }
    }
    // fn(0 .. parts-1), the caller included
    void run(int parts, const std::function<void(int)>& fn) {
        if (parts <= 1 || workers == 0 || g_plannerPoolForked.load(std::memory_order_relaxed)) { for (int i = 0; i < parts; i++) fn(i); return; }
        Group grp;
        for (int i = 1; i < parts; i++) spawn(grp, [&fn, i]() { fn(i); }, SHORT);
        fn(0);
        wait(grp, SHORT);
    }
};

// ---- libstdc++'s std::sort, restated so that it can run on several threads --------------------------------------------
struct KeyTri { float key; int tri; };           // the key is a vertex coordinate, a float: comparing it as such = comparing the reference's doubles
static inline bool keyLess(const KeyTri& a, const KeyTri& b) { return a.key < b.key; }


struct IntroSortLike {
    int maxThreads = 1;
    size_t minParallel = 1u << 13;             // ranges below this are finished by the calling thread
    std::mutex cutMutex; std::vector<size_t> cuts; KeyTri* base = nullptr;

    static void moveMedianToFirst(KeyTri* result, KeyTri* a, KeyTri* b, KeyTri* c) {
        if (keyLess(*a, *b)) {
            if (keyLess(*b, *c)) std::iter_swap(result, b);
            else if (keyLess(*a, *c)) std::iter_swap(result, c);
            else std::iter_swap(result, a);
        } else if (keyLess(*a, *c)) std::iter_swap(result, a);
        else if (keyLess(*b, *c)) std::iter_swap(result, c);
        else std::iter_swap(result, b);
    }
    static KeyTri* unguardedPartition(KeyTri* first, KeyTri* last, KeyTri* pivot) {
        for (;;) {
            while (keyLess(*first, *pivot)) ++first;
            --last;
            while (keyLess(*pivot, *last)) --last;
            if (!(first < last)) return first;
            std::iter_swap(first, last);
            ++first;
        }
    }
    // unguardedPartition(first, last, pivot) on several threads, same final arrangement and return value.  The sequential scans only
    // ever look at elements no swap has touched yet, so what they do is fixed by the ORIGINAL content: the t-th swap exchanges the
    // t-th element from the left that is not less than the pivot (L_t) with the t-th from the right that is not greater (R_t), for
    // as long as L_t < R_t (m swaps), and the scan that ends the loop stops at L_(m+1) or at R_m — now holding a not-less element —
    // whichever comes first.  So: count both kinds per chunk, lay out the two index lists by prefix sums, find m by bisection,
    // swap the m pairs in parallel.
    size_t minParPartition = 1u << 17;         // ranges from this size on are partitioned by several threads
    // index scratch of the whole sort (2 x one uint32 per element of [base, base + n)); sub-ranges use their own slices.  No allocation
    // here: a multi-megabyte new / delete is an mmap / munmap, and those serialise every page fault of the process behind them.
    uint32_t* scratchL = nullptr; uint32_t* scratchR = nullptr;
    KeyTri* parallelPartition(KeyTri* first, KeyTri* last, KeyTri* pivot) {
        const size_t n = (size_t)(last - first);
        int parts = (int)(n / 16384); if (parts > 16) parts = 16; if (parts < 2) parts = 2;
        const float pk = pivot->key;
        std::vector<size_t> cl((size_t)parts + 1, 0), cr((size_t)parts + 1, 0);
        auto lo = [&](int c) { return n * (size_t)c / (size_t)parts; };
        PlannerPool& pool = PlannerPool::get();
        pool.run(parts, [&](int c) {
            size_t a = 0, b = 0;
            for (size_t i = lo(c), e = lo(c + 1); i < e; i++) { const float k = first[i].key; a += !(k < pk); b += !(pk < k); }
            cl[(size_t)c + 1] = a; cr[(size_t)c + 1] = b;
        });
        for (int c = 0; c < parts; c++) { cl[(size_t)c + 1] += cl[(size_t)c]; cr[(size_t)c + 1] += cr[(size_t)c]; }
        const size_t nl = cl[(size_t)parts], nr = cr[(size_t)parts];
        // L ascending; R stored ascending too (R_t = Rasc[nr - t])
        uint32_t* L = scratchL + (first - base); uint32_t* R = scratchR + (first - base);
        pool.run(parts, [&](int c) {
            size_t a = cl[(size_t)c], b = cr[(size_t)c];
            for (size_t i = lo(c), e = lo(c + 1); i < e; i++) { const float k = first[i].key; if (!(k < pk)) L[a++] = (uint32_t)i; if (!(pk < k)) R[b++] = (uint32_t)i; }
        });
        // m = number of t in [1, min(nl, nr)] with L_t < R_t  (L_t increases, R_t decreases with t)
        size_t lo_t = 0, hi_t = nl < nr ? nl : nr;
        while (lo_t < hi_t) { const size_t t = (lo_t + hi_t + 1) >> 1; if (L[t - 1] < R[nr - t]) lo_t = t; else hi_t = t - 1; }
        const size_t m = lo_t;
        if (m > 0) {
            int sp = (int)(m / 8192); if (sp > 16) sp = 16; if (sp < 1) sp = 1;
            pool.run(sp, [&](int c) {
                for (size_t t = m * (size_t)c / (size_t)sp + 1, e = m * (size_t)(c + 1) / (size_t)sp; t <= e; t++) std::iter_swap(first + L[t - 1], first + R[nr - t]);
            });
        }
        // the scan that ends the loop
        size_t stop = (m < nl) ? (size_t)L[m] : n;             // L_(m+1) in the original content (n: none; the median-of-3 sentinel rules that out when m == 0)
        if (m > 0 && (size_t)R[nr - m] < stop) stop = R[nr - m];
        return first + stop;
    }
    // The same construction on one thread: two passes without a data-dependent branch (the scans of the textbook loop mispredict every
    // other element on unsorted keys: 8-10 cycles per element against ~4 here), same arrangement, same return value.
    size_t minListPartition = 96;
    KeyTri* listPartition(KeyTri* first, KeyTri* last, KeyTri* pivot) {
        const size_t n = (size_t)(last - first);
        const float pk = pivot->key;
        uint32_t* L = scratchL + (first - base); uint32_t* R = scratchR + (first - base);
        size_t nl = 0, nr = 0;
        for (size_t i = 0; i < n; i++) { const float k = first[i].key; L[nl] = (uint32_t)i; nl += !(k < pk); R[nr] = (uint32_t)i; nr += !(pk < k); }
        size_t lo_t = 0, hi_t = nl < nr ? nl : nr;
        while (lo_t < hi_t) { const size_t t = (lo_t + hi_t + 1) >> 1; if (L[t - 1] < R[nr - t]) lo_t = t; else hi_t = t - 1; }
        const size_t m = lo_t;
        for (size_t t = 1; t <= m; t++) std::iter_swap(first + L[t - 1], first + R[nr - t]);
        size_t stop = (m < nl) ? (size_t)L[m] : n;
        if (m > 0 && (size_t)R[nr - m] < stop) stop = R[nr - m];
        return first + stop;
    }
    void loop(KeyTri* first, KeyTri* last, int depthLimit) {
        PlannerPool& pool = PlannerPool::get();
        PlannerPool::Group helpers;
        while (last - first > 16) {
            if (depthLimit == 0) { std::make_heap(first, last, keyLess); std::sort_heap(first, last, keyLess); break; }
            --depthLimit;
            KeyTri* mid = first + (last - first) / 2;
            moveMedianToFirst(first, first + 1, mid, last - 1);
            const size_t len = (size_t)(last - first);
            KeyTri* cut = (scratchL && len >= minParPartition && maxThreads > 1) ? parallelPartition(first + 1, last, first)
                        : (scratchL && len >= minListPartition) ? listPartition(first + 1, last, first) : unguardedPartition(first + 1, last, first);
            if (maxThreads > 1 && (size_t)(last - cut) >= minParallel && (size_t)(cut - first) >= minParallel) {
                { std::lock_guard<std::mutex> g(cutMutex); cuts.push_back((size_t)(cut - base)); }
                KeyTri* l = last; const int dl = depthLimit;
                pool.spawn(helpers, [this, cut, l, dl]() { loop(cut, l, dl); }, PlannerPool::MEDIUM);
            } else loop(cut, last, depthLimit);
            last = cut;
        }
        pool.wait(helpers, PlannerPool::MEDIUM);
    }
    static void insertionSort(KeyTri* first, KeyTri* last) {       // guarded form; same result as the reference's guarded + unguarded pair
        if (first == last) return;
        for (KeyTri* i = first + 1; i != last; ++i) {
            KeyTri val = *i;
            if (keyLess(val, *first)) { std::move_backward(first, i, i + 1); *first = val; }
            else { KeyTri* pos = i; KeyTri* next = i - 1; while (keyLess(val, *next)) { *pos = *next; pos = next; --next; } *pos = val; }
        }
    }
    void sort(KeyTri* first, KeyTri* last) {
        if (first == last) return;
        base = first; cuts.clear();
        int lg = 0; for (size_t n = (size_t)(last - first); n > 1; n >>= 1) lg++;
        const bool trace = (last - first) > 1000000 && getenv("SDFHIP_TIMING"); const double ts0 = nowSeconds();
        loop(first, last, 2 * lg);
        if (trace) fprintf(stderr, "[sdfhip] root sort: partition phase %.4f s, %zu cuts\n", nowSeconds() - ts0, cuts.size());
        // final insertion sort: elements never cross a partition cut, so the ranges between recorded cuts are independent
        std::sort(cuts.begin(), cuts.end());
        PlannerPool& pool = PlannerPool::get();
        PlannerPool::Group helpers;
        size_t begin = 0;
        for (size_t k = 0; k <= cuts.size(); k++) {
            const size_t end = (k < cuts.size()) ? cuts[k] : (size_t)(last - first);
            if (k < cuts.size()) pool.spawn(helpers, [this, begin, end]() { insertionSort(base + begin, base + end); }, PlannerPool::SHORT);
            else insertionSort(base + begin, base + end);
            begin = end;
        }
        pool.wait(helpers, PlannerPool::SHORT);
    }
};

struct BvhTask { int innerId; uint32_t begin, end, parentSlot; };      // a range left to the device; parentSlot: index of the subtree's sphere in units of 4 doubles

struct HostBvhBuilder {
    const float* verts; const uint32_t* idx;
    double* sph;       // 8 doubles per inner node: spheres of the left and of the right child
    int* kids;         // 2 ints per inner node: child references (>= 0 inner node, < 0 ~triangle)
    std::vector<int> order;
    int maxParallelDepth = 0;
    // offload (planBvhHost): ranges of at most offloadMax triangles are not planned here but listed for k_bvh_subtrees; the inner nodes
    // that ARE planned here are listed too (their records are scattered into the device arrays)
    uint32_t offloadMax = 0;
    std::vector<BvhTask> tasks; std::vector<int> hostNodes; std::mutex listLock;

    struct D { double x, y, z; };
    KeyTri* scratchKeys = nullptr; float* scratchLoc = nullptr; uint32_t* scratchL = nullptr; uint32_t* scratchR = nullptr;      // T entries / 9 T floats, uninitialised, sliced by range
    const float* triV = nullptr;       // 9 floats per triangle, gathered once (the planner reads every vertex ~2 log2(T) times)
    int sortThreads = 1;
    D vtx(int t, int k) const { const float* q = triV + 9 * (size_t)t + 3 * k; return D{(double)q[0], (double)q[1], (double)q[2]}; }
    static double comp(const D& a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

    static constexpr int kWideRange = 1 << 16;
    // chunks of [0,n) on the planner's pool (at most 16, at least 16k items each); fn(i0, i1)
    template <typename F> static void parallelFor(int n, F fn) {
        int parts = n / 16384; if (parts > 16) parts = 16; if (parts < 1) parts = 1;
        PlannerPool::get().run(parts, [&](int p) { fn((int)((long long)n * p / parts), (int)((long long)n * (p + 1) / parts)); });
    }

    // Plans the subtree over order[begin,end): writes its bounding sphere into out[0..3] and returns the reference to it.
    // `innerId` = pre-order index this subtree's root gets if it is an inner node (n > 1).
    int build(int innerId, double* out, int begin, int end, int depth) {
        const int n = end - begin;
        if (offloadMax && n > 1 && (uint32_t)n <= offloadMax && out >= sph) {      // (`out` outside the array: the root's own sphere, which nobody reads)
            std::lock_guard<std::mutex> g(listLock);
            tasks.push_back(BvhTask{innerId, (uint32_t)begin, (uint32_t)end, (uint32_t)((out - sph) / 4)});
            return innerId;
        }
        if (offloadMax && n > 1) { std::lock_guard<std::mutex> g(listLock); hostNodes.push_back(innerId); }
        if (n == 1) {
            const int t = order[begin];
            const D a = vtx(t, 0), b = vtx(t, 1), c = vtx(t, 2);
            const D s = D{(a.x + b.x) + c.x, (a.y + b.y) + c.y, (a.z + b.z) + c.z};
            const D ce = D{s.x / 3.0, s.y / 3.0, s.z / 3.0};
            auto dist = [&](const D& p) { const double dx = p.x - ce.x, dy = p.y - ce.y, dz = p.z - ce.z; return std::sqrt(dx * dx + dy * dy + dz * dz); };
            out[0] = ce.x; out[1] = ce.y; out[2] = ce.z;
            out[3] = std::max(std::max(dist(a), dist(b)), dist(c));
            return ~t;
        }
        const double lo = std::numeric_limits<double>::lowest(), hi = std::numeric_limits<double>::max();
        D top{lo, lo, lo}, bot{hi, hi, hi}, ce{0, 0, 0};
        double r2 = 0.0;
        int dim = 0;
        if (n >= kWideRange) {
            // Large ranges (the top of the tree = the planner's critical path): only the centre sum depends on the order of its
            // operands; the gather of the range's vertices, the AABB, the radius (max of identical expressions), the keys and
            // the write-back are order independent and run on helper threads over a contiguous copy of the range.
            const bool trace = depth == 0 && getenv("SDFHIP_TIMING"); double tq = nowSeconds();
            auto lapq = [&](const char* what) { if (trace) { const double now = nowSeconds(); fprintf(stderr, "[sdfhip] bvh root: %s %.4f s\n", what, now - tq); tq = now; } };
            const float* loc = triV;                          // the root's range is the identity: no copy needed
            if (depth != 0) {
                float* copy = scratchLoc + 9 * (size_t)begin;     // this node's slice of the planner-wide scratch (ranges in flight are disjoint)
                parallelFor(n, [&](int i0, int i1) { for (int i = i0; i < i1; i++) std::memcpy(&copy[9 * (size_t)i], triV + 9 * (size_t)order[begin + i], 36); });
                loc = copy;
            }
            lapq("gather");
            // the centre sum is a serial chain of 3 n additions per coordinate (2 ms for 1.3 M triangles) that nothing but the radius
            // waits for: it runs as a task of its own under the AABB, the keys and the sort
            PlannerPool::Group sumTask;
            PlannerPool::get().spawn(sumTask, [&]() {
                double sx = 0.0, sy = 0.0, sz = 0.0;
                for (size_t j = 0; j < 3 * (size_t)n; j++) { sx += (double)loc[3 * j]; sy += (double)loc[3 * j + 1]; sz += (double)loc[3 * j + 2]; }
                const double cnt = (double)(3 * n);
                ce.x = sx / cnt; ce.y = sy / cnt; ce.z = sz / cnt;
            }, PlannerPool::MEDIUM);
            std::mutex m;
            parallelFor(n, [&](int i0, int i1) {
                D t{lo, lo, lo}, bt{hi, hi, hi};
                for (size_t j = 3 * (size_t)i0; j < 3 * (size_t)i1; j++) {
                    const D p{(double)loc[3 * j], (double)loc[3 * j + 1], (double)loc[3 * j + 2]};
                    t.x = std::max(t.x, p.x); bt.x = std::min(bt.x, p.x); t.y = std::max(t.y, p.y); bt.y = std::min(bt.y, p.y); t.z = std::max(t.z, p.z); bt.z = std::min(bt.z, p.z);
                }
                std::lock_guard<std::mutex> g(m);
                top.x = std::max(top.x, t.x); top.y = std::max(top.y, t.y); top.z = std::max(top.z, t.z);
                bot.x = std::min(bot.x, bt.x); bot.y = std::min(bot.y, bt.y); bot.z = std::min(bot.z, bt.z);
            });
            lapq("aabb");
            const double diag[3] = {top.x - bot.x, top.y - bot.y, top.z - bot.z};
            dim = (int)(std::max_element(diag, diag + 3) - diag);
            KeyTri* tmp = scratchKeys + begin;
            parallelFor(n, [&](int i0, int i1) { for (int i = i0; i < i1; i++) tmp[i] = KeyTri{loc[9 * (size_t)i + dim], order[begin + i]}; });
            IntroSortLike sorter; sorter.maxThreads = sortThreads; sorter.scratchL = scratchL + begin; sorter.scratchR = scratchR + begin;
            { const char* e1 = getenv("SDFHIP_BVH_MIN_PARALLEL"); const char* e2 = getenv("SDFHIP_BVH_PAR_PARTITION");
              if (e1) sorter.minParallel = (size_t)atol(e1); if (e2) sorter.minParPartition = (size_t)atol(e2); }
            lapq("keys");
            sorter.sort(tmp, tmp + n);
            lapq("sort");
            parallelFor(n, [&](int i0, int i1) { for (int i = i0; i < i1; i++) order[begin + i] = tmp[i].tri; });
            lapq("write back");
            PlannerPool::get().wait(sumTask, PlannerPool::MEDIUM);
            lapq("wait for the centre sum");
            parallelFor(n, [&](int i0, int i1) {
                double rr = 0.0;
                for (size_t j = 3 * (size_t)i0; j < 3 * (size_t)i1; j++) {
                    const double dx = ce.x - (double)loc[3 * j], dy = ce.y - (double)loc[3 * j + 1], dz = ce.z - (double)loc[3 * j + 2];
                    rr = std::max(rr, dx * dx + dy * dy + dz * dz);
                }
                std::lock_guard<std::mutex> g(m);
                r2 = std::max(r2, rr);
            });
            lapq("radius");
        } else {
            // one sweep gathers the node's vertices into its slice of the scratch, sums them in range order (the only order-dependent
            // quantity) and takes the AABB on the floats themselves (exact; same doubles after conversion); the radius and the keys
            // then read the contiguous copy
            float* loc = scratchLoc + 9 * (size_t)begin;
            double sx = 0.0, sy = 0.0, sz = 0.0;
            const float fhi = std::numeric_limits<float>::max();
            float tx = -fhi, ty = -fhi, tz = -fhi, bx = fhi, by = fhi, bz = fhi;
            for (int i = 0; i < n; i++) {
                const float* q = triV + 9 * (size_t)order[begin + i];
                float* w = loc + 9 * (size_t)i;
                for (int k = 0; k < 9; k++) w[k] = q[k];
                for (int k = 0; k < 3; k++) {
                    const float px = q[3 * k], py = q[3 * k + 1], pz = q[3 * k + 2];
                    sx += (double)px; sy += (double)py; sz += (double)pz;
                    tx = px > tx ? px : tx; bx = px < bx ? px : bx; ty = py > ty ? py : ty; by = py < by ? py : by; tz = pz > tz ? pz : tz; bz = pz < bz ? pz : bz;
                }
            }
            const double cnt = (double)(3 * n);
            ce.x = sx / cnt; ce.y = sy / cnt; ce.z = sz / cnt;
            top = D{(double)tx, (double)ty, (double)tz}; bot = D{(double)bx, (double)by, (double)bz};
            const double diag[3] = {top.x - bot.x, top.y - bot.y, top.z - bot.z};
            dim = (int)(std::max_element(diag, diag + 3) - diag);
            double ra = 0.0, rb = 0.0, rc = 0.0;                 // a maximum: any grouping gives the same value
            for (int i = 0; i < n; i++) {
                const float* w = loc + 9 * (size_t)i;
                const double ax = ce.x - (double)w[0], ay = ce.y - (double)w[1], az = ce.z - (double)w[2];
                const double bx2 = ce.x - (double)w[3], by2 = ce.y - (double)w[4], bz2 = ce.z - (double)w[5];
                const double cx2 = ce.x - (double)w[6], cy2 = ce.y - (double)w[7], cz2 = ce.z - (double)w[8];
                ra = std::max(ra, ax * ax + ay * ay + az * az); rb = std::max(rb, bx2 * bx2 + by2 * by2 + bz2 * bz2); rc = std::max(rc, cx2 * cx2 + cy2 * cy2 + cz2 * cz2);
            }
            r2 = std::max(ra, std::max(rb, rc));
            // median split: sort the range by the first vertex's coordinate along `dim`
            KeyTri* tmp = scratchKeys + begin;
            for (int i = 0; i < n; i++) tmp[i] = KeyTri{loc[9 * (size_t)i + dim], order[begin + i]};
            if (n <= 16) IntroSortLike::insertionSort(tmp, tmp + n);          // what introsort does with a range this short
            else {
                IntroSortLike sorter; sorter.maxThreads = sortThreads; sorter.scratchL = scratchL + begin; sorter.scratchR = scratchR + begin;
                if (const char* e = getenv("SDFHIP_BVH_LIST_PARTITION")) sorter.minListPartition = (size_t)atol(e);
                sorter.sort(tmp, tmp + n);
            }
            for (int i = 0; i < n; i++) order[begin + i] = tmp[i].tri;
        }
        out[0] = ce.x; out[1] = ce.y; out[2] = ce.z; out[3] = std::sqrt(r2);
        const int mid = (int)(0.5 * (begin + end));
        // pre-order numbering of inner nodes: the left subtree holds (mid - begin) - 1 of them
        const int leftId = innerId + 1, rightId = innerId + (mid - begin);
        double* nd = sph + 8 * (size_t)innerId;
        int refs[2];
        if (depth < maxParallelDepth && n > 4096) {
            PlannerPool::Group both;
            PlannerPool::get().spawn(both, [&]() { refs[0] = build(leftId, nd, begin, mid, depth + 1); }, PlannerPool::LONG);
            refs[1] = build(rightId, nd + 4, mid, end, depth + 1);
            PlannerPool::get().wait(both, PlannerPool::LONG);
        } else {
            refs[0] = build(leftId, nd, begin, mid, depth + 1);
            refs[1] = build(rightId, nd + 4, mid, end, depth + 1);
        }
        kids[2 * (size_t)innerId] = refs[0]; kids[2 * (size_t)innerId + 1] = refs[1];
        return innerId;
    }
};

// ---- the bottom of the tree on the device ---------------------------------------------------------------------------------------------
// OPT-IN (SDFHIP_BVH_DEVICE_SUBTREES=1).  The host planner hands every range of at most kDevSubtreeMax triangles to the device (planBvhHost
// with offload): those ranges are the bottom twelve of the tree's twenty levels, 45 % of the planner's CPU time.  The trees are identical
// (tests/test_gpu_octree.py::test_hybrid_bvh_plan_equals_the_oracles_tree) but the build is not faster, see sdfhip_mesh_build_bvh.
// One workgroup builds one such subtree, level by level, its {key, triangle} array in LDS:
//   * a node is ONE lane's work for everything whose result depends on an order: the vertices are summed in range order in fp64 (the
//     reference's centre), AABB -> split axis, radius, keys; the same expressions as HostBvhBuilder::build, operand for operand;
//   * the range is then sorted by libstdc++'s introsort, restated once more (IntroSortLike above is the host's): median of three to the
//     front, unguarded Hoare partition, ranges of at most 16 finished by insertion sort — a partition is one lane's work, the two
//     parts it leaves are independent tasks of the next round, so a level's sorts cost about 3 n sequential steps, not n log n.  The
//     permutation among tied keys is the sequential algorithm's because every comparison and swap is.  A range that exhausts
//     introsort's depth limit (heap sort in libstdc++) raises a flag and the whole tree is planned on the host instead;
//   * node ids follow from the pre-order numbering (left child = id + 1, right child = id + (mid - begin)), so the subtree writes its
//     records straight into the device arrays, and its own sphere into its parent's record (planned on the host).
struct BvhDevNode { int id; uint32_t b, e, slot; };
constexpr uint32_t kDevSubtreeMaxLimit = 16384;          // 128 KB of LDS for the keys + the sort task lists
constexpr uint32_t kDevSortTasks = 1024;

SDF_DEV void devInsertionSort(KeyTri* a, int first, int last) {
    for (int i = first + 1; i < last; i++) {
        const KeyTri val = a[i];
        if (val.key < a[first].key) { for (int k = i; k > first; k--) a[k] = a[k - 1]; a[first] = val; }
        else { int pos = i; while (val.key < a[pos - 1].key) { a[pos] = a[pos - 1]; pos--; } a[pos] = val; }
    }
}
SDF_DEV void devSwap(KeyTri* a, int i, int j) { const KeyTri t = a[i]; a[i] = a[j]; a[j] = t; }
// libstdc++'s heap sort (what std::sort falls back to when a range exhausts introsort's depth limit: __partial_sort(first, last, last) =
// __make_heap + __sort_heap), restated move for move: __push_heap, __adjust_heap, __pop_heap (bits/stl_heap.h).  Compiled for the host too:
// sdfhip_test_heap_sort_matches_std compares it with std::make_heap / std::sort_heap on tie-heavy keys.
SDF_HD void stdPushHeap(KeyTri* first, int holeIndex, int topIndex, KeyTri value) {
    int parent = (holeIndex - 1) / 2;
    while (holeIndex > topIndex && first[parent].key < value.key) { first[holeIndex] = first[parent]; holeIndex = parent; parent = (holeIndex - 1) / 2; }
    first[holeIndex] = value;
}
SDF_HD void stdAdjustHeap(KeyTri* first, int holeIndex, int len, KeyTri value) {
    const int topIndex = holeIndex;
    int secondChild = holeIndex;
    while (secondChild < (len - 1) / 2) {
        secondChild = 2 * (secondChild + 1);
        if (first[secondChild].key < first[secondChild - 1].key) secondChild--;
        first[holeIndex] = first[secondChild];
        holeIndex = secondChild;
    }
    if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
        secondChild = 2 * (secondChild + 1);
        first[holeIndex] = first[secondChild - 1];
        holeIndex = secondChild - 1;
    }
    stdPushHeap(first, holeIndex, topIndex, value);
}
SDF_HD void stdHeapSort(KeyTri* first, int len) {
    if (len >= 2) for (int parent = (len - 2) / 2;; parent--) { stdAdjustHeap(first, parent, len, first[parent]); if (parent == 0) break; }      // __make_heap
    for (int last = len; last > 1;) { --last; const KeyTri value = first[last]; first[last] = first[0]; stdAdjustHeap(first, 0, last, value); }   // __sort_heap / __pop_heap
}
// libstdc++'s __move_median_to_first(result, a, b, c)
SDF_DEV void devMedianToFirst(KeyTri* k, int result, int a, int b, int c) {
    if (k[a].key < k[b].key) {
        if (k[b].key < k[c].key) devSwap(k, result, b);
        else if (k[a].key < k[c].key) devSwap(k, result, c);
        else devSwap(k, result, a);
    } else if (k[a].key < k[c].key) devSwap(k, result, a);
    else if (k[b].key < k[c].key) devSwap(k, result, c);
    else devSwap(k, result, b);
}
// libstdc++'s __unguarded_partition(first, last, pivot)
SDF_DEV int devPartition(KeyTri* k, int first, int last, int pivot) {
    const float pk = k[pivot].key;
    for (;;) {
        while (k[first].key < pk) ++first;
        --last;
        while (pk < k[last].key) --last;
        if (!(first < last)) return first;
        devSwap(k, first, last);
        ++first;
    }
}
struct DevV3 { float x, y, z; };
SDF_DEV void devTriVerts(const float4* __restrict__ triV, int t, DevV3& a, DevV3& b, DevV3& c) {
    const float4 q0 = triV[3 * (size_t)t], q1 = triV[3 * (size_t)t + 1], q2 = triV[3 * (size_t)t + 2];
    a = DevV3{q0.x, q0.y, q0.z}; b = DevV3{q0.w, q1.x, q1.y}; c = DevV3{q1.z, q1.w, q2.x};
}
SDF_DEV float devComp(const DevV3& v, int d) { return d == 0 ? v.x : (d == 1 ? v.y : v.z); }

__global__ void __launch_bounds__(256) k_bvh_subtrees(const BvhTask* __restrict__ tasks, const uint32_t* __restrict__ order, const float4* __restrict__ triV,
                                                      double* __restrict__ sph, int* __restrict__ kids, BvhDevNode* __restrict__ nodeScratch, uint32_t* __restrict__ failed, uint32_t kDevSubtreeMax) {
    extern __shared__ unsigned char s_bvh_raw[];
    KeyTri* keys = reinterpret_cast<KeyTri*>(s_bvh_raw);
    uint32_t* sortA = reinterpret_cast<uint32_t*>(s_bvh_raw + sizeof(KeyTri) * kDevSubtreeMax);        // {first, last, depth limit} x 3 words
    uint32_t* sortB = sortA + 3 * kDevSortTasks;
    __shared__ uint32_t s_count[4];                  // [0] nodes of the next level, [1] sort tasks of the next round, [2] overflow / depth-limit flag
    const BvhTask T = tasks[blockIdx.x];
    const uint32_t n = T.end - T.begin;
    const int tid = threadIdx.x;
    BvhDevNode* cur = nodeScratch + (size_t)blockIdx.x * 2u * (kDevSubtreeMax / 2u + 1u);
    BvhDevNode* nxt = cur + (kDevSubtreeMax / 2u + 1u);
    for (uint32_t i = tid; i < n; i += 256) keys[i] = KeyTri{0.f, (int)order[T.begin + i]};
    if (tid == 0) { cur[0] = BvhDevNode{T.innerId, T.begin, T.end, T.parentSlot}; s_count[2] = 0; }
    uint32_t nCur = 1;
    __syncthreads();
    while (nCur > 0) {
        if (tid == 0) { s_count[0] = 0; s_count[1] = 0; }
        __syncthreads();
        // ---- A. one lane per node: centre (ordered fp64 sum), AABB -> axis, radius, keys; short ranges sorted at once
        for (uint32_t j = tid; j < nCur; j += 256) {
            const BvhDevNode nd = cur[j];
            const int lo = (int)(nd.b - T.begin), nn = (int)(nd.e - nd.b);
            double sx = 0.0, sy = 0.0, sz = 0.0;
            const float fhi = 3.402823466e+38f;
            float tx = -fhi, ty = -fhi, tz = -fhi, bx = fhi, by = fhi, bz = fhi;
            for (int i = 0; i < nn; i++) {
                DevV3 v[3]; devTriVerts(triV, keys[lo + i].tri, v[0], v[1], v[2]);
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    sx += (double)v[k].x; sy += (double)v[k].y; sz += (double)v[k].z;
                    tx = v[k].x > tx ? v[k].x : tx; bx = v[k].x < bx ? v[k].x : bx; ty = v[k].y > ty ? v[k].y : ty; by = v[k].y < by ? v[k].y : by;
                    tz = v[k].z > tz ? v[k].z : tz; bz = v[k].z < bz ? v[k].z : bz;
                }
            }
            const double cnt = (double)(3 * nn);
            const double cx = sx / cnt, cy = sy / cnt, cz = sz / cnt;
            const double d0 = (double)tx - (double)bx, d1 = (double)ty - (double)by, d2 = (double)tz - (double)bz;
            int dim = 0; double dm = d0;                       // std::max_element: the first of equal maxima
            if (dm < d1) { dim = 1; dm = d1; }
            if (dm < d2) dim = 2;
            double r2 = 0.0;
            for (int i = 0; i < nn; i++) {
                DevV3 v[3]; devTriVerts(triV, keys[lo + i].tri, v[0], v[1], v[2]);
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const double dx = cx - (double)v[k].x, dy = cy - (double)v[k].y, dz = cz - (double)v[k].z;
                    const double q = dx * dx + dy * dy + dz * dz;
                    r2 = r2 < q ? q : r2;
                }
                keys[lo + i].key = devComp(v[0], dim);
            }
            if (nd.slot != 0xFFFFFFFFu) { double* o = sph + 4 * (size_t)nd.slot; o[0] = cx; o[1] = cy; o[2] = cz; o[3] = sqrt(r2); }
            if (nn <= 16) devInsertionSort(keys, lo, lo + nn);
            else {
                int lg = 0; for (int m = nn; m > 1; m >>= 1) lg++;
                const uint32_t at = atomicAdd(&s_count[1], 1u);
                if (at < kDevSortTasks) { sortA[3 * at] = (uint32_t)lo; sortA[3 * at + 1] = (uint32_t)(lo + nn); sortA[3 * at + 2] = (uint32_t)(2 * lg); }
                else atomicOr(&s_count[2], 1u);
            }
        }
        __syncthreads();
        // ---- B. introsort rounds: every pending range is partitioned by one lane; parts of at most 16 are finished on the spot
        uint32_t nSort = s_count[1] < kDevSortTasks ? s_count[1] : kDevSortTasks;
        uint32_t* in = sortA; uint32_t* out = sortB;
        __syncthreads();
        while (nSort > 0) {
            if (tid == 0) s_count[1] = 0;
            __syncthreads();
            for (uint32_t t = tid; t < nSort; t += 256) {
                const int first = (int)in[3 * t], last = (int)in[3 * t + 1]; const int depth = (int)in[3 * t + 2];
                if (depth == 0) { stdHeapSort(keys + first, last - first); continue; }        // libstdc++ switches to heap sort here (it does happen: 1.31 M triangles)
                devMedianToFirst(keys, first, first + 1, first + (last - first) / 2, last - 1);
                const int cut = devPartition(keys, first + 1, last, first);
                const int parts[2][2] = {{first, cut}, {cut, last}};
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int f = parts[h][0], l = parts[h][1];
                    if (l - f > 16) {
                        const uint32_t at = atomicAdd(&s_count[1], 1u);
                        if (at < kDevSortTasks) { out[3 * at] = (uint32_t)f; out[3 * at + 1] = (uint32_t)l; out[3 * at + 2] = (uint32_t)(depth - 1); }
                        else atomicOr(&s_count[2], 4u);
                    } else devInsertionSort(keys, f, l);
                }
            }
            __syncthreads();
            nSort = s_count[1] < kDevSortTasks ? s_count[1] : kDevSortTasks;
            uint32_t* sw = in; in = out; out = sw;
            __syncthreads();
        }
        // ---- C. one lane per node: the children (leaves get their sphere here, inner children at the next level)
        for (uint32_t j = tid; j < nCur; j += 256) {
            const BvhDevNode nd = cur[j];
            const uint32_t mid = (nd.b + nd.e) >> 1;           // (int)(0.5 * (begin + end))
            const uint32_t rb[2] = {nd.b, mid}, re[2] = {mid, nd.e};
            const int childId[2] = {nd.id + 1, nd.id + (int)(mid - nd.b)};
#pragma unroll
            for (int side = 0; side < 2; side++) {
                const uint32_t slot = 2u * (uint32_t)nd.id + (uint32_t)side;
                if (re[side] - rb[side] == 1u) {
                    const int t = keys[rb[side] - T.begin].tri;
                    DevV3 a, b, c; devTriVerts(triV, t, a, b, c);
                    const double sx = ((double)a.x + (double)b.x) + (double)c.x, sy = ((double)a.y + (double)b.y) + (double)c.y, sz = ((double)a.z + (double)b.z) + (double)c.z;
                    const double cx = sx / 3.0, cy = sy / 3.0, cz = sz / 3.0;
                    auto dist = [&](const DevV3& p) { const double dx = (double)p.x - cx, dy = (double)p.y - cy, dz = (double)p.z - cz; return sqrt(dx * dx + dy * dy + dz * dz); };
                    const double da = dist(a), db = dist(b), dc = dist(c);
                    const double m1 = da < db ? db : da;
                    double* o = sph + 4 * (size_t)slot; o[0] = cx; o[1] = cy; o[2] = cz; o[3] = m1 < dc ? dc : m1;
                    kids[slot] = ~t;
                } else {
                    kids[slot] = childId[side];
                    const uint32_t at = atomicAdd(&s_count[0], 1u);
                    nxt[at] = BvhDevNode{childId[side], rb[side], re[side], slot};
                }
            }
        }
        __syncthreads();
        nCur = s_count[0];
        BvhDevNode* sw = cur; cur = nxt; nxt = sw;
        __syncthreads();
    }
    if (tid == 0 && s_count[2]) atomicOr(failed, s_count[2]);
}

// the records the host planned (the top of the tree), scattered to their pre-order positions; a half whose child is a device subtree is
// left alone (the subtree writes its own sphere there)
__global__ void k_bvh_scatter_top(const int* __restrict__ ids, const double* __restrict__ sph8, const int* __restrict__ kids2, const unsigned char* __restrict__ halfOwned, uint32_t count,
                                  double* __restrict__ sph, int* __restrict__ kids) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int id = ids[i];
    for (int h = 0; h < 2; h++) {
        kids[2 * (size_t)id + h] = kids2[2 * (size_t)i + h];
        if (halfOwned[2 * (size_t)i + h]) for (int k = 0; k < 4; k++) sph[8 * (size_t)id + 4 * h + k] = sph8[8 * (size_t)i + 4 * h + k];
    }
}

__global__ void k_sph32(const double* __restrict__ sph, uint64_t n, float* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)sph[i];            // round to nearest: the bracket in dev_bvh.h assumes |c32 - c| <= 2^-24 |c|
}

__global__ void k_tri_verts(const float* __restrict__ verts, const uint32_t* __restrict__ idx, uint32_t numTriangles, float* __restrict__ triV) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t = gid / 12u, k = gid - 12u * t;
    if (t >= numTriangles) return;
    float v = 0.f;
    if (k < 9u) v = verts[3 * (size_t)idx[3 * (size_t)t + k / 3u] + (k % 3u)];
    else if (k == 9u) {
        // flag (nearly) degenerate triangles: sin^2 of the angle at v0 below 1e-3, a zero edge, or anything not finite.  The fp32
        // point/triangle distance divides by det = |e0|^2 |e1|^2 sin^2; its error bound (dev_bvh_fast.h) holds for the others only.
        const uint32_t a = idx[3 * (size_t)t], b = idx[3 * (size_t)t + 1], c = idx[3 * (size_t)t + 2];
        const F3 v0 = F3{verts[3 * (size_t)a], verts[3 * (size_t)a + 1], verts[3 * (size_t)a + 2]};
        const F3 e0 = F3{verts[3 * (size_t)b], verts[3 * (size_t)b + 1], verts[3 * (size_t)b + 2]} - v0, e1 = F3{verts[3 * (size_t)c], verts[3 * (size_t)c + 1], verts[3 * (size_t)c + 2]} - v0;
        const float a00 = dot(e0, e0), a01 = dot(e0, e1), a11 = dot(e1, e1);
        const float det = fabsf(a00 * a11 - a01 * a01);
        v = (det > 1e-3f * (a00 * a11) && a00 * a11 > 0.f && a00 * a11 < 1e37f) ? 0.f : 1.f;
    }
    triV[gid] = v;
}

// Leaf-order position of every triangle from the child references alone: inner nodes are numbered in pre-order and the node over
// [b, e) splits at (b + e) / 2, so the range of inner node i follows from i by arithmetic (descend from the root: the left subtree
// of a node holds the inner indices node + 1 .. node + (mid - b) - 1); its leaf children sit at the ends of that range.
__global__ void k_tri_ranks(const int2* __restrict__ kids, uint32_t numTriangles, uint32_t* __restrict__ triRank) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (numTriangles == 1u) { if (i == 0u) triRank[0] = 0u; return; }
    if (i >= numTriangles - 1u) return;
    uint32_t node = 0, b = 0, e = numTriangles;
    while (node != i) {
        const uint32_t mid = (b + e) >> 1;
        if (i < node + (mid - b)) { node = node + 1u; e = mid; } else { node = node + (mid - b); b = mid; }
    }
    const int2 k = kids[i];
    if (k.x < 0) triRank[~k.x] = b;
    if (k.y < 0) triRank[~k.y] = e - 1u;
}

// 4-wide nodes for the candidate search (layout: dev_bvh.h).  One thread per binary inner node; nodes at odd depths are skipped.
// The radius of a child is inflated by the MEASURED distance between its fp64 centre and the centre the traversal will decode
// (same expression: fmaf(q, scale, origin)), then rounded up to a half: the decoded sphere contains the reference's sphere.
__device__ __forceinline__ unsigned short halfRoundedUp(float f) {
    const _Float16 h = (_Float16)f;
    unsigned short bits = __builtin_bit_cast(unsigned short, h);
    if ((float)h < f) bits = (bits == 0x8000u) ? (unsigned short)0x0001u : ((bits & 0x8000u) ? (unsigned short)(bits - 1u) : (unsigned short)(bits + 1u));
    return bits;
}
__global__ void k_tri_at_rank(const uint32_t* __restrict__ triRank, uint32_t numTriangles, uint32_t* __restrict__ triAtRank) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < numTriangles) triAtRank[triRank[t]] = t;
}
// A child of a wide node = a sphere AND a slab: every vertex x of its triangles satisfies |x - c'| <= r' and |m . (x - c')| <= W, with
// c' the DECODED centre, m the DECODED direction (3 x snorm16 of the normalised sum of the subtree's area normals: a smooth patch is
// thin along it) and r', W measured against those decoded values here, so that nothing about the quantisation has to be bounded
// analytically.  Subtrees above WIDE_SLAB_MAX triangles get no slab (W = +inf): it would not be thin, and the loops below are per thread.
constexpr uint32_t WIDE_SLAB_MAX = 2048;
constexpr uint32_t WIDE_SLAB_SPLIT = 192;          // children with more triangles get their slab from k_wide_slabs_big (a block each) instead of 16 lanes
__global__ void k_wide_nodes(const int2* __restrict__ kids, const double2* __restrict__ sph, const float4* __restrict__ triV, const uint32_t* __restrict__ triAtRank,
                             uint32_t numTriangles, uint4* __restrict__ wide, uint4* __restrict__ bigList, uint32_t* __restrict__ bigCount, uint32_t bigCap) {
    // 16 lanes per node: they share the (cheap) header work and stride over the children's triangles for the two reductions
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = gid >> 4, sub = gid & 15u;
    if (numTriangles < 2u || i >= numTriangles - 1u) return;
    uint32_t node = 0, b = 0, e = numTriangles, depth = 0;
    while (node != i) {
        const uint32_t mid = (b + e) >> 1;
        if (i < node + (mid - b)) { node = node + 1u; e = mid; } else { node = node + (mid - b); b = mid; }
        depth++;
    }
    if (depth & 1u) return;
    double cx[4], cy[4], cz[4], cr[4]; int ref[4]; uint32_t rb[4], re[4]; int n = 0;
    const int2 k0 = kids[i];
    const uint32_t mid0 = (b + e) >> 1;
    for (int s1 = 0; s1 < 2; s1++) {
        const int c1 = s1 ? k0.y : k0.x;
        const uint32_t b1 = s1 ? mid0 : b, e1 = s1 ? e : mid0;
        if (c1 < 0) {
            const double2 a = sph[4 * (size_t)i + 2 * s1], bb = sph[4 * (size_t)i + 2 * s1 + 1];
            cx[n] = a.x; cy[n] = a.y; cz[n] = bb.x; cr[n] = bb.y; ref[n] = c1; rb[n] = b1; re[n] = e1; n++;
        } else {
            const int2 k1 = kids[c1];
            const uint32_t mid1 = (b1 + e1) >> 1;
            for (int s2 = 0; s2 < 2; s2++) {
                const double2 a = sph[4 * (size_t)c1 + 2 * s2], bb = sph[4 * (size_t)c1 + 2 * s2 + 1];
                cx[n] = a.x; cy[n] = a.y; cz[n] = bb.x; cr[n] = bb.y; ref[n] = s2 ? k1.y : k1.x; rb[n] = s2 ? mid1 : b1; re[n] = s2 ? e1 : mid1; n++;
            }
        }
    }
    double lo[3] = {cx[0], cy[0], cz[0]}, hi[3] = {cx[0], cy[0], cz[0]};
    for (int c = 1; c < n; c++) {
        lo[0] = fmin(lo[0], cx[c]); hi[0] = fmax(hi[0], cx[c]); lo[1] = fmin(lo[1], cy[c]); hi[1] = fmax(hi[1], cy[c]); lo[2] = fmin(lo[2], cz[c]); hi[2] = fmax(hi[2], cz[c]);
    }
    const float ox = (float)lo[0], oy = (float)lo[1], oz = (float)lo[2];
    float scale = (float)(fmax(fmax(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]) / 65535.0);
    if (!(scale > 1e-30f)) scale = 1e-30f;
    uint4* out = wide + 8 * (size_t)i;
    if (sub == 0u) out[0] = make_uint4(__float_as_uint(ox), __float_as_uint(oy), __float_as_uint(oz), __float_as_uint(scale));
    uint32_t refs[4] = {0u, 0u, 0u, 0u};
    for (int c = 0; c < 4; c++) {
        uint32_t qx = 0, qy = 0, qz = 0, mx = 0, my = 0, mz = 0; unsigned short rh = 0xFC00u, wh = 0x7C00u;       // empty slot: radius -inf
        if (c < n) {
            auto quant = [&](double v, float o) { double q = rint((v - (double)o) / (double)scale); if (!(q >= 0.0)) q = 0.0; if (q > 65535.0) q = 65535.0; return (uint32_t)q; };
            qx = quant(cx[c], ox); qy = quant(cy[c], oy); qz = quant(cz[c], oz);
            const double dcx = (double)fmaf((float)qx, scale, ox), dcy = (double)fmaf((float)qy, scale, oy), dcz = (double)fmaf((float)qz, scale, oz);   // the decoded centre
            const double ex = dcx - cx[c], ey = dcy - cy[c], ez = dcz - cz[c];
            const double rInfl = (cr[c] + sqrt(ex * ex + ey * ey + ez * ez)) * (1.0 + 1e-9) + 1e-300;
            float rf = (float)rInfl; if ((double)rf < rInfl) rf = nextafterf(rf, 3.0e38f);
            rh = halfRoundedUp(rf);                      // +inf when the radius exceeds the half range: the child is then always visited
            refs[c] = (uint32_t)ref[c];
            const uint32_t cnt = re[c] - rb[c];
            if (cnt > WIDE_SLAB_SPLIT && cnt <= WIDE_SLAB_MAX) {
                // a long strided loop here would hold the 16 lanes (and the wave) for thousands of steps: listed for k_wide_slabs_big, which
                // fills in the slab words of this child (until then: no slab, W = +inf)
                if (sub == 0u) { const uint32_t at = atomicAdd(bigCount, 1u); if (at < bigCap) bigList[at] = make_uint4(i, (uint32_t)c, rb[c], re[c]); }
            } else if (cnt <= WIDE_SLAB_MAX) {
                double sx = 0, sy = 0, sz = 0;
                for (uint32_t k = rb[c] + sub; k < re[c]; k += 16u) {
                    const uint32_t t = triAtRank[k];
                    const float4 q0 = triV[3 * (size_t)t], q1 = triV[3 * (size_t)t + 1], q2 = triV[3 * (size_t)t + 2];
                    const double ux = (double)q0.w - q0.x, uy = (double)q1.x - q0.y, uz = (double)q1.y - q0.z, vx = (double)q1.z - q0.x, vy = (double)q1.w - q0.y, vz = (double)q2.x - q0.z;
                    sx += uy * vz - uz * vy; sy += uz * vx - ux * vz; sz += ux * vy - uy * vx;
                }
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) { sx += __shfl_xor(sx, o, 16); sy += __shfl_xor(sy, o, 16); sz += __shfl_xor(sz, o, 16); }
                const double len = sqrt(sx * sx + sy * sy + sz * sz);
                if (len > 1e-300 && len < 1e300) {
                    auto snorm = [&](double v) { double q = rint(v / len * 32767.0); if (q < -32767.0) q = -32767.0; if (q > 32767.0) q = 32767.0; return (int)q; };
                    const int ix = snorm(sx), iy = snorm(sy), iz = snorm(sz);
                    const double dmx = (double)((float)ix * (1.0f / 32767.0f)), dmy = (double)((float)iy * (1.0f / 32767.0f)), dmz = (double)((float)iz * (1.0f / 32767.0f));   // the decoded direction
                    double W = 0.0;
                    for (uint32_t k = rb[c] + sub; k < re[c]; k += 16u) {
                        const uint32_t t = triAtRank[k];
                        const float4 q0 = triV[3 * (size_t)t], q1 = triV[3 * (size_t)t + 1], q2 = triV[3 * (size_t)t + 2];
                        const double px[3] = {q0.x, q0.w, q1.z}, py[3] = {q0.y, q1.x, q1.w}, pz[3] = {q0.z, q1.y, q2.x};
                        for (int j = 0; j < 3; j++) W = fmax(W, fabs(dmx * (px[j] - dcx) + dmy * (py[j] - dcy) + dmz * (pz[j] - dcz)));
                    }
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) W = fmax(W, __shfl_xor(W, o, 16));
                    const double Winfl = W * (1.0 + 1e-9) + 1e-300;
                    float wf = (float)Winfl; if ((double)wf < Winfl) wf = nextafterf(wf, 3.0e38f);
                    wh = halfRoundedUp(wf);
                    mx = (uint32_t)(ix & 0xFFFF); my = (uint32_t)(iy & 0xFFFF); mz = (uint32_t)(iz & 0xFFFF);
                }
            }
        }
        if (sub == 0u) out[1 + c] = make_uint4(qx | (qy << 16), qz | ((uint32_t)rh << 16), mx | (my << 16), mz | ((uint32_t)wh << 16));
    }
    if (sub == 0u) out[5] = make_uint4(refs[0], refs[1], refs[2], refs[3]);
}

// The slab of one large child (WIDE_SLAB_SPLIT < triangles <= WIDE_SLAB_MAX): a block per listed (node, child).  Same construction as
// in k_wide_nodes — direction = normalised sum of the subtree's area normals, quantised; W measured against the DECODED centre and
// direction, rounded up — so the bound is conservative whatever the summation order; only the order of the fp64 sum differs.
__global__ void __launch_bounds__(256) k_wide_slabs_big(const uint4* __restrict__ bigList, uint32_t count, const float4* __restrict__ triV, const uint32_t* __restrict__ triAtRank,
                                                        uint4* __restrict__ wide) {
    __shared__ double s_a[3][4]; __shared__ double s_w[4];
    const uint4 job = bigList[blockIdx.x];
    if (blockIdx.x >= count) return;
    uint4* out = wide + 8 * (size_t)job.x;
    const uint4 hdr = out[0], rec = out[1 + job.y];
    const float ox = __uint_as_float(hdr.x), oy = __uint_as_float(hdr.y), oz = __uint_as_float(hdr.z), scale = __uint_as_float(hdr.w);
    const double dcx = (double)fmaf((float)(rec.x & 0xFFFFu), scale, ox), dcy = (double)fmaf((float)(rec.x >> 16), scale, oy), dcz = (double)fmaf((float)(rec.y & 0xFFFFu), scale, oz);
    const int tid = threadIdx.x, w = tid >> 6;
    double sx = 0, sy = 0, sz = 0;
    for (uint32_t k = job.z + (uint32_t)tid; k < job.w; k += 256u) {
        const uint32_t t = triAtRank[k];
        const float4 q0 = triV[3 * (size_t)t], q1 = triV[3 * (size_t)t + 1], q2 = triV[3 * (size_t)t + 2];
        const double ux = (double)q0.w - q0.x, uy = (double)q1.x - q0.y, uz = (double)q1.y - q0.z, vx = (double)q1.z - q0.x, vy = (double)q1.w - q0.y, vz = (double)q2.x - q0.z;
        sx += uy * vz - uz * vy; sy += uz * vx - ux * vz; sz += ux * vy - uy * vx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); sz += __shfl_xor(sz, o); }
    if ((tid & 63) == 0) { s_a[0][w] = sx; s_a[1][w] = sy; s_a[2][w] = sz; }
    __syncthreads();
    sx = s_a[0][0] + s_a[0][1] + s_a[0][2] + s_a[0][3]; sy = s_a[1][0] + s_a[1][1] + s_a[1][2] + s_a[1][3]; sz = s_a[2][0] + s_a[2][1] + s_a[2][2] + s_a[2][3];
    const double len = sqrt(sx * sx + sy * sy + sz * sz);
    if (!(len > 1e-300 && len < 1e300)) return;           // uniform: no slab, as k_wide_nodes leaves it
    auto snorm = [&](double v) { double q = rint(v / len * 32767.0); if (q < -32767.0) q = -32767.0; if (q > 32767.0) q = 32767.0; return (int)q; };
    const int ix = snorm(sx), iy = snorm(sy), iz = snorm(sz);
    const double dmx = (double)((float)ix * (1.0f / 32767.0f)), dmy = (double)((float)iy * (1.0f / 32767.0f)), dmz = (double)((float)iz * (1.0f / 32767.0f));
    double W = 0.0;
    for (uint32_t k = job.z + (uint32_t)tid; k < job.w; k += 256u) {
        const uint32_t t = triAtRank[k];
        const float4 q0 = triV[3 * (size_t)t], q1 = triV[3 * (size_t)t + 1], q2 = triV[3 * (size_t)t + 2];
        const double px[3] = {q0.x, q0.w, q1.z}, py[3] = {q0.y, q1.x, q1.w}, pz[3] = {q0.z, q1.y, q2.x};
        for (int j = 0; j < 3; j++) W = fmax(W, fabs(dmx * (px[j] - dcx) + dmy * (py[j] - dcy) + dmz * (pz[j] - dcz)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) W = fmax(W, __shfl_xor(W, o));
    if ((tid & 63) == 0) s_w[w] = W;
    __syncthreads();
    if (tid == 0) {
        W = fmax(fmax(s_w[0], s_w[1]), fmax(s_w[2], s_w[3]));
        const double Winfl = W * (1.0 + 1e-9) + 1e-300;
        float wf = (float)Winfl; if ((double)wf < Winfl) wf = nextafterf(wf, 3.0e38f);
        const unsigned short wh = halfRoundedUp(wf);
        out[1 + job.y] = make_uint4(rec.x, rec.y, (uint32_t)(ix & 0xFFFF) | ((uint32_t)(iy & 0xFFFF) << 16), (uint32_t)(iz & 0xFFFF) | ((uint32_t)wh << 16));
    }
}

__global__ void __launch_bounds__(128) k_nearest(BvhDev bvh, const float* __restrict__ pts, uint64_t n, uint32_t* __restrict__ out) {
    __shared__ uint32_t s_stack[BVH_STACK * 128];
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = bvhNearest<128>(bvh, F3{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]}, s_stack + threadIdx.x);
}

// dev probe: traversal statistics per query
__global__ void __launch_bounds__(128) k_nearest_stats(BvhDev bvh, const float* __restrict__ pts, uint64_t n, uint32_t* __restrict__ out4) {
    __shared__ uint32_t s_stack[BVH_STACK * 128];
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t c[3] = {0, 0, 0};
    out4[4 * i] = bvhNearest<128, true>(bvh, F3{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]}, s_stack + threadIdx.x, c);
    out4[4 * i + 1] = c[0]; out4[4 * i + 2] = c[1]; out4[4 * i + 3] = c[2];
}

__global__ void k_point_values(const float* __restrict__ verts, const uint32_t* __restrict__ idx, const float* __restrict__ td,
                               const float* __restrict__ pts, const uint32_t* __restrict__ tris, uint64_t n, float* __restrict__ out8) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = tris[i];
    const uint32_t a = idx[3 * t], b = idx[3 * t + 1], c = idx[3 * t + 2];
    F3 g;
    const float d = signedDistPointTriangleGrad(F3{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]}, td + (size_t)TD_FLOATS * t,
                                                F3{verts[3 * a], verts[3 * a + 1], verts[3 * a + 2]}, F3{verts[3 * b], verts[3 * b + 1], verts[3 * b + 2]},
                                                F3{verts[3 * c], verts[3 * c + 1], verts[3 * c + 2]}, g);
    float* o = out8 + 8 * i;
    o[0] = d; o[1] = g.x; o[2] = g.y; o[3] = g.z; o[4] = 0.f; o[5] = 0.f; o[6] = 0.f; o[7] = 0.f;
}

}  // namespace sdfhip

using namespace sdfhip;

int sdfhip_mesh_ensure_bvh(sdfhip_mesh* mesh) {
    SDF_API_BEGIN
    if (mesh->hasBvh) return SDFHIP_OK;
    return sdfhip_mesh_build_bvh(mesh, nullptr);
    SDF_API_END
}

// Upload a planned tree (8 doubles + 2 ints per inner node, host or device memory) and derive what the traversal needs besides:
// the fp32 copy of the spheres, the coordinate scale bounding its rounding, and the per-triangle vertex records.
// The two-phase nearest search navigates the tree WITHOUT loading it (k_tri_ranks, k_wide_nodes, resolveTies): it assumes what the planner
// produces — inner nodes numbered in pre-order, a node over the sorted range [b, e) split at (b + e) / 2, one-triangle ranges stored as
// ~triangle.  An imported tree of any other shape would silently give wrong ids, so an import is checked against that shape (host walk
// of the child array, O(T), overlapped with the device-side derivations) and refused otherwise.
static bool isPlannerShaped(const int* kids, uint32_t T) {
    if (T < 2) return true;
    std::vector<uint8_t> seen(T, 0);
    struct Item { uint32_t node, b, e; };
    std::vector<Item> stack; stack.push_back(Item{0u, 0u, T});
    uint64_t leaves = 0;
    while (!stack.empty()) {
        const Item it = stack.back(); stack.pop_back();
        if (it.node >= T - 1) return false;
        const uint32_t mid = (uint32_t)(0.5 * ((double)it.b + (double)it.e));
        const uint32_t ranges[2][2] = {{it.b, mid}, {mid, it.e}};
        const uint32_t expectInner[2] = {it.node + 1u, it.node + (mid - it.b)};
        for (int side = 0; side < 2; side++) {
            const int k = kids[2 * (size_t)it.node + side];
            const uint32_t n = ranges[side][1] - ranges[side][0];
            if (n == 1) {
                if (k >= 0) return false;
                const uint32_t t = (uint32_t)~k;
                if (t >= T || seen[t]) return false;
                seen[t] = 1; leaves++;
            } else {
                if (k < 0 || (uint32_t)k != expectInner[side]) return false;
                stack.push_back(Item{(uint32_t)k, ranges[side][0], ranges[side][1]});
            }
        }
    }
    return leaves == T;
}

// The device's share of a hybrid plan (planBvhHost with offload): the host-planned records go to their places, then one workgroup per
// listed range builds its subtree (k_bvh_subtrees).  SDFHIP_E_UNSUPPORTED: a sort ran into introsort's depth limit — the caller plans on the host.
struct PlannedBvh;
static int finishOnDevice(sdfhip_mesh* mesh, const PlannedBvh& P, hipStream_t st);

static int installBvh(sdfhip_mesh* mesh, const double* sph, const int* kids, int where, bool validate = false, const PlannedBvh* hybrid = nullptr) {
    const uint32_t T = mesh->numTriangles;
    const uint64_t nn = T - 1;
    const size_t nSph = 8 * (size_t)(nn ? nn : 1), nKids = 2 * (size_t)(nn ? nn : 1);
    SDF_HIP_CHECK(hipSetDevice(mesh->ctx->device));
    hipStream_t st = mesh->ctx->stream;
    AllocScope allocScope(st);
    const hipMemcpyKind kind = where == SDFHIP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    SDF_TRY(mesh->dBvhSph.reserve(nSph)); SDF_TRY(mesh->dBvhKids.reserve(nKids)); SDF_TRY(mesh->dTriVerts.reserve(12ull * T));
    if (!hybrid) {
        SDF_HIP_CHECK(hipMemcpyAsync(mesh->dBvhSph.p, sph, nSph * sizeof(double), kind, st));
        SDF_HIP_CHECK(hipMemcpyAsync(mesh->dBvhKids.p, kids, nKids * sizeof(int), kind, st));
    }
    k_tri_verts<<<gridFor(12ull * T, 256), 256, 0, st>>>(mesh->dVerts.p, mesh->dIdx.p, T, mesh->dTriVerts.p);
    if (hybrid) SDF_TRY(finishOnDevice(mesh, *hybrid, st));
    {
        float scale = 0.f;
        for (float c : mesh->hVerts) scale = std::max(scale, std::fabs(c));
        mesh->bvhCoordScale = scale;
        SDF_TRY(mesh->dBvhSph32.reserve(nSph));
        k_sph32<<<gridFor(nSph, 256), 256, 0, st>>>(mesh->dBvhSph.p, nSph, mesh->dBvhSph32.p);
    }
    SDF_TRY(mesh->dTriRank.reserve(T));
    k_tri_ranks<<<gridFor(T, 256), 256, 0, st>>>(reinterpret_cast<const int2*>(mesh->dBvhKids.p), T, mesh->dTriRank.p);
    DevBuf<uint32_t> triAtRank;
    SDF_TRY(triAtRank.reserve(T));
    k_tri_at_rank<<<gridFor(T, 256), 256, 0, st>>>(mesh->dTriRank.p, T, triAtRank.p);
    SDF_TRY(mesh->dBvhWide.reserve(32 * (size_t)(nn ? nn : 1)));
    DevBuf<uint32_t> bigList, bigCount;                  // children too large for 16 lanes: at most T / WIDE_SLAB_SPLIT per level pair, 12 level pairs
    const size_t bigCap = (size_t)T / WIDE_SLAB_SPLIT * 16 + 64;
    SDF_TRY(bigList.reserve(4 * bigCap)); SDF_TRY(bigCount.reserve(1));
    SDF_HIP_CHECK(hipMemsetAsync(bigCount.p, 0, 4, st));
    k_wide_nodes<<<gridFor(16ull * T, 256), 256, 0, st>>>(reinterpret_cast<const int2*>(mesh->dBvhKids.p), reinterpret_cast<const double2*>(mesh->dBvhSph.p), reinterpret_cast<const float4*>(mesh->dTriVerts.p),
                                               triAtRank.p, T, reinterpret_cast<uint4*>(mesh->dBvhWide.p), reinterpret_cast<uint4*>(bigList.p), bigCount.p, (uint32_t)bigCap);
    {
        uint32_t nBig = 0;
        SDF_HIP_CHECK(hipMemcpyAsync(&nBig, bigCount.p, 4, hipMemcpyDeviceToHost, st));
        SDF_HIP_CHECK(hipStreamSynchronize(st));
        SDF_REQUIRE(nBig <= bigCap, "wide-node work list overflow");
        if (nBig) k_wide_slabs_big<<<nBig, 256, 0, st>>>(reinterpret_cast<const uint4*>(bigList.p), nBig, reinterpret_cast<const float4*>(mesh->dTriVerts.p), triAtRank.p, reinterpret_cast<uint4*>(mesh->dBvhWide.p));
    }
    SDF_HIP_CHECK(hipGetLastError());
    bool shapeOk = true;
    if (validate) {              // while the device derives its records
        std::vector<int> hostKids;
        const int* hk = kids;
        if (where != SDFHIP_HOST) { hostKids.resize(nKids); SDF_HIP_CHECK(hipMemcpy(hostKids.data(), kids, nKids * sizeof(int), hipMemcpyDeviceToHost)); hk = hostKids.data(); }
        shapeOk = isPlannerShaped(hk, T);
    }
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    if (!shapeOk) {
        mesh->hasBvh = false;
        setError("imported BVH is not a midpoint-split tree numbered in pre-order (what sdfhip_mesh_bvh_export / the planner produce): refused");
        return SDFHIP_E_INVALID;
    }
    mesh->numBvhNodes = nn;
    mesh->hasBvh = true;
    return SDFHIP_OK;
}

extern "C" {

// The planner proper: host memory in, host memory out, no device involved.  sph / kids as in HostBvhBuilder (8 doubles + 2 ints per inner
// node; one dummy node for a one-triangle mesh).
// The planner's arrays — 8 doubles + 2 ints per node of output, 72 bytes per triangle of scratch: 200 MB at 1.31 M triangles — are 2 MB-aligned
// blocks with a huge-page hint, touched up front by a few threads (first touch of a fresh mapping by all the planner's workers at once was
// measured to stall single nodes for tens of milliseconds).  Allocating and faulting them in is a fifth of a plan's wall time, so blocks
// given back are kept (up to SDFHIP_PLANNER_CACHE_MB, default 512) and handed to the next plan as they are.
struct PlannerBlocks {
    std::mutex m; std::vector<std::pair<void*, size_t>> idle, live; size_t idleBytes = 0;
    static PlannerBlocks& get() { static PlannerBlocks* p = new PlannerBlocks(); return *p; }
    static size_t cap() { static const size_t v = (size_t)(getenv("SDFHIP_PLANNER_CACHE_MB") ? strtoull(getenv("SDFHIP_PLANNER_CACHE_MB"), nullptr, 10) : 512ull) << 20; return v; }
    void* take(size_t rounded) {
        std::lock_guard<std::mutex> g(m);
        size_t best = (size_t)-1;
        for (size_t i = 0; i < idle.size(); i++) if (idle[i].second >= rounded && idle[i].second <= 2 * rounded && (best == (size_t)-1 || idle[i].second < idle[best].second)) best = i;
        if (best == (size_t)-1) return nullptr;
        void* p = idle[best].first; live.push_back(idle[best]); idleBytes -= idle[best].second; idle[best] = idle.back(); idle.pop_back();
        return p;
    }
    void track(void* p, size_t bytes) { std::lock_guard<std::mutex> g(m); live.emplace_back(p, bytes); }
    void give(void* p) {
        size_t bytes = 0;
        {
            std::lock_guard<std::mutex> g(m);
            for (size_t i = 0; i < live.size(); i++) if (live[i].first == p) { bytes = live[i].second; live[i] = live.back(); live.pop_back(); break; }
            if (bytes && idleBytes + bytes <= cap()) { idle.emplace_back(p, bytes); idleBytes += bytes; return; }
        }
        free(p);
    }
};
struct FreeDeleter { void operator()(void* p) const { if (p) PlannerBlocks::get().give(p); } };
static void* plannerAlloc(size_t bytes) {
    const size_t rounded0 = (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
    const size_t rounded = rounded0 ? rounded0 : (2u << 20);
    if (void* cached = PlannerBlocks::get().take(rounded)) return cached;
    void* p = nullptr;
    if (posix_memalign(&p, 2u << 20, rounded) != 0) throw std::bad_alloc();
    static const bool noThp = getenv("SDFHIP_BVH_NO_THP") != nullptr;
    if (!noThp) madvise(p, rounded, MADV_HUGEPAGE);
    int parts = (int)std::min<size_t>(16, rounded >> 21); if (parts < 1) parts = 1;
    PlannerPool::get().run(parts, [&](int c) {
        char* q = (char*)p;
        for (size_t off = (rounded * (size_t)c / (size_t)parts) & ~(size_t)4095, e = rounded * (size_t)(c + 1) / (size_t)parts; off < e; off += 4096) q[off] = 0;
    });
    PlannerBlocks::get().track(p, rounded);
    return p;
}
struct PlannedBvh {
    std::unique_ptr<double, FreeDeleter> sph; std::unique_ptr<int, FreeDeleter> kids; double gatherSeconds = 0, planSeconds = 0; int sortThreads = 0, parallelDepth = 0;
    std::vector<BvhTask> tasks; std::vector<int> hostNodes; std::vector<int> order;       // offload only: what is left to the device, what was planned here, the triangle order so far
};
// offloadMax > 0: ranges of at most that many triangles are left to the device (k_bvh_subtrees); the arrays then hold the top of the tree only
static PlannedBvh planBvhHost(const float* hVerts, const uint32_t* hIdx, uint32_t T, uint32_t offloadMax = 0) {
    PlannedBvh R;
    if (T <= offloadMax) offloadMax = 0;
    const double t0 = nowSeconds();
    const uint64_t nn = T - 1;                                   // inner nodes
    const size_t nSph = 8 * (size_t)(nn ? nn : 1), nKids = 2 * (size_t)(nn ? nn : 1);
    R.sph.reset((double*)plannerAlloc(8 * nSph));
    R.kids.reset((int*)plannerAlloc(4 * nKids));
    if (nn == 0) { for (size_t i = 0; i < nSph; i++) R.sph.get()[i] = 0.0; R.kids.get()[0] = R.kids.get()[1] = ~0; }
    std::unique_ptr<float, FreeDeleter> htvBuf((float*)plannerAlloc(36 * (size_t)T));
    float* htv = htvBuf.get();
    HostBvhBuilder::parallelFor((int)T, [&](int t0, int t1) {
        for (size_t t = (size_t)t0; t < (size_t)t1; t++) for (int k = 0; k < 3; k++) {
            const uint32_t v = hIdx[3 * t + k];
            htv[9 * t + 3 * k] = hVerts[3 * (size_t)v]; htv[9 * t + 3 * k + 1] = hVerts[3 * (size_t)v + 1]; htv[9 * t + 3 * k + 2] = hVerts[3 * (size_t)v + 2];
        }
    });
    const double tGather = nowSeconds();
    HostBvhBuilder b;
    b.verts = hVerts; b.idx = hIdx; b.sph = R.sph.get(); b.kids = R.kids.get(); b.triV = htv;
    std::unique_ptr<KeyTri, FreeDeleter> sk((KeyTri*)plannerAlloc(sizeof(KeyTri) * (size_t)T)); std::unique_ptr<float, FreeDeleter> sl((float*)plannerAlloc(36 * (size_t)T));
    std::unique_ptr<uint32_t, FreeDeleter> sL((uint32_t*)plannerAlloc(4 * (size_t)T)), sR((uint32_t*)plannerAlloc(4 * (size_t)T));
    b.scratchKeys = sk.get(); b.scratchLoc = sl.get(); b.scratchL = sL.get(); b.scratchR = sR.get();
    b.order.resize(T);
    for (uint32_t i = 0; i < T; i++) b.order[i] = (int)i;
    unsigned hc = std::thread::hardware_concurrency();
    int pd = 0; while ((1u << pd) < (hc ? hc : 1u) && pd < 8) pd++;
    b.maxParallelDepth = pd;
    b.sortThreads = (int)(hc ? hc : 1u);
    if (getenv("SDFHIP_BVH_PAR_DEPTH")) b.maxParallelDepth = atoi(getenv("SDFHIP_BVH_PAR_DEPTH"));
    if (getenv("SDFHIP_BVH_SORT_THREADS")) b.sortThreads = atoi(getenv("SDFHIP_BVH_SORT_THREADS"));
    double rootSphere[4];
    b.offloadMax = offloadMax;
    const double tBuild = nowSeconds();
    b.build(0, rootSphere, 0, (int)T, 0);
    if (getenv("SDFHIP_TIMING")) fprintf(stderr, "[sdfhip] bvh plan: output + gather %.4f s, scratch %.4f s, tree %.4f s\n", tGather - t0, tBuild - tGather, nowSeconds() - tBuild);
    if (offloadMax) { R.tasks = std::move(b.tasks); R.hostNodes = std::move(b.hostNodes); R.order = std::move(b.order); }
    R.gatherSeconds = tGather - t0; R.planSeconds = nowSeconds() - tGather; R.sortThreads = b.sortThreads; R.parallelDepth = b.maxParallelDepth;
    return R;
}

}  // extern "C"

static uint32_t bvhOffloadMax() {
    // SDFHIP_BVH_DEVICE_SUBTREES=1: ranges of at most 4096 triangles are built on the device (k_bvh_subtrees).  Off by default — measured on
    // the 16-CPU box it does not shorten the build: the planner's wall time is the critical path through the TOP levels (the root's sort,
    // then its children's ...), the bottom levels already run on otherwise idle pool threads, and the subtree kernel adds 7 ms.
    static const uint32_t offload = [] {
        const char* e = getenv("SDFHIP_BVH_DEVICE_SUBTREES");
        if (!e) return 0u;
        uint32_t v = (uint32_t)atoi(e); if (v <= 1u) v = 4096u;                 // =1: the default size; =N: ranges of at most N triangles
        return v > kDevSubtreeMaxLimit ? kDevSubtreeMaxLimit : (v < 32u ? 32u : v);
    }();
    return offload;
}
namespace sdfhip {
void startEarlyBvhPlan(sdfhip_mesh* mesh) {
    if (mesh->early.th.joinable() || mesh->early.plan) return;
    mesh->early.drop = [](void* p) { delete static_cast<PlannedBvh*>(p); };
    mesh->early.th = std::thread([mesh]() {
        try { mesh->early.plan = new PlannedBvh(planBvhHost(mesh->hVerts.data(), mesh->hIdx.data(), mesh->numTriangles, bvhOffloadMax())); }
        catch (...) { mesh->early.plan = nullptr; }          // (out of memory: sdfhip_mesh_build_bvh plans again and reports)
    });
}
}

static int finishOnDevice(sdfhip_mesh* mesh, const PlannedBvh& P, hipStream_t st) {
    const uint32_t T = mesh->numTriangles;
    const uint32_t kDevSubtreeMax = bvhOffloadMax();
    const size_t nt = P.tasks.size(), nh = P.hostNodes.size();
    // the host's records, compacted: ids, 8 doubles, 2 child references, and per half whether the host owns that sphere
    std::vector<double> s8(8 * nh); std::vector<int> k2(2 * nh); std::vector<unsigned char> own(2 * nh, 1);
    {
        std::vector<uint32_t> taskSlots(nt);
        for (size_t i = 0; i < nt; i++) taskSlots[i] = P.tasks[i].parentSlot;
        std::sort(taskSlots.begin(), taskSlots.end());
        for (size_t i = 0; i < nh; i++) {
            const int id = P.hostNodes[i];
            memcpy(&s8[8 * i], P.sph.get() + 8 * (size_t)id, 64); k2[2 * i] = P.kids.get()[2 * (size_t)id]; k2[2 * i + 1] = P.kids.get()[2 * (size_t)id + 1];
            for (int h = 0; h < 2; h++) if (std::binary_search(taskSlots.begin(), taskSlots.end(), 2u * (uint32_t)id + (uint32_t)h)) own[2 * i + h] = 0;
        }
    }
    DevBuf<int> dIds, dK2; DevBuf<double> dS8; DevBuf<unsigned char> dOwn; DevBuf<uint32_t> dOrder, dFail; DevBuf<BvhTask> dTasks; DevBuf<BvhDevNode> dScratch;
    SDF_TRY(dIds.reserve(nh)); SDF_TRY(dK2.reserve(2 * nh)); SDF_TRY(dS8.reserve(8 * nh)); SDF_TRY(dOwn.reserve(2 * nh));
    SDF_TRY(dOrder.reserve(T)); SDF_TRY(dFail.reserve(1)); SDF_TRY(dTasks.reserve(nt)); SDF_TRY(dScratch.reserve(nt * 2 * (kDevSubtreeMax / 2 + 1)));
    SDF_HIP_CHECK(hipMemcpyAsync(dIds.p, P.hostNodes.data(), 4 * nh, hipMemcpyHostToDevice, st));
    SDF_HIP_CHECK(hipMemcpyAsync(dK2.p, k2.data(), 8 * nh, hipMemcpyHostToDevice, st));
    SDF_HIP_CHECK(hipMemcpyAsync(dS8.p, s8.data(), 64 * nh, hipMemcpyHostToDevice, st));
    SDF_HIP_CHECK(hipMemcpyAsync(dOwn.p, own.data(), 2 * nh, hipMemcpyHostToDevice, st));
    SDF_HIP_CHECK(hipMemcpyAsync(dOrder.p, P.order.data(), 4 * (size_t)T, hipMemcpyHostToDevice, st));
    SDF_HIP_CHECK(hipMemcpyAsync(dTasks.p, P.tasks.data(), sizeof(BvhTask) * nt, hipMemcpyHostToDevice, st));
    SDF_HIP_CHECK(hipMemsetAsync(dFail.p, 0, 4, st));
    k_bvh_scatter_top<<<gridFor(nh, 256), 256, 0, st>>>(dIds.p, dS8.p, dK2.p, dOwn.p, (uint32_t)nh, mesh->dBvhSph.p, mesh->dBvhKids.p);
    const size_t lds = sizeof(KeyTri) * kDevSubtreeMax + 2 * 3 * 4 * kDevSortTasks;
    static bool ldsRaised = false;
    if (!ldsRaised && lds > (48u << 10)) { SDF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bvh_subtrees), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(KeyTri) * kDevSubtreeMaxLimit + 2 * 3 * 4 * kDevSortTasks))); ldsRaised = true; }
    k_bvh_subtrees<<<(unsigned)nt, 256, lds, st>>>(dTasks.p, dOrder.p, reinterpret_cast<const float4*>(mesh->dTriVerts.p), mesh->dBvhSph.p, mesh->dBvhKids.p, dScratch.p, dFail.p, kDevSubtreeMax);
    SDF_HIP_CHECK(hipGetLastError());
    uint32_t failed = 0;
    SDF_HIP_CHECK(hipMemcpyAsync(&failed, dFail.p, 4, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));         // (the staging vectors above are released here)
    if (failed && getenv("SDFHIP_TIMING")) fprintf(stderr, "[sdfhip] bvh: device subtrees gave up (reason bits %u: 1 / 4 = sort task list full, 2 = introsort depth limit): planning on the host\n", failed);
    return failed ? SDFHIP_E_UNSUPPORTED : SDFHIP_OK;
}

extern "C" {

int sdfhip_mesh_build_bvh(sdfhip_mesh* mesh, double* seconds) {
    SDF_API_BEGIN
    SDF_REQUIRE(mesh != nullptr, "mesh is NULL");
    std::lock_guard<std::recursive_mutex> building(mesh->ctx->buildLock);
    if (mesh->hasBvh) { if (seconds) *seconds = 0.0; return SDFHIP_OK; }
    { int depth = 1; while ((1ull << (depth - 1)) < mesh->numTriangles) depth++; SDF_REQUIRE(depth + 1 <= BVH_STACK, "mesh too large for the traversal stack"); }
    const double t0 = nowSeconds();
    const uint32_t offload = bvhOffloadMax();
    PlannedBvh P;
    if (mesh->early.th.joinable()) mesh->early.th.join();        // a plan started under the mesh preparation (sdfhip_mesh_create_opt)
    if (mesh->early.plan) { P = std::move(*static_cast<PlannedBvh*>(mesh->early.plan)); delete static_cast<PlannedBvh*>(mesh->early.plan); mesh->early.plan = nullptr; }
    else P = planBvhHost(mesh->hVerts.data(), mesh->hIdx.data(), mesh->numTriangles, offload);
    double tPlanned = nowSeconds();
    int rc = installBvh(mesh, P.sph.get(), P.kids.get(), SDFHIP_HOST, false, P.tasks.empty() ? nullptr : &P);
    if (rc == SDFHIP_E_UNSUPPORTED && !P.tasks.empty()) {         // a device sort met introsort's depth limit: libstdc++'s heap sort decides that order
        P = planBvhHost(mesh->hVerts.data(), mesh->hIdx.data(), mesh->numTriangles, 0);
        tPlanned = nowSeconds();
        rc = installBvh(mesh, P.sph.get(), P.kids.get(), SDFHIP_HOST);
    }
    SDF_TRY(rc);
    if (getenv("SDFHIP_TIMING")) fprintf(stderr, "[sdfhip] bvh: gather %.3f s, planner %.3f s (%d sort threads, parallel depth %d), upload + device prep %.3f s\n", P.gatherSeconds, P.planSeconds, P.sortThreads, P.parallelDepth, nowSeconds() - tPlanned);
    if (seconds) *seconds = nowSeconds() - t0;
    return SDFHIP_OK;
    SDF_API_END
}

// Test hook (no GPU needed): heap-sorts n {key, id} pairs with the restated libstdc++ heap sort (the device subtrees' fallback at introsort's
// depth limit) and with std::make_heap + std::sort_heap; returns the number of positions where the permutations differ.
int sdfhip_test_heap_sort_matches_std(const double* keys, uint64_t n) {
    SDF_API_BEGIN
    if (!keys) return -1;
    std::vector<KeyTri> a(n), b(n);
    for (uint64_t i = 0; i < n; i++) a[i] = b[i] = KeyTri{(float)keys[i], (int)i};
    stdHeapSort(a.data(), (int)n);
    std::make_heap(b.begin(), b.end(), keyLess); std::sort_heap(b.begin(), b.end(), keyLess);
    int diff = 0;
    for (uint64_t i = 0; i < n; i++) diff += (a[i].tri != b[i].tri || a[i].key != b[i].key) ? 1 : 0;
    return diff;
    SDF_API_END
}

// Test hook (no GPU needed): plans the tree of a mesh given in host memory.  out_spheres: 8 doubles, out_children: 2 ints per inner node
// (max(T - 1, 1) of them).  The vertices must be finite and the indices in range (sdfhip_mesh_create checks that for real meshes).
int sdfhip_test_plan_bvh(const float* xyz, uint32_t num_vertices, const uint32_t* indices, uint32_t num_triangles, double* out_spheres, int32_t* out_children, double* seconds) {
    SDF_API_BEGIN
    SDF_REQUIRE(xyz && indices && out_spheres && out_children && num_triangles >= 1, "bad argument");
    for (size_t i = 0; i < 3 * (size_t)num_triangles; i++) SDF_REQUIRE(indices[i] < num_vertices, "index out of range");
    const double t0 = nowSeconds();
    PlannedBvh P = planBvhHost(xyz, indices, num_triangles);
    if (seconds) *seconds = nowSeconds() - t0;
    const size_t nn = num_triangles > 1 ? num_triangles - 1 : 1;
    memcpy(out_spheres, P.sph.get(), 64 * nn); memcpy(out_children, P.kids.get(), 8 * nn);
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_mesh_bvh_export(sdfhip_mesh* mesh, double* out_spheres, int32_t* out_children, int where) {
    SDF_API_BEGIN
    SDF_REQUIRE(mesh && out_spheres && out_children, "NULL argument");
    SDF_TRY(sdfhip_mesh_ensure_bvh(mesh));
    const uint64_t nn = mesh->numTriangles - 1;
    const size_t nSph = 8 * (size_t)(nn ? nn : 1), nKids = 2 * (size_t)(nn ? nn : 1);
    SDF_HIP_CHECK(hipSetDevice(mesh->ctx->device));
    hipStream_t st = mesh->ctx->stream;
    const hipMemcpyKind kind = where == SDFHIP_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    SDF_HIP_CHECK(hipMemcpyAsync(out_spheres, mesh->dBvhSph.p, nSph * sizeof(double), kind, st));
    SDF_HIP_CHECK(hipMemcpyAsync(out_children, mesh->dBvhKids.p, nKids * sizeof(int), kind, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_mesh_bvh_import(sdfhip_mesh* mesh, const double* spheres, const int32_t* children, int where) {
    SDF_API_BEGIN
    SDF_REQUIRE(mesh && spheres && children, "NULL argument");
    std::lock_guard<std::recursive_mutex> building(mesh->ctx->buildLock);
    { int depth = 1; while ((1ull << (depth - 1)) < mesh->numTriangles) depth++; SDF_REQUIRE(depth + 1 <= BVH_STACK, "mesh too large for the traversal stack"); }
    return installBvh(mesh, spheres, children, where, true);
    SDF_API_END
}

int sdfhip_mesh_nearest(sdfhip_mesh* mesh, const float* xyz, uint64_t n, uint32_t* out_ids, int where) {
    SDF_API_BEGIN
    SDF_REQUIRE(mesh && xyz && out_ids, "NULL argument");
    SDF_TRY(sdfhip_mesh_ensure_bvh(mesh));
    if (n == 0) return SDFHIP_OK;
    hipStream_t st = mesh->ctx->stream;
    SDF_HIP_CHECK(hipSetDevice(mesh->ctx->device));
    DevBuf<float> dp; DevBuf<uint32_t> dout;
    const float* p = xyz; uint32_t* o = out_ids;
    if (where == SDFHIP_HOST) {
        SDF_TRY(dp.reserve(3 * n)); SDF_TRY(dout.reserve(n));
        SDF_HIP_CHECK(hipMemcpyAsync(dp.p, xyz, sizeof(float) * 3 * n, hipMemcpyHostToDevice, st));
        p = dp.p; o = dout.p;
    }
    if (nearestExactOnly()) k_nearest<<<gridFor(n, 128), 128, 0, st>>>(meshBvh(mesh), p, n, o);
    else {
        std::lock_guard<std::recursive_mutex> building(mesh->ctx->buildLock);      // the context's scratch is shared with the builders
        int depth = 1; while ((1ull << (depth - 1)) < mesh->numTriangles) depth++;
        // the search keeps ~140 B of candidate lists per query in the context's scratch: batches go through in pieces of 4 M queries
        // (0.6 GB), whatever their size
        constexpr uint64_t kPiece = 1ull << 22;
        for (uint64_t off = 0; off < n; off += kPiece) {
            const uint64_t m = n - off < kPiece ? n - off : kPiece;
            SDF_TRY(nearestTwoPhase(st, meshBvh(mesh), p + 3 * off, (uint32_t)m, o + off, mesh->ctx->nearScratch, depth + 2, 0u, 1u));
        }
        SDF_HIP_CHECK(hipStreamSynchronize(st));
    }
    SDF_HIP_CHECK(hipGetLastError());
    if (where == SDFHIP_HOST) SDF_HIP_CHECK(hipMemcpyAsync(out_ids, dout.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    return SDFHIP_OK;
    SDF_API_END
}

// Test hook (no GPU needed): sorts n {key, id} pairs with the planner's multi-threaded introsort and with std::sort and
// returns the number of positions where the two permutations differ (0 = identical).
int sdfhip_test_sort_matches_std(const double* keys, uint64_t n, int threads) {
    SDF_API_BEGIN
    if (!keys) return -1;
    std::vector<KeyTri> a(n), b(n);
    for (uint64_t i = 0; i < n; i++) a[i] = b[i] = KeyTri{(float)keys[i], (int)i};
    IntroSortLike s; s.maxThreads = threads; s.minParallel = 64; s.minParPartition = 200;          // small thresholds: exercise the threaded paths
    std::vector<uint32_t> sl(n + 1), sr(n + 1); s.scratchL = sl.data(); s.scratchR = sr.data();
    s.sort(a.data(), a.data() + n);
    std::sort(b.begin(), b.end(), keyLess);
    int diff = 0;
    for (uint64_t i = 0; i < n; i++) diff += (a[i].tri != b[i].tri || a[i].key != b[i].key) ? 1 : 0;
    return diff;
    SDF_API_END
}

int sdfhip_mesh_nearest_stats(sdfhip_mesh* mesh, const float* xyz, uint64_t n, uint32_t* out4) {
    SDF_API_BEGIN
    SDF_REQUIRE(mesh && xyz && out4, "NULL argument");
    SDF_TRY(sdfhip_mesh_ensure_bvh(mesh));
    hipStream_t st = mesh->ctx->stream;
    DevBuf<float> dp; DevBuf<uint32_t> dout;
    SDF_TRY(dp.reserve(3 * n)); SDF_TRY(dout.reserve(4 * n));
    SDF_HIP_CHECK(hipMemcpyAsync(dp.p, xyz, sizeof(float) * 3 * n, hipMemcpyHostToDevice, st));
    k_nearest_stats<<<gridFor(n, 128), 128, 0, st>>>(meshBvh(mesh), dp.p, n, dout.p);
    SDF_HIP_CHECK(hipGetLastError());
    SDF_HIP_CHECK(hipMemcpyAsync(out4, dout.p, sizeof(uint32_t) * 4 * n, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_mesh_point_values(sdfhip_mesh* mesh, const float* xyz, const uint32_t* tri_ids, uint64_t n, float* out8, int where) {
    SDF_API_BEGIN
    SDF_REQUIRE(mesh && xyz && tri_ids && out8, "NULL argument");
    if (n == 0) return SDFHIP_OK;
    hipStream_t st = mesh->ctx->stream;
    SDF_HIP_CHECK(hipSetDevice(mesh->ctx->device));
    DevBuf<float> dp, dout; DevBuf<uint32_t> dt;
    const float* p = xyz; const uint32_t* t = tri_ids; float* o = out8;
    if (where == SDFHIP_HOST) {
        SDF_TRY(dp.reserve(3 * n)); SDF_TRY(dt.reserve(n)); SDF_TRY(dout.reserve(8 * n));
        SDF_HIP_CHECK(hipMemcpyAsync(dp.p, xyz, sizeof(float) * 3 * n, hipMemcpyHostToDevice, st));
        SDF_HIP_CHECK(hipMemcpyAsync(dt.p, tri_ids, sizeof(uint32_t) * n, hipMemcpyHostToDevice, st));
        p = dp.p; t = dt.p; o = dout.p;
    }
    k_point_values<<<gridFor(n, 128), 128, 0, st>>>(mesh->dVerts.p, mesh->dIdx.p, mesh->dTri.p, p, t, n, o);
    SDF_HIP_CHECK(hipGetLastError());
    if (where == SDFHIP_HOST) SDF_HIP_CHECK(hipMemcpyAsync(out8, dout.p, sizeof(float) * 8 * n, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    return SDFHIP_OK;
    SDF_API_END
}

}  // extern "C"
