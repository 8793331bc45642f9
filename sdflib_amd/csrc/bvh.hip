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
constexpr uint32_t kDevSubtreeMaxLimit = 8192;           // 64 KB of LDS for the keys + the sort task lists + the per-node tables below
constexpr uint32_t kSumChunk = 32;          // triangles per chunk: 96 additions per chain and lane
constexpr uint32_t kCoopNodes = 128;                     // levels of at most this many nodes ...
constexpr uint32_t kCoopMin = 32;                        // ... of more than this many triangles each are prepared by the whole workgroup
SDF_DEV uint32_t devOrdKey(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }      // monotone float -> uint
SDF_DEV float devOrdVal(uint32_t k) { return __uint_as_float(k ^ (0x80000000u | ~(uint32_t)((int32_t)k >> 31))); }      // (top bit set: clear it; else: all bits flipped)
constexpr uint32_t kDevSortTasks = 512;           // (with the tables below: 72 KB at 4096 triangles, two workgroups per CU)

// LDS views with a skew: the ranges a workgroup's lanes work on start at regular strides (a level's nodes are 2560, 1280, ... 80, 40, 20
// triangles apart), which without it puts every lane of a wave on the same two banks (measured: the sort rounds were 32-way conflicts).
struct KeyArr {                      // 8-byte elements: one element of padding per 16
    KeyTri* p; int base;
    SDF_HD KeyTri& operator[](int i) const { const int j = base + i; return p[j + (j >> 4)]; }
    SDF_HD KeyArr operator+(int o) const { return KeyArr{p, base + o}; }
    static constexpr size_t bytes(uint32_t n) { return sizeof(KeyTri) * ((size_t)n + (n >> 4) + 1); }
};
struct IdxArr {                      // 2-byte elements: two elements of padding per 64
    unsigned short* p;
    SDF_HD unsigned short& operator[](int i) const { return p[i + 2 * (i >> 6)]; }
    static constexpr size_t bytes(uint32_t n) { return (2 * ((size_t)n + 2 * (n >> 6) + 2) + 7) & ~(size_t)7; }
};
template <class A> SDF_DEV void devInsertionSort(A a, int first, int last) {
    for (int i = first + 1; i < last; i++) {
        const KeyTri val = a[i];
        if (val.key < a[first].key) { for (int k = i; k > first; k--) a[k] = a[k - 1]; a[first] = val; }
        else { int pos = i; while (val.key < a[pos - 1].key) { a[pos] = a[pos - 1]; pos--; } a[pos] = val; }
    }
}
template <class A> SDF_DEV void devSwap(A a, int i, int j) { const KeyTri t = a[i]; a[i] = a[j]; a[j] = t; }
// libstdc++'s heap sort (what std::sort falls back to when a range exhausts introsort's depth limit: __partial_sort(first, last, last) =
// __make_heap + __sort_heap), restated move for move: __push_heap, __adjust_heap, __pop_heap (bits/stl_heap.h).  Compiled for the host too:
// sdfhip_test_heap_sort_matches_std compares it with std::make_heap / std::sort_heap on tie-heavy keys.
template <class A> SDF_HD void stdPushHeap(A first, int holeIndex, int topIndex, KeyTri value) {
    int parent = (holeIndex - 1) / 2;
    while (holeIndex > topIndex && first[parent].key < value.key) { first[holeIndex] = first[parent]; holeIndex = parent; parent = (holeIndex - 1) / 2; }
    first[holeIndex] = value;
}
template <class A> SDF_HD void stdAdjustHeap(A first, int holeIndex, int len, KeyTri value) {
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
template <class A> SDF_HD void stdHeapSort(A first, int len) {
    if (len >= 2) for (int parent = (len - 2) / 2;; parent--) { stdAdjustHeap(first, parent, len, first[parent]); if (parent == 0) break; }      // __make_heap
    for (int last = len; last > 1;) { --last; const KeyTri value = first[last]; first[last] = first[0]; stdAdjustHeap(first, 0, last, value); }   // __sort_heap / __pop_heap
}
// libstdc++'s __move_median_to_first(result, a, b, c)
template <class A> SDF_DEV void devMedianToFirst(A k, int result, int a, int b, int c) {
    if (k[a].key < k[b].key) {
        if (k[b].key < k[c].key) devSwap(k, result, b);
        else if (k[a].key < k[c].key) devSwap(k, result, c);
        else devSwap(k, result, a);
    } else if (k[a].key < k[c].key) devSwap(k, result, a);
    else if (k[b].key < k[c].key) devSwap(k, result, c);
    else devSwap(k, result, b);
}
// libstdc++'s __unguarded_partition(first, last, pivot)
SDF_DEV int devPartition(KeyArr k, int first, int last, int pivot) {
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
// libstdc++'s __unguarded_partition(first, last, pivot) by its result instead of its scans (the construction of IntroSortLike::listPartition):
// the t-th swap exchanges the t-th element from the left that is not less than the pivot with the t-th from the right that is not greater,
// while they have not crossed.  One lane, no data-dependent branch in the pass over the keys; L / R: the lane's slices of the index lists.
SDF_DEV int devListPartition(KeyArr k, IdxArr L, IdxArr R, int first, int last, int pivot) {
    const float pk = k[pivot].key;
    const int n = last - first;
    int nl = 0, nr = 0;
    for (int i = 0; i < n; i++) { const float key = k[first + i].key; L[first + nl] = (unsigned short)i; nl += !(key < pk); R[first + nr] = (unsigned short)i; nr += !(pk < key); }
    int lo = 0, hi = nl < nr ? nl : nr;
    while (lo < hi) { const int t = (lo + hi + 1) >> 1; if (L[first + t - 1] < R[first + nr - t]) lo = t; else hi = t - 1; }
    const int m = lo;
    for (int t = 1; t <= m; t++) devSwap(k, first + (int)L[first + t - 1], first + (int)R[first + nr - t]);
    int stop = (m < nl) ? (int)L[first + m] : n;
    if (m > 0 && (int)R[first + nr - m] < stop) stop = (int)R[first + nr - m];
    return first + stop;
}
SDF_DEV void devWaveSync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
constexpr int kWaveSortLen = 160;        // pending ranges from this length on are partitioned by a whole wave
// The introsort rounds of a workgroup (256 threads) over keys[] in LDS: `in` holds nSort pending ranges {first, last, depth limit}; every
// round partitions each of them once — a long range by a whole wave (both index lists by ballots, 64 keys per step; the swaps in
// parallel), a short one by sixteen lanes — and lists the parts for the next round; parts of at most 16 are finished by insertion sort (by
// rank), a range out of depth by libstdc++'s heap sort.  Every comparison and exchange is the sequential algorithm's, so is the order among ties.
// s_count[1] = next round's count, s_count[2] |= 4 when the lists (maxTasks entries) overflow.
SDF_DEV void devSortRounds(KeyArr keys, IdxArr listL, IdxArr listR, uint32_t* in, uint32_t* out, uint32_t* s_count, uint32_t nSort, uint32_t maxTasks, int tid, uint32_t nThreads = 256u) {
    const int lane = tid & 63, wave = tid >> 6;
    const unsigned long long ltMask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    auto push = [&](int f, int l, int depth) {
        const uint32_t at = atomicAdd(&s_count[1], 1u);
        if (at < maxTasks) { out[3 * at] = (uint32_t)f; out[3 * at + 1] = (uint32_t)l; out[3 * at + 2] = (uint32_t)depth; }
        else atomicOr(&s_count[2], 4u);
    };
    while (nSort > 0) {
        if (tid == 0) s_count[1] = 0;
        __syncthreads();
        // long ranges: one wave each
        for (uint32_t t = (uint32_t)wave; t < nSort; t += nThreads >> 6) {
            const int first = (int)in[3 * t], last = (int)in[3 * t + 1], depth = (int)in[3 * t + 2];
            if (last - first < kWaveSortLen) continue;
            if (depth == 0) { if (lane == 0) stdHeapSort(keys + first, last - first); continue; }
            if (lane == 0) devMedianToFirst(keys, first, first + 1, first + (last - first) / 2, last - 1);
            devWaveSync();
            const float pk = keys[first].key;
            const int f = first + 1, m = last - f;
            int nl = 0, nr = 0;
            for (int base = 0; base < m; base += 64) {
                const int i = base + lane;
                const bool valid = i < m;
                const float key = valid ? keys[f + i].key : 0.f;
                const bool pl = valid && !(key < pk), pr = valid && !(pk < key);
                const unsigned long long bl = __ballot(pl), br = __ballot(pr);
                if (pl) listL[f + nl + (int)__popcll(bl & ltMask)] = (unsigned short)i;
                if (pr) listR[f + nr + (int)__popcll(br & ltMask)] = (unsigned short)i;
                nl += (int)__popcll(bl); nr += (int)__popcll(br);
            }
            devWaveSync();
            int lo = 0, hi = nl < nr ? nl : nr;
            while (lo < hi) { const int q = (lo + hi + 1) >> 1; if (listL[f + q - 1] < listR[f + nr - q]) lo = q; else hi = q - 1; }
            const int ms = lo;
            int stop = (ms < nl) ? (int)listL[f + ms] : m;
            if (ms > 0 && (int)listR[f + nr - ms] < stop) stop = (int)listR[f + nr - ms];
            for (int q = 1 + lane; q <= ms; q += 64) devSwap(keys, f + (int)listL[f + q - 1], f + (int)listR[f + nr - q]);
            const int cut = f + stop;
            if (lane == 0) {
                if (cut - first > 1) push(first, cut, depth - 1);
                if (last - cut > 1) push(cut, last, depth - 1);
            }
        }
        // short ranges: sixteen lanes each (four ranges per wave at a time; a lane of its own per range was a serial chain of ~450 cycles per
        // element and step).  Partition as above with the group's 16 bits of the ballots; a part of at most 16 — what introsort leaves to its
        // final insertion sort, a STABLE sort — is placed by rank: element i goes behind the elements before it that are not greater and the
        // elements after it that are less (the same arrangement; keys that are not ordered at all, NaN, take the literal insertion sort).
        {
            const int gl = lane & 15, gshift = lane & 48;
            auto groupBits = [&](unsigned long long b) { return (unsigned)((b >> gshift) & 0xFFFFull); };
            const unsigned below = (1u << gl) - 1u;
            auto rankSort = [&](int first, int last) {                 // 2 .. 16 elements, all lanes of the group
                const int len = last - first;
                if (len < 2) return;
                const bool mine = gl < len;
                const KeyTri e = mine ? (KeyTri)keys[first + gl] : KeyTri{0.f, 0};
                const bool unordered = groupBits(__ballot(mine && e.key != e.key)) != 0u;
                int rank = 0;
                for (int j = 0; j < len; j++) { const float kj = keys[first + j].key; rank += (j < gl) ? !(e.key < kj) : (kj < e.key); }
                devWaveSync();
                if (unordered) { if (gl == 0) devInsertionSort(keys, first, last); }
                else if (mine) keys[first + rank] = e;
                devWaveSync();
            };
            for (uint32_t t = (uint32_t)(tid >> 4); t < nSort; t += nThreads >> 4) {
                const int first = (int)in[3 * t], last = (int)in[3 * t + 1], depth = (int)in[3 * t + 2];
                if (last - first >= kWaveSortLen) continue;
                if (last - first <= 16) { rankSort(first, last); continue; }
                if (depth == 0) { if (gl == 0) stdHeapSort(keys + first, last - first); continue; }        // libstdc++ switches to heap sort here (it does happen: 1.31 M triangles)
                if (gl == 0) devMedianToFirst(keys, first, first + 1, first + (last - first) / 2, last - 1);
                devWaveSync();
                const float pk = keys[first].key;
                const int f = first + 1, m = last - f;
                int nl = 0, nr = 0;
                for (int base = 0; base < m; base += 16) {
                    const int i = base + gl;
                    const bool valid = i < m;
                    const float key = valid ? keys[f + i].key : 0.f;
                    const bool pl = valid && !(key < pk), pr = valid && !(pk < key);
                    const unsigned bl = groupBits(__ballot(pl)), br = groupBits(__ballot(pr));
                    if (pl) listL[f + nl + (int)__popc(bl & below)] = (unsigned short)i;
                    if (pr) listR[f + nr + (int)__popc(br & below)] = (unsigned short)i;
                    nl += (int)__popc(bl); nr += (int)__popc(br);
                }
                devWaveSync();
                const int lim = nl < nr ? nl : nr;
                int ms = 0;                                           // pairs that have not crossed: a prefix of 1 .. lim
                for (int base = 0; base < lim; base += 16) {
                    const int q = base + gl + 1;
                    const bool go = q <= lim && listL[f + q - 1] < listR[f + nr - q];
                    ms += (int)__popc(groupBits(__ballot(go)));
                }
                int stop = (ms < nl) ? (int)listL[f + ms] : m;
                if (ms > 0 && (int)listR[f + nr - ms] < stop) stop = (int)listR[f + nr - ms];
                for (int q = 1 + gl; q <= ms; q += 16) devSwap(keys, f + (int)listL[f + q - 1], f + (int)listR[f + nr - q]);
                devWaveSync();
                const int cut = f + stop;
                const int parts[2][2] = {{first, cut}, {cut, last}};
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int pf = parts[h][0], pe = parts[h][1];
                    if (pe - pf > 16) { if (gl == 0) push(pf, pe, depth - 1); }
                    else rankSort(pf, pe);
                }
            }
        }
        __syncthreads();
        nSort = s_count[1] < maxTasks ? s_count[1] : maxTasks;
        uint32_t* sw = in; in = out; out = sw;
        __syncthreads();
    }
}
struct DevV3 { float x, y, z; };
SDF_DEV void devTriVerts(const float4* __restrict__ triV, int t, DevV3& a, DevV3& b, DevV3& c) {
    const float4 q0 = triV[3 * (size_t)t], q1 = triV[3 * (size_t)t + 1], q2 = triV[3 * (size_t)t + 2];
    a = DevV3{q0.x, q0.y, q0.z}; b = DevV3{q0.w, q1.x, q1.y}; c = DevV3{q1.z, q1.w, q2.x};
}
SDF_DEV float devComp(const DevV3& v, int d) { return d == 0 ? v.x : (d == 1 ? v.y : v.z); }

// kSubThreads threads per workgroup — 1024, 512 or 256 (256 until round 5), by how many workgroups there are: the sort rounds and the per-element passes
// have a subtree's hundreds of ranges to share out, but 128 registers per thread leave room for 2048 threads per CU
template <int kSubThreads>
__global__ void __launch_bounds__(kSubThreads) k_bvh_subtrees(const BvhTask* __restrict__ tasks, const uint32_t* __restrict__ order, const float4* __restrict__ triV,
                                                      double* __restrict__ sph, int* __restrict__ kids, BvhDevNode* __restrict__ nodeScratch, uint32_t* __restrict__ failed, uint32_t kDevSubtreeMax, unsigned long long* __restrict__ phaseClocks /* SDFHIP_TIMING: 100 MHz ticks of block 0 in A (workgroup), A (lane per node), B, C; else null */) {
    extern __shared__ unsigned char s_bvh_raw[];
    const KeyArr keys{reinterpret_cast<KeyTri*>(s_bvh_raw), 0};
    const IdxArr listL{reinterpret_cast<unsigned short*>(s_bvh_raw + KeyArr::bytes(kDevSubtreeMax))};      // index lists of the partitions, by position
    const IdxArr listR{reinterpret_cast<unsigned short*>(s_bvh_raw + KeyArr::bytes(kDevSubtreeMax) + IdxArr::bytes(kDevSubtreeMax))};
    uint32_t* sortA = reinterpret_cast<uint32_t*>(s_bvh_raw + KeyArr::bytes(kDevSubtreeMax) + 2 * IdxArr::bytes(kDevSubtreeMax));        // {first, last, depth limit} x 3 words
    uint32_t* sortB = sortA + 3 * kDevSortTasks;
    __shared__ uint32_t s_count[4];                  // [0] nodes of the next level, [1] sort tasks of the next round, [2] overflow / depth-limit flag
    // tables of a level prepared by the whole workgroup (see A below): first element, AABB (ordered-uint floats), centre, radius^2 bits, axis
    __shared__ uint32_t s_nb[kCoopNodes + 1]; __shared__ uint32_t s_box[kCoopNodes][6]; __shared__ double s_ctr[kCoopNodes][3];
    __shared__ unsigned long long s_r2[kCoopNodes]; __shared__ int s_dim[kCoopNodes];
    const BvhTask T = tasks[blockIdx.x];
    const uint32_t n = T.end - T.begin;
    const int tid = threadIdx.x;
    BvhDevNode* cur = nodeScratch + (size_t)blockIdx.x * 2u * (kDevSubtreeMax / 2u + 1u);
    BvhDevNode* nxt = cur + (kDevSubtreeMax / 2u + 1u);
    for (uint32_t i = tid; i < n; i += kSubThreads) keys[i] = KeyTri{0.f, (int)order[T.begin + i]};
    if (tid == 0) { cur[0] = BvhDevNode{T.innerId, T.begin, T.end, T.parentSlot}; s_count[2] = 0; }
    uint32_t nCur = 1, level = 0;
    __syncthreads();
    const bool clocked = phaseClocks != nullptr && blockIdx.x == 0 && tid == 0;
    unsigned long long tick = clocked ? wall_clock64() : 0ull;
    auto lapClock = [&](int which) { if (clocked) { const unsigned long long now = wall_clock64(); phaseClocks[which] += now - tick; tick = now; } };
    while (nCur > 0) {
        if (tid == 0) { s_count[0] = 0; s_count[1] = 0; }
        __syncthreads();
        // A level whose nodes are all large (the top levels of the subtree; sizes differ by at most one, the smallest is n >> level) is
        // prepared by the whole workgroup: only the centre SUM depends on the order of its operands and stays one lane's chain per node
        // (vertex loads eight triangles ahead); AABB, radius (maxima of identical expressions) and keys are taken per element.  cur[] is in
        // left-to-right order on such levels (C keeps it so).  One lane per node was 2 x 4096 dependent global loads at the subtree's root.
        const bool coop = level < 31u && (n >> level) > kCoopMin && nCur <= kCoopNodes;
        if (coop) {
            const float fhi = 3.402823466e+38f;
            for (uint32_t j = tid; j < nCur; j += kSubThreads) {
                s_nb[j] = cur[j].b - T.begin; s_r2[j] = 0ull;
#pragma unroll
                for (int k = 0; k < 3; k++) { s_box[j][k] = devOrdKey(-fhi); s_box[j][3 + k] = devOrdKey(fhi); }
            }
            if (tid == 0) s_nb[nCur] = n;
            __syncthreads();
            const uint32_t per = (n + (uint32_t)kSubThreads - 1u) / (uint32_t)kSubThreads, i0 = (uint32_t)tid * per, i1 = (i0 + per < n) ? i0 + per : n;
            uint32_t node0 = 0;
            if (i0 < i1) { uint32_t lo_ = 0, hi_ = nCur - 1; while (lo_ < hi_) { const uint32_t m_ = (lo_ + hi_ + 1) >> 1; if (s_nb[m_] <= i0) lo_ = m_; else hi_ = m_ - 1; } node0 = lo_; }
            {   // 1. AABB per node
                uint32_t node = node0;
                float tx = -fhi, ty = -fhi, tz = -fhi, bx = fhi, by = fhi, bz = fhi;
                auto flush = [&]() {
                    atomicMax(&s_box[node][0], devOrdKey(tx)); atomicMax(&s_box[node][1], devOrdKey(ty)); atomicMax(&s_box[node][2], devOrdKey(tz));
                    atomicMin(&s_box[node][3], devOrdKey(bx)); atomicMin(&s_box[node][4], devOrdKey(by)); atomicMin(&s_box[node][5], devOrdKey(bz));
                    tx = ty = tz = -fhi; bx = by = bz = fhi;
                };
                for (uint32_t i = i0; i < i1; i++) {
                    while (i >= s_nb[node + 1]) { flush(); node++; }
                    DevV3 v[3]; devTriVerts(triV, keys[i].tri, v[0], v[1], v[2]);
#pragma unroll
                    for (int k = 0; k < 3; k++) {            // (a NaN coordinate is never taken, as in `v > t ? v : t`)
                        tx = v[k].x > tx ? v[k].x : tx; bx = v[k].x < bx ? v[k].x : bx; ty = v[k].y > ty ? v[k].y : ty; by = v[k].y < by ? v[k].y : by;
                        tz = v[k].z > tz ? v[k].z : tz; bz = v[k].z < bz ? v[k].z : bz;
                    }
                }
                if (i0 < i1) flush();
            }
            __syncthreads();
            // 2a. the centre sums, in parallel and verified like the top levels' (k_csum_*): every chunk of 8 - 32 triangles is added up from zero, a
            // per-node scan gives each chunk the value it would start from, every chunk is re-run from that value with the reference's very
            // additions and its end compared with the next chunk's start bit for bit; a node with a disagreeing boundary is summed by one
            // lane below.  (One lane per node was the longest part of this kernel: 2560 + 1280 + 640 + ... dependent steps per subtree.)  The
            // scratch is the sort's index lists and task lists, free until the level's sorts begin.
            const size_t scratchBytes = 2 * IdxArr::bytes(kDevSubtreeMax) + 2 * 3 * 4 * (size_t)kDevSortTasks;
            uint32_t CH = 8u;                                        // triangles per chunk: the smallest of 8 / 16 / 32 whose tables fit (more chunks = more lanes at work)
            while (CH < 32u && (size_t)(n / CH + nCur) * 50u + 4u * (size_t)nCur + 8u > scratchBytes) CH *= 2u;
            const uint32_t maxChunks = n / CH + nCur;
            double* cs = reinterpret_cast<double*>(s_bvh_raw + KeyArr::bytes(kDevSubtreeMax));
            double* ci = cs + 3 * (size_t)maxChunks;
            unsigned short* cnode = reinterpret_cast<unsigned short*>(ci + 3 * (size_t)maxChunks);
            unsigned short* cbase = cnode + maxChunks;               // nCur + 1
            unsigned short* cbad = cbase + (nCur + 1u);              // nCur
            const bool par = (size_t)maxChunks * 50u + 4u * (size_t)nCur + 8u <= scratchBytes;
            auto chunkAdd = [&](uint32_t lo_, uint32_t hi_, double& sx, double& sy, double& sz) {
                for (uint32_t i = lo_; i < hi_; i++) {
                    const size_t t = (size_t)keys[i].tri;
                    const float4 q0 = triV[3 * t], q1 = triV[3 * t + 1]; const float q2 = reinterpret_cast<const float*>(triV)[12 * t + 8];
                    sx += (double)q0.x; sy += (double)q0.y; sz += (double)q0.z;
                    sx += (double)q0.w; sy += (double)q1.x; sz += (double)q1.y;
                    sx += (double)q1.z; sy += (double)q1.w; sz += (double)q2;
                }
            };
            if (par) {
                for (uint32_t j = tid; j < nCur; j += kSubThreads) { cbase[j] = (unsigned short)((s_nb[j + 1] - s_nb[j] + CH - 1u) / CH); cbad[j] = 0; }
                __syncthreads();
                if (tid == 0) { uint32_t run = 0; for (uint32_t j = 0; j < nCur; j++) { const uint32_t c = cbase[j]; cbase[j] = (unsigned short)run; run += c; } cbase[nCur] = (unsigned short)run; }
                __syncthreads();
                for (uint32_t j = tid; j < nCur; j += kSubThreads) for (uint32_t c = cbase[j]; c < cbase[j + 1]; c++) cnode[c] = (unsigned short)j;
                __syncthreads();
                const uint32_t total = cbase[nCur];
                for (uint32_t c = tid; c < total; c += kSubThreads) {
                    const uint32_t j = cnode[c], lo_ = s_nb[j] + (c - cbase[j]) * CH, hi_ = (lo_ + CH < s_nb[j + 1]) ? lo_ + CH : s_nb[j + 1];
                    double sx = 0.0, sy = 0.0, sz = 0.0;
                    chunkAdd(lo_, hi_, sx, sy, sz);
                    cs[3 * c] = sx; cs[3 * c + 1] = sy; cs[3 * c + 2] = sz;
                }
                __syncthreads();
                for (uint32_t j = tid; j < nCur; j += kSubThreads) {
                    double px = 0.0, py = 0.0, pz = 0.0;
                    for (uint32_t c = cbase[j]; c < cbase[j + 1]; c++) { ci[3 * c] = px; ci[3 * c + 1] = py; ci[3 * c + 2] = pz; px += cs[3 * c]; py += cs[3 * c + 1]; pz += cs[3 * c + 2]; }
                }
                __syncthreads();
                for (uint32_t c = tid; c < total; c += kSubThreads) {
                    const uint32_t j = cnode[c], lo_ = s_nb[j] + (c - cbase[j]) * CH, hi_ = (lo_ + CH < s_nb[j + 1]) ? lo_ + CH : s_nb[j + 1];
                    double sx = ci[3 * c], sy = ci[3 * c + 1], sz = ci[3 * c + 2];
                    chunkAdd(lo_, hi_, sx, sy, sz);
                    if (c + 1u < cbase[j + 1]) {
                        if (__double_as_longlong(sx) != __double_as_longlong(ci[3 * (c + 1)]) || __double_as_longlong(sy) != __double_as_longlong(ci[3 * (c + 1) + 1]) ||
                            __double_as_longlong(sz) != __double_as_longlong(ci[3 * (c + 1) + 2])) cbad[j] = 1;
                    } else {
                        const double cnt = (double)(3 * (s_nb[j + 1] - s_nb[j]));
                        s_ctr[j][0] = sx / cnt; s_ctr[j][1] = sy / cnt; s_ctr[j][2] = sz / cnt;
                    }
                }
                __syncthreads();
            }
            // 2b. one lane per node: split axis; the centre by the serial chain where the parallel form was not verified
            for (uint32_t j = tid; j < nCur; j += kSubThreads) {
                const double d0 = (double)devOrdVal(s_box[j][0]) - (double)devOrdVal(s_box[j][3]), d1 = (double)devOrdVal(s_box[j][1]) - (double)devOrdVal(s_box[j][4]),
                             d2 = (double)devOrdVal(s_box[j][2]) - (double)devOrdVal(s_box[j][5]);
                int dim = 0; double dm = d0;
                if (dm < d1) { dim = 1; dm = d1; }
                if (dm < d2) dim = 2;
                s_dim[j] = dim;
                if (par && !cbad[j]) continue;
                double sx = 0.0, sy = 0.0, sz = 0.0;
                chunkAdd(s_nb[j], s_nb[j + 1], sx, sy, sz);
                const double cnt = (double)(3 * (s_nb[j + 1] - s_nb[j]));
                s_ctr[j][0] = sx / cnt; s_ctr[j][1] = sy / cnt; s_ctr[j][2] = sz / cnt;
            }
            __syncthreads();
            {   // 3. radius and keys per element
                uint32_t node = node0;
                double r2 = 0.0;
                double cx = 0, cy = 0, cz = 0; int dim = 0;
                if (i0 < i1) { cx = s_ctr[node][0]; cy = s_ctr[node][1]; cz = s_ctr[node][2]; dim = s_dim[node]; }
                for (uint32_t i = i0; i < i1; i++) {
                    while (i >= s_nb[node + 1]) {
                        atomicMax(&s_r2[node], (unsigned long long)__double_as_longlong(r2));
                        node++; r2 = 0.0; cx = s_ctr[node][0]; cy = s_ctr[node][1]; cz = s_ctr[node][2]; dim = s_dim[node];
                    }
                    DevV3 v[3]; devTriVerts(triV, keys[i].tri, v[0], v[1], v[2]);
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const double dx = cx - (double)v[k].x, dy = cy - (double)v[k].y, dz = cz - (double)v[k].z;
                        const double q = dx * dx + dy * dy + dz * dz;
                        r2 = r2 < q ? q : r2;                  // (never a NaN, never negative: the bits order like the values)
                    }
                    keys[i].key = devComp(v[0], dim);
                }
                if (i0 < i1) atomicMax(&s_r2[node], (unsigned long long)__double_as_longlong(r2));
            }
            __syncthreads();
            // 4. spheres and the level's sort tasks (every range is longer than 16)
            for (uint32_t j = tid; j < nCur; j += kSubThreads) {
                const BvhDevNode nd = cur[j];
                if (nd.slot != 0xFFFFFFFFu) { double* o = sph + 4 * (size_t)nd.slot; o[0] = s_ctr[j][0]; o[1] = s_ctr[j][1]; o[2] = s_ctr[j][2]; o[3] = sqrt(__longlong_as_double((long long)s_r2[j])); }
                const int nn = (int)(nd.e - nd.b);
                int lg = 0; for (int m = nn; m > 1; m >>= 1) lg++;
                sortA[3 * j] = s_nb[j]; sortA[3 * j + 1] = s_nb[j] + (uint32_t)nn; sortA[3 * j + 2] = (uint32_t)(2 * lg);
            }
            if (tid == 0) s_count[1] = nCur;
        }
        lapClock(0);
        // ---- A. one lane per node: centre (ordered fp64 sum), AABB -> axis, radius, keys; short ranges sorted at once
        for (uint32_t j = tid; j < (coop ? 0u : nCur); j += kSubThreads) {
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
        lapClock(1);
        // ---- B. introsort rounds: every pending range is partitioned by one lane; parts of at most 16 are finished on the spot
        {
            const uint32_t nSort = s_count[1] < kDevSortTasks ? s_count[1] : kDevSortTasks;
            __syncthreads();
            devSortRounds(keys, listL, listR, sortA, sortB, s_count, nSort, kDevSortTasks, tid, (uint32_t)kSubThreads);
        }
        lapClock(2);
        // ---- C. one lane per node: the children (leaves get their sphere here, inner children at the next level)
        for (uint32_t j = tid; j < nCur; j += kSubThreads) {
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
                    const uint32_t at = coop ? 2u * j + (uint32_t)side : atomicAdd(&s_count[0], 1u);       // (a coop level's children are all inner nodes: left-to-right order is kept)
                    nxt[at] = BvhDevNode{childId[side], rb[side], re[side], slot};
                }
            }
        }
        __syncthreads();
        lapClock(3);
        nCur = coop ? 2u * nCur : s_count[0];
        level++;
        BvhDevNode* sw = cur; cur = nxt; nxt = sw;
        __syncthreads();
    }
    if (tid == 0 && s_count[2]) atomicOr(failed, s_count[2]);
}

static int raiseSubtreeLds(int bytes) {
    SDF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bvh_subtrees<256>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    SDF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bvh_subtrees<512>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    SDF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bvh_subtrees<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return SDFHIP_OK;
}
// one workgroup per CU at 1024 threads (128 registers each), two at 512: as many threads as leave every workgroup resident at once
// (327 680 triangles, 128 subtrees: 1.5 -> 1.2 ms; 1.31 M, 512 subtrees: 1024 threads would run them in two turns, 2.4 ms against 1.9)
template <typename... A>
static void launchSubtrees(unsigned nt, size_t lds, hipStream_t st, A... a) {
    if (nt <= 256u) k_bvh_subtrees<1024><<<nt, 1024, lds, st>>>(a...);
    else if (nt <= 512u) k_bvh_subtrees<512><<<nt, 512, lds, st>>>(a...);
    else k_bvh_subtrees<256><<<nt, 256, lds, st>>>(a...);
}

// ---- the top of the tree on the device too ------------------------------------------------------------------------------------------------
// (The default; SDFHIP_BVH_BUILD=host switches it off.)  The levels whose nodes hold more than kDevSubtreeMax triangles are sorted in global memory, level by level, all
// nodes of a level at once; k_bvh_subtrees then finishes every range.  The sort is libstdc++'s introsort once more, as ROUNDS: a round
// partitions every pending range (> kDevSubtreeMax) once with the result of __unguarded_partition — the index lists of the elements not less /
// not greater than the pivot, the t-th from the left exchanged with the t-th from the right while they have not crossed (see
// IntroSortLike::parallelPartition) — on as many workgroups as the range has chunks; parts that fit in LDS are finished by k_sort_parts
// (devSortRounds), parts of at most 16 by k_sort_tiny.  What depends on an order besides — the fp64 centre sums in range order — runs on a
// second stream behind each level's sort: three lanes per node add x, y and z (k_top_sums), the rest of the wave gathers.
// Measured in round 4 and NOT kept (profiles/r04_bvh_persistent_rounds.txt): all rounds of a level in ONE persistent kernel — one workgroup per
// CU striding over the chunks, the five phases separated by a device-wide barrier (monotone counter, release / acquire fences at agent
// scope), prepare + emit on workgroup 0.  Identical trees, but 9 levels of the 1.31 M mesh took 19.4 ms instead of 9.2 ms: a barrier
// across the eight XCDs costs ~25 us (every workgroup's release writes its L2 back, every acquire invalidates it), 624 of them per build,
// while a launch of the five-kernel form costs ~7 us in a queue the host fills ahead.  The launches stay.
struct TopNode { int id; uint32_t b, e, slot; };                    // slot: where the node's sphere goes, in units of 4 doubles (NONE32: the root's, nowhere)
struct GTask { uint32_t first, last, depth; };
constexpr uint32_t kGsChunk = 2048;          // elements per workgroup in the round kernels (256 threads x 8)
constexpr uint32_t kGsMaxRangeChunks = 7000; // chunks of the longest range k_gs_swap can hold prefix sums for (56 KB of LDS): 14 M triangles; beyond -> host planner
struct GsRound {
    KeyTri* K; uint32_t* Ll; uint32_t* Rl;                          // keys; index lists, chunk by chunk (a chunk's entries start at its first position)
    const GTask* tasks; const uint32_t* nTasksPtr; uint32_t maxTasks;
    float* pk; uint32_t* chunkBase; uint32_t* cut;                   // per range
    uint32_t* cntL; uint32_t* cntR; uint32_t* chunkTask;             // per chunk (chunkTask: the range a chunk belongs to, written with the round's layout)
};
SDF_DEV uint32_t gsTaskCount(const GsRound& R) { const uint32_t n = *R.nTasksPtr; return n < R.maxTasks ? n : R.maxTasks; }

// one workgroup of BLOCK threads: pivots (median of three to the front), chunk layout of the round; zeroes the next round's counter
template <int BLOCK>
SDF_DEV void gsPrepare(const GsRound& R, uint32_t* __restrict__ nextCount, uint32_t* __restrict__ flags) {
    __shared__ uint32_t s_part[BLOCK / 64]; __shared__ uint32_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nT = gsTaskCount(R);
    if (tid == 0) { s_carry = 0; *nextCount = 0; if (*R.nTasksPtr > R.maxTasks) atomicOr(flags, 1u); }
    __syncthreads();
    for (uint32_t base = 0; base < nT; base += (uint32_t)BLOCK) {
        const uint32_t t = base + (uint32_t)tid;
        uint32_t nch = 0;
        if (t < nT) {
            const GTask task = R.tasks[t];
            const int first = (int)task.first, last = (int)task.last;
            if (task.depth == 0) atomicOr(flags, 2u);                 // heap sort of a range this long: left to the host planner
            else devMedianToFirst(R.K, first, first + 1, first + (last - first) / 2, last - 1);
            R.pk[t] = R.K[first].key;
            nch = ((uint32_t)(last - first - 1) + kGsChunk - 1u) / kGsChunk;
        }
        uint32_t incl = nch;                                          // inclusive scan over the workgroup
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        if (lane == 63) s_part[wave] = incl;
        __syncthreads();
        uint32_t before = 0;
        for (int w = 0; w < wave; w++) before += s_part[w];
        if (t < nT) {
            const uint32_t cb = s_carry + before + incl - nch;
            R.chunkBase[t] = cb;
            for (uint32_t k = 0; k < nch; k++) R.chunkTask[cb + k] = t;      // (the round kernels' workgroups read their range here)
        }
        __syncthreads();
        if (tid == BLOCK - 1) s_carry += before + incl;
        __syncthreads();
    }
    if (tid == 0) R.chunkBase[nT] = s_carry;
}
// a level's first round: its ranges are the level's nodes (introsort's depth limit 2 floor(log2 n)), the level's counters start over; then
// the round is prepared like every other (one workgroup; the ranges and their count are read by this workgroup only)
__global__ void __launch_bounds__(1024) k_gs_first_prepare(const TopNode* __restrict__ nodes, uint32_t count, GTask* __restrict__ tasks, uint32_t* __restrict__ ctr, GsRound R,
                                                           uint32_t* __restrict__ nextCount, uint32_t* __restrict__ flags) {
    if (threadIdx.x == 0) { ctr[0] = count; ctr[1] = 0; ctr[2] = 0; ctr[3] = 0; }
    for (uint32_t j = threadIdx.x; j < count; j += 1024u) {
        const uint32_t len = nodes[j].e - nodes[j].b;
        tasks[j] = GTask{nodes[j].b, nodes[j].e, 2u * (31u - (uint32_t)__clz((int)(len | 1u)))};
    }
    __threadfence_block();
    __syncthreads();
    gsPrepare<1024>(R, nextCount, flags);
}
// which range and which of its chunks a workgroup of the round kernels works on
struct GsChunk { uint32_t task; int f, m; uint32_t k; float pk; };
SDF_DEV bool gsLocate(const GsRound& R, uint32_t block, GsChunk& c) {
    const uint32_t nT = gsTaskCount(R);
    if (block >= R.chunkBase[nT]) return false;                      // (uniform per workgroup)
    c.task = R.chunkTask[block];
    const GTask task = R.tasks[c.task];
    c.f = (int)task.first + 1; c.m = (int)task.last - c.f; c.k = block - R.chunkBase[c.task]; c.pk = R.pk[c.task];
    return true;
}
// A round is THREE launches (four until round 5: count, fill, swap, emit + prepare, each ~15 us of a dependent chain whatever it does).
// k_gs_mark, a workgroup per chunk: the chunk's entries of the two index lists — positions (relative to the range) of the elements not
// less / not greater than the pivot, ascending — written from the chunk's first position on, and their numbers.  No workgroup needs
// another's result.
__global__ void __launch_bounds__(256) k_gs_mark(GsRound R) {
    __shared__ uint32_t s_wave[2][4];
    GsChunk c; if (!gsLocate(R, blockIdx.x, c)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = (int)(c.k * kGsChunk) + 8 * tid;
    uint32_t a = 0, b = 0; unsigned fl = 0, fr = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) if (i0 + j < c.m) { const float key = R.K[c.f + i0 + j].key; const bool l = !(key < c.pk), r = !(c.pk < key); fl |= (unsigned)l << j; fr |= (unsigned)r << j; a += l; b += r; }
    uint32_t ia = a, ib = b;                                          // inclusive scans in thread (= element) order
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t va = __shfl_up(ia, o), vb = __shfl_up(ib, o); if (lane >= o) { ia += va; ib += vb; } }
    if (lane == 63) { s_wave[0][wave] = ia; s_wave[1][wave] = ib; }
    __syncthreads();
    uint32_t ba = 0, bb = 0;
    for (int w = 0; w < wave; w++) { ba += s_wave[0][w]; bb += s_wave[1][w]; }
    const int base = c.f + (int)(c.k * kGsChunk);
    uint32_t wa = ba + ia - a, wb = bb + ib - b;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if ((fl >> j) & 1u) R.Ll[base + (int)wa++] = (uint32_t)(i0 + j);
        if ((fr >> j) & 1u) R.Rl[base + (int)wb++] = (uint32_t)(i0 + j);
    }
    if (tid == 0) { R.cntL[blockIdx.x] = s_wave[0][0] + s_wave[0][1] + s_wave[0][2] + s_wave[0][3]; R.cntR[blockIdx.x] = s_wave[1][0] + s_wave[1][1] + s_wave[1][2] + s_wave[1][3]; }
}
// the cut of a range and what becomes of its two parts
SDF_DEV void gsEmitOne(const GsRound& R, uint32_t t, GTask* __restrict__ next, uint32_t* __restrict__ nextCount, GTask* __restrict__ parts, uint32_t* __restrict__ partCount, uint32_t maxParts,
                       GTask* __restrict__ tiny, uint32_t* __restrict__ tinyCount, uint32_t maxTiny, uint32_t ldsMax, uint32_t* __restrict__ flags) {
    const GTask task = R.tasks[t];
    const uint32_t cut = R.cut[t];
    const uint32_t pb[2] = {task.first, cut}, pe[2] = {cut, task.last};
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint32_t len = pe[h] - pb[h];
        const GTask out{pb[h], pe[h], task.depth - 1u};
        if (len > ldsMax) { const uint32_t at = atomicAdd(nextCount, 1u); if (at < R.maxTasks) next[at] = out; else atomicOr(flags, 1u); }
        else if (len > 16u) { const uint32_t at = atomicAdd(partCount, 1u); if (at < maxParts) parts[at] = out; else atomicOr(flags, 1u); }
        else if (len > 1u) { const uint32_t at = atomicAdd(tinyCount, 1u); if (at < maxTiny) tiny[at] = out; else atomicOr(flags, 1u); }
    }
}
// k_gs_swap, eight workgroups per chunk, a thread per exchange: pair q (1-based) is the q-th entry of the range's L list and the q-th entry
// of its R list from the END, exchanged while they have not crossed (see IntroSortLike::parallelPartition).  Where the q-th entry lies
// follows from the prefix sums of the range's chunk counts, which every workgroup builds in LDS (a range has at most a few hundred chunks);
// since the L entries ascend and the R entries from the end descend, "not crossed" holds for q <= ms and for no q beyond: the thread that
// sees the boundary knows ms and writes the range's cut — __unguarded_partition's result — itself.
__global__ void __launch_bounds__(256) k_gs_swap(GsRound R, uint32_t nchCap) {
    extern __shared__ uint32_t s_pre[];                               // pL[0 .. nchCap], pR[0 .. nchCap]
    __shared__ uint32_t s_scan[2][4]; __shared__ unsigned char s_sq[256];
    GsChunk c; if (!gsLocate(R, blockIdx.x >> 3, c)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* pL = s_pre; uint32_t* pR = s_pre + (nchCap + 1u);
    const uint32_t cb = R.chunkBase[c.task], nch = R.chunkBase[c.task + 1u] - cb;
    {   // exclusive prefix sums of the range's chunk counts
        const uint32_t per = (nch + 255u) / 256u, k0 = ((uint32_t)tid * per < nch) ? (uint32_t)tid * per : nch, k1 = (k0 + per < nch) ? k0 + per : nch;
        uint32_t sa = 0, sb = 0;
        for (uint32_t k = k0; k < k1; k++) { sa += R.cntL[cb + k]; sb += R.cntR[cb + k]; }
        uint32_t ia = sa, ib = sb;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t va = __shfl_up(ia, o), vb = __shfl_up(ib, o); if (lane >= o) { ia += va; ib += vb; } }
        if (lane == 63) { s_scan[0][wave] = ia; s_scan[1][wave] = ib; }
        __syncthreads();
        uint32_t ea = ia - sa, eb = ib - sb;
        for (int w = 0; w < wave; w++) { ea += s_scan[0][w]; eb += s_scan[1][w]; }
        for (uint32_t k = k0; k < k1; k++) { pL[k] = ea; pR[k] = eb; ea += R.cntL[cb + k]; eb += R.cntR[cb + k]; }
        if (tid == 0) { pL[nch] = s_scan[0][0] + s_scan[0][1] + s_scan[0][2] + s_scan[0][3]; pR[nch] = s_scan[1][0] + s_scan[1][1] + s_scan[1][2] + s_scan[1][3]; }
        __syncthreads();
    }
    const uint32_t nl = pL[nch], nr = pR[nch], lim = nl < nr ? nl : nr;
    auto nth = [&](const uint32_t* pre, const uint32_t* list, uint32_t idx) {       // the idx-th entry (0-based) of a list
        uint32_t lo = 0, hi = nch - 1u;
        while (lo < hi) { const uint32_t mid = (lo + hi + 1u) >> 1; if (pre[mid] <= idx) lo = mid; else hi = mid - 1u; }
        return list[c.f + (int)(lo * kGsChunk + (idx - pre[lo]))];
    };
    const uint32_t q = c.k * kGsChunk + (blockIdx.x & 7u) * 256u + (uint32_t)tid + 1u;
    uint32_t a = 0, b = 0; bool sq = false;
    if (q <= lim) {
        a = nth(pL, R.Ll, q - 1u); b = nth(pR, R.Rl, nr - q);
        sq = a < b;
        if (sq) devSwap(R.K, c.f + (int)a, c.f + (int)b);
    }
    s_sq[tid] = sq ? 1 : 0;
    __syncthreads();
    {   // the boundary: ms = the number of exchanges of the range
        uint32_t ms = 0xFFFFFFFFu;
        if (lim == 0u) { if (q == 1u) ms = 0u; }
        else if (q <= lim) {
            if (sq) { if (q == lim) ms = lim; }
            else {
                bool prev = true;                                     // pair q - 1 exchanged?  (q == 1: there is none)
                if (q > 1u) prev = (tid > 0) ? (s_sq[tid - 1] != 0) : (nth(pL, R.Ll, q - 2u) < nth(pR, R.Rl, nr - (q - 1u)));
                if (prev) ms = q - 1u;
            }
        }
        if (ms != 0xFFFFFFFFu) {
            uint32_t stop = (ms < nl) ? nth(pL, R.Ll, ms) : (uint32_t)c.m;
            if (ms > 0u) { const uint32_t r = nth(pR, R.Rl, nr - ms); if (r < stop) stop = r; }
            R.cut[c.task] = (uint32_t)c.f + stop;
        }
    }
}
// ONE workgroup ends a round and begins the next: the parts of every range go to the next round's list, to the LDS sorts or to the tiny
// sorts; then — the list complete — the next round's pivots and chunk layout (gsPrepare).  (Round 5 also tried this as the tail of
// k_gs_swap, run by the workgroup that finishes last behind a device-scope fence and a ticket: identical trees, but every workgroup's
// fence writes its XCD's L2 back — 9 levels of the 1.31 M mesh took 30 ms instead of 7.4.  A kernel boundary is the cheap way to make
// eight L2s agree, as round 4's persistent form had already shown.)
__global__ void __launch_bounds__(1024) k_gs_emit_prepare(GsRound R, GsRound Rnext, uint32_t* __restrict__ afterNextCount, GTask* __restrict__ parts, uint32_t* __restrict__ partCount, uint32_t maxParts,
                                                          GTask* __restrict__ tiny, uint32_t* __restrict__ tinyCount, uint32_t maxTiny, uint32_t ldsMax, uint32_t* __restrict__ flags) {
    const uint32_t nT = gsTaskCount(R);
    for (uint32_t t = threadIdx.x; t < nT; t += 1024u)
        gsEmitOne(R, t, const_cast<GTask*>(Rnext.tasks), const_cast<uint32_t*>(Rnext.nTasksPtr), parts, partCount, maxParts, tiny, tinyCount, maxTiny, ldsMax, flags);
    __threadfence_block();           // (the list and its count are read by this workgroup only)
    __syncthreads();
    gsPrepare<1024>(Rnext, afterNextCount, flags);
}
// a part that fits in LDS: the rest of its introsort in one workgroup
// (1024 threads: a round's ranges are shared out among the waves and the groups of sixteen lanes, and a part of 4096 keys has hundreds of them)
__global__ void __launch_bounds__(1024) k_sort_parts(KeyTri* __restrict__ K, const GTask* __restrict__ parts, uint32_t ldsMax, uint32_t* __restrict__ flags) {
    extern __shared__ unsigned char s_bvh_raw[];
    const KeyArr keys{reinterpret_cast<KeyTri*>(s_bvh_raw), 0};
    const IdxArr listL{reinterpret_cast<unsigned short*>(s_bvh_raw + KeyArr::bytes(ldsMax))};
    const IdxArr listR{reinterpret_cast<unsigned short*>(s_bvh_raw + KeyArr::bytes(ldsMax) + IdxArr::bytes(ldsMax))};
    uint32_t* sortA = reinterpret_cast<uint32_t*>(s_bvh_raw + KeyArr::bytes(ldsMax) + 2 * IdxArr::bytes(ldsMax));
    uint32_t* sortB = sortA + 3 * kDevSortTasks;
    __shared__ uint32_t s_count[4];
    const GTask part = parts[blockIdx.x];
    const int n = (int)(part.last - part.first), tid = threadIdx.x;
    for (int i = tid; i < n; i += (int)blockDim.x) keys[i] = K[part.first + i];
    if (tid == 0) { sortA[0] = 0; sortA[1] = (uint32_t)n; sortA[2] = part.depth; s_count[1] = 1; s_count[2] = 0; }
    __syncthreads();
    devSortRounds(keys, listL, listR, sortA, sortB, s_count, 1u, kDevSortTasks, tid, blockDim.x);
    for (int i = tid; i < n; i += (int)blockDim.x) K[part.first + i] = keys[i];
    if (tid == 0 && s_count[2]) atomicOr(flags, s_count[2]);
}
__global__ void k_sort_tiny(KeyTri* __restrict__ K, const GTask* __restrict__ tiny, uint32_t count) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < count) devInsertionSort(K, (int)tiny[t].first, (int)tiny[t].last);
}

// ---- per level: AABB -> axis, keys; afterwards the centre sums and radii of its nodes
SDF_DEV uint32_t topNodeOf(const TopNode* __restrict__ nodes, uint32_t count, uint32_t i) {
    uint32_t lo = 0, hi = count - 1;
    while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (nodes[mid].b <= i) lo = mid; else hi = mid - 1; }
    return lo;
}
__global__ void k_top_init(uint32_t* __restrict__ box, unsigned long long* __restrict__ r2, uint32_t count) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const float fhi = 3.402823466e+38f;
    for (int k = 0; k < 3; k++) { box[6 * j + k] = devOrdKey(-fhi); box[6 * j + 3 + k] = devOrdKey(fhi); }
    r2[j] = 0ull;
}
__global__ void k_key_init(KeyTri* __restrict__ K, uint32_t n) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) K[i] = KeyTri{0.f, (int)i}; }
// 4 consecutive elements per thread; a wave whose lanes all sit in one node (the rule: nodes are longer than a workgroup's 1024 elements) reduces first
__global__ void __launch_bounds__(256) k_top_aabb(const KeyTri* __restrict__ K, const float4* __restrict__ triV, const TopNode* __restrict__ nodes, uint32_t count, uint32_t n, uint32_t* __restrict__ box) {
    const uint32_t i0 = 4u * (blockIdx.x * 256u + threadIdx.x);
    const float fhi = 3.402823466e+38f;
    float tx = -fhi, ty = -fhi, tz = -fhi, bx = fhi, by = fhi, bz = fhi;
    uint32_t node = 0xFFFFFFFFu;
    auto flush = [&]() {
        atomicMax(&box[6 * node], devOrdKey(tx)); atomicMax(&box[6 * node + 1], devOrdKey(ty)); atomicMax(&box[6 * node + 2], devOrdKey(tz));
        atomicMin(&box[6 * node + 3], devOrdKey(bx)); atomicMin(&box[6 * node + 4], devOrdKey(by)); atomicMin(&box[6 * node + 5], devOrdKey(bz));
        tx = ty = tz = -fhi; bx = by = bz = fhi;
    };
    if (i0 < n) node = topNodeOf(nodes, count, i0);
    const uint32_t nodeAtStart = node;
    for (uint32_t i = i0; i < i0 + 4u && i < n; i++) {
        while (i >= nodes[node].e) { flush(); node++; }
        DevV3 v[3]; devTriVerts(triV, K[i].tri, v[0], v[1], v[2]);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            tx = v[k].x > tx ? v[k].x : tx; bx = v[k].x < bx ? v[k].x : bx; ty = v[k].y > ty ? v[k].y : ty; by = v[k].y < by ? v[k].y : by;
            tz = v[k].z > tz ? v[k].z : tz; bz = v[k].z < bz ? v[k].z : bz;
        }
    }
    // a workgroup whose 1024 elements all lie in one node (most: nodes are longer than that) folds its box in LDS first: 6 atomics on the
    // node's record instead of 24 (at the root every workgroup of the launch meets on the same six words)
    __shared__ uint32_t s_fold[6];
    const uint32_t nodeFirst = topNodeOf(nodes, count, 1024u * blockIdx.x);
    const uint32_t lastEl = (1024u * blockIdx.x + 1023u < n) ? 1024u * blockIdx.x + 1023u : n - 1u;
    const bool whole = lastEl < nodes[nodeFirst].e;                  // (uniform)
    if (whole) {
        if (threadIdx.x < 3) s_fold[threadIdx.x] = devOrdKey(-fhi); else if (threadIdx.x < 6) s_fold[threadIdx.x] = devOrdKey(fhi);
        __syncthreads();
    }
    const uint32_t first = __shfl(nodeAtStart, 0);
    if (__all(nodeAtStart == first && node == first && first != 0xFFFFFFFFu)) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float a = __shfl_xor(tx, o), b = __shfl_xor(ty, o), c = __shfl_xor(tz, o), d = __shfl_xor(bx, o), e = __shfl_xor(by, o), f = __shfl_xor(bz, o);
            tx = a > tx ? a : tx; ty = b > ty ? b : ty; tz = c > tz ? c : tz; bx = d < bx ? d : bx; by = e < by ? e : by; bz = f < bz ? f : bz;
        }
        if ((threadIdx.x & 63) == 0) {
            if (whole) {
                atomicMax(&s_fold[0], devOrdKey(tx)); atomicMax(&s_fold[1], devOrdKey(ty)); atomicMax(&s_fold[2], devOrdKey(tz));
                atomicMin(&s_fold[3], devOrdKey(bx)); atomicMin(&s_fold[4], devOrdKey(by)); atomicMin(&s_fold[5], devOrdKey(bz));
            } else flush();
        }
    } else if (node != 0xFFFFFFFFu) flush();
    if (whole) {
        __syncthreads();
        if (threadIdx.x < 3) atomicMax(&box[6 * nodeFirst + threadIdx.x], s_fold[threadIdx.x]);
        else if (threadIdx.x < 6) atomicMin(&box[6 * nodeFirst + threadIdx.x], s_fold[threadIdx.x]);
    }
}
// (the split axis of the element's node straight from the node's box — std::max_element: the first of equal maxima — a launch per level of its own until round 5)
__global__ void __launch_bounds__(256) k_top_keys(KeyTri* __restrict__ K, const float4* __restrict__ triV, const TopNode* __restrict__ nodes, uint32_t count, uint32_t n, const uint32_t* __restrict__ box) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = topNodeOf(nodes, count, i);
    double d[3];
    for (int k = 0; k < 3; k++) d[k] = (double)devOrdVal(box[6 * j + k]) - (double)devOrdVal(box[6 * j + 3 + k]);
    int dim = 0;
    for (int k = 1; k < 3; k++) if (d[dim] < d[k]) dim = k;
    const size_t t = (size_t)K[i].tri;
    const float4 q0 = triV[3 * t];
    K[i].key = dim == 0 ? q0.x : (dim == 1 ? q0.y : q0.z);
}
// the order a level's sorts left (the centre sums of the next level's nodes run over it on a side stream); its first threads also write the level's child links
__global__ void k_top_snapshot(const KeyTri* __restrict__ K, uint32_t n, uint32_t* __restrict__ snap, const TopNode* __restrict__ nodes, uint32_t count, int* __restrict__ kids) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) snap[i] = (uint32_t)K[i].tri;
    if (i < count) { const TopNode nd = nodes[i]; const uint32_t mid = (nd.b + nd.e) >> 1; kids[2 * (size_t)nd.id] = nd.id + 1; kids[2 * (size_t)nd.id + 1] = nd.id + (int)(mid - nd.b); }
}
// Centre of a node = its vertices summed in range order by the reference, three chains (x, y, z) of 3 n dependent fp64 additions each.
// Round 5: the chains are reproduced IN PARALLEL and verified.  A chain is cut into chunks of kSumChunk triangles; (1) k_csum_local adds
// every chunk up from zero, (2) k_csum_scan turns the chunk sums of a node into the value each chunk would START from, (3) k_csum_check
// re-runs every chunk from that value with the reference's very additions and compares where it ends with where the next chunk starts,
// bit for bit.  If every boundary of a node agrees, the chunks laid end to end ARE the sequential chain (by induction from the first chunk,
// which starts from 0 like the reference) and the last chunk's end is the reference's sum — whether or not an addition rounded on the way.
// The guesses of (1) + (2) are right whenever no addition of the chain rounds, which is the rule for fp32 coordinates summed in fp64
// (3.9 M additions of the 1.31 M-triangle mesh's longest chain: none rounds); a node with a disagreeing boundary is flagged and summed by
// the serial chain (k_top_sums, which then runs for flagged nodes only).  Rounds 3-4 ran the serial chain for every node (13 ms per
// 655 360 triangles on a lane, bound by the latency of dependent additions) and sent the longest nodes' coordinates to HOST threads.
struct CsumLevel { const uint32_t* order; const float4* triV; const TopNode* nodes; uint32_t count; const uint32_t* chunkBase; uint32_t totalChunks; double* csum; double* cin; uint32_t* status; double* centres; };
SDF_DEV uint32_t csumNodeOf(const uint32_t* __restrict__ chunkBase, uint32_t count, uint32_t c) {
    uint32_t lo = 0, hi = count - 1;
    while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (chunkBase[mid] <= c) lo = mid; else hi = mid - 1; }
    return lo;
}
// the additions of one chunk, in the reference's order: per triangle its three vertices, per vertex x, y, z into their chains
SDF_DEV void csumChunk(const CsumLevel& L, const TopNode& nd, uint32_t k, double& sx, double& sy, double& sz) {
    const uint32_t b = nd.b + k * kSumChunk, e = (nd.e - b < kSumChunk) ? nd.e : b + kSumChunk;
    for (uint32_t i = b; i < e; i++) {
        const size_t t = (size_t)L.order[i];
        const float4 q0 = L.triV[3 * t], q1 = L.triV[3 * t + 1]; const float q2 = reinterpret_cast<const float*>(L.triV)[12 * t + 8];
        sx += (double)q0.x; sy += (double)q0.y; sz += (double)q0.z;
        sx += (double)q0.w; sy += (double)q1.x; sz += (double)q1.y;
        sx += (double)q1.z; sy += (double)q1.w; sz += (double)q2;
    }
}
__global__ void __launch_bounds__(256) k_csum_local(CsumLevel L) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= L.totalChunks) return;
    const uint32_t j = csumNodeOf(L.chunkBase, L.count, c);
    double sx = 0.0, sy = 0.0, sz = 0.0;
    csumChunk(L, L.nodes[j], c - L.chunkBase[j], sx, sy, sz);
    L.csum[3 * (size_t)c] = sx; L.csum[3 * (size_t)c + 1] = sy; L.csum[3 * (size_t)c + 2] = sz;
}
// one workgroup per node: exclusive prefix sums of its chunk sums (thread t takes a contiguous share, thread 0 chains the 256 share totals)
__global__ void __launch_bounds__(256) k_csum_scan(CsumLevel L) {
    __shared__ double s_tot[256][3];
    const uint32_t j = blockIdx.x, c0 = L.chunkBase[j], n = L.chunkBase[j + 1] - c0, t = threadIdx.x;
    const uint32_t per = (n + 255u) / 256u, a = t * per < n ? t * per : n, z = a + per < n ? a + per : n;
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (uint32_t k = a; k < z; k++) { sx += L.csum[3 * (size_t)(c0 + k)]; sy += L.csum[3 * (size_t)(c0 + k) + 1]; sz += L.csum[3 * (size_t)(c0 + k) + 2]; }
    s_tot[t][0] = sx; s_tot[t][1] = sy; s_tot[t][2] = sz;
    __syncthreads();
    if (t == 0) {
        double px = 0.0, py = 0.0, pz = 0.0;
        for (int i = 0; i < 256; i++) { const double x = s_tot[i][0], y = s_tot[i][1], w = s_tot[i][2]; s_tot[i][0] = px; s_tot[i][1] = py; s_tot[i][2] = pz; px += x; py += y; pz += w; }
        L.status[j] = 0u;
    }
    __syncthreads();
    sx = s_tot[t][0]; sy = s_tot[t][1]; sz = s_tot[t][2];
    for (uint32_t k = a; k < z; k++) {
        const size_t c = (size_t)(c0 + k);
        L.cin[3 * c] = sx; L.cin[3 * c + 1] = sy; L.cin[3 * c + 2] = sz;
        sx += L.csum[3 * c]; sy += L.csum[3 * c + 1]; sz += L.csum[3 * c + 2];
    }
}
__global__ void __launch_bounds__(256) k_csum_check(CsumLevel L) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= L.totalChunks) return;
    const uint32_t j = csumNodeOf(L.chunkBase, L.count, c);
    const TopNode nd = L.nodes[j];
    double sx = L.cin[3 * (size_t)c], sy = L.cin[3 * (size_t)c + 1], sz = L.cin[3 * (size_t)c + 2];
    csumChunk(L, nd, c - L.chunkBase[j], sx, sy, sz);
    if (c + 1u < L.chunkBase[j + 1]) {
        const bool same = __double_as_longlong(sx) == __double_as_longlong(L.cin[3 * (size_t)(c + 1)]) && __double_as_longlong(sy) == __double_as_longlong(L.cin[3 * (size_t)(c + 1) + 1]) &&
                          __double_as_longlong(sz) == __double_as_longlong(L.cin[3 * (size_t)(c + 1) + 2]);
        if (!same) atomicOr(&L.status[j], 1u);
    } else {
        const double cnt = (double)(3u * (nd.e - nd.b));
        L.centres[3 * (size_t)j] = sx / cnt; L.centres[3 * (size_t)j + 1] = sy / cnt; L.centres[3 * (size_t)j + 2] = sz / cnt;
    }
}
// The serial chain: one workgroup per node, all lanes gather 256 triangles and lay their coordinates out as doubles, lanes 0..2 add them in
// order.  status != nullptr: only the nodes k_csum_check flagged (counted in *serialNodes).
__global__ void __launch_bounds__(256) k_top_sums(const uint32_t* __restrict__ order, const float4* __restrict__ triV, const TopNode* __restrict__ nodes, uint32_t count, double* __restrict__ centres,
                                                  const uint32_t* __restrict__ status, uint32_t* __restrict__ serialNodes) {
    __shared__ double s_v[256][9];
    if (status && status[blockIdx.x] == 0u) return;
    if (status && threadIdx.x == 0) atomicAdd(serialNodes, 1u);
    const TopNode nd = nodes[blockIdx.x];
    const int tid = threadIdx.x;
    const uint32_t nn = nd.e - nd.b;
    double sum = 0.0;
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0; float q2 = 0.f;
    auto load = [&](uint32_t base) { const uint32_t i = base + (uint32_t)tid; if (i < nn) { const size_t t = (size_t)order[nd.b + i]; q0 = triV[3 * t]; q1 = triV[3 * t + 1]; q2 = reinterpret_cast<const float*>(triV)[12 * t + 8]; } };
    load(0);
    for (uint32_t base = 0; base < nn; base += 256u) {
        double* w = s_v[tid];
        w[0] = (double)q0.x; w[1] = (double)q0.y; w[2] = (double)q0.z; w[3] = (double)q0.w; w[4] = (double)q1.x; w[5] = (double)q1.y; w[6] = (double)q1.z; w[7] = (double)q1.w; w[8] = (double)q2;
        __syncthreads();
        if (base + 256u < nn) load(base + 256u);                     // the next 256 triangles: two dependent gathers, in flight under the additions below
        if (tid < 3) {
            const uint32_t cnt = (nn - base < 256u) ? nn - base : 256u;
            uint32_t i = 0;
            for (; i + 8u <= cnt; i += 8u) {                         // the operands of eight triangles first, then their 24 dependent additions
                double a[24];
#pragma unroll
                for (int u = 0; u < 8; u++) { a[3 * u] = s_v[i + u][tid]; a[3 * u + 1] = s_v[i + u][3 + tid]; a[3 * u + 2] = s_v[i + u][6 + tid]; }
#pragma unroll
                for (int u = 0; u < 24; u++) sum += a[u];
            }
            for (; i < cnt; i++) { sum += s_v[i][tid]; sum += s_v[i][3 + tid]; sum += s_v[i][6 + tid]; }
        }
        __syncthreads();
    }
    if (tid < 3) centres[3 * (size_t)blockIdx.x + tid] = sum / (double)(3u * nn);
}
__global__ void __launch_bounds__(256) k_top_radius(const uint32_t* __restrict__ order, const float4* __restrict__ triV, const TopNode* __restrict__ nodes, uint32_t count, uint32_t n,
                                                    const double* __restrict__ centres, unsigned long long* __restrict__ r2bits) {
    const uint32_t i0 = 4u * (blockIdx.x * 256u + threadIdx.x);
    uint32_t node = 0xFFFFFFFFu;
    double cx = 0.0, cy = 0.0, cz = 0.0, r2 = 0.0;
    if (i0 < n) { node = topNodeOf(nodes, count, i0); cx = centres[3 * (size_t)node]; cy = centres[3 * (size_t)node + 1]; cz = centres[3 * (size_t)node + 2]; }
    const uint32_t nodeAtStart = node;
    for (uint32_t i = i0; i < i0 + 4u && i < n; i++) {
        while (i >= nodes[node].e) { atomicMax(&r2bits[node], (unsigned long long)__double_as_longlong(r2)); node++; r2 = 0.0; cx = centres[3 * (size_t)node]; cy = centres[3 * (size_t)node + 1]; cz = centres[3 * (size_t)node + 2]; }
        DevV3 v[3]; devTriVerts(triV, (int)order[i], v[0], v[1], v[2]);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const double dx = cx - (double)v[k].x, dy = cy - (double)v[k].y, dz = cz - (double)v[k].z;
            const double q = dx * dx + dy * dy + dz * dz;
            r2 = r2 < q ? q : r2;                                     // (never a NaN, never negative: the bits order like the values)
        }
    }
    const uint32_t first = __shfl(nodeAtStart, 0);
    if (__all(nodeAtStart == first && node == first && first != 0xFFFFFFFFu)) {      // the whole wave in one node: one atomic
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const double other = __shfl_xor(r2, o); r2 = r2 < other ? other : r2; }
        if ((threadIdx.x & 63) == 0) atomicMax(&r2bits[node], (unsigned long long)__double_as_longlong(r2));
    } else if (node != 0xFFFFFFFFu) atomicMax(&r2bits[node], (unsigned long long)__double_as_longlong(r2));
}
// the records of a level's nodes: their sphere into the parent's record, their own child references (both children are inner nodes)
__global__ void k_top_write(const TopNode* __restrict__ nodes, uint32_t count, const double* __restrict__ centres, const unsigned long long* __restrict__ r2bits, int haveSpheres,
                            double* __restrict__ sph, int* __restrict__ kids) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const TopNode nd = nodes[j];
    if (haveSpheres && nd.slot != 0xFFFFFFFFu) { double* o = sph + 4 * (size_t)nd.slot; o[0] = centres[3 * (size_t)j]; o[1] = centres[3 * (size_t)j + 1]; o[2] = centres[3 * (size_t)j + 2]; o[3] = sqrt(__longlong_as_double((long long)r2bits[j])); }
    const uint32_t mid = (nd.b + nd.e) >> 1;
    kids[2 * (size_t)nd.id] = nd.id + 1; kids[2 * (size_t)nd.id + 1] = nd.id + (int)(mid - nd.b);
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
constexpr uint32_t WIDE_SLAB_SPLIT = 192;          // children with more triangles get their slab from k_wide_slabs_big (a block each)
constexpr uint32_t WIDE_SLAB_TINY = 8;             // children with at most this many get it from the node's own thread; between: k_wide_slabs_mid, 16 lanes each
// the slab of triangles [rb, re) of the rank order around the decoded centre (dcx, dcy, dcz), by ONE lane: direction words and W as a half
__device__ __forceinline__ void wideSlabSerial(const float4* __restrict__ triV, const uint32_t* __restrict__ triAtRank, uint32_t rb, uint32_t re, double dcx, double dcy, double dcz,
                                               uint32_t& mxy, uint32_t& mzw) {
    double sx = 0, sy = 0, sz = 0;
    for (uint32_t k = rb; k < re; k++) {
        const uint32_t t = triAtRank[k];
        const float4 q0 = triV[3 * (size_t)t], q1 = triV[3 * (size_t)t + 1], q2 = triV[3 * (size_t)t + 2];
        const double ux = (double)q0.w - q0.x, uy = (double)q1.x - q0.y, uz = (double)q1.y - q0.z, vx = (double)q1.z - q0.x, vy = (double)q1.w - q0.y, vz = (double)q2.x - q0.z;
        sx += uy * vz - uz * vy; sy += uz * vx - ux * vz; sz += ux * vy - uy * vx;
    }
    const double len = sqrt(sx * sx + sy * sy + sz * sz);
    if (!(len > 1e-300 && len < 1e300)) return;
    auto snorm = [&](double v) { double q = rint(v / len * 32767.0); if (q < -32767.0) q = -32767.0; if (q > 32767.0) q = 32767.0; return (int)q; };
    const int ix = snorm(sx), iy = snorm(sy), iz = snorm(sz);
    const double dmx = (double)((float)ix * (1.0f / 32767.0f)), dmy = (double)((float)iy * (1.0f / 32767.0f)), dmz = (double)((float)iz * (1.0f / 32767.0f));   // the decoded direction
    double W = 0.0;
    for (uint32_t k = rb; k < re; k++) {
        const uint32_t t = triAtRank[k];
        const float4 q0 = triV[3 * (size_t)t], q1 = triV[3 * (size_t)t + 1], q2 = triV[3 * (size_t)t + 2];
        const double px[3] = {q0.x, q0.w, q1.z}, py[3] = {q0.y, q1.x, q1.w}, pz[3] = {q0.z, q1.y, q2.x};
        for (int j = 0; j < 3; j++) W = fmax(W, fabs(dmx * (px[j] - dcx) + dmy * (py[j] - dcy) + dmz * (pz[j] - dcz)));
    }
    const double Winfl = W * (1.0 + 1e-9) + 1e-300;
    float wf = (float)Winfl; if ((double)wf < Winfl) wf = nextafterf(wf, 3.0e38f);
    mxy = (uint32_t)(ix & 0xFFFF) | ((uint32_t)(iy & 0xFFFF) << 16); mzw = (uint32_t)(iz & 0xFFFF) | ((uint32_t)halfRoundedUp(wf) << 16);
}
// a slot of a work list for every lane that wants one: one atomic per wave (0xFFFFFFFF for the lanes that do not)
__device__ __forceinline__ uint32_t waveAppend(uint32_t* __restrict__ counter, bool want) {
    const unsigned long long m = __ballot(want);
    if (!m) return 0xFFFFFFFFu;
    const uint32_t lane = __lane_id(); const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader);
    return want ? base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)) : 0xFFFFFFFFu;
}
// One THREAD per binary inner node (nodes at odd depths are skipped).  Until round 5 sixteen lanes shared a node - for the loops over its
// children's triangles - and all sixteen ran the header's fp64 arithmetic (divisions, square roots, roundings) with it: four nodes per
// wave, 1.6 ms for the 1.31 M-triangle tree.  Now the thread does the header and the slabs of its TINY children (<= 8 triangles: nearly all
// of them) itself; larger children are listed for k_wide_slabs_mid (16 lanes each) / k_wide_slabs_big (a workgroup each), which fill in the
// slab words of the child's record (until then: no slab, W = +inf).
__global__ void __launch_bounds__(256) k_wide_nodes(const int2* __restrict__ kids, const double2* __restrict__ sph, const float4* __restrict__ triV, const uint32_t* __restrict__ triAtRank,
                                                    uint32_t numTriangles, uint4* __restrict__ wide, uint4* __restrict__ midList, uint32_t* __restrict__ midCount, uint32_t midCap,
                                                    uint4* __restrict__ bigList, uint32_t* __restrict__ bigCount, uint32_t bigCap) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
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
    out[0] = make_uint4(__float_as_uint(ox), __float_as_uint(oy), __float_as_uint(oz), __float_as_uint(scale));
    uint32_t refs[4] = {0u, 0u, 0u, 0u};
    for (int c = 0; c < 4; c++) {
        uint32_t qx = 0, qy = 0, qz = 0, mxy = 0, mzw = 0x7C00u << 16, cnt = 0; unsigned short rh = 0xFC00u;       // empty slot: radius -inf; no slab: W = +inf
        if (c < n) {
            auto quant = [&](double v, float o) { double q = rint((v - (double)o) / (double)scale); if (!(q >= 0.0)) q = 0.0; if (q > 65535.0) q = 65535.0; return (uint32_t)q; };
            qx = quant(cx[c], ox); qy = quant(cy[c], oy); qz = quant(cz[c], oz);
            const double dcx = (double)fmaf((float)qx, scale, ox), dcy = (double)fmaf((float)qy, scale, oy), dcz = (double)fmaf((float)qz, scale, oz);   // the decoded centre
            const double ex = dcx - cx[c], ey = dcy - cy[c], ez = dcz - cz[c];
            const double rInfl = (cr[c] + sqrt(ex * ex + ey * ey + ez * ez)) * (1.0 + 1e-9) + 1e-300;
            float rf = (float)rInfl; if ((double)rf < rInfl) rf = nextafterf(rf, 3.0e38f);
            rh = halfRoundedUp(rf);                      // +inf when the radius exceeds the half range: the child is then always visited
            refs[c] = (uint32_t)ref[c];
            cnt = re[c] - rb[c];
            if (cnt <= WIDE_SLAB_TINY) wideSlabSerial(triV, triAtRank, rb[c], re[c], dcx, dcy, dcz, mxy, mzw);
        }
        // (one atomic per wave and list: hundreds of thousands of single increments of one counter would cost what the kernel saves)
        const uint32_t atMid = waveAppend(midCount, cnt > WIDE_SLAB_TINY && cnt <= WIDE_SLAB_SPLIT), atBig = waveAppend(bigCount, cnt > WIDE_SLAB_SPLIT && cnt <= WIDE_SLAB_MAX);
        if (atMid != 0xFFFFFFFFu && atMid < midCap) midList[atMid] = make_uint4(i, (uint32_t)c, rb[c], re[c]);
        if (atBig != 0xFFFFFFFFu && atBig < bigCap) bigList[atBig] = make_uint4(i, (uint32_t)c, rb[c], re[c]);
        out[1 + c] = make_uint4(qx | (qy << 16), qz | ((uint32_t)rh << 16), mxy, mzw);
    }
    out[5] = make_uint4(refs[0], refs[1], refs[2], refs[3]);
}
// The slab of a child of 9 .. WIDE_SLAB_SPLIT triangles: sixteen lanes per listed (node, child) stride over its triangles for the two
// reductions.  Same construction as everywhere: direction = normalised sum of the area normals, quantised; W measured against the DECODED
// centre and direction, rounded up - conservative whatever the summation order.
__global__ void __launch_bounds__(256) k_wide_slabs_mid(const uint4* __restrict__ midList, uint32_t count, const float4* __restrict__ triV, const uint32_t* __restrict__ triAtRank,
                                                        uint4* __restrict__ wide) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t jb = gid >> 4, sub = gid & 15u;
    if (jb >= count) return;                               // (uniform per group of sixteen lanes)
    const uint4 job = midList[jb];
    uint4* out = wide + 8 * (size_t)job.x;
    const uint4 hdr = out[0], rec = out[1 + job.y];
    const float ox = __uint_as_float(hdr.x), oy = __uint_as_float(hdr.y), oz = __uint_as_float(hdr.z), scale = __uint_as_float(hdr.w);
    const double dcx = (double)fmaf((float)(rec.x & 0xFFFFu), scale, ox), dcy = (double)fmaf((float)(rec.x >> 16), scale, oy), dcz = (double)fmaf((float)(rec.y & 0xFFFFu), scale, oz);
    double sx = 0, sy = 0, sz = 0;
    for (uint32_t k = job.z + sub; k < job.w; k += 16u) {
        const uint32_t t = triAtRank[k];
        const float4 q0 = triV[3 * (size_t)t], q1 = triV[3 * (size_t)t + 1], q2 = triV[3 * (size_t)t + 2];
        const double ux = (double)q0.w - q0.x, uy = (double)q1.x - q0.y, uz = (double)q1.y - q0.z, vx = (double)q1.z - q0.x, vy = (double)q1.w - q0.y, vz = (double)q2.x - q0.z;
        sx += uy * vz - uz * vy; sy += uz * vx - ux * vz; sz += ux * vy - uy * vx;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { sx += __shfl_xor(sx, o, 16); sy += __shfl_xor(sy, o, 16); sz += __shfl_xor(sz, o, 16); }
    const double len = sqrt(sx * sx + sy * sy + sz * sz);
    if (!(len > 1e-300 && len < 1e300)) return;           // (the sixteen lanes hold the same sums)
    auto snorm = [&](double v) { double q = rint(v / len * 32767.0); if (q < -32767.0) q = -32767.0; if (q > 32767.0) q = 32767.0; return (int)q; };
    const int ix = snorm(sx), iy = snorm(sy), iz = snorm(sz);
    const double dmx = (double)((float)ix * (1.0f / 32767.0f)), dmy = (double)((float)iy * (1.0f / 32767.0f)), dmz = (double)((float)iz * (1.0f / 32767.0f));
    double W = 0.0;
    for (uint32_t k = job.z + sub; k < job.w; k += 16u) {
        const uint32_t t = triAtRank[k];
        const float4 q0 = triV[3 * (size_t)t], q1 = triV[3 * (size_t)t + 1], q2 = triV[3 * (size_t)t + 2];
        const double px[3] = {q0.x, q0.w, q1.z}, py[3] = {q0.y, q1.x, q1.w}, pz[3] = {q0.z, q1.y, q2.x};
        for (int j = 0; j < 3; j++) W = fmax(W, fabs(dmx * (px[j] - dcx) + dmy * (py[j] - dcy) + dmz * (pz[j] - dcz)));
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) W = fmax(W, __shfl_xor(W, o, 16));
    if (sub == 0u) {
        const double Winfl = W * (1.0 + 1e-9) + 1e-300;
        float wf = (float)Winfl; if ((double)wf < Winfl) wf = nextafterf(wf, 3.0e38f);
        out[1 + job.y] = make_uint4(rec.x, rec.y, (uint32_t)(ix & 0xFFFF) | ((uint32_t)(iy & 0xFFFF) << 16), (uint32_t)(iz & 0xFFFF) | ((uint32_t)halfRoundedUp(wf) << 16));
    }
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

static int buildTreeOnDevice(sdfhip_mesh* mesh, hipStream_t st);
static int installBvh(sdfhip_mesh* mesh, const double* sph, const int* kids, int where, bool validate = false, const PlannedBvh* hybrid = nullptr, bool onDevice = false) {
    const uint32_t T = mesh->numTriangles;
    const uint64_t nn = T - 1;
    const size_t nSph = 8 * (size_t)(nn ? nn : 1), nKids = 2 * (size_t)(nn ? nn : 1);
    SDF_HIP_CHECK(hipSetDevice(mesh->ctx->device));
    hipStream_t st = mesh->ctx->stream;
    AllocScope allocScope(st);
    const hipMemcpyKind kind = where == SDFHIP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    SDF_TRY(mesh->dBvhSph.reserve(nSph)); SDF_TRY(mesh->dBvhKids.reserve(nKids)); SDF_TRY(mesh->dTriVerts.reserve(12ull * T));
    if (!hybrid && !onDevice) {
        SDF_HIP_CHECK(hipMemcpyAsync(mesh->dBvhSph.p, sph, nSph * sizeof(double), kind, st));
        SDF_HIP_CHECK(hipMemcpyAsync(mesh->dBvhKids.p, kids, nKids * sizeof(int), kind, st));
    }
    k_tri_verts<<<gridFor(12ull * T, 256), 256, 0, st>>>(mesh->dVerts.p, mesh->dIdx.p, T, mesh->dTriVerts.p);
    if (hybrid) SDF_TRY(finishOnDevice(mesh, *hybrid, st));
    if (onDevice) SDF_TRY(buildTreeOnDevice(mesh, st));
    {
        SDF_TRY(mesh->dBvhSph32.reserve(nSph));
        k_sph32<<<gridFor(nSph, 256), 256, 0, st>>>(mesh->dBvhSph.p, nSph, mesh->dBvhSph32.p);
    }
    SDF_TRY(mesh->dTriRank.reserve(T));
    k_tri_ranks<<<gridFor(T, 256), 256, 0, st>>>(reinterpret_cast<const int2*>(mesh->dBvhKids.p), T, mesh->dTriRank.p);
    DevBuf<uint32_t> triAtRank;
    SDF_TRY(triAtRank.reserve(T));
    k_tri_at_rank<<<gridFor(T, 256), 256, 0, st>>>(mesh->dTriRank.p, T, triAtRank.p);
    SDF_TRY(mesh->dBvhWide.reserve(32 * (size_t)(nn ? nn : 1)));
    DevBuf<uint32_t> bigList, midList, listCount;        // children whose slab the node's thread leaves to sixteen lanes / to a workgroup
    const size_t bigCap = (size_t)T / WIDE_SLAB_SPLIT * 16 + 64, midCap = (size_t)T / WIDE_SLAB_TINY * 4 + 64;      // (sizes of a level pair's children x level pairs that fall in the class)
    SDF_TRY(bigList.reserve(4 * bigCap)); SDF_TRY(midList.reserve(4 * midCap)); SDF_TRY(listCount.reserve(2));
    SDF_HIP_CHECK(hipMemsetAsync(listCount.p, 0, 8, st));
    k_wide_nodes<<<gridFor(T, 256), 256, 0, st>>>(reinterpret_cast<const int2*>(mesh->dBvhKids.p), reinterpret_cast<const double2*>(mesh->dBvhSph.p), reinterpret_cast<const float4*>(mesh->dTriVerts.p),
                                                 triAtRank.p, T, reinterpret_cast<uint4*>(mesh->dBvhWide.p), reinterpret_cast<uint4*>(midList.p), listCount.p, (uint32_t)midCap,
                                                 reinterpret_cast<uint4*>(bigList.p), listCount.p + 1, (uint32_t)bigCap);
    {
        uint32_t nList[2] = {0, 0};
        SDF_TRY(readBackWords(st, listCount.p, nullptr, 2, nList));
        SDF_REQUIRE(nList[0] <= midCap && nList[1] <= bigCap, "wide-node work list overflow");
        if (nList[0]) k_wide_slabs_mid<<<gridFor(16ull * nList[0], 256), 256, 0, st>>>(reinterpret_cast<const uint4*>(midList.p), nList[0], reinterpret_cast<const float4*>(mesh->dTriVerts.p), triAtRank.p, reinterpret_cast<uint4*>(mesh->dBvhWide.p));
        if (nList[1]) k_wide_slabs_big<<<nList[1], 256, 0, st>>>(reinterpret_cast<const uint4*>(bigList.p), nList[1], reinterpret_cast<const float4*>(mesh->dTriVerts.p), triAtRank.p, reinterpret_cast<uint4*>(mesh->dBvhWide.p));
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
// given back are kept (up to 512 MB) and handed to the next plan as they are.
struct PlannerBlocks {
    std::mutex m; std::vector<std::pair<void*, size_t>> idle, live; size_t idleBytes = 0;
    static PlannerBlocks& get() { static PlannerBlocks* p = new PlannerBlocks(); return *p; }
    static size_t cap() { return (size_t)512 << 20; }
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
    madvise(p, rounded, MADV_HUGEPAGE);
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
        uint32_t v = (uint32_t)atoi(e); if (v == 0u) return 0u; if (v == 1u) v = 4096u;      // =0: off; =1: the default size; =N: ranges of at most N triangles
        return v > kDevSubtreeMaxLimit ? kDevSubtreeMaxLimit : (v < 32u ? 32u : v);
    }();
    return offload;
}
namespace sdfhip {
bool bvhBuildOnDevice() {
    static const bool v = [] { const char* e = getenv("SDFHIP_BVH_BUILD"); return e == nullptr || strcmp(e, "host") != 0; }();      // the default; =host: the host planner
    return v;
}
void startEarlyBvhPlan(sdfhip_mesh* mesh) {
    if (mesh->early.th.joinable() || mesh->early.plan || bvhBuildOnDevice()) return;      // (built on the device: nothing to plan)
    mesh->early.drop = [](void* p) { delete static_cast<PlannedBvh*>(p); };
    mesh->early.th = std::thread([mesh]() {
        try { mesh->early.plan = new PlannedBvh(planBvhHost(mesh->hVerts.data(), mesh->hIdx.data(), mesh->numTriangles, bvhOffloadMax())); }
        catch (...) { mesh->early.plan = nullptr; }          // (out of memory: sdfhip_mesh_build_bvh plans again and reports)
    });
}
}

// The whole tree on the device (see k_gs_first_prepare); SDFHIP_BVH_BUILD=host plans on the host instead.  SDFHIP_E_UNSUPPORTED: a long range ran out of introsort's depth
// limit or a work list overflowed — the caller plans on the host.
static int buildTreeOnDevice(sdfhip_mesh* mesh, hipStream_t st) {
    const uint32_t T = mesh->numTriangles;
    uint32_t S = bvhOffloadMax(); if (S == 0) S = 4096u;
    // a range leaves the rounds over global memory for k_sort_parts (LDS) at this length: a round costs ~50 us whatever its ranges, a workgroup's
    // serial tail grows with the part (SDFHIP_BVH_PART; default 4096 since round 4: 72 instead of 104 rounds at 327 680 triangles, 116 instead
    // of 156 at 1.31 M - build_bvh 7.4 -> 6.8 ms and 16.0 -> 15.4 ms, profiles/r04_bvh_persistent_rounds.txt)
    uint32_t partMax = 4096u; if (const char* e = getenv("SDFHIP_BVH_PART")) { const uint32_t v = (uint32_t)atoi(e); if (v >= 32u) partMax = v; }
    if (partMax > S) partMax = S;
    const bool timing = getenv("SDFHIP_TIMING") != nullptr;
    const double t0 = nowSeconds();
    const float4* triV = reinterpret_cast<const float4*>(mesh->dTriVerts.p);
    // the levels sorted in global memory: while some node is longer than S
    std::vector<std::vector<TopNode>> levels;
    std::vector<TopNode> cur{TopNode{0, 0u, T, 0xFFFFFFFFu}};
    for (;;) {
        uint32_t longest = 0; for (const TopNode& nd : cur) longest = std::max(longest, nd.e - nd.b);
        if (longest <= S) break;
        std::vector<TopNode> next; next.reserve(2 * cur.size());
        for (const TopNode& nd : cur) {
            const uint32_t mid = (nd.b + nd.e) >> 1;
            next.push_back(TopNode{nd.id + 1, nd.b, mid, 2u * (uint32_t)nd.id});
            next.push_back(TopNode{nd.id + (int)(mid - nd.b), mid, nd.e, 2u * (uint32_t)nd.id + 1u});
        }
        levels.push_back(std::move(cur)); cur = std::move(next);
    }
    const size_t nTop = levels.size();                  // cur: the subtrees' roots
    size_t tableNodes = 0; std::vector<size_t> levelAt(nTop + 1, 0);
    for (size_t l = 0; l < nTop; l++) { levelAt[l] = tableNodes; tableNodes += levels[l].size(); }
    levelAt[nTop] = tableNodes;
    const uint32_t maxTasks = 2u * (T / partMax) + 16u, maxParts = T / 16u + 16u, maxTiny = T / 2u + 16u;
    const uint32_t maxChunks = T / kGsChunk + maxTasks + 1u;
    DevBuf<KeyTri> K; DevBuf<uint32_t> Ll, Rl, snaps, box, ctr, chunkBase, cutAt, cntL, cntR, chunkTask, dFail; DevBuf<float> pk; DevBuf<GTask> tasks, parts, tiny; DevBuf<TopNode> dNodes;
    DevBuf<double> centres; DevBuf<unsigned long long> r2, dClk; DevBuf<BvhTask> dTasks; DevBuf<BvhDevNode> dScratch;
    DevBuf<uint32_t> sumBase, sumStatus; DevBuf<double> csum, cin; std::vector<uint32_t> sumBaseH; std::vector<size_t> sumChunkAt;
    SDF_TRY(K.reserve(T)); SDF_TRY(snaps.reserve((size_t)T * (nTop ? nTop : 1))); SDF_TRY(ctr.reserve(8)); SDF_TRY(dFail.reserve(1));
    if (nTop) {
        SDF_TRY(Ll.reserve(T)); SDF_TRY(Rl.reserve(T)); SDF_TRY(box.reserve(6 * tableNodes)); SDF_TRY(chunkBase.reserve(maxTasks + 1)); SDF_TRY(cutAt.reserve(maxTasks));
        SDF_TRY(cntL.reserve(maxChunks)); SDF_TRY(cntR.reserve(maxChunks)); SDF_TRY(chunkTask.reserve(maxChunks)); SDF_TRY(pk.reserve(maxTasks)); SDF_TRY(tasks.reserve(2 * (size_t)maxTasks)); SDF_TRY(parts.reserve(maxParts));
        SDF_TRY(tiny.reserve(maxTiny)); SDF_TRY(dNodes.reserve(tableNodes)); SDF_TRY(centres.reserve(3 * tableNodes)); SDF_TRY(r2.reserve(tableNodes));
        std::vector<TopNode> flat; flat.reserve(tableNodes);
        for (size_t l = 0; l < nTop; l++) flat.insert(flat.end(), levels[l].begin(), levels[l].end());
        SDF_HIP_CHECK(hipMemcpyAsync(dNodes.p, flat.data(), sizeof(TopNode) * tableNodes, hipMemcpyHostToDevice, st));
        // the chunks of the parallel centre sums (k_csum_*): per level the first chunk of every node (+ the end), numbered through the levels
        sumBaseH.assign(tableNodes + nTop + 1, 0u); sumChunkAt.assign(nTop + 1, 0);
        size_t chunksAll = 0;
        for (size_t l = 0; l < nTop; l++) {
            sumChunkAt[l] = chunksAll;
            uint32_t run = 0;
            for (size_t j = 0; j < levels[l].size(); j++) { sumBaseH[levelAt[l] + l + j] = run; run += (levels[l][j].e - levels[l][j].b + kSumChunk - 1u) / kSumChunk; }
            sumBaseH[levelAt[l] + l + levels[l].size()] = run;
            chunksAll += run;
        }
        sumChunkAt[nTop] = chunksAll;
        SDF_TRY(sumBase.reserve(sumBaseH.size())); SDF_TRY(csum.reserve(3 * chunksAll + 3)); SDF_TRY(cin.reserve(3 * chunksAll + 3)); SDF_TRY(sumStatus.reserve(tableNodes + 1));
        SDF_HIP_CHECK(hipMemcpyAsync(sumBase.p, sumBaseH.data(), 4 * sumBaseH.size(), hipMemcpyHostToDevice, st));
        SDF_HIP_CHECK(hipMemsetAsync(sumStatus.p + tableNodes, 0, 4, st));          // [tableNodes]: nodes summed by the serial chain
        SDF_HIP_CHECK(hipStreamSynchronize(st));        // (flat goes out of scope)
        k_top_init<<<gridFor(tableNodes, 256), 256, 0, st>>>(box.p, r2.p, (uint32_t)tableNodes);
    }
    SDF_HIP_CHECK(hipMemsetAsync(ctr.p, 0, 32, st)); SDF_HIP_CHECK(hipMemsetAsync(dFail.p, 0, 4, st));
    k_key_init<<<gridFor(T, 256), 256, 0, st>>>(K.p, T);
    const size_t lds = KeyArr::bytes(S) + 2 * IdxArr::bytes(S) + 2 * 3 * 4 * kDevSortTasks;
    {
        static bool raised = false;
        if (!raised) {
            const int most = (int)(KeyArr::bytes(kDevSubtreeMaxLimit) + 2 * IdxArr::bytes(kDevSubtreeMaxLimit) + 2 * 3 * 4 * kDevSortTasks);
            SDF_TRY(raiseSubtreeLds(most));
            SDF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sort_parts), hipFuncAttributeMaxDynamicSharedMemorySize, most));
            raised = true;
        }
    }
    struct EventHolder { hipEvent_t e = nullptr; ~EventHolder() { if (e) (void)hipEventDestroy(e); } };
    std::vector<EventHolder> sorted(nTop);
    // two side streams of the context: level 1 alone on the second (its two chains are as long as all deeper levels' together)
    struct Side { hipStream_t s; } side{nullptr}, side1{nullptr};
    const bool useSide = true;
    if (nTop && useSide) {
        for (hipStream_t& s : mesh->ctx->bvhSide) if (!s) SDF_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        side.s = mesh->ctx->bvhSide[0]; side1.s = mesh->ctx->bvhSide[1];
    } else { side.s = st; side1.s = st; }
    // whatever happens below, nothing of this call may still run on the side streams when its buffers are released
    struct SideGuard { hipStream_t a, b; ~SideGuard() { if (a) (void)hipStreamSynchronize(a); if (b) (void)hipStreamSynchronize(b); } } sideGuard{useSide ? side.s : nullptr, useSide ? side1.s : nullptr};
    // ctr: [0], [1] = pending ranges of this / the next round (alternating), [2] = parts for k_sort_parts, [3] = for k_sort_tiny, [4] = flags
    uint32_t hostCtr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t rounds = 0, groups = 0;
    for (size_t l = 0; l < nTop; l++) {
        const std::vector<TopNode>& nodes = levels[l];
        const uint32_t count = (uint32_t)nodes.size();
        const TopNode* dN = dNodes.p + levelAt[l];
        k_top_aabb<<<gridFor(T, 1024), 256, 0, st>>>(K.p, triV, dN, count, T, box.p + 6 * levelAt[l]);
        k_top_keys<<<gridFor(T, 256), 256, 0, st>>>(K.p, triV, dN, count, T, box.p + 6 * levelAt[l]);
        // the level's nodes are the first round's ranges
        SDF_REQUIRE(count <= maxTasks, "internal: more nodes on a level than ranges provided for");
        int curBuf = 0;
        hostCtr[0] = count; hostCtr[1] = 0; hostCtr[2] = 0; hostCtr[3] = 0;
        uint32_t pending = count;
        // The host learns a round's outcome only by waiting for it: rounds are queued in groups, sized for the most ranges they can have (a
        // range leaves at most two); a round without ranges costs three empty launches.  The first group of a level is as long as its longest
        // node needs with halving cuts (+ 2: ranges of at most partMax keys leave the rounds), the groups behind it take three rounds each
        // (rounds 1-4: always four at a time, 156 rounds for the 9 levels of the 1.31 M mesh, of which the levels needed about 60).
        uint32_t longest = 0; for (const TopNode& nd : nodes) longest = std::max(longest, nd.e - nd.b);
        int groupRounds = 1; for (uint32_t m = longest; m > partMax; m >>= 1) groupRounds++;
        groupRounds++;             // (one more: 36 -> 23 read-backs for the 9 levels of the 1.31 M mesh, same number of rounds run)
        const uint32_t nchCap = longest / kGsChunk + 2u;            // chunks of the level's longest range (ranges only get shorter)
        if (nchCap > kGsMaxRangeChunks) { if (timing) fprintf(stderr, "[sdfhip] bvh on the device: a range of %u triangles is more than the round kernels hold prefix sums for\n", longest); return SDFHIP_E_UNSUPPORTED; }
        auto roundOf = [&](int buf) { return GsRound{K.p, Ll.p, Rl.p, tasks.p + (size_t)buf * maxTasks, ctr.p + buf, maxTasks, pk.p, chunkBase.p, cutAt.p, cntL.p, cntR.p, chunkTask.p}; };
        k_gs_first_prepare<<<1, 1024, 0, st>>>(dN, count, tasks.p, ctr.p, roundOf(curBuf), ctr.p + (curBuf ^ 1), ctr.p + 4);      // the level's first round; every other round is prepared by the round before it
        while (pending > 0) {
            uint32_t bound = pending;
            const int nowRounds = groupRounds;
            for (int q = 0; q < nowRounds; q++) {
                const GsRound R = roundOf(curBuf), Rn = roundOf(curBuf ^ 1);
                const unsigned chunkGrid = (unsigned)(T / kGsChunk + bound + 1u);
                k_gs_mark<<<chunkGrid, 256, 0, st>>>(R);
                k_gs_swap<<<8u * chunkGrid, 256, 8u * (nchCap + 1u), st>>>(R, nchCap);
                k_gs_emit_prepare<<<1, 1024, 0, st>>>(R, Rn, ctr.p + curBuf, parts.p, ctr.p + 2, maxParts, tiny.p, ctr.p + 3, maxTiny, partMax, ctr.p + 4);
                curBuf ^= 1;
                rounds++;
                bound = (bound > maxTasks / 2u) ? maxTasks : 2u * bound;
            }
            groupRounds = 3;
            groups++;
            SDF_TRY(readBackWords(st, ctr.p, nullptr, 5, hostCtr));
            if (hostCtr[4]) { if (timing) fprintf(stderr, "[sdfhip] bvh on the device: gave up at level %zu (flags %u: 1 = a work list overflowed, 2 = a long range out of introsort's depth)\n", l, hostCtr[4]); return SDFHIP_E_UNSUPPORTED; }
            pending = hostCtr[curBuf];
        }
        if (hostCtr[2]) k_sort_parts<<<hostCtr[2], 1024, KeyArr::bytes(partMax) + 2 * IdxArr::bytes(partMax) + 2 * 3 * 4 * kDevSortTasks, st>>>(K.p, parts.p, partMax, ctr.p + 4);
        if (hostCtr[3]) k_sort_tiny<<<gridFor(hostCtr[3], 64), 64, 0, st>>>(K.p, tiny.p, hostCtr[3]);
        uint32_t* snap = snaps.p + (size_t)T * l;
        k_top_snapshot<<<gridFor(T, 256), 256, 0, st>>>(K.p, T, snap, dN, count, mesh->dBvhKids.p);
        SDF_HIP_CHECK(hipGetLastError());
        // behind this level's sort, on the side stream: centres (sums in this order), radii and sphere records of the NEXT level's nodes
        if (l + 1 < nTop) {
            hipStream_t ss = (l == 0) ? side1.s : side.s;
            if (useSide) {
                SDF_HIP_CHECK(hipEventCreateWithFlags(&sorted[l].e, hipEventDisableTiming));
                SDF_HIP_CHECK(hipEventRecord(sorted[l].e, st));
                SDF_HIP_CHECK(hipStreamWaitEvent(ss, sorted[l].e, 0));
            }
            const uint32_t nc = (uint32_t)levels[l + 1].size(); const size_t at = levelAt[l + 1];
            // centres: in parallel and verified, the serial chain for the nodes whose verification failed (k_csum_*, k_top_sums)
            const uint32_t chunks = (uint32_t)(sumChunkAt[l + 2 <= nTop ? l + 2 : nTop] - sumChunkAt[l + 1]);
            const CsumLevel CL{snap, triV, dNodes.p + at, nc, sumBase.p + at + (l + 1), chunks, csum.p + 3 * sumChunkAt[l + 1], cin.p + 3 * sumChunkAt[l + 1], sumStatus.p + at, centres.p + 3 * at};
            k_csum_local<<<gridFor(chunks, 256), 256, 0, ss>>>(CL);
            k_csum_scan<<<nc, 256, 0, ss>>>(CL);
            k_csum_check<<<gridFor(chunks, 256), 256, 0, ss>>>(CL);
            k_top_sums<<<nc, 256, 0, ss>>>(snap, triV, dNodes.p + at, nc, centres.p + 3 * at, sumStatus.p + at, sumStatus.p + tableNodes);
            k_top_radius<<<gridFor(T, 1024), 256, 0, ss>>>(snap, triV, dNodes.p + at, nc, T, centres.p + 3 * at, r2.p + at);
            k_top_write<<<gridFor(nc, 256), 256, 0, ss>>>(dNodes.p + at, nc, centres.p + 3 * at, r2.p + at, 1, mesh->dBvhSph.p, mesh->dBvhKids.p);
        }
    }
    const double tTop = nowSeconds();
    // every remaining range: one workgroup each
    const size_t nt = cur.size();
    std::vector<BvhTask> ht(nt);
    for (size_t i = 0; i < nt; i++) ht[i] = BvhTask{cur[i].id, cur[i].b, cur[i].e, cur[i].slot};
    const uint32_t* order = snaps.p + (size_t)T * (nTop ? nTop - 1 : 0);
    if (!nTop) k_top_snapshot<<<gridFor(T, 256), 256, 0, st>>>(K.p, T, snaps.p, nullptr, 0u, nullptr);
    SDF_TRY(dTasks.reserve(nt)); SDF_TRY(dScratch.reserve(nt * 2 * (S / 2 + 1)));
    SDF_HIP_CHECK(hipMemcpyAsync(dTasks.p, ht.data(), sizeof(BvhTask) * nt, hipMemcpyHostToDevice, st));
    unsigned long long clk[4] = {0, 0, 0, 0};
    if (timing) { SDF_TRY(dClk.reserve(4)); SDF_HIP_CHECK(hipMemsetAsync(dClk.p, 0, 32, st)); }
    launchSubtrees((unsigned)nt, lds, st, dTasks.p, order, triV, mesh->dBvhSph.p, mesh->dBvhKids.p, dScratch.p, dFail.p, S, timing ? dClk.p : nullptr);
    SDF_HIP_CHECK(hipGetLastError());
    uint32_t failed = 0;
    SDF_HIP_CHECK(hipMemcpyAsync(&failed, dFail.p, 4, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipMemcpyAsync(hostCtr, ctr.p, 20, hipMemcpyDeviceToHost, st));
    if (timing) SDF_HIP_CHECK(hipMemcpyAsync(clk, dClk.p, 32, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    const double tSub = nowSeconds();
    if (nTop && useSide) { SDF_HIP_CHECK(hipStreamSynchronize(side.s)); SDF_HIP_CHECK(hipStreamSynchronize(side1.s)); }
    if (timing && nTop) {
        uint32_t serial = 0;
        if (hipMemcpy(&serial, sumStatus.p + tableNodes, 4, hipMemcpyDeviceToHost) != hipSuccess) (void)hipGetLastError();
        fprintf(stderr, "[sdfhip] bvh on the device: centre sums of %zu nodes in parallel (verified chunk by chunk), %u of them redone by the serial chain\n", tableNodes - levels[0].size(), serial);
    }
    if (timing) fprintf(stderr, "[sdfhip] bvh on the device: %zu levels in global memory (%u rounds in %u groups) %.4f s, %zu ranges of <= %u in LDS %.4f s (block 0: levels prepared by the workgroup %.3f ms, by a lane per node %.3f, sorts %.3f, children %.3f), waiting for the centre sums %.4f s\n",
                        nTop, rounds, groups, tTop - t0, nt, S, tSub - tTop, clk[0] * 1e-5, clk[1] * 1e-5, clk[2] * 1e-5, clk[3] * 1e-5, nowSeconds() - tSub);
    if (failed || hostCtr[4]) return SDFHIP_E_UNSUPPORTED;
    return SDFHIP_OK;
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
    const bool timing = getenv("SDFHIP_TIMING") != nullptr;
    DevBuf<unsigned long long> dClk; unsigned long long clk[4] = {0, 0, 0, 0};
    if (timing) { SDF_TRY(dClk.reserve(4)); SDF_HIP_CHECK(hipMemsetAsync(dClk.p, 0, 32, st)); }
    const double tUp = nowSeconds();
    if (timing) SDF_HIP_CHECK(hipStreamSynchronize(st));
    const double tKer = nowSeconds();
    k_bvh_scatter_top<<<gridFor(nh, 256), 256, 0, st>>>(dIds.p, dS8.p, dK2.p, dOwn.p, (uint32_t)nh, mesh->dBvhSph.p, mesh->dBvhKids.p);
    const size_t lds = KeyArr::bytes(kDevSubtreeMax) + 2 * IdxArr::bytes(kDevSubtreeMax) + 2 * 3 * 4 * kDevSortTasks;
    static bool ldsRaised = false;
    if (!ldsRaised && lds > (48u << 10)) { SDF_TRY(raiseSubtreeLds((int)(KeyArr::bytes(kDevSubtreeMaxLimit) + 2 * IdxArr::bytes(kDevSubtreeMaxLimit) + 2 * 3 * 4 * kDevSortTasks))); ldsRaised = true; }
    launchSubtrees((unsigned)nt, lds, st, dTasks.p, dOrder.p, reinterpret_cast<const float4*>(mesh->dTriVerts.p), mesh->dBvhSph.p, mesh->dBvhKids.p, dScratch.p, dFail.p, kDevSubtreeMax, timing ? dClk.p : nullptr);
    SDF_HIP_CHECK(hipGetLastError());
    uint32_t failed = 0;
    if (timing) SDF_HIP_CHECK(hipMemcpyAsync(clk, dClk.p, 32, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipMemcpyAsync(&failed, dFail.p, 4, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));         // (the staging vectors above are released here)
    if (timing) fprintf(stderr, "[sdfhip] bvh device subtrees: %zu ranges of <= %u, uploads %.4f s, kernels %.4f s; block 0: workgroup-prepared levels %.3f ms, lane-per-node levels %.3f ms, sorts %.3f ms, children %.3f ms\n",
                        nt, kDevSubtreeMax, tKer - tUp, nowSeconds() - tKer, clk[0] * 1e-5, clk[1] * 1e-5, clk[2] * 1e-5, clk[3] * 1e-5);
    if (failed && getenv("SDFHIP_TIMING")) fprintf(stderr, "[sdfhip] bvh: device subtrees gave up (reason bits %u: 1 / 4 = sort task list full, 2 = introsort depth limit): planning on the host\n", failed);
    return failed ? SDFHIP_E_UNSUPPORTED : SDFHIP_OK;
}

// the mesh's arrays on the host (the host planner's input): kept from the mesh's creation when the host planner is the builder, fetched
// from the device when the device builder hands a tree over to it
static int hostArrays(sdfhip_mesh* mesh) {
    if (!mesh->hVerts.empty() && !mesh->hIdx.empty()) return SDFHIP_OK;
    SDF_HIP_CHECK(hipSetDevice(mesh->ctx->device));        // the caller's current device need not be the context's
    hipStream_t st = mesh->ctx->stream;
    mesh->hVerts.resize(3ull * mesh->numVertices); mesh->hIdx.resize(3ull * mesh->numTriangles);
    SDF_HIP_CHECK(hipMemcpyAsync(mesh->hVerts.data(), mesh->dVerts.p, 4 * mesh->hVerts.size(), hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipMemcpyAsync(mesh->hIdx.data(), mesh->dIdx.p, 4 * mesh->hIdx.size(), hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    return SDFHIP_OK;
}

extern "C" {

int sdfhip_mesh_build_bvh(sdfhip_mesh* mesh, double* seconds) {
    SDF_API_BEGIN
    SDF_REQUIRE(mesh != nullptr, "mesh is NULL");
    std::lock_guard<std::recursive_mutex> building(mesh->ctx->buildLock);
    if (mesh->hasBvh) { if (seconds) *seconds = 0.0; return SDFHIP_OK; }
    SDF_HIP_CHECK(hipSetDevice(mesh->ctx->device));
    { int depth = 1; while ((1ull << (depth - 1)) < mesh->numTriangles) depth++; SDF_REQUIRE(depth + 1 <= BVH_STACK, "mesh too large for the traversal stack"); }
    const double t0 = nowSeconds();
    const uint32_t offload = bvhOffloadMax();
    if (bvhBuildOnDevice() && mesh->numTriangles >= 64u && !mesh->early.th.joinable() && !mesh->early.plan) {
        const int rcDev = installBvh(mesh, nullptr, nullptr, SDFHIP_HOST, false, nullptr, true);
        if (rcDev == SDFHIP_OK) {
            if (getenv("SDFHIP_TIMING")) fprintf(stderr, "[sdfhip] bvh: built on the device, %.3f s with the derived records\n", nowSeconds() - t0);
            if (seconds) *seconds = nowSeconds() - t0;
            return SDFHIP_OK;
        }
        if (rcDev != SDFHIP_E_UNSUPPORTED) return rcDev;
    }
    PlannedBvh P;
    if (mesh->early.th.joinable()) mesh->early.th.join();        // a plan started under the mesh preparation (sdfhip_mesh_create_opt)
    if (mesh->early.plan) { P = std::move(*static_cast<PlannedBvh*>(mesh->early.plan)); delete static_cast<PlannedBvh*>(mesh->early.plan); mesh->early.plan = nullptr; }
    else { SDF_TRY(hostArrays(mesh)); P = planBvhHost(mesh->hVerts.data(), mesh->hIdx.data(), mesh->numTriangles, offload); }
    double tPlanned = nowSeconds();
    int rc = installBvh(mesh, P.sph.get(), P.kids.get(), SDFHIP_HOST, false, P.tasks.empty() ? nullptr : &P);
    if (rc == SDFHIP_E_UNSUPPORTED && !P.tasks.empty()) {         // a device sort met introsort's depth limit: libstdc++'s heap sort decides that order
        SDF_TRY(hostArrays(mesh));
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

static int nearestStats(sdfhip_mesh* mesh, const float* xyz, uint64_t n, uint32_t* out4, bool preseed) {
    SDF_API_BEGIN
    SDF_REQUIRE(mesh && xyz && out4, "NULL argument");
    SDF_HIP_CHECK(hipSetDevice(mesh->ctx->device));
    SDF_TRY(sdfhip_mesh_ensure_bvh(mesh));
    hipStream_t st = mesh->ctx->stream;
    DevBuf<float> dp; DevBuf<uint32_t> dout;
    SDF_TRY(dp.reserve(3 * n)); SDF_TRY(dout.reserve(4 * n));
    SDF_HIP_CHECK(hipMemcpyAsync(dp.p, xyz, sizeof(float) * 3 * n, hipMemcpyHostToDevice, st));
    if (nearestExactOnly()) k_nearest_stats<<<gridFor(n, 128), 128, 0, st>>>(meshBvh(mesh), dp.p, n, dout.p);
    else {
        // the two-phase search's own counters: [id, wide-node expansions, wave iterations alive, triangle evaluations]; with
        // preseed of a second run seeded with the first run's answers (the fewest visits any visiting order can need)
        std::lock_guard<std::recursive_mutex> building(mesh->ctx->buildLock);
        SDF_REQUIRE(n < (1ull << 22), "stats: at most 4 M points");
        int depth = 1; while ((1ull << (depth - 1)) < mesh->numTriangles) depth++;
        DevBuf<uint32_t> ids; SDF_TRY(ids.reserve(n));
        SDF_HIP_CHECK(hipMemsetAsync(dout.p, 0, sizeof(uint32_t) * 4 * n, st));
        SDF_TRY(nearestTwoPhase(st, meshBvh(mesh), dp.p, (uint32_t)n, ids.p, mesh->ctx->nearScratch, depth + 2, 0u, 1u, dout.p, nullptr));
        if (preseed) {
            SDF_HIP_CHECK(hipMemsetAsync(dout.p, 0, sizeof(uint32_t) * 4 * n, st));
            DevBuf<uint32_t> ids2; SDF_TRY(ids2.reserve(n));
            SDF_TRY(nearestTwoPhase(st, meshBvh(mesh), dp.p, (uint32_t)n, ids2.p, mesh->ctx->nearScratch, depth + 2, 0u, 1u, dout.p, ids.p));
        }
        SDF_HIP_CHECK(hipMemcpy2DAsync(dout.p, 16, ids.p, 4, 4, n, hipMemcpyDeviceToDevice, st));
        SDF_HIP_CHECK(hipStreamSynchronize(st));
    }
    SDF_HIP_CHECK(hipGetLastError());
    SDF_HIP_CHECK(hipMemcpyAsync(out4, dout.p, sizeof(uint32_t) * 4 * n, hipMemcpyDeviceToHost, st));
    SDF_HIP_CHECK(hipStreamSynchronize(st));
    return SDFHIP_OK;
    SDF_API_END
}

int sdfhip_mesh_nearest_stats(sdfhip_mesh* mesh, const float* xyz, uint64_t n, uint32_t* out4) { return nearestStats(mesh, xyz, n, out4, false); }
int sdfhip_mesh_nearest_stats_preseeded(sdfhip_mesh* mesh, const float* xyz, uint64_t n, uint32_t* out4) { return nearestStats(mesh, xyz, n, out4, true); }

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

// (sdfhip_ctx_create: the runtime loads a translation unit's code object on the first use of one of its kernels — milliseconds that would
// otherwise land in the first build or the first query of a process)
namespace sdfhip { void loadKernelsBvh() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_top_init)); (void)hipGetLastError(); } }
