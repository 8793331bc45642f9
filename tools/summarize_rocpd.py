"""Turn the rocprofv3 (rocpd sqlite) outputs of tools/profile_bench.sh into small text summaries for profiles/.

Usage: python tools/summarize_rocpd.py gpurun_out/prof_r01 profiles/r01
Writes <prefix>_kernel_stats.csv (rocprofv3 --kernel-trace --stats equivalent) and <prefix>_pmc.csv
(per-kernel averages of FETCH_SIZE / WRITE_SIZE / TCC hit+miss from the separate --pmc passes)."""
import csv
import os
import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "")
    if "rocprim" in name:
        m = re.search(r"detail::(\w+?)(?:_config|_impl|<)", name)
        return "rocprim::" + (m.group(1) if m else "kernel")
    return name[:90]


SPLIT_BY_GRID = ("k_octree_query<", "k_octree_query_coop<", "k_gather_blocks")     # the same kernel is launched on different workloads: keep them apart


def keyed(name, grid):
    k = short(name)
    return f"{k}@{int(grid)}" if grid is not None and any(t in k for t in SPLIT_BY_GRID) else k


def main(src, prefix):
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    db = sqlite3.connect(os.path.join(src, "trace", "bench_results.db"))
    rows = db.execute("select name, grid_x, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name, grid_x order by sum(duration) desc").fetchall()
    tot = sum(r[3] for r in rows)
    with open(prefix + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "percent"])
        agg = {}
        for name, grid, n, s, a, mn, mx in rows:
            k = keyed(name, grid)
            e = agg.setdefault(k, [0, 0, 1e30, 0])
            e[0] += n; e[1] += s; e[2] = min(e[2], mn); e[3] = max(e[3], mx)
        for k, (n, s, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, n, int(s), int(s / n), int(mn), int(mx), f"{100 * s / tot:.3f}"])
    out = {}
    SQ = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE"]
    EA = ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"]
    for sub, counters in (("pmc_fetch", ["FETCH_SIZE"]), ("pmc_write", ["WRITE_SIZE"]), ("pmc_l2", ["TCC_HIT_sum", "TCC_MISS_sum"]), ("pmc_sq", SQ), ("pmc_ea", EA), ("pmc_ea2", EA)):
        p = os.path.join(src, sub, "bench_results.db")
        if not os.path.exists(p):
            continue
        d = sqlite3.connect(p)
        for name, grid, cname, n, avg in d.execute("select kernel_name, grid_size_x, counter_name, count(*), avg(value) from counters_collection group by kernel_name, grid_size_x, counter_name"):
            if cname in counters:
                k = keyed(name, grid)
                if k in out and cname in out[k]:          # same short name from several template instances: weighted mean
                    n0, a0 = out[k][cname]; out[k][cname] = (n0 + n, (a0 * n0 + avg * n) / (n0 + n))
                else:
                    out.setdefault(k, {})[cname] = (n, avg)
    with open(prefix + "_pmc.csv", "w", newline="") as f:
        w = csv.writer(f)
        cols = ["FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"]
        w.writerow(["kernel", "dispatches"] + [c + "_avg_per_dispatch" for c in cols] + ["note"])
        for k, v in sorted(out.items()):
            if not k.startswith("sdfhip"):
                continue
            n = max(x[0] for x in v.values())
            w.writerow([k, n] + [f"{v[c][1]:.3f}" if c in v else "" for c in cols] + ["FETCH/WRITE_SIZE in KB as reported; kernel@N = launches over N work-items"])
    if any(c in v for v in out.values() for c in EA):
        with open(prefix + "_pmc_ea.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "dispatches"] + [c + "_avg_per_dispatch" for c in EA])
            for k, v in sorted(out.items()):
                if k.startswith("sdfhip") and any(c in v for c in EA):
                    w.writerow([k, max(v[c][0] for c in EA if c in v)] + [f"{v[c][1]:.6g}" if c in v else "" for c in EA])
    with open(prefix + "_pmc_sq.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches"] + [c + "_avg_per_dispatch" for c in SQ])
        for k, v in sorted(out.items()):
            if k.startswith("sdfhip") and any(c in v for c in SQ):
                n = max(v[c][0] for c in SQ if c in v)
                w.writerow([k, n] + [f"{v[c][1]:.6g}" if c in v else "" for c in SQ])
    # Where the runtime's own copy / fill kernels come from (hipMemcpyAsync device-to-device and pageable host copies run as
    # __amd_rocclr_copyBuffer, hipMemsetAsync as fillBufferAligned): by duration class and by the library kernel launched just before them.
    with open(prefix + "_copies.txt", "w") as f:
        seq = db.execute("select name, start, end, grid_x from kernels order by start").fetchall()
        for what in ("__amd_rocclr_copyBuffer", "__amd_rocclr_fillBufferAligned"):
            classes = {"< 10 us": [0, 0], "10 - 100 us": [0, 0], "0.1 - 1 ms": [0, 0], ">= 1 ms": [0, 0]}
            before = {}
            last = "(start)"
            for name, st, en, grid in seq:
                if what in name:
                    d = en - st
                    c = "< 10 us" if d < 10_000 else ("10 - 100 us" if d < 100_000 else ("0.1 - 1 ms" if d < 1_000_000 else ">= 1 ms"))
                    classes[c][0] += 1; classes[c][1] += d
                    e = before.setdefault(last, [0, 0]); e[0] += 1; e[1] += d
                elif "rocclr" not in name:
                    last = short(name)
            n = sum(v[0] for v in classes.values()); t = sum(v[1] for v in classes.values())
            f.write(f"{what}: {n} launches, {t / 1e6:.2f} ms ({100 * t / tot:.1f} % of the trace's kernel time)\n")
            for c, (k, d) in classes.items():
                f.write(f"    {c:12s} {k:6d} launches {d / 1e6:9.3f} ms\n")
            f.write("  by the library kernel launched before them:\n")
            for k, (c, d) in sorted(before.items(), key=lambda kv: -kv[1][1])[:14]:
                f.write(f"    {d / 1e6:9.3f} ms in {c:5d}  after {k}\n")
    print(open(prefix + "_copies.txt").read())
    print(open(prefix + "_kernel_stats.csv").read()[:3000])
    print(open(prefix + "_pmc.csv").read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
