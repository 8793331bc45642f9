#!/usr/bin/env python
"""bench.py — headline benchmark of the hot path on MI355X.

Workload (BASELINE.json configs[1], synthetic stand-in because no Armadillo file exists here):
  bumpy icosphere s=7 (327 680 triangles), OctreeSdf depth 8, start depth 3, threshold 1e-3 (NO_CONTINUITY),
  built on the GPU; a "step" is ONE batched getDistance() pass over 10 M uniform-random points resident in HBM.
value = whole-job M queries/s (all ranks' points / max-over-ranks time).  With N > 1 every rank holds the tree
(built SHARDED by start-grid cell + RCCL all-gather) and its own 10 M points: weak scaling, no data-path collective.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import sdflib_amd as S  # noqa: E402
from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin  # noqa: E402
from sdflib_amd import distributed as sdist  # noqa: E402

# where sdfhip_mesh_build_bvh builds the tree: on the device (the default: introsort rounds over global memory + k_bvh_subtrees) or by the host planner
BVH_BUILT_ON = "host" if os.environ.get("SDFHIP_BVH_BUILD") == "host" else "device"

MESH_BBOX, MESH_VERTS = None, -1        # the loader's box of a --mesh file (and its vertex count: build_1m's own mesh has none)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured copy ceiling


def source_hashes():
    """sha256 of every source file the kernels are compiled from (the GPU box has no .git, so a commit id cannot be read there)."""
    import glob, hashlib
    files = sorted(glob.glob(os.path.join(ROOT, "sdflib_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "sdflib_amd", "csrc", "*.h"))) + [os.path.join(ROOT, "include", "sdfhip.h"),
                                                                                                                                               os.path.join(ROOT, "sdflib_amd", "csrc", "Makefile")]      # (the compiler flags)
    return {os.path.relpath(p, ROOT): hashlib.sha256(open(p, "rb").read()).hexdigest()[:16] for p in files}


# the files a kernel's code comes from: a counter profile is accepted for a kernel only if none of them changed since it was recorded
KERNEL_SOURCES = {          # (sdfhip_internal.h holds host-side plumbing — allocation, error handling — and no kernel code: not listed)
    "octree_query": ["sdflib_amd/csrc/octree_query.hip", "sdflib_amd/csrc/octree_lattice.hip", "sdflib_amd/csrc/octree_internal.h", "sdflib_amd/csrc/dev_tricubic.h", "sdflib_amd/csrc/dev_math.h",
                     "sdflib_amd/csrc/blocks.hip", "sdflib_amd/csrc/Makefile"],
    "exact_query": ["sdflib_amd/csrc/exact_query.hip", "sdflib_amd/csrc/exact_internal.h", "sdflib_amd/csrc/dev_math.h", "sdflib_amd/csrc/Makefile"],
    "fit_mfma": ["sdflib_amd/csrc/dev_fit_mfma.h", "sdflib_amd/csrc/dev_tricubic.h", "sdflib_amd/csrc/dev_math.h", "sdflib_amd/csrc/octree_build.hip", "sdflib_amd/csrc/Makefile"],
}


class Profile:
    """The newest committed rocprofv3 summary of this same command (profiles/rNN_bench_{kernel_stats,pmc,pmc_sq}.csv + rNN_bench_meta.json,
    written by tools/profile_bench.sh + tools/summarize_rocpd.py: kernel trace, then separate FETCH_SIZE / WRITE_SIZE / TCC / SQ passes,
    per-dispatch averages, the query and calibration kernels split by grid size).  A bench run cannot collect counters itself; what ties
    the counters to the code being timed is the meta file's per-source-file hashes, compared with the files on disk now."""

    def __init__(self, enabled=True):
        import csv, glob
        self.prefix = None; self.pmc = {}; self.sq = {}; self.stats = {}; self.meta = {}; self.ea = {}
        if not enabled:
            return
        for path in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_meta.json")))):
            pre = path[:-len("_meta.json")]
            try:
                self.meta = json.load(open(path))
                rd = lambda suf: {row["kernel"].replace(" ", ""): row for row in csv.DictReader(open(pre + suf))} if os.path.exists(pre + suf) else {}
                self.pmc, self.sq, self.stats, self.ea = rd("_pmc.csv"), rd("_pmc_sq.csv"), rd("_kernel_stats.csv"), rd("_pmc_ea.csv")
                if self.pmc:
                    self.prefix = os.path.relpath(pre, ROOT); break
            except Exception:
                continue
        self.now = source_hashes()

    def stale(self, group):
        """None if the profile was recorded with today's sources of `group`, else the reason it is refused."""
        if not self.prefix:
            return "no committed profile with a meta file (profiles/rNN_bench_meta.json)"
        rec = self.meta.get("source_hashes", {})
        changed = [f for f in KERNEL_SOURCES[group] if rec.get(f) != self.now.get(f)]
        return f"{', '.join(changed)} changed since {self.prefix} was recorded" if changed else None

    @staticmethod
    def _find(table, kernel, n_threads, need):
        grid = (n_threads + 255) // 256 * 256 if n_threads else 0
        for key in ((f"{kernel}@{grid}", f"{kernel}@{grid // 256}", kernel) if n_threads else (kernel,)):
            r = table.get(key)
            if r and all(r.get(c) for c in need):
                return r
        return None

    def pmc_row(self, kernel, n_threads=0):
        return self._find(self.pmc, kernel, n_threads, ("FETCH_SIZE_avg_per_dispatch", "WRITE_SIZE_avg_per_dispatch"))

    def sq_row(self, kernel, n_threads=0):
        return self._find(self.sq, kernel, n_threads, ("SQ_ACTIVE_INST_VALU_avg_per_dispatch",))

    def stats_row(self, kernel, n_threads=0):
        return self._find(self.stats, kernel, n_threads, ("avg_ns",))

    def fetch_calibration(self):
        """MI355X_MICROARCH.md calibrates FETCH_SIZE (x2) for coalesced 16-B/lane streaming reads ONLY; the query kernels gather 256-B blocks, so
        the factor is measured on that pattern: the calibration kernel (sdfhip_test_gather_blocks: the same cooperative block loads, every 256-B
        block of a 2.56 GB array exactly once -> known bytes) runs in the same profiled command; factor = known read bytes / reported FETCH_SIZE."""
        c = self.pmc_row("sdfhip::k_gather_blocks_coop", CALIB_BLOCKS) or self.pmc_row("sdfhip::k_gather_blocks", CALIB_BLOCKS)
        if c is None:
            return 1.0, "no calibration row: FETCH_SIZE as reported"
        known = CALIB_BLOCKS * (256 + 4)            # every block once + its 4-byte id
        rep = float(c["FETCH_SIZE_avg_per_dispatch"]) * 1024
        return known / rep, f"k_gather_blocks (same cooperative 256-B block loads as the query kernel): {known} B known / {int(rep)} B reported"

    def traffic(self, group, kernel, n_threads=0):
        """Measured bytes per launch that crossed the L2 -> fabric boundary (FETCH_SIZE x calibration + WRITE_SIZE; both are reported in KB), with
        provenance, or {"traffic": None, "traffic_refused": why}."""
        why = self.stale(group)
        if why:
            return {"traffic": None, "traffic_refused": why}
        r = self.pmc_row(kernel, n_threads)
        if r is None:
            return {"traffic": None, "traffic_refused": f"no row for {kernel} in {self.prefix}_pmc.csv"}
        fetch, write = float(r["FETCH_SIZE_avg_per_dispatch"]) * 1024, float(r["WRITE_SIZE_avg_per_dispatch"]) * 1024
        hit, miss = float(r.get("TCC_HIT_sum_avg_per_dispatch") or 0), float(r.get("TCC_MISS_sum_avg_per_dispatch") or 0)
        factor, note = self.fetch_calibration()
        ea = self._find(self.ea, kernel, n_threads, ("TCC_EA0_RDREQ_128B_sum_avg_per_dispatch", "TCC_EA0_RDREQ_sum_avg_per_dispatch"))
        if ea:      # the L2's read requests by size (round 6): exact bytes, no calibration factor (on the calibration kernel: 2.0313e7 x 128 B = 2.600 GB for 2.6 GB known)
            n128, n64, n32 = (float(ea.get(f"TCC_EA0_RDREQ_{b}_sum_avg_per_dispatch") or 0) for b in ("128B", "64B", "32B"))
            rd_bytes = 128 * n128 + 64 * n64 + 32 * n32
            return self._traffic_tail({"traffic": int(rd_bytes + write), "traffic_source": f"{self.prefix}_pmc_ea.csv (TCC_EA0_RDREQ by request size: 128 B x {n128:.6g} + 64 B x {n64:.4g} + 32 B x {n32:.4g}) + WRITE_SIZE, per launch; sources unchanged since",
                                       "fabric_read_bytes": int(rd_bytes), "write_bytes_reported": int(write), "fetch_size_x_calibration": int(fetch * factor),
                                       "l2_hit": round(hit / (hit + miss), 4) if hit + miss > 0 else None}, kernel, n_threads)
        out = {"traffic": int(fetch * factor + write), "traffic_source": f"{self.prefix}_pmc.csv (FETCH_SIZE x fetch_calibration + WRITE_SIZE, per launch; sources unchanged since)",
               "fabric_bytes_reported": int(fetch), "write_bytes_reported": int(write), "fetch_calibration": round(factor, 3), "fetch_calibration_from": note,
               "l2_hit": round(hit / (hit + miss), 4) if hit + miss > 0 else None}
        return self._traffic_tail(out, kernel, n_threads)

    def _traffic_tail(self, out, kernel, n_threads):
        sq = self.sq_row(kernel, n_threads)
        if sq and sq.get("SQ_THREAD_CYCLES_VALU_avg_per_dispatch"):
            out["valu_lanes_active"] = round(float(sq["SQ_THREAD_CYCLES_VALU_avg_per_dispatch"]) / (64.0 * float(sq["SQ_ACTIVE_INST_VALU_avg_per_dispatch"])), 3)
        out.update(self.valu_issue(kernel, n_threads))
        return out

    def valu_ceiling(self):
        """SQ_ACTIVE_INST_VALU per second of the calibration kernel (k_valu_peak: nothing but independent unpacked v_fma_f32, 32 waves per CU)
        in the SAME profile: the counter's own ceiling, whatever the shader clock did and whatever a count stands for.  None without the row."""
        sq, st = self.sq_row("sdfhip::k_valu_peak"), self.stats_row("sdfhip::k_valu_peak")
        if not (sq and st and float(st["avg_ns"]) > 0):
            return None
        return float(sq["SQ_ACTIVE_INST_VALU_avg_per_dispatch"]) / (float(st["avg_ns"]) * 1e-9)

    def valu_issue(self, kernel, n_threads=0):
        """Share of the chip's VALU issue capacity the kernel used = its SQ_ACTIVE_INST_VALU per second / the same counter per second of the
        calibration kernel in the same profile (valu_ceiling) - self-calibrating: no clock, SIMD count or cycles-per-instruction constant enters.
        Profiles recorded before the calibration kernel existed fall back to 4 x counter / (time x 2.4 GHz x 1024 SIMDs), labelled as such
        (that constant-based form read 1.03 for k_exact_lists: the sustained clock is not the 2.4 GHz sheet figure)."""
        sq, st = self.sq_row(kernel, n_threads), self.stats_row(kernel, n_threads)
        if not (sq and st and float(st["avg_ns"]) > 0):
            return {}
        rate = float(sq["SQ_ACTIVE_INST_VALU_avg_per_dispatch"]) / (float(st["avg_ns"]) * 1e-9)
        ceil = self.valu_ceiling()
        if ceil:
            return {"valu_issue_frac": round(rate / ceil, 3),
                    "valu_issue_basis": f"SQ_ACTIVE_INST_VALU per second of the kernel ({rate:.4g}) / of k_valu_peak in the same profile ({ceil:.4g}: unpacked v_fma_f32 back to back on every SIMD)"}
        busy = 4.0 * rate / (GPU_CLOCK_HZ * GPU_SIMDS)
        return {"valu_issue_frac": round(busy, 3), "valu_issue_basis": f"UNCALIBRATED (no k_valu_peak row in {self.prefix}): 4 x SQ_ACTIVE_INST_VALU / (profiled dispatch {float(st['avg_ns']) / 1e6:.3f} ms x {GPU_CLOCK_HZ / 1e9:.1f} GHz x {GPU_SIMDS} SIMDs)"}


def roofline_block(kernel, kernel_ms, algorithmic_bytes, compulsory_bytes, tr):
    """One rule for every HBM-bound kernel, so that no fraction can exceed what physically moved:
         achieved = min(ALGORITHMIC bytes, MEASURED traffic) / kernel time
       i.e. a byte counts only if the algorithm needs it AND it actually crossed the L2 -> fabric boundary (counters of the
       committed profile of this same command, accepted only if the kernel's sources are unchanged).  Lanes that share a cache line do not get
       credit for the bytes they did not move; wasted re-reads do not count either.  Without an accepted profile the compulsory bytes
       (every input / output / touched structure byte exactly once) stand in for the traffic: a lower bound, labelled as such."""
    t = kernel_ms * 1e-3
    moved, basis = (tr["traffic"], "measured traffic") if tr.get("traffic") else (compulsory_bytes, "compulsory bytes (no accepted counter profile)")
    counted = min(algorithmic_bytes, moved)
    r = {"bound": "hbm", "kernel": kernel.split("::")[-1], "achieved": round(counted / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(counted / t / 1e9 / HBM_PEAK_GBS, 4), "frac_basis": f"min(algorithmic, {basis}) / kernel time / peak",
         "traffic": tr.get("traffic"), "kernel_ms": round(kernel_ms, 4),
         "algorithmic_bytes_per_launch": int(algorithmic_bytes), "algorithmic_gb_s": round(algorithmic_bytes / t / 1e9, 1),
         "algorithmic_over_traffic": (round(algorithmic_bytes / tr["traffic"], 3) if tr.get("traffic") else None),
         "compulsory_bytes_per_launch": int(compulsory_bytes), "compulsory_frac": round(compulsory_bytes / t / 1e9 / HBM_PEAK_GBS, 4),
         "traffic_over_compulsory": (round(tr["traffic"] / compulsory_bytes, 2) if tr.get("traffic") and compulsory_bytes else None)}
    r.update({k: v for k, v in tr.items() if k != "traffic"})
    return r


CALIB_BLOCKS = 10_000_000      # 256-B blocks of the calibration array (2.56 GB: ten times the Infinity Cache)


def octree_query_roofline(info, start_depth, n, kernel_ms, gradient, prof, kernel, ic_ceiling=None):
    """SURVEY.md 8(d) "Q": algorithmic bytes per query = point 12 + distance 4 (+ gradient 12) + 256 coefficients + 4 x mean dependent node loads
    (mean over uniform-random points, from the leaf-per-depth histogram).  Compulsory bytes per launch = the streams + every coefficient block and
    node word that can be touched, once.  See roofline_block for what `achieved` / `frac` count."""
    lpd = np.array(list(info.leaves_per_depth), dtype=np.float64)
    prob = np.array([lpd[d] / 8.0 ** d for d in range(16)])
    mean_loads = float(sum(prob[d] * (d - start_depth + 1) for d in range(16)) / max(prob.sum(), 1e-30))
    bytes_per_query = 12 + 4 + (12 if gradient else 0) + 256 + 4 * mean_loads
    tree_bytes = 4 * int(info.num_words)
    io_bytes = n * (16 + (12 if gradient else 0))
    working_set = tree_bytes + io_bytes
    fits = working_set < 256 * 2 ** 20
    compulsory = io_bytes + min(tree_bytes, int((256 + 4 * mean_loads) * n))
    r = roofline_block(kernel, kernel_ms, bytes_per_query * n, compulsory, prof.traffic("octree_query", kernel, n))
    r.update({"bytes_per_query": round(bytes_per_query, 2), "mean_node_loads": round(mean_loads, 3), "working_set_bytes": int(working_set), "infinity_cache_resident": bool(fits)})
    if not fits:
        r["bound_regime"] = "hbm"
        r["regime"] = "HBM gather (working set exceeds the 256 MB Infinity Cache)"
        return r
    # Cache-resident regime: the working set (tree + streams) fits the 256 MB Infinity Cache, so the bytes that leave the L2s never reach the HBM
    # pins and SURVEY 8(d)'s formula (algorithmic bytes / time / 8 TB/s) is not a fraction of anything: it is printed, labelled, under
    # "hbm_formula".  The ceiling those bytes DO meet is the fabric + Infinity Cache serving 256-byte gathers, measured in this same run with the
    # query kernel's own load pattern on a 200 MB array (ic_ceiling: sdfhip_test_gather_blocks, random blocks, same byte accounting per lane).
    t = kernel_ms * 1e-3
    r["hbm_formula"] = {"achieved": r["algorithmic_gb_s"], "peak": HBM_PEAK_GBS, "frac": round(r["algorithmic_gb_s"] / HBM_PEAK_GBS, 4),
                        "label": "cache-resident: not an HBM fraction (algorithmic bytes / kernel time / 8 TB/s; the bytes are served by the L2s and the Infinity Cache)"}
    r["bound"] = "infinity_cache"
    r["bound_regime"] = "fabric / Infinity Cache (working set below 256 MB: see roofline_hbm for the HBM-resident figure of this kernel)"
    if ic_ceiling:
        moved = r["traffic"] if r.get("traffic") else compulsory
        r["achieved"] = round(min(bytes_per_query * n, moved) / t / 1e9, 1)
        r["peak"] = ic_ceiling["gb_s"]
        r["peak_name"] = "measured Infinity-Cache 256-B gather ceiling: k_gather_blocks_coop over a 200 MB array in this run (%d random blocks, %d B per lane) = %.0f GB/s" % (
            ic_ceiling["lanes"], ic_ceiling["bytes_per_lane"], ic_ceiling["gb_s"])
        r["frac"] = round(r["achieved"] / r["peak"], 4)
        r["frac_basis"] = ("min(algorithmic bytes, measured L2 -> fabric traffic) / kernel time / measured Infinity-Cache gather ceiling" if r.get("traffic")
                           else "compulsory bytes / kernel time / measured Infinity-Cache gather ceiling (no accepted counter profile)")
    r["regime"] = ("L2-miss gather served by the 256 MB Infinity Cache: numerator = bytes that crossed the L2 -> fabric boundary (counters), denominator = the rate "
                   "at which the same boundary serves the calibration gather in this run; the HBM-resident figure of the same kernel is roofline_hbm")
    return r


def _self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run with one rank per GPU (RCCL), the command the
    driver itself uses.  The re-executed ranks see WORLD_SIZE and take the normal path; rank 0 prints the one JSON line."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--subdiv", type=int, default=7, help="icosphere subdivisions (7 -> 327 680 triangles)")
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--start-depth", type=int, default=3)
    ap.add_argument("--queries", type=int, default=10_000_000)
    ap.add_argument("--eval", choices=["exact", "fast"], default="exact")
    ap.add_argument("--gradient", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000, help="queries timed on the host cores for cpu_baseline")
    ap.add_argument("--mesh", default=None, help="a PLY / OBJ file to run instead of the synthetic stand-in (sdflib_amd.meshio: first mesh, triangulated; the real Bunny / "
                    "Armadillo when supplied); the box is the exporter's: bbox + 20 %% of the largest extent, cube-ified by the build")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-build-1m", action="store_true", help="skip the depth-8 build of the 1.31 M-triangle mesh (BASELINE configs[3])")
    ap.add_argument("--no-forecast", action="store_true", help="skip build_1m.forecast (28 shard builds of the 1.31 M mesh): the profile passes use it so that per-kernel averages describe the benchmark's own builds")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (gradient, fast eval, 256^3 grid, ExactOctreeSdf)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _self_launch(args.gpus)          # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook: several ranks on ONE device with gloo collectives (the real launch is one rank per GPU over RCCL)
    one_device = os.environ.get("SDFHIP_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} (or without a launcher)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    # ---- setup (untimed): mesh, context on torch's stream, tree -------------------------------------------
    if args.mesh:
        from sdflib_amd import meshio
        v, f = meshio.read_mesh(args.mesh)
        mesh_name = f"{os.path.basename(args.mesh)} ({len(f)} tris, file)"
    else:
        v, f = bumpy_icosphere(args.subdiv)
        mesh_name = f"bumpy icosphere s={args.subdiv} ({len(f)} tris, Armadillo-scale stand-in)"
    box = box_with_margin(v)
    ctx = S.Context(dev.index, use_torch_stream=True)
    # a mesh that comes from a FILE carries its bounding box like the reference's Mesh(filePath) (src/utils/Mesh.cpp:9-88), which is what
    # enables calculateMeshTriangleData's seam welding (TriangleUtils.cpp:292-420); the synthetic stand-ins are raw arrays (no box, no welding)
    global MESH_BBOX, MESH_VERTS
    MESH_BBOX = np.concatenate([v.min(axis=0), v.max(axis=0)]).astype(np.float32) if args.mesh else None
    MESH_VERTS = len(v)
    mesh = S.Mesh(v, f, ctx, bbox=MESH_BBOX)
    bvh_s = sdist.share_bvh(mesh, rank, world, dev) if world > 1 else mesh.build_bvh()      # N > 1: every rank builds it on its own device (host planner: rank 0 + broadcast)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if world > 1:
        tree, binfo = sdist.build_octree_sharded(mesh, box, args.depth, args.start_depth, 1e-3, rank, world, dev)
    else:
        tree = S.OctreeSdf(mesh, box, args.depth, args.start_depth, 1e-3, num_threads=2)
        binfo = {"shard_build_s": tree.info.seconds_total, "exchange_s": 0.0}
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    info = tree.info
    rebuild_s = None
    if world == 1:              # the same build again: steady state (the first one in a context also grows its scratch buffers)
        t0 = time.perf_counter()
        again = S.OctreeSdf(mesh, box, args.depth, args.start_depth, 1e-3, num_threads=2)
        torch.cuda.synchronize(); rebuild_s = time.perf_counter() - t0
        rebuild_info = again.info
        again.close()
        e2e = end_to_end_build(ctx, v, f, box, args.depth, args.start_depth, dev)

    gen = torch.Generator(device=dev); gen.manual_seed(1234 + rank)
    bb = tree.get_grid_bounding_box()
    lo = torch.tensor(bb[:3], device=dev)
    size = float(bb[3] - bb[0])
    pts = (lo + torch.rand((args.queries, 3), generator=gen, device=dev) * (size * 0.999999)).contiguous()
    out = torch.empty(args.queries, dtype=torch.float32, device=dev)
    outg = torch.empty((args.queries, 3), dtype=torch.float32, device=dev) if args.gradient else None
    mode = S.EVAL_EXACT if args.eval == "exact" else S.EVAL_FAST

    def step():
        tree.get_distance(pts, gradient=args.gradient, eval_mode=mode, out=out, out_grad=outg)

    for _ in range(args.warmup):
        step()
    # ---- timed region: exactly K steps, barrier + synchronize on both sides ------------------------------
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in evs:
        a.record(); step(); b.record()     # events on the stream the kernel is launched on (ctx = torch's stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=torch.device("cpu") if one_device else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))

    prof = Profile(enabled=(not args.mesh and args.subdiv == 7 and args.depth == 8 and args.start_depth == 3))
    valu_cal = valu_calibration(ctx, dev) if world == 1 else None
    build_roof = build_roofline(rebuild_info, rebuild_s, len(v), len(f), prof, valu_cal) if world == 1 else None
    kname = f"sdfhip::k_octree_query_coop<{0 if args.eval == 'exact' else 1},{'true' if args.gradient else 'false'}>"
    ic_ceiling = ic_gather_ceiling(ctx, dev) if rank == 0 else None           # outside the timed region
    roof = octree_query_roofline(info, args.start_depth, args.queries, kernel_ms, args.gradient, prof, kname, ic_ceiling)
    if ic_ceiling:
        roof["ic_gather_ceiling"] = ic_ceiling
    if build_roof and valu_cal:
        build_roof["valu_calibration"] = valu_cal

    total_queries = args.queries * world * args.steps
    value = total_queries / elapsed / 1e6

    copy_gbs = measured_copy_gbs(dev)
    roof["copy_bw_measured_gbs"] = round(copy_gbs, 1); roof["frac_of_measured_copy"] = round(min(roof["achieved"] / copy_gbs, 1.0), 4)
    result = {
        "metric": "Mqueries/sec getDistance() (OctreeSdf, whole job)", "value": round(value, 2), "unit": "Mqueries/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": ("file" if args.mesh else "synthetic"),
        "config": {"workload": f"{mesh_name} OctreeSdf depth {args.depth} "
                               f"start {args.start_depth} thr 1e-3 NO_CONTINUITY; {args.queries} uniform-random getDistance per GPU per step",
                   "eval": args.eval, "gradient": bool(args.gradient), "queries_per_gpu": args.queries,
                   **({"mesh_edges": mesh.edge_stats(), "mesh_box": "the file's bounding box (loader semantics: seam welding on)"} if args.mesh else {}),
                   "octree_words": int(info.num_words), "octree_leaves": int(info.num_leaves), "parallelism": f"replicated tree x{world}, sharded build"},
        "per_gpu_mqueries_s": round(value / world, 2),
        "roofline": roof,
        "build": {**(e2e if world == 1 else {}), **({"roofline": build_roof} if build_roof else {}), "octree_build_s": round(build_s, 4), "octree_rebuild_s": (round(rebuild_s, 4) if rebuild_s is not None else None), "bvh_build_s": round(bvh_s, 4), "bvh_built_on": BVH_BUILT_ON, "samples": int(info.num_samples), "bvh_traversals": int(info.num_traversals), "nearest_fallbacks": int(info.num_nearest_fallbacks), **_r4(binfo)},
    }

    if world > 1:       # collective sanity: every rank contributes its rank + 1; the sum proves all N ranks were in the communicator
        chk = torch.tensor([rank + 1], dtype=torch.int64, device=torch.device("cpu") if one_device else dev)
        dist.all_reduce(chk)
        result["collectives"] = {"backend": dist.get_backend(), "ranks_seen": int(binfo.get("ranks_seen", 0)), "rank_sum_ok": bool(int(chk.item()) == world * (world + 1) // 2),
                                 "octree_bytes_all_gathered_per_rank": int(binfo.get("exchange_bytes", 0))}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # the CPU baseline is reported at N = 1 only
        result["cpu_baseline"] = cpu_baseline(v, f, box, args, pts)
    if not args.no_extras:
        result["extras"] = extras(tree, mesh, box, pts, out, dev, rank, world, prof)
    if not args.no_build_1m:
        result["build_1m"] = build_1m(ctx, rank, world, dev, forecast=not args.no_forecast)
    # the HBM-honest figure of the headline kernel beside the headline (whose working set sits in the Infinity Cache): the depth-9 tree's block
    deep = (result.get("extras") or {}).get("deep_tree_d9")
    if deep and deep.get("roofline"):
        result["roofline_hbm"] = {**deep["roofline"], "workload": f"same kernel, depth-9 tree ({deep['words'] * 4 / 1e9:.2f} GB, beyond the 256 MB Infinity Cache), {deep['queries']} queries, one per distinct depth-9 cell near the surface (every coefficient block fetched once): extras.deep_tree_d9; the volume-uniform 12 M figure on the same tree is extras.deep_tree_d9.uniform_random"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


NEAR_KERNEL = "sdfhip::k_near_quads<256,false,1536u>"
KERNEL_SOURCES["near_search"] = ["sdflib_amd/csrc/dev_bvh_fast.h", "sdflib_amd/csrc/dev_bvh.h", "sdflib_amd/csrc/dev_math.h", "sdflib_amd/csrc/octree_sampler.h", "sdflib_amd/csrc/octree_build.hip", "sdflib_amd/csrc/Makefile"]
FP32_VECTOR_PEAK_TFLOPS = 157.3
GPU_CLOCK_HZ, GPU_SIMDS = 2.4e9, 1024          # MI355X: 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)


def build_roofline(info, build_s, nv, nt, prof, valu_cal=None):
    """SURVEY.md 8(d) "B" for the build's dominant kernel, the fp32 candidate search of the nearest-triangle queries (k_near_quads, ~2/3 of a
    NO_CONTINUITY build's GPU time).  Live: the kernel's device time inside the build just run (HIP events on the build's stream, summed over
    its batches) and its work counters (wide-node expansions, triangle tests: counted by the kernel).  From the committed counter profile,
    when its sources are unchanged: lanes active per VALU instruction, L2 -> fabric traffic per dispatch (an average over ALL dispatches of the
    profiled bench run, whose meshes differ: indicative).  Algorithmic bytes per query = 12 (point) + 5 (candidate count, bound) + 96 per
    expansion (header + four child records + references of a 128-byte node) + 48 per triangle test; they are gathered through the L2 (hit rate
    above 90 %), so the HBM fraction says how far the kernel is from being byte bound: it is bound by instruction issue (valu_lanes_active,
    expansions_per_query).  build_compulsory = 8(d)'s per-node bytes of the whole build over the build's wall time."""
    q = int(info.num_traversals); t = float(info.seconds_near_candidates)
    if q == 0 or t <= 0:
        return None
    ex, tr = int(info.near_expansions), int(info.near_triangle_tests)
    alg = q * 17 + ex * 96 + tr * 48
    nodes, leaves = int(info.num_nodes), int(info.num_leaves)
    compulsory_build = 12 * nv + 12 * nt + 148 * nt + nodes * (8 * 36 + 19 * 36 + 20) + leaves * 256
    # The kernel is bound by VALU instruction issue, not by bytes (its records are gathered through the L2): `achieved` = useful lane-operations per
    # second = lanes active per VALU instruction x VALU instructions issued per second (both from the committed counter profile of this command,
    # when its sources are unchanged), `peak` = what 1024 SIMDs of 16 lanes issue at 2.4 GHz (39.3 T lane-operations/s: separate multiplies and
    # adds, no packed FMA), `frac` = lanes x issue.  Without an accepted profile achieved / frac are null; the L2-served byte figure stays as a note.
    # (rounds 1-5 priced this against 1024 SIMDs x 16 lanes x 2.4 GHz = 39.3 T lane-ops/s, which a CDNA4 SIMD exceeds: it retires a wave64 fp32
    # instruction in 2 cycles, not 4 - hence the "issue fractions" above 1 of round 5.  The ceiling is now MEASURED: k_valu_peak, live in this run.)
    valu_peak = valu_cal["t_lane_ops_s"] if valu_cal else GPU_CLOCK_HZ * GPU_SIMDS * 32 / 1e12
    r = {"bound": "valu", "kernel": NEAR_KERNEL.split("::")[-1], "kernel_ms_per_build": round(t * 1e3, 3), "search_ms_per_build": round(float(info.seconds_near_search) * 1e3, 3),
         "share_of_build": round(t / build_s, 3), "queries": q, "mqueries_s": round(q / t / 1e6, 1),
         "expansions_per_query": round(ex / q, 1), "triangle_tests_per_query": round(tr / q, 1),
         "achieved": None, "peak": round(valu_peak, 1), "peak_name": ("measured in this run: k_valu_peak, unpacked v_fma_f32 back to back on every SIMD" if valu_cal else "1024 SIMDs x 32 lanes x 2.4 GHz"), "unit": "T lane-ops/s", "frac": None,
         "l2_served_bytes_per_build": int(alg), "l2_served_gb_s": round(alg / t / 1e9, 1), "l2_served_over_hbm_peak": round(alg / t / 1e9 / HBM_PEAK_GBS, 4),
         "child_tests_per_s_g": round(4 * ex / t / 1e9, 1), "fp32_vector_frac": round((4 * ex * 55 + tr * 130) / t / 1e12 / FP32_VECTOR_PEAK_TFLOPS * 2, 4),
         "build_compulsory_bytes": int(compulsory_build), "build_compulsory_gb_s": round(compulsory_build / build_s / 1e9, 1), "build_compulsory_frac": round(compulsory_build / build_s / 1e9 / HBM_PEAK_GBS, 4),
         "note": "instruction bound (about 55 VALU instructions per child test, 130 per triangle test; fp32_vector_frac counts them as lane-instructions against the 2-flop-per-lane FMA peak), "
                 "not byte bound: the records are gathered through the L2"}
    why = prof.stale("near_search")
    if why:
        r["profile_refused"] = why
    else:
        sq, pm, st = prof.sq_row(NEAR_KERNEL), prof.pmc_row(NEAR_KERNEL), prof.stats_row(NEAR_KERNEL)
        if sq and sq.get("SQ_THREAD_CYCLES_VALU_avg_per_dispatch"):
            r["valu_lanes_active"] = round(float(sq["SQ_THREAD_CYCLES_VALU_avg_per_dispatch"]) / (64.0 * float(sq["SQ_ACTIVE_INST_VALU_avg_per_dispatch"])), 3)
        if pm:
            hit, miss = float(pm.get("TCC_HIT_sum_avg_per_dispatch") or 0), float(pm.get("TCC_MISS_sum_avg_per_dispatch") or 0)
            r["traffic_per_dispatch_avg"] = int((float(pm["FETCH_SIZE_avg_per_dispatch"]) * 2.0 + float(pm["WRITE_SIZE_avg_per_dispatch"])) * 1024)
            r["l2_hit"] = round(hit / (hit + miss), 4) if hit + miss > 0 else None
        if st:
            r["profile_avg_dispatch_ms"] = round(float(st["avg_ns"]) / 1e6, 3)
        r.update(prof.valu_issue(NEAR_KERNEL))
        r["profile"] = prof.prefix
        if r.get("valu_lanes_active") and r.get("valu_issue_frac"):
            r["frac"] = round(r["valu_lanes_active"] * r["valu_issue_frac"], 3)
            r["achieved"] = round(r["frac"] * valu_peak, 2)
            r["frac_basis"] = "valu_lanes_active x valu_issue_frac (useful lane issue of the chip)"
    return r


def end_to_end_build(ctx, v, f, box, depth, start_depth, dev):
    """What a caller of the class constructor waits for: host arrays in -> tree ready (mesh upload + TriangleData, BVH plan + install, octree
    build), then the first query's one-off cost (none since round 5: the builders emit the query layout; a tree that arrives as an array makes it then).  Steady state of this context: its
    scratch buffers exist already, nothing else is reused."""
    torch.cuda.synchronize(); t0 = time.perf_counter()
    box_arg = MESH_BBOX if (MESH_BBOX is not None and len(v) == MESH_VERTS) else None
    m = S.Mesh(v, f, ctx, plan_bvh_early=True, bbox=box_arg); torch.cuda.synchronize(); t1 = time.perf_counter()       # as the OctreeSdf constructors do: the BVH plan starts under the mesh preparation
    planner_s = m.build_bvh(); torch.cuda.synchronize(); t2 = time.perf_counter()
    t = S.OctreeSdf(m, box, depth, start_depth, 1e-3, num_threads=2); torch.cuda.synchronize(); tb = time.perf_counter()
    bb = t.get_grid_bounding_box()
    q = (torch.tensor(bb[:3], device=dev) + torch.rand((65536, 3), device=dev) * float(bb[3] - bb[0]) * 0.999).contiguous()
    o = torch.empty(65536, dtype=torch.float32, device=dev)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    t.get_distance(q, out=o); torch.cuda.synchronize(); t4 = time.perf_counter()
    t.get_distance(q, out=o); torch.cuda.synchronize(); t5 = time.perf_counter()
    words = t.get_octree_data(); t6 = time.perf_counter()       # what the C++ class's constructor does for getOctreeData(): the reference's array rebuilt from the layout + downloaded
    array_s, array_mb = t6 - t5, 4 * len(words) / 1e6
    del words
    t.close()
    return {"end_to_end_s": round(tb - t0, 4), "mesh_prep_s": round(t1 - t0, 4), "bvh_s_after_mesh": round(t2 - t1, 4),
            "octree_s": round(tb - t2, 4), "query_layout_s": round(max((t4 - t3) - (t5 - t4), 0.0), 5),
            "time_to_first_query_s": round((tb - t0) + (t4 - t3), 4),
            "array_download_s": round(array_s, 4), "array_download_note": f"the reference's node array ({array_mb:.0f} MB) rebuilt from the query layout and copied to a pageable host array (getOctreeData / saveToFile)",
            "time_to_first_query_note": "host arrays in -> the first batch of 65 536 distances back (mesh upload + TriangleData, BVH, octree build, first query incl. anything it makes once)"}


def _r4(d):
    return {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items()}


def measured_copy_gbs(dev):
    """The practical HBM ceiling next to the datasheet one: device-to-device copy of 1 GiB, read + write bytes
    over the best of 5 timings (HIP events)."""
    n = 256 << 20
    a = torch.empty(n, dtype=torch.float32, device=dev).fill_(1.0); b = torch.empty_like(a)
    best = 1e30
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b.copy_(a); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    del a, b
    return 2 * 4 * n / (best * 1e-3) / 1e9


def cpu_baseline(v, f, box, args, pts):
    """The oracle (CPU restatement of the reference, kind 'port') timed on this box's host cores: bounded sample."""
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")      # the oracle's idle OpenMP workers must not spin into the GPU measurements that follow
    from oracle import pyoracle as O
    cores = os.cpu_count() or 1
    om = O.Mesh(v, f, MESH_BBOX) if MESH_BBOX is not None else O.Mesh(v, f)
    t0 = time.perf_counter()
    # build at the bench depth is minutes of CPU on few cores: bound it by building one level shallower if needed
    cpu_depth = args.depth if cores >= 32 else min(args.depth, 7)
    ot = O.Octree(om, box, cpu_depth, args.start_depth, 1e-3, vertex_cache=False, layout=O.LAYOUT_SUBTREES)
    cpu_build_s = time.perf_counter() - t0
    sample = pts[:args.cpu_sample].cpu().numpy()
    # a container may grant fewer CPUs than the host has threads (cgroup v2 cpu.max): more OpenMP threads than about twice the quota are
    # only throttled, so the team sizes tried are the hardware threads, twice the quota and the quota; the best one is reported
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = float(q) / float(per) if q != "max" else None
    except Exception:
        pass
    teams = sorted({cores} | ({max(1, min(cores, int(2 * quota))), max(1, min(cores, int(quota)))} if quota else set()), reverse=True)
    ot.query(sample, grad=args.gradient, threads=cores)           # warm the thread team and the caches
    dt, used = 1e30, cores
    for team in teams:
        for _ in range(3):
            t0 = time.perf_counter()
            ot.query(sample, grad=args.gradient, threads=team)
            e = time.perf_counter() - t0
            if e < dt: dt, used = e, team
    host_threads, cores = cores, used
    t0 = time.perf_counter()
    ot.query(sample[:len(sample) // 8], grad=args.gradient, threads=1)
    dt1 = time.perf_counter() - t0
    return {"value": round(len(sample) / dt / 1e6, 3), "unit": "Mqueries/s", "cores": cores, "kind": "port",
            "sample": f"{len(sample)} of the same random points, oracle getDistance under OpenMP static schedule, {cores} threads; "
                      f"tree = oracle build depth {cpu_depth} ({cpu_build_s:.1f} s, OpenMP over start cells)",
            "single_thread_mqueries_s": round(len(sample) // 8 / dt1 / 1e6, 3), "cpu_build_s": round(cpu_build_s, 2), "cpu_build_depth": cpu_depth,
            "host_hardware_threads": host_threads, "cpu_quota_cpus": quota}


def reference_nearest(mesh, v, f, box, dev):
    """Row a5 with the REAL reference timed beside it: tmd::TriangleMeshDistance (oracle/_ref/libtmd_ref.so, the reference's own header
    compiled as it lies; test infrastructure) answers a bounded sample of nearest-triangle queries on the host cores, the engine
    (sdfhip_mesh_nearest: two-phase search on the planner's tree) the same points on the GPU; ids must agree."""
    try:
        from oracle import pyref
        if not pyref.available():
            return {"skipped": "oracle/_ref/libtmd_ref.so not present"}
    except Exception as e:      # noqa: BLE001
        return {"skipped": f"oracle/_ref unavailable: {e}"}
    from sdflib_amd.meshgen import random_points_in_box
    t0 = time.perf_counter(); ref = pyref.RefMesh(v, f); ref_build = time.perf_counter() - t0
    pts = random_points_in_box(box, 400_000, seed=4242)
    t0 = time.perf_counter(); ids_ref = ref.nearest(pts); t_ref = time.perf_counter() - t0
    big = random_points_in_box(box, 4_000_000, seed=4243)
    mesh.nearest_triangle(big[:1000])
    t0 = time.perf_counter(); mesh.nearest_triangle(big); t_gpu = time.perf_counter() - t0
    same = bool(np.array_equal(ids_ref, mesh.nearest_triangle(pts)))
    return {"kind": "reference", "reference_bvh_build_s": round(ref_build, 3), "reference_mqueries_s": round(len(pts) / t_ref / 1e6, 3), "reference_threads": os.cpu_count(),
            "reference_sample": f"{len(pts)} uniform points of the box, tmd::TriangleMeshDistance::signed_distance under OpenMP (dynamic schedule)",
            "engine_mqueries_s": round(len(big) / t_gpu / 1e6, 1), "engine_sample": f"{len(big)} points, host arrays in / out (PCIe inside)", "ids_identical": same}


def _time_ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def extras(tree, mesh, box, pts, out, dev, rank, world=1, prof=None):
    prof = prof or Profile(enabled=False)
    """Secondary per-GPU measurements (rank-local, untimed region): the other BASELINE.json configs on the same mesh."""
    n = pts.shape[0]
    outg = torch.empty((n, 3), dtype=torch.float32, device=dev)
    r = {}
    ms = _time_ms(lambda: tree.get_distance(pts, gradient=True, eval_mode=S.EVAL_EXACT, out=out, out_grad=outg))
    r["value_and_gradient_exact"] = {"ms": round(ms, 4), "mqueries_s": round(n / ms / 1e3, 1)}
    ms = _time_ms(lambda: tree.get_distance(pts, gradient=False, eval_mode=S.EVAL_FAST, out=out))
    r["value_fast_eval"] = {"ms": round(ms, 4), "mqueries_s": round(n / ms / 1e3, 1)}
    bb = tree.get_grid_bounding_box(); size = float(bb[3] - bb[0])
    step = np.full(3, size / 256, dtype=np.float32); origin = (bb[:3] + 0.5 * step).astype(np.float32)
    ms = _time_ms(lambda: tree.get_distance_grid(origin, step, (256, 256, 256), gradient=True, eval_mode=S.EVAL_EXACT, device_out=True), reps=20)
    words = int(tree.info.num_words)
    gbytes = 256 ** 3 * 16 + words * 4          # SURVEY 8(d) "G": 16 B written per point + the tree read once
    r["grid256_value_and_gradient"] = {"ms": round(ms, 4), "mqueries_s": round(256 ** 3 / ms / 1e3, 1), "algorithmic_gb_s": round(gbytes / ms / 1e6, 1),
                                       "hbm_frac": round(gbytes / ms / 1e6 / HBM_PEAK_GBS, 4), "note": "reference-order polynomial, leaf-driven (k_lattice_columns_exact: the z-independent prefix of every term once per column, ~500 flop/point instead of 1100): ALU bound, not HBM bound; bit-identical to the point kernel and the oracle"}
    ms = _time_ms(lambda: tree.get_distance_grid(origin, step, (256, 256, 256), gradient=True, eval_mode=S.EVAL_FAST, device_out=True), reps=40)
    r["grid256_value_and_gradient_fast_eval"] = {"ms": round(ms, 4), "mqueries_s": round(256 ** 3 / ms / 1e3, 1), "algorithmic_gb_s": round(gbytes / ms / 1e6, 1),
                                                 "hbm_frac": round(gbytes / ms / 1e6 / HBM_PEAK_GBS, 4),
                                                 "note": "leaf-driven (k_lattice_columns): leaves emit their lattice points, no per-point walk or division; same bits as the point kernel's EVAL_FAST"}
    # ExactOctreeSdf (BASELINE configs[2]): depth 7, start 3, min_triangles_per_node 128
    einfo = {}
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    if world > 1:       # start cells sharded over the ranks, one exchange (distributed.build_exact_sharded)
        ex, einfo = sdist.build_exact_sharded(mesh, box, 7, 3, 128, rank, world, dev)
        einfo = _r4(einfo)
    else:
        ex = S.ExactOctreeSdf(mesh, box, 7, 3, 128)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    i = ex.info
    q = pts
    if world == 1:      # the same build again: without the first build's one-time costs (scratch blocks of the context, code objects)
        reb = []
        for _ in range(2):      # (two, both reported: on some boxes a multi-GB hipMalloc takes 0.2-0.3 s; since r05z2 the caches keep the build's scratch below their mark, 1/32 of the device, and a rebuild allocates nothing)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ex2 = S.ExactOctreeSdf(mesh, box, 7, 3, 128)
            torch.cuda.synchronize(); reb.append(round(time.perf_counter() - t0, 4))
            ex2.close()
        einfo = dict(einfo, rebuild_s=min(reb), rebuilds_s=reb)
    ms = _time_ms(lambda: ex.get_distance(q, out=out), reps=3)
    r["exact_octree_d7_min128"] = {"build_s": round(dt, 4), "nodes": int(i.num_nodes), "cull_tests": int(i.cull_tests), "max_triangles_in_leafs": int(i.max_triangles_in_leafs),
                                  "queries": int(len(q)), "query_ms": round(ms, 3), "mqueries_s": round(len(q) / ms / 1e3, 1), **einfo,
                                  "roofline": exact_query_roofline(ex, q, ms, prof)}
    ex_for_scalar = ex
    # CONTINUITY builder (SdfExporter's default) on the same mesh / depth
    # (N > 1: every rank builds the whole tree, the BVH traversals of each sample batch are shared out, one all-reduce per batch)
    cinfo = {}
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    if world > 1:
        ct, cinfo = sdist.build_continuity_sharded(mesh, box, int(tree.info.max_depth), int(round(np.log2(tree.info.start_grid_size))), 1e-3, rank, world, dev)
        cinfo = {"exchange_s": round(cinfo["exchange_s"], 4), "exchange_bytes": int(cinfo["exchange_bytes"]), "ranks_sharing_traversals": world}
    else:
        ct = S.OctreeSdf(mesh, box, int(tree.info.max_depth), int(round(np.log2(tree.info.start_grid_size))), 1e-3, init_algorithm=S.ALG_CONTINUITY, num_threads=2)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ci = ct.info
    if world == 1:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ct2 = S.OctreeSdf(mesh, box, int(tree.info.max_depth), int(round(np.log2(tree.info.start_grid_size))), 1e-3, init_algorithm=S.ALG_CONTINUITY, num_threads=2)
        torch.cuda.synchronize(); cinfo = dict(cinfo, rebuild_s=round(time.perf_counter() - t0, 4))
        ct2.close()
    ms = _time_ms(lambda: ct.get_distance(pts, eval_mode=S.EVAL_EXACT, out=out))
    r["continuity_octree"] = {"build_s": round(dt, 4), "words": int(ci.num_words), "leaves": int(ci.num_leaves), "query_ms": round(ms, 4), "mqueries_s": round(n / ms / 1e3, 1), **cinfo}
    ct.close()
    # the 64x64 fit on the matrix cores (SDFHIP_FIT_MFMA): same topology, coefficients within the reference's own rounding noise
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mt = S.OctreeSdf(mesh, box, int(tree.info.max_depth), int(round(np.log2(tree.info.start_grid_size))), 1e-3, fit_mode=S.FIT_MFMA, num_threads=2)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    mi = mt.info
    r["fit_mfma_build"] = {"build_s": round(dt, 4), "words": int(mi.num_words), "same_size_as_exact_fit": bool(int(mi.num_words) == int(tree.info.num_words)),
                           "decisions_rechecked_with_exact_fit": int(mi.fit_rechecks), "nodes": int(mi.num_nodes), "roofline": mfma_roofline(prof)}
    mt.close()
    r["host_pointer_api"] = host_pointer(tree, ex_for_scalar, pts, dev)
    if rank == 0 and world == 1:
        r["nearest_triangle_vs_reference"] = reference_nearest(mesh, np.asarray(mesh.vertices), np.asarray(mesh.indices), box, dev)
    ex_for_scalar.close()
    if len(mesh.indices) >= 300_000:
        r["torus_knot_328k"] = knot_workload(mesh.ctx, dev, pts.shape[0])
    if len(mesh.indices) >= 300_000:          # the headline configuration only (short test runs of this script skip the 1.7 GB tree)
        r["deep_tree_d9"] = deep_tree(mesh, box, dev, prof)
        r["gather_calibration"] = gather_calibration(mesh.ctx, dev)
    return r


def exact_query_roofline(ex, pts, ms, prof, sample=20000):
    """SURVEY.md 8(d) "X": per query 16 B of I/O + k x 148 B of TriangleData + the packed-set bytes, k = triangles of the query's leaf after the
    mask chain.  The mean k is taken over a sample of the timed points by walking the DOWNLOADED arrays on the host (measurement code).
    The leaf-sorted kernel stages a leaf's triangles once per tile for all its queries, so the bytes it moves are far below this
    per-query figure (frac may exceed 1): the honest limiter is the k point/triangle evaluations per query, reported as a flop rate."""
    nodes, has, sets, masks = ex.download()
    i = ex.info
    bb = ex.get_grid_bounding_box(); G = int(i.start_grid_size); cell = float(i.start_grid_cell_size)
    p = pts[:sample].cpu().numpy().astype(np.float32)
    ks = np.zeros(len(p), dtype=np.int64); setbytes = np.zeros(len(p), dtype=np.int64)
    pop8 = np.array([bin(x).count("1") for x in range(256)], dtype=np.int64)
    def popfirst(off, n):                  # set bits among the first n bits (MSB first) of the mask starting at byte `off`
        full, rem = n // 8, n % 8
        c = int(pop8[masks[off:off + full]].sum())
        if rem: c += int(pop8[masks[off + full] >> (8 - rem)])
        return c
    for qi in range(len(p)):
        f = (p[qi] - bb[:3]) / np.float32(cell)
        ijk = np.floor(f).astype(np.int64)
        if (ijk < 0).any() or (ijk >= G).any(): continue
        f = f - np.floor(f)
        node = int((ijk[2] * G + ijk[1]) * G + ijk[0]); depth = int(i.start_depth)
        def child(nd, f):
            c = (4 if f[2] > 0.5 else 0) + (2 if f[1] > 0.5 else 0) + (1 if f[0] > 0.5 else 0)
            f = 2 * f; f = f - np.floor(f)
            return int(nodes[nd, 0] & 0x7FFFFFFF) + c, f
        leaf = lambda nd: bool(nodes[nd, 0] >> 31)
        while not leaf(node) and depth < int(i.bit_encoding_start_depth):
            node, f = child(node, f); depth += 1
        cnt = int(sets[nodes[node, 1]]); k = cnt
        setbytes[qi] = 4 + (cnt * int(i.bits_per_index) + 7) // 8
        if not leaf(node):
            node, f = child(node, f); k = popfirst(int(nodes[node, 1]), cnt)
            if not leaf(node):
                node, f = child(node, f); k = popfirst(int(nodes[node, 1]), k)
        ks[qi] = k
    mean_k = float(ks.mean()); bytes_q = 16 + 148 * mean_k + float(setbytes.mean())
    n = pts.shape[0]
    flops = 70.0 * mean_k * n / (ms * 1e-3)            # ~70 flop per fp32 point/triangle squared distance in the triangle's frame
    # compulsory: the streams + every array of the structure once (nodes, packed sets, byte masks, the packed 80-B triangle frames)
    compulsory = 16 * n + nodes.nbytes + sets.nbytes + masks.nbytes + 80 * int(i.num_triangles if hasattr(i, "num_triangles") else 0)
    r = roofline_block(EXACT_KERNEL, ms, bytes_q * n, compulsory, prof.traffic("exact_query", EXACT_KERNEL, 0))
    r["kernel_ms_note"] = "locate + radix sort by leaf + the sorted kernel (k_exact_lists since round 4: decoded leaf lists, pipelined tiles), HIP events around the whole call; traffic and VALU figures are the sorted kernel's"
    st = prof.stats_row(EXACT_KERNEL, 0) if not prof.stale("exact_query") else None
    if st:
        r["sorted_kernel_ms_profiled"] = round(float(st["avg_ns"]) / 1e6, 3); r["locate_and_sort_ms"] = round(ms - float(st["avg_ns"]) / 1e6, 3)
    r["bound_regime"] = "VALU issue (valu_issue_frac): the kernel reproduces the reference's separate fp32 multiplies and adds, no FMA contraction"
    r.update({"mean_k": round(mean_k, 1), "max_k_sampled": int(ks.max()), "bytes_per_query_8d": round(bytes_q, 1),
              "pair_evaluations_per_s": round(mean_k * n / (ms * 1e-3) / 1e9, 1), "tflop_s": round(flops / 1e12, 2), "fp32_vector_frac": round(flops / 157.3e12, 4), "lane_ops_peak_no_fma_tops": round(GPU_CLOCK_HZ * GPU_SIMDS * 16 / 1e12, 1),
              "note": "8(d) bytes are per query (k x 148 B of TriangleData each); a leaf's triangles are staged once per tile for all its queries, so the traffic that moves is "
                      "far below that figure and `achieved` counts the moved bytes only; the kernel is bound by its k distance evaluations per query (tflop_s, valu_lanes_active), not by bytes"})
    return r


EXACT_KERNEL = "sdfhip::k_exact_lists<false>"
MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 matrix peak


def mfma_roofline(prof):
    """The 64x64 tricubic fit on the matrix cores (k_fit_mfma, v_mfma_f32_32x32x2_f32): the one MFMA-bound kernel of the path.  The kernel runs
    inside a build, so its time and instruction counts come from the committed profile of this same command (kernel trace + SQ pass, accepted
    only if the kernel's sources are unchanged): flop = SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 per dispatch, time = the trace's average duration."""
    why = prof.stale("fit_mfma")
    if why:
        return {"bound": "mfma", "achieved": None, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "refused": why}
    best = None
    for slots in (4, 2, 1):
        k = f"sdfhip::k_fit_mfma<{slots}>"
        sq, st = prof.sq_row(k), prof.stats_row(k)
        if sq and st and sq.get("SQ_INSTS_VALU_MFMA_MOPS_F32_avg_per_dispatch"):
            best = (k, float(sq["SQ_INSTS_VALU_MFMA_MOPS_F32_avg_per_dispatch"]), float(st["avg_ns"]), float(sq.get("SQ_VALU_MFMA_BUSY_CYCLES_avg_per_dispatch") or 0),
                    float(sq.get("SQ_BUSY_CYCLES_avg_per_dispatch") or 0)); break
    if not best:
        return {"bound": "mfma", "achieved": None, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "refused": f"no k_fit_mfma rows in {prof.prefix}"}
    k, mops, ns, busy, total = best
    tf = mops * 512 / ns / 1e3
    return {"bound": "mfma", "kernel": k.split("::")[-1], "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4),
            "mfma_mops_per_dispatch": int(mops), "avg_kernel_us": round(ns / 1e3, 2), "mfma_busy_cycles_per_dispatch": int(busy),
            "source": f"{prof.prefix}_pmc_sq.csv + _kernel_stats.csv (sources unchanged since)",
            "note": "hi/lo split: two exact-in-fp32 passes per node tile; the fit is 0.1 % of a build, the figure is reported because the north star asks for it"}
DEEP_QUERIES = 12_000_000      # not 10 M: the profile summaries tell the two launches of the same kernel apart by grid size


DEEP_THRESHOLD = 1e-4


def band_cell_points(t, dev, want=8_000_000, res=512, seed=4321):
    """Centres of `want` DISTINCT depth-9 lattice cells nearest the surface, in random order: the tree's own values on the 512^3 lattice pick the
    band |d| < w (w by bisection), so almost every point falls into a depth-9 leaf of its own -> every 256-byte block is fetched once."""
    bb = t.get_grid_bounding_box(); size = float(bb[3] - bb[0])
    step = np.full(3, size / res, dtype=np.float32); origin = (bb[:3] + 0.5 * step).astype(np.float32)
    d = t.get_distance_grid(origin, step, (res, res, res), gradient=False, eval_mode=S.EVAL_FAST, device_out=True).abs_()
    lo, hi = 0.0, 64.0 * size / res
    for _ in range(24):
        mid = 0.5 * (lo + hi)
        if int((d < mid).sum().item()) > want: hi = mid
        else: lo = mid
    idx = torch.nonzero(d < lo).squeeze(1)
    del d
    g = torch.Generator(device=dev); g.manual_seed(seed)
    idx = idx[torch.randperm(idx.numel(), generator=g, device=dev)]
    x, y, z = idx % res, (idx // res) % res, idx // (res * res)
    o = torch.tensor(origin, device=dev); st = torch.tensor(step, device=dev)
    return (o + torch.stack([x, y, z], dim=1).to(torch.float32) * st).contiguous(), lo


def deep_tree(mesh, box, dev, prof):
    """The same query kernel on a tree that does NOT fit the 256 MB Infinity Cache: depth 9, threshold 1e-4 = 3.1 GB of node array (12 M leaves;
    the reference layout's 30-bit word index allows 4.29 GB, and 5e-5 already needs 4.9).  Two query sets:
      * `distinct_cells` (the roofline_hbm figure): one query per distinct depth-9 cell of the band around the surface, random order - every
        coefficient block is fetched ONCE, so at most the Infinity Cache's 256 MB of the ~2 GB gathered can be cache-served: an HBM figure;
      * `uniform_random`: 12 M volume-uniform points, the metric's distribution - 12 M such points fall into only ~2.6 M distinct leaves (large
        leaves are hit again and again), every XCD's L2 re-fetches them through the fabric and the Infinity Cache serves an unknown share of
        those re-reads: its fabric traffic (7.2 TB/s in profiles/r06d) exceeds what the HBM pins deliver - reported, but not an HBM fraction.
    (tests/test_gpu_octree.py::test_deep_tree_depth_9_matches_oracle checks the 2e-4 tree, 1.6 GB, against the oracle: same kernel, same builder.)"""
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t = S.OctreeSdf(mesh, box, 9, 3, DEEP_THRESHOLD, num_threads=2)
    torch.cuda.synchronize(); build_s = time.perf_counter() - t0
    i = t.info
    kern = "sdfhip::k_octree_query_coop<0,false>"
    gen = torch.Generator(device=dev); gen.manual_seed(4321)
    bb = t.get_grid_bounding_box(); size = float(bb[3] - bb[0])
    pts = (torch.tensor(bb[:3], device=dev) + torch.rand((DEEP_QUERIES, 3), generator=gen, device=dev) * (size * 0.999999)).contiguous()
    out = torch.empty(DEEP_QUERIES, dtype=torch.float32, device=dev)
    ms_u = _time_ms(lambda: t.get_distance(pts, eval_mode=S.EVAL_EXACT, out=out), reps=10)
    roof_u = octree_query_roofline(i, 3, DEEP_QUERIES, ms_u, False, prof, kern)
    del pts
    cpts, band = band_cell_points(t, dev)
    nq = int(cpts.shape[0])
    ms = _time_ms(lambda: t.get_distance(cpts, eval_mode=S.EVAL_EXACT, out=out[:nq]), reps=10)
    # algorithmic bytes of THIS query set: every query walks to a depth-9 leaf (7 dependent node words from the start grid) and reads its own block
    alg = nq * (16 + 256 + 4 * 7)
    distinct = nq * 256
    roof = roofline_block(kern, ms, alg, nq * 16 + distinct, prof.traffic("octree_query", kern, nq))
    roof.update({"bound_regime": "hbm", "queries": nq, "bytes_per_query": 300, "distinct_block_bytes": distinct,
                 "cache_servable_share_max": round(256 * 2 ** 20 / distinct, 3),
                 "hbm_bytes_at_least": int(distinct - 256 * 2 ** 20), "hbm_gb_s_at_least": round((distinct - 256 * 2 ** 20) / ms / 1e6, 1),
                 "regime": "HBM gather: one query per distinct depth-9 cell within %.4f of the surface, random order; each 256-byte block is needed once per launch" % band})
    t.close()
    return {"build_s": round(build_s, 4), "words": int(i.num_words), "leaves": int(i.num_leaves), "queries": nq, "query_ms": round(ms, 4),
            "mqueries_s": round(nq / ms / 1e3, 1), "roofline": roof,
            "uniform_random": {"queries": DEEP_QUERIES, "query_ms": round(ms_u, 4), "mqueries_s": round(DEEP_QUERIES / ms_u / 1e3, 1), "roofline": roof_u}}


def host_pointer(tree, ex, pts, dev):
    """The drop-in boundary as the reference's callers use it: HOST arrays in, host arrays out (PCIe inside the call), and the scalar
    getDistance.  Large calls are plain copies between the caller's pageable arrays and the context's buffers around the kernel (the
    in-place-pinned two-stream pipeline of rounds 2-3 was removed in round 4: it ended in a GPU memory access fault whose cause was never found).
    One-point calls are answered on host copies of the arrays.  The link ceilings are measured here with pinned torch tensors."""
    import ctypes as C
    from sdflib_amd._lib import lib, check
    n = pts.shape[0]
    hp = pts.cpu().numpy().copy()
    hd = np.empty(n, dtype=np.float32); hg = np.empty((n, 3), dtype=np.float32)
    def best(fn, reps=4):
        fn(); b = 1e30
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); b = min(b, time.perf_counter() - t0)
        return b
    t_val = best(lambda: tree.get_distance(hp, out=hd))
    t_grad = best(lambda: tree.get_distance(hp, gradient=True, out=hd, out_grad=hg))
    # link ceilings: pinned host tensors, one direction at a time
    pin_in = torch.empty((n, 3), dtype=torch.float32).pin_memory(); pin_out = torch.empty(n, dtype=torch.float32).pin_memory()
    dv = torch.empty((n, 3), dtype=torch.float32, device=dev); do = torch.empty(n, dtype=torch.float32, device=dev)
    def h2d(): dv.copy_(pin_in, non_blocking=True); torch.cuda.synchronize()
    def d2h(): pin_out.copy_(do, non_blocking=True); torch.cuda.synchronize()
    t_up, t_down = best(h2d), best(d2h)
    up_gbs, down_gbs = 12 * n / t_up / 1e9, 4 * n / t_down / 1e9
    bound_seq = t_up + t_down                       # upload then download, kernel hidden: what one direction at a time allows
    # scalar calls through the C ABI: 64 different points, pointers converted once (a ctypes call costs ~0.5 us itself); best of 3 rounds
    # (the first thousand calls after a bulk phase run several times slower: interpreter / allocator warm-up, not the library)
    sp = hp[:64].copy(); d1 = np.empty(1, dtype=np.float32)
    dptr = C.c_void_p(d1.ctypes.data)
    ptrs = [C.c_void_p(sp[i].ctypes.data) for i in range(len(sp))]
    L = lib()
    def scalar(fn, rounds=3, reps=20):
        b = float("inf")
        for _ in range(rounds):
            t0 = time.perf_counter()
            for _ in range(reps):
                for q in ptrs: fn(q)
            b = min(b, (time.perf_counter() - t0) / (reps * len(ptrs)) * 1e6)
        return b
    us_oct = scalar(lambda q: L.sdfhip_octree_query(tree.h, q, 1, dptr, None, 0, S.EVAL_EXACT))
    us_ex = scalar(lambda q: L.sdfhip_exact_query(ex.h, q, 1, dptr, None, None, 0))
    path = "plain pageable copies, pieces of 2^20 points: upload + kernel of piece k + 1 overlap the download of piece k (two or three host threads, no registration of the caller's memory)"
    bound_ovl = max(t_up, t_down)
    return {"queries": int(n), "path": path, "value_ms": round(t_val * 1e3, 3), "host_pointer_mqueries_s": round(n / t_val / 1e6, 1), "value_and_gradient_ms": round(t_grad * 1e3, 3),
            "pcie_pinned_h2d_gb_s": round(up_gbs, 1), "pcie_pinned_d2h_gb_s": round(down_gbs, 1), "pcie_bound_ms": round(bound_seq * 1e3, 3),
            "frac_of_pcie_bound": round(bound_seq / t_val, 3), "pcie_overlapped_bound_ms": round(bound_ovl * 1e3, 3), "frac_of_overlapped_bound": round(bound_ovl / t_val, 3), "scalar_us_per_call_octree": round(us_oct, 2), "scalar_us_per_call_exact": round(us_ex, 2),
            "note": "pcie_bound = pinned upload + pinned download of the same arrays, one after the other; overlapped bound = the larger of the two (full duplex), both measured on this box with PINNED tensors - the call itself copies from and to PAGEABLE arrays; scalar = one point per call through the C ABI from ctypes, mean over 64 random points of the box"}


def knot_workload(ctx, dev, n):
    """The C2 / C3 pipeline on geometry that is not a displaced sphere: a 327 680-triangle tube around a (2,3) torus knot (genus 1, not
    star-shaped, thin, close to itself): BVH planner, both structures' builds, 10 M queries each."""
    from sdflib_amd.meshgen import torus_knot
    v, f = torus_knot()
    box = box_with_margin(v)
    t0 = time.perf_counter(); m = S.Mesh(v, f, ctx); prep = time.perf_counter() - t0
    bvh_s = m.build_bvh()
    builds = []
    for _ in range(2):          # the first build grows the context's scratch buffers for this mesh, the second is the steady state
        torch.cuda.synchronize(); t0 = time.perf_counter()
        t = S.OctreeSdf(m, box, 8, 3, 1e-3, num_threads=2)
        torch.cuda.synchronize(); builds.append(time.perf_counter() - t0)
    build_s = builds[1]
    i = t.info
    gen = torch.Generator(device=dev); gen.manual_seed(77)
    bb = t.get_grid_bounding_box(); size = float(bb[3] - bb[0])
    pts = (torch.tensor(bb[:3], device=dev) + torch.rand((n, 3), generator=gen, device=dev) * (size * 0.999999)).contiguous()
    out = torch.empty(n, dtype=torch.float32, device=dev)
    ms = _time_ms(lambda: t.get_distance(pts, eval_mode=S.EVAL_EXACT, out=out))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ex = S.ExactOctreeSdf(m, box, 7, 3, 128)
    torch.cuda.synchronize(); ebuild = time.perf_counter() - t0
    ems = _time_ms(lambda: ex.get_distance(pts, out=out), reps=3)
    r = {"triangles": int(len(f)), "mesh_prep_s": round(prep, 4), "bvh_build_s": round(bvh_s, 4), "octree_build_s": round(build_s, 4), "octree_first_build_s": round(builds[0], 4), "words": int(i.num_words), "leaves": int(i.num_leaves),
         "bvh_traversals": int(i.num_traversals), "nearest_fallbacks": int(i.num_nearest_fallbacks), "query_ms": round(ms, 4), "mqueries_s": round(n / ms / 1e3, 1),
         "exact_build_s": round(ebuild, 4), "exact_nodes": int(ex.info.num_nodes), "exact_max_triangles_in_leafs": int(ex.info.max_triangles_in_leafs),
         "exact_query_ms": round(ems, 3), "exact_mqueries_s": round(n / ems / 1e3, 1)}
    ex.close(); t.close()
    return r


def gather_calibration(ctx, dev):
    """One launch pair of the counter-calibration kernel (see _traffic_block): also a direct measurement of the 256-B-gather ceiling from HBM."""
    import ctypes as C
    from sdflib_amd._lib import lib, check
    data = torch.empty(64 * CALIB_BLOCKS, dtype=torch.int32, device=dev).fill_(1)
    ids = torch.randperm(CALIB_BLOCKS, device=dev, dtype=torch.int64).to(torch.int32).contiguous()
    out = torch.empty(CALIB_BLOCKS, dtype=torch.float32, device=dev)
    fn = lambda: check(lib().sdfhip_test_gather_blocks(ctx.h, C.c_void_p(data.data_ptr()), C.c_void_p(ids.data_ptr()), CALIB_BLOCKS, C.c_void_p(out.data_ptr())))
    ms = _time_ms(fn, reps=5)
    byts = CALIB_BLOCKS * (256 + 4 + 4)
    return {"blocks": CALIB_BLOCKS, "ms": round(ms, 4), "known_bytes": byts, "gb_s": round(byts / ms / 1e6, 1), "hbm_frac": round(byts / ms / 1e6 / HBM_PEAK_GBS, 4),
            "note": "random permutation of 256-B blocks over 2.56 GB, fetched like the query kernel fetches a leaf (16 lanes x dwordx4 per block, rows through LDS): the HBM ceiling of that access pattern"}


IC_BLOCKS = 781_250            # 200 MB of 256-byte blocks: inside the 256 MB Infinity Cache, fifty times one XCD's L2


def ic_gather_ceiling(ctx, dev, lanes=9_000_000):        # (not 10 M: the profile summaries tell launches of one kernel apart by grid size)
    """The rate at which L2 -> fabric -> Infinity Cache serves random 256-byte gathers: the query kernel's cooperative block loads
    (k_gather_blocks_coop) on a 200 MB array, block ids uniform-random (with repeats, like leaves), bytes per lane = 256 block + 4 id + 4 out."""
    import ctypes as C
    from sdflib_amd._lib import lib, check
    data = torch.empty(64 * IC_BLOCKS, dtype=torch.int32, device=dev).fill_(1)
    g = torch.Generator(device=dev); g.manual_seed(99)
    ids = torch.randint(0, IC_BLOCKS, (lanes,), generator=g, device=dev, dtype=torch.int64).to(torch.int32).contiguous()
    out = torch.empty(lanes, dtype=torch.float32, device=dev)
    fn = lambda: check(lib().sdfhip_test_gather_blocks(ctx.h, C.c_void_p(data.data_ptr()), C.c_void_p(ids.data_ptr()), lanes, C.c_void_p(out.data_ptr())))
    ms = _time_ms(fn, reps=10)
    return {"array_mb": round(IC_BLOCKS * 256 / 1e6, 1), "lanes": lanes, "bytes_per_lane": 264, "ms": round(ms, 4), "gb_s": round(lanes * 264 / ms / 1e6, 1)}


def valu_calibration(ctx, dev, blocks=4096, iters=20000):
    """The chip's unpacked fp32 VALU issue rate, live: k_valu_peak = 8 independent v_fma_f32 chains per lane, 16 waves per SIMD."""
    import ctypes as C
    from sdflib_amd._lib import lib, check
    out = torch.empty(blocks * 256, dtype=torch.float32, device=dev)
    ms = _time_ms(lambda: check(lib().sdfhip_test_valu_peak(ctx.h, blocks, iters, C.c_void_p(out.data_ptr()))), reps=5)
    winst = blocks * 4 * 8 * iters
    rate = winst / ms * 1e3
    return {"ms": round(ms, 4), "wave_instructions_per_s": float(f"{rate:.4g}"), "t_lane_ops_s": round(rate * 64 / 1e12, 2),
            "implied_clock_ghz": round(rate * 2 / GPU_SIMDS / 1e9, 3), "implied_clock_basis": "a wave64 fp32 instruction occupies a CDNA4 SIMD for 2 cycles (157 TFLOP/s = 256 CUs x 4 SIMDs x 32 lanes x 2 flop x 2.4 GHz), 1024 SIMDs; GRBM_GUI_ACTIVE of the profiled run reads 2.05 GHz",
            "tflops_fma": round(rate * 128 / 1e12, 1)}


XGMI_LINK_GBS = 153.0       # per direction and link (7 links per GPU, point to point); MI355X_MICROARCH.md


def shard_forecast(mesh, box, depth, start_depth, steady, words):
    """What an N-GPU run of this build should show (N = 2, 4, 8), from THIS GPU: every rank's shard of the start cells is built here on its
    own (the other ranks absent, best of two), the slowest one is the sharded part; the serial part (mesh preparation + BVH) is what every
    rank repeats; the exchange is the all-gather-v of the shard bodies: each rank receives the other ranks' words, at best over its N - 1
    direct xGMI links in parallel.  A later SCALE curve can be held against these numbers."""
    from sdflib_amd import api
    serial = float(steady.get("mesh_prep_s", 0.0)) + float(steady.get("bvh_s_after_mesh", 0.0))
    G3 = 8 ** start_depth
    weights = sdist.cell_weights(mesh.vertices, box, start_depth)
    out = {"serial_s": round(serial, 4), "single_gpu_octree_s": steady.get("octree_s"), "tree_bytes": 4 * words}
    for world in (2, 4, 8):
        ranges = sdist.partition_cells(G3, world, weights)
        ts, sizes = [], []
        for rk in range(world):
            best = 1e9
            for _ in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                sh = api.OctreeShard(mesh, box, depth, start_depth, 1e-3, cells=ranges[rk]); torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
                nb = int(sh.info.body_words)
                sh.close()
            ts.append(best); sizes.append(4 * nb)
        recv = sum(sizes) - min(sizes)                     # the rank with the smallest shard receives the most
        exch = recv / ((world - 1) * XGMI_LINK_GBS * 1e9)
        out[f"n{world}"] = {"slowest_shard_s": round(max(ts), 4), "mean_shard_s": round(float(np.mean(ts)), 4), "exchange_bytes_received_max": int(recv),
                            "exchange_s_at_link_rate": round(exch, 5), "end_to_end_s": round(serial + max(ts) + exch, 4),
                            "speedup_over_one": round((serial + float(steady.get("octree_s", 0.0))) / (serial + max(ts) + exch), 2)}
    out["note"] = ("shards measured one at a time on this GPU; the sharded part does not shrink like 1/N because a launch of the nearest search costs ~0.45 ms "
                   "however few queries it holds (profiles/r06_near_small_batches.txt) and the serial part is repeated by every rank")
    return out


def build_1m(ctx, rank, world, dev, forecast=True):
    v, f = bumpy_icosphere(8)
    box = box_with_margin(v)
    t0 = time.perf_counter()
    mesh = S.Mesh(v, f, ctx)
    prep = time.perf_counter() - t0
    bvh_s = sdist.share_bvh(mesh, rank, world, dev) if world > 1 else mesh.build_bvh()      # N > 1: every rank builds it on its own device (host planner: rank 0 + broadcast)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    if world > 1:
        tree, binfo = sdist.build_octree_sharded(mesh, box, 8, 3, 1e-3, rank, world, dev)
    else:
        tree = S.OctreeSdf(mesh, box, 8, 3, 1e-3, num_threads=2); binfo = {}
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    i = tree.info
    r = {"triangles": int(len(f)), "octree_build_s": round(dt, 4), "mesh_prep_s": round(prep, 4), "bvh_build_s": round(bvh_s, 4), "bvh_built_on": BVH_BUILT_ON,
         "words": int(i.num_words), "leaves": int(i.num_leaves), **_r4(binfo)}
    tree.close()
    if world == 1:
        r["steady_state"] = end_to_end_build(ctx, v, f, box, 8, 3, dev)
        r["end_to_end_s"] = r["steady_state"]["end_to_end_s"]
        if forecast:
            r["forecast"] = shard_forecast(mesh, box, 8, 3, r["steady_state"], int(i.num_words))
    else:
        # what north_star asks of the N-GPU build: seconds at N and where they go.  serial = what every rank repeats (mesh upload +
        # TriangleData, the sphere BVH - built by every rank on its own device, or planned by rank 0 and broadcast under
        # SDFHIP_BVH_BUILD=host); sharded = the slowest rank's shard of the start cells; exchange = sizes + all-gather-v + reductions
        one_device = os.environ.get("SDFHIP_BENCH_ONE_DEVICE") == "1"
        t = torch.tensor([prep, bvh_s, binfo.get("shard_build_s", 0.0), binfo.get("exchange_s", 0.0)], dtype=torch.float64, device=torch.device("cpu") if one_device else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        prep_m, bvh_m, shard_m, exch_m = (float(x) for x in t.tolist())
        r["end_to_end_s"] = round(prep_m + bvh_m + dt, 4)
        r["n_gpus"] = world
        r["split"] = {"serial_s": round(prep_m + bvh_m, 4), "serial_mesh_prep_s": round(prep_m, 4), "serial_bvh_s": round(bvh_m, 4), "sharded_s": round(shard_m, 4),
                      "exchange_s": round(exch_m, 4), "exchange_bytes_per_rank": int(binfo.get("exchange_bytes", 0)),
                      "bvh": "every rank builds it on its own device" if BVH_BUILT_ON == "device" else "planned by rank 0, broadcast",
                      "note": "max over ranks of every part; octree_build_s = sharded + exchange between two barriers"}
    return r


if __name__ == "__main__":
    main()
