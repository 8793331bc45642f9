"""bench.py's roofline accounting (CPU): no fraction can exceed what moved, stale counter profiles are refused, the committed profile
belongs to the sources in the tree."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_achieved_never_exceeds_algorithmic_or_moved_bytes():
    import bench
    rng = np.random.default_rng(0)
    for _ in range(200):
        alg, comp = float(rng.uniform(1e6, 1e10)), float(rng.uniform(1e5, 1e9))
        traffic = None if rng.random() < 0.3 else int(rng.uniform(1e5, 1e10))
        ms = float(rng.uniform(0.01, 10))
        r = bench.roofline_block("sdfhip::k", ms, alg, comp, {"traffic": traffic})
        moved = traffic if traffic else comp
        assert r["achieved"] <= min(alg, moved) / (ms * 1e-3) / 1e9 + 0.1
        assert abs(r["frac"] - r["achieved"] / bench.HBM_PEAK_GBS) < 1e-3
        assert ("measured traffic" in r["frac_basis"]) == bool(traffic)
    # the round-2 headline: 2.85 GB algorithmic, 1.91 GB moved, 0.33 ms -> 0.72, not 1.08
    r = bench.roofline_block("sdfhip::k", 0.3316, 2.8537e9, 2.4e8, {"traffic": 1909740902})
    assert 0.70 < r["frac"] < 0.74 and r["algorithmic_over_traffic"] > 1.4


def test_stale_profiles_are_refused_and_the_committed_one_is_current(tmp_path):
    import bench
    p = bench.Profile()
    assert p.prefix, "no committed profile with a meta file under profiles/"
    # the headline kernel's counters must belong to the sources in the tree (the other groups are reported: bench.py falls back to
    # the compulsory bytes for them and says so)
    assert p.stale("octree_query") is None, f"the newest committed profile ({p.prefix}) was recorded with other sources: {p.stale('octree_query')} - run tools/profile_bench.sh + tools/keep_profile.sh"
    import warnings
    for group in bench.KERNEL_SOURCES:
        if p.stale(group): warnings.warn(f"profile {p.prefix} is stale for {group}: {p.stale(group)}")
    # a changed source file must be noticed
    victim = bench.KERNEL_SOURCES["exact_query"][0]
    p.now = dict(p.now); p.now[victim] = "0" * 16
    assert victim in p.stale("exact_query")
    t = p.traffic("exact_query", bench.EXACT_KERNEL, 0)
    assert t["traffic"] is None and victim in t["traffic_refused"]
    assert p.stale("octree_query") is None or victim in bench.KERNEL_SOURCES["octree_query"]
    meta = json.load(open(os.path.join(ROOT, p.prefix + "_meta.json")))
    assert set(meta["source_hashes"]) >= {f for fs in bench.KERNEL_SOURCES.values() for f in fs}
