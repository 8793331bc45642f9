"""A bounded, seeded slice of the differential fuzzers in the suite (they ran by hand before: tools/gpu_fuzz.py, tools/gpu_fuzz_ranks.py).

Every case draws a random mesh (spheres, cube, nested / disjoint components, triangle soups; anisotropic transforms, noise, holes,
coincident duplicates, flipped winding, degenerate triangles, scales 1e-3 .. 1e3 far from the origin), box, depth, rule, layout and
builder, and compares the GPU (through the C ABI) with the CPU oracle bit for bit: TriangleData, nearest-triangle ids (the two-phase
search, whose error bounds this is the standing check of), node arrays of both OctreeSdf builders, the ExactOctreeSdf arrays, queries
with gradients, shards == single build, lattice == point queries.  A second test drives the nearest search's escape routes on purpose:
more than sixteen candidates within rounding of the minimum (candidate overflow) and more than NEAR_MAX_TIES exact ties, both of which
must leave through k_near_fallback and still return the reference's id."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import ROOT, bits

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_seeded_fuzz_slice_against_the_oracle(oracle):
    import gpu_fuzz
    from conftest import SOAK
    budget, cases, t0, skipped = (60.0 if SOAK else 15.0), 0, time.time(), 0
    for seed in range(910000, 910200):
        if time.time() - t0 > budget:
            break
        r = gpu_fuzz.one(seed)          # raises AssertionError with the differing quantity on a mismatch
        cases += 1
        skipped += r == "skip"
    assert cases - skipped >= (40 if SOAK else 10), f"only {cases} cases in {budget:.0f} s"


@pytest.mark.parametrize("mode", ["continuity", "exact"])
def test_seeded_fuzz_slice_biased(oracle, mode):
    """The same generator biased towards the CONTINUITY builder / deeper ExactOctreeSdf trees (FUZZ_MODE of the tool): 8 s each (20 s
    with SDFHIP_TEST_SOAK=1)."""
    import gpu_fuzz
    from conftest import SOAK
    old = gpu_fuzz.MODE
    gpu_fuzz.MODE = mode
    try:
        t0, cases = time.time(), 0
        base = 920000 if mode == "continuity" else 930000
        for seed in range(base, base + 100):
            if time.time() - t0 > (20.0 if SOAK else 8.0):
                break
            gpu_fuzz.one(seed); cases += 1
        assert cases >= 5
    finally:
        gpu_fuzz.MODE = old


def _fan(n, height=0.0, radius=1.0):
    """n triangles around the origin sharing the apex (0, 0, height): for points above the apex every triangle is at the same distance."""
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    ring = np.stack([radius * np.cos(ang), radius * np.sin(ang), np.zeros(n)], 1)
    v = np.concatenate([[[0, 0, height]], ring]).astype(np.float32)
    f = np.array([[0, 1 + i, 1 + (i + 1) % n] for i in range(n)], np.uint32)
    return v, f


def test_nearest_search_escape_routes_return_the_reference_id(oracle, gpu_ctx):
    import sdflib_amd as S
    from sdflib_amd.meshgen import box_with_margin, bumpy_icosphere
    rng = np.random.default_rng(7)
    cases = []
    # (a) 40 triangles meeting in one vertex: above the apex all 40 tie (candidate list of 16 overflows)
    cases.append(("fan40", *_fan(40, height=0.3)))
    # (b) a flat fan: every point of the axis is equidistant from all of them, on both sides
    cases.append(("flatfan24", *_fan(24)))
    # (c) 14 coincident copies of every triangle of a small mesh: more exact ties than NEAR_MAX_TIES everywhere
    v, f = bumpy_icosphere(1)
    cases.append(("copies14", v, np.concatenate([f] * 14)))
    # (d) the same under an extreme scale / offset
    cases.append(("copies12_scaled", (v * np.float32(2e-3) + np.float32(37.0)).astype(np.float32), np.concatenate([f] * 12)))
    total_fallbacks = 0
    for name, v, f in cases:
        v = np.ascontiguousarray(v, np.float32); f = np.ascontiguousarray(f, np.uint32)
        box = box_with_margin(v, margin=0.3)
        om, gm = oracle.Mesh(v, f), S.Mesh(v, f, gpu_ctx)
        size = float((box[3:] - box[:3]).max())
        pts = (box[:3] + rng.random((30000, 3), dtype=np.float32) * size).astype(np.float32)
        axis = np.zeros((4000, 3), np.float32); axis[:, 2] = np.linspace(-size, size, 4000)       # on the fan's axis: all triangles tie
        axis += v[0] * np.float32([1, 1, 0])
        onv = v[rng.integers(0, len(v), 2000)]                                                     # exactly on vertices
        q = np.ascontiguousarray(np.concatenate([pts, axis, onv]), np.float32)
        assert np.array_equal(om.nearest(q), gm.nearest_triangle(q)), name
        ot = oracle.Octree(om, box, 5, 2, 1e-3, vertex_cache=False, layout=oracle.LAYOUT_SUBTREES)
        gt = S.OctreeSdf(gm, box, 5, 2, 1e-3, num_threads=2)
        assert np.array_equal(ot.data(), gt.get_octree_data()), name
        total_fallbacks += int(gt.info.num_nearest_fallbacks)
    assert total_fallbacks > 0, "none of the constructed cases left through k_near_fallback: the test no longer exercises it"


def test_rank_fuzz_slice_on_one_gpu():
    """tools/gpu_fuzz_ranks.py: 2 and 3 ranks on this GPU (gloo collectives), sharded OctreeSdf / ExactOctreeSdf, CONTINUITY with shared
    traversals, broadcast — all identical to the single-process builds on every rank."""
    env = dict(os.environ, FUZZ_WORLDS="2,3", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_fuzz_ranks.py"), "3", "940000"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
