"""Meshes that are NOT closed manifolds, at BASELINE configs[1] scale (the synthetic stand-ins are closed: the path a real Bunny with
holes or an unwelded export takes — open edges, calculateMeshTriangleData's seam welding, src/utils/TriangleUtils.cpp:292-420 — would
otherwise only be exercised on 1 280 triangles):
  (a) the s=7 bumpy icosphere with five caps punched out (open boundary loops, nothing to weld),
  (b) the same 327 680 triangles as an unwelded soup (983 040 private vertices: EVERY edge is open and is re-paired by position).
Both go through sdfhip_mesh_create_ex with the loader's box (the mesh's bounding box, as Mesh(filePath) computes it): TriangleData bits and
the depth-7 NO_CONTINUITY node array must equal the oracle's."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def punched(v, f, caps=5, radius=0.16, seed=5):
    rng = np.random.default_rng(seed)
    c = v[f].mean(axis=1)
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    keep = np.ones(len(f), bool)
    for _ in range(caps):
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        keep &= np.arccos(np.clip(c @ d, -1, 1)) > radius
    return v, np.ascontiguousarray(f[keep])


@pytest.fixture(scope="module")
def base():
    from sdflib_amd.meshgen import bumpy_icosphere
    return bumpy_icosphere(7)


def check(oracle, gpu_ctx, v, f, what, expect_welded):
    import sdflib_amd as S
    from sdflib_amd.meshgen import box_with_margin
    bbox = np.concatenate([v.min(axis=0), v.max(axis=0)]).astype(np.float32)
    t0 = time.perf_counter()
    gm = S.Mesh(v, f, gpu_ctx, bbox=bbox)
    td_g = gm.triangle_data()
    prep_s = time.perf_counter() - t0
    st = gm.edge_stats()
    om = oracle.Mesh(v, f, bbox)
    td_o = om.triangle_data()
    assert st["unmatched_edges"] > 0, what
    assert (st["welded_half_edges"] > 0) == expect_welded, (what, st)
    bad = np.nonzero((bits(td_o) != bits(td_g)).any(axis=1))[0]
    assert len(bad) == 0, f"{what}: TriangleData of {len(bad)} triangles differs from the oracle's, first {bad[0]}"
    box = box_with_margin(v)
    gt = S.OctreeSdf(gm, box, 7, 3, 1e-3, num_threads=2)
    ot = oracle.Octree(om, box, 7, 3, 1e-3, vertex_cache=False, layout=oracle.LAYOUT_SUBTREES)
    go, oo = gt.get_octree_data(), ot.data()
    assert go.shape == oo.shape, (what, go.shape, oo.shape)
    assert np.array_equal(go, oo), what
    print(f"{what}: {len(f)} triangles, {len(v)} vertices, {st}; mesh preparation {prep_s * 1e3:.1f} ms (upload, frames, edge pairing, welding, download of TriangleData); {len(go)} words equal")
    return prep_s


def test_punched_sphere_at_c2_scale(oracle, gpu_ctx, base):
    v, f = punched(*base)
    assert 300_000 < len(f) < 327_680
    check(oracle, gpu_ctx, v, f, "s=7 with five caps punched out", expect_welded=False)


def test_unwelded_soup_at_c2_scale(oracle, gpu_ctx, base):
    from sdflib_amd.meshgen import triangle_soup
    v, f = triangle_soup(*base)
    assert len(v) == 983_040
    check(oracle, gpu_ctx, v, f, "s=7 as an unwelded soup", expect_welded=True)


def test_punched_soup_with_jittered_seams(oracle, gpu_ctx):
    """holes AND seams, the seam copies moved by less than the welding threshold (1e-5 / size) on part of the surface and by more on
    another part (those stay open): what a scanned, re-exported mesh looks like.  s=5: 20 480 triangles."""
    from sdflib_amd.meshgen import bumpy_icosphere, triangle_soup
    v, f = punched(*bumpy_icosphere(5), caps=3, radius=0.3)
    v, f = triangle_soup(v, f)
    rng = np.random.default_rng(11)
    size = float((v.max(axis=0) - v.min(axis=0)).max())
    j = rng.normal(size=v.shape).astype(np.float32)
    j /= np.linalg.norm(j, axis=1, keepdims=True)
    amp = np.where(v[:, 2] > 0.3, 3e-5 / size, np.where(v[:, 2] > -0.3, 3e-6 / size, 0.0)).astype(np.float32)      # above / below the threshold / exact copies
    v = (v + j * amp[:, None]).astype(np.float32)
    check(oracle, gpu_ctx, v, f, "punched s=5 soup with jittered seams", expect_welded=True)
