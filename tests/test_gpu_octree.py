"""GPU parity: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


def _mesh(s):
    from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin
    v, f = bumpy_icosphere(s)
    return v, f, box_with_margin(v)


@pytest.fixture(scope="module")
def small(oracle, gpu_ctx):
    import sdflib_amd as S
    v, f, box = _mesh(3)
    return dict(v=v, f=f, box=box, om=oracle.Mesh(v, f), gm=S.Mesh(v, f, gpu_ctx))


def test_triangle_data_matches_oracle(small):
    a = small["om"].triangle_data()
    b = small["gm"].triangle_data()
    # frames and edge pseudonormals: bit-exact
    assert np.array_equal(bits(a[:, :28]), bits(b[:, :28]))
    # vertex pseudonormals go through acosf (glibc vs ocml may differ in the last ulp): tolerance 1e-5
    np.testing.assert_allclose(a[:, 28:], b[:, 28:], rtol=0, atol=1e-5)


def test_nearest_triangle_ids_bit_exact(small, oracle):
    rng = np.random.default_rng(7)
    pts = ((rng.random((20000, 3), dtype=np.float32) * 2 - 1) * 1.4).astype(np.float32)
    ids_o = small["om"].nearest(pts)
    ids_g = small["gm"].nearest_triangle(pts)
    assert np.array_equal(ids_o, ids_g)


def test_point_values_match(small):
    rng = np.random.default_rng(8)
    pts = ((rng.random((5000, 3), dtype=np.float32) * 2 - 1) * 1.4).astype(np.float32)
    ids = small["om"].nearest(pts)
    a = small["om"].point_values(pts, ids)
    b = small["gm"].point_values(pts, ids)
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-5)
    # everything except a possible sign decision through the vertex pseudonormal is bit-exact
    assert (bits(a) == bits(b)).mean() > 0.999


def test_tricubic_fit_bit_exact(oracle, gpu_ctx):
    import sdflib_amd as S
    rng = np.random.default_rng(3)
    vals = (rng.standard_normal((512, 8, 8))).astype(np.float32)
    ns = (0.01 + rng.random(512)).astype(np.float32)
    ref = np.stack([oracle.tricubic_fit(vals[i], ns[i]) for i in range(512)])
    got = S.tricubic_fit(vals, ns, gpu_ctx)
    assert np.array_equal(bits(ref), bits(got))


@pytest.mark.parametrize("depth,start,layout", [(5, 2, 1), (5, 2, 0), (4, 0, 1), (4, 1, 0), (6, 3, 1)])
def test_octree_build_bit_exact_topology_and_coefficients(small, oracle, depth, start, layout):
    import sdflib_amd as S
    oc = oracle.Octree(small["om"], small["box"], depth, start, 1e-3, vertex_cache=False, layout=layout)
    gt = S.OctreeSdf(small["gm"], small["box"], depth, start, 1e-3, num_threads=2 if layout == 1 else 1)
    a, b = oc.data(), gt.get_octree_data()
    assert a.shape == b.shape
    assert np.array_equal(a, b), f"first mismatch at word {np.flatnonzero(a != b)[:5]}"
    i = gt.info
    assert i.start_grid_size == oc.start_grid_size
    assert np.float32(i.value_range) == np.float32(oc.value_range)
    assert np.float32(i.min_border_value) == np.float32(oc.min_border)
    np.testing.assert_array_equal(np.array(list(i.box_min) + list(i.box_max), dtype=np.float32), oc.box)


def test_octree_query_exact_and_fast(small, oracle):
    import sdflib_amd as S
    from sdflib_amd.meshgen import random_points_in_box
    oc = oracle.Octree(small["om"], small["box"], 6, 3, 1e-3)
    gt = S.OctreeSdf(small["gm"], small["box"], 6, 3, 1e-3)
    pts = random_points_in_box(small["box"], 200000, seed=11)
    # a few points outside the grid as well
    pts[:200] *= 3.0
    d_o, g_o = oc.query(pts, grad=True)
    d_g, g_g = gt.get_distance(pts, gradient=True, eval_mode=S.EVAL_EXACT)
    inside = np.ones(len(pts), bool); inside[:200] = False
    assert np.array_equal(bits(d_o), bits(d_g))
    assert np.array_equal(bits(g_o[inside]), bits(g_g[inside]))
    d_f, g_f = gt.get_distance(pts, gradient=True, eval_mode=S.EVAL_FAST)
    np.testing.assert_allclose(d_f, d_o, rtol=0, atol=1e-5)
    np.testing.assert_allclose(g_f[inside], g_o[inside], rtol=0, atol=2e-4)
    d_v = gt.get_distance(pts, eval_mode=S.EVAL_EXACT)
    assert np.array_equal(bits(d_v), bits(d_o))
