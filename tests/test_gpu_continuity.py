"""GPU parity of the CONTINUITY OctreeSdf builder (through the C ABI) vs the CPU oracle."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("subdiv,depth,start", [(2, 4, 2), (3, 5, 2), (3, 5, 1), (3, 6, 3), (4, 6, 3), (2, 4, 0)])
def test_continuity_build_bit_exact(oracle, gpu_ctx, subdiv, depth, start):
    import sdflib_amd as S
    from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin, random_points_in_box
    v, f = bumpy_icosphere(subdiv)
    box = box_with_margin(v)
    om = oracle.Mesh(v, f)
    oc = oracle.Octree(om, box, depth, start, 1e-3, continuity=True)
    gm = S.Mesh(v, f, gpu_ctx)
    gt = S.OctreeSdf(gm, box, depth, start, 1e-3, init_algorithm=S.ALG_CONTINUITY)
    a, b = oc.data(), gt.get_octree_data()
    assert a.shape == b.shape
    assert np.array_equal(a, b), f"first mismatch at word {np.flatnonzero(a != b)[:5]} of {len(a)}"
    assert np.float32(gt.info.value_range) == np.float32(oc.value_range)
    assert np.float32(gt.info.min_border_value) == np.float32(oc.min_border)
    pts = random_points_in_box(box, 50000, seed=3)
    assert np.array_equal(bits(oc.query(pts)), bits(gt.get_distance(pts)))
