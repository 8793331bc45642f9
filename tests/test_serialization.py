"""`.bin` files in the reference's (restated cereal PortableBinary) layout: CPU round trip + GPU save/load/query."""
import struct

import numpy as np
import pytest

from conftest import bits


def test_octree_file_layout_and_round_trip(tmp_path):
    from sdflib_amd import serialization as ser
    words = np.arange(100, dtype=np.uint32) * 7
    box = np.array([-1, -2, -3, 1, 2, 3], np.float32)
    p = str(tmp_path / "o.bin")
    ser.save_octree(p, box, 4, 8, 1.5, 0.25, words)
    raw = open(p, "rb").read()
    assert raw[0] == 1 and struct.unpack_from("<i", raw, 1)[0] == ser.FORMAT_OCTREE
    assert len(raw) == 1 + 4 + 24 + 16 + 8 + 400 and struct.unpack_from("<Q", raw, 45)[0] == 100
    kind, d = ser.load(p)
    assert kind == "octree" and d["start_grid_size"] == 4 and d["max_depth"] == 8 and d["value_range"] == 1.5 and d["min_border_value"] == 0.25
    assert np.array_equal(d["words"], words) and np.array_equal(d["box"], box)
    with pytest.raises(ValueError):
        open(p, "wb").write(b"\x00abcd"); ser.load(p)


@pytest.mark.gpu
def test_save_load_query_round_trip_on_gpu(tmp_path, gpu_ctx):
    import sdflib_amd as S
    from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin, random_points_in_box
    v, f = bumpy_icosphere(2)
    box = box_with_margin(v)
    m = S.Mesh(v, f, gpu_ctx)
    pts = random_points_in_box(box, 20000, seed=1)
    t = S.OctreeSdf(m, box, 5, 2, 1e-3, init_algorithm=S.ALG_CONTINUITY, num_threads=2)
    t.save_to_file(str(tmp_path / "oct.bin"))
    t2 = S.load_from_file(str(tmp_path / "oct.bin"), gpu_ctx)
    assert isinstance(t2, S.OctreeSdf) and np.array_equal(t.get_octree_data(), t2.get_octree_data())
    assert np.array_equal(bits(t.get_distance(pts, gradient=True)[1]), bits(t2.get_distance(pts, gradient=True)[1]))
    assert np.array_equal(bits(t.get_distance(pts)), bits(t2.get_distance(pts)))
    e = S.ExactOctreeSdf(m, box, 5, 1, 16)
    e.save_to_file(str(tmp_path / "ex.bin"), m)
    e2 = S.load_from_file(str(tmp_path / "ex.bin"), gpu_ctx)
    assert isinstance(e2, S.ExactOctreeSdf)
    assert np.array_equal(bits(e.get_distance(pts)), bits(e2.get_distance(pts)))
    assert e2.info.num_nodes == e.info.num_nodes and e2.info.bits_per_index == e.info.bits_per_index
