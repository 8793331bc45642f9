"""Device-memory behaviour of the engine: the packed query layout replaces (not doubles) a compacted tree's node array and gives it back
bit for bit; the allocation caches keep a bounded amount after a large build and give everything back on request."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


def test_compacted_tree_answers_and_downloads_the_same_bits(oracle, gpu_ctx):
    import sdflib_amd as S
    from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin, random_points_in_box
    v, f = bumpy_icosphere(4)
    box = box_with_margin(v)
    gm = S.Mesh(v, f, gpu_ctx)
    pts = random_points_in_box(box, 50000, seed=3)
    for alg in (S.ALG_NO_CONTINUITY, S.ALG_CONTINUITY):
        # a BUILT tree is born with the query layout and without a resident copy of the reference's array (round 5): one copy on the device
        t = S.OctreeSdf(gm, box, 6, 2, 1e-3, init_algorithm=alg, num_threads=2)
        array_bytes = 4 * int(t.info.num_words)
        born = t.device_bytes()
        assert born < 1.1 * array_bytes, (born, array_bytes)          # layout = array + 4 bytes per node + block alignment
        d0, g0 = t.get_distance(pts, gradient=True)
        assert t.device_bytes() == born                               # the first query made nothing
        words = t.get_octree_data()                                   # rebuilt from the layout, transiently
        assert t.device_bytes() == born
        # the same tree arriving as an ARRAY (a loaded file, reassembled shards): the first query derives the layout, compact() drops the array
        i = t.info
        a = S.OctreeSdf.from_data(gpu_ctx, words, i.box_min, i.box_max, i.start_grid_size, i.max_depth, i.value_range, i.min_border_value, cell_size=i.start_grid_cell_size)
        d1, g1 = a.get_distance(pts, gradient=True)
        assert np.array_equal(bits(d0), bits(d1)) and np.array_equal(bits(g0), bits(g1))
        before = a.device_bytes()
        a.compact()
        after = a.device_bytes()
        assert after < 0.62 * before, (before, after)                 # array gone
        assert np.array_equal(a.get_octree_data(), words)             # rebuilt from the layout, transiently
        assert a.device_bytes() == after
        d2, g2 = a.get_distance(pts, gradient=True)
        assert np.array_equal(bits(d0), bits(d2)) and np.array_equal(bits(g0), bits(g2))
        n = 24
        bb = t.get_grid_bounding_box(); step = np.full(3, (bb[3] - bb[0]) / n, np.float32); org = (bb[:3] + 0.5 * step).astype(np.float32)
        dl = t.get_distance_grid(org, step, (n, n, n), eval_mode=S.EVAL_FAST)
        da = a.get_distance_grid(org, step, (n, n, n), eval_mode=S.EVAL_FAST)
        assert np.isfinite(dl).all() and np.array_equal(bits(dl), bits(da))
        t.close(); a.close()
    # an array with words that belong to no node keeps its array
    t = S.OctreeSdf(gm, box, 4, 1, 1e-3, num_threads=2)
    w = np.concatenate([t.get_octree_data(), np.arange(100, dtype=np.uint32)])
    i = t.info
    loose = S.OctreeSdf.from_data(gpu_ctx, w, i.box_min, i.box_max, i.start_grid_size, i.max_depth, i.value_range, i.min_border_value)
    loose.get_distance(pts[:1000])
    b0 = loose.device_bytes(); loose.compact()
    assert loose.device_bytes() == b0 and np.array_equal(loose.get_octree_data(), w)


def test_large_tree_holds_one_copy_and_the_caches_give_memory_back(oracle):
    """Depth-9 tree (1.6 GB node array, the largest transient blocks of any build in the suite), then a small build in the same
    context: ONE copy of the tree is on the device from the start (the builders emit the query layout), the context's caches sit below their high-water mark after the
    build, and after closing the tree and trimming the device is back where it started."""
    import torch
    import sdflib_amd as S
    from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin, random_points_in_box
    ctx = S.Context(0)
    v, f = bumpy_icosphere(7)
    box = box_with_margin(v)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    gm = S.Mesh(v, f, ctx)
    t = S.OctreeSdf(gm, box, 9, 3, 2e-4, num_threads=2)
    array_bytes = 4 * int(t.info.num_words)
    assert array_bytes > (1 << 30)
    mark = torch.cuda.mem_get_info(0)[1] // 32                            # the caches' default high-water mark: 1/32 of the device (SDFHIP_CACHE_KEEP_MB overrides)
    assert ctx.cached_bytes() <= mark + (1 << 30), (ctx.cached_bytes(), mark)         # blocks above the mark were freed when the build returned; the nearest search's lists and slabs with a live block stay
    pts = random_points_in_box(box, 100000, seed=1)
    assert t.device_bytes() < 1.1 * array_bytes, (t.device_bytes(), array_bytes)      # born with the layout only: no resident copy of the array
    d0 = t.get_distance(pts)
    assert t.device_bytes() < 1.1 * array_bytes, (t.device_bytes(), array_bytes)
    words = t.get_octree_data()                                           # rebuilt from the layout
    assert len(words) == int(t.info.num_words)
    raw = oracle.octree_query_raw(words, t.get_grid_bounding_box(), t.info.start_grid_size, t.info.min_border_value, pts)
    assert np.array_equal(bits(raw), bits(d0))
    del words
    t.close()
    small_v, small_f = bumpy_icosphere(4)
    sm = S.Mesh(small_v, small_f, ctx)
    st = S.OctreeSdf(sm, box_with_margin(small_v), 6, 3, 1e-3, num_threads=2)          # BASELINE configs[0] size
    st.get_distance(pts[:1000])
    st.close(); sm.close(); gm.close()
    ctx.trim(0)
    assert ctx.cached_bytes() == 0
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(0)[0]
    assert free0 - free1 < (256 << 20), (free0, free1)                    # what stays is the runtime's own (streams, code objects, pool slack)
    ctx.close()
