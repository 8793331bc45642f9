"""CPU (gloo, world_size 2) coverage of the N>1 path: cell partitioning and the one-exchange reassembly.

The shards come from the ORACLE's full tree (cut per start-grid cell and re-based the way a rank emits them), so the
collective/assembly logic of sdflib_amd/distributed.py is exercised without a GPU."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from sdflib_amd.distributed import partition_cells, cell_weights
from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin


def test_partition_cells_covers_everything():
    for n, w in ((512, 8), (512, 3), (8, 8), (64, 5), (1, 1)):
        r = partition_cells(n, w)
        assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:])) and all(e > b for b, e in r)
    v, _ = bumpy_icosphere(3)
    wts = cell_weights(v, box_with_margin(v), 3)
    r = partition_cells(512, 4, wts)
    loads = [wts[b:e].sum() for b, e in r]
    assert max(loads) < 1.6 * (sum(loads) / 4)


def test_more_ranks_than_cells_is_a_clear_error():
    from sdflib_amd.distributed import partition_cells
    with pytest.raises(ValueError, match="start_depth"):
        partition_cells(1, 2)
    assert partition_cells(8, 8) == [(i, i + 1) for i in range(8)]


def _subtree_words(data, root_word):
    """number of words of the body hanging under a start-grid word (block + descendants)"""
    if root_word & 0x80000000:
        return 64
    base = root_word & 0x3FFFFFFF
    return 8 + sum(_subtree_words(data, int(data[base + c])) for c in range(8))


def _rebase(words, data_from, delta, root_word):
    """add delta to every node word of the body under root_word (coefficients untouched); returns the new root word"""
    leaf = root_word & 0x80000000
    base = root_word & 0x3FFFFFFF
    if not leaf:
        for c in range(8):
            words[base + c - data_from] = _rebase(words, data_from, delta, int(words[base + c - data_from]))
    return (leaf | ((base + delta) & 0x3FFFFFFF))


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from sdflib_amd.distributed import exchange_and_assemble, body_offset_for_rank
    full = np.load(os.path.join(tmp, "full.npy"))
    G3 = 64
    ranges = partition_cells(G3, world)
    b, e = ranges[rank]
    # this rank's shard, first with shard-local indices (as if built alone), then emitted at its absolute offset
    sizes = [_subtree_words(full, int(full[c])) for c in range(G3)]
    starts = np.concatenate([[G3], G3 + np.cumsum(sizes)]).astype(np.int64)
    my_words = int(sum(sizes[b:e]))
    offset, all_sizes = body_offset_for_rank(my_words, G3)
    assert offset == starts[b] and sum(all_sizes) == starts[-1] - G3
    body = full[starts[b]:starts[e]].copy()
    grid = full[b:e].copy()
    out = exchange_and_assemble(torch.from_numpy(grid.view(np.int32)), torch.from_numpy(body.view(np.int32)), my_words, (b, e), G3)
    got = out.numpy().view(np.uint32)
    assert np.array_equal(got, full), f"rank {rank}: assembled array differs"
    t = torch.tensor([1.0 + rank, -(0.5 - rank)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t[0].item() == float(world)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_reassembly(tmp_path, oracle):
    v, f = bumpy_icosphere(2)
    box = box_with_margin(v)
    m = oracle.Mesh(v, f)
    full = oracle.Octree(m, box, 4, 2, 1e-3, vertex_cache=False, layout=oracle.LAYOUT_SUBTREES).data()
    np.save(os.path.join(tmp_path, "full.npy"), full)
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)


# ---- ExactOctreeSdf: partition in emission order, three arrays + scattered start-grid slots --------------------------------

def test_exact_emission_rank_is_the_dfs_order():
    from sdflib_amd.distributed import exact_emission_rank
    r = exact_emission_rank(2)                                   # 4^3 cells, z-major ids
    assert sorted(r.tolist()) == list(range(64))
    # the reference visits children 7..0: the first emitted cell is the (+x,+y,+z) corner, the last one the origin cell
    assert r[63] == 0 and r[0] == 63
    # one level: child c (bit0 = x) has rank 7 - c
    assert exact_emission_rank(1).tolist() == [7, 6, 5, 4, 3, 2, 1, 0]
    assert exact_emission_rank(0).tolist() == [0]


def _fake_exact_parts(world, num_cells, seed=3):
    """Synthetic shards with ragged sizes (one of them with empty masks) + the arrays a single build would hold."""
    from sdflib_amd.distributed import exact_emission_rank
    rng = np.random.default_rng(seed)
    order = np.argsort(exact_emission_rank(int(round(np.log2(num_cells) / 3))), kind="stable")
    ranges = partition_cells(num_cells, world)
    parts = []
    for r, (b, e) in enumerate(ranges):
        nb, ns, nm = int(rng.integers(8, 200)) * 8, int(rng.integers(5, 300)), (0 if r == 0 else int(rng.integers(1, 500)))
        parts.append(dict(cells=np.sort(order[b:e]), grid_nodes=rng.integers(0, 2**31, (e - b, 2)).astype(np.int32), grid_has=rng.integers(0, 2, e - b).astype(np.uint8),
                          body_nodes=rng.integers(0, 2**31, (nb, 2)).astype(np.int32), body_has=rng.integers(0, 2, nb).astype(np.uint8),
                          sets=rng.integers(0, 2**31, ns).astype(np.int32), masks=rng.integers(0, 256, nm).astype(np.uint8)))
    grid = np.zeros((num_cells, 2), np.int32); ghas = np.zeros(num_cells, np.uint8)
    for p in parts:
        grid[p["cells"]] = p["grid_nodes"]; ghas[p["cells"]] = p["grid_has"]
    expect = (np.concatenate([grid] + [p["body_nodes"] for p in parts]), np.concatenate([ghas] + [p["body_has"] for p in parts]),
              np.concatenate([p["sets"] for p in parts]), np.concatenate([p["masks"] for p in parts]))
    return ranges, parts, expect


def test_exact_assembly_and_offsets():
    from sdflib_amd.distributed import assemble_exact, exact_offsets
    ranges, parts, expect = _fake_exact_parts(3, 64)
    got = assemble_exact(parts, 64)
    assert all(np.array_equal(a, b) for a, b in zip(got, expect))
    sizes = [(len(p["body_nodes"]), len(p["sets"]), len(p["masks"])) for p in parts]
    offs = exact_offsets(sizes, 64)
    assert offs[0] == (64, 0, 0) and offs[2] == (64 + sizes[0][0] + sizes[1][0], sizes[0][1] + sizes[1][1], sizes[0][2] + sizes[1][2])


def _exact_worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from sdflib_amd.distributed import assemble_exact, _all_gather_padded
    ranges, parts, expect = _fake_exact_parts(world, 64)
    mine = {k: torch.from_numpy(v) for k, v in parts[rank].items() if k != "cells"}
    lens = {"grid_nodes": [e - b for b, e in ranges], "grid_has": [e - b for b, e in ranges], "body_nodes": [len(p["body_nodes"]) for p in parts],
            "body_has": [len(p["body_nodes"]) for p in parts], "sets": [len(p["sets"]) for p in parts], "masks": [len(p["masks"]) for p in parts]}
    gathered = {k: _all_gather_padded(mine[k], lens[k], None) for k in mine}
    got = assemble_exact([dict(cells=parts[r]["cells"], **{k: v[r] for k, v in gathered.items()}) for r in range(world)], 64)
    for a, b in zip(got, expect):
        assert np.array_equal(a.numpy(), b), f"rank {rank}: assembled ExactOctreeSdf arrays differ"
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_exact_exchange():
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_exact_worker, args=(2, port), nprocs=2, join=True)


def _sample_exchange_worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import ctypes
    from sdflib_amd.distributed import SampleExchange
    x = SampleExchange(None, rank, world, torch.device("cpu"))
    for n in (1, 127, 128, 1000, 128 * 7 + 5):
        ptr = x._acquire(None, n)
        buf = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_int32)), shape=(n,))
        assert not buf.any(), "acquire must hand out zeros"
        ids = (np.arange(n, dtype=np.int64) * 2654435761 % 1000003).astype(np.int32)        # what the traversals would return
        for b in range(rank, (n + 127) // 128, world):                                       # this rank's 128-sample blocks, dealt round-robin
            buf[128 * b: 128 * (b + 1)] = ids[128 * b: 128 * (b + 1)]
        assert x._all_reduce(None, n) == 0, x.error
        assert np.array_equal(buf, ids), f"rank {rank}: exchange of {n} ids incomplete"
    assert x.bytes_reduced == 4 * (1 + 127 + 128 + 1000 + 128 * 7 + 5)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sample_exchange():
    """The torch.distributed side of sdfhip_exchange (CONTINUITY build, traversals shared out): zero-filled buffer, every rank
    fills its round-robin blocks, one all-reduce completes it on both ranks."""
    port = 30500 + (os.getpid() % 2000)
    mp.spawn(_sample_exchange_worker, args=(2, port), nprocs=2, join=True)
