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
