"""Sharded OctreeSdf construction: one process per GPU, start-grid cells partitioned across ranks.

The reference's own OpenMP decomposition makes every start-grid cell an independent sub-octree
(src/sdf/OctreeSdfDepthFirst.h:433-503); here the cells are split into contiguous z-major ranges, every rank
builds its range on its GPU, and ONE exchange reassembles the node array:
    all-gather(body sizes) -> prefix sums give every rank's absolute body offset
    rank r emits its start-grid words and bodies with ABSOLUTE indices (no rebase pass needed afterwards)
    all-gather(padded bodies), all-gather(padded grid slices)  [RCCL over xGMI; gloo in the CPU tests]
    max-reduce(valueRange), min-reduce(minBorderValue)         (OctreeSdfDepthFirst.h:505-509)
Every rank ends with the full, identical array (queries are then embarrassingly parallel).
The partition / assembly logic is backend-agnostic torch code so that it is covered by world_size-2 gloo tests.
"""
import time

import numpy as np
import torch
import torch.distributed as dist


def partition_cells(num_cells, world, weights=None):
    """Contiguous ranges [(begin, end)] * world covering [0, num_cells), balanced by `weights` (>= 1 cell each)."""
    assert 1 <= world <= num_cells
    if weights is None:
        weights = np.ones(num_cells, dtype=np.float64)
    w = np.asarray(weights, dtype=np.float64) + 1e-9
    cum = np.cumsum(w)
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        c = int(np.searchsorted(cum, target, side="left")) + 1
        c = max(c, cuts[-1] + 1)
        c = min(c, num_cells - (world - r))
        cuts.append(c)
    cuts.append(num_cells)
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def cell_weights(vertices, box6, start_depth):
    """Cheap work estimate per start-grid cell: vertices inside the cell and its 26 neighbours (surface cells subdivide)."""
    box6 = np.asarray(box6, dtype=np.float32)
    lo, hi = box6[:3], box6[3:]
    size = float((hi - lo).max())
    cmin = (lo + 0.5 * (hi - lo)) - 0.5 * size
    G = 1 << start_depth
    ijk = np.clip(((np.asarray(vertices) - cmin) / (size / G)).astype(np.int64), 0, G - 1)
    occ = np.zeros((G, G, G), dtype=np.float64)            # [z, y, x]
    np.add.at(occ, (ijk[:, 2], ijk[:, 1], ijk[:, 0]), 1.0)
    pad = np.pad(occ, 1)
    acc = np.zeros_like(occ)
    for dz in range(3):
        for dy in range(3):
            for dx in range(3):
                acc += pad[dz:dz + G, dy:dy + G, dx:dx + G]
    return acc.reshape(-1) + 1.0


def exchange_and_assemble(grid_local, body_local, body_words, cells, num_cells, group=None):
    """Collective part: returns the full node array (int32 view of the u32 words) on every rank.

    grid_local: this rank's start-grid words (len = cells[1]-cells[0]); body_local: its bodies with ABSOLUTE indices
    already applied for offset = num_cells + sum(body_words of lower ranks)."""
    world = dist.get_world_size(group)
    dev = body_local.device
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([int(body_words), cells[0], cells[1]], dtype=torch.int64, device=dev)
    meta = [torch.zeros(3, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(meta, mine, group=group)
    meta = torch.stack(meta).cpu().numpy()
    sizes = meta[:, 0]; begins = meta[:, 1]; ends = meta[:, 2]
    max_body = int(sizes.max()); max_cells = int((ends - begins).max())
    pb = torch.zeros(max(max_body, 1), dtype=torch.int32, device=dev); pb[:int(body_words)] = body_local[:int(body_words)]
    pg = torch.zeros(max_cells, dtype=torch.int32, device=dev); pg[:cells[1] - cells[0]] = grid_local[:cells[1] - cells[0]]
    gb = torch.empty(world * max(max_body, 1), dtype=torch.int32, device=dev)
    gg = torch.empty(world * max_cells, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(gb, pb, group=group)
    dist.all_gather_into_tensor(gg, pg, group=group)
    total = num_cells + int(sizes.sum())
    full = torch.empty(total, dtype=torch.int32, device=dev)
    off = num_cells
    for r in range(world):
        full[int(begins[r]):int(ends[r])] = gg[r * max_cells: r * max_cells + int(ends[r] - begins[r])]
        full[off: off + int(sizes[r])] = gb[r * max(max_body, 1): r * max(max_body, 1) + int(sizes[r])]
        off += int(sizes[r])
    return full


def body_offset_for_rank(body_words, num_cells, group=None, device=None):
    """Absolute word offset of this rank's bodies = num_cells + sum of lower ranks' body sizes (one tiny all-gather)."""
    world = dist.get_world_size(group); rank = dist.get_rank(group)
    t = torch.tensor([int(body_words)], dtype=torch.int64, device=device)
    lst = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(lst, t, group=group)
    sizes = [int(x.item()) for x in lst]
    return num_cells + sum(sizes[:rank]), sizes


def build_octree_sharded(mesh, box, depth, start_depth, max_error, rank, world, dev, group=None):
    """Sharded GPU build + RCCL reassembly.  Returns (OctreeSdf with the full array on this GPU, timing dict)."""
    from . import api
    num_cells = 8 ** start_depth
    ranges = partition_cells(num_cells, world, cell_weights(mesh.vertices, box, start_depth))
    t0 = time.perf_counter()
    shard = api.OctreeShard(mesh, box, depth, start_depth, max_error, cells=ranges[rank])
    info = shard.info
    t1 = time.perf_counter()
    offset, _ = body_offset_for_rank(info.body_words, num_cells, group, dev)
    ncell = ranges[rank][1] - ranges[rank][0]
    grid_local = torch.empty(ncell, dtype=torch.int32, device=dev)
    body_local = torch.empty(max(int(info.body_words), 1), dtype=torch.int32, device=dev)
    shard.emit(offset, grid_local, body_local)
    full = exchange_and_assemble(grid_local, body_local, info.body_words, ranges[rank], num_cells, group)
    stats = torch.tensor([info.value_range, -info.min_border_value], dtype=torch.float32, device=dev)
    dist.all_reduce(stats, op=dist.ReduceOp.MAX, group=group)
    lpd = torch.tensor(list(info.leaves_per_depth), dtype=torch.int64, device=dev)
    cnt = torch.tensor([info.num_leaves, info.num_nodes, info.num_samples], dtype=torch.int64, device=dev)
    dist.all_reduce(lpd, group=group); dist.all_reduce(cnt, group=group)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    tree = api.OctreeSdf.from_data(mesh.ctx, full, info.box_min, info.box_max, info.start_grid_size, info.max_depth,
                                   float(stats[0].item()), float(-stats[1].item()), where=api.DEVICE)
    tree._override = {"leaves_per_depth": [int(x) for x in lpd.cpu().tolist()], "num_leaves": int(cnt[0]), "num_nodes": int(cnt[1]),
                      "num_samples": int(cnt[2])}
    shard.close()
    return tree, {"shard_build_s": t1 - t0, "exchange_s": t2 - t1}
