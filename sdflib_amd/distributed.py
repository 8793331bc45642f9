"""Sharded OctreeSdf / ExactOctreeSdf construction: one process per GPU, start-grid cells partitioned across ranks.

The reference's own OpenMP decomposition makes every start-grid cell an independent sub-octree
(src/sdf/OctreeSdfDepthFirst.h:433-503); here the cells are split into contiguous z-major ranges, every rank
builds its range on its GPU, and ONE exchange reassembles the node array:
    all-gather(body sizes) -> prefix sums give every rank's absolute body offset
    rank r emits its start-grid words and bodies with ABSOLUTE indices (no rebase pass needed afterwards)
    in-place all-gather-v of bodies and grid slices (rank r's segments broadcast from r at their final offsets: no padding)
                                                               [RCCL over xGMI; gloo in the CPU tests]
    max-reduce(valueRange), min-reduce(minBorderValue)         (OctreeSdfDepthFirst.h:505-509)
Every rank ends with the full, identical array (queries are then embarrassingly parallel).
The partition / assembly logic is backend-agnostic torch code so that it is covered by world_size-2 gloo tests.
"""
import ctypes as C
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


def partition_cells(num_cells, world, weights=None):
    """Contiguous ranges [(begin, end)] * world covering [0, num_cells), balanced by `weights` (>= 1 cell each)."""
    if not 1 <= world <= num_cells:
        raise ValueError(f"{world} ranks cannot share {num_cells} start-grid cells: the build shards by start cell (8^start_depth of them); "
                         "use a larger start_depth or fewer ranks")
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


def _collective_device(dev, group=None):
    """Device the collectives run on: the GPU with RCCL ('nccl'); host memory with gloo (CPU tests, and 2-rank tests that
    share one GPU), in which case shard outputs are staged through the host."""
    return torch.device("cpu") if dist.get_backend(group) == "gloo" else dev


def exchange_and_assemble(grid_local, body_local, body_words, cells, num_cells, group=None, stats=None):
    """Collective part: returns the full node array (int32 view of the u32 words) on every rank.
    `stats` (dict, optional) receives ranks_seen and bytes_all_gathered (the size of the assembled array: what the exchange left on THIS rank).

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
    total = num_cells + int(sizes.sum())
    if total >= (1 << 30):
        raise ValueError(f"assembled node array has {total} words: beyond the 30-bit node index of the reference layout")
    # In-place all-gather-v (the construction of csrc/multi.hip): every rank writes its grid slice and its bodies at their FINAL
    # position in the full array, then rank r's two segments are broadcast from r.  No padding to the largest shard (start cells are
    # balanced by an estimate; bodies can differ by 2x), no reassembly copies; with RCCL the broadcasts are enqueued back to back.
    full = torch.empty(total, dtype=torch.int32, device=dev)
    offs = num_cells + np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    me = dist.get_rank(group)
    full[int(begins[me]):int(ends[me])] = grid_local[:cells[1] - cells[0]]
    full[int(offs[me]):int(offs[me]) + int(body_words)] = body_local[:int(body_words)]
    work = []
    for r in range(world):
        src = dist.get_global_rank(group, r) if group is not None else r
        if ends[r] > begins[r]:
            work.append(dist.broadcast(full[int(begins[r]):int(ends[r])], src=src, group=group, async_op=True))
        if sizes[r] > 0:
            work.append(dist.broadcast(full[int(offs[r]):int(offs[r]) + int(sizes[r])], src=src, group=group, async_op=True))
    for w in work:
        w.wait()
    if stats is not None:
        stats["ranks_seen"] = int(len(meta)); stats["bytes_all_gathered"] = int(4 * total)
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
    offset, _ = body_offset_for_rank(info.body_words, num_cells, group, _collective_device(dev, group))
    ncell = ranges[rank][1] - ranges[rank][0]
    grid_local = torch.empty(ncell, dtype=torch.int32, device=dev)
    body_local = torch.empty(max(int(info.body_words), 1), dtype=torch.int32, device=dev)
    shard.emit(offset, grid_local, body_local)
    cdev = _collective_device(dev, group)
    xstats = {}
    full = exchange_and_assemble(grid_local.to(cdev), body_local.to(cdev), info.body_words, ranges[rank], num_cells, group, xstats).to(dev)
    stats = torch.tensor([info.value_range, -info.min_border_value], dtype=torch.float32, device=cdev)
    dist.all_reduce(stats, op=dist.ReduceOp.MAX, group=group)
    lpd = torch.tensor(list(info.leaves_per_depth), dtype=torch.int64, device=cdev)
    cnt = torch.tensor([info.num_leaves, info.num_nodes, info.num_samples, info.num_traversals, info.num_nearest_fallbacks], dtype=torch.int64, device=cdev)
    dist.all_reduce(lpd, group=group); dist.all_reduce(cnt, group=group)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    tree = api.OctreeSdf.from_data(mesh.ctx, full, info.box_min, info.box_max, info.start_grid_size, info.max_depth,
                                   float(stats[0].item()), float(-stats[1].item()), where=api.DEVICE, cell_size=info.start_grid_cell_size)
    tree._override = {"leaves_per_depth": [int(x) for x in lpd.cpu().tolist()], "num_leaves": int(cnt[0]), "num_nodes": int(cnt[1]), "num_traversals": int(cnt[3]),
                      "num_samples": int(cnt[2]), "num_nearest_fallbacks": int(cnt[4])}
    shard.close()
    return tree, {"shard_build_s": t1 - t0, "exchange_s": t2 - t1, "exchange_bytes": xstats.get("bytes_all_gathered", 0), "ranks_seen": xstats.get("ranks_seen", 0),
                  "backend": dist.get_backend(group)}


# ---------------------------------------------------------------------------------------------------------------------
# The sphere BVH.  Built on the device (the default since round 3: 7 ms at 327 680 triangles, 16 ms at 1.31 M) every rank builds its
# own — the builder reproduces the reference's tree bit for bit, so the ranks agree without talking, and a broadcast of the two
# arrays (80 B per triangle: 105 MB at 1.31 M) would only add a serial step.  The HOST planner (SDFHIP_BVH_BUILD=host) wants all
# the cores of the node: planned by every rank at the same time it takes several times longer than planned once, so there rank `src`
# plans, the arrays are broadcast and the others import them.

def bvh_built_on_device():
    return os.environ.get("SDFHIP_BVH_BUILD") != "host"


def share_bvh(mesh, rank, world, dev, group=None, src=0):
    """Collective under SDFHIP_BVH_BUILD=host (plan on `src`, broadcast, import); a local device build otherwise.  Returns the seconds
    this rank spent."""
    if bvh_built_on_device():
        t0 = time.perf_counter()
        mesh.build_bvh()
        return time.perf_counter() - t0
    t0 = time.perf_counter()
    n = max(len(mesh.indices) - 1, 1)
    cdev = _collective_device(dev, group)
    sph = torch.empty(8 * n, dtype=torch.float64, device=cdev); kids = torch.empty(2 * n, dtype=torch.int32, device=cdev)
    if rank == src:
        mesh.build_bvh()
        if cdev.type == "cpu":
            a, b = mesh.bvh_arrays(); sph.copy_(torch.from_numpy(a)); kids.copy_(torch.from_numpy(b))
        else:
            mesh.bvh_arrays(sph, kids)
    dist.broadcast(sph, src, group=group); dist.broadcast(kids, src, group=group)
    if rank != src:
        if cdev.type == "cpu":
            mesh.set_bvh(sph.numpy(), kids.numpy())
        else:
            mesh.set_bvh(sph, kids)
    return time.perf_counter() - t0


# ---------------------------------------------------------------------------------------------------------------------
# CONTINUITY build: the tree is not separable by start cell (the second iteration couples neighbouring cells,
# src/sdf/OctreeSdfBreadthFirstNoDelay.h:440-482), so every rank builds the whole tree and only the nearest-triangle
# traversals of each sample batch are shared out (include/sdfhip.h, sdfhip_exchange): one all-reduce of 4 B per unique
# sample point and batch (a-priori levels, then one per level).

class SampleExchange:
    """torch.distributed side of sdfhip_exchange.  Use as a context manager around collective CONTINUITY builds."""

    def __init__(self, ctx, rank, world, dev, group=None):
        self.ctx, self.rank, self.world, self.dev, self.group = ctx, int(rank), int(world), dev, group
        self.buf = None
        self.error = None
        self.bytes_reduced = 0
        self.seconds = 0.0
        self._acq = _lib.ACQUIRE_FN(self._acquire)
        self._red = _lib.ALL_REDUCE_FN(self._all_reduce)
        self._x = _lib.Exchange(None, self._acq, self._red, self.rank, self.world)

    def _sync(self):
        if torch.device(self.dev).type == "cuda":
            torch.cuda.synchronize(self.dev)

    def _acquire(self, _user, count):
        try:
            n = max(int(count), 1)
            if self.buf is None or self.buf.numel() < n:
                self.buf = torch.empty(int(n * 1.25) + 1024, dtype=torch.int32, device=self.dev)
            self.buf[:n].zero_()
            self._sync()      # the library writes from its own stream
            return self.buf.data_ptr()
        except Exception as e:      # never let an exception cross the C boundary
            self.error = e
            return None

    def _all_reduce(self, _user, count):
        try:
            t0 = time.perf_counter()
            n = int(count)
            view = self.buf[:n]
            if _collective_device(self.dev, self.group).type == "cpu":
                h = view.cpu()
                dist.all_reduce(h, group=self.group)
                view.copy_(h)
            else:
                dist.all_reduce(view, group=self.group)
            self._sync()
            self.bytes_reduced += 4 * n
            self.seconds += time.perf_counter() - t0
            return 0
        except Exception as e:
            self.error = e
            return 1

    def __enter__(self):
        _lib.check(_lib.lib().sdfhip_ctx_set_exchange(self.ctx.h, C.byref(self._x)))
        return self

    def __exit__(self, *exc):
        _lib.lib().sdfhip_ctx_set_exchange(self.ctx.h, None)
        return False


def build_continuity_sharded(mesh, box, depth, start_depth, max_error, rank, world, dev, group=None, **kw):
    """CONTINUITY OctreeSdf on `world` GPUs: identical tree on every rank, traversals shared.  Returns (tree, timing dict)."""
    from . import api
    t0 = time.perf_counter()
    with SampleExchange(mesh.ctx, rank, world, dev, group) as x:
        try:
            tree = api.OctreeSdf(mesh, box, depth, start_depth, max_error, init_algorithm=api.ALG_CONTINUITY, **kw)
        except Exception as e:
            raise (x.error or e)
    return tree, {"build_s": time.perf_counter() - t0, "exchange_s": x.seconds, "exchange_bytes": x.bytes_reduced}


# ---------------------------------------------------------------------------------------------------------------------
# ExactOctreeSdf: same decomposition (include/SdfLib/ExactOctreeSdfDepthFirst.h:534-622), three arrays to concatenate.
# Cells are partitioned in the reference's EMISSION order (children 7..0 at every level), so that the concatenation of the
# ranks' bodies / sets / masks is exactly the single-thread build's layout; the start-grid slots are scattered by cell id.

def exact_emission_rank(start_depth):
    """rank[cell] for z-major cell ids (x fastest): position of the cell's subtree in the single-thread array."""
    G = 1 << start_depth
    z, y, x = np.meshgrid(np.arange(G), np.arange(G), np.arange(G), indexing="ij")
    r = np.zeros_like(x)
    for level in range(start_depth):
        sh = start_depth - 1 - level
        c = ((x >> sh) & 1) | (((y >> sh) & 1) << 1) | (((z >> sh) & 1) << 2)
        r = r * 8 + (7 - c)
    return r.reshape(-1)


def assemble_exact(parts, num_cells):
    """parts: per rank dict(cells, grid_nodes, grid_has, body_nodes, body_has, sets, masks) in rank order (numpy or torch, all the
    same kind).  Returns (nodes[(num_cells + sum bodies), 2], has, sets, masks)."""
    cat = np.concatenate if isinstance(parts[0]["sets"], np.ndarray) else __import__("torch").cat
    first = parts[0]
    if isinstance(first["sets"], np.ndarray):
        grid = np.zeros((num_cells, 2), dtype=first["grid_nodes"].dtype); ghas = np.zeros(num_cells, dtype=np.uint8)
        idx = lambda c: np.asarray(c, dtype=np.int64)
    else:
        import torch
        grid = torch.zeros((num_cells, 2), dtype=first["grid_nodes"].dtype, device=first["grid_nodes"].device)
        ghas = torch.zeros(num_cells, dtype=torch.uint8, device=grid.device)
        idx = lambda c: torch.as_tensor(np.asarray(c, dtype=np.int64), device=grid.device)
    for p in parts:
        grid[idx(p["cells"])] = p["grid_nodes"]; ghas[idx(p["cells"])] = p["grid_has"]
    nodes = cat([grid] + [p["body_nodes"] for p in parts]); has = cat([ghas] + [p["body_has"] for p in parts])
    return nodes, has, cat([p["sets"] for p in parts]), cat([p["masks"] for p in parts])


def exact_offsets(sizes, num_cells):
    """sizes[r] = (body_nodes, set_words, mask_bytes) of every rank -> [(node_offset, set_offset, mask_offset)] per rank."""
    out, n, s, m = [], num_cells, 0, 0
    for bn, sw, mb in sizes:
        out.append((n, s, m)); n += int(bn); s += int(sw); m += int(mb)
    return out


def _all_gather_padded(t, length, group):
    """all-gather of 1-D / 2-D tensors whose first dimension differs per rank (lengths[r] known on every rank)."""
    world = dist.get_world_size(group)
    mx = max(int(max(length)), 1)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device); pad[:t.shape[0]] = t
    out = torch.empty((world * mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return [out[r * mx: r * mx + int(length[r])] for r in range(world)]


def build_exact_sharded(mesh, box, max_depth, start_depth, min_triangles_per_node, rank, world, dev, group=None):
    """Sharded GPU build + one exchange.  Returns (ExactOctreeSdf holding the full arrays on this GPU, timing dict)."""
    from . import api
    num_cells = 8 ** start_depth
    order = np.argsort(exact_emission_rank(start_depth), kind="stable")          # cell id at every emission rank
    ranges = partition_cells(num_cells, world, cell_weights(mesh.vertices, box, start_depth)[order])
    t0 = time.perf_counter()
    shard = api.ExactShard(mesh, box, max_depth, start_depth, min_triangles_per_node, ranges[rank])
    info = shard.info
    t1 = time.perf_counter()
    cdev = _collective_device(dev, group)
    mine = torch.tensor([info.num_nodes, info.num_set_words, info.num_mask_bytes, info.max_triangles_in_leafs, info.max_triangles_encoded_in_leafs,
                         info.cull_tests], dtype=torch.int64, device=cdev)
    meta = [torch.zeros(6, dtype=torch.int64, device=cdev) for _ in range(world)]
    dist.all_gather(meta, mine, group=group)
    meta = torch.stack(meta).cpu().numpy()
    offs = exact_offsets(meta[:, :3], num_cells)
    part = {k: v.to(cdev) for k, v in shard.emit(*offs[rank], device=dev).items()}
    ncell = [b - a for a, b in ranges]
    gathered = {"grid_nodes": _all_gather_padded(part["grid_nodes"], ncell, group), "grid_has": _all_gather_padded(part["grid_has"], ncell, group),
                "body_nodes": _all_gather_padded(part["body_nodes"], meta[:, 0], group), "body_has": _all_gather_padded(part["body_has"], meta[:, 0], group),
                "sets": _all_gather_padded(part["sets"], meta[:, 1], group), "masks": _all_gather_padded(part["masks"], meta[:, 2], group)}
    parts = []
    for r in range(world):
        cells = np.sort(order[ranges[r][0]:ranges[r][1]])                        # shard_cells() order: ascending z-major
        parts.append(dict(cells=cells, **{k: v[r] for k, v in gathered.items()}))
    nodes, has, sets, masks = (t.to(dev) for t in assemble_exact(parts, num_cells))
    full = api.ExactInfo.from_buffer_copy(info)
    full.num_nodes, full.num_set_words, full.num_mask_bytes = int(nodes.shape[0]), int(sets.shape[0]), int(masks.shape[0])
    full.max_triangles_in_leafs, full.max_triangles_encoded_in_leafs = int(meta[:, 3].max()), int(meta[:, 4].max())
    full.cull_tests = int(meta[:, 5].sum())          # whole job (levels above the start depth are tested by every rank)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    tree = api.ExactOctreeSdf.from_parts(mesh, full, nodes.contiguous(), has.contiguous(), sets.contiguous() if len(sets) else torch.zeros(1, dtype=torch.int32, device=dev),
                                         masks.contiguous() if len(masks) else torch.zeros(1, dtype=torch.uint8, device=dev), where=api.DEVICE)
    shard.close()
    return tree, {"shard_build_s": t1 - t0, "exchange_s": t2 - t1}


# ---------------------------------------------------------------------------------------------------------------------
# Queries: the tree is replicated, the query array is split by contiguous ranges (first row).  A tree that only
# one rank holds (built there, or loaded from a .bin file) reaches the others with ONE broadcast of its node array.

def broadcast_octree(tree, ctx, dev, src=0, group=None):
    """Every rank returns an OctreeSdf holding rank `src`'s node array (rank `src` passes its tree, the others pass None)."""
    from . import api
    rank = dist.get_rank(group)
    cdev = _collective_device(dev, group)
    head = torch.zeros(12, dtype=torch.float64, device=cdev)
    if rank == src:
        i = tree.info
        head = torch.tensor(list(i.box_min) + list(i.box_max) + [i.start_grid_size, i.max_depth, i.value_range, i.min_border_value, i.num_words, i.start_grid_cell_size],
                            dtype=torch.float64, device=cdev)
    dist.broadcast(head, src, group=group)
    h = head.cpu().tolist()
    n = int(h[10])
    if rank == src:
        words = torch.from_numpy(tree.get_octree_data().view(np.int32)).to(cdev)
    else:
        words = torch.empty(n, dtype=torch.int32, device=cdev)
    dist.broadcast(words, src, group=group)
    if rank == src:
        return tree
    return api.OctreeSdf.from_data(ctx, words.to(dev), np.float32(h[0:3]), np.float32(h[3:6]), int(h[6]), int(h[7]), float(np.float32(h[8])), float(np.float32(h[9])),
                                   where=api.DEVICE, cell_size=float(np.float32(h[11])))      # answers like rank src's tree, built or loaded


def query_range(n, rank, world):
    """Contiguous share [begin, end) of n queries for `rank`."""
    return (n * rank) // world, (n * (rank + 1)) // world
