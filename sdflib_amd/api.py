"""Host-side mirror of the reference's interface for the hot path, on top of the libsdfhip C ABI.

Names and argument meaning follow the reference (include/SdfLib/OctreeSdf.h:156-219,
include/SdfLib/ExactOctreeSdf.h:91-134, include/SdfLib/SdfFunction.h:12-58); the batched entry points
(`get_distance` over arrays / torch tensors / a lattice) are the data-parallel form of `getDistance`.
"""
import ctypes as C
import numpy as np

from ._lib import lib, check, OctreeInfo, OctreeParams, ExactInfo, SdfHipError  # noqa: F401

HOST, DEVICE = 0, 1
RULE_NONE, RULE_TRAPEZOIDAL, RULE_SIMPSONS, RULE_BY_DISTANCE = 0, 1, 2, 3
ALG_UNIFORM, ALG_NO_CONTINUITY, ALG_CONTINUITY = 0, 1, 2
LAYOUT_GLOBAL_DFS, LAYOUT_SUBTREES = 0, 1
EVAL_EXACT, EVAL_FAST = 0, 1
FIT_EXACT, FIT_MFMA = 0, 1

_TERMINATION_RULES = {"none": RULE_NONE, "trapezoidal_rule": RULE_TRAPEZOIDAL, "simpsons_rule": RULE_SIMPSONS,
                      "by_distance_rule": RULE_BY_DISTANCE}


def string_to_termination_rule(text):
    """OctreeSdf::stringToTerminationRule (include/SdfLib/OctreeSdf.h:128-148); None when unknown."""
    return _TERMINATION_RULES.get(text.lower())


def _np(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class Context:
    """One (process, device) context.  With ``use_torch_stream`` the engine runs on torch's current stream."""

    def __init__(self, device=0, stream=None, use_torch_stream=False):
        borrowed = stream is not None
        if use_torch_stream and stream is None:
            import torch
            stream = torch.cuda.current_stream(device).cuda_stream     # 0 = the default (null) stream
            borrowed = True
        h = C.c_void_p()
        check(lib().sdfhip_ctx_create(int(device), C.c_void_p(stream) if stream else None, 1 if borrowed else 0, C.byref(h)))
        self.h, self.device = h, int(device)
        self.stream = int(stream) if (borrowed and stream) else (0 if borrowed else None)      # None: a private stream of the library

    # torch tensors in / out: the engine is asynchronous on ITS stream.  When that is torch's current stream nothing is needed;
    # otherwise the call is fenced on both sides so that results are never read (or inputs written) across streams.
    def _shares_torch_stream(self):
        import torch
        return self.stream is not None and self.stream == int(torch.cuda.current_stream(self.device).cuda_stream)

    def _torch_inputs_ready(self):
        if not self._shares_torch_stream():
            import torch
            torch.cuda.current_stream(self.device).synchronize()

    def _torch_outputs_ready(self):
        if not self._shares_torch_stream():
            self.synchronize()

    def synchronize(self):
        check(lib().sdfhip_ctx_synchronize(self.h))

    def trim(self, keep_bytes=0):
        """Free the device memory the context keeps for reuse beyond keep_bytes (sdfhip_ctx_trim)."""
        check(lib().sdfhip_ctx_trim(self.h, int(keep_bytes)))

    def cached_bytes(self):
        n = C.c_uint64(0)
        check(lib().sdfhip_ctx_cached_bytes(self.h, C.byref(n)))
        return int(n.value)

    def close(self):
        if getattr(self, "h", None):
            lib().sdfhip_ctx_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


class Mesh:
    """sdflib::Mesh(vertices, indices) + the per-mesh acceleration data (TriangleData, sphere BVH) on the device."""

    def __init__(self, vertices, indices, ctx=None, bbox=None, plan_bvh_early=False):
        """bbox (6 floats, optional) = the box the reference's file loader computes; it switches on the non-manifold seam
        welding (TriangleUtils.cpp:292-420).  None = the raw-pointer constructor's behaviour (no welding).
        plan_bvh_early: the sphere BVH is planned on host threads while the device prepares the TriangleData (for a mesh an OctreeSdf
        will be built from; SDFHIP_MESH_PLAN_BVH_EARLY)."""
        self.ctx = ctx or default_context()
        self.vertices = _np(vertices, np.float32).reshape(-1, 3)
        self.indices = _np(indices, np.uint32).reshape(-1, 3)
        self.bbox = None if bbox is None else _np(bbox, np.float32).reshape(6)
        h = C.c_void_p()
        check(lib().sdfhip_mesh_create_opt(self.ctx.h, _ptr(self.vertices), len(self.vertices), _ptr(self.indices), len(self.indices),
                                           None if self.bbox is None else _ptr(self.bbox), 1 if plan_bvh_early else 0, C.byref(h)))
        self.h = h

    @classmethod
    def from_file(cls, path, ctx=None):
        """sdflib::Mesh(path) (src/utils/Mesh.cpp:44-62): load + compute the bounding box (=> seam welding enabled)."""
        from . import meshio
        v, f = meshio.read_mesh(path)
        return cls(v, f, ctx=ctx, bbox=np.concatenate([v.min(axis=0), v.max(axis=0)]))

    def edge_stats(self):
        a, b = C.c_uint32(), C.c_uint32()
        check(lib().sdfhip_mesh_edge_stats(self.h, C.byref(a), C.byref(b)))
        return {"unmatched_edges": a.value, "welded_half_edges": b.value}

    def close(self):
        if getattr(self, "h", None):
            lib().sdfhip_mesh_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def triangle_data(self):
        out = np.empty((len(self.indices), 37), dtype=np.float32)
        check(lib().sdfhip_mesh_triangle_data(self.h, _ptr(out)))
        return out

    def build_bvh(self):
        s = C.c_double()
        check(lib().sdfhip_mesh_build_bvh(self.h, C.byref(s)))
        return s.value

    def bvh_arrays(self, spheres=None, children=None):
        """The planned BVH: (float64[8 (T-1)], int32[2 (T-1)]) as numpy arrays, or written into the given torch CUDA tensors."""
        n = max(len(self.indices) - 1, 1)
        if spheres is not None and _is_torch(spheres):
            self.ctx._torch_inputs_ready()
            check(lib().sdfhip_mesh_bvh_export(self.h, C.c_void_p(spheres.data_ptr()), C.c_void_p(children.data_ptr()), DEVICE))
            self.ctx._torch_outputs_ready()
            return spheres, children
        sph = np.empty(8 * n, dtype=np.float64); kids = np.empty(2 * n, dtype=np.int32)
        check(lib().sdfhip_mesh_bvh_export(self.h, _ptr(sph), _ptr(kids), HOST))
        return sph, kids

    def set_bvh(self, spheres, children):
        """Install a BVH planned elsewhere (another rank) instead of planning it here."""
        if _is_torch(spheres):
            self.ctx._torch_inputs_ready()
            check(lib().sdfhip_mesh_bvh_import(self.h, C.c_void_p(spheres.data_ptr()), C.c_void_p(children.data_ptr()), DEVICE))
            self.ctx._torch_outputs_ready()
        else:
            sph = _np(spheres, np.float64); kids = _np(children, np.int32)
            check(lib().sdfhip_mesh_bvh_import(self.h, _ptr(sph), _ptr(kids), HOST))

    def nearest_triangle(self, points):
        pts = _np(points, np.float32).reshape(-1, 3)
        out = np.empty(len(pts), dtype=np.uint32)
        check(lib().sdfhip_mesh_nearest(self.h, _ptr(pts), len(pts), _ptr(out), HOST))
        return out

    def point_values(self, points, triangle_ids):
        pts = _np(points, np.float32).reshape(-1, 3); t = _np(triangle_ids, np.uint32)
        out = np.empty((len(pts), 8), dtype=np.float32)
        check(lib().sdfhip_mesh_point_values(self.h, _ptr(pts), _ptr(t), len(pts), _ptr(out), HOST))
        return out


def _params(box, depth, start_depth, rule, rule_params, algorithm, layout, fit_mode, cells):
    p = OctreeParams()
    box = _np(box, np.float32).reshape(6)
    for i in range(3):
        p.box_min[i] = box[i]; p.box_max[i] = box[3 + i]
    p.depth, p.start_depth, p.rule = depth, start_depth, rule
    p.rule_params[0] = rule_params[0]; p.rule_params[1] = rule_params[1] if len(rule_params) > 1 else 0.0
    p.algorithm, p.layout, p.fit_mode = algorithm, layout, fit_mode
    p.cell_begin, p.cell_end = cells
    return p


class OctreeSdf:
    """sdflib::OctreeSdf built on the GPU.

    OctreeSdf(mesh, box, depth, start_depth, max_error=1e-3, init_algorithm=NO_CONTINUITY, num_threads=1)
    mirrors the reference constructor (include/SdfLib/OctreeSdf.h:156-160); ``num_threads`` selects which of the
    reference's two array layouts is produced (1 -> numThreads<2 array, >=2 -> per-start-cell sub-octrees).
    """

    def __init__(self, mesh=None, box=None, depth=None, start_depth=None, max_error=1e-3, init_algorithm=ALG_NO_CONTINUITY,
                 num_threads=1, termination_rule=RULE_TRAPEZOIDAL, rule_params=None, fit_mode=FIT_EXACT, _handle=None, _ctx=None):
        if _handle is not None:
            self.h, self.ctx = _handle, _ctx
        else:
            self.ctx = mesh.ctx
            rp = rule_params if rule_params is not None else (max_error, 0.0)
            layout = LAYOUT_GLOBAL_DFS if num_threads < 2 else LAYOUT_SUBTREES
            p = _params(box, depth, start_depth, termination_rule, rp, init_algorithm, layout, fit_mode, (0, 0))
            h = C.c_void_p()
            check(lib().sdfhip_octree_build(self.ctx.h, mesh.h, C.byref(p), C.byref(h)))
            self.h = h
        self._info = None

    @classmethod
    def from_data(cls, ctx, words, box_min, box_max, start_grid_size, max_depth, value_range, min_border_value, where=HOST, cell_size=None):
        """Wrap a node array.  Without ``cell_size`` the tree behaves like a LOADED one (cell size from the box, OctreeSdf.h:233);
        reassembled shards of a build pass the build's (info.start_grid_cell_size), see sdfhip.h."""
        h = C.c_void_p()
        bmin, bmax = _np(box_min, np.float32), _np(box_max, np.float32)
        if where == HOST:
            words = _np(words, np.uint32); ptr, n = _ptr(words), len(words)
        else:
            ptr, n = C.c_void_p(words.data_ptr()), words.numel()
            ctx._torch_inputs_ready()
        check(lib().sdfhip_octree_from_data(ctx.h, ptr, n, where, _ptr(bmin), _ptr(bmax), int(start_grid_size), int(max_depth),
                                            float(value_range), float(min_border_value), C.byref(h)))
        if where != HOST: ctx._torch_outputs_ready()       # the words are copied on the engine's stream
        if cell_size is not None:
            check(lib().sdfhip_octree_set_start_grid_cell_size(h, float(cell_size)))
        return cls(_handle=h, _ctx=ctx)

    def close(self):
        if getattr(self, "h", None):
            lib().sdfhip_octree_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def info(self):
        i = OctreeInfo()
        check(lib().sdfhip_octree_get_info(self.h, C.byref(i)))
        ov = getattr(self, "_override", None)
        if ov:
            for d, n in enumerate(ov["leaves_per_depth"]):
                i.leaves_per_depth[d] = n
            i.num_leaves, i.num_nodes, i.num_samples = ov["num_leaves"], ov["num_nodes"], ov["num_samples"]
            i.num_traversals = ov.get("num_traversals", 0); i.num_nearest_fallbacks = ov.get("num_nearest_fallbacks", 0)
        return i

    # reference getters
    def get_octree_value_range(self): return self.info.value_range
    def get_octree_min_border_value(self): return self.info.min_border_value
    def get_start_grid_size(self): return (self.info.start_grid_size,) * 3
    def get_grid_bounding_box(self): i = self.info; return np.array(list(i.box_min) + list(i.box_max), dtype=np.float32)
    get_sample_area = get_grid_bounding_box
    def get_octree_max_depth(self): return self.info.max_depth

    def save_to_file(self, path):
        """SdfFunction::saveToFile (cereal PortableBinary layout, see sdflib_amd/serialization.py)."""
        from . import serialization
        i = self.info
        serialization.save_octree(path, self.get_grid_bounding_box(), i.start_grid_size, i.max_depth, i.value_range, i.min_border_value, self.get_octree_data())
        return True

    def compact(self):
        """Keep only the packed query layout on the device (sdfhip_octree_compact); get_octree_data() rebuilds the array when asked."""
        check(lib().sdfhip_octree_compact(self.h))

    def device_bytes(self):
        n = C.c_uint64(0)
        check(lib().sdfhip_octree_device_bytes(self.h, C.byref(n)))
        return int(n.value)

    def get_octree_data(self):
        """getOctreeData(): the flat u32 node array (host copy)."""
        out = np.empty(self.info.num_words, dtype=np.uint32)
        check(lib().sdfhip_octree_download(self.h, _ptr(out), HOST))
        return out

    def get_distance(self, points, gradient=False, eval_mode=EVAL_EXACT, out=None, out_grad=None):
        """Batched getDistance.  numpy in -> numpy out; torch CUDA tensor in -> torch tensors out (async on the ctx stream)."""
        if _is_torch(points):
            import torch
            pts = points.contiguous()
            assert pts.is_cuda and pts.dtype == torch.float32 and pts.shape[-1] == 3
            n = pts.numel() // 3
            d = out if out is not None else torch.empty(n, dtype=torch.float32, device=pts.device)
            g = (out_grad if out_grad is not None else torch.empty((n, 3), dtype=torch.float32, device=pts.device)) if gradient else None
            self.ctx._torch_inputs_ready()
            check(lib().sdfhip_octree_query(self.h, C.c_void_p(pts.data_ptr()), n, C.c_void_p(d.data_ptr()),
                                            C.c_void_p(g.data_ptr()) if gradient else None, DEVICE, eval_mode))
            self.ctx._torch_outputs_ready()
            return (d, g) if gradient else d
        pts = _np(points, np.float32).reshape(-1, 3)
        d = out if out is not None else np.empty(len(pts), dtype=np.float32)
        g = (out_grad if out_grad is not None else np.zeros((len(pts), 3), dtype=np.float32)) if gradient else None
        assert d.dtype == np.float32 and d.flags.c_contiguous and d.size == len(pts)
        check(lib().sdfhip_octree_query(self.h, _ptr(pts), len(pts), _ptr(d), _ptr(g), HOST, eval_mode))
        return (d, g) if gradient else d

    def get_distance_grid(self, origin, step, shape, gradient=False, eval_mode=EVAL_EXACT, device_out=False):
        """Lattice origin + (i,j,k)*step, x fastest; shape = (nx, ny, nz)."""
        o, s = _np(origin, np.float32), _np(step, np.float32)
        nx, ny, nz = (int(v) for v in shape)
        n = nx * ny * nz
        if device_out:
            import torch
            dev = torch.device("cuda", self.ctx.device)
            d = torch.empty(n, dtype=torch.float32, device=dev)
            g = torch.empty((n, 3), dtype=torch.float32, device=dev) if gradient else None
            self.ctx._torch_inputs_ready()
            check(lib().sdfhip_octree_query_grid(self.h, _ptr(o), _ptr(s), nx, ny, nz, C.c_void_p(d.data_ptr()),
                                                 C.c_void_p(g.data_ptr()) if gradient else None, DEVICE, eval_mode))
            self.ctx._torch_outputs_ready()
            return (d, g) if gradient else d
        d = np.empty(n, dtype=np.float32)
        g = np.zeros((n, 3), dtype=np.float32) if gradient else None
        check(lib().sdfhip_octree_query_grid(self.h, _ptr(o), _ptr(s), nx, ny, nz, _ptr(d), _ptr(g), HOST, eval_mode))
        return (d, g) if gradient else d


def load_from_file(path, ctx=None):
    """SdfFunction::loadFromFile: returns an OctreeSdf or an ExactOctreeSdf living on the GPU."""
    from . import serialization
    ctx = ctx or default_context()
    kind, d = serialization.load(path)
    if kind == "octree":
        return OctreeSdf.from_data(ctx, d["words"], d["box"][:3], d["box"][3:], d["start_grid_size"], d["max_depth"], d["value_range"], d["min_border_value"])
    return ExactOctreeSdf.from_data(ctx, d)


class OctreeShard:
    """One rank's part of a sharded OctreeSdf build (start-grid cells [cell_begin, cell_end))."""

    def __init__(self, mesh, box, depth, start_depth, max_error=1e-3, cells=(0, 0), termination_rule=RULE_TRAPEZOIDAL, rule_params=None):
        self.ctx = mesh.ctx
        rp = rule_params if rule_params is not None else (max_error, 0.0)
        p = _params(box, depth, start_depth, termination_rule, rp, ALG_NO_CONTINUITY, LAYOUT_SUBTREES, FIT_EXACT, cells)
        h = C.c_void_p()
        check(lib().sdfhip_octree_build_shard(self.ctx.h, mesh.h, C.byref(p), C.byref(h)))
        self.h = h

    @property
    def info(self):
        i = OctreeInfo()
        check(lib().sdfhip_octree_get_info(self.h, C.byref(i)))
        return i

    def emit(self, body_offset, dst_grid, dst_body):
        """Write this shard's start-grid words and bodies (absolute indices) into numpy arrays or torch CUDA tensors."""
        if _is_torch(dst_grid):
            self.ctx._torch_inputs_ready()
            check(lib().sdfhip_octree_emit_shard(self.h, int(body_offset), C.c_void_p(dst_grid.data_ptr()), C.c_void_p(dst_body.data_ptr()), DEVICE))
            self.ctx._torch_outputs_ready()
        else:
            check(lib().sdfhip_octree_emit_shard(self.h, int(body_offset), _ptr(dst_grid), _ptr(dst_body), HOST))

    def close(self):
        if getattr(self, "h", None):
            lib().sdfhip_octree_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ExactShard:
    """One rank's part of a multi-GPU ExactOctreeSdf build (sdfhip_exact_build_shard / emit_shard): the start cells whose
    position in the reference's emission order lies in rank_range."""

    def __init__(self, mesh, box, max_depth, start_depth, min_triangles_per_node, rank_range):
        self.ctx, self.mesh = mesh.ctx, mesh
        box = _np(box, np.float32).reshape(6)
        bmin, bmax = box[:3].copy(), box[3:].copy()
        h = C.c_void_p()
        check(lib().sdfhip_exact_build_shard(self.ctx.h, mesh.h, _ptr(bmin), _ptr(bmax), int(max_depth), int(start_depth), int(min_triangles_per_node),
                                             int(rank_range[0]), int(rank_range[1]), C.byref(h)))
        self.h = h
        self.num_cells = int(rank_range[1]) - int(rank_range[0])

    @property
    def info(self):
        i = ExactInfo()
        check(lib().sdfhip_exact_get_info(self.h, C.byref(i)))
        return i

    def cells(self):
        out = np.zeros(self.num_cells, dtype=np.uint32)
        check(lib().sdfhip_exact_shard_cells(self.h, _ptr(out)))
        return out

    def emit(self, node_offset, set_offset, mask_offset, device=None):
        """Returns dict(grid_nodes[ncells,2], grid_has, body_nodes[n,2], body_has, sets, masks): numpy arrays, or torch tensors on
        `device` (int32 / uint8) for the RCCL exchange."""
        i = self.info
        nb, ns, nm, nc = int(i.num_nodes), int(i.num_set_words), int(i.num_mask_bytes), self.num_cells
        if device is None:
            mk32 = lambda *shape: np.zeros(shape, dtype=np.uint32); mk8 = lambda n: np.zeros(n, dtype=np.uint8)
            ptr, where = _ptr, HOST
        else:
            import torch
            mk32 = lambda *shape: torch.zeros(shape, dtype=torch.int32, device=device); mk8 = lambda n: torch.zeros(n, dtype=torch.uint8, device=device)
            ptr, where = (lambda t: C.c_void_p(t.data_ptr())), DEVICE
        out = dict(grid_nodes=mk32(nc, 2), grid_has=mk8(nc), body_nodes=mk32(max(nb, 1), 2), body_has=mk8(max(nb, 1)), sets=mk32(max(ns, 1)), masks=mk8(max(nm, 1)))
        if where == DEVICE: self.ctx._torch_inputs_ready()
        check(lib().sdfhip_exact_emit_shard(self.h, int(node_offset), int(set_offset), int(mask_offset), ptr(out["grid_nodes"]), ptr(out["grid_has"]),
                                            ptr(out["body_nodes"]), ptr(out["body_has"]), ptr(out["sets"]), ptr(out["masks"]), where))
        if where == DEVICE: self.ctx._torch_outputs_ready()
        out["body_nodes"] = out["body_nodes"][:nb]; out["body_has"] = out["body_has"][:nb]; out["sets"] = out["sets"][:ns]; out["masks"] = out["masks"][:nm]
        return out

    def close(self):
        if getattr(self, "h", None):
            lib().sdfhip_exact_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ExactOctreeSdf:
    """sdflib::ExactOctreeSdf(mesh, box, maxDepth, startDepth=1, minTrianglesPerNode=128) on the GPU."""

    def __init__(self, mesh=None, box=None, max_depth=None, start_depth=1, min_triangles_per_node=128, num_threads=1, _handle=None, _ctx=None):
        if _handle is not None:
            self.h, self.ctx = _handle, _ctx
            return
        self.ctx = mesh.ctx
        self._mesh = mesh           # the tree queries the mesh's TriangleData (device memory owned by the Mesh): keep it alive
        box = _np(box, np.float32).reshape(6)
        bmin, bmax = box[:3].copy(), box[3:].copy()
        h = C.c_void_p()
        check(lib().sdfhip_exact_build(self.ctx.h, mesh.h, _ptr(bmin), _ptr(bmax), int(max_depth), int(start_depth),
                                       int(min_triangles_per_node), C.byref(h)))
        self.h = h

    @property
    def info(self):
        i = ExactInfo()
        check(lib().sdfhip_exact_get_info(self.h, C.byref(i)))
        return i

    @classmethod
    def from_data(cls, ctx, d):
        i = ExactInfo()
        for k in range(3):
            i.box_min[k] = d["box"][k]; i.box_max[k] = d["box"][3 + k]
        for key in ("start_grid_size", "start_depth", "max_depth", "bit_encoding_start_depth", "bits_per_index", "min_triangles_in_leafs",
                    "max_triangles_in_leafs", "max_triangles_encoded_in_leafs"):
            setattr(i, key, int(d[key]))
        nodes = _np(d["nodes"], np.uint32); sets = _np(d["sets"], np.uint32); masks = _np(d["masks"], np.uint8); td = _np(d["triangle_data"], np.float32)
        i.num_nodes, i.num_set_words, i.num_mask_bytes, i.num_triangles = len(nodes), len(sets), len(masks), len(td)
        if len(masks) == 0:
            masks = np.zeros(1, np.uint8)
        h = C.c_void_p()
        check(lib().sdfhip_exact_from_data(ctx.h, C.byref(i), _ptr(nodes), _ptr(sets), _ptr(masks), _ptr(td), C.byref(h)))
        return cls(_handle=h, _ctx=ctx)

    @classmethod
    def from_parts(cls, mesh, info, nodes, has, sets, masks, where=HOST):
        """Assembled arrays (e.g. after the sharded build's all-gather) + the mesh's TriangleData -> queryable tree.
        Arrays are numpy (where=HOST) or torch device tensors (where=DEVICE); `info` is an ExactInfo with the totals."""
        h = C.c_void_p()
        ptr = (lambda a: None if a is None else _ptr(a)) if where == HOST else (lambda a: None if a is None else C.c_void_p(a.data_ptr()))
        if where == DEVICE: mesh.ctx._torch_inputs_ready()
        check(lib().sdfhip_exact_from_parts(mesh.ctx.h, mesh.h, C.byref(info), ptr(nodes), ptr(has), ptr(sets), ptr(masks), where, C.byref(h)))
        if where == DEVICE: mesh.ctx._torch_outputs_ready()       # the arrays are copied on the engine's stream: keep them alive / unchanged until then
        t = cls(_handle=h, _ctx=mesh.ctx)
        t._mesh = mesh              # TriangleData lives in the mesh
        return t

    def save_to_file(self, path, mesh):
        """SdfFunction::saveToFile; `mesh` supplies the TriangleData block the reference stores in the file."""
        from . import serialization
        i = self.info
        nodes, has, sets, masks = self.download()
        info = {k: getattr(i, k) for k in ("start_grid_size", "start_depth", "min_triangles_in_leafs", "max_triangles_in_leafs",
                                           "max_triangles_encoded_in_leafs", "bit_encoding_start_depth", "bits_per_index", "max_depth")}
        serialization.save_exact(path, self.get_grid_bounding_box(), info, nodes, sets, masks, mesh.triangle_data())
        return True

    def get_start_grid_size(self): return (self.info.start_grid_size,) * 3
    def get_grid_bounding_box(self): i = self.info; return np.array(list(i.box_min) + list(i.box_max), dtype=np.float32)
    get_sample_area = get_grid_bounding_box
    def get_max_triangles_in_leafs(self): return self.info.max_triangles_in_leafs
    def get_min_triangles_in_leafs(self): return self.info.min_triangles_in_leafs
    def get_octree_max_depth(self): return self.info.max_depth

    def download(self):
        i = self.info
        nodes = np.zeros((i.num_nodes, 2), dtype=np.uint32); has = np.zeros(i.num_nodes, dtype=np.uint8)
        sets = np.zeros(i.num_set_words, dtype=np.uint32); masks = np.zeros(max(i.num_mask_bytes, 1), dtype=np.uint8)
        check(lib().sdfhip_exact_download(self.h, _ptr(nodes), _ptr(has), _ptr(sets), _ptr(masks)))
        return nodes, has, sets, masks[:i.num_mask_bytes]

    def get_distance(self, points, gradient=False, triangle=False, out=None):
        if _is_torch(points):
            import torch
            pts = points.contiguous(); n = pts.numel() // 3
            d = out if out is not None else torch.empty(n, dtype=torch.float32, device=pts.device)
            g = torch.empty((n, 3), dtype=torch.float32, device=pts.device) if gradient else None
            t = torch.empty(n, dtype=torch.int32, device=pts.device) if triangle else None
            self.ctx._torch_inputs_ready()
            check(lib().sdfhip_exact_query(self.h, C.c_void_p(pts.data_ptr()), n, C.c_void_p(d.data_ptr()),
                                           C.c_void_p(g.data_ptr()) if gradient else None, C.c_void_p(t.data_ptr()) if triangle else None, DEVICE))
            self.ctx._torch_outputs_ready()
        else:
            pts = _np(points, np.float32).reshape(-1, 3)
            d = np.empty(len(pts), dtype=np.float32)
            g = np.zeros((len(pts), 3), dtype=np.float32) if gradient else None
            t = np.empty(len(pts), dtype=np.uint32) if triangle else None
            check(lib().sdfhip_exact_query(self.h, _ptr(pts), len(pts), _ptr(d), _ptr(g), _ptr(t), HOST))
        res = [d]
        if gradient: res.append(g)
        if triangle: res.append(t)
        return res[0] if len(res) == 1 else tuple(res)

    def close(self):
        if getattr(self, "h", None):
            lib().sdfhip_exact_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tricubic_fit(values_8x8, node_sizes, ctx=None, fit_mode=FIT_EXACT):
    ctx = ctx or default_context()
    v = _np(values_8x8, np.float32).reshape(-1, 64); ns = _np(node_sizes, np.float32).reshape(-1)
    out = np.empty_like(v)
    check(lib().sdfhip_tricubic_fit(ctx.h, _ptr(v), _ptr(ns), len(v), _ptr(out), fit_mode))
    return out
