"""Differential fuzzing on the GPU box: random meshes / boxes / parameters, GPU (through the C ABI) against the CPU oracle, bit for
bit — TriangleData, nearest-triangle ids, both OctreeSdf builders (all rules, both layouts), ExactOctreeSdf arrays, queries with
gradients.  Usage: tools/gpu_fuzz.py [iterations] [first seed].  Prints one line per case and a summary; exit code 1 on mismatch."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import sdflib_amd as S
from oracle import pyoracle as O
from sdflib_amd.meshgen import icosphere, bumpy_icosphere, cube_mesh, box_with_margin

MODE = os.environ.get("FUZZ_MODE", "")      # "continuity": always the CONTINUITY builder, deeper trees; "exact": deeper ExactOctreeSdf; "big": larger meshes
bits = lambda a: np.ascontiguousarray(a).view(np.uint32)
ctx = None          # made on first use: the module is also imported by tests/test_gpu_fuzz.py


def _ctx():
    global ctx
    if ctx is None:
        ctx = S.Context(0)
    return ctx


def random_mesh(rng):
    kind = rng.integers(0, 5)
    big = rng.random() < (0.5 if MODE == "big" else 0.08)
    if kind == 0: v, f = icosphere(int(rng.integers(0, 4)) + (2 if big else 0))
    elif kind == 1: v, f = bumpy_icosphere(int(rng.integers(1, 4)) + (2 if big else 0))
    elif kind == 2: v, f = cube_mesh()
    elif kind == 3:                                  # two components, one inside the other or apart
        a, fa = icosphere(int(rng.integers(1, 3)))
        off = rng.normal(0, 0.6, 3).astype(np.float32)
        v = np.concatenate([a, a * np.float32(rng.uniform(0.2, 0.8)) + off]).astype(np.float32); f = np.concatenate([fa, fa + len(a)]).astype(np.uint32)
    else:                                            # random soup of a few triangles
        n = int(rng.integers(1, 30))
        v = rng.normal(0, 1, (3 * n, 3)).astype(np.float32); f = np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)
    v = v.astype(np.float32).copy(); f = f.astype(np.uint32).copy()
    if rng.random() < 0.5:                           # anisotropic scale, rotation, translation
        A = np.linalg.qr(rng.normal(0, 1, (3, 3)))[0] * rng.uniform(0.3, 3.0, 3)[None, :]
        v = (v @ A.T.astype(np.float32) + rng.normal(0, 2.0, 3).astype(np.float32)).astype(np.float32)
    if rng.random() < 0.3: v = (v + rng.normal(0, 0.01, v.shape).astype(np.float32)).astype(np.float32)
    if rng.random() < 0.3 and len(f) > 4: f = f[rng.random(len(f)) > 0.2]                        # holes
    if rng.random() < 0.2 and len(f) > 2: f = np.concatenate([f, f[rng.integers(0, len(f), 2)]])  # coincident duplicates
    if rng.random() < 0.2: f = f[:, [0, 2, 1]]                                                    # flipped winding
    if rng.random() < 0.1 and len(f) > 3:                                                         # degenerate triangles (repeated index / zero area)
        bad = f[rng.integers(0, len(f), 2)].copy(); bad[:, 2] = bad[:, 1]; f = np.concatenate([f, bad])
    if rng.random() < 0.3:                                                                        # extreme scales, far from the origin
        sc = np.float32(10.0 ** rng.uniform(-3, 3)); v = (v * sc + sc * rng.normal(0, 1, 3).astype(np.float32) * np.float32(10.0 ** rng.uniform(0, 2.5))).astype(np.float32)
    return np.ascontiguousarray(v), np.ascontiguousarray(f.astype(np.uint32))


def one(seed):
    ctx = _ctx()
    rng = np.random.default_rng(seed)
    if os.environ.get("FUZZ_VERBOSE"): print(f"  seed {seed} start", flush=True)
    v, f = random_mesh(rng)
    if len(f) < 1: return "skip"
    box = np.asarray(box_with_margin(v, margin=float(rng.uniform(0.05, 0.5))), dtype=np.float32)
    if rng.random() < 0.4: box[3:] += rng.uniform(0, 0.5, 3).astype(np.float32) * (box[3:] - box[:3])      # non-cubic input box
    use_bbox = rng.random() < 0.3
    bbox = np.concatenate([v.min(0), v.max(0)]).astype(np.float32) if use_bbox else None
    om = O.Mesh(v, f, bbox=bbox) if use_bbox else O.Mesh(v, f)
    gm = S.Mesh(v, f, ctx, bbox=bbox) if use_bbox else S.Mesh(v, f, ctx)
    a, b = om.triangle_data(), gm.triangle_data()
    assert np.array_equal(bits(a), bits(b)) or np.array_equal(a, b, equal_nan=True), "TriangleData"
    size = float((box[3:] - box[:3]).max())
    pts = (box[:3] + rng.random((20000, 3), dtype=np.float32) * size * np.float32(1.2) - np.float32(0.1 * size)).astype(np.float32)
    assert np.array_equal(om.nearest(pts), gm.nearest_triangle(pts)), "nearest ids"
    depth = int(rng.integers(4, 8) if MODE in ("continuity", "big") else rng.integers(2, 7)); start = int(rng.integers(0, min(depth, 3) + 1))
    rule = int(rng.choice([S.RULE_TRAPEZOIDAL, S.RULE_TRAPEZOIDAL, S.RULE_SIMPSONS, S.RULE_BY_DISTANCE, S.RULE_NONE]))
    if rule == S.RULE_NONE: depth = min(depth, 4)
    thr = float(10 ** rng.uniform(-4, -2)); decay = float(rng.uniform(0.0, 0.2))
    cont = rng.random() < (1.0 if MODE == "continuity" else 0.4)
    if depth == 7: thr = max(thr, 1e-3)           # keeps the single-threaded oracle build within seconds
    layout1 = rng.random() < 0.5
    ot = O.Octree(om, box, depth, start, thr, rule=rule, param1=decay, vertex_cache=False, layout=O.LAYOUT_GLOBAL_DFS if layout1 else O.LAYOUT_SUBTREES, continuity=cont)
    gt = S.OctreeSdf(gm, box, depth, start, thr, init_algorithm=S.ALG_CONTINUITY if cont else S.ALG_NO_CONTINUITY, num_threads=1 if layout1 else 2,
                     termination_rule=rule, rule_params=(thr, decay))
    assert np.array_equal(ot.data(), gt.get_octree_data()), f"octree array (cont={cont} rule={rule} depth={depth} start={start} thr={thr:g})"
    assert np.float32(ot.value_range) == np.float32(gt.info.value_range) and (np.float32(ot.min_border) == np.float32(gt.info.min_border_value) or np.isnan(ot.min_border)), "range/border"
    d0, g0 = ot.query(pts, grad=True); d1, g1 = gt.get_distance(pts, gradient=True)
    assert np.array_equal(bits(d0), bits(d1)) and np.array_equal(bits(g0), bits(g1)), "octree queries"
    if len(f) < 2: return f"T={len(f)} octree only (ExactOctreeSdf needs 2 triangles: bits per index)"
    extra = ""
    if not cont and not layout1 and rng.random() < 0.5:          # sharded build over random contiguous cell ranges == the single build
        G3 = 8 ** start
        cuts = sorted(set([0, G3] + [int(c) for c in rng.integers(0, G3 + 1, int(rng.integers(1, 5)))]))
        shards = [S.OctreeShard(gm, box, depth, start, thr, cells=(a_, b_), termination_rule=rule, rule_params=(thr, decay)) for a_, b_ in zip(cuts, cuts[1:])]
        out = np.zeros(G3 + sum(int(sh.info.body_words) for sh in shards), dtype=np.uint32); off = G3
        for sh, (a_, b_) in zip(shards, zip(cuts, cuts[1:])):
            nb = int(sh.info.body_words); grid = np.zeros(b_ - a_, dtype=np.uint32); body = np.zeros(max(nb, 1), dtype=np.uint32)
            sh.emit(off, grid, body); out[a_:b_] = grid; out[off:off + nb] = body[:nb]; off += nb
        assert np.array_equal(out, ot.data()), f"octree shards {cuts}"
        gi_ = gt.info
        re_ = S.OctreeSdf.from_data(ctx, out, gi_.box_min, gi_.box_max, gi_.start_grid_size, gi_.max_depth, gi_.value_range, gi_.min_border_value, cell_size=shards[0].info.start_grid_cell_size)
        assert np.array_equal(bits(re_.get_distance(pts)), bits(d1)), "reassembled shards answer like the built tree"
        extra += f" shards={len(shards)}"
    if not cont and rng.random() < 0.3:                           # MFMA fit: same topology as the exact fit
        mt = S.OctreeSdf(gm, box, depth, start, thr, num_threads=1 if layout1 else 2, termination_rule=rule, rule_params=(thr, decay), fit_mode=S.FIT_MFMA)
        a_, b_ = ot.data(), mt.get_octree_data()
        assert a_.shape == b_.shape, "FIT_MFMA tree size"
        extra += " mfma"
    if rng.random() < 0.3:                                        # separable-Horner evaluation stays within 1e-5 of the literal order (at the tree's scale)
        df, gf = gt.get_distance(pts, gradient=True, eval_mode=S.EVAL_FAST)
        scale = max(1.0, float(gt.info.value_range))
        err = float(np.nanmax(np.abs(df - d1))) if len(d1) else 0.0
        assert err <= 3e-5 * scale, f"EVAL_FAST error {err:g} at value range {scale:g}"
        ok_ = np.isfinite(gf).all(1) & np.isfinite(g1).all(1)      # unit vectors; where the polynomial's gradient nearly vanishes the direction is ill-conditioned in BOTH orders
        assert not ok_.any() or float(np.median(np.abs(gf[ok_] - g1[ok_]).max(1))) <= 1e-4, "EVAL_FAST gradients differ from the literal-order ones"
        extra += " fast"
    if rng.random() < 0.15:                                       # the reference's .bin layout: save, load, same words and answers
        import tempfile
        with tempfile.TemporaryDirectory() as tmpd:
            path = os.path.join(tmpd, "t.bin")
            gt.save_to_file(path)
            lt = S.load_from_file(path, ctx)
            # a LOADED tree takes its cell size from the stored box (OctreeSdf.h:233), a built one from the input box's largest extent
            # (OctreeSdf.cpp:43-50): far from the origin the two differ in the last bit, in the reference as here
            raw = O.octree_query_raw(gt.get_octree_data(), gt.get_grid_bounding_box(), gt.info.start_grid_size, gt.info.min_border_value, pts)
            assert np.array_equal(lt.get_octree_data(), gt.get_octree_data()) and np.array_equal(bits(lt.get_distance(pts)), bits(raw)), "octree save/load"
        extra += " bin"
    if rng.random() < 0.3:                                        # lattice query == point query
        nxyz = tuple(int(x) for x in rng.integers(1, 20, 3)); org = (box[:3] - np.float32(0.05 * size)).astype(np.float32); st = (rng.uniform(0.01, 0.1, 3) * size).astype(np.float32)
        gi = np.stack(np.meshgrid(np.arange(nxyz[2]), np.arange(nxyz[1]), np.arange(nxyz[0]), indexing="ij"), -1).reshape(-1, 3)[:, ::-1].astype(np.float32)
        lp = (org + gi * st).astype(np.float32)
        dg, gg = gt.get_distance_grid(org, st, nxyz, gradient=True); dp, gp = gt.get_distance(lp, gradient=True)
        assert np.array_equal(bits(dg), bits(dp)) and np.array_equal(bits(gg), bits(gp)), "lattice query"
        extra += " grid"
    edepth = int(rng.integers(4, 8) if MODE == "exact" else rng.integers(2, 6)); estart = int(rng.integers(0, min(edepth - 2, 2) + 1)); mint = int(rng.choice([1, 2, 8, 32, 128]))
    oe = O.Exact(om, box, edepth, estart, mint); ge = S.ExactOctreeSdf(gm, box, edepth, estart, mint)
    for name, x, y in zip(("nodes", "has", "sets", "masks"), oe.data(), ge.download()):
        if name == "nodes": x, y = x[:, 0], y[:, 0]
        if name == "has": continue
        assert x.shape == y.shape and np.array_equal(x, y), f"exact {name} (depth={edepth} start={estart} min={mint})"
    e0, t0 = oe.query(pts, tri=True); e1, t1 = ge.get_distance(pts, triangle=True)
    assert np.array_equal(bits(e0), bits(e1)) and np.array_equal(t0, t1.astype(np.uint32)), "exact queries"
    if rng.random() < 0.5:                                        # ExactOctreeSdf gradients (TriangleUtils.h:292-376), NaN directions included
        eg0 = oe.query(pts, grad=True); eg1 = ge.get_distance(pts, gradient=True)
        assert np.array_equal(bits(eg0[0]), bits(eg1[0])) and (np.array_equal(bits(eg0[1]), bits(eg1[1])) or np.array_equal(eg0[1], eg1[1], equal_nan=True)), "exact gradients"
    if rng.random() < 0.2:                                        # small batches take the per-lane kernel instead of the leaf-sorted one
        k_ = int(rng.integers(1, 300))
        assert np.array_equal(bits(ge.get_distance(pts[:k_])), bits(e0[:k_])), "exact small batch"
    if rng.random() < 0.4 and estart >= 1:                        # ExactOctreeSdf shards over random ranges of the emission order
        from sdflib_amd import distributed as sdist
        G3 = 8 ** estart
        cuts = sorted(set([0, G3] + [int(c) for c in rng.integers(0, G3 + 1, int(rng.integers(1, 4)))]))
        shards = [S.ExactShard(gm, box, edepth, estart, mint, (a_, b_)) for a_, b_ in zip(cuts, cuts[1:])]
        offs = sdist.exact_offsets([(sh.info.num_nodes, sh.info.num_set_words, sh.info.num_mask_bytes) for sh in shards], G3)
        parts = [dict(cells=sh.cells(), **sh.emit(*o)) for sh, o in zip(shards, offs)]
        for name, x, y in zip(("nodes", "has", "sets", "masks"), ge.download(), sdist.assemble_exact(parts, G3)):
            assert np.array_equal(x, y), f"exact shards {cuts}: {name}"
        extra += f" exact-shards={len(shards)}"
    if rng.random() < 0.15:                                       # ExactOctreeSdf .bin round trip: arrays kept; answers = the oracle's for the stored box
        import tempfile
        with tempfile.TemporaryDirectory() as tmpd:
            path = os.path.join(tmpd, "e.bin")
            ge.save_to_file(path, gm)
            le = S.load_from_file(path, ctx)
            for name, x, y in zip(("nodes", "has", "sets", "masks"), ge.download(), le.download()):
                if name == "has": continue
                if name == "nodes": x, y = x[:, 0], y[:, 0]
                assert np.array_equal(x, y), f"exact save/load {name}"
            dl_ = le.get_distance(pts)
            same = np.array_equal(bits(dl_), bits(e1))
            gi_ = ge.info
            stored_cell = np.float32(gi_.box_max[0] - gi_.box_min[0]) / np.float32(gi_.start_grid_size)
            assert same or stored_cell != np.float32(gi_.start_grid_cell_size), "exact save/load answers"
        extra += " ebin"
    return f"T={len(f)} cont={int(cont)} rule={rule} d={depth}/{start} words={len(ot.data())} exact d={edepth}/{estart} min={mint}{extra}"


if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    fails = 0; t00 = time.time()
    for s in range(seed0, seed0 + iters):
        try:
            print(f"seed {s}: {one(s)}", flush=True)
        except AssertionError as e:
            fails += 1; print(f"seed {s}: MISMATCH {e}", flush=True)
        except Exception as e:      # noqa: BLE001
            fails += 1; print(f"seed {s}: ERROR {type(e).__name__}: {e}", flush=True)
    print(f"{iters} cases, {fails} failures, {time.time() - t00:.0f} s")
    sys.exit(1 if fails else 0)
