"""Synthetic closed manifold meshes for tests and bench.

No Bunny / Armadillo file exists in this environment, so displaced icospheres stand in:
``bumpy_icosphere(4)`` = 5 120 triangles (Bunny scale), ``(7)`` = 327 680 (Armadillo scale),
``(8)`` = 1 310 720 (the "1 M triangle" mesh).  Everything is evaluated in float32 and is deterministic.
The box follows SdfExporter: mesh bbox + 20 % of the largest extent on every side
(reference src/tools/SdfExporter/main.cpp:92-95).
"""
import numpy as np

_X = np.float32(0.525731112119133606)
_Z = np.float32(0.850650808352039932)

_ICO_V = np.array([
    [-_X, 0, _Z], [_X, 0, _Z], [-_X, 0, -_Z], [_X, 0, -_Z],
    [0, _Z, _X], [0, _Z, -_X], [0, -_Z, _X], [0, -_Z, -_X],
    [_Z, _X, 0], [-_Z, _X, 0], [_Z, -_X, 0], [-_Z, -_X, 0]], dtype=np.float32)

# outward (counter-clockwise seen from outside) winding
_ICO_F = np.array([
    [0, 1, 4], [0, 4, 9], [9, 4, 5], [4, 8, 5], [4, 1, 8],
    [8, 1, 10], [8, 10, 3], [5, 8, 3], [5, 3, 2], [2, 3, 7],
    [7, 3, 10], [7, 10, 6], [7, 6, 11], [11, 6, 0], [0, 6, 1],
    [6, 10, 1], [9, 11, 0], [9, 2, 11], [9, 5, 2], [7, 11, 2]], dtype=np.uint32)


def icosphere(subdivisions):
    """Unit icosphere with 20*4**s triangles: (float32 [V,3], uint32 [T,3]), outward winding."""
    v = _ICO_V.copy()
    f = _ICO_F.copy()
    for _ in range(subdivisions):
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0).astype(np.int64)
        e.sort(axis=1)
        key = e[:, 0] * (len(v) + 1) + e[:, 1]
        uniq, inv = np.unique(key, return_inverse=True)
        a = (uniq // (len(v) + 1)).astype(np.int64)
        b = (uniq % (len(v) + 1)).astype(np.int64)
        mid = np.float32(0.5) * (v[a] + v[b])
        mid = mid / np.sqrt((mid * mid).sum(axis=1, dtype=np.float32))[:, None].astype(np.float32)
        base = len(v)
        v = np.concatenate([v, mid.astype(np.float32)], axis=0)
        T = len(f)
        m01 = (base + inv[0:T]).astype(np.uint32)
        m12 = (base + inv[T:2 * T]).astype(np.uint32)
        m20 = (base + inv[2 * T:3 * T]).astype(np.uint32)
        f = np.concatenate([
            np.stack([f[:, 0], m01, m20], axis=1),
            np.stack([m01, f[:, 1], m12], axis=1),
            np.stack([m20, m12, f[:, 2]], axis=1),
            np.stack([m01, m12, m20], axis=1)], axis=0).astype(np.uint32)
    return np.ascontiguousarray(v, dtype=np.float32), np.ascontiguousarray(f, dtype=np.uint32)


def bumpy_icosphere(subdivisions):
    """Icosphere radially displaced by r = 1 + 0.08 sin9x sin(7y+1) sin(8z+2) + 0.02 sin(31x+y) sin29z (fp32)."""
    v, f = icosphere(subdivisions)
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    f32 = np.float32
    r = (f32(1.0) + f32(0.08) * np.sin(f32(9) * x) * np.sin(f32(7) * y + f32(1)) * np.sin(f32(8) * z + f32(2))
         + f32(0.02) * np.sin(f32(31) * x + y) * np.sin(f32(29) * z)).astype(np.float32)
    return np.ascontiguousarray(v * r[:, None], dtype=np.float32), f


def torus_knot(nu=1024, nv=160, p=2, q=3, big=1.0, small=0.45, tube=0.16, ripple=0.12):
    """Tube around a (p, q) torus knot: a closed manifold of genus 1 that is NOT star-shaped — thin, strongly curved, passing close to
    itself, with a rippled cross-section (creases of varying sharpness) — so that tree sizes and traversal statistics measured on the
    bumpy spheres can be checked against a different kind of geometry.  2 * nu * nv triangles (defaults: 327 680), outward winding.
    The frame comes from the carrier torus: n1 = the torus normal at the curve point (perpendicular to every curve on the torus),
    n2 = tangent x n1; everything is evaluated in float64 and rounded once to float32 (deterministic)."""
    u = np.arange(nu, dtype=np.float64) * (2.0 * np.pi / nu)
    cu, su, cq, sq = np.cos(p * u), np.sin(p * u), np.cos(q * u), np.sin(q * u)
    rad = big + small * cq
    c = np.stack([rad * cu, rad * su, small * sq], axis=1)
    dc = np.stack([-small * q * sq * cu - rad * p * su, -small * q * sq * su + rad * p * cu, small * q * cq], axis=1)
    t = dc / np.linalg.norm(dc, axis=1)[:, None]
    n1 = np.stack([cu * cq, su * cq, sq], axis=1)
    n1 = n1 - (n1 * t).sum(axis=1)[:, None] * t
    n1 /= np.linalg.norm(n1, axis=1)[:, None]
    n2 = np.cross(t, n1)
    v = np.arange(nv, dtype=np.float64) * (2.0 * np.pi / nv)
    rr = tube * (1.0 + ripple * np.cos(5.0 * v)[None, :] * np.cos(7.0 * u)[:, None])          # [nu, nv]
    pts = c[:, None, :] + rr[:, :, None] * (np.cos(v)[None, :, None] * n1[:, None, :] + np.sin(v)[None, :, None] * n2[:, None, :])
    verts = np.ascontiguousarray(pts.reshape(-1, 3), dtype=np.float32)
    i = np.arange(nu)[:, None]; j = np.arange(nv)[None, :]
    a = (i * nv + j).reshape(-1); b = (((i + 1) % nu) * nv + j).reshape(-1)
    cidx = (((i + 1) % nu) * nv + (j + 1) % nv).reshape(-1); d = (i * nv + (j + 1) % nv).reshape(-1)
    f = np.concatenate([np.stack([a, cidx, b], axis=1), np.stack([a, d, cidx], axis=1)], axis=0).astype(np.uint32)
    return verts, np.ascontiguousarray(f)


def cube_mesh():
    """Axis-aligned unit cube (12 triangles, 8 shared vertices), outward winding."""
    v = np.array([[-.5, -.5, -.5], [.5, -.5, -.5], [.5, .5, -.5], [-.5, .5, -.5],
                  [-.5, -.5, .5], [.5, -.5, .5], [.5, .5, .5], [-.5, .5, .5]], dtype=np.float32)
    f = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4],
                  [1, 2, 6], [1, 6, 5], [2, 3, 7], [2, 7, 6], [3, 0, 4], [3, 4, 7]], dtype=np.uint32)
    return v, f


def triangle_soup(vertices, triangles):
    """Every triangle gets private copies of its three vertices (what an STL-style export produces): all edges become
    single-owner seams, which the reference's loader path re-pairs by position (TriangleUtils.cpp:292-420)."""
    v = np.ascontiguousarray(vertices, dtype=np.float32)[np.asarray(triangles).reshape(-1)]
    f = np.arange(len(v), dtype=np.uint32).reshape(-1, 3)
    return v, f


def scan_like_mesh(seed=0, subdivisions=4):
    """What scanned / badly exported meshes bring and the analytic generators above do not, in one mesh (deterministic for a seed):
      * SLIVERS: a band of triangles with aspect ratios 1e3 .. 1e5 (an edge split extremely close to one end);
      * a HIGH-VALENCE vertex: a pole whose fan has 64 triangles;
      * T-JUNCTIONS: edges split on one side only (the neighbour keeps the long edge: a crack with coincident geometry);
      * SELF-INTERSECTION: a second, smaller sphere poking through the first (no boolean: both surfaces stay);
      * UNWELDED SEAMS: part of the surface as a soup (private vertex copies), part of it with jittered duplicates (1e-7);
      * a patch with FLIPPED winding, a few zero-area triangles and one isolated far-away triangle."""
    rng = np.random.default_rng(seed)
    v, f = bumpy_icosphere(subdivisions)
    v = v.astype(np.float64); f = f.astype(np.int64)
    verts = [v]; faces = []
    nv = len(v)
    tri = f.copy()
    # T-junctions + slivers: split an edge of every 7th triangle at t (tiny for slivers), on THIS triangle only
    keep = np.ones(len(tri), bool)
    extra_v, extra_f = [], []
    for k in range(0, len(tri), 7):
        a, b, c = tri[k]
        t = 10.0 ** -rng.uniform(3, 5) if (k // 7) % 2 == 0 else rng.uniform(0.3, 0.7)
        m = nv + sum(len(x) for x in extra_v); extra_v.append(((1 - t) * v[a] + t * v[b])[None])
        extra_f += [[a, m, c], [m, b, c]]
        keep[k] = False
    verts.append(np.concatenate(extra_v)); faces.append(tri[keep]); faces.append(np.array(extra_f, np.int64))
    nv += len(verts[-1])
    # high-valence pole: a 64-triangle fan capping a small circle above the north pole
    ang = np.linspace(0, 2 * np.pi, 64, endpoint=False)
    ring = np.stack([0.12 * np.cos(ang), 0.12 * np.sin(ang), np.full(64, 1.2)], 1); apex = np.array([[0.0, 0.0, 1.32]])
    verts.append(np.concatenate([apex, ring])); faces.append(np.array([[nv, nv + 1 + i, nv + 1 + (i + 1) % 64] for i in range(64)], np.int64)); nv += 65
    # self-intersection: a smaller sphere pushed through the surface
    v2, f2 = icosphere(2)
    verts.append(v2.astype(np.float64) * 0.35 + np.array([0.85, 0.1, 0.0])); faces.append(f2.astype(np.int64) + nv); nv += len(v2)
    V = np.concatenate(verts); F = np.concatenate(faces)
    # unwelded: a third of the triangles get private vertex copies, half of those jittered by 1e-7
    sel = rng.random(len(F)) < 0.33
    soup = V[F[sel].reshape(-1)].copy()
    jit = rng.random(len(soup)) < 0.5
    soup[jit] += rng.normal(0, 1e-7, (int(jit.sum()), 3))
    Fs = np.arange(len(soup), dtype=np.int64).reshape(-1, 3) + len(V)
    V = np.concatenate([V, soup]); F = np.concatenate([F[~sel], Fs])
    # flipped patch, zero-area triangles, an isolated triangle far away
    flip = V[F].mean(1)[:, 0] < -0.8
    F[flip] = F[flip][:, [0, 2, 1]]
    deg = F[rng.integers(0, len(F), 6)].copy(); deg[:, 2] = deg[:, 1]
    far = np.array([[4.0, 4.0, 4.0], [4.1, 4.0, 4.0], [4.0, 4.1, 4.05]]); Ffar = np.array([[len(V), len(V) + 1, len(V) + 2]], np.int64)
    V = np.concatenate([V, far]); F = np.concatenate([F, deg, Ffar])
    return np.ascontiguousarray(V, np.float32), np.ascontiguousarray(F, np.uint32)


def box_with_margin(vertices, margin=0.2):
    """(min xyz, max xyz) float32[6]: mesh bbox grown by margin * largest extent on every side."""
    lo = vertices.min(axis=0).astype(np.float32)
    hi = vertices.max(axis=0).astype(np.float32)
    m = np.float32(margin) * np.float32((hi - lo).max())
    return np.concatenate([lo - m, hi + m]).astype(np.float32)


def random_points_in_box(box6, n, seed=1234):
    """n uniform float32 points inside the cube-ified box the octree will cover."""
    lo, hi = box6[:3].astype(np.float32), box6[3:].astype(np.float32)
    size = np.float32((hi - lo).max())
    c = lo + np.float32(0.5) * (hi - lo)
    cmin = c - np.float32(0.5) * size
    rng = np.random.default_rng(seed)
    u = rng.random((n, 3), dtype=np.float32)
    return np.ascontiguousarray(cmin + u * size * np.float32(0.999999), dtype=np.float32)
