"""Seeded query-point sets shared by the reference-pin tests and the fixture generator (tests/golden/make_golden.py)."""
import numpy as np

from sdflib_amd import meshgen


def tie_points(v, f, rng, n_random=20000):
    """Random points in the padded box, far-field points, near-surface points, and points EXACTLY on vertices,
    edge midpoints and centroids (several nearest triangles at equal distance: the id depends on the traversal order)."""
    v = np.asarray(v, np.float32); f = np.asarray(f)
    box = meshgen.box_with_margin(v)
    lo, hi = box[:3], box[3:]
    pts = [meshgen.random_points_in_box(box, n_random, seed=int(rng.integers(1 << 30)))]
    ext = (hi - lo).max()
    pts.append((lo + (rng.random((n_random // 10, 3)).astype(np.float32) - 0.5) * 6 * ext).astype(np.float32))     # outside, far
    sel = rng.integers(0, len(f), min(len(f), 6000))
    tri = v[f[sel]]
    pts.append(v[rng.integers(0, len(v), min(len(v), 4000))])                                                       # on vertices
    pts.append(((tri[:, 0] + tri[:, 1]) * np.float32(0.5)).astype(np.float32))                                      # on edges
    pts.append(((tri[:, 1] + tri[:, 2]) * np.float32(0.5)).astype(np.float32))
    pts.append(((tri[:, 0] + tri[:, 1] + tri[:, 2]) / np.float32(3)).astype(np.float32))                            # on faces
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]).astype(np.float32)
    nn = np.linalg.norm(n, axis=1, keepdims=True); nn[nn == 0] = 1
    for eps in (1e-6, 1e-3, 5e-2):                                                                                  # near the surface, both sides
        pts.append((tri[:, 0] + (n / nn) * np.float32(eps) * ext).astype(np.float32))
        pts.append((tri[:, 2] - (n / nn) * np.float32(eps) * ext).astype(np.float32))
    pts.append(np.zeros((1, 3), np.float32))
    pts.append(v.mean(axis=0, keepdims=True).astype(np.float32))                                                    # centre: everything is far and tied-ish
    return np.ascontiguousarray(np.concatenate(pts), np.float32)


def ref_fixture_cases():
    """(name, vertices, triangles, points) of tests/golden/ref_nearest_large.npz: the full-size meshes of the BASELINE configs."""
    out = []
    for name, (v, f), seed in (("bumpy7", meshgen.bumpy_icosphere(7), 11), ("knot", meshgen.torus_knot(), 12), ("bumpy8", meshgen.bumpy_icosphere(8), 13)):
        v = np.ascontiguousarray(v, np.float32); f = np.ascontiguousarray(f, np.uint32)
        out.append((name, v, f, tie_points(v, f, np.random.default_rng(seed), n_random=60000)))
    return out


def points_digest(pts):
    """Order-dependent 64-bit digest of the points' bit patterns (the fixture holds ids only; this ties them to their points)."""
    w = np.ascontiguousarray(pts, np.float32).view(np.uint32).astype(np.uint64).ravel()
    k = (np.arange(len(w), dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) | np.uint64(1)
    return int(np.bitwise_xor.reduce(w * k))
