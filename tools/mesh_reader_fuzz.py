"""Differential fuzzing of the two mesh readers (CPU only): sdflib_amd.meshio (Python) against sdflib::Mesh(path) of
include/SdfLib/utils/Mesh.h (C++) on randomly written OBJ / PLY files — polygons, negative OBJ indices, v/vt/vn tokens, comments,
CRLF, extra vertex and face properties, ASCII and both binary byte orders.  Usage: tools/mesh_reader_fuzz.py <dump-exe> [cases]."""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from sdflib_amd import meshio

exe = sys.argv[1]; cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200


def write_obj(path, v, faces, rng):
    nl = "\r\n" if rng.random() < 0.3 else "\n"
    lines = ["# fuzz"]
    order = rng.random() < 0.5
    def fline(f, nv_so_far):
        toks = []
        style = rng.integers(0, 4)
        for i in f:
            idx = (i + 1) if (order or rng.random() < 0.7) else (i - nv_so_far)          # negative = relative to the vertices read so far
            toks.append(str(idx) if style == 0 else (f"{idx}/1" if style == 1 else (f"{idx}//1" if style == 2 else f"{idx}/1/1")))
        return "f " + " ".join(toks)
    for p in v: lines.append(f"v {p[0]:.9g} {p[1]:.9g} {p[2]:.9g}")
    if rng.random() < 0.5: lines += ["vt 0 0", "vn 0 0 1", "", "g part", "s off", "usemtl m"]
    for f in faces: lines.append(fline(f, len(v)))
    open(path, "w", newline="").write(nl.join(lines) + nl)


def write_ply(path, v, faces, rng):
    mode = rng.choice(["ascii", "binary_little_endian", "binary_big_endian"])
    e = "<" if mode != "binary_big_endian" else ">"
    extra_v = rng.random() < 0.5; extra_f = rng.random() < 0.3 and mode != "ascii"
    vt = rng.choice(["float", "double", "float32"]); ct = rng.choice(["uchar", "uint8", "int"]); it = rng.choice(["int", "uint", "int32", "ushort"]) if len(v) < 60000 else "int"
    nl = "\r\n" if rng.random() < 0.3 else "\n"
    h = ["ply", f"format {mode} 1.0", "comment fuzz", f"element vertex {len(v)}"]
    props = [("x", vt), ("y", vt), ("z", vt)]
    if extra_v: props = [("nx", "float")] + props + [("red", "uchar")] if rng.random() < 0.5 else props + [("nx", "float"), ("red", "uchar")]
    for n, t in props: h.append(f"property {t} {n}")
    h += [f"element face {len(faces)}", f"property list {ct} {it} vertex_indices"]
    if extra_f: h.append("property uchar flags")
    h.append("end_header")
    T = meshio._PLY_TYPES
    with open(path, "wb") as fh:
        fh.write((nl.join(h) + nl).encode())
        if mode == "ascii":
            for p in v:
                row = []
                for n, t in props: row.append(f"{p['xyz'.index(n)]:.9g}" if n in "xyz" else ("0.5" if n == "nx" else "7"))
                fh.write((" ".join(row) + "\n").encode())
            for f in faces: fh.write((" ".join([str(len(f))] + [str(i) for i in f]) + "\n").encode())
        else:
            dt = np.dtype([(n, e + T[t]) for n, t in props]); arr = np.zeros(len(v), dtype=dt)
            for k, n in enumerate("xyz"): arr[n] = v[:, k]
            fh.write(arr.tobytes())
            for f in faces:
                fh.write(np.array([len(f)], dtype=e + T[ct]).tobytes()); fh.write(np.array(f, dtype=e + T[it]).tobytes())
                if extra_f: fh.write(b"\x01")


fails = 0
with tempfile.TemporaryDirectory() as tmp:
    for seed in range(cases):
        rng = np.random.default_rng(seed)
        nv = int(rng.integers(5, 40)); v = rng.normal(0, 1, (nv, 3)).astype(np.float32)
        faces = [list(rng.choice(nv, int(rng.choice([3, 3, 3, 4, 5])), replace=False)) for _ in range(int(rng.integers(1, 30)))]
        ext = "obj" if rng.random() < 0.4 else "ply"
        path = os.path.join(tmp, f"m{seed}.{ext}")
        (write_obj if ext == "obj" else write_ply)(path, v, faces, rng)
        try:
            pv, pf = meshio.read_mesh(path)
            out = os.path.join(tmp, "dump.bin")
            r = subprocess.run([exe, path, out], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            raw = open(out, "rb").read(); n_v, n_i = np.frombuffer(raw, dtype=np.uint32, count=2)
            cv = np.frombuffer(raw, dtype=np.float32, count=3 * n_v, offset=8).reshape(-1, 3); ci = np.frombuffer(raw, dtype=np.uint32, count=n_i, offset=8 + 12 * n_v).reshape(-1, 3)
            bb = np.frombuffer(raw, dtype=np.float32, count=6, offset=8 + 12 * n_v + 4 * n_i)
            assert cv.shape == pv.shape and np.array_equal(cv.view(np.uint32), pv.view(np.uint32)), "vertices"
            assert ci.shape == pf.shape and np.array_equal(ci, pf), "triangles"
            assert np.array_equal(bb, np.concatenate([pv.min(0), pv.max(0)])), "bounding box"
            want = np.array([(f[0], f[k], f[k + 1]) for f in faces for k in range(1, len(f) - 1)], dtype=np.uint32)
            assert np.array_equal(pf, want) and np.allclose(pv, v, rtol=1e-6, atol=0), "content"
        except Exception as e:      # noqa: BLE001
            fails += 1; print(f"seed {seed} ({ext}): {type(e).__name__} {e}")
print(f"{cases} files, {fails} failures")
sys.exit(1 if fails else 0)
