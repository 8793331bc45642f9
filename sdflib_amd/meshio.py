"""Minimal triangle-mesh readers (OBJ, ASCII / binary PLY) standing in for the reference's assimp loader
(src/utils/Mesh.cpp:9-88: faces triangulated).  Polygons are fan-triangulated.  Host-side IO only.
Difference from assimp, on purpose: an OBJ file's objects / groups / materials are read as ONE mesh (all faces of the file);
assimp splits them into aiMeshes and the reference keeps mMeshes[0] only.  Single-object files behave identically."""
import numpy as np


def _fan(faces):
    tris = []
    for f in faces:
        for k in range(1, len(f) - 1):
            tris.append((f[0], f[k], f[k + 1]))
    return np.array(tris, dtype=np.uint32).reshape(-1, 3)


def read_obj(path):
    verts, faces = [], []
    with open(path, "r", errors="ignore") as fh:
        for line in fh:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                verts.append((float(p[1]), float(p[2]), float(p[3])))
            elif p[0] == "f":
                idx = [int(tok.split("/")[0]) for tok in p[1:]]
                faces.append([i - 1 if i > 0 else len(verts) + i for i in idx])
    return np.array(verts, dtype=np.float32).reshape(-1, 3), _fan(faces)


_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4", "double": "f8",
              "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8"}


def read_ply(path):
    with open(path, "rb") as fh:
        data = fh.read()
    end = data.index(b"end_header") + len(b"end_header")
    header = data[:end].decode("ascii", errors="ignore").splitlines()
    body = data[end + 2:] if data[end:end + 2] == b"\r\n" else data[end + 1:]      # exactly ONE line terminator: the binary body may start with 0x0A
    fmt, elements = None, []
    for line in header:
        p = line.split()
        if not p:
            continue
        if p[0] == "format":
            fmt = p[1]
        elif p[0] == "element":
            elements.append({"name": p[1], "count": int(p[2]), "props": []})
        elif p[0] == "property":
            elements[-1]["props"].append(p[1:])
    verts, faces = None, []
    if fmt == "ascii":
        toks = body.split()
        pos = 0
        for el in elements:
            if el["name"] == "vertex":
                k = len(el["props"])
                arr = np.array(toks[pos:pos + k * el["count"]], dtype=np.float64).reshape(el["count"], k)
                names = [pr[-1] for pr in el["props"]]
                verts = arr[:, [names.index("x"), names.index("y"), names.index("z")]].astype(np.float32)
                pos += k * el["count"]
            elif el["name"] == "face":
                for _ in range(el["count"]):
                    n = int(toks[pos]); faces.append([int(t) for t in toks[pos + 1:pos + 1 + n]]); pos += 1 + n
            else:
                raise ValueError("unsupported PLY element " + el["name"])
        return verts, _fan(faces)
    endian = "<" if fmt == "binary_little_endian" else ">"
    pos = 0
    for el in elements:
        if el["name"] == "vertex":
            dt = np.dtype([(pr[-1], endian + _PLY_TYPES[pr[0]]) for pr in el["props"]])
            arr = np.frombuffer(body, dtype=dt, count=el["count"], offset=pos)
            verts = np.stack([arr["x"], arr["y"], arr["z"]], axis=1).astype(np.float32)
            pos += dt.itemsize * el["count"]
        elif el["name"] == "face":
            pr = el["props"][0]
            assert pr[0] == "list", "face element must be a list property"
            ct, it = np.dtype(endian + _PLY_TYPES[pr[1]]), np.dtype(endian + _PLY_TYPES[pr[2]])
            extra = [np.dtype(endian + _PLY_TYPES[q[0]]).itemsize for q in el["props"][1:]]
            for _ in range(el["count"]):
                n = int(np.frombuffer(body, dtype=ct, count=1, offset=pos)[0]); pos += ct.itemsize
                faces.append(np.frombuffer(body, dtype=it, count=n, offset=pos).astype(np.int64).tolist()); pos += it.itemsize * n + sum(extra)
        else:
            raise ValueError("unsupported PLY element " + el["name"])
    return verts, _fan(faces)


def read_mesh(path):
    low = path.lower()
    if low.endswith(".obj"):
        return read_obj(path)
    if low.endswith(".ply"):
        return read_ply(path)
    raise ValueError("supported mesh formats: .obj, .ply")


def write_ply(path, vertices, triangles):
    v = np.ascontiguousarray(vertices, dtype="<f4"); f = np.ascontiguousarray(triangles, dtype="<i4")
    with open(path, "wb") as fh:
        fh.write((f"ply\nformat binary_little_endian 1.0\nelement vertex {len(v)}\nproperty float x\nproperty float y\nproperty float z\n"
                  f"element face {len(f)}\nproperty list uchar int vertex_indices\nend_header\n").encode())
        fh.write(v.tobytes())
        rec = np.empty(len(f), dtype=[("n", "u1"), ("i", "<i4", 3)]); rec["n"] = 3; rec["i"] = f
        fh.write(rec.tobytes())
