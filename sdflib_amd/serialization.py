"""`.bin` files in the reference's on-disk format (SdfFunction::saveToFile / loadFromFile).

The reference serialises with cereal 1.3.2's PortableBinary archive (src/sdf/SdfFunction.cpp:9-79); cereal is a
third-party dependency that is NOT vendored under /root/reference, so its byte layout is restated here from its
published format: one leading byte = 1 (little-endian archive), then every field in `archive(...)` order as raw
little-endian bytes; enums as their underlying int32; std::vector<T> = uint64 element count followed by the elements;
fixed-size std::array and glm vectors/matrices = just their scalars (include/SdfLib/utils/UsefullSerializations.h:6-35).

The writers and the reader below are driven by the field tables OCTREE_FIELDS / EXACT_FIELDS / TRIANGLE_DATA_FIELDS.
The tables (names, order, scalar types) are compared with the reference's own `archive(...)` lists and member
declarations by tools/refpin (group_archive) — OctreeSdf.h:222-226, ExactOctreeSdf.h:138-142, TriangleUtils.h:50-54,
Mesh.h:65-69, the node `serialize` bodies.  What stays unpinned is cereal's encoding of those fields (the leading byte,
the uint64 vector count): no file written by an upstream build is available.
"""
import struct

import numpy as np

FORMAT_GRID, FORMAT_OCTREE, FORMAT_EXACT_OCTREE, FORMAT_NONE = 0, 1, 2, 3      # SdfFunction::SdfFormat (SdfFunction.h:16-22)

# (reference member name, layout, key used on the Python side).  Layout: "<n><type>" = n scalars; "vec:<n><type>" = uint64 count + elements.
TRIANGLE_DATA_FIELDS = [("origin", "3f4"), ("transform", "9f4"), ("b", "2f4"), ("c", "2f4"), ("v2", "1f4"), ("v3", "2f4"),
                        ("edgesNormal", "9f4"), ("verticesNormal", "9f4")]
TRIANGLE_DATA_FLOATS = sum(int(l[:-2]) for _, l in TRIANGLE_DATA_FIELDS)          # 37
OCTREE_FIELDS = [("mBox", "6f4", "box"), ("mStartGridSize", "1i4", "start_grid_size"), ("mMaxDepth", "1u4", "max_depth"),
                 ("mValueRange", "1f4", "value_range"), ("mMinBorderValue", "1f4", "min_border_value"), ("mOctreeData", "vec:1u4", "words")]
EXACT_FIELDS = [("mBox", "6f4", "box"), ("mStartGridSize", "1i4", "start_grid_size"), ("mStartDepth", "1u4", "start_depth"),
                ("mMinTrianglesInLeafs", "1u4", "min_triangles_in_leafs"), ("mMaxTrianglesInLeafs", "1u4", "max_triangles_in_leafs"),
                ("mMaxTrianglesEncodedInLeafs", "1u4", "max_triangles_encoded_in_leafs"), ("mBitEncodingStartDepth", "1u4", "bit_encoding_start_depth"),
                ("mBitsPerIndex", "1u4", "bits_per_index"), ("mMaxDepth", "1u4", "max_depth"),
                ("mOctreeData", "vec:2u4", "nodes"), ("mTrianglesSets", "vec:1u4", "sets"), ("mTrianglesMasks", "vec:1u1", "masks"),
                ("mTrianglesData", "vec:%df4" % TRIANGLE_DATA_FLOATS, "triangle_data")]
_NP = {"f4": "<f4", "i4": "<i4", "u4": "<u4", "u1": "u1"}


def _split(layout):
    vec = layout.startswith("vec:")
    body = layout[4:] if vec else layout
    return vec, int(body[:-2]), _NP[body[-2:]]


def _write(f, fields, values):
    for name, layout, key in fields:
        vec, width, dt = _split(layout)
        a = np.ascontiguousarray(values[key], dtype=dt).reshape(-1)
        if vec:
            assert a.size % width == 0, (name, a.size, width)
            f.write(struct.pack("<Q", a.size // width))
        else:
            assert a.size == width, (name, a.size, width)
        f.write(a.tobytes())


def _read(buf, pos, fields):
    out = {}
    for name, layout, key in fields:
        vec, width, dt = _split(layout)
        n = 1
        if vec:
            (n,) = struct.unpack_from("<Q", buf, pos); pos += 8
        a = np.frombuffer(buf, dtype=dt, count=n * width, offset=pos).copy(); pos += a.nbytes
        if vec:
            out[key] = a.reshape(-1, width) if width > 1 else a
        else:
            out[key] = a if width > 1 else a[0].item()
    return out, pos


def save_octree(path, box6, start_grid_size, max_depth, value_range, min_border_value, words):
    with open(path, "wb") as f:
        f.write(struct.pack("<B", 1))
        f.write(struct.pack("<i", FORMAT_OCTREE))
        _write(f, OCTREE_FIELDS, dict(box=box6, start_grid_size=int(start_grid_size), max_depth=int(max_depth), value_range=float(value_range),
                                      min_border_value=float(min_border_value), words=words))


def save_exact(path, box6, info, nodes, sets, masks, triangle_data):
    with open(path, "wb") as f:
        f.write(struct.pack("<B", 1))
        f.write(struct.pack("<i", FORMAT_EXACT_OCTREE))
        v = {k: int(info[k]) for _, l, k in EXACT_FIELDS if not l.startswith("vec:") and k != "box"}
        v.update(box=box6, nodes=nodes, sets=sets, masks=masks, triangle_data=triangle_data)
        _write(f, EXACT_FIELDS, v)


def load(path):
    """Returns ('octree', dict) or ('exact_octree', dict) with numpy arrays; raises ValueError on anything else."""
    buf = open(path, "rb").read()
    if len(buf) < 5 or buf[0] != 1:
        raise ValueError("not a little-endian cereal PortableBinary archive")
    (fmt,) = struct.unpack_from("<i", buf, 1)
    if fmt == FORMAT_OCTREE:
        d, _ = _read(buf, 5, OCTREE_FIELDS)
        return "octree", d
    if fmt == FORMAT_EXACT_OCTREE:
        d, _ = _read(buf, 5, EXACT_FIELDS)
        return "exact_octree", d
    raise ValueError(f"unsupported SdfFormat {fmt} (only OCTREE and EXACT_OCTREE are in scope)")
