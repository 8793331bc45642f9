"""`.bin` files in the reference's on-disk format (SdfFunction::saveToFile / loadFromFile).

The reference serialises with cereal 1.3.2's PortableBinary archive (src/sdf/SdfFunction.cpp:9-79); cereal is a
third-party dependency that is NOT vendored under /root/reference, so its byte layout is restated here from its
published format: one leading byte = 1 (little-endian archive), then every field in declaration order as raw
little-endian bytes; enums as their underlying int32; std::vector<T> = uint64 element count followed by the elements;
fixed-size std::array and glm vectors/matrices = just their scalars (include/SdfLib/utils/UsefullSerializations.h:6-35).
Field orders: OctreeSdf (include/SdfLib/OctreeSdf.h:222-226), ExactOctreeSdf (include/SdfLib/ExactOctreeSdf.h:138-142),
TriangleData (include/SdfLib/utils/TriangleUtils.h:50-54), BoundingBox (include/SdfLib/utils/Mesh.h:65-69).
PARITY UNPINNED: no file written by an upstream build is available to check against.
"""
import struct

import numpy as np

FORMAT_GRID, FORMAT_OCTREE, FORMAT_EXACT_OCTREE, FORMAT_NONE = 0, 1, 2, 3


def save_octree(path, box6, start_grid_size, max_depth, value_range, min_border_value, words):
    words = np.ascontiguousarray(words, dtype="<u4")
    with open(path, "wb") as f:
        f.write(struct.pack("<B", 1))
        f.write(struct.pack("<i", FORMAT_OCTREE))
        f.write(np.asarray(box6, dtype="<f4").tobytes())
        f.write(struct.pack("<iIffQ", int(start_grid_size), int(max_depth), float(value_range), float(min_border_value), len(words)))
        f.write(words.tobytes())


def save_exact(path, box6, info, nodes, sets, masks, triangle_data):
    nodes = np.ascontiguousarray(nodes, dtype="<u4").reshape(-1, 2)
    sets = np.ascontiguousarray(sets, dtype="<u4"); masks = np.ascontiguousarray(masks, dtype=np.uint8)
    td = np.ascontiguousarray(triangle_data, dtype="<f4").reshape(-1, 37)
    with open(path, "wb") as f:
        f.write(struct.pack("<B", 1))
        f.write(struct.pack("<i", FORMAT_EXACT_OCTREE))
        f.write(np.asarray(box6, dtype="<f4").tobytes())
        f.write(struct.pack("<iIIIIIII", int(info["start_grid_size"]), int(info["start_depth"]), int(info["min_triangles_in_leafs"]),
                            int(info["max_triangles_in_leafs"]), int(info["max_triangles_encoded_in_leafs"]), int(info["bit_encoding_start_depth"]),
                            int(info["bits_per_index"]), int(info["max_depth"])))
        f.write(struct.pack("<Q", len(nodes))); f.write(nodes.tobytes())
        f.write(struct.pack("<Q", len(sets))); f.write(sets.tobytes())
        f.write(struct.pack("<Q", len(masks))); f.write(masks.tobytes())
        f.write(struct.pack("<Q", len(td))); f.write(td.tobytes())


def load(path):
    """Returns ('octree', dict) or ('exact_octree', dict) with numpy arrays; raises ValueError on anything else."""
    buf = open(path, "rb").read()
    if len(buf) < 5 or buf[0] != 1:
        raise ValueError("not a little-endian cereal PortableBinary archive")
    pos = 1
    (fmt,) = struct.unpack_from("<i", buf, pos); pos += 4
    box = np.frombuffer(buf, dtype="<f4", count=6, offset=pos).copy(); pos += 24

    def vec(dtype, width=1):
        nonlocal pos
        (n,) = struct.unpack_from("<Q", buf, pos); pos += 8
        a = np.frombuffer(buf, dtype=dtype, count=n * width, offset=pos).copy(); pos += a.nbytes
        return a

    if fmt == FORMAT_OCTREE:
        g, depth, vr, mb = struct.unpack_from("<iIff", buf, pos); pos += 16
        words = vec("<u4")
        return "octree", dict(box=box, start_grid_size=g, max_depth=depth, value_range=vr, min_border_value=mb, words=words)
    if fmt == FORMAT_EXACT_OCTREE:
        keys = ("start_grid_size", "start_depth", "min_triangles_in_leafs", "max_triangles_in_leafs", "max_triangles_encoded_in_leafs",
                "bit_encoding_start_depth", "bits_per_index", "max_depth")
        vals = struct.unpack_from("<iIIIIIII", buf, pos); pos += 32
        d = dict(zip(keys, vals)); d["box"] = box
        d["nodes"] = vec("<u4", 2).reshape(-1, 2); d["sets"] = vec("<u4"); d["masks"] = vec(np.uint8); d["triangle_data"] = vec("<f4", 37).reshape(-1, 37)
        return "exact_octree", d
    raise ValueError(f"unsupported SdfFormat {fmt} (only OCTREE and EXACT_OCTREE are in scope)")
