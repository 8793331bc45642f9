"""SdfExporter-compatible command line (reference src/tools/SdfExporter/main.cpp:20-171): mesh -> (normalise) -> box with
margin -> build on the GPU -> .bin file in the reference's layout.  Same flag names and defaults as the reference tool.

    python -m sdflib_amd.exporter model.ply out.bin --sdf_format octree --algorithm continuity -d 8 --start_depth 1
"""
import argparse
import sys
import time

import numpy as np

from . import api, meshio


def main(argv=None):
    ap = argparse.ArgumentParser(prog="SdfExporter", description="SdfExporter export an sdf")
    ap.add_argument("model_path"); ap.add_argument("output_path", nargs="?", default="sdf.bin")
    ap.add_argument("-c", "--cell_size", type=float)
    ap.add_argument("-d", "--depth", type=int)
    ap.add_argument("--start_depth", type=int, default=1)
    ap.add_argument("--termination_rule", default="trapezoidal_rule")
    ap.add_argument("--termination_threshold", type=float, default=1e-3)
    ap.add_argument("--termination_threshold_by_distance", type=float, default=0.0)
    ap.add_argument("--min_triangles_per_node", type=int, default=32)
    ap.add_argument("--sdf_format", default="octree")
    ap.add_argument("--algorithm", default="continuity")
    ap.add_argument("-n", "--normalize", action="store_true")
    ap.add_argument("--bb_margin", type=float, default=20.0)
    ap.add_argument("--num_threads", type=int, default=1)
    a = ap.parse_args(argv)

    v, f = meshio.read_mesh(a.model_path)
    lo, hi = v.min(axis=0).astype(np.float32), v.max(axis=0).astype(np.float32)
    if a.normalize:                                   # scale the largest extent to 2 and centre (main.cpp:83-90)
        size = np.float32((hi - lo).max())
        centre = (lo + np.float32(0.5) * (hi - lo)).astype(np.float32)
        s = np.float32(2.0) / size                        # scale(mat4(1), 2/size) * translate(mat4(1), -centre) applied as mat4 * vec4:
        v = ((s * v).astype(np.float32) + (s * -centre).astype(np.float32)).astype(np.float32)      # fl(fl(s x) + fl(s (-c))), Mesh.cpp:131-139
        lo, hi = v.min(axis=0).astype(np.float32), v.max(axis=0).astype(np.float32)
    margin = np.float32(a.bb_margin / 100.0) * np.float32((hi - lo).max())
    box = np.concatenate([lo - margin, hi + margin]).astype(np.float32)

    mesh = api.Mesh(v, f, bbox=np.concatenate([lo, hi]))   # loader-computed box => seam welding as in the reference tool
    t0 = time.perf_counter()
    if a.sdf_format == "octree":
        alg = {"uniform": api.ALG_UNIFORM, "no_continuity": api.ALG_NO_CONTINUITY, "continuity": api.ALG_CONTINUITY}.get(a.algorithm)
        if alg is None:
            print(f"{a.algorithm} is not a valid supported octree generation algorithm", file=sys.stderr); return 0
        rule = api.string_to_termination_rule(a.termination_rule)
        if rule is None:
            print(f"{a.termination_rule} is not a valid termination rule", file=sys.stderr); return 0
        params = (a.termination_threshold, a.termination_threshold_by_distance if rule == api.RULE_BY_DISTANCE else 0.0)
        sdf = api.OctreeSdf(mesh, box, a.depth if a.depth is not None else 8, a.start_depth, init_algorithm=alg, num_threads=a.num_threads,
                            termination_rule=rule, rule_params=params)
        print(f"[info] Computation time {time.perf_counter() - t0:.3f}s", file=sys.stderr)
        sdf.save_to_file(a.output_path)
    elif a.sdf_format == "exact_octree":
        sdf = api.ExactOctreeSdf(mesh, box, a.depth if a.depth is not None else 5, a.start_depth, a.min_triangles_per_node, a.num_threads)
        print(f"[info] Computation time {time.perf_counter() - t0:.3f}s", file=sys.stderr)
        sdf.save_to_file(a.output_path, mesh)
    else:
        print("The sdf_format can only be octree or exact_octree (the uniform grid is out of scope)", file=sys.stderr); return 1
    print("[info] Saving the model", file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
