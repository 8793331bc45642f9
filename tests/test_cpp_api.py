"""The reference-compatible C++ API (include/SdfLib/*.h) on top of the C ABI: compiles on CPU, runs on the GPU."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, bits

EXE = "/tmp/sdflib_amd_test_cpp_api"


def _compile():
    import sdflib_amd
    if not os.path.exists(sdflib_amd.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "sdflib_amd", "csrc"), "-j8"])
    libdir = os.path.join(ROOT, "sdflib_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_cpp_api.cpp"), "-L", libdir, "-lsdfhip", f"-Wl,-rpath,{libdir}", "-o", EXE])


def test_cpp_headers_compile_and_link():
    _compile()
    assert os.path.exists(EXE)
    # the Unity-style handle interface compiles too
    src = '#include "SdfLib/SdfExportFunc.h"\nint main(){return 0;}\n'
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-x", "c++", "-", "-I", os.path.join(ROOT, "include")], input=src.encode(), check=True)


@pytest.mark.gpu
def test_cpp_api_matches_oracle(tmp_path, oracle):
    from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin, random_points_in_box
    _compile()
    v, f = bumpy_icosphere(2)
    box = box_with_margin(v)
    pts = random_points_in_box(box, 5000, seed=17)
    pts[:50] *= 3.0
    p = lambda n: os.path.join(tmp_path, n)
    v.tofile(p("v.bin")); f.tofile(p("f.bin")); pts.tofile(p("p.bin"))
    r = subprocess.run([EXE, p("v.bin"), p("f.bin"), p("p.bin"), p("d.bin"), p("e.bin"), p("oct.bin"), p("exact.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "scalar-vs-batched mismatches 0" in r.stdout and "reloaded-vs-built mismatches 0" in r.stdout
    # a start grid of 4^3 unit cells: the leaf volumes add up to 8^-startDepth * 64 = 1 (OctreeSdf.cpp:270-276 weights from depth 0)
    total = float(r.stdout.split("depth density levels")[1].split("total")[1].split()[0])
    assert abs(total - 1.0) < 1e-6
    om = oracle.Mesh(v, f)
    # the C++ test computes its box like SdfExporter: bbox + 20 % of the largest extent
    oc = oracle.Octree(om, box, 5, 2, 1e-3, vertex_cache=False, layout=oracle.LAYOUT_SUBTREES)
    d = np.fromfile(p("d.bin"), dtype=np.float32)
    assert np.array_equal(bits(d), bits(oc.query(pts)))
    ex = oracle.Exact(om, box, 5, 1, 16)
    e = np.fromfile(p("e.bin"), dtype=np.float32)
    assert np.array_equal(bits(e), bits(ex.query(pts)))
    # the files the C++ classes wrote parse with the Python restatement of the format and hold the oracle's arrays
    from sdflib_amd import serialization
    kind, d_oct = serialization.load(p("oct.bin"))
    assert kind == "octree" and np.array_equal(d_oct["words"], oc.data()) and d_oct["start_grid_size"] == 4 and d_oct["max_depth"] == 5
    kind, d_ex = serialization.load(p("exact.bin"))
    nodes, has, sets, masks = ex.data()
    assert kind == "exact_octree" and np.array_equal(d_ex["sets"], sets) and np.array_equal(d_ex["masks"], masks)
    assert np.array_equal(d_ex["nodes"][:, 0], nodes[:, 0]) and np.array_equal(d_ex["nodes"][has == 1, 1], nodes[has == 1, 1])
    assert np.array_equal(bits(d_ex["triangle_data"][:, :28]), bits(om.triangle_data()[:, :28]))


@pytest.mark.gpu
def test_cpp_host_drives_the_sharded_build(tmp_path):
    """tests/cpp/test_cpp_sharded.cpp: the shard protocol of the C ABI (build_shard -> sizes -> emit_shard with absolute offsets
    -> gather -> from_data) driven from C++ alone, for 2, 3 and 8 ranks; the reassembled array must equal the single build."""
    from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin
    exe = "/tmp/sdflib_amd_test_cpp_sharded"
    libdir = os.path.join(ROOT, "sdflib_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_cpp_sharded.cpp"),
                           "-L", libdir, "-lsdfhip", f"-Wl,-rpath,{libdir}", "-o", exe])
    v, f = bumpy_icosphere(4)
    box = np.asarray(box_with_margin(v), dtype=np.float32)
    p = lambda n: os.path.join(tmp_path, n)
    v.tofile(p("v.bin")); f.tofile(p("f.bin")); box.tofile(p("box.bin"))
    for ranks in (2, 3, 8):
        r = subprocess.run([exe, p("v.bin"), p("f.bin"), p("box.bin"), str(ranks)], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "sharded-vs-single mismatches 0 value_range_equal 1 min_border_equal 1" in r.stdout and "query_equal 1" in r.stdout, r.stdout
