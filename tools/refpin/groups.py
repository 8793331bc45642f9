"""The pinned groups: which function of the reference is compared with which restatement in oracle/, and the vocabulary
that maps one onto the other.  Run through tools/check_ref_expressions.py (and tests/test_oracle_ref_pin.py)."""
import os
import re
from . import cparse, symex

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def read(path):
    with open(path) as f:
        return f.read()


def fn(text, header, nth=0):
    params, body = cparse.find_function(text, header, nth)
    return cparse.parse_params(params), cparse.parse_body(body)


def compare(name, a, b):
    d = symex.first_difference(a, b)
    if d:
        raise AssertionError("%s: reference and oracle differ at %s" % (name, d))


def paths_of(ex, params, body, **kw):
    return ex.run(params, body, **kw)


# ---------------------------------------------------------------------------------------------------------------------
def group_triangle_data(orc_triangle=None):
    """TriangleData constructor (TriangleUtils.h:23-42) and getTriangleNormal (:45-48) vs makeTriangleData / normal()."""
    ref = read(REF + "/include/SdfLib/utils/TriangleUtils.h")
    orc = orc_triangle or read(REPO + "/oracle/orc_triangle.h")
    rp, rb = fn(ref, r"TriangleData\s*\(glm::vec3 v1")
    op, ob = fn(orc, r"TriangleData makeTriangleData\s*\(")
    e = symex.Exec(); e.ctor_mode = True
    r = e.run(rp, rb, ctor=True)
    o = symex.Exec().run(op, ob)
    assert len(r) == 1 and len(o) == 1 and not r[0]["conds"] and not o[0]["conds"]
    ret = o[0]["end"]
    assert ret[0] == "ret" and ret[1][0] == "agg"
    compare("TriangleData ctor", dict(r[0]["obj"]), dict(ret[1][1:]))
    rp, rb = fn(ref, r"glm::vec3 getTriangleNormal\s*\(")
    op, ob = fn(orc, r"V3 normal\s*\(")
    r = symex.Exec(id_alias={}).run(rp, rb)
    o = symex.Exec().run(op, ob)
    # reference: transform[c][2]; oracle: transform.c[c].z   (glm's m[col][row])
    compare("getTriangleNormal", r, o)
    return "TriangleData constructor: 8 fields identical; getTriangleNormal identical"


DIST_VARIANTS = [
    ("getSqDistPointAndTriangle(point, data)", r"float getSqDistPointAndTriangle\s*\(glm::vec3 point", 0, r"float sqDistPointTriangle\s*\("),
    ("getSignedDistPointAndTriangle(point, data)", r"float getSignedDistPointAndTriangle\s*\(", 0, r"float signedDistPointTriangle\s*\("),
    ("getSignedDistPointAndTriangle(point, data, v1, v2, v3, outNormal)", r"float getSignedDistPointAndTriangle\s*\(", 1, r"float signedDistPointTriangleGrad\s*\("),
    ("getSignedDistPointAndTriangle(point, data, outNormal)", r"float getSignedDistPointAndTriangle\s*\(", 2, r"float signedDistPointTriangleGradLocal\s*\("),
    ("getSqDistPointAndTriangle(p, a, b, c)", r"float getSqDistPointAndTriangle\s*\(glm::vec3 p,", 0, r"float sqDistPointTriangleRaw\s*\("),
]


def group_point_triangle(orc_triangle=None):
    """The four point/triangle routines + the raw-vertex variant (TriangleUtils.h:76-404): per Voronoi region, the same
    region tests in the same order and the same returned / out-normal expressions."""
    ref = read(REF + "/include/SdfLib/utils/TriangleUtils.h")
    orc = orc_triangle or read(REPO + "/oracle/orc_triangle.h")
    classify = fn(orc, r"Proj classify\s*\(")
    dot2 = fn(ref, r"float dot2\s*\(")
    n_paths = 0
    for name, rh, nth, oh in DIST_VARIANTS:
        rp, rb = fn(ref, rh, nth)
        op, ob = fn(orc, oh)
        r = symex.Exec(funcs={"dot2": [dot2]}).run(rp, rb)
        o = symex.Exec(funcs={"classify": [classify]}, member_alias={"normal": "getTriangleNormal"}).run(op, ob)
        compare(name, r, o)
        n_paths += len(r)
    return "point/triangle distance: 5 routines, %d region paths identical (tests, order, returned and out-normal expressions)" % n_paths


GROUPS = [group_triangle_data, group_point_triangle]


# ---------------------------------------------------------------------------------------------------------------------
def vec_table(text, anchor):
    """the 8 corner offsets: numbers of the brace list that follows `anchor`"""
    blk = text[text.index(anchor):]
    blk = blk[:blk.index("};")]
    blk = blk[blk.index("=") + 1:]
    nums = [float(x.rstrip("f")) for x in re.findall(r"-?\d+\.\d*f?", cparse.preprocess(blk))]
    assert len(nums) == 24, (anchor, len(nums))
    return [tuple(nums[i:i + 3]) for i in range(0, 24, 3)]


def group_gjk(orc_exact=None, orc_octree=None):
    """GJK.cpp:644-652 (triangle support), :702-738 (corner table, sphere-hull support, difference), :830-866 (IsNearMinimize)."""
    ref = read(REF + "/src/utils/GJK.cpp")
    orc = orc_exact or read(REPO + "/oracle/orc_exact.h")
    octree = orc_octree or read(REPO + "/oracle/orc_octree.h")
    # ignored on both sides: the iteration-count out-parameter (statistics only)
    ref_p = ref.replace("uint32_t& iter = (pIter == nullptr) ? dIter : *pIter;", "uint32_t iter;")
    assert ref_p != ref
    orc_p = orc.replace("if (iters) *iters = iter;", "")
    assert orc_p.count("iters") == 1, "oracle isNearMinimize: unexpected uses of the iteration out-parameter"
    corners = vec_table(octree, "CORNER_REL[8]")
    assert vec_table(ref, "std::array<glm::vec3, 8> childrens") == corners
    assert vec_table(read(REF + "/include/SdfLib/TrianglesInfluence.h"), "std::array<glm::vec3, 8> childrens") == corners
    assert vec_table(read(REF + "/src/sdf/OctreeSdf.cpp"), "std::array<glm::vec3, 8> childrens") == corners
    alias = {"furthestOnHull": "findFurthestPoint", "furthestOnTriangle": "findFurthestPoint"}
    ids = {"CORNER_REL": "childrens"}
    # triangle support
    r = symex.Exec().run(*fn(ref_p, r"glm::vec3 findFurthestPoint\s*\(const std::array<glm::vec3, 3>& triangle, const glm::vec3& direction"))
    o = symex.Exec(fn_alias=alias, id_alias=ids).run(*fn(orc_p, r"V3 furthestOnTriangle\s*\("))
    compare("findFurthestPoint(triangle, direction)", r, o)
    n = len(r)
    # hull-of-spheres support: the 7-step argmax unrolls into 128 paths on each side
    r = symex.Exec().run(*fn(ref_p, r"glm::vec3 findFurthestPoint\s*\(float halfNodeSize, const std::array<float, 8>& vertRadius, const glm::vec3& direction"))
    o = symex.Exec(fn_alias=alias, id_alias=ids).run(*fn(orc_p, r"V3 furthestOnHull\s*\("))
    compare("findFurthestPoint(halfNodeSize, vertRadius, direction)", r, o)
    n += len(r)
    # IsNearMinimize, with the reference's 4-argument support inlined (hull support minus triangle support of -direction)
    diff = fn(ref_p, r"glm::vec3 findFurthestPoint\s*\(float halfNodeSize,\s*const std::array<float, 8>& vertRadius,\s*const std::array<glm::vec3, 3>& triangle")
    r = symex.Exec(funcs={"findFurthestPoint": [diff]}).run(*fn(ref_p, r"bool IsNearMinimize\s*\(float halfNodeSize"))
    o = symex.Exec(fn_alias=alias, id_alias=ids, local_alias={"cur": "currentPoint"}).run(*fn(orc_p, r"bool isNearMinimize\s*\("))
    compare("IsNearMinimize", r, o)
    return "GJK: corner table (3 copies), 2 supports (%d paths), Frank-Wolfe step, exits and iteration cap of IsNearMinimize identical" % n


GROUPS.append(group_gjk)


# ---------------------------------------------------------------------------------------------------------------------
def group_filter_triangles(orc_exact=None):
    """PerNodeRegionTrianglesInfluence::filterTriangles (TrianglesInfluence.h:767-860): the 8x8 corner radii, the centroid
    octant and the keep rule."""
    ref = read(REF + "/include/SdfLib/TrianglesInfluence.h")
    orc = orc_exact or read(REPO + "/oracle/orc_exact.h")
    seg = ref[ref.index("struct PerNodeRegionTrianglesInfluence"):]
    ref_p = seg.replace(", minDistToVertices[vId], &iter))", ", minDistToVertices[vId]))")      # iteration-count out-parameter (statistics)
    assert ref_p != seg
    orc_p = orc.replace("if (n.vi[vId] != idx) cullTests++;", "")                                # oracle-only statistics counter
    assert orc_p != orc
    rp, rb = fn(ref_p, r"inline void filterTriangles\s*\(")
    op, ob = fn(orc_p, r"void filterTriangles\s*\(")

    def rw_ref(x):
        if x[0] == "mcall" and x[1] == ("id", "mesh") and x[2] in ("getVertices", "getIndices"):
            return ("id", "vertices" if x[2] == "getVertices" else "indices")
        return x

    def rw_orc(x):
        if x[0] == "member" and x[1] == ("id", "n"):
            return ("id", {"vi": "verticesInfo", "center": "nodeCenter", "size": "nodeHalfSize"}[x[2]])
        if x[0] == "member" and x[1] == ("id", "mesh"):
            return ("id", x[2])
        if x == ("un", "*", ("id", "tris")):
            return ("id", "trianglesData")
        return x
    er = symex.Exec(); er.rewrite = rw_ref
    eo = symex.Exec(fn_alias={"sqDistPointTriangle": "getSqDistPointAndTriangle", "isNearMinimize": "IsNearMinimize"},
                    id_alias={"CORNER_REL": "childrens", "in": "inTriangles", "outList": "outTriangles"}, local_alias={"keep": "isInside"})
    eo.rewrite = rw_orc
    r = er.run([], rb)
    o = eo.run([], ob)
    compare("filterTriangles", r, o)
    return "filterTriangles: 64 corner radii (sqrt of the squared distance, minus the per-vertex minimum), centroid octant, keep rule identical (%d paths)" % len(r)


def leaf_rewrites(x):
    """oracle spelling of the node word -> the reference's accessors"""
    if x[0] == "bin" and x[1] == "&" and x[3] == ("id", "LEAF_BIT"):
        return ("mcall", x[2], "isLeaf", ())
    if x[0] == "bin" and x[1] == "&" and x[3] == ("id", "INDEX_MASK"):
        return ("mcall", x[2], "getChildrenIndex", ())
    return x


def coeff_rewrites(x):
    """`reinterpret_cast<...>(&data[i])` and its dereference name the 64 coefficients that start at word i, on both sides"""
    if x[0] == "cast" and x[1].rstrip().endswith("*"):
        inner = x[2]
        return ("coeffs", inner[2] if isinstance(inner, tuple) and inner[0] == "un" and inner[1] == "&" else inner)
    if x[0] == "un" and x[1] == "*" and isinstance(x[2], tuple) and x[2][0] == "coeffs":
        return x[2]
    return x


def group_min_border(orc_octree=None):
    """OctreeSdf::computeMinBorderValue (OctreeSdf.cpp:155-230)."""
    ref = read(REF + "/src/sdf/OctreeSdf.cpp")
    orc = orc_octree or read(REPO + "/oracle/orc_octree.h")
    rp, rb = fn(ref, r"void OctreeSdf::computeMinBorderValue\s*\(")
    op, ob = fn(orc, r"static inline void computeMinBorder\s*\(OctreeSdfData& out\)\s*\{")

    def rw_orc(x):
        x = coeff_rewrites(leaf_rewrites(x))
        if x[0] == "member" and x[1] == ("id", "out"):
            return ("id", {"data": "mOctreeData", "startGridSize": "mStartGridSize", "startGridXY": "mStartGridXY", "minBorderValue": "mMinBorderValue"}[x[2]])
        return x
    er = symex.Exec(); er.rewrite = coeff_rewrites
    eo = symex.Exec(fn_alias={"tricubicValue": "interpolateValue", "rec": "processNode"}, id_alias={"CORNER_REL": "childrens", "d": "mOctreeData"},
                    local_alias={"mn": "minValue"})
    eo.rewrite = rw_orc
    # the reference declares the corner table locally; the oracle reads its file-scope copy (values compared in group_gjk too)
    eo.globals["CORNER_REL"] = symex.Agg("corners", {i: ("vec3",) + tuple(("lit", "float", c) for c in v) for i, v in enumerate(vec_table(orc, "CORNER_REL[8]"))})
    r = er.run([], rb)
    o = eo.run([], ob)
    compare("computeMinBorderValue", r, o)
    return "computeMinBorderValue: border tests, corner evaluation points, recursion arguments and grid walk identical (%d paths)" % len(r)


GROUPS += [group_filter_triangles, group_min_border]


# ---------------------------------------------------------------------------------------------------------------------
def group_box_and_query(orc_octree=None):
    """BoundingBox::getSize/getCenter/getDistance x2 (Mesh.h:26-63) and OctreeSdf::getDistance x2 + roundFloat (OctreeSdf.cpp:88-152)."""
    mesh_h = read(REF + "/include/SdfLib/utils/Mesh.h")
    ref = read(REF + "/src/sdf/OctreeSdf.cpp")
    orc = orc_octree or read(REPO + "/oracle/orc_octree.h")
    box = mesh_h[mesh_h.index("struct BoundingBox"):]

    def rw_box(x):                       # oracle: free functions over a Box b; reference: methods of the box
        if x[0] == "mcall" and x[1] == ("id", "b") and x[2] in ("size", "center"):
            return ("call", "getSize" if x[2] == "size" else "getCenter", ())
        if x[0] == "call" and x[1] == "boxDistance" and len(x[2]) == 2 and x[2][0] == ("id", "b"):
            return ("call", "getDistance", (x[2][1],))
        return x
    n = 0
    for name, rh, oh in [("getSize", r"glm::vec3 getSize\s*\(", r"V3 size\s*\("), ("getCenter", r"glm::vec3 getCenter\s*\(", r"V3 center\s*\(")]:
        r = symex.Exec().run([], fn(box, rh)[1])
        eo = symex.Exec(); eo.rewrite = lambda x: ("call", "getSize", ()) if x == ("call", "size", ()) else x
        compare("BoundingBox::" + name, r, eo.run([], fn(orc, oh)[1]))
    for name, rh, nth, oh in [("getDistance(point)", r"float getDistance\s*\(glm::vec3 point\)", 0, r"float boxDistance\s*\("),
                              ("getDistance(point, outGradient)", r"float getDistance\s*\(glm::vec3 point, glm::vec3& outGradient\)", 0, r"float boxDistanceGrad\s*\(")]:
        r = symex.Exec().run([], fn(box, rh, nth)[1])
        eo = symex.Exec(id_alias={"p": "point", "g": "outGradient"}); eo.rewrite = rw_box
        o = eo.run([], fn(orc, oh)[1])
        compare("BoundingBox::" + name, r, o)
        n += len(r)
    # roundFloat
    compare("roundFloat", symex.Exec().run(*fn(ref, r"inline uint32_t roundFloat\s*\(")), symex.Exec().run(*fn(orc, r"uint32_t roundFloatGE\s*\(")))

    # OctreeSdf::getDistance: the oracle holds both variants in one function, selected by `grad`
    def rw_ref(x):
        x = coeff_rewrites(x)
        if x[0] == "un" and x[1] in ("&", "*"):                     # node pointers: `currentNode = &data[i]`, `currentNode->f()`
            return x[2]
        if x[0] == "mcall" and x[1] == ("id", "mBox") and x[2] == "getDistance":
            return ("call", "boxDistance", x[3])
        return x

    def rw_orc(x):
        x = coeff_rewrites(leaf_rewrites(x))
        if x[0] == "un" and x[1] in ("&", "*"):
            return x[2]
        if x[0] == "member" and x[1] == ("id", "o"):
            return ("id", {"box": "mBox", "startGridCellSize": "mStartGridCellSize", "startGridSize": "mStartGridSize", "startGridXY": "mStartGridXY",
                           "data": "mOctreeData", "minBorderValue": "mMinBorderValue"}[x[2]])
        if x[0] == "call" and x[1] in ("boxDistance", "boxDistanceGrad") and x[2][0] == ("id", "mBox"):
            return ("call", "boxDistance", x[2][1:])
        return x
    for k, (name, flag) in enumerate([("getDistance(sample)", "false"), ("getDistance(sample, outGradient)", "true")]):
        rp, rb = fn(ref, r"float OctreeSdf::getDistance\s*\(", k)
        orc_p = orc.replace("if (grad)", "if (%s)" % flag)
        assert orc_p.count("if (%s)" % flag) == 2
        op, ob = fn(orc_p, r"float octreeDistance\s*\(")
        er = symex.Exec(); er.rewrite = rw_ref
        eo = symex.Exec(fn_alias={"tricubicValue": "interpolateValue", "tricubicGradient": "interpolateGradient", "roundFloatGE": "roundFloat"},
                        id_alias={"p": "sample", "grad": "outGradient"}, local_alias={"w": "currentNode", "f": "fracPart"})
        eo.rewrite = rw_orc
        r = er.run([], rb)
        o = eo.run([], ob)
        compare("OctreeSdf::" + name, r, o)
        n += len(r)
    return "BoundingBox size / centre / distance / distance+gradient, roundFloat, OctreeSdf::getDistance x2 (start cell, descent, child index, fract, leaf evaluation): identical (%d paths)" % n


GROUPS.append(group_box_and_query)


# ---------------------------------------------------------------------------------------------------------------------
def block_after(text, anchor_regex, nth=0):
    """the `{...}` block that follows the nth match of anchor_regex"""
    m = list(re.finditer(anchor_regex, text))[nth]
    b = text.index("{", m.end() - 1)
    depth, e = 0, b
    while True:
        depth += (text[e] == "{") - (text[e] == "}")
        e += 1
        if depth == 0:
            return text[b:e]


def statement_at(text, anchor_regex, nth=0):
    """the `for (...) {...}` statement that starts at the nth match of anchor_regex (which matches its `for`)"""
    m = list(re.finditer(anchor_regex, text))[nth]
    i = text.index("(", m.start())
    depth, j = 0, i
    while True:
        depth += (text[j] == "(") - (text[j] == ")")
        j += 1
        if depth == 0:
            break
    k = j
    while text[k].isspace():
        k += 1
    if text[k] == "{":
        return text[m.start():m.start() + len(text[m.start():k]) + len(block_after(text[k:], r"\{"))]
    return text[m.start():text.index(";", k) + 1]


def without_conds(paths):
    """paths that differ only in conditions that select nothing (dead locals) collapse to one record"""
    out = []
    for p in paths:
        r = {k: v for k, v in p.items() if k != "conds"}
        if r not in out:
            out.append(r)
    return out


def group_mesh_triangle_data(orc_triangle=None):
    """calculateMeshTriangleData, live branches (TriangleUtils.cpp:20-57 constructor loop, :58-86 edge / vertex pseudonormals,
    :292-420 seam welding, :422-425 vertex normals into the triangle frames)."""
    ref = cparse.preprocess(read(REF + "/src/utils/TriangleUtils.cpp"))
    orc = cparse.preprocess(orc_triangle or read(REPO + "/oracle/orc_triangle.h"))
    orc = orc[orc.index("meshTriangleData("):]
    # the degenerate-triangle machinery of the reference is dead code: its only entry is guarded by `if(false && ...)`
    assert "if(false && triangleArea < zeroAngleThreshold" in ref
    ids_o = {"tris": "triangles", "vertexNormal": "verticesNormal", "openEdges": "edgesNormal", "numIndices": "indices.size()"}

    def rw_ref(x):
        if x == ("mcall", ("id", "indices"), "size", ()):
            return ("id", "indices.size()")
        if x[0] == "mcall" and x[2] == "getSize" and x[1] == ("mcall", ("id", "mesh"), "getBoundingBox", ()):
            return ("bin", "-", ("id", "meshBox.max"), ("id", "meshBox.min"))          # BoundingBox::getSize is pinned as max - min
        if x[0] == "member" and x[1] == ("mcall", ("id", "mesh"), "getBoundingBox", ()):
            return ("id", "meshBox." + x[2])
        return x

    def rw_orc(x):
        if x[0] == "member" and x[1] == ("id", "meshBox"):
            return ("id", "meshBox." + x[2])
        if x == ("un", "*", ("id", "meshBox")):
            return ("id", "meshBox")
        return x

    def run_ref(text, **kw):
        e = symex.Exec(**kw); e.rewrite = rw_ref
        return e.run([], cparse.parse_body("{" + text + "}"))

    def run_orc(text, **kw):
        kw.setdefault("id_alias", ids_o)
        e = symex.Exec(fn_alias={"makeTriangleData": "TriangleData"}, member_alias={"normal": "getTriangleNormal"}, **kw); e.rewrite = rw_orc
        return e.run([], cparse.parse_body("{" + text + "}"))
    # 1. constructor loop (the reference's loop also computes area / degeneracy values that only the dead branch reads)
    r = run_ref(statement_at(ref, r"for \(int i = 0, tIndex = 0;"))
    o = run_orc(statement_at(orc, r"for \(uint32_t i = 0, t = 0;", 0), local_alias={"t": "tIndex"})
    compare("calculateMeshTriangleData / constructor loop", without_conds(r), without_conds(o))
    # 2. edge pseudonormals (insert / pair / erase) and angle-weighted vertex normals
    loop2 = statement_at(ref, r"for\(int i = 0, tIndex = 0;").replace("if(isTriangleDegenerated[tIndex]) continue;", "")
    r = run_ref(loop2)
    o = run_orc(statement_at(orc, r"for \(uint32_t i = 0, t = 0;", 1), local_alias={"t": "tIndex"})
    compare("calculateMeshTriangleData / edge and vertex pseudonormals", r, o)
    n = len(r)
    # 3. seam welding
    r = run_ref(block_after(ref, r"if\(edgesNormal\.size\(\) > 0\)\s*\{")[1:-1])
    o = run_orc(block_after(orc, r"if \(!openEdges\.empty\(\) && meshBox\)\s*\{")[1:-1],
                id_alias=dict(ids_o, start="gridStartPos"),
                local_alias={"vmap": "verticesMap", "nm": "nonManifoldVertices", "set1": "pointSet1", "set2": "pointSet2", "repaired": "newEdgesNormals",
                             "v": "vId", "fill": "index"})
    compare("calculateMeshTriangleData / seam welding", r, o)
    n += len(r)
    # 4. vertex normals into each triangle's frame
    r = run_ref(statement_at(ref, r"for\(int i = 0; i < indices\.size\(\); i\+\+\)"))
    o = run_orc(statement_at(orc, r"for \(uint32_t i = 0; i < numIndices; i\+\+\)"))
    compare("calculateMeshTriangleData / vertex normals to triangle frames", r, o)
    return ("calculateMeshTriangleData: constructor loop, edge pairing + angle weights, seam welding (grid keys, threshold, union-find, "
            "re-pairing, parent sums), final transform identical (%d paths)" % n)


GROUPS.append(group_mesh_triangle_data)


# ---------------------------------------------------------------------------------------------------------------------
SCALARS = {"float": "f4", "int": "i4", "uint32_t": "u4", "uint8_t": "u1", "int32_t": "i4"}


def archive_list(text, fn_name="save"):
    """argument names of the archive(...) / ar(...) calls inside `fn_name` (in order)"""
    m = re.search(r"void " + fn_name + r"\s*\(\s*Archive\s*&\s*(\w+)\s*\)(?:\s*const)?\s*\{", text)
    body = block_after(text[m.start():], r"\{")
    out = []
    for call in re.findall(re.escape(m.group(1)) + r"\s*\(([^;]*)\)\s*;", body):
        out += [a.strip() for a in call.split(",")]
    return out


def member_type(text, name):
    m = re.search(r"^\s*([A-Za-z_][\w:<>, ]*?)\s+" + re.escape(name) + r"\s*(?:=[^;]*)?;", text, re.M)
    assert m, name
    return re.sub(r"\s+", "", m.group(1))


def group_archive():
    """The on-disk field order: archive(...) lists and member types of OctreeSdf.h:222-226, ExactOctreeSdf.h:138-142, the two OctreeNode
    serialize bodies, TriangleUtils.h:50-54, Mesh.h:65-69, UsefullSerializations.h:6-35, the SdfFormat enum — against the tables that
    drive sdflib_amd/serialization.py and against the put() sequence of the C++ writers in include/SdfLib/."""
    import sys
    sys.path.insert(0, REPO)
    from sdflib_amd import serialization as S
    octree_h = cparse.preprocess(read(REF + "/include/SdfLib/OctreeSdf.h"))
    exact_h = cparse.preprocess(read(REF + "/include/SdfLib/ExactOctreeSdf.h"))
    tri_h = cparse.preprocess(read(REF + "/include/SdfLib/utils/TriangleUtils.h"))
    mesh_h = cparse.preprocess(read(REF + "/include/SdfLib/utils/Mesh.h"))
    ser_h = cparse.preprocess(read(REF + "/include/SdfLib/utils/UsefullSerializations.h"))
    fn_h = cparse.preprocess(read(REF + "/include/SdfLib/SdfFunction.h"))
    # glm types: which scalars, in which order
    glm = {}
    for m in re.finditer(r"glm::(\w+)\s*&\s*m\s*\)\s*\{\s*archive\(([^;]*)\);", ser_h):
        glm[m.group(1)] = [a.strip() for a in m.group(2).split(",")]
    assert glm["vec3"] == ["m.x", "m.y", "m.z"] and glm["vec2"] == ["m.x", "m.y"] and glm["mat3"] == ["m[0]", "m[1]", "m[2]"], glm
    flat = {"glm::vec3": "3f4", "glm::vec2": "2f4", "glm::mat3x3": "9f4", "glm::mat3": "9f4"}
    box = mesh_h[mesh_h.index("struct BoundingBox"):]
    assert archive_list(box, "serialize") == ["min", "max"] and member_type(box, "min") == "glm::vec3" and member_type(box, "max") == "glm::vec3"
    flat["BoundingBox"] = "6f4"
    # TriangleData
    td = tri_h[tri_h.index("struct TriangleData"):]
    td = td[:td.index("calculateMeshTriangleData")]
    td_fields = []
    for name in archive_list(td, "serialize"):
        t = member_type(td, name)
        am = re.fullmatch(r"std::array<(glm::\w+),(\d+)>", t)
        if am:
            lay = "%df4" % (int(flat[am.group(1)][:-2]) * int(am.group(2)))
        else:
            lay = flat.get(t) or ("1" + SCALARS[t])
        td_fields.append((name, lay))
    assert td_fields == S.TRIANGLE_DATA_FIELDS, (td_fields, S.TRIANGLE_DATA_FIELDS)
    flat["TriangleUtils::TriangleData"] = "%df4" % sum(int(l[:-2]) for _, l in td_fields)
    # node types
    onode = octree_h[octree_h.index("struct OctreeNode"):]
    assert archive_list(onode, "serialize") == ["childrenIndex"] and re.search(r"uint32_t childrenIndex;", onode)
    enode = exact_h[exact_h.index("struct OctreeNode"):]
    assert archive_list(enode, "serialize") == ["childrenIndex", "trianglesArrayIndex"]
    assert re.search(r"uint32_t childrenIndex;", enode) and re.search(r"uint32_t trianglesArrayIndex;", enode)

    def fields(header, node_layout):
        assert archive_list(header, "save") == archive_list(header, "load")
        out = []
        for name in archive_list(header, "save"):
            t = member_type(header, name)
            vm = re.fullmatch(r"std::vector<([\w:]+)>", t)
            if vm:
                e = vm.group(1)
                lay = "vec:" + (node_layout if e == "OctreeNode" else flat.get(e) or ("1" + SCALARS[e]))
            else:
                lay = flat.get(t) or ("1" + SCALARS[t])
            out.append((name, lay))
        return out
    ref_oct = fields(octree_h, "1u4")
    ref_ex = fields(exact_h, "2u4")
    assert ref_oct == [(n, l) for n, l, _ in S.OCTREE_FIELDS], (ref_oct, S.OCTREE_FIELDS)
    assert ref_ex == [(n, l) for n, l, _ in S.EXACT_FIELDS], (ref_ex, S.EXACT_FIELDS)
    # the format tag: SdfFormat enumerators in order, written before the object (SdfFunction.cpp:27-33)
    en = re.search(r"enum SdfFormat\s*\{([^}]*)\}", fn_h).group(1)
    assert [e.strip() for e in en.split(",")] == ["GRID", "OCTREE", "EXACT_OCTREE", "NONE"]
    assert (S.FORMAT_GRID, S.FORMAT_OCTREE, S.FORMAT_EXACT_OCTREE, S.FORMAT_NONE) == (0, 1, 2, 3)
    cpp = cparse.preprocess(read(REF + "/src/sdf/SdfFunction.cpp"))
    save = cpp[cpp.index("SdfFunction::saveToFile"):cpp.index("SdfFunction::loadFromFile")]
    assert re.search(r"SdfFormat::OCTREE\)\s*\{\s*archive\(format\);\s*archive\(\*reinterpret_cast<OctreeSdf\*>\(this\)\);", save)
    assert re.search(r"SdfFormat::EXACT_OCTREE\)\s*\{\s*archive\(format\);\s*archive\(\*reinterpret_cast<ExactOctreeSdf\*>\(this\)\);", save)
    # the C++ writers of the drop-in headers: order and width of every put()
    info_decl = read(REPO + "/include/sdfhip.h")
    info_decl = info_decl[info_decl.index("typedef struct sdfhip_exact_info"):info_decl.index("} sdfhip_exact_info")]
    info_types = {}
    for t, names in re.findall(r"^\s*(int32_t|uint32_t|uint64_t|float|double)\s+([^;]+);", info_decl, re.M):
        for nm in names.split(","):
            info_types[nm.strip().split("[")[0]] = t

    def cpp_sequence(path, member_types, vec_elems):
        text = cparse.preprocess(read(path))
        body = block_after(text[text.index("void writePayload(std::ostream& os) const override"):], r"\{")
        seq = []
        for kind, arg in re.findall(r"detail::(put|putVec)\(os,\s*([^;]*?)\)\s*;", body):
            arg = arg.strip()
            if kind == "put":
                cm = re.fullmatch(r"\((int32_t|uint32_t)\)\s*([\w.]+)", arg)
                if cm:
                    seq.append((cm.group(2), "1" + SCALARS[cm.group(1)]))
                elif arg == "box":
                    assert re.search(r"const float box\[6\] = \{mBox\.min\.x, mBox\.min\.y, mBox\.min\.z, mBox\.max\.x, mBox\.max\.y, mBox\.max\.z\};", body)
                    seq.append(("mBox", "6f4"))
                else:
                    seq.append((arg, "1" + SCALARS[member_types[arg.split(".")[-1]]]))
            else:
                first = arg.split(",")[0].strip()
                key = re.sub(r"reinterpret_cast<const uint32_t\*>\((\w+)\.data\(\)\)", r"\1", first).replace(".data()", "")
                seq.append((key, "vec:" + vec_elems[key]))
        return seq
    ours_oct = cpp_sequence(REPO + "/include/SdfLib/OctreeSdf.h", {"mValueRange": "float", "mMinBorderValue": "float"}, {"mOctreeData": "1u4"})
    assert ours_oct == ref_oct, (ours_oct, ref_oct)
    ours_ex = cpp_sequence(REPO + "/include/SdfLib/ExactOctreeSdf.h", info_types, {"nodes": "2u4", "sets": "1u4", "masks": "1u1", "td": "37f4"})
    key_of = {"mInfo." + k: n for n, _, k in S.EXACT_FIELDS}
    key_of.update({"mBox": "mBox", "nodes": "mOctreeData", "sets": "mTrianglesSets", "masks": "mTrianglesMasks", "td": "mTrianglesData"})
    assert [(key_of[k], l) for k, l in ours_ex] == ref_ex, (ours_ex, ref_ex)
    return ".bin layout: SdfFormat tag, %d OctreeSdf fields, %d ExactOctreeSdf fields, 8 TriangleData fields (37 floats), node words: names, order and scalar types identical in serialization.py and the C++ writers" % (len(ref_oct), len(ref_ex))


GROUPS.append(group_archive)


# ---------------------------------------------------------------------------------------------------------------------
def group_exact_bits(orc_exact=None):
    """ExactOctreeSdf's bit-packed triangle sets: the writer (ExactOctreeSdfDepthFirst.h:261-283) and the four reader expressions of
    ExactOctreeSdf::getDistance (ExactOctreeSdf.cpp:80-81, 125-126 and their gradient twins), roundFloat (:33-36, '>' not '>=')."""
    hdr = cparse.preprocess(read(REF + "/include/SdfLib/ExactOctreeSdfDepthFirst.h"))
    cpp = cparse.preprocess(read(REF + "/src/sdf/ExactOctreeSdf.cpp"))
    orc = orc_exact or read(REPO + "/oracle/orc_exact.h")
    # ---- writer
    blk = block_after(hdr, r"if\(node\.depth == tContext\.bitEncodingStartDepth\)\s*\{")
    cut = blk.index("tContext.maxTrianglesEncodedInLeafs = ")          # (a statistic the oracle keeps elsewhere)
    ref_w = blk[:cut] + "}"
    orc_p = orc.replace("out.nodeHasTriIdx[nodeIndex] = 1;", "")          # oracle-only bookkeeping ("this node's index was written")
    assert orc_p != orc
    op, ob = fn(orc_p, r"void emitSet\s*\(")

    def rw_ref(x):
        if x[0] == "member" and x[1] == ("id", "tContext"):
            return ("id", x[2])
        if x == ("mcall", ("id", "nodeTriangles"), "size", ()):
            return ("id", "numTriangles@list")
        return x

    def rw_orc(x):
        if x[0] == "member" and x[1] == ("id", "out"):
            return ("id", {"sets": "outputTrianglesSets", "bitsPerIndex": "bitsPerIndex", "nodes": "nodes"}[x[2]])
        if x == ("mcall", ("id", "nodeTriangles"), "size", ()):
            return ("id", "numTriangles@list")
        if x[0] == "mcall" and x[2] == "resize" and len(x[3]) == 2 and x[3][1] == ("lit", "uint", 0):
            return ("mcall", x[1], "resize", x[3][:1])                       # resize(n, 0u) == resize(n): value-initialised words
        if x[0] == "cast" and x[1] == "uint32_t" and x[2] in (("id", "numTriangles@list"), ("mcall", ("id", "outputTrianglesSets"), "size", ())):
            return x[2]                                                      # (uint32_t)vector.size(): the reference converts implicitly
        if x[0] == "index" and x[1] == ("id", "nodes"):
            return ("member", ("id", "octreeNode"), "trianglesArrayIndex")  # the node's second word
        return x
    er = symex.Exec(); er.rewrite = rw_ref
    eo = symex.Exec(id_alias={"list": "nodeTriangles"}, local_alias={"at": "arrayStartIndex", "bIdx": "bIdx"}); eo.rewrite = rw_orc
    r = er.run([], cparse.parse_body(ref_w))
    o = eo.run([], ob)

    def strip(paths):          # `octreeNode->x = v` stores through a pointer on one side, names the node's word on the other: compare target names only
        out = []
        for p in paths:
            ev = tuple(("store", ("member", ("id", "octreeNode"), "trianglesArrayIndex"), e[2]) if e[0] == "store" and symex.show(e[1]).endswith("trianglesArrayIndex") else e for e in p["events"])
            out.append(dict(p, events=ev))
        return out
    compare("bit-packed set writer", strip(r), strip(o))
    # ---- readers
    exprs = re.findall(r"=\s*(\(\(mTrianglesSets\[(\w+) \+ idx\] << bit\) >> \(32-mBitsPerIndex\)\)\s*\|\s*static_cast<uint32_t>\(static_cast<uint64_t>\(mTrianglesSets\[\w+ \+ idx \+ 1\]\) >> \(64 - \(bit \+ mBitsPerIndex\)\)\))", cpp)
    assert len(exprs) == 4, len(exprs)
    assert len(re.findall(r"uint32_t idx = bIdx >> 5;\s*uint32_t bit = bIdx & 0b0011111;", cpp)) == 4
    op, ob = fn(orc, r"uint32_t unpackIndex\s*\(")
    want = symex.Exec().run(op, ob)
    assert len(want) == 1
    for text, base in exprs:
        body = cparse.parse_body("{ uint32_t idx = bIdx >> 5; uint32_t bit = bIdx & 0b0011111; return %s; }" % text)

        def rw(x, base=base):
            if x[0] == "index" and x[1] == ("id", "mTrianglesSets") and x[2][0] == "bin" and x[2][1] == "+":
                inner = x[2]
                if inner[2] == ("id", base):
                    return ("index", ("arg", 0), inner[3])
                if inner[2][0] == "bin" and inner[2][2] == ("id", base):        # (base + idx) + 1
                    return ("index", ("arg", 0), ("bin", "+", inner[2][3], inner[3]))
            return x
        e = symex.Exec(id_alias={"bIdx": "$1", "mBitsPerIndex": "$2"}); e.rewrite = rw
        got = e.run([], body)

        def argify(x):
            if isinstance(x, tuple):
                if x == ("id", "$1"):
                    return ("arg", 1)
                if x == ("id", "$2"):
                    return ("arg", 2)
                return tuple(argify(y) for y in x)
            return x
        compare("bit-packed set reader", [dict(p, end=argify(p["end"])) for p in got], want)
    compare("ExactOctreeSdf roundFloat", symex.Exec().run(*fn(cpp, r"inline uint32_t roundFloat\s*\(")), symex.Exec().run(*fn(orc.replace("? 1u : 0u", "? 1 : 0"), r"uint32_t roundFloatGT\s*\(")))
    return "ExactOctreeSdf bit-packed sets: writer (word count, MSB-first shifts, the spill into the next word), 4 reader expressions, roundFloat ('>'): identical"


GROUPS.append(group_exact_bits)


# ---------------------------------------------------------------------------------------------------------------------
def snippet(text, start_regex, end_regex):
    a = re.search(start_regex, text); assert a, start_regex
    b = re.search(end_regex, text[a.start():]); assert b, end_regex
    return text[a.start():a.start() + b.end()]


def strip_at(x):
    if isinstance(x, tuple):
        if x and x[0] == "at":
            return strip_at(x[2])
        return tuple(strip_at(y) for y in x)
    return x


def group_octree_setup(orc_octree=None, orc_tricubic=None):
    """The float side of OctreeSdf's set-up and of a NO_CONTINUITY node: the box made a cube and the start-grid cell size
    (OctreeSdf.cpp:40-50), root nodes (OctreeSdfDepthFirst.h:115-125), the start-cell index (:406-410), a node's fit size, rule test, child
    centres and value range (:205-225, 349-361), calculatePointValues and the derivative scaling of calculateCoefficients
    (InterpolationMethods.h:273-312)."""
    cpp = cparse.preprocess(read(REF + "/src/sdf/OctreeSdf.cpp"))
    df = cparse.preprocess(read(REF + "/src/sdf/OctreeSdfDepthFirst.h"))
    im = cparse.preprocess(read(REF + "/include/SdfLib/InterpolationMethods.h"))
    orc = cparse.preprocess(orc_octree or read(REPO + "/oracle/orc_octree.h"))
    tri = cparse.preprocess(orc_tricubic or read(REPO + "/oracle/orc_tricubic.h"))
    members = {"box": "mBox", "startGridSize": "mStartGridSize", "startGridXY": "mStartGridXY", "startGridCellSize": "mStartGridCellSize", "maxDepth": "mMaxDepth"}

    def rw_orc(x):
        if x[0] == "member" and x[1] == ("id", "out") and x[2] in members:
            return ("id", members[x[2]])
        if x[0] == "mcall" and x[2] in ("size", "center") and x[3] == ():
            return ("mcall", x[1], "getSize" if x[2] == "size" else "getCenter", ())
        if x[0] == "cast" and x[1] == "float" and "pow" in symex.show(x[2]):
            return x[2]                      # std::pow(float, unsigned) is a double in C++; 0.5^k and the products with it are exact either way
        return x

    def run(text, rewrite=None, **kw):
        e = symex.Exec(**kw); e.rewrite = rewrite
        return e.run([], cparse.parse_body("{" + text + "}"))
    # 1. the cube
    r = run(snippet(cpp, r"const glm::vec3 bbSize = box\.getSize\(\);", r"mStartGridCellSize = [^;]*;"))
    o = run(snippet(orc, r"const V3 bs = inBox\.size\(\);", r"out\.startGridCellSize = [^;]*;"), rw_orc, id_alias={"inBox": "box"})
    compare("buildOctree: cube and start-grid cell size", r, o)
    # 2. root nodes: size, first centre, centre of node (i, j, k)
    r = run(snippet(df, r"float newSize = 0\.5f \* mBox\.getSize\(\)\.x", r"const glm::vec3 startCenter = [^;]*;") + " return startCenter + glm::vec3(i, j, k) * 2.0f * newSize;")
    o = run(snippet(orc, r"const float newSize = \(float\)\(0\.5f \* out\.box\.size\(\)\.x", r"const V3 startCenter = [^;]*;") + " return startCenter + V3{(float)i, (float)j, (float)k} * 2.0f * newSize;", rw_orc)

    def uncast(x):       # glm::vec3(i, j, k) converts its unsigned arguments; the oracle spells the conversion out
        if isinstance(x, tuple):
            if x and x[0] == "cast" and x[1] == "float" and x[2][0] == "id":
                return x[2]
            return tuple(uncast(y) for y in x)
        return x
    compare("initOctree: root nodes", r, [dict(p, end=uncast(p["end"])) for p in o])
    # 3. start-cell index of a node
    r = run(snippet(df, r"glm::ivec3 startArrayPos = glm::floor\(\(node\.center - mBox\.min\) / mStartGridCellSize\);", r"node\.nodeIndex = [^;]*;"))
    o = run(snippet(orc, r"V3 f = \(n\.center - out\.box\.min\) / out\.startGridCellSize;", r"return \(uint32_t\)\([^;]*;"), rw_orc, id_alias={"n": "node"})
    want = r[0]["events"][-1][2]
    got = o[0]["end"][1]
    assert got[0] == "cast" and got[1] == "uint32_t", got
    compare("initOctree: start-cell index", want, got[2])
    # 4. a node: the fit's size argument, the rule's comparison, child centres, the value range
    assert re.search(r"calculateCoefficients\(node\.verticesValues, 2\.0f \* node\.size,", df) and len(re.findall(r"tricubicFit\(node\.vv, 2\.0f \* node\.size, coeff\)", orc)) == 2
    assert re.search(r"generateTerminalNodes = value < tContext\.sqTerminationThreshold;", df) and re.search(r"sqTerminationThreshold = terminationRuleParams\[0\] \* terminationRuleParams\[0\];", df)
    assert re.search(r"terminal = ruleValue\(rule, coeff, mid, param1\) < sqThreshold;", orc) and re.search(r"sqThreshold = p0 \* p0;", orc)
    signs = re.findall(r"node\.center \+ glm::vec3\((-?)newSize, (-?)newSize, (-?)newSize\), newSize", df)
    assert [tuple(s == "" for s in t) for t in signs] == [((c & 1) != 0, (c & 2) != 0, (c & 4) != 0) for c in range(8)], signs
    assert re.search(r"const float newSize = 0\.5f \* node\.size;", df) and re.search(r"const float ns = 0\.5f \* node\.size;", orc)
    assert re.search(r"ch\.center = node\.center \+ V3\{\(c & 1\) \? ns : -ns, \(c & 2\) \? ns : -ns, \(c & 4\) \? ns : -ns\};", orc)
    r = run(snippet(df, r"for\(uint32_t i=0; i < 8; i\+\+\)\s*\{\s*tContext\.valueRange", r"\}"), lambda x: ("id", "valueRange") if x == ("member", ("id", "tContext"), "valueRange") else (("id", "vv") if x == ("member", ("id", "node"), "verticesValues") else x))
    o = run("for (int i = 0; i < 8; i++) valueRange = gmax(valueRange, std::fabs(node.vv[i][0]));", lambda x: ("id", "vv") if x == ("member", ("id", "node"), "vv") else x)
    assert "for (int i = 0; i < 8; i++) valueRange = gmax(valueRange, std::fabs(node.vv[i][0]));" in orc
    compare("processNode: value range", strip_at(r), strip_at(o))
    # 5. calculatePointValues
    tc = im[im.index("struct TriCubicInterpolation"):]
    # (the gradient local is an out-argument of the distance routine: left undeclared on both sides so that it is one opaque name)
    ref_pv = tc.replace("glm::vec3 gradient;", "", 1)
    orc_pv = orc.replace("V3 g;", "", 1)
    assert ref_pv != tc and orc_pv != orc
    rp, rb = fn(ref_pv, r"inline static void calculatePointValues\s*\(")
    op, ob = fn(orc_pv, r"static inline void pointValues\s*\(")

    def rw_pv_ref(x):
        if x[0] == "mcall" and x[1] == ("arg", 2) and x[2] in ("getIndices", "getVertices"):
            return ("member", ("arg", 2), "indices" if x[2] == "getIndices" else "vertices")
        return x
    er = symex.Exec(); er.rewrite = rw_pv_ref
    r = er.run(rp, rb)
    # oracle argument order: (p, t, mesh, td, out) - the same positions as the reference's (point, index, mesh, trianglesData, outValues)
    o = symex.Exec(fn_alias={"signedDistPointTriangleGrad": "getSignedDistPointAndTriangle"}, id_alias={"g": "gradient"}).run(op, ob)
    compare("calculatePointValues", strip_at(r), strip_at(o))
    # 6. derivative scaling in calculateCoefficients: slot q of vertex v is multiplied by nodeSize^order
    scal = snippet(im[im.index("inline static void calculateCoefficients"):], r"for\(uint32_t i=0; i < 8; i\+\+\)", r"inValues\[i\]\[7\] \*= sqNodeSize \* nodeSize;\s*\}")
    r = run(scal)
    refmap = {}
    for e in r[0]["events"]:
        assert e[0] == "store"
        tgt, val = strip_at(e[1]), strip_at(e[2])
        refmap[(tgt[1][2][2], tgt[2][2])] = val                     # inValues[v][q]
    o = run(snippet(tri, r"float s\[64\];", r"s\[8 \* v \+ 7\] = [^;]*;\s*\}") + " return s;")
    got = dict(o[0]["end"][1][1:])
    assert len(refmap) == 56 and len(got) == 64
    for (v, q), val in refmap.items():
        want_val = symex.first_difference(val, got[8 * v + q])
        # reference: inValues[v][q] * k   (in place);  oracle: in[v][q] * k
        a = symex.show(val).replace("inValues", "in"); b = symex.show(got[8 * v + q])
        assert a == b, (v, q, a, b)
    for v in range(8):
        assert symex.show(got[8 * v]) == "in[%d][0]" % v
    # 7. VHQueries::calculateVerticesInfo (TrianglesInfluence.h:951-996): where a sample is taken, the parent-interpolation arguments, the cache key
    ti = cparse.preprocess(read(REF + "/include/SdfLib/TrianglesInfluence.h"))
    vh = ti[ti.index("struct VHQueries"):]

    def expr(text, **kw):
        return run("return %s;" % text, **kw)[0]["end"]
    assert "inPoints[i] = nodeCenter + pointsRelPos[i] * nodeHalfSize;" in vh and "const V3 p = center + rel[i] * half;" in orc
    compare("sample position", expr("nodeCenter + pointsRelPos[i] * nodeHalfSize"), expr("center + rel[i] * half", id_alias={"center": "nodeCenter", "rel": "pointsRelPos", "half": "nodeHalfSize"}))
    assert "InterpolationMethod::interpolateVertexValues(interpolationCoeff, 0.5f * pointsRelPos[i] + 0.5f, 2.0f * nodeHalfSize, outPointsValues[i]);" in vh
    assert "const glm::uvec3 pointId = glm::uvec3(glm::round((inPoints[i] - minPoint) * coordToId));" in vh and "const V3 q = (p - minPoint) * coordToId;" in orc
    assert "coordToId = glm::vec3(static_cast<float>((1 << maxDepth)) / box.getSize());" in vh
    assert re.search(r"const float s = \(float\)\(1 << maxDepth\);\s*const V3 sz = box\.size\(\);\s*coordToId = V3\{s / sz\.x, s / sz\.y, s / sz\.z\};", orc)
    # (the 32^3 direct-mapped cache is OFF in canonical mode; its key and slot are checked as text: mask (1 << 5) - 1, shifts 2 * 5 and 5)
    assert "const uint32_t cacheId = ((pointId.z & CACHE_AXIS_MASK) << (2*CACHE_AXIS_POWER)) |" in vh and re.search(r"CACHE_AXIS_MASK = \(1 << CACHE_AXIS_POWER\) - 1;", ti)
    assert re.search(r"CACHE_AXIS_POWER = 5;", ti) and re.search(r"CACHE_AXIS_MASK = [^;]*;", ti) and "((iz & 31u) << 10) | ((iy & 31u) << 5) | (ix & 31u)" in orc
    return "OctreeSdf set-up: cube / cell size, root nodes, start-cell index, fit size, rule test, child centres, value range, calculatePointValues, derivative scaling, sample positions: identical"


GROUPS.append(group_octree_setup)


# ---------------------------------------------------------------------------------------------------------------------
def group_continuity_floats(orc_cont=None):
    """The floating-point decisions of the CONTINUITY builder (OctreeSdfBreadthFirstNoDelay.h): Iter 2's "can this shared mid-point be taken
    from the coarser neighbour's interpolant" test (:486-503), the post-pass's twin (:924-939), the interpolation arguments of
    calculateVerticesInfo (TrianglesInfluence.h:966-969) and the child size (:941)."""
    ref = cparse.preprocess(read(REF + "/src/sdf/OctreeSdfBreadthFirstNoDelay.h"))
    orc = cparse.preprocess(orc_cont or read(REPO + "/oracle/orc_continuity.h"))
    # the NoDelay builder is the one the exporter runs (OctreeSdf.cpp:66-73)
    live = ref[ref.index("void OctreeSdf::initOctreeWithContinuityNoDelay"):]

    def rw_ref(x):
        if x[0] == "member" and x[1] == ("id", "node"):
            return ("id", {"interpolationCoeff": "coeff", "midPointsValues": "mid", "size": "size"}.get(x[2], x[2]))
        return x

    def rw_orc(x):
        if x[0] == "member" and x[1] in (("id", "node"), ("id", "n")):
            return ("id", x[2])
        if x[0] == "member" and x[1] == ("id", "st") and x[2] == "midRel":
            return ("id", "nodeSamplePoints")
        return x

    def run(text, rw, **kw):
        # ONE iteration with a free `i` (the 19 iterations are the same statement; unrolled they would fork 3^19 ways)
        body = text[text.index("{", text.index(")")):]            # the for statement's own block
        e = symex.Exec(fn_alias={"tricubicValue": "interpolateValue", "tricubicVertexValues": "interpolateVertexValues"}, **kw); e.rewrite = rw
        return e.run([], cparse.parse_body(body))
    # Iter 2
    r = run(snippet(live, r"for\(uint32_t i=0; i < 19; i\+\+\)\s*\{\s*if\(samplesMask & \(1 << \(18-i\)\)\)", r"\n\s*\}\s*\}\s*\}"), rw_ref, funcs={"pow2": [(cparse.parse_params("float a"), cparse.parse_body("{ return a * a; }"))]},
            id_alias={"sqTerminationThreshold": "sqThr"})
    o = run(snippet(orc, r"for \(int i = 0; i < 19; i\+\+\) \{\s*if \(!\(samplesMask & \(1u << \(18 - i\)\)\)\) continue;", r"else tricubicVertexValues\(node\.coeff, f, 2\.0f \* node\.size, node\.mid\[i\]\);\s*\}").replace("continue;", "return;"), rw_orc)
    compare("CONTINUITY Iter 2: interpolate-or-subdivide test", _masks_as_sets(r), _masks_as_sets(o))
    # post-pass
    r2 = run(snippet(live, r"for\(uint32_t i=0; i < 19; i\+\+\)\s*\{\s*if \(\(samplesMask & \(1 << \(18-i\)\)\) == 0\)", r"else if\(recycleMidPointsValues\)\s*\{[^}]*\}\s*\}"), rw_ref,
             funcs={"pow2": [(cparse.parse_params("float a"), cparse.parse_body("{ return a * a; }"))]}, id_alias={"sqTerminationThreshold": "sqThr", "recycleMidPointsValues": "recycleMid"})
    o2 = run(snippet(orc, r"for \(int i = 0; i < 19; i\+\+\) \{\s*const V3 f = 0\.5f \* st\.midRel\[i\] \+ 0\.5f;\s*if \(\(samplesMask & \(1u << \(18 - i\)\)\) == 0\)", r"else if \(recycleMid\) tricubicVertexValues\([^;]*;\s*\}"), rw_orc)
    compare("CONTINUITY post-pass: interpolate test", _masks_as_sets(r2), _masks_as_sets(o2))
    assert "0.5f * pointsRelPos[i] + 0.5f, 2.0f * nodeHalfSize" in cparse.preprocess(read(REF + "/include/SdfLib/TrianglesInfluence.h"))
    assert "tricubicVertexValues(n.coeff, 0.5f * st.midRel[i] + 0.5f, 2.0f * n.size, n.mid[i]);" in orc and "sample(n.center + st.midRel[i] * n.size, n.mid[i], n.midInfo[i]);" in orc
    assert len(re.findall(r"const float newSize = 0\.5f \* node\.size;", live)) >= 2 and "0.5f * node.size" in orc
    return "CONTINUITY builder: Iter-2 and post-pass interpolate-or-subdivide tests (value, squared error against the squared threshold, strictness), interpolation arguments: identical (%d + %d paths)" % (len(r), len(o2))


def _masks_as_sets(paths):
    """one loop iteration's paths, comparable across `if (bit) {...}` and `if (!bit) continue; ...`: a negated condition is the condition with
    the other outcome, an early return is the end of the body, and the integer literal of the bit test has no kind (1 << k vs 1u << k)"""
    def unkind(x):
        if isinstance(x, tuple):
            if x and x[0] == "lit" and x[1] in ("int", "uint"):
                return ("lit", "int", x[2])
            return tuple(unkind(y) for y in x)
        return x
    out = []
    for p in paths:
        conds = []
        for c, taken in p["conds"]:
            while isinstance(c, tuple) and c[0] == "un" and c[1] == "!":
                c, taken = c[2], not taken
            conds.append((symex.show(unkind(c)), taken))
        out.append((tuple(conds), tuple(symex.show(unkind(e)) for e in p["events"])))
    return sorted(out)


GROUPS.append(group_continuity_floats)


# ---------------------------------------------------------------------------------------------------------------------
def group_exact_setup(orc_exact=None):
    """ExactOctreeSdf's set-up and per-node float / index expressions: the cube and cell size (ExactOctreeSdf.cpp:13-22), bits per index and
    root nodes (ExactOctreeSdfDepthFirst.h:68-71, 130-140), start-cell index (:506), the brute-force argmin of
    PerNodeRegionTrianglesInfluence::calculateVerticesInfo (TrianglesInfluence.h:712-733), the terminal test and child size (:301, 320)."""
    cpp = cparse.preprocess(read(REF + "/src/sdf/ExactOctreeSdf.cpp"))
    df = cparse.preprocess(read(REF + "/include/SdfLib/ExactOctreeSdfDepthFirst.h"))
    ti = cparse.preprocess(read(REF + "/include/SdfLib/TrianglesInfluence.h"))
    orc = cparse.preprocess(orc_exact or read(REPO + "/oracle/orc_exact.h"))
    members = {"box": "mBox", "startGridSize": "mStartGridSize", "startGridXY": "mStartGridXY", "startGridCellSize": "mStartGridCellSize", "maxDepth": "mMaxDepth",
               "startDepth": "mStartDepth"}

    def rw_orc(x):
        if x[0] == "member" and x[1] == ("id", "out") and x[2] in members:
            return ("id", members[x[2]])
        if x[0] == "mcall" and x[2] in ("size", "center") and x[3] == () and x[1] in (("id", "inBox"), ("id", "box"), ("id", "mBox")):
            return ("mcall", x[1], "getSize" if x[2] == "size" else "getCenter", ())
        if x[0] == "cast" and x[1] == "float" and "pow" in symex.show(x[2]):
            return x[2]
        return x

    def run(text, rewrite=None, **kw):
        e = symex.Exec(**kw); e.rewrite = rewrite
        return e.run([], cparse.parse_body("{" + text + "}"))
    r = run(snippet(cpp, r"const glm::vec3 bbSize = box\.getSize\(\);", r"mStartGridCellSize = [^;]*;"))
    o = run(snippet(orc, r"const V3 bs = inBox\.size\(\);", r"out\.startGridCellSize = [^;]*;"), rw_orc, id_alias={"inBox": "box"})
    compare("ExactOctreeSdf: cube and start-grid cell size", r, o)
    # bits per index: ceil(log2(float(#triangles))) through int32 into uint32
    r = run("return " + re.search(r"uint32_t bitsPerIndex = ([^;]*);", df).group(1) + ";", id_alias={"trianglesData": "T"})
    o = run("return " + re.search(r"out\.bitsPerIndex = ([^;]*);", orc).group(1) + ";", lambda x: ("id", "T") if x == ("member", ("id", "out"), "triangles") else x)
    got = o[0]["end"][1]
    assert got[0] == "cast" and got[1] == "uint32_t", got                      # the reference's declaration converts; the oracle spells it
    want = r[0]["end"][1]
    assert want[0] == "cast" and want[1] == "int32_t" and got[2][0] == "cast" and got[2][1] == "int32_t"
    compare("bits per index", want[2], got[2][2])
    assert re.search(r"uint32_t bitEncodingStartDepth = maxDepth - BIT_ENCODING_DEPTH;", df) and re.search(r"BIT_ENCODING_DEPTH = 2;", read(REF + "/include/SdfLib/ExactOctreeSdf.h")) and "out.bitEncodingStartDepth = depth - 2;" in orc
    # root nodes and the start-cell index: the same statements as OctreeSdf's
    r = run(snippet(df, r"float newSize = 0\.5f \* mBox\.getSize\(\)\.x", r"const glm::vec3 startCenter = [^;]*;") + " return startCenter + glm::vec3(i, j, k) * 2.0f * newSize;")
    o = run(snippet(orc, r"const float newSize = \(float\)\(0\.5f \* out\.box\.size\(\)\.x", r"const V3 startCenter = [^;]*;") + " return startCenter + V3{(float)i, (float)j, (float)k} * 2.0f * newSize;", rw_orc,
            id_alias={"sod": "startOctreeDepth"})

    def uncast(x):
        if isinstance(x, tuple):
            if x and x[0] == "cast" and x[1] == "float" and x[2][0] == "id":
                return x[2]
            return tuple(uncast(y) for y in x)
        return x
    compare("ExactOctreeSdf: root nodes", r, [dict(p, end=uncast(p["end"])) for p in o])
    assert len(re.findall(r"glm::ivec3 startArrayPos = glm::floor\(\(node\.center - mBox\.min\) / mStartGridCellSize\);", df)) >= 1
    assert len(re.findall(r"V3 f = \((?:cn|n)\.center - out\.box\.min\) / out\.startGridCellSize;", orc)) == 2
    # the brute-force nearest triangle of a node's list: first strict minimum in list order
    arg = snippet(ti[ti.index("struct PerNodeRegionTrianglesInfluence"):], r"for\(const uint32_t& t : triangles\)", r"minDistanceToPoint\[i\] = dist;\s*\}\s*\}")
    r = run(arg, lambda x: x, fn_alias={}, id_alias={})
    o = run("uint32_t best = 0; float bd = INFINITY; for (uint32_t t : list) { const float d = sqDistPointTriangle(p, (*tris)[t]); if (d < bd) { best = t; bd = d; } }",
            lambda x: ("id", "trianglesData") if x == ("un", "*", ("id", "tris")) else x, fn_alias={"sqDistPointTriangle": "getSqDistPointAndTriangle"},
            id_alias={"list": "triangles", "p": "inPoints[i]"})
    assert "uint32_t best = 0; float bd = INFINITY;\n        for (uint32_t t : list) { const float d = sqDistPointTriangle(p, (*tris)[t]); if (d < bd) { best = t; bd = d; } }" in orc
    assert "minDistanceToPoint.fill(INFINITY);" in ti

    def conds_only(paths):         # the reference keeps (id, distance) in arrays, the oracle in two locals: compare what is tested and in which order
        return [tuple((symex.show(c).replace("minDistanceToPoint[i]", "<bd@loop1>").replace("inPoints[i]", "inPoints[i]"), t) for c, t in p["conds"]) for p in paths]
    a, b = conds_only(r), conds_only(o)
    assert len(a) == len(b) == 3 and [x for x in a] == [x for x in b], (a, b)
    assert re.search(r"isTerminalNode = nodeTriangles\.size\(\) <= tContext\.minTrianglesPerNode;", df) and "terminal = nodeList.size() <= minTriangles;" in orc
    assert re.search(r"const float newSize = 0\.5f \* node\.size;", df) and "const float ns = 0.5f * n.size;" in orc
    return "ExactOctreeSdf set-up: cube / cell size, bits per index, root nodes, start-cell index, brute-force argmin (strict <, list order), terminal test, child size: identical"


GROUPS.append(group_exact_setup)


# ---------------------------------------------------------------------------------------------------------------------
def group_exact_query(orc_exact=None):
    """ExactOctreeSdf::getDistance x2 (ExactOctreeSdf.cpp:38-320), the floating-point side: start cell and the outside-the-grid value, the descent
    (child index from roundFloat, fract(2 f)), the `dist < minDist` scan and the signed distance / gradient of the winner.  (The mask decoding
    between them is integer bookkeeping: compared only through GPU-vs-oracle equality of every query, not here.)"""
    cpp = cparse.preprocess(read(REF + "/src/sdf/ExactOctreeSdf.cpp"))
    orc = cparse.preprocess(orc_exact or read(REPO + "/oracle/orc_exact.h"))
    members = {"box": "mBox", "startGridSize": "mStartGridSize", "startGridXY": "mStartGridXY", "startGridCellSize": "mStartGridCellSize"}

    def rw_orc(x):
        if x[0] == "member" and x[1] == ("id", "o") and x[2] in members:
            return ("id", members[x[2]])
        if x[0] == "call" and x[1] == "boxDistance" and x[2][0] == ("id", "mBox"):
            return ("mcall", ("id", "mBox"), "getDistance", x[2][1:])
        if x[0] == "mcall" and x[1] == ("id", "mBox") and x[2] == "size":
            return ("mcall", x[1], "getSize", ())
        return x

    def run(text, rewrite=None, **kw):
        e = symex.Exec(**kw); e.rewrite = rewrite
        return e.run([], cparse.parse_body("{" + text + "}"))
    n = 0
    for k in range(2):
        fnc = cpp[[m.start() for m in re.finditer(r"float ExactOctreeSdf::getDistance\(", cpp)][k]:]
        head = snippet(fnc, r"glm::vec3 fracPart = ", r"return mBox\.getDistance\(sample\) \+ glm::sqrt\(3\.0f\) \* mBox\.getSize\(\)\.x;\s*\}")
        r = run(head + " return startArrayPos.z * mStartGridXY + startArrayPos.y * mStartGridSize + startArrayPos.x;")
        o = run(snippet(orc[orc.index("static inline float exactDistance"):], r"V3 f = \(p - o\.box\.min\)", r"return boxDistance\(o\.box, p\) \+ std::sqrt\(3\.0f\) \* o\.box\.size\(\)\.x;")
                + " return (uint32_t)(iz * o.startGridXY + iy * o.startGridSize + ix);", rw_orc, id_alias={"p": "sample"})
        # the node index: the oracle converts to uint32_t explicitly where the reference indexes the array with an int expression
        o = [dict(p, end=("ret", p["end"][1][2]) if (p["end"][1][0] == "cast" and p["end"][1][1] == "uint32_t") else p["end"]) for p in o]
        compare("ExactOctreeSdf::getDistance #%d: start cell / outside value" % k, r, o)
        n += len(r)
        # descent step, three times in the function (before the encoding depth, the step onto the first masked level, below it)
        assert len(re.findall(r"const uint32_t childIdx = \(roundFloat\(fracPart\.z\) << 2\) \+\s*\(roundFloat\(fracPart\.y\) << 1\) \+\s*roundFloat\(fracPart\.x\);\s*currentNode = &mOctreeData\[currentNode->getChildrenIndex\(\) \+ childIdx\];\s*fracPart = glm::fract\(2\.0f \* fracPart\);", fnc[:fnc.index("float ExactOctreeSdf", 10) if "float ExactOctreeSdf" in fnc[10:] else len(fnc)])) == 3
        # the scan: strict `<`, first minimum wins; two scans (leaf above the encoding depth, masked leaf)
        body = fnc[:fnc.index("float ExactOctreeSdf", 10)] if "float ExactOctreeSdf" in fnc[10:] else fnc
        assert len(re.findall(r"const float dist = TriangleUtils::getSqDistPointAndTriangle\(sample, mTrianglesData\[tIndex\]\);\s*if\(dist < minDist\)\s*\{\s*minIndex = tIndex;\s*minDist = dist;\s*\}", body)) == 2
        if k == 0:
            assert len(re.findall(r"return TriangleUtils::getSignedDistPointAndTriangle\(sample, mTrianglesData\[minIndex\]\);", body)) == 2
        else:
            assert len(re.findall(r"return TriangleUtils::getSignedDistPointAndTriangle\(sample, mTrianglesData\[minIndex\], outGradient\);", body)) == 2
    oq = orc[orc.index("static inline float exactDistance"):]
    assert "const uint32_t c = (roundFloatGT(f.z) << 2) + (roundFloatGT(f.y) << 1) + roundFloatGT(f.x);" in oq and "f = gfract(2.0f * f);" in oq
    assert "float minDist = INFINITY; uint32_t minIndex = 0;" in oq and len(re.findall(r"const float d = sqDistPointTriangle\(p, o\.triangles\[ti\]\);\s*if \(d < minDist\) \{ minIndex = ti; minDist = d; \}", oq)) == 2
    assert "if (grad) return signedDistPointTriangleGradLocal(p, o.triangles[minIndex], *grad);" in oq and "return signedDistPointTriangle(p, o.triangles[minIndex]);" in oq
    assert re.search(r"float minDist = INFINITY;\s*uint32_t minIndex = 0;", cpp)
    compare("descent step", run("return (roundFloat(fracPart.z) << 2) + (roundFloat(fracPart.y) << 1) + roundFloat(fracPart.x);"),
            run("return (roundFloatGT(f.z) << 2) + (roundFloatGT(f.y) << 1) + roundFloatGT(f.x);", fn_alias={"roundFloatGT": "roundFloat"}, id_alias={"f": "fracPart"}))
    return "ExactOctreeSdf::getDistance x2: start cell, outside-the-grid value (box distance + sqrt(3) size), descent step, strict `<` scans, winner's signed distance / gradient: identical (%d paths + text)" % n


GROUPS.append(group_exact_query)
