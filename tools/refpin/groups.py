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
    if x[0] == "cast" and isinstance(x[2], tuple) and x[2][0] == "un" and x[2][1] == "&":
        return ("coeffs", x[2][2])
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
