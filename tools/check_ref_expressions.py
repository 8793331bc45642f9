"""Dev-time pinning of the oracle against the reference's literal expressions.

Part 1 (below): the generated straight-line code of InterpolationMethods.h and the rule / stencil / mask tables.
Part 2 (tools/refpin): a C++-subset parser + symbolic path executor compares, function by function, the reference's
hand-written geometry / GJK / culling / border / query code and its archive(...) lists with oracle/orc_*.h and the .bin writers.

Parses (never copies) the generated straight-line code in /root/reference/include/SdfLib/
InterpolationMethods.h and checks, term by term, that the rule-based restatement in oracle/orc_tricubic.h
(fit matrix = H(x)H(x)H in (vertex, slot) order; value / derivative sums in ascending coefficient order with
left-to-right power products) generates exactly the same expressions.  Skips when the reference is absent.
Run:  python tools/check_ref_expressions.py
"""
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
REF = "/root/reference/include/SdfLib/InterpolationMethods.h"


def parse_sum(expr):
    """'a + b + c' -> list of terms; each term = (int factor or None, coeff index, [axis factors])."""
    terms = []
    for t in [s.strip() for s in expr.replace("\n", " ").split(" + ")]:
        if t in ("0.0f", ""):
            terms.append(("zero",)); continue
        toks = [x.strip() for x in t.split("*")]
        fac = None
        if re.fullmatch(r"-?\d+", toks[0]):
            fac = int(toks[0]); toks = toks[1:]
        m = re.fullmatch(r"values\[(\d+)\]", toks[0])
        assert m, t
        axes = []
        for x in toks[1:]:
            mm = re.fullmatch(r"fracPart\[(\d)\]", x); assert mm, t
            axes.append(int(mm.group(1)))
        terms.append((fac, int(m.group(1)), axes))
    return terms


def gen_value():
    out = [("zero",)]
    for n in range(64):
        i, j, k = n & 3, (n >> 2) & 3, n >> 4
        out.append((None, n, [0] * i + [1] * j + [2] * k))
    return out


def gen_deriv(ex, ey, ez):
    out = []
    for n in range(64):
        i, j, k = n & 3, (n >> 2) & 3, n >> 4
        fac = (i if ex else 1) * (j if ey else 1) * (k if ez else 1)
        if fac == 0:
            continue
        out.append((fac, n, [0] * (i - ex) + [1] * (j - ey) + [2] * (k - ez)))
    return out


def main():
    if not os.path.exists(REF):
        print("reference not present: skipped"); return 0
    src = open(REF).read()
    live = src[src.index("struct TriCubicInterpolation"):]
    # ---- fit matrix
    from oracle import pyoracle as O
    M = O.fit_matrix()
    rows = {}
    for m in re.finditer(r"outCoeff\[(\d+)\] = (.*?);", live):
        terms = re.findall(r"(-?\d+) \* inValues\[(\d)\]\[(\d)\]", m.group(2))
        rows[int(m.group(1))] = [(int(c), int(v), int(q)) for c, v, q in terms]
    assert len(rows) == 64
    for r in range(64):
        mine = [(int(M[r, 8 * v + q]), v, q) for v in range(8) for q in range(8) if M[r, 8 * v + q] != 0]
        assert mine == rows[r], ("fit row", r)
    print("fit matrix: 64 rows identical (coefficients and term order)")
    # ---- scalar interpolateValue (ENOKI off)
    scalar = live[live.index("#else"):live.index("#endif")]
    body = re.search(r"return (.*?);", scalar, re.S).group(1)
    assert parse_sum(body) == gen_value()
    print("interpolateValue: identical")
    # ---- gradient
    ga = live.index("inline static glm::vec3 interpolateGradient")
    gb = live.index("inline static void interpolateVertexValues", ga)
    g = live[ga:gb]
    body = re.search(r"return glm::vec3\((.*)\);", g, re.S).group(1)
    # split the three components at top-level commas (no nested parentheses in the body)
    comps = [c.strip() for c in body.split(",")]
    assert len(comps) == 3
    for comp, d in zip(comps, [(1, 0, 0), (0, 1, 0), (0, 0, 1)]):
        assert parse_sum(comp) == gen_deriv(*d), d
    print("interpolateGradient: identical")
    # ---- vertex values
    vv = live[gb:]
    exprs = re.findall(r"outValues\[(\d)\] = (.*?);", vv, re.S)
    want = {1: ((1, 0, 0), "nodeSize"), 2: ((0, 1, 0), "nodeSize"), 3: ((0, 0, 1), "nodeSize"),
            4: ((1, 1, 0), "sqNodeSize"), 5: ((1, 0, 1), "sqNodeSize"), 6: ((0, 1, 1), "sqNodeSize"),
            7: ((1, 1, 1), "(sqNodeSize * nodeSize)")}
    seen = set()
    for idx, e in exprs:
        idx = int(idx); e = e.strip()
        if idx == 0:
            assert parse_sum(e) == gen_value(); seen.add(0); continue
        m = re.fullmatch(r"\((.*)\) / (.*)", e, re.S)
        assert m, e[:80]
        assert m.group(2).strip() == want[idx][1], (idx, m.group(2))
        assert parse_sum(m.group(1)) == gen_deriv(*want[idx][0]), idx
        seen.add(idx)
    assert seen == set(range(8))
    print("interpolateVertexValues: 8 expressions identical")
    # ---- CONTINUITY builder: the 24-entry neighbour-mask table is derived from the stencil geometry in the oracle
    nd = open("/root/reference/src/sdf/OctreeSdfBreadthFirstNoDelay.h").read()
    blk = nd[nd.index("neigbourMasks ="):]
    blk = blk[:blk.index("};")]
    lits = [int(x, 2) for x in re.findall(r"0b([01]{20})", blk)]
    assert len(lits) == 24
    import ctypes as C
    import numpy as np
    L = O.lib()
    L.orc_neighbour_masks.restype = None; L.orc_neighbour_masks.argtypes = [C.c_void_p]
    mine = np.zeros(24, dtype=np.uint32)
    L.orc_neighbour_masks(mine.ctypes.data_as(C.c_void_p))
    assert [int(x) for x in mine] == lits, (list(mine), lits)
    print("neighbour masks: 24 entries identical")
    # ---- NO_CONTINUITY builder: the 19 sample offsets and the hand-down of the 27-point stencil to the 8 children
    df = open("/root/reference/src/sdf/OctreeSdfDepthFirst.h").read()
    blk = df[df.index("nodeSamplePoints ="):]
    blk = blk[:blk.index("};")]
    pts = []
    for mm in re.finditer(r"glm::vec3\(([^)]*)\)", blk):
        a = [float(x.strip().rstrip("f")) for x in mm.group(1).split(",")]
        pts.append(a * 3 if len(a) == 1 else a)
    assert len(pts) == 19
    L.orc_stencil_tables.restype = None; L.orc_stencil_tables.argtypes = [C.c_void_p] * 3
    rel = np.zeros((19, 3), np.float32); src = np.zeros((8, 8), np.int32); wt = np.zeros(19, np.float32)
    L.orc_stencil_tables(rel.ctypes.data_as(C.c_void_p), src.ctypes.data_as(C.c_void_p), wt.ctypes.data_as(C.c_void_p))
    assert rel.tolist() == pts, (rel.tolist(), pts)
    print("nodeSamplePoints: 19 offsets identical")
    body = df[df.index("// Generate new childrens"):]
    blocks = re.findall(r"nodesStack\.push\(NodeInfo\((.*?)\)\);\s*\{(.*?)\n\t*\s*\}", body, re.S)
    seen = {}
    for head, inner in blocks[:8]:
        sg = re.search(r"glm::vec3\((-?)newSize, (-?)newSize, (-?)newSize\)", head)
        c = (0 if sg.group(1) else 1) | ((0 if sg.group(2) else 1) << 1) | ((0 if sg.group(3) else 1) << 2)
        row = [None] * 8
        for j, kind, idx in re.findall(r"child\.verticesValues\[(\d)\] = (midPointsValues|node\.verticesValues)\[(\d+)\]", inner):
            row[int(j)] = int(idx) if kind == "midPointsValues" else -int(idx) - 1
        info = [None] * 8
        for j, kind, idx in re.findall(r"child\.verticesInfo\[(\d)\] = (pointsInfo|node\.verticesInfo)\[(\d+)\]", inner):
            info[int(j)] = int(idx) if kind == "pointsInfo" else -int(idx) - 1
        assert None not in row and row == info, (c, row, info)
        seen[c] = row
    assert sorted(seen) == list(range(8))
    assert [seen[c] for c in range(8)] == src.tolist(), ([seen[c] for c in range(8)], src.tolist())
    print("child stencil hand-down: 8 x 8 sources identical")
    # trapezoid rule (OctreeSdfUtils.h:60-85): term m =  w/64 * pow2(middlePoints[m][0] - interpolateValue(coeff, p_m))  in order m = 0..18,
    # w in {2,4,8}, p_m = the sample offset mapped to [0,1]^3
    ut = open("/root/reference/include/SdfLib/OctreeSdfUtils.h").read()
    fn = ut[ut.index("estimateErrorFunctionIntegralByTrapezoidRule"):]
    fn = fn[fn.index("return"):fn.index(";")]
    terms = re.findall(r"(\d)\.0f / 64\.0f \* pow2\(middlePoints\[(\d+)\]\[0\] - Inter::interpolateValue\(interpolationCoeff, glm::vec3\(([^)]*)\)\)\)", fn)
    assert len(terms) == 19 and [int(t[1]) for t in terms] == list(range(19))
    assert [float(t[0]) for t in terms] == wt.tolist(), ([t[0] for t in terms], wt.tolist())
    frac = [[float(x.strip().rstrip("f")) for x in t[2].split(",")] for t in terms]
    assert frac == (0.5 * rel + 0.5).tolist(), (frac, (0.5 * rel + 0.5).tolist())
    print("trapezoid rule: 19 weights, evaluation points and their order identical")
    # Simpson's rule (OctreeSdfUtils.h:213-238): same points and order, weights (w^2)/216
    fn = ut[ut.index("estimateErrorFunctionIntegralBySimpsonsRule"):]
    fn = fn[fn.index("return"):fn.index(";")]
    terms = re.findall(r"(\d+)\.0f / 216\.0f \* pow2\(middlePoints\[(\d+)\]\[0\] - Inter::interpolateValue\(interpolationCoeff, glm::vec3\(([^)]*)\)\)\)", fn)
    assert len(terms) == 19 and [int(t[1]) for t in terms] == list(range(19))
    assert [float(t[0]) for t in terms] == (wt * wt).tolist(), ([t[0] for t in terms], (wt * wt).tolist())
    assert [[float(x.strip().rstrip("f")) for x in t[2].split(",")] for t in terms] == frac
    print("Simpson's rule: 19 weights (w^2/216), points and order identical")
    # by-distance (decay) rule (OctreeSdfUtils.h:87-138): value at p_m, then  w/64 * pow2(max(|mid - value| - decay * |value|, 0))
    fn = ut[ut.index("estimateDecayErrorFunctionIntegralByTrapezoidRule"):]
    fn = fn[:fn.index("return error;")]
    pts_d = [[float(x.strip().rstrip("f")) for x in t.split(",")] for t in re.findall(r"value = Inter::interpolateValue\(interpolationCoeff, glm::vec3\(([^)]*)\)\);", fn)]
    terms = re.findall(r"error \+= (\d)\.0f / 64\.0f \* pow2\(glm::max\(glm::abs\(middlePoints\[(\d+)\]\[0\] ?- ?value\) - errorDecayByDistance \* glm::abs\(value\), 0\.0f\)\);", fn)
    assert len(terms) == 19 and len(pts_d) == 19 and [int(t[1]) for t in terms] == list(range(19))
    assert [float(t[0]) for t in terms] == wt.tolist() and pts_d == frac
    print("by-distance rule: 19 weights, points, order and the max(|e| - decay |v|, 0) form identical")
    from tools.refpin import groups
    for g in groups.GROUPS:
        print(g())
    return 0


if __name__ == "__main__":
    sys.exit(main())
