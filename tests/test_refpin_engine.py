"""The text-pin machinery itself (tools/refpin: C++-subset parser + symbolic path executor), on hand-made snippets: what it must call
identical, and what it must not.  CPU only, no reference needed."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tools.refpin import cparse, symex  # noqa: E402


def run(text, params=(), **kw):
    return symex.Exec(**kw).run([("float", p) for p in params], cparse.parse_body("{" + text + "}"))


def same(a, b):
    return symex.first_difference(a, b) is None


def test_locals_and_vocabulary_vanish():
    a = run("glm::vec3 d = p - glm::vec3(s, 0.0, 0.0); const float q = glm::dot(d, d); return glm::sign(q) * glm::sqrt(q);", ("p", "s"))
    b = run("V3 e = x - v3(t, 0.f, 0.f); float w = dot(e, e); return gsign(w) * std::sqrt(w);", ("x", "t"))
    assert same(a, b)
    c = run("V3 e = v3(t, 0.f, 0.f) - x; float w = dot(e, e); return gsign(w) * std::sqrt(w);", ("x", "t"))          # operands swapped
    assert not same(a, c)
    d = run("V3 e = x - v3(t, 0.f, 0.f); float w = dot(e, e); return std::sqrt(w) * gsign(w);", ("x", "t"))          # product commuted
    assert not same(a, d)


def test_literal_kinds_matter_outside_vector_constructors():
    assert not same(run("return a < 1.0e-5;", ("a",)), run("return a < 1.0e-5f;", ("a",)))
    assert same(run("return a * glm::vec3(1.0, 0, 2.0f);", ("a",)), run("return a * V3{1.f, 0.f, 2.f};", ("a",)))


def test_branches_fork_and_their_order_is_compared():
    a = run("if (x >= 0) { if (y <= 0) return x; else return y; } return x + y;", ("x", "y"))
    b = run("float r; if (x >= 0) { r = (y <= 0) ? x : y; return r; } return x + y;", ("x", "y"))
    assert len(a) == 3 and same(a, b)                    # a returned ternary forks like the if / else it stands for
    c = run("if (x > 0) { if (y <= 0) return x; else return y; } return x + y;", ("x", "y"))
    assert not same(a, c)


def test_literal_loops_unroll_and_symbolic_ones_are_summarised():
    a = run("float s = 0.0f; for (int i = 0; i < 3; i++) s = s + v[i] * v[i]; return s;", ("v",))
    b = run("return ((0.0f + v[0] * v[0]) + v[1] * v[1]) + v[2] * v[2];", ("v",))
    assert same(a, b)
    c = run("float s = 0.0f; for (uint32_t i = 0; i < n; i++) { s += v[i]; } return s;", ("v", "n"))
    d = run("float acc = 0.0f; for (uint32_t k = 0; k < n; k++) { acc = acc + v[k]; } return acc;", ("v", "n"), local_alias={"acc": "s", "k": "i"})
    assert same(c, d)
    e = run("float acc = 0.0f; for (uint32_t k = 0; k < n; k++) { acc = v[k] + acc; } return acc;", ("v", "n"), local_alias={"acc": "s", "k": "i"})
    assert not same(c, e)
    f = run("float acc = 1.0f; for (uint32_t k = 0; k < n; k++) { acc = acc + v[k]; } return acc;", ("v", "n"), local_alias={"acc": "s", "k": "i"})
    assert not same(c, f)                                # the value a loop-carried variable enters with is part of the summary


def test_a_read_cannot_move_across_a_store():
    a = run("out[0] = x; float t = out[0]; out[1] = t + 1.0f;", ("x",))
    b = run("float t = out[0]; out[0] = x; out[1] = t + 1.0f;", ("x",))
    assert not same(a, b)
    c = run("out[0] = x; out[1] = out[0] + 1.0f;", ("x",))
    assert same(a, c)


def test_inlined_helpers_and_switches():
    helper = (cparse.parse_params("float a, float b"), cparse.parse_body("{ return (b < a) ? b : a; }"))
    a = symex.Exec(funcs={"lo": [helper]}).run([("float", "x"), ("float", "y")], cparse.parse_body("{ return lo(x, y) + 1.0f; }"))
    b = run("return ((y < x) ? y : x) + 1.0f;", ("x", "y"))
    assert same(a, b)
    s1 = run("int r = R_A; if (x > 0) r = R_B; switch (r) { case R_A: return x; case R_B: return -x; default: return 0.0f; }", ("x",))
    s2 = run("if (x > 0) return -x; return x;", ("x",))
    assert same(s1, s2)
