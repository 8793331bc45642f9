"""Symbolic path execution of the C++ subset parsed by cparse.py.

Purpose (dev-time test infrastructure): turn a function of the reference and its restatement in oracle/ into the same
canonical object — an ordered list of paths, each `(branch conditions taken, memory stores / opaque calls in order, returned
expression)` with every local variable substituted by its defining expression — so that the two can be compared for
equality.  Equal canonical forms mean: the same floating-point operations on the same operands in the same order, under
the same branch conditions.  What it does NOT look into: functions it is told to treat as opaque (compared separately) and
glm's own operators (`*` on a matrix, dot, normalize, inverse ...), which are vocabulary here.

Model
  * locals hold expression trees (or aggregates of them: structs / arrays with concrete indices); reads substitute.
  * `if` on a non-constant condition forks the path; integer expressions on literals fold, so loops with literal bounds
    unroll and their indices become concrete.
  * a loop whose condition is not concrete is summarised: variables assigned in it are replaced by ('lv', loop, name)
    symbols ("the value at the start of an arbitrary iteration"); the body runs once; the summary records entry values
    (of the loop-carried variables actually read), the condition, and per body path its stores and the end-of-iteration
    values.  `break` / condition-false paths continue behind the loop.
  * stores to anything that is not a local (out-parameters, members, containers) are events, kept in program order; reads
    of a root object that was stored to earlier on the path carry the store count ('at', n, expr) so that a read cannot
    silently move across a store.
"""
import re
import sys
from . import cparse

sys.setrecursionlimit(20000)

NS_STRIP = ("glm::", "std::", "TriangleUtils::", "GJK::", "InterpolationMethod::", "Inter::", "orc::", "sdflib::")
MUTATORS = {"insert", "erase", "push_back", "clear", "fill", "resize", "emplace_back", "pop_back", "assign", "reserve"}
VEC3_CTORS = {"vec3", "V3", "v3"}
VEC2_CTORS = {"vec2", "V2"}
MAT3_CTORS = {"mat3x3", "mat3"}
DEFAULT_FN_ALIAS = {"gsign": "sign", "gmin": "min", "gmax": "max", "gclamp": "clamp", "gabs": "abs", "fabs": "abs", "gfract": "fract",
                    "fabsf": "abs", "sqrtf": "sqrt", "floorf": "floor", "acosf": "acos"}


def strip_ns(name):
    changed = True
    while changed:
        changed = False
        for p in NS_STRIP:
            if name.startswith(p):
                name = name[len(p):]; changed = True
    return name


class Agg:
    """struct / array local with concrete keys."""
    __slots__ = ("type", "items")

    def __init__(self, type_, items=None):
        self.type = type_; self.items = items or {}

    def clone(self):
        return Agg(self.type, {k: (v.clone() if isinstance(v, Agg) else v) for k, v in self.items.items()})


class LoopInfo:
    def __init__(self, lid, kind):
        self.lid, self.kind, self.entry, self.used, self.header = lid, kind, {}, set(), None


class St:
    def __init__(self):
        self.scopes = [{}]
        self.obj = {}            # stores to undeclared identifiers (constructor members)
        self.events = []
        self.conds = []
        self.epoch = {}
        self.loops = [0]         # shared counter cell (per top-level run)

    def clone(self):
        s = St.__new__(St)
        s.scopes = [{k: (v.clone() if isinstance(v, Agg) else v) for k, v in sc.items()} for sc in self.scopes]
        s.obj = {k: (v.clone() if isinstance(v, Agg) else v) for k, v in self.obj.items()}
        s.events = list(self.events); s.conds = list(self.conds); s.epoch = dict(self.epoch); s.loops = self.loops
        return s

    def find(self, name):
        for sc in reversed(self.scopes):
            if name in sc:
                return sc
        return None


def is_lit(x):
    return isinstance(x, tuple) and x[0] == "lit"


def is_int(x):
    return is_lit(x) and x[1] in ("int", "uint", "bool")


def ival(x):
    return int(x[2])


def to_float_lit(x):
    if is_lit(x) and x[1] in ("int", "uint", "double", "float"):
        return ("lit", "float", float(x[2]))
    return x


class PinError(Exception):
    pass


class Exec:
    def __init__(self, funcs=None, fn_alias=None, member_alias=None, id_alias=None, local_alias=None, opaque_roots=(), drop_calls=("SPDLOG_INFO", "SPDLOG_ERROR", "SPDLOG_DEBUG", "SPDLOG_WARN"),
                 unroll_limit=4096):
        self.funcs = funcs or {}              # name -> (params, body) to inline
        self.fn_alias = dict(DEFAULT_FN_ALIAS); self.fn_alias.update(fn_alias or {})
        self.member_alias = member_alias or {}
        self.id_alias = id_alias or {}          # free identifiers (members of the enclosing object, globals)
        self.local_alias = local_alias or {}    # names of loop-carried locals
        self.drop_calls = set(drop_calls)
        self.unroll_limit = unroll_limit
        self.call_stack = []
        self.loop_infos = {}
        self.globals = {}                       # file-scope constants given by value (name -> expression / Agg)
        self.rewrite = None                     # optional fn(node) -> node, applied to every freshly built opaque node

    def rw(self, x):
        return self.rewrite(x) if self.rewrite is not None and isinstance(x, tuple) else x

    # ------------------------------------------------------------------ running a function
    def run(self, params, body, ctor=False):
        """params: [(type, name)]; returns the list of canonical path records."""
        st = St()
        self.loop_infos = {}
        for i, (t, n) in enumerate(params):
            st.scopes[0][n] = ("arg", i)
        paths = []
        self.params, self.ctor_rec, self.paths = params, ctor, paths
        for s, c in self.exec_stmt(body, st):
            self.finished(s, c)
        return [self.finish(p) for p in paths]

    def finished(self, s, c):
        """a path ended (function end / return, or the back edge of a summarised loop — possibly inside an inlined callee)"""
        params, ctor, paths = self.params, self.ctor_rec, self.paths
        if True:
            rec = {"conds": tuple(s.conds), "events": tuple(s.events), "end": c}
            outs = {}
            for i, (t, n) in enumerate(params):          # parameters written through (references / out-parameters)
                v = self.snapshot(s.scopes[0][n])
                if v != ("arg", i):
                    outs[i] = v
            rec["outs"] = outs
            if ctor:
                rec["obj"] = {k: self.snapshot(v) for k, v in s.obj.items()}
            paths.append(rec)

    def finish(self, p):
        def fix(x):
            if isinstance(x, LoopInfo):
                return ("loop", x.lid, x.kind, x.header, tuple(sorted((k, fix(v)) for k, v in x.entry.items() if k in x.used)))
            if isinstance(x, tuple):
                return tuple(fix(y) for y in x)
            if isinstance(x, list):
                return tuple(fix(y) for y in x)
            if isinstance(x, dict):
                return tuple(sorted((k, fix(v)) for k, v in x.items()))
            if isinstance(x, Agg):
                return fix(self.snapshot(x))
            return x
        return {k: fix(v) for k, v in p.items()}

    # ------------------------------------------------------------------ values
    def snapshot(self, v):
        """aggregate -> canonical expression"""
        if not isinstance(v, Agg):
            return v
        t = strip_ns(v.type or "")
        it = {k: self.snapshot(x) for k, x in v.items.items()}
        if t in ("V3",) and set(it) <= {"x", "y", "z"}:
            return ("vec3", it.get("x", ("undef",)), it.get("y", ("undef",)), it.get("z", ("undef",)))
        if t in ("V2",):
            return ("vec2", it.get("x", ("undef",)), it.get("y", ("undef",)))
        if t in ("M3",) and "c" in it:
            c = dict(it["c"][1:]) if isinstance(it["c"], tuple) and it["c"][0] == "agg" else {}
            return ("mat3", c.get(0, ("undef",)), c.get(1, ("undef",)), c.get(2, ("undef",)))
        return ("agg",) + tuple(sorted(it.items(), key=lambda kv: (str(type(kv[0])), kv[0])))

    def lookup(self, name, st):
        sc = st.find(name)
        if sc is not None:
            v = sc[name]
            self.note_lv(v)
            return v
        if name in st.obj:
            return st.obj[name]
        if name in self.globals:
            return self.globals[name]
        n = strip_ns(name)
        return ("id", self.id_alias.get(n, n))

    def note_lv(self, v):
        if isinstance(v, tuple) and v and v[0] == "lv":
            self.loop_infos[v[1]].used.add(v[2])

    def root_of(self, x):
        while isinstance(x, tuple) and x[0] in ("member", "index", "at", "un", "mcall"):
            x = x[2] if x[0] in ("at", "un") else x[1]
        return x

    def tag_read(self, x, st):
        r = self.root_of(x)
        e = st.epoch.get(r, 0)
        return ("at", e, x) if e else x

    # ------------------------------------------------------------------ expression evaluation (generator: paths may fork in inlined calls)
    def ev(self, x, st):
        k = x[0]
        if k == "lit":
            yield st, x
        elif k == "id":
            yield st, self.lookup(x[1], st)
        elif k == "bin":
            if x[1] == ",":
                for s1, _ in self.ev(x[2], st):
                    yield from self.ev(x[3], s1)
                return
            for s1, l in self.ev(x[2], st):
                for s2, r in self.ev(x[3], s1):
                    yield s2, self.binop(x[1], l, r)
        elif k == "un":
            op = x[1]
            if op in ("++", "--"):
                for s1, v in self.ev(x[2], st):
                    nv = self.binop("+" if op == "++" else "-", v, ("lit", "int", 1))
                    yield from self.assign_to(x[2], nv, s1, result=nv)
                return
            for s1, v in self.ev(x[2], st):
                yield s1, self.unop(op, v, s1)
        elif k == "post":
            for s1, v in self.ev(x[2], st):
                nv = self.binop("+" if x[1] == "++" else "-", v, ("lit", "int", 1))
                yield from self.assign_to(x[2], nv, s1, result=v)
        elif k == "assign":
            op, lhs, rhs = x[1], x[2], x[3]
            if op == "=":
                for s1, v in self.ev_init(rhs, st):
                    yield from self.assign_to(lhs, v, s1, result=v)
            else:
                for s1, cur in self.ev(lhs, st):
                    for s2, v in self.ev(rhs, s1):
                        nv = self.binop(op[:-1], cur, v)
                        yield from self.assign_to(lhs, nv, s2, result=nv)
        elif k == "tern":
            for s1, c in self.ev(x[1], st):
                if is_int(c):
                    yield from self.ev(x[2] if ival(c) else x[3], s1)
                else:
                    for s2, a in self.ev(x[2], s1):
                        for s3, b in self.ev(x[3], s2):
                            yield s3, ("tern", c, a, b)
        elif k == "cast":
            for s1, v in self.ev(x[2], st):
                t = strip_ns(x[1]).replace("const ", "").strip()
                if is_lit(v) and t in ("float",) and v[1] in ("int", "uint", "float"):
                    yield s1, ("lit", "float", float(v[2]))
                elif is_int(v) and t in ("int", "uint32_t", "uint64_t", "uint8_t", "size_t", "int32_t"):
                    yield s1, v
                else:
                    yield s1, self.rw(("cast", t, self.snapshot(v)))
        elif k == "member":
            for s1, o in self.ev(x[1], st):
                yield s1, self.member(o, x[2], s1)
        elif k == "index":
            for s1, o in self.ev(x[1], st):
                for s2, i in self.ev(x[2], s1):
                    yield s2, self.index(o, i, s2)
        elif k == "brace":
            yield from self.ev_brace(x, st)
        elif k == "lambda":
            yield st, ("closure", x[1], x[2])
        elif k == "call":
            yield from self.ev_call(x, st)
        else:
            raise PinError("cannot evaluate %r" % (x,))

    def ev_init(self, x, st):
        if x[0] == "brace":
            yield from self.ev_brace(x, st)
        else:
            yield from self.ev(x, st)

    def ev_list(self, xs, st, i=0, acc=()):
        if i == len(xs):
            yield st, list(acc); return
        for s1, v in self.ev_init(xs[i], st):
            yield from self.ev_list(xs, s1, i + 1, acc + (v,))

    def ev_brace(self, x, st):
        t = strip_ns(x[1]) if x[1] else None
        for s1, items in self.ev_list(x[2], st):
            if t in VEC3_CTORS or t in VEC2_CTORS or t in MAT3_CTORS:
                yield s1, self.ctor(t, items)
            else:
                yield s1, Agg(t, {i: v for i, v in enumerate(items)})

    def ctor(self, t, a):
        a = [self.snapshot(v) for v in a]
        if t in VEC3_CTORS:
            if len(a) == 1:
                a = a * 3
            if len(a) != 3:
                raise PinError("vec3 constructor with %d arguments" % len(a))
            return ("vec3",) + tuple(to_float_lit(v) for v in a)
        if t in VEC2_CTORS:
            if len(a) == 1:
                return ("vec2", self.member(a[0], "x", None), self.member(a[0], "y", None))
            return ("vec2",) + tuple(to_float_lit(v) for v in a)
        if t in MAT3_CTORS:
            return ("mat3",) + tuple(a)
        raise PinError(t)

    def member(self, o, name, st):
        if isinstance(o, Agg):
            if name not in o.items:
                o.items[name] = Agg(None)          # struct field touched before being written (e.g. frame.c[0] = ...)
            v = o.items[name]
            self.note_lv(v)
            return v
        if isinstance(o, tuple) and o[0] in ("vec3", "vec2") and name in ("x", "y", "z"):
            i = "xyz".index(name)
            if i + 1 < len(o):
                return o[1 + i]
        if isinstance(o, tuple) and o[0] == "mat3" and name == "c":
            return Agg(None, {0: o[1], 1: o[2], 2: o[3]})
        if name == "c" and self.is_matrix(o):
            return o                                               # oracle M3 keeps its columns in .c[3]
        name = self.member_alias.get(name, name)
        r = self.rw(("member", o, name))
        return self.tag_read(r, st) if st is not None and r[0] == "member" else r

    def component(self, v, ax):
        """component of a vector expression; glm's floor / fract / abs on vectors act per component"""
        if isinstance(v, tuple) and v[0] == "call" and v[1] in ("floor", "fract", "abs") and len(v[2]) == 1:
            return ("call", v[1], (self.component(v[2][0], ax),))
        return self.member(v, ax, None)

    def index(self, o, i, st):
        if isinstance(o, Agg):
            if is_int(i):
                if ival(i) not in o.items:
                    o.items[ival(i)] = Agg(None)
                return o.items[ival(i)]
            return ("select", self.snapshot(o), i)
        if isinstance(o, tuple) and o[0] in ("vec3", "vec2", "mat3") and is_int(i) and 1 + ival(i) < len(o):
            return o[1 + ival(i)]
        if isinstance(o, tuple) and o[0] == "index" and is_int(i) and is_int(o[2]) and self.is_matrix(o[1]):
            return ("member", o, "xyz"[ival(i)])                     # glm mat3 m[col][row]  ==  oracle m.c[col].{x,y,z}
        r = self.rw(("index", o, i))
        return self.tag_read(r, st) if r[0] == "index" else r

    def is_matrix(self, x):
        x = self.strip_at(x)
        return isinstance(x, tuple) and ((x[0] == "member" and x[2] == "transform") or (x[0] == "id" and x[1] == "transform"))

    def binop(self, op, l, r):
        l, r = self.snapshot(l), self.snapshot(r)
        if is_int(l) and is_int(r):
            a, b = ival(l), ival(r)
            kind = "uint" if "uint" in (l[1], r[1]) else "int"
            try:
                if op in ("+", "-", "*", "<<", ">>", "&", "|", "^"):
                    v = {"+": a + b, "-": a - b, "*": a * b, "<<": a << b, ">>": a >> b, "&": a & b, "|": a | b, "^": a ^ b}[op]
                    if kind == "uint":
                        v &= 0xFFFFFFFF
                    return ("lit", kind, v)
                if op == "/" and b:
                    return ("lit", kind, int(a / b) if kind == "int" else a // b)
                if op == "%" and b:
                    return ("lit", kind, a - b * int(a / b))
                if op in ("<", ">", "<=", ">=", "==", "!="):
                    return ("lit", "bool", {"<": a < b, ">": a > b, "<=": a <= b, ">=": a >= b, "==": a == b, "!=": a != b}[op])
                if op == "&&":
                    return ("lit", "bool", bool(a) and bool(b))
                if op == "||":
                    return ("lit", "bool", bool(a) or bool(b))
            except (ValueError, ZeroDivisionError):
                pass
        if op == "+" and is_int(r) and ival(r) == 0 and r[1] != "bool":          # index arithmetic of unrolled loops: (3 * t + 0)
            return l
        if op in ("==", "!=") and l == r and l[0] == "id":
            return ("lit", "bool", op == "==")
        if op in ("==", "!=") and l[0] == "id" and r[0] == "id" and l[1].isupper() and r[1].isupper():
            return ("lit", "bool", (l == r) == (op == "=="))
        if op == "&&" and is_int(l):
            return r if ival(l) else ("lit", "bool", False)
        if op == "||" and is_int(l):
            return ("lit", "bool", True) if ival(l) else r
        return self.rw(("bin", op, l, r))

    def unop(self, op, v, st):
        v = self.snapshot(v)
        if op == "-" and is_lit(v) and v[1] in ("int", "float", "double"):
            return ("lit", v[1], -v[2])
        if op == "!" and is_int(v):
            return ("lit", "bool", not ival(v))
        if op == "+":
            return v
        if op == "*":
            if isinstance(v, tuple) and v[0] == "un" and v[1] == "&":
                return v[2]
            v2 = self.rw(("un", "*", v))
            return self.tag_read(v2, st) if v2[0] == "un" else v2
        return self.rw(("un", op, v))

    # ------------------------------------------------------------------ calls
    def canon_fn(self, name):
        n = strip_ns(name)
        return self.fn_alias.get(n, n)

    def ev_call(self, x, st):
        fn, args = x[1], x[2]
        if fn[0] == "id":
            raw = strip_ns(fn[1])
            if raw in self.drop_calls:
                yield st, ("void",); return
            target = None
            sc = st.find(fn[1])
            if sc is not None and isinstance(sc[fn[1]], tuple) and sc[fn[1]][0] == "closure":
                target = sc[fn[1]]
            for s1, a in self.ev_list(args, st):
                if target is not None and fn[1] not in self.call_stack:
                    yield from self.inline(fn[1], [(None, p[1]) for p in target[1]], target[2], a, s1, closure=True)
                    continue
                cands = self.funcs.get(raw)
                if cands and raw not in self.call_stack:
                    match = [c for c in cands if len(c[0]) == len(a)]
                    if match:
                        yield from self.inline(raw, match[0][0], match[0][1], a, s1)
                        continue
                if raw in VEC3_CTORS or raw in VEC2_CTORS or raw in MAT3_CTORS:
                    yield s1, self.ctor(raw, a); continue
                if raw in ("ivec3", "iv3") and len(a) == 1:            # component-wise conversion of a vec3 expression to int
                    yield s1, ("vec3",) + tuple(("cast", "int", self.component(self.snapshot(a[0]), ax)) for ax in "xyz"); continue
                name = self.canon_fn(fn[1])
                a = [self.snapshot(v) for v in a]
                if name == "mul" and len(a) == 2:
                    yield s1, ("bin", "*", a[0], a[1]); continue
                if name == "mulT" and len(a) == 2:
                    yield s1, ("bin", "*", ("call", "transpose", (a[0],)), a[1]); continue
                if name == "make_pair" or name == "pair":
                    yield s1, ("pair",) + tuple(a); continue
                yield s1, self.rw(("call", name, tuple(a)))
            return
        if fn[0] == "member":
            for s1, o in self.ev(fn[1], st):
                for s2, a in self.ev_list(args, s1):
                    a = tuple(self.snapshot(v) for v in a)
                    m = self.member_alias.get(fn[2], fn[2])
                    if isinstance(o, Agg):
                        if m == "fill" and len(a) == 1:
                            mm = re.search(r",\s*(\d+)\s*>$", o.type or "")
                            if not mm:
                                raise PinError("fill() on an aggregate of unknown size (%r)" % o.type)
                            for k in range(int(mm.group(1))):
                                o.items[k] = a[0]
                            yield s2, ("void",); continue
                        if m == "size":
                            yield s2, ("lit", "uint", len([k for k in o.items if isinstance(k, int)])); continue
                        raise PinError("method %s on a local aggregate" % m)
                    o = self.snapshot(o)
                    if m in MUTATORS:
                        r = self.root_of(o)
                        s2.events.append(self.rw(("mcall", o, m, a)))
                        s2.epoch[r] = s2.epoch.get(r, 0) + 1
                        yield s2, ("mresult", len(s2.events) - 1, m)
                    else:
                        r = self.rw(("mcall", o, m, a))
                        yield s2, self.tag_read(r, s2) if r[0] == "mcall" else r
            return
        raise PinError("call of %r" % (fn,))

    def inline(self, name, params, body, args, st, closure=False):
        st.scopes.append({"__fn__": name})
        for (t, n), v in zip(params, args):
            st.scopes[-1][n] = v.clone() if isinstance(v, Agg) else v
        depth = len(st.scopes)
        self.call_stack.append(name)
        gen = self.exec_stmt(body, st)
        while True:
            try:
                s1, c = next(gen)
            except StopIteration:
                break
            finally:
                self.call_stack.pop()        # the consumer of a yielded path runs outside the callee
            del s1.scopes[depth - 1:]
            if isinstance(c, tuple) and c[0] == "ret":
                yield s1, c[1]
            elif c is None:
                yield s1, ("void",)
            elif isinstance(c, tuple) and c[0] == "back":
                self.finished(s1, c)             # back edge of a loop inside the callee: the path ends here
            else:
                raise PinError("control %r escapes inlined %s" % (c, name))
            self.call_stack.append(name)

    # ------------------------------------------------------------------ stores
    def assign_to(self, lhs, v, st, result):
        """store v into the l-value lhs; yields (state, result)"""
        k = lhs[0]
        if k == "id":
            sc = st.find(lhs[1])
            if sc is not None:
                old = sc[lhs[1]]
                if isinstance(old, tuple) and old[0] == "ref":          # reference parameter / alias of non-local memory
                    yield from self.store_mem(old[1], v, st, result); return
                sc[lhs[1]] = v.clone() if isinstance(v, Agg) else v
                yield st, result; return
            n = strip_ns(lhs[1])
            if n in st.obj or self.ctor_mode:
                st.obj[n] = v
                yield st, result; return
            yield from self.store_mem(("id", self.id_alias.get(n, n)), v, st, result); return
        if k == "member":
            if lhs[1] == ("un", "*", ("id", "this")):
                st.obj[lhs[2]] = v
                yield st, result; return
            for s1, o in self.ev(lhs[1], st):
                if isinstance(o, Agg):
                    o.items[lhs[2]] = v
                    yield s1, result
                elif isinstance(o, tuple) and o[0] in ("vec3", "vec2") and lhs[2] in "xyz":
                    no = list(o); no[1 + "xyz".index(lhs[2])] = self.snapshot(v)
                    yield from self.assign_to(lhs[1], tuple(no), s1, result)
                else:
                    yield from self.store_mem(("member", self.strip_at(o), self.member_alias.get(lhs[2], lhs[2])), v, s1, result)
            return
        if k == "index":
            for s1, o in self.ev(lhs[1], st):
                for s2, i in self.ev(lhs[2], s1):
                    if isinstance(o, Agg) and is_int(i):
                        o.items[ival(i)] = v
                        yield s2, result
                    elif isinstance(o, tuple) and o[0] in ("vec3", "vec2") and is_int(i):
                        no = list(o); no[1 + ival(i)] = self.snapshot(v)
                        yield from self.assign_to(lhs[1], tuple(no), s2, result)
                    elif isinstance(o, Agg):
                        raise PinError("store into a local array at a symbolic index")
                    else:
                        yield from self.store_mem(("index", self.strip_at(o), i), v, s2, result)
            return
        if k == "un" and lhs[1] == "*":
            for s1, p in self.ev(lhs[2], st):
                yield from self.store_mem(("un", "*", p), v, s1, result)
            return
        raise PinError("cannot assign to %r" % (lhs,))

    ctor_mode = False

    def strip_at(self, x):
        return x[2] if isinstance(x, tuple) and x[0] == "at" else x

    def store_mem(self, target, v, st, result):
        target = self.rw(target)
        r = self.root_of(target)
        st.events.append(("store", target, self.snapshot(v)))
        st.epoch[r] = st.epoch.get(r, 0) + 1
        yield st, result

    # ------------------------------------------------------------------ statements
    def exec_seq(self, items, i, st):
        if i == len(items):
            yield st, None; return
        for s1, c in self.exec_stmt(items[i], st):
            if c is not None:
                yield s1, c
            else:
                yield from self.exec_seq(items, i + 1, s1)

    def exec_stmt(self, s, st):
        k = s[0]
        if k == "block":
            st.scopes.append({})
            depth = len(st.scopes)
            for s1, c in self.exec_seq(s[1], 0, st):
                del s1.scopes[depth - 1:]
                yield s1, c
        elif k == "empty":
            yield st, None
        elif k == "expr":
            x = s[1]
            for s1, v in self.ev(x, st):
                if x[0] == "call" and isinstance(v, tuple) and v[0] == "call":       # opaque call evaluated for its effect
                    s1.events.append(("callstmt", v))
                yield s1, None
        elif k == "decl":
            yield from self.exec_decl(s, 0, st)
        elif k == "return":
            if s[1] is None:
                yield st, ("ret", None); return
            for s1, v in self.ev_init(s[1], st):
                if self.call_stack:                                  # inlined callee: aggregates stay aggregates
                    yield s1, ("ret", v.clone() if isinstance(v, Agg) else v)
                else:
                    yield from self.ret_split(self.snapshot(v), s1)
        elif k == "break":
            yield st, "brk"
        elif k == "continue":
            yield st, "cont"
        elif k == "if":
            if self.is_noop(s[2]) and (s[3] is None or self.is_noop(s[3])):
                yield st, None; return
            for s1, c in self.ev(s[1], st):
                c = self.snapshot(c)
                if is_int(c):
                    br = s[2] if ival(c) else s[3]
                    if br is None:
                        yield s1, None
                    else:
                        yield from self.exec_stmt(br, s1)
                else:
                    s_then = s1.clone(); s_then.conds.append((c, True))
                    s1.conds.append((c, False))
                    yield from self.exec_stmt(s[2], s_then)
                    if s[3] is None:
                        yield s1, None
                    else:
                        yield from self.exec_stmt(s[3], s1)
        elif k == "switch":
            for s1, v in self.ev(s[1], st):
                v = self.snapshot(v)
                start = None
                for ci, (lab, _) in enumerate(s[2]):
                    if lab is None:
                        continue
                    lv = next(self.ev(lab, s1))[1]
                    eq = self.binop("==", v, lv)
                    if not is_int(eq):
                        raise PinError("switch on a symbolic value %r" % (v,))
                    if ival(eq):
                        start = ci; break
                if start is None:
                    start = next((ci for ci, (lab, _) in enumerate(s[2]) if lab is None), None)
                if start is None:
                    yield s1, None; continue
                items = [x for _, body in s[2][start:] for x in body]
                for s2, c in self.exec_stmt(("block", items), s1):
                    yield s2, (None if c == "brk" else c)
        elif k == "for":
            st.scopes.append({})
            depth = len(st.scopes)
            init = s[1] or ("empty",)
            for s1, _ in self.exec_stmt(init, st):
                for s2, c in self.loop(s1, "for", s[2], s[3], s[4]):
                    del s2.scopes[depth - 1:]
                    yield s2, c
        elif k == "while":
            yield from self.loop(st, "while", s[1], None, s[2])
        elif k == "do":
            yield from self.loop(st, "do", s[2], None, s[1])
        elif k == "rangefor":
            yield from self.rangefor(s, st)
        else:
            raise PinError("statement %r" % (k,))

    def is_noop(self, s):
        if s[0] == "empty":
            return True
        if s[0] == "block":
            return all(self.is_noop(x) for x in s[1])
        if s[0] == "expr" and s[1][0] == "call" and s[1][1][0] == "id" and strip_ns(s[1][1][1]) in self.drop_calls:
            return True
        return False

    def ret_split(self, v, st):
        if isinstance(v, tuple) and v[0] == "tern" and not self.call_stack:
            s_a = st.clone(); s_a.conds.append((v[1], True))
            st.conds.append((v[1], False))
            yield from self.ret_split(v[2], s_a)
            yield from self.ret_split(v[3], st)
        else:
            yield st, ("ret", v)

    def exec_decl(self, s, i, st):
        if i == len(s[2]):
            yield st, None; return
        t = s[1]
        name, dims, init, kind = s[2][i]
        tt = strip_ns(t.replace("&", "").replace("*", ""))
        if init is None:
            if self.is_container(tt):
                v = ("id", self.local_alias.get(name, name))           # an object with identity (map / vector / function): named, not valued
                st.events.append(("construct", v, ()))
            elif dims or tt.startswith("array<") or tt in ("M3", "Proj", "TriangleData", "V3", "V2") or (tt[:1].isupper() and not tt.startswith("INF")):
                v = Agg(tt)
            else:
                v = ("undef",)
            st.scopes[-1][name] = v
            yield from self.exec_decl(s, i + 1, st); return
        if kind == "(":
            for s1, a in self.ev_list(init, st):
                if tt in VEC3_CTORS or tt in VEC2_CTORS or tt in MAT3_CTORS:
                    v = self.ctor(tt, a)
                elif self.is_container(tt):
                    v = ("id", self.local_alias.get(name, name))
                    s1.events.append(("construct", v, tuple(self.snapshot(q) for q in a)))
                elif len(a) == 1:
                    v = a[0]
                else:
                    v = ("call", tt, tuple(self.snapshot(q) for q in a))
                s1.scopes[-1][name] = v
                yield from self.exec_decl(s, i + 1, s1)
            return
        if init[0] == "lambda":
            st.scopes[-1][name] = ("closure", init[1], init[2])
            yield from self.exec_decl(s, i + 1, st); return
        for s1, v in self.ev_init(init, st):
            if tt == "ivec3" and not isinstance(v, Agg) and not (v[0] == "vec3" and all(c[0] == "cast" and c[1] == "int" for c in v[1:])):
                v = ("vec3",) + tuple(("cast", "int", self.component(v, ax)) for ax in "xyz")
            if isinstance(v, Agg):
                v = v.clone()
                if v.type is None:
                    v.type = tt
            elif "&" in t and isinstance(v, tuple) and v[0] in ("member", "index", "at", "un", "id") and tt not in ("auto",) and not t.strip().startswith("const") and False:
                v = ("ref", v)
            elif tt in ("V2",) and isinstance(v, tuple) and v[0] == "agg":
                pass
            s1.scopes[-1][name] = v
            yield from self.exec_decl(s, i + 1, s1)

    def is_container(self, tt):
        return tt.split("<")[0].strip() in ("map", "vector", "unordered_map", "set")

    # ------------------------------------------------------------------ loops
    def assigned(self, *nodes):
        out = []

        def root(x):
            while x[0] in ("member", "index"):
                x = x[1]
            if x[0] == "un" and x[1] == "*":
                return None
            return x[1] if x[0] == "id" else None

        def walk(x):
            if isinstance(x, tuple):
                if x and x[0] == "assign":
                    r = root(x[2])
                    walk(x[3])
                    if r and r not in out:
                        out.append(r)
                    walk(x[2])
                    return
                if x and x[0] in ("un", "post") and x[1] in ("++", "--"):
                    r = root(x[2])
                    if r and r not in out:
                        out.append(r)
                for y in x:
                    walk(y)
            elif isinstance(x, list):
                for y in x:
                    walk(y)
        for n in nodes:
            if n is not None:
                walk(n)
        return out

    def loop(self, st, kind, cond, step, body, it=0):
        """for / while / do.  Concrete conditions unroll; a symbolic one is summarised."""
        if kind != "do" or it > 0:
            if cond is None:
                raise PinError("loop without a condition")
            probe = st.clone()
            res = list(self.ev(cond, probe))
            if len(res) == 1 and is_int(self.snapshot(res[0][1])):
                cst = res[0][0]
                if not ival(self.snapshot(res[0][1])):
                    yield cst, None; return
                st = cst
            else:
                if it > 0:
                    raise PinError("loop condition became symbolic after %d iterations" % it)
                yield from self.summarise(st, kind, cond, step, body); return
        elif kind == "do":
            # a do-loop is summarised unless its condition is concrete after the first pass; the pinned code only has the former
            yield from self.summarise(st, kind, cond, step, body); return
        if it > self.unroll_limit:
            raise PinError("unroll limit")
        for s1, c in self.exec_stmt(body, st):
            if c == "brk":
                yield s1, None
            elif isinstance(c, tuple):
                yield s1, c
            else:
                if step is not None:
                    for s2, _ in self.ev(step, s1):
                        yield from self.loop(s2, kind, cond, step, body, it + 1)
                else:
                    yield from self.loop(s1, kind, cond, step, body, it + 1)

    def new_loop(self, st, kind):
        st.loops[0] += 1
        info = LoopInfo(st.loops[0], kind)
        self.loop_infos[info.lid] = info
        return info

    def havoc(self, st, info, names):
        for n in names:
            sc = st.find(n)
            if sc is None:
                continue
            cn = self.local_alias.get(n, n)
            v = sc[n]
            if isinstance(v, Agg):          # an array (re)filled inside the loop at concrete indices: havoc element-wise
                for kk in list(v.items):
                    if isinstance(v.items[kk], Agg):
                        raise PinError("nested local aggregate %s is modified inside a summarised loop" % n)
                    info.entry["%s[%s]" % (cn, kk)] = v.items[kk]
                    v.items[kk] = ("lv", info.lid, "%s[%s]" % (cn, kk))
                continue
            info.entry[cn] = self.snapshot(v)
            sc[n] = ("lv", info.lid, cn)

    def end_values(self, st, info, names):
        out = []
        for n in names:
            sc = st.find(n)
            if sc is not None and not isinstance(sc[n], Agg):
                cn = self.local_alias.get(n, n)
                v = self.snapshot(sc[n])
                if v != ("lv", info.lid, cn):
                    out.append((cn, v))
        return tuple(out)

    def summarise(self, st, kind, cond, step, body):
        info = self.new_loop(st, kind)
        names = self.assigned(body, cond, step)
        self.havoc(st, info, names)
        st.events.append(info)
        if kind != "do":
            res = list(self.ev(cond, st))
            if len(res) != 1:
                raise PinError("forking loop condition")
            st, c = res[0]
            c = self.snapshot(c)
            s_exit = st.clone(); s_exit.conds.append((("loopcond", info.lid, c), False))
            st.conds.append((("loopcond", info.lid, c), True))
        for s1, ctl in self.exec_stmt(body, st):
            if ctl == "brk":
                s1.events.append(("loopbreak", info.lid))
                yield s1, None
            elif isinstance(ctl, tuple):
                yield s1, ctl
            else:
                if step is not None:
                    s1 = next(self.ev(step, s1))[0]
                if kind == "do":
                    res = list(self.ev(cond, s1))
                    if len(res) != 1:
                        raise PinError("forking loop condition")
                    s1, c = res[0]
                    c = self.snapshot(c)
                    s_out = s1.clone(); s_out.conds.append((("loopcond", info.lid, c), False))
                    s1.conds.append((("loopcond", info.lid, c), True))
                    s1.events.append(("loopback", info.lid, self.end_values(s1, info, names)))
                    yield s1, ("back", info.lid)
                    yield s_out, None
                else:
                    s1.events.append(("loopback", info.lid, self.end_values(s1, info, names)))
                    yield s1, ("back", info.lid)
        if kind != "do":
            yield s_exit, None

    def rangefor(self, s, st):
        decl, seq, body = s[1], s[2], s[3]
        for s1, q in self.ev(seq, st):
            if isinstance(q, Agg):
                keys = sorted(k for k in q.items if isinstance(k, int))
                yield from self.range_unrolled(decl[2][0][0], [q.items[k] for k in keys], 0, body, s1)
                continue
            info = self.new_loop(s1, "range")
            info.header = self.snapshot(q)
            names = self.assigned(body)
            self.havoc(s1, info, names)
            s1.events.append(info)
            s_exit = s1.clone()
            s1.scopes.append({decl[2][0][0]: ("iter", info.lid)})
            depth = len(s1.scopes)
            for s2, ctl in self.exec_stmt(body, s1):
                del s2.scopes[depth - 1:]
                if ctl == "brk":
                    s2.events.append(("loopbreak", info.lid))
                    yield s2, None
                elif isinstance(ctl, tuple):
                    yield s2, ctl
                else:
                    s2.events.append(("loopback", info.lid, self.end_values(s2, info, names)))
                    yield s2, ("back", info.lid)
            yield s_exit, None

    def range_unrolled(self, var, vals, i, body, st):
        if i == len(vals):
            yield st, None; return
        st.scopes.append({var: vals[i]})
        depth = len(st.scopes)
        for s1, c in self.exec_stmt(body, st):
            del s1.scopes[depth - 1:]
            if c == "brk":
                yield s1, None
            elif isinstance(c, tuple):
                yield s1, c
            else:
                yield from self.range_unrolled(var, vals, i + 1, body, s1)


# ---------------------------------------------------------------------- printing / comparing
def show(x, depth=0):
    if isinstance(x, tuple) and x:
        k = x[0]
        if k == "lit":
            return repr(x[2]) + {"float": "f", "uint": "u"}.get(x[1], "")
        if k == "id":
            return x[1]
        if k == "arg":
            return "$%d" % x[1]
        if k == "bin":
            return "(%s %s %s)" % (show(x[2]), x[1], show(x[3]))
        if k == "un":
            return "%s%s" % (x[1], show(x[2]))
        if k == "member":
            return "%s.%s" % (show(x[1]), x[2])
        if k == "index":
            return "%s[%s]" % (show(x[1]), show(x[2]))
        if k == "call":
            return "%s(%s)" % (x[1], ", ".join(show(a) for a in x[2]))
        if k == "mcall":
            return "%s.%s(%s)" % (show(x[1]), x[2], ", ".join(show(a) for a in x[3]))
        if k == "tern":
            return "(%s ? %s : %s)" % (show(x[1]), show(x[2]), show(x[3]))
        if k in ("vec3", "vec2", "mat3", "pair"):
            return "%s(%s)" % (k, ", ".join(show(a) for a in x[1:]))
        if k == "lv" and len(x) == 3 and isinstance(x[1], int):
            return "<%s@loop%d>" % (x[2], x[1])
        if k == "iter" and len(x) == 2 and isinstance(x[1], int):
            return "<elem@loop%d>" % x[1]
        if k == "at":
            return "%s@%d" % (show(x[2]), x[1])
        if k == "cast":
            return "(%s)%s" % (x[1], show(x[2]))
        return "%s[%s]" % (k, ", ".join(show(a) for a in x[1:]))
    return repr(x)


def first_difference(a, b, path="$"):
    if type(a) != type(b):
        return "%s: %s  vs  %s" % (path, show(a), show(b))
    if isinstance(a, dict):
        for k in sorted(set(a) | set(b), key=str):
            if k not in a or k not in b:
                return "%s: key %r only on one side" % (path, k)
            d = first_difference(a[k], b[k], path + "." + str(k))
            if d:
                return d
        return None
    if isinstance(a, (tuple, list)):
        if len(a) != len(b):
            return "%s: lengths %d vs %d:\n   %s\n   %s" % (path, len(a), len(b), show(a)[:600], show(b)[:600])
        atom = a and isinstance(a[0], str)
        for i, (x, y) in enumerate(zip(a, b)):
            d = first_difference(x, y, path + "/" + str(i))
            if d:
                if atom and len(show(a)) < 400:
                    return "%s:\n   %s\n   %s" % (path, show(a), show(b))
                return d
        return None
    if a != b:
        return "%s: %r vs %r" % (path, a, b)
    return None
