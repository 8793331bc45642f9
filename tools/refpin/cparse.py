"""A parser for the subset of C++ that the pinned functions of SdfLib (and their restatements in oracle/) are written in.

Dev-time test infrastructure: it READS function text (from /root/reference, or from oracle/*.h) and turns it into a
statement / expression tree for tools/refpin/symex.py.  Nothing of the reference is stored.

Expression nodes (tuples):
  ('lit', kind, value)        kind in int|uint|float|double|bool|str        ('id', name)
  ('bin', op, l, r)  ('un', op, x)  ('post', op, x)  ('assign', op, lhs, rhs)  ('tern', c, a, b)
  ('call', fn_expr, [args])   ('member', obj, name)   ('index', obj, idx)
  ('cast', type_text, x)      ('brace', type_text or None, [items])         ('lambda', [params], body)
Statements:
  ('block', [stmts]) ('if', c, then, else|None) ('for', init|None, cond|None, step|None, body) ('rangefor', decl, seq, body)
  ('while', c, body) ('do', body, c) ('switch', x, [(label|None, [stmts])]) ('return', x|None) ('break',) ('continue',)
  ('decl', type_text, [(name, [array dims], init|None, init_kind)])   init_kind in None|'='|'('|'{'
  ('expr', x) ('empty',)
"""
import re

TOKEN_RE = re.compile(r"""
    (?P<ws>\s+)
  | (?P<num>0[xX][0-9a-fA-F]+[uUlL]*|0[bB][01]+[uUlL]*|(?:\d+\.\d*|\.\d+|\d+)(?:[eE][+-]?\d+)?[fFuUlL]*)
  | (?P<id>[A-Za-z_]\w*)
  | (?P<str>"(?:\\.|[^"\\])*")
  | (?P<chr>'(?:\\.|[^'\\])')
  | (?P<op><<=|>>=|->|\+\+|--|<<|>>|<=|>=|==|!=|&&|\|\||\+=|-=|\*=|/=|%=|&=|\|=|\^=|::|[-+*/%<>=!&|^~?:;,.(){}\[\]])
""", re.X)

KEYWORD_TYPES = {"float", "double", "int", "unsigned", "bool", "char", "void", "auto", "uint32_t", "uint64_t", "uint8_t",
                 "uint16_t", "int32_t", "int64_t", "size_t", "long", "short"}
QUALIFIERS = {"const", "constexpr", "static", "inline", "volatile", "mutable", "typename"}
# names after which '<' opens a template argument list inside an EXPRESSION (at statement start any IDENT '<' is tried as a type)
TEMPLATE_NAMES = {"static_cast", "reinterpret_cast", "const_cast", "dynamic_cast", "array", "vector", "pair", "map", "numeric_limits",
                  "make_shared", "shared_ptr", "function", "greater", "unordered_map", "make_pair_t"}


def preprocess(text, defined=()):
    """Strip comments; resolve #ifdef/#ifndef/#else/#endif against `defined`; drop the other directives."""
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out, stack = [], []          # stack of (active_before, taking)
    for line in text.split("\n"):
        s = line.strip()
        if s.startswith("#"):
            m = re.match(r"#\s*(ifdef|ifndef|if|else|elif|endif)\b\s*(.*)", s)
            if m:
                d, arg = m.group(1), m.group(2).strip()
                if d in ("ifdef", "ifndef", "if"):
                    on = (arg in defined) if d == "ifdef" else (arg not in defined) if d == "ifndef" else False
                    stack.append([all(t for _, t in stack), on])
                elif d in ("else", "elif"):
                    stack[-1][1] = not stack[-1][1] if d == "else" else False
                else:
                    stack.pop()
            continue
        if all(t for _, t in stack):
            out.append(line)
    return "\n".join(out)


def tokenize(text):
    toks, pos = [], 0
    while pos < len(text):
        m = TOKEN_RE.match(text, pos)
        if not m:
            raise SyntaxError("cannot tokenize at: %r" % text[pos:pos + 40])
        pos = m.end()
        k = m.lastgroup
        if k != "ws":
            toks.append((k, m.group(k)))
    return toks


def parse_number(s):
    t = s.lower()
    if t.startswith("0x"):
        return ("lit", "uint" if "u" in t else "int", int(t.rstrip("ul"), 16))
    if t.startswith("0b"):
        return ("lit", "uint" if "u" in t else "int", int(t.rstrip("ul")[2:], 2))
    if re.fullmatch(r"\d+[ul]*", t):
        return ("lit", "uint" if "u" in t else "int", int(t.rstrip("ul")))
    if t.endswith("f"):
        return ("lit", "float", float(t[:-1]))
    return ("lit", "double", float(t))


class Parser:
    def __init__(self, text, defined=()):
        self.toks = tokenize(preprocess(text, defined))
        self.i = 0

    # ---- token helpers
    def peek(self, k=0):
        j = self.i + k
        return self.toks[j] if j < len(self.toks) else ("eof", "")

    def at(self, val, k=0):
        return self.peek(k)[1] == val and self.peek(k)[0] in ("op", "id")

    def eat(self, val=None):
        t = self.peek()
        if val is not None and t[1] != val:
            raise SyntaxError("expected %r, got %r near %s" % (val, t[1], self.context()))
        self.i += 1
        return t

    def context(self):
        return " ".join(t[1] for t in self.toks[max(0, self.i - 8):self.i + 8])

    # ---- types
    def skip_template_args(self):
        """at '<': consume a balanced <...> (parentheses inside are balanced too) and return its text."""
        depth, out = 0, []
        while True:
            k, v = self.eat()
            out.append(v)
            if v == "<":
                depth += 1
            elif v == ">":
                depth -= 1
                if depth == 0:
                    return " ".join(out)
            elif v == ">>":
                depth -= 2
                if depth <= 0:
                    return " ".join(out)
            elif v == "(":
                d2 = 1
                while d2:
                    k, v = self.eat(); out.append(v)
                    d2 += (v == "(") - (v == ")")
            elif k == "eof":
                raise SyntaxError("unterminated template arguments")

    def try_type(self, free_templates):
        """Parse [qualifiers] name(::name)*[<...>](::name)* [const] [&|*]*; return the type text or None (position restored)."""
        save = self.i
        parts = []
        while self.peek()[0] == "id" and self.peek()[1] in QUALIFIERS:
            self.eat()
        if self.peek()[0] != "id":
            self.i = save; return None
        while True:
            k, v = self.peek()
            if k != "id":
                self.i = save; return None
            self.eat(); parts.append(v)
            if v in ("unsigned", "long", "short") and self.peek()[0] == "id" and self.peek()[1] in KEYWORD_TYPES:
                continue
            if self.at("<") and (free_templates or v in TEMPLATE_NAMES):
                try:
                    parts.append(self.skip_template_args())
                except SyntaxError:
                    self.i = save; return None
            if self.at("::"):
                self.eat(); parts.append("::"); continue
            break
        while self.peek()[0] == "id" and self.peek()[1] in QUALIFIERS:
            self.eat()
        while self.at("&") or self.at("*") or self.at("&&"):
            parts.append(self.eat()[1])
            while self.peek()[0] == "id" and self.peek()[1] in QUALIFIERS:
                self.eat()
        return "".join(parts)

    # ---- expressions
    BIN_PREC = [("||",), ("&&",), ("|",), ("^",), ("&",), ("==", "!="), ("<", ">", "<=", ">="), ("<<", ">>"), ("+", "-"), ("*", "/", "%")]
    ASSIGN_OPS = {"=", "+=", "-=", "*=", "/=", "%=", "&=", "|=", "^=", "<<=", ">>="}

    def expr(self):
        return self.assignment()

    def assignment(self):
        lhs = self.ternary()
        if self.peek()[0] == "op" and self.peek()[1] in self.ASSIGN_OPS:
            op = self.eat()[1]
            rhs = self.brace(None) if self.at("{") else self.assignment()
            return ("assign", op, lhs, rhs)
        return lhs

    def ternary(self):
        c = self.binary(0)
        if self.at("?"):
            self.eat()
            a = self.assignment()
            self.eat(":")
            b = self.assignment()
            return ("tern", c, a, b)
        return c

    def binary(self, level):
        if level == len(self.BIN_PREC):
            return self.unary()
        l = self.binary(level + 1)
        while self.peek()[0] == "op" and self.peek()[1] in self.BIN_PREC[level]:
            op = self.eat()[1]
            r = self.binary(level + 1)
            l = ("bin", op, l, r)
        return l

    def unary(self):
        k, v = self.peek()
        if k == "op" and v in ("-", "+", "!", "~", "*", "&"):
            self.eat()
            return ("un", v, self.unary())
        if k == "op" and v in ("++", "--"):
            self.eat()
            return ("un", v, self.unary())
        if k == "op" and v == "(":
            # C-style cast:  ( type ) unary      -- only for built-in type names
            save = self.i
            self.eat()
            if self.peek()[0] == "id" and (self.peek()[1] in KEYWORD_TYPES or self.peek()[1] in QUALIFIERS):
                t = self.try_type(True)
                if t is not None and self.at(")"):
                    self.eat()
                    return ("cast", t, self.unary())
            self.i = save
        return self.postfix()

    def args(self, close):
        out = []
        while not self.at(close):
            out.append(self.brace(None) if self.at("{") else self.assignment())
            if self.at(","):
                self.eat()
        self.eat(close)
        return out

    def brace(self, type_text):
        self.eat("{")
        return ("brace", type_text, self.args("}"))

    def postfix(self):
        x = self.primary()
        while True:
            if self.at("("):
                self.eat(); x = ("call", x, self.args(")"))
            elif self.at("["):
                self.eat(); i = self.expr(); self.eat("]"); x = ("index", x, i)
            elif self.at(".") or self.at("->"):
                arrow = self.eat()[1] == "->"
                name = self.eat()[1]
                x = ("member", ("un", "*", x) if arrow else x, name)
            elif self.at("++") or self.at("--"):
                x = ("post", self.eat()[1], x)
            elif self.at("{") and x[0] == "id" and self.brace_init_ok(x[1]):
                x = self.brace(x[1])
            else:
                return x

    def brace_init_ok(self, name):
        # `Type{...}` inside an expression (oracle style: V3{a,b,c}); never an `if (x) {` because callers parse conditions in ()
        return bool(re.fullmatch(r"(?:\w+::)*[A-Z]\w*", name)) or name in KEYWORD_TYPES

    def lambda_(self):
        self.eat("[")
        depth = 1
        while depth:
            v = self.eat()[1]
            depth += (v == "[") - (v == "]")
        params = []
        if self.at("("):
            self.eat()
            while not self.at(")"):
                t = self.try_type(True)
                name = self.eat()[1] if self.peek()[0] == "id" else None
                params.append((t, name))
                if self.at(","):
                    self.eat()
            self.eat(")")
        if self.at("->"):
            self.eat(); self.try_type(True)
        body = self.statement()
        return ("lambda", params, body)

    def primary(self):
        k, v = self.peek()
        if k == "num":
            self.eat(); return parse_number(v)
        if k == "str" or k == "chr":
            self.eat(); return ("lit", "str", v)
        if k == "op" and v == "(":
            self.eat(); x = self.expr(); self.eat(")"); return x
        if k == "op" and v == "[":
            return self.lambda_()
        if k == "id":
            if v in ("true", "false"):
                self.eat(); return ("lit", "bool", v == "true")
            if v == "nullptr":
                self.eat(); return ("lit", "int", 0)
            if v == "sizeof":
                self.eat(); self.eat("("); t = self.try_type(True); self.eat(")"); return ("call", ("id", "sizeof"), [("id", t)])
            # qualified name, with template arguments after known template names
            parts = []
            while True:
                kk, vv = self.eat()
                parts.append(vv)
                if self.at("<") and vv in TEMPLATE_NAMES:
                    targs = self.skip_template_args()
                    if vv.endswith("_cast"):
                        self.eat("("); x = self.expr(); self.eat(")")
                        return ("cast", targs[1:-1].strip(), x)
                    parts.append(targs)
                if self.at("::"):
                    self.eat(); parts.append("::"); continue
                break
            name = "".join(parts)
            # functional cast of a built-in type:  float(x)
            if name in KEYWORD_TYPES and self.at("("):
                self.eat(); x = self.expr(); self.eat(")")
                return ("cast", name, x)
            return ("id", name)
        raise SyntaxError("unexpected token %r near %s" % (v, self.context()))

    # ---- statements
    def block_items(self):
        out = []
        while not self.at("}"):
            out.append(self.statement())
        return out

    def declaration(self, allow_range=False):
        """Try `type declarator[, declarator]* ;`.  Returns a node or None (position restored)."""
        save = self.i
        t = self.try_type(True)
        if t is None or self.peek()[0] != "id" or self.peek()[1] in ("operator",):
            self.i = save; return None
        decls = []
        while True:
            if self.peek()[0] != "id":
                self.i = save; return None
            name = self.eat()[1]
            dims = []
            while self.at("["):
                self.eat(); dims.append(None if self.at("]") else self.expr()); self.eat("]")
            init, kind = None, None
            if self.at("="):
                self.eat(); kind = "="
                init = self.brace(None) if self.at("{") else self.assignment()
            elif self.at("("):
                self.eat(); kind = "("; init = self.args(")")
            elif self.at("{"):
                kind = "{"; init = self.brace(None)
            elif allow_range and self.at(":"):
                decls.append((name, dims, None, None))
                return ("decl", t, decls)
            elif not (self.at(";") or self.at(",")):
                self.i = save; return None
            decls.append((name, dims, init, kind))
            if self.at(","):
                self.eat()
                while self.at("*") or self.at("&"):
                    self.eat()
                continue
            break
        if not self.at(";"):
            self.i = save; return None
        self.eat(";")
        return ("decl", t, decls)

    def statement(self):
        k, v = self.peek()
        if k == "op" and v == "{":
            self.eat(); items = self.block_items(); self.eat("}")
            return ("block", items)
        if k == "op" and v == ";":
            self.eat(); return ("empty",)
        if k == "id":
            if v == "if":
                self.eat(); self.eat("("); c = self.expr(); self.eat(")")
                a = self.statement()
                b = None
                if self.at("else"):
                    self.eat(); b = self.statement()
                return ("if", c, a, b)
            if v == "for":
                self.eat(); self.eat("(")
                save = self.i
                d = self.declaration(allow_range=True)
                if d is not None and self.at(":"):
                    self.eat(); seq = self.expr(); self.eat(")")
                    return ("rangefor", d, seq, self.statement())
                self.i = save
                init = None
                if self.at(";"):
                    self.eat()
                else:
                    init = self.declaration()
                    if init is None:
                        init = ("expr", self.comma_expr()); self.eat(";")
                cond = None if self.at(";") else self.expr()
                self.eat(";")
                step = None if self.at(")") else self.comma_expr()
                self.eat(")")
                return ("for", init, cond, step, self.statement())
            if v == "while":
                self.eat(); self.eat("("); c = self.expr(); self.eat(")")
                return ("while", c, self.statement())
            if v == "do":
                self.eat(); body = self.statement(); self.eat("while"); self.eat("("); c = self.expr(); self.eat(")"); self.eat(";")
                return ("do", body, c)
            if v == "switch":
                self.eat(); self.eat("("); x = self.expr(); self.eat(")"); self.eat("{")
                cases = []
                while not self.at("}"):
                    if self.at("case"):
                        self.eat(); lab = self.ternary(); self.eat(":"); cases.append((lab, []))
                    elif self.at("default"):
                        self.eat(); self.eat(":"); cases.append((None, []))
                    else:
                        cases[-1][1].append(self.statement())
                self.eat("}")
                return ("switch", x, cases)
            if v == "return":
                self.eat()
                if self.at(";"):
                    self.eat(); return ("return", None)
                x = self.brace(None) if self.at("{") else self.expr()
                self.eat(";")
                return ("return", x)
            if v == "break":
                self.eat(); self.eat(";"); return ("break",)
            if v == "continue":
                self.eat(); self.eat(";"); return ("continue",)
            d = self.declaration()
            if d is not None:
                return d
        x = self.comma_expr()
        self.eat(";")
        return ("expr", x)

    def comma_expr(self):
        x = self.expr()
        while self.at(","):
            self.eat()
            x = ("bin", ",", x, self.expr())
        return x


def find_function(text, header_regex, nth=0):
    """Return (params_text, body_text) of the nth function whose header matches header_regex (its first '(' opens the parameter list)."""
    ms = list(re.finditer(header_regex, text))
    if len(ms) <= nth:
        raise LookupError("function %r (#%d) not found" % (header_regex, nth))
    i = text.index("(", ms[nth].start())
    depth, j = 0, i
    while True:
        depth += (text[j] == "(") - (text[j] == ")")
        j += 1
        if depth == 0:
            break
    params = text[i + 1:j - 1]
    b = text.index("{", j)
    depth, e = 0, b
    while True:
        depth += (text[e] == "{") - (text[e] == "}")
        e += 1
        if depth == 0:
            break
    return params, text[b:e]


def parse_params(params_text):
    """'glm::vec3 point, const TriangleData& data' -> [('glm::vec3','point'), ...] (default values dropped)."""
    p = Parser(params_text)
    out = []
    while p.peek()[0] != "eof":
        t = p.try_type(True)
        name = p.eat()[1]
        while p.at("["):
            p.eat()
            if not p.at("]"):
                p.expr()
            p.eat("]"); t += "[]"
        if p.at("="):
            p.eat(); p.assignment()
        out.append((t, name))
        if p.at(","):
            p.eat()
    return out


def parse_body(body_text, defined=()):
    p = Parser(body_text, defined)
    s = p.statement()
    assert p.peek()[0] == "eof", p.context()
    return s
