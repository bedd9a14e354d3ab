"""ORACLE / TEST INFRASTRUCTURE -- never imported by the product path.

A small lexer + recursive-descent parser for the GLSL that WebRender's shaders
are once `swgl/build.rs:40-105` has expanded their `#include`s and run them
through the C preprocessor.  It produces plain tuples (the "syntax" tree) that
`glsl_hir.py` lowers the way `glsl-to-cxx/src/hir.rs` lowers the `glsl` crate's
syntax tree.  Only the language subset the reference's shaders use is accepted;
anything else raises, so a silent mistranslation is not possible.
"""
import re

PRIMITIVE_TYPES = {
    "void", "bool", "int", "uint", "float", "double",
    "vec2", "vec3", "vec4", "bvec2", "bvec3", "bvec4",
    "ivec2", "ivec3", "ivec4", "uvec2", "uvec3", "uvec4",
    "mat2", "mat3", "mat4", "mat3x4", "mat4x3",
    "sampler2D", "sampler2DRect", "isampler2D", "sampler2DArray",
}
STORAGE = {"in", "out", "inout", "uniform", "const", "attribute", "varying"}
INTERP = {"flat", "smooth", "noperspective"}
PRECISION = {"highp", "mediump", "lowp"}

_TOKEN = re.compile(r"""
    (?P<ws>\s+|//[^\n]*|/\*.*?\*/)
  | (?P<float>(?:\d+\.\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?|\d+[eE][+-]?\d+)[fF]?)
  | (?P<int>0[xX][0-9a-fA-F]+[uU]?|\d+[uU]?)
  | (?P<id>[A-Za-z_]\w*)
  | (?P<op><<=|>>=|\+\+|--|<<|>>|<=|>=|==|!=|&&|\|\||\^\^|\+=|-=|\*=|/=|%=|&=|\|=|\^=|[-+*/%<>=!~&|^?:;,.(){}\[\]])
""", re.X | re.S)


def lex(src):
    toks = []
    pos = 0
    while pos < len(src):
        m = _TOKEN.match(src, pos)
        if not m:
            raise SyntaxError(f"bad character {src[pos:pos+20]!r}")
        pos = m.end()
        kind = m.lastgroup
        if kind == "ws":
            continue
        toks.append((kind, m.group(kind)))
    toks.append(("eof", ""))
    return toks


class Parser:
    def __init__(self, src):
        self.toks = lex(src)
        self.i = 0
        self.struct_names = set()

    # -- token helpers -----------------------------------------------------
    def peek(self, k=0):
        return self.toks[self.i + k]

    def at(self, text, k=0):
        t = self.toks[self.i + k]
        return t[1] == text and t[0] in ("op", "id")

    def next(self):
        t = self.toks[self.i]
        self.i += 1
        return t

    def expect(self, text):
        t = self.next()
        if t[1] != text:
            ctx = " ".join(x[1] for x in self.toks[max(0, self.i - 8):self.i + 4])
            raise SyntaxError(f"expected {text!r}, got {t[1]!r} near: {ctx}")
        return t

    def accept(self, text):
        if self.at(text):
            self.i += 1
            return True
        return False

    def is_type_name(self, k=0):
        t = self.peek(k)
        return t[0] == "id" and (t[1] in PRIMITIVE_TYPES or t[1] in self.struct_names)

    # -- translation unit ----------------------------------------------------
    def translation_unit(self):
        out = []
        while self.peek()[0] != "eof":
            if self.accept(";"):
                continue
            out.append(self.external_declaration())
        return out

    def qualifiers(self):
        q = {"storage": [], "interp": None, "layout": [], "precision": None}
        seen = False
        while True:
            t = self.peek()
            if t[0] != "id":
                break
            if t[1] == "layout":
                self.next()
                self.expect("(")
                while True:
                    key = self.next()[1]
                    val = None
                    if self.accept("="):
                        val = self.conditional()
                    q["layout"].append((key, val))
                    if not self.accept(","):
                        break
                self.expect(")")
            elif t[1] in STORAGE:
                q["storage"].append(self.next()[1])
            elif t[1] in INTERP:
                if q["interp"] is not None:
                    raise SyntaxError("multiple interpolation")
                q["interp"] = self.next()[1]
            elif t[1] in PRECISION:
                q["precision"] = self.next()[1]
            elif t[1] in ("invariant", "precise"):
                self.next()
            else:
                break
            seen = True
        return q if seen else None

    def type_specifier(self):
        """-> (name, array_size_expr_or_None)"""
        t = self.next()
        if t[0] != "id":
            raise SyntaxError(f"type expected, got {t[1]!r}")
        arr = None
        if self.at("["):
            self.next()
            arr = self.conditional()
            self.expect("]")
        return (t[1], arr)

    def external_declaration(self):
        if self.at("precision"):
            self.next()
            self.next()
            self.next()
            self.expect(";")
            return ("precision",)
        if self.at("struct"):
            return self.struct_declaration()
        quals = self.qualifiers()
        if quals is not None and self.at(";"):
            # `layout(blend_support_all_equations) out;`
            self.next()
            return ("globalqual", quals)
        ty = self.type_specifier()
        name = self.next()
        if name[0] != "id":
            raise SyntaxError(f"declarator expected, got {name[1]!r}")
        if self.at("("):
            params = self.parameters()
            if self.accept(";"):
                return ("proto", quals, ty, name[1], params)
            body = self.compound()
            return ("funcdef", quals, ty, name[1], params, body)
        return self.finish_vardecl(quals, ty, name[1])

    def struct_declaration(self):
        self.expect("struct")
        name = self.next()[1]
        self.expect("{")
        fields = []
        while not self.accept("}"):
            quals = self.qualifiers()
            ty = self.type_specifier()
            while True:
                fname = self.next()[1]
                arr = ty[1]
                if self.accept("["):
                    arr = self.conditional()
                    self.expect("]")
                fields.append(((ty[0], arr), fname, quals))
                if not self.accept(","):
                    break
            self.expect(";")
        self.expect(";")
        self.struct_names.add(name)
        return ("struct", name, fields)

    def finish_vardecl(self, quals, ty, first_name):
        decls = []
        name = first_name
        while True:
            arr = None
            if self.accept("["):
                arr = self.conditional()
                self.expect("]")
            init = None
            if self.accept("="):
                init = self.assignment()
            decls.append((name, arr, init))
            if not self.accept(","):
                break
            name = self.next()[1]
        self.expect(";")
        return ("vardecl", quals, ty, decls)

    def parameters(self):
        self.expect("(")
        params = []
        if self.accept(")"):
            return params
        if self.at("void") and self.at(")", 1):
            self.next()
            self.next()
            return params
        while True:
            quals = self.qualifiers()
            ty = self.type_specifier()
            name = None
            arr = ty[1]
            if self.peek()[0] == "id":
                name = self.next()[1]
                if self.accept("["):
                    arr = self.conditional()
                    self.expect("]")
            params.append((quals, (ty[0], arr), name))
            if not self.accept(","):
                break
        self.expect(")")
        return params

    # -- statements ----------------------------------------------------------
    def compound(self):
        self.expect("{")
        stmts = []
        while not self.accept("}"):
            stmts.append(self.statement())
        return ("compound", stmts)

    def looks_like_declaration(self):
        t = self.peek()
        if t[0] != "id":
            return False
        if t[1] in STORAGE or t[1] in PRECISION or t[1] in INTERP:
            return True
        if not self.is_type_name():
            return False
        k = 1
        if self.at("[", k):
            depth = 0
            while True:
                tt = self.peek(k)
                if tt[1] == "[":
                    depth += 1
                elif tt[1] == "]":
                    depth -= 1
                    if depth == 0:
                        k += 1
                        break
                k += 1
        return self.peek(k)[0] == "id"

    def declaration_statement(self):
        quals = self.qualifiers()
        ty = self.type_specifier()
        name = self.next()[1]
        return self.finish_vardecl(quals, ty, name)

    def statement(self):
        t = self.peek()
        if t[1] == "{" and t[0] == "op":
            return self.compound()
        if t[0] == "id":
            kw = t[1]
            if kw == "if":
                self.next()
                self.expect("(")
                cond = self.expression()
                self.expect(")")
                then = self.statement()
                els = None
                if self.accept("else"):
                    els = self.statement()
                return ("if", cond, then, els)
            if kw == "switch":
                self.next()
                self.expect("(")
                head = self.expression()
                self.expect(")")
                self.expect("{")
                body = []
                while not self.accept("}"):
                    if self.accept("case"):
                        e = self.expression()
                        self.expect(":")
                        body.append(("case", e))
                    elif self.accept("default"):
                        self.expect(":")
                        body.append(("default",))
                    else:
                        body.append(self.statement())
                return ("switch", head, body)
            if kw == "while":
                self.next()
                self.expect("(")
                cond = self.expression()
                self.expect(")")
                return ("while", cond, self.statement())
            if kw == "do":
                self.next()
                body = self.statement()
                self.expect("while")
                self.expect("(")
                cond = self.expression()
                self.expect(")")
                self.expect(";")
                return ("do", body, cond)
            if kw == "for":
                self.next()
                self.expect("(")
                if self.accept(";"):
                    init = ("expr", None)
                elif self.looks_like_declaration():
                    init = ("decl", self.declaration_statement())
                else:
                    init = ("expr", self.expression())
                    self.expect(";")
                cond = None if self.at(";") else self.expression()
                self.expect(";")
                post = None if self.at(")") else self.expression()
                self.expect(")")
                return ("for", init, cond, post, self.statement())
            if kw == "return":
                self.next()
                e = None if self.at(";") else self.expression()
                self.expect(";")
                return ("return", e)
            if kw in ("break", "continue", "discard"):
                self.next()
                self.expect(";")
                return (kw,)
            if self.looks_like_declaration():
                return ("decl", self.declaration_statement())
        if self.accept(";"):
            return ("expr", None)
        e = self.expression()
        self.expect(";")
        return ("expr", e)

    # -- expressions ---------------------------------------------------------
    def expression(self):
        e = self.assignment()
        while self.accept(","):
            e = ("comma", e, self.assignment())
        return e

    ASSIGN_OPS = {"=", "*=", "/=", "%=", "+=", "-=", "<<=", ">>=", "&=", "^=", "|="}

    def assignment(self):
        lhs = self.conditional()
        t = self.peek()
        if t[0] == "op" and t[1] in self.ASSIGN_OPS:
            self.next()
            rhs = self.assignment()
            return ("assign", t[1], lhs, rhs)
        return lhs

    def conditional(self):
        c = self.binary(0)
        if self.accept("?"):
            a = self.expression()
            self.expect(":")
            b = self.assignment()
            return ("ternary", c, a, b)
        return c

    LEVELS = [["||"], ["^^"], ["&&"], ["|"], ["^"], ["&"], ["==", "!="],
              ["<", ">", "<=", ">="], ["<<", ">>"], ["+", "-"], ["*", "/", "%"]]

    def binary(self, level):
        if level == len(self.LEVELS):
            return self.unary()
        lhs = self.binary(level + 1)
        while True:
            t = self.peek()
            if t[0] == "op" and t[1] in self.LEVELS[level]:
                self.next()
                rhs = self.binary(level + 1)
                lhs = ("binary", t[1], lhs, rhs)
            else:
                return lhs

    def unary(self):
        t = self.peek()
        if t[0] == "op" and t[1] in ("+", "-", "!", "~", "++", "--"):
            self.next()
            return ("unary", t[1], self.unary())
        return self.postfix()

    def postfix(self):
        e = self.primary()
        while True:
            if self.accept("["):
                idx = self.expression()
                self.expect("]")
                e = ("bracket", e, idx)
            elif self.at("(") and e[0] in ("var", "bracket"):
                self.next()
                args = []
                if not self.accept(")"):
                    if self.at("void") and self.at(")", 1):
                        self.next()
                    else:
                        while True:
                            args.append(self.assignment())
                            if not self.accept(","):
                                break
                    self.expect(")")
                e = ("call", e, args)
            elif self.accept("."):
                e = ("dot", e, self.next()[1])
            elif self.accept("++"):
                e = ("postinc", e)
            elif self.accept("--"):
                e = ("postdec", e)
            else:
                return e

    def primary(self):
        t = self.next()
        if t[0] == "int":
            s = t[1]
            if s[-1] in "uU":
                return ("uint", int(s[:-1], 0))
            return ("int", int(s, 0))
        if t[0] == "float":
            s = t[1]
            if s[-1] in "fF":
                return ("float", float(s[:-1]))
            return ("double", float(s))
        if t[0] == "id":
            if t[1] in ("true", "false"):
                return ("bool", t[1] == "true")
            return ("var", t[1])
        if t[1] == "(":
            e = self.expression()
            self.expect(")")
            return e
        raise SyntaxError(f"unexpected token {t[1]!r}")


def parse(src):
    return Parser(src).translation_unit()
