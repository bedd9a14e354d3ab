"""ORACLE / TEST INFRASTRUCTURE -- never imported by the product path.

Lowers the syntax tree of `glsl_parse.py` to a typed, symbol-resolved tree and
infers which values are uniform across the four SIMD lanes swgl shades at once
("run class": scalar / vector / dependent on the caller's arguments), following
`glsl-to-cxx/src/hir.rs`:

  * symbol table, scoping quirks and declaration order   hir.rs:963-1070, 2783-2890
  * storage classes / sampler formats / flat => scalar   hir.rs:1728-1890
  * expression typing (swizzles vs. fields, matrix mult) hir.rs:2199-2610
  * bool <-> int constructor fix-ups                     hir.rs:2157-2197
  * texelFetchOffset bookkeeping                         hir.rs:2115-2139, 2330-2345
  * run-class inference                                  hir.rs:4234-4545

The table of built-in functions (hir.rs:2950-4230) is not restated signature by
signature: only the result's *shape* (vector / ivec / bool / struct / matrix) and
the few run-class overrides matter to the emitted C++, and those are given by
rules below (`native_return_type`, `SCALAR_NATIVES`).
"""
from glsl_parse import PRIMITIVE_TYPES

# ---- run classes (hir.rs:885-902) --------------------------------------------
U, S, V = "U", "S", "V"


def D(mask):
    return ("D", mask)


def is_dep(rc):
    return isinstance(rc, tuple)


def merge(a, b):
    if a == V or b == V:
        return V
    if is_dep(a) and is_dep(b):
        return D(a[1] | b[1])
    if a == U or is_dep(b):
        return b
    return a


# ---- types ---------------------------------------------------------------------
class Type:
    __slots__ = ("kind", "array", "struct")

    def __init__(self, kind, array=None, struct=None):
        self.kind = kind      # GLSL primitive name, or "struct"
        self.array = array    # lowered size expression or None
        self.struct = struct  # Sym of the struct when kind == "struct"

    def same(self, o):
        return self.kind == o.kind and self.struct is o.struct and (self.array is None) == (o.array is None)

    def __repr__(self):
        return f"Type({self.kind}{'[]' if self.array is not None else ''})"


VEC = {"vec2": ("float", 2), "vec3": ("float", 3), "vec4": ("float", 4),
       "ivec2": ("int", 2), "ivec3": ("int", 3), "ivec4": ("int", 4),
       "bvec2": ("bool", 2), "bvec3": ("bool", 3), "bvec4": ("bool", 4),
       "uvec2": ("uint", 2), "uvec3": ("uint", 3), "uvec4": ("uint", 4)}
MAT_COL = {"mat2": "vec2", "mat3": "vec3", "mat4": "vec4", "mat3x4": "vec4", "mat4x3": "vec3"}
SAMPLERS = {"sampler2D", "sampler2DRect", "isampler2D", "sampler2DArray"}


def is_vector(ty):  # hir.rs:1948-1961 (uvec is not a "vector" there)
    return ty.array is None and ty.kind in VEC and VEC[ty.kind][0] != "uint"


def is_ivec(ty):
    return ty.array is None and ty.kind in ("ivec2", "ivec3", "ivec4")


def is_bool_kind(kind):
    return kind in ("bool", "bvec2", "bvec3", "bvec4")


def to_bool(kind):
    if kind in ("int", "uint", "float", "double"):
        return "bool"
    if kind in VEC:
        return "bvec%d" % VEC[kind][1]
    return kind


def to_int(kind):
    if kind in ("bool", "uint", "float", "double"):
        return "int"
    if kind in VEC and VEC[kind][0] != "int":
        return "ivec%d" % VEC[kind][1]
    return kind


def to_scalar(kind):
    if kind in VEC:
        return VEC[kind][0]
    return kind


def promoted_type(l, r):  # hir.rs:2049-2089, shape only
    scal = ("float", "double", "int")
    if is_vector(l) and r.array is None and r.kind in scal:
        return l
    if is_vector(r) and l.array is None and l.kind in scal:
        return r
    if l.kind == "double" or r.kind == "double":
        if l.kind in scal and r.kind in scal:
            return Type("double")
    return l


# ---- symbols -------------------------------------------------------------------
class Sym:
    def __init__(self, id, name, kind, **kw):
        self.id = id
        self.name = name
        self.kind = kind  # native | user | local | global | struct
        self.__dict__.update(kw)

    def __repr__(self):
        return f"Sym({self.id},{self.name},{self.kind})"


class E:
    """Lowered expression: kind + type + operands."""

    def __init__(self, kind, ty, **kw):
        self.kind = kind
        self.ty = ty
        self.__dict__.update(kw)


class Param:
    def __init__(self, qual, ty, name, sym):
        self.qual = qual  # None | const | in | out | inout
        self.ty = ty
        self.name = name
        self.sym = sym


class FunctionDef:
    def __init__(self, ret, name, params):
        self.ret = ret
        self.name = name
        self.params = params
        self.body = ("compound", [])
        self.globals = []
        self.texel_fetches = {}

    def has_parameter(self, sym):
        return any(p.sym is sym for p in self.params)


# natives whose result is uniform whatever their arguments (declare_function_ext
# ... RunClass::Scalar, hir.rs:3478-3500, 3829-3870, 3950-4010)
SCALAR_NATIVES = {"fwidth", "dFdx", "swgl_forceScalar", "anyInvocations", "allInvocations",
                  "allInvocationsEqual", "swgl_interpStep", "swgl_validateGradient",
                  "swgl_isTextureLinear", "swgl_isTextureRGBA8", "swgl_isTextureR8"}
CXX_NAMES = {"anyInvocations": "test_any", "allInvocations": "test_all", "allInvocationsEqual": "test_equal"}
CTOR_TYPES = ["vec2", "vec3", "vec4", "bvec2", "bvec3", "bvec4", "int", "float", "uint", "bool",
              "ivec2", "ivec3", "ivec4", "mat2", "mat3", "mat4", "mat3x4"]
FLOAT_RESULT = {"dot", "length", "distance"}
BVEC_RESULT = {"equal", "notEqual", "lessThan", "lessThanEqual", "greaterThan", "greaterThanEqual"}
SAME_AS_WIDEST = {"abs", "sign", "min", "max", "mix", "step", "clamp", "fwidth", "dFdx", "cos", "sin", "tan",
                  "atan", "pow", "exp", "exp2", "log", "log2", "recip", "inversesqrt", "sqrt", "floor", "ceil",
                  "round", "fract", "mod", "normalize", "inverse", "swgl_forceScalar", "swgl_interpStep",
                  "smoothstep", "radians", "degrees", "cross", "not"}


def native_return_type(name, args):
    if name in FLOAT_RESULT:
        return Type("float")
    if name in BVEC_RESULT:
        k = args[0].ty.kind
        return Type("bvec%d" % VEC[k][1])
    if name in ("any", "all", "anyInvocations", "allInvocations", "allInvocationsEqual",
                "swgl_isTextureLinear", "swgl_isTextureRGBA8", "swgl_isTextureR8"):
        return Type("bool")
    if name in ("texelFetch", "texelFetchOffset"):
        return Type("ivec4" if args[0].ty.kind == "isampler2D" else "vec4")
    if name == "texture":
        return Type("vec4")
    if name == "textureSize":
        return Type("ivec2")
    if name == "swgl_validateGradient":
        return Type("int")
    if name == "transpose":
        k = args[0].ty.kind
        return Type({"mat3x4": "mat4x3", "mat4x3": "mat3x4"}.get(k, k))
    if name == "if_then_else":
        return args[1].ty
    if name in SAME_AS_WIDEST:
        for a in args:
            if a.ty.kind in VEC or a.ty.kind in MAT_COL:
                return Type(a.ty.kind)
        return Type(args[0].ty.kind)
    if name.startswith("swgl_"):
        return Type("void")
    raise KeyError(f"no rule for native function {name}")


class State:
    def __init__(self):
        self.scopes = [{}]
        self.syms = []
        self.in_function = None
        self.run_class_changed = False
        self.last_declaration = 0
        self.branch_run_class = U
        self.branch_declaration = 0
        self.modified_globals = []
        self.used_globals = []
        self.texel_fetches = {}
        self.used_clip_dist = 0
        self.declare_builtins()

    # hir.rs:995-1010
    def lookup(self, name):
        for s in reversed(self.scopes):
            if name in s:
                return s[name]
        return None

    def declare(self, name, kind, **kw):
        sym = Sym(len(self.syms), name, kind, **kw)
        self.syms.append(sym)
        self.scopes[-1][name] = sym
        return sym

    def declare_builtins(self):
        for name in CTOR_TYPES:
            self.declare(name, "native", cxx_name="make_" + name, ret_class=U)
        natives = (SAME_AS_WIDEST | FLOAT_RESULT | BVEC_RESULT |
                   {"any", "all", "if_then_else", "texelFetch", "texelFetchOffset", "texture", "textureSize",
                    "transpose", "anyInvocations", "allInvocations", "allInvocationsEqual", "swgl_stepInterp",
                    "swgl_validateGradient", "swgl_isTextureLinear", "swgl_isTextureRGBA8", "swgl_isTextureR8",
                    "swgl_clipMask", "swgl_antiAlias", "swgl_blendDropShadow", "swgl_blendSubpixelText"})
        for name in sorted(natives):
            self.declare(name, "native", cxx_name=CXX_NAMES.get(name),
                         ret_class=S if name in SCALAR_NATIVES else U)
        # hir.rs:3803-3827
        g = lambda n, st, ty, rc: self.declare(n, "global", storage=st, interp=None, ty=ty, run_class=rc)
        g("gl_FragCoord", "in", Type("vec4"), V)
        g("gl_FragColor", "out", Type("vec4"), V)
        g("gl_Position", "out", Type("vec4"), V)
        self.clip_dist_sym = g("gl_ClipDistance", "out", Type("float", array=E("int", Type("int"), value=4)), V)
        g("swgl_SpanLength", "in", Type("int"), S)
        g("swgl_StepSize", "const", Type("int"), S)

    def native(self, name):
        """swgl_commit* and friends are declared on first use (all return void)."""
        sym = self.lookup(name)
        if sym is None and name.startswith("swgl_commit"):
            sym = Sym(len(self.syms), name, "native", cxx_name=None, ret_class=U)
            self.syms.append(sym)
            self.scopes[0][name] = sym
        return sym

    # hir.rs:1012-1020
    def return_run_class(self, rc):
        rc = merge(self.branch_run_class, rc)
        if self.in_function is not None and self.in_function.kind == "user":
            self.in_function.run_class = merge(self.in_function.run_class, rc)

    # hir.rs:1030-1046
    def merge_run_class(self, sym, rc):
        if sym.id <= self.branch_declaration:
            rc = merge(self.branch_run_class, rc)
        old = rc
        if sym.kind == "local":
            old = sym.run_class
            rc = merge(old, rc)
            sym.run_class = rc
        if old != U and old != rc:
            self.run_class_changed = True
        return rc


# ---- lowering --------------------------------------------------------------------
class Lower:
    def __init__(self):
        self.st = State()

    def lift_type(self, spec):
        name, arr = spec
        st = self.st
        array = self.expr(arr) if arr is not None else None
        if name in PRIMITIVE_TYPES:
            return Type(name, array)
        sym = st.lookup(name)
        if sym is None or sym.kind != "struct":
            raise KeyError(f"unknown type {name}")
        return Type("struct", array, sym)

    def translation_unit(self, tu):
        out = []
        for ed in tu:
            k = ed[0]
            if k == "precision":
                continue
            if k == "globalqual":
                for key, _ in ed[1]["layout"]:
                    assert key == "blend_support_all_equations", key
                continue
            if k == "struct":
                fields = [(self.lift_type(ty), fname) for ty, fname, _ in ed[2]]
                sym = self.st.declare(ed[1], "struct", fields=fields)
                out.append(("structdef", sym))
            elif k == "vardecl":
                out.append(("decl", self.variable_declaration(ed, U)))
            elif k == "proto":
                self.prototype(ed[2], ed[3], ed[4])
                out.append(("proto",))
            elif k == "funcdef":
                out.append(("funcdef", self.function_definition(ed)))
            else:
                raise ValueError(k)
        return out

    # hir.rs:1728-1890
    def variable_declaration(self, d, default_run_class):
        _, quals, spec, decls = d
        st = self.st
        name0, arr0, init0 = decls[0]
        ty = self.lift_type(spec)
        if arr0 is not None:
            ty = Type(ty.kind, self.expr(arr0), ty.struct)
        storage = None
        interp = None
        if quals:
            interp = quals["interp"]
            for key, val in quals["layout"]:
                if val is None:
                    if key in ("rgba8", "rgba32f", "rgba32i", "r8", "rg8"):
                        storage = ("sampler", key.upper())
            loc = index = -1
            for key, val in quals["layout"]:
                if val is not None:
                    if key == "location":
                        loc = val[1]
                    elif key == "index":
                        index = val[1]
            if index >= 0:
                assert loc == 0 and index <= 1 and storage is None
                storage = ("fragcolor", index)
            for s in quals["storage"]:
                if isinstance(storage, tuple) and storage[0] == "fragcolor" and s == "out":
                    pass
                elif isinstance(storage, tuple) and storage[0] == "sampler" and s == "uniform":
                    pass
                elif storage is None and s in ("out", "in", "const"):
                    storage = s
                elif storage is None and s == "uniform":
                    storage = ("sampler", None) if ty.kind in SAMPLERS else "uniform"
                else:
                    raise ValueError(f"bad storage {storage} {s}")
        if st.in_function is not None:
            if storage == "const":
                rc = S
            elif storage is None:
                rc = default_run_class
            else:
                raise ValueError(f"bad local storage {storage}")
            mk = lambda n: st.declare(n, "local", storage=storage, ty=ty, run_class=rc)
        else:
            if storage in ("const", "uniform") or (isinstance(storage, tuple) and storage[0] == "sampler"):
                rc = S
            elif (storage in ("in", "out") or (isinstance(storage, tuple) and storage[0] == "fragcolor")) \
                    and interp == "flat":
                rc = S
            else:
                rc = V
            mk = lambda n: st.declare(n, "global", storage=storage, interp=interp, ty=ty, run_class=rc)
        head_sym = mk(name0)
        head_init = self.expr(init0) if init0 is not None else None
        tail = []
        for name, arr, init in decls[1:]:
            assert arr is None, "unhandled array"
            sym = mk(name)
            tail.append((sym, self.expr(init) if init is not None else None))
        return {"sym": head_sym, "ty": ty, "init": head_init, "tail": tail}

    # hir.rs:2783-2854
    def prototype(self, ret_spec, name, params):
        st = self.st
        ret = self.lift_type(ret_spec)
        ps = []
        for index, (quals, spec, pname) in enumerate(params):
            assert pname is not None, "unnamed parameter"
            ty = self.lift_type(spec)
            qual = None
            if quals:
                for s in quals["storage"]:
                    assert qual is None and s in ("const", "in", "out", "inout"), s
                    qual = s
            sym = st.declare(pname, "local", storage=None, ty=ty, run_class=D(1 << index))
            ps.append(Param(qual, ty, pname, sym))
        fd = FunctionDef(ret, name, ps)
        sym = st.lookup(name)
        if sym is not None:
            assert sym.kind == "user", f"prototype conflicts with existing symbol: {name}"
        else:
            sym = st.declare(name, "user", fd=FunctionDef(ret, name, ps), run_class=U)
        return fd, sym

    # hir.rs:2864-2890
    def function_definition(self, ed):
        _, quals, ret_spec, name, params, body = ed
        st = self.st
        fd, sym = self.prototype(ret_spec, name, params)
        st.scopes.append({})
        st.in_function = sym
        st.modified_globals = []
        st.texel_fetches = {}
        fd.body = self.statement(body)
        fd.globals = st.modified_globals
        fd.texel_fetches = st.texel_fetches
        st.modified_globals = []
        st.texel_fetches = {}
        st.in_function = None
        st.scopes.pop()
        sym.fd = fd
        sym.run_class = U
        return fd

    def statement(self, s):
        k = s[0]
        if k == "compound":
            return ("compound", [self.statement(x) for x in s[1]])
        if k == "decl":
            return ("decl", self.variable_declaration(s[1], U))
        if k == "expr":
            return ("expr", self.expr(s[1]) if s[1] is not None else None)
        if k == "if":
            cond = self.expr(s[1])
            then = self.statement(s[2])
            els = self.statement(s[3]) if s[3] is not None else None
            return ("if", cond, then, els)
        if k == "switch":
            cases = []
            case = None
            for st in s[2]:
                if st[0] in ("case", "default"):
                    if case is not None:
                        cases.append(case)
                    case = {"label": self.expr(st[1]) if st[0] == "case" else None, "stmts": []}
                else:
                    assert case is not None, "switch must start with case"
                    case["stmts"].append(self.statement(st))
            if case is not None:
                cases.append(case)
            return ("switch", self.expr(s[1]), cases)
        if k == "while":
            return ("while", self.expr(s[1]), self.statement(s[2]))
        if k == "do":
            return ("do", self.statement(s[1]), self.expr(s[2]))
        if k == "for":
            init = s[1]
            if init[0] == "decl":
                init = ("decl", self.variable_declaration(init[1], S))
            else:
                init = ("expr", self.expr(init[1]) if init[1] is not None else None)
            cond = self.expr(s[2]) if s[2] is not None else None
            post = self.expr(s[3]) if s[3] is not None else None
            return ("for", init, cond, post, self.statement(s[4]))
        if k == "return":
            return ("return", self.expr(s[1]) if s[1] is not None else None)
        if k in ("break", "continue", "discard"):
            return (k,)
        raise ValueError(k)

    # hir.rs:2091-2113
    def is_output(self, e):
        if e.kind == "var":
            if e.sym.kind == "global" and e.sym.storage in ("in", "out"):
                return e.sym
            return None
        if e.kind in ("swizzle", "bracket", "dot"):
            return self.is_output(e.e)
        return None

    # hir.rs:2115-2139
    def get_texel_fetch_offset(self, sampler_e, uv_e, off_e):
        if sampler_e.kind == "var" and uv_e.kind == "var" and off_e.kind == "call" and off_e.ctor is None \
                and off_e.fun.name == "ivec2" and len(off_e.args) == 2 \
                and off_e.args[0].kind == "int" and off_e.args[1].kind == "int":
            return (sampler_e.sym, uv_e.sym, off_e.args[0].value, off_e.args[1].value)
        return None

    def make_const(self, kind, v):
        if kind == "int":
            return E("int", Type("int"), value=v)
        if kind == "uint":
            return E("uint", Type("uint"), value=v)
        if kind == "bool":
            return E("bool", Type("bool"), value=v != 0)
        if kind in ("float", "double"):
            return E("float", Type(kind), value=float(v))
        raise ValueError("bad constant type")

    # hir.rs:2199-2610
    def expr(self, e):
        st = self.st
        k = e[0]
        if k == "var":
            sym = st.lookup(e[1])
            if sym is None:
                raise KeyError(f"missing declaration {e[1]}")
            if sym.kind == "global":
                if sym not in st.used_globals:
                    st.used_globals.append(sym)
            elif sym.kind != "local":
                raise ValueError(f"bad variable type {e[1]}")
            return E("var", sym.ty, sym=sym)
        if k == "assign":
            lhs = self.expr(e[2])
            rhs = self.expr(e[3])
            if e[1] == "*=" and lhs.ty.kind == "vec4" and rhs.ty.kind == "float":
                ty = lhs.ty
            else:
                ty = promoted_type(lhs.ty, rhs.ty)
            g = self.is_output(lhs)
            if g is not None:
                if g not in st.modified_globals:
                    st.modified_globals.append(g)
                if g is st.clip_dist_sym and lhs.kind == "bracket":
                    idx = lhs.index
                    assert idx.kind in ("int", "uint") and 0 <= idx.value < 4, "bad index for gl_ClipDistance"
                    st.used_clip_dist |= 1 << idx.value
            return E("assign", ty, op=e[1], lhs=lhs, rhs=rhs)
        if k == "binary":
            op = e[1]
            lhs = self.expr(e[2])
            rhs = self.expr(e[3])
            if op in ("==", "!=", ">", ">=", "<", "<="):
                ty = Type("bool")
            elif op == "*":
                lk, rk = lhs.ty.kind, rhs.ty.kind
                if (lk, rk) in (("mat2", "vec2"), ("mat3", "vec3"), ("mat3", "mat3"), ("mat3", "mat4x3"),
                                ("mat4", "vec4")):
                    ty = rhs.ty
                elif (lk, rk) == ("mat4x3", "vec4"):
                    ty = Type("vec3")
                elif lk in ("mat2", "mat3", "mat4") and rk == "float":
                    ty = lhs.ty
                else:
                    ty = promoted_type(lhs.ty, rhs.ty)
            else:
                ty = promoted_type(lhs.ty, rhs.ty)
            return E("binary", ty, op=op, lhs=lhs, rhs=rhs)
        if k == "unary":
            x = self.expr(e[2])
            return E("unary", x.ty, op=e[1], e=x)
        if k == "bool":
            return E("bool", Type("bool"), value=e[1])
        if k == "comma":
            a = self.expr(e[1])
            b = self.expr(e[2])
            return E("comma", a.ty, a=a, b=b)
        if k == "double":
            return E("float", Type("double"), value=e[1])
        if k == "float":
            return E("float", Type("float"), value=e[1])
        if k == "int":
            return E("int", Type("int"), value=e[1])
        if k == "uint":
            return E("uint", Type("uint"), value=e[1])
        if k == "postinc" or k == "postdec":
            x = self.expr(e[1])
            return E(k, x.ty, e=x)
        if k == "ternary":
            c = self.expr(e[1])
            a = self.expr(e[2])
            b = self.expr(e[3])
            return E("ternary", promoted_type(a.ty, b.ty), c=c, a=a, b=b)
        if k == "dot":
            x = self.expr(e[1])
            name = e[2]
            ty = x.ty
            if is_vector(ty):
                base = "int" if is_ivec(ty) else "float"
                n = len(name)
                assert 1 <= n <= 4
                if n == 1:
                    rty = Type(base)
                else:
                    rty = Type(("ivec%d" if base == "int" else "vec%d") % n)
                comps = []
                sets = []
                for c in name:
                    for fs in ("rgba", "xyzw", "stpq"):
                        if c in fs:
                            comps.append(fs.index(c))
                            sets.append(fs)
                            break
                    else:
                        raise ValueError(f"bad selector {name}")
                assert all(s == sets[0] for s in sets)
                return E("swizzle", rty, e=x, comps=comps, text=name)
            if ty.kind == "struct" and ty.array is None:
                for fty, fname in ty.struct.fields:
                    if fname == name:
                        return E("dot", fty, e=x, name=name)
                raise KeyError(f"missing field `{name}` in `{ty.struct.name}`")
            raise ValueError(f"expected struct, found {ty} for .{name}")
        if k == "bracket":
            x = self.expr(e[1])
            ty = x.ty
            if ty.array is None and ty.kind in VEC:
                rty = Type(VEC[ty.kind][0])
            elif ty.array is None and ty.kind in MAT_COL:
                rty = Type(MAT_COL[ty.kind])
            else:
                assert ty.array is not None, f"indexing {ty}"
                rty = Type(ty.kind, None, ty.struct)
            return E("bracket", rty, e=x, index=self.expr(e[2]))
        if k == "call":
            fun = e[1]
            args = [self.expr(a) for a in e[2]]
            if fun[0] == "bracket":
                # array constructor `vec4[2](...)`
                assert fun[1][0] == "var" and fun[1][1] in ("vec4", "vec2", "int"), fun
                ty = Type(fun[1][1], self.expr(fun[2]))
                return E("call", ty, fun=None, ctor=ty, args=args)
            name = fun[1]
            if name == "texelFetchOffset" and len(args) >= 4:
                tf = self.get_texel_fetch_offset(args[0], args[1], args[3])
                if tf is not None:
                    sampler, base, x, y = tf
                    key = (sampler, base)
                    if key in st.texel_fetches:
                        o = st.texel_fetches[key]
                        o[0] = min(o[0], x)
                        o[1] = max(o[1], x)
                        o[2] = min(o[2], y)
                        o[3] = max(o[3], y)
                    else:
                        st.texel_fetches[key] = [x, x, y, y]
            elif name == "swgl_stepInterp":
                for sym in st.syms:
                    if sym.kind == "global" and sym.storage == "in" and sym.run_class == V:
                        if sym not in st.modified_globals:
                            st.modified_globals.append(sym)
            sym = st.lookup(name) or st.native(name)
            if sym is None:
                raise KeyError(f"missing symbol {name}")
            if name in PRIMITIVE_TYPES:
                if is_bool_kind(name):
                    # hir.rs:2157-2174
                    for i, a in enumerate(args):
                        if not is_bool_kind(a.ty.kind):
                            kk = a.ty.kind
                            args[i] = E("binary", Type(to_bool(kk)), op="!=", lhs=a,
                                        rhs=self.make_const(to_scalar(kk), 0))
                else:
                    # hir.rs:2176-2197
                    for i, a in enumerate(args):
                        if is_bool_kind(a.ty.kind):
                            kk = to_int(a.ty.kind)
                            conv = st.lookup(kk)
                            assert conv is not None, kk
                            inner = E("call", Type(kk), fun=conv, ctor=None, args=[a])
                            args[i] = E("binary", Type(kk), op="&", lhs=inner, rhs=self.make_const("int", 1))
            if sym.kind == "native":
                if name in PRIMITIVE_TYPES:
                    rty = Type(name)
                else:
                    rty = native_return_type(name, args)
            elif sym.kind == "user":
                fd = sym.fd
                for g in fd.globals:
                    if g not in st.modified_globals:
                        st.modified_globals.append(g)
                for a, p in zip(args, fd.params):
                    if p.qual in ("inout", "out"):
                        g = self.is_output(a)
                        if g is not None and g not in st.modified_globals:
                            st.modified_globals.append(g)
                rty = fd.ret
            elif sym.kind == "struct":
                rty = Type("struct", None, sym)
            else:
                raise ValueError(f"can only call functions: {name}")
            return E("call", rty, fun=sym, ctor=None, args=args)
        raise ValueError(k)


# ---- run-class inference (hir.rs:4234-4545) -----------------------------------------
class Infer:
    def __init__(self, st):
        self.st = st

    def expr_inner(self, e, assign):
        """returns (run_class, assigned_sym)"""
        st = self.st
        k = e.kind
        if k == "var":
            return e.sym.run_class, e.sym
        if k in ("int", "uint", "bool", "float"):
            return S, assign
        if k == "unary":
            return self.expr(e.e), assign
        if k == "binary":
            return merge(self.expr(e.lhs), self.expr(e.rhs)), assign
        if k == "ternary":
            return merge(merge(self.expr(e.c), self.expr(e.a)), self.expr(e.b)), assign
        if k == "assign":
            rc_v, sym = self.expr_inner(e.lhs, None)
            rc = merge(rc_v, self.expr(e.rhs))
            assert sym is not None
            return st.merge_run_class(sym, rc), assign
        if k == "bracket":
            rc, assign = self.expr_inner(e.e, assign)
            return merge(rc, self.expr(e.index)), assign
        if k == "call":
            arg_classes = [self.expr_inner(a, None) for a in e.args]
            if not e.args:
                run_class = S
            else:
                run_class = U
                for rc, _ in arg_classes:
                    run_class = merge(run_class, rc)
            if e.ctor is not None:
                return run_class, assign
            fun = e.fun
            if fun.kind == "native":
                return (fun.ret_class if fun.ret_class != U else run_class), assign
            if fun.kind == "user":
                fd = fun.fd
                for (arg_class, asym), p in zip(arg_classes, fd.params):
                    if p.qual in ("inout", "out"):
                        pc = p.sym.run_class
                        if pc == U or pc == V:
                            arg_class = V
                        elif is_dep(pc):
                            for i in range(31):
                                if pc[1] & (1 << i):
                                    arg_class = merge(arg_class, arg_classes[i][0])
                        assert asym is not None
                        st.merge_run_class(asym, arg_class)
                if fd.ret.kind == "void" and fd.ret.array is None:
                    return S, assign
                frc = fun.run_class
                if frc == U or frc == V:
                    return V, assign
                if is_dep(frc):
                    ret = U
                    for i in range(31):
                        if frc[1] & (1 << i):
                            ret = merge(ret, arg_classes[i][0])
                    return ret, assign
                return S, assign
            if fun.kind == "struct":
                return run_class, assign
            raise ValueError(fun)
        if k in ("dot", "swizzle", "postinc", "postdec"):
            return self.expr_inner(e.e, assign)
        if k == "comma":
            self.expr(e.a)
            return self.expr(e.b), assign
        raise ValueError(k)

    def expr(self, e):
        return self.expr_inner(e, None)[0]

    def declaration(self, d):
        st = self.st
        st.last_declaration = d["sym"].id
        rc = U
        for _, init in d["tail"]:
            if init is not None:
                rc = merge(rc, self.expr(init))
        if d["init"] is not None:
            rc = merge(rc, self.expr(d["init"]))
            st.merge_run_class(d["sym"], rc)

    def statement(self, s):
        st = self.st
        k = s[0]
        if k == "compound":
            for x in s[1]:
                self.statement(x)
        elif k == "decl":
            self.declaration(s[1])
        elif k == "expr":
            if s[1] is not None:
                self.expr(s[1])
        elif k == "if":
            saved = st.branch_run_class
            st.branch_run_class = merge(saved, self.expr(s[1]))
            saved_decl = st.branch_declaration
            st.branch_declaration = st.last_declaration
            self.statement(s[2])
            if s[3] is not None:
                self.statement(s[3])
            st.branch_run_class = saved
            st.branch_declaration = saved_decl
        elif k == "switch":
            saved = st.branch_run_class
            st.branch_run_class = merge(saved, self.expr(s[1]))
            saved_decl = st.branch_declaration
            st.branch_declaration = st.last_declaration
            for case in s[2]:
                for x in case["stmts"]:
                    self.statement(x)
            st.branch_run_class = saved
            st.branch_declaration = saved_decl
        elif k in ("while", "do", "for"):
            changed = st.run_class_changed
            st.run_class_changed = True
            if k == "while":
                while st.run_class_changed:
                    st.run_class_changed = False
                    self.expr(s[1])
                    self.statement(s[2])
            elif k == "do":
                while st.run_class_changed:
                    st.run_class_changed = False
                    self.statement(s[1])
                    self.expr(s[2])
            else:
                init = s[1]
                if init[0] == "decl":
                    self.declaration(init[1])
                elif init[1] is not None:
                    self.expr(init[1])
                while st.run_class_changed:
                    st.run_class_changed = False
                    if s[2] is not None:
                        self.expr(s[2])
                    if s[3] is not None:
                        self.expr(s[3])
                    self.statement(s[4])
            st.run_class_changed = changed
        elif k == "return":
            if s[1] is not None:
                st.return_run_class(self.expr(s[1]))
        elif k in ("break", "continue", "discard"):
            pass
        else:
            raise ValueError(k)

    def run(self, tu):
        st = self.st
        for ed in tu:
            if ed[0] != "funcdef":
                continue
            fd = ed[1]
            st.in_function = st.lookup(fd.name)
            st.run_class_changed = True
            while st.run_class_changed:
                st.run_class_changed = False
                for x in fd.body[1]:
                    self.statement(x)
            st.in_function = None
