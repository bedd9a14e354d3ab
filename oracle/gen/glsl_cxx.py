"""ORACLE / TEST INFRASTRUCTURE -- never imported by the product path.

Emits, for one shader key, the C++ header swgl compiles: `NAME_common`,
`NAME_vert`, `NAME_frag`, `NAME_program` on top of the reference's own
`glsl.h` / `program.h` / `swgl_ext.h` types.  This is a restatement of the
emitter half of the reference's shader translator,
`glsl-to-cxx/src/lib.rs`:

  * struct layout, uniform / attribute / interpolant plumbing   lib.rs:120-743
  * scalar vs. vector spelling of types and constructors        lib.rs:1061-1170
  * expressions: swizzles -> `.sel()`, masked assignment,
    out-parameter adaptation, texelFetchOffset pointers         lib.rs:1495-2040
  * declarations, function instantiation per argument mask      lib.rs:2290-2640, 3490-3540
  * `if` on a per-lane condition -> masks, masked `return`,
    `discard`                                                   lib.rs:2760-2860, 3230-3330
  * ABI thunks and constructor wiring                           lib.rs:3560-3650

The generated text is a build product (oracle/_ref/gen/, git-ignored): it is
derived from the reference's GLSL and is never committed.
"""
from glsl_hir import (E, Type, U, S, V, is_dep, merge, Lower, Infer, VEC, is_bool_kind)

CXX_VECTOR_NAME = {"bool": "Bool", "int": "I32", "uint": "U32", "float": "Float", "double": "Double"}
CXX_SCALAR_NAME = {"void": "void", "bool": "bool", "int": "int32_t", "uint": "uint32_t", "float": "float",
                   "double": "double"}
SAMPLER_KINDS = ("sampler2D", "sampler2DRect", "isampler2D", "sampler2DArray")
GLSL_KIND = {"mat3x4": "mat34", "mat4x3": "mat43"}  # hir.rs glsl_primitive_type_name


def cxx_primitive_type_name(kind):
    return CXX_VECTOR_NAME.get(kind, GLSL_KIND.get(kind, kind))


def cxx_primitive_scalar_type_name(kind):
    if kind in CXX_SCALAR_NAME:
        return CXX_SCALAR_NAME[kind]
    if kind in SAMPLER_KINDS:
        return kind
    return None


def symbol_run_class(sym, vector_mask):  # lib.rs:2484-2500
    rc = sym.run_class if sym.kind in ("global", "local") else V
    if rc == S:
        return S
    if is_dep(rc):
        return V if (rc[1] & vector_mask) else S
    return V


def fmt_float(x):  # lib.rs:1464-1480
    if x == int(x) and abs(x) < 1e300:
        return "%d.f" % int(x)
    r = repr(float(x))
    return r + "f"


class Out:
    """OutputState of lib.rs:743-812 (the C++ half only)."""

    def __init__(self, st, name, is_frag):
        self.hir = st
        self.name = name
        self.is_frag = is_frag
        self.output = []
        self.buffer = []
        self.indent = 0
        self.mask = None
        self.cond_index = 0
        self.return_type = None
        self.return_declared = False
        self.return_vector = False
        self.is_scalar = False
        self.is_lval = False
        self.functions = {}
        self.deps = []
        self.vector_mask = 0
        self.uses_discard = False
        self.used_fragcoord = 0
        self.use_perspective = False
        self.used_globals = []
        self.texel_fetches = []

    def w(self, s):
        self.buffer.append(s)

    def flush_buffer(self):
        self.output.extend(self.buffer)
        self.buffer = []

    def push_buffer(self):
        b = self.buffer
        self.buffer = []
        return b

    def pop_buffer(self, b):
        old = self.buffer
        self.buffer = b
        return old

    def show_indent(self):
        self.w(" " * self.indent)

    def add_used_global(self, sym):
        if sym not in self.used_globals:
            self.used_globals.append(sym)

    # ---- types (lib.rs:1061-1135) ---------------------------------------------
    def show_type_kind(self, ty):
        kind = ty.kind
        if kind == "struct":
            self.w(ty.struct.name + ("_scalar" if self.is_scalar else ""))
            return
        if self.is_scalar:
            n = cxx_primitive_scalar_type_name(kind)
            if n is not None:
                self.w(n)
            else:
                self.w(cxx_primitive_type_name(kind) + "_scalar")
        else:
            self.w(cxx_primitive_type_name(kind))

    def show_type(self, ty):
        if ty.array is not None:
            self.w("Array<")
            self.show_type_kind(ty)
            self.w(",")
            self.show_expr(ty.array)
            self.w(">")
        else:
            self.show_type_kind(ty)

    def with_scalar(self, flag, fn):
        old = self.is_scalar
        self.is_scalar = flag
        fn()
        self.is_scalar = old

    # ---- struct definitions (lib.rs:885-1060) ------------------------------------
    def show_struct_field(self, fty, fname):
        self.show_type(fty)
        self.w(" " + fname + ";\n")

    def write_constructor(self, name, fields):
        if len(fields) == 1:
            self.w("explicit ")
        self.w(name + "(")
        for i, (fty, fname) in enumerate(fields):
            if i:
                self.w(", ")
            self.show_type(fty)
            self.w(" " + fname)
        self.w(") : ")
        self.w(", ".join(f"{fname}({fname})" for _, fname in fields))
        self.w("{}\n")

    def write_convert_constructor(self, name, fields):
        if len(fields) == 1:
            self.w("explicit ")
        self.w(name + "(")
        for i, (fty, fname) in enumerate(fields):
            if i:
                self.w(", ")
            self.with_scalar(True, lambda: self.show_type(fty))
            self.w(" " + fname)
        self.w(")")
        first = True
        for fty, fname in fields:
            if fty.array is None:
                self.w(":" if first else ",")
                self.w(f"{fname}({fname})")
                first = False
        self.w("{\n")
        for fty, fname in fields:
            if fty.array is not None:
                self.w(f"this->{fname}.convert({fname});\n")
        self.w("}\n")
        self.w(f"IMPLICIT {name}({name}_scalar s)")
        first = True
        for fty, fname in fields:
            if fty.array is None:
                self.w(":" if first else ",")
                self.w(f"{fname}(s.{fname})")
                first = False
        self.w("{\n")
        for fty, fname in fields:
            if fty.array is not None:
                self.w(f"{fname}.convert(s.{fname});\n")
        self.w("}\n")

    def show_struct(self, sym):
        name = sym.name
        fields = sym.fields
        sname = name + "_scalar"
        self.w(f"struct {sname} {{\n")
        old = self.is_scalar
        self.is_scalar = True
        for fty, fname in fields:
            self.show_struct_field(fty, fname)
        self.w(f"{sname}() = default;\n")
        self.write_constructor(sname, fields)
        self.is_scalar = old
        self.w("};\n")
        self.w(f"struct {name} {{\n")
        for fty, fname in fields:
            self.show_struct_field(fty, fname)
        self.w(f"{name}() = default;\n")
        self.write_constructor(name, fields)
        self.write_convert_constructor(name, fields)
        self.w(f"friend {name} if_then_else(I32 c, {name} t, {name} e) {{ return {name}(\n")
        self.w(", ".join(f"if_then_else(c, t.{fname}, e.{fname})" for _, fname in fields))
        self.w(");\n}")
        self.w("}")

    # ---- symbols ---------------------------------------------------------------------
    def show_sym(self, sym):
        if sym.kind == "native":
            self.w(sym.cxx_name or sym.name)
        elif sym.kind == "global":
            self.add_used_global(sym)
            self.w(sym.name)
        else:
            self.w(sym.name)

    # ---- run class of an expression under the current instantiation (lib.rs:1482-1590)
    def expr_run_class(self, e):
        k = e.kind
        if k == "var":
            return symbol_run_class(e.sym, self.vector_mask)
        if k in ("int", "uint", "bool", "float"):
            return S
        if k == "unary":
            return self.expr_run_class(e.e)
        if k == "binary":
            return merge(self.expr_run_class(e.lhs), self.expr_run_class(e.rhs))
        if k == "ternary":
            return merge(merge(self.expr_run_class(e.c), self.expr_run_class(e.a)), self.expr_run_class(e.b))
        if k == "assign":
            return merge(self.expr_run_class(e.lhs), self.expr_run_class(e.rhs))
        if k == "bracket":
            return merge(self.expr_run_class(e.e), self.expr_run_class(e.index))
        if k == "call":
            arg_mask = 0
            for idx, a in enumerate(e.args):
                if self.expr_run_class(a) == V:
                    arg_mask |= 1 << idx
            if e.ctor is not None:
                return V if arg_mask else S
            fun = e.fun
            if fun.kind == "native":
                if fun.ret_class != U:
                    return fun.ret_class
                return V if arg_mask else S
            if fun.kind == "user":
                param_mask = arg_mask
                for idx, p in enumerate(fun.fd.params):
                    if p.qual in ("inout", "out"):
                        if symbol_run_class(p.sym, arg_mask) == V:
                            param_mask |= 1 << idx
                rc = fun.run_class
                if rc == S:
                    return S
                if is_dep(rc):
                    return V if (rc[1] & param_mask) else S
                return V
            if fun.kind == "struct":
                return V if arg_mask else S
            raise ValueError(fun)
        if k in ("dot", "swizzle", "postinc", "postdec"):
            return self.expr_run_class(e.e)
        if k == "comma":
            return self.expr_run_class(e.b)
        if k == "cond":
            return self.expr_run_class(e.e)
        if k == "condmask":
            return V
        raise ValueError(k)

    # ---- expressions (lib.rs:1592-2040) ---------------------------------------------
    ASSIGN_BIN = {"=": "", "*=": "*", "/=": "/", "%=": "%", "+=": "+", "-=": "-", "<<=": "<<", ">>=": ">>",
                  "&=": "&", "^=": "^", "|=": "|"}

    def is_output(self, e):
        if e.kind == "var":
            if e.sym.kind == "global" and e.sym.storage in ("in", "out"):
                return e.sym
            return None
        if e.kind in ("swizzle", "bracket", "dot"):
            return self.is_output(e.e)
        return None

    def show_lval(self, e):
        self.is_lval = True
        self.show_expr(e)
        self.is_lval = False

    def show_expr(self, e, top_level=False):
        k = e.kind
        w = self.w
        if k == "var":
            self.show_sym(e.sym)
        elif k == "int":
            w("%d" % e.value)
        elif k == "uint":
            w("%du" % e.value)
        elif k == "bool":
            w("true" if e.value else "false")
        elif k == "float":
            w(fmt_float(e.value))
        elif k == "unary":
            w(e.op)
            w("(")
            self.show_expr(e.e)
            w(")")
        elif k == "binary":
            w("(")
            self.show_expr(e.lhs)
            w(")")
            w(e.op)
            w("(")
            self.show_expr(e.rhs)
            w(")")
        elif k == "ternary":
            if self.expr_run_class(e.c) != S:
                w("if_then_else(")
                self.show_expr(e.c)
                w(", ")
                self.show_expr(e.a)
                w(", ")
                self.show_expr(e.b)
                w(")")
            else:
                self.show_expr(e.c)
                w(" ? ")
                self.show_expr(e.a)
                w(" : ")
                self.show_expr(e.b)
        elif k == "assign":
            self.show_assignment(e, top_level)
        elif k == "bracket":
            self.show_expr(e.e)
            w("[")
            self.show_expr(e.index)
            w("]")
        elif k == "call":
            self.show_call(e, top_level)
        elif k == "dot":
            w("(")
            self.show_expr(e.e)
            w(")")
            w(".")
            w(e.name)
        elif k == "swizzle":
            if e.e.kind == "var" and e.e.sym.name == "gl_FragCoord":
                for c in e.comps:
                    self.used_fragcoord |= 1 << c
            w("(")
            self.show_expr(e.e)
            w(").")
            if len(e.comps) == 1:
                w("xyzw"[e.comps[0]])
            else:
                w("lsel(" if self.is_lval else "sel(")
                w(",".join(c.upper() for c in e.text))
                w(")")
        elif k == "postinc":
            self.show_expr(e.e)
            w("++")
        elif k == "postdec":
            self.show_expr(e.e)
            w("--")
        elif k == "comma":
            self.show_expr(e.a)
            w(", ")
            self.show_expr(e.b)
        elif k == "cond":
            w("_c%d_" % e.index)
        elif k == "condmask":
            w("_cond_mask_")
        else:
            raise ValueError(k)

    def show_assignment(self, e, top_level):
        w = self.w
        v, op, rhs = e.lhs, e.op, e.rhs
        is_output = self.is_output(v) is not None
        is_scalar_var = self.expr_run_class(v) == S
        is_scalar_expr = self.expr_run_class(rhs) == S
        force_scalar = is_scalar_var and not is_scalar_expr
        if self.mask is not None:
            mask = self.mask
            is_scalar_mask = self.expr_run_class(mask) == S
            force_scalar_mask = is_scalar_var and is_scalar_expr and not is_scalar_mask
            if force_scalar or force_scalar_mask:
                w("if (" if top_level else "(")
            else:
                self.show_lval(v)
                w(" = if_then_else(")
            if is_output and self.return_declared:
                w("((")
                self.show_expr(mask)
                w(")&ret_mask)")
            else:
                self.show_expr(mask)
            if force_scalar or force_scalar_mask:
                w("[0]) { " if top_level else "[0] ? ")
                self.show_lval(v)
                w(" = ")
            else:
                w(",")
            if op != "=":
                self.show_expr(v)
            w(self.ASSIGN_BIN[op])
            if force_scalar:
                w("force_scalar(")
            self.show_expr(rhs)
            if force_scalar:
                w(")")
            if force_scalar or force_scalar_mask:
                if top_level:
                    w("; }")
                else:
                    w(" : ")
                    self.show_expr(v)
                    w(")")
            else:
                w(",")
                self.show_expr(v)
                w(")")
        else:
            self.show_lval(v)
            w(" ")
            if is_output and self.return_declared:
                w("= ")
                if force_scalar:
                    w("force_scalar(")
                w("if_then_else(ret_mask,")
                if op != "=":
                    self.show_expr(v)
                w(self.ASSIGN_BIN[op])
                self.show_expr(rhs)
                w(",")
                self.show_expr(v)
                w(")")
            else:
                w(op)
                w(" ")
                if force_scalar:
                    w("force_scalar(")
                self.show_expr(rhs)
            if force_scalar:
                w(")")

    def show_call(self, e, top_level):
        w = self.w
        args = e.args
        cond_mask = 0
        adapt_mask = 0
        has_ret = False
        array_constructor = False
        arg_mask = 0
        for idx, a in enumerate(args):
            if self.expr_run_class(a) == V:
                arg_mask |= 1 << idx
        if e.ctor is not None:
            self.with_scalar(arg_mask == 0, lambda: self.show_type(e.ctor))
            array_constructor = e.ctor.array is not None
        else:
            fun = e.fun
            if fun.kind == "native":
                if fun.name == "texelFetchOffset" and len(args) >= 4:
                    tf = self.texel_fetch_offset(args[0], args[1], args[3])
                    if tf is not None:
                        sampler, base, x, y = tf
                        self.add_used_global(sampler)
                        if base.kind == "global":
                            self.add_used_global(base)
                        w(f"texelFetchUnchecked({sampler.name}, {sampler.name}_{base.name}_fetch, {x}, {y})")
                        return
                self.show_sym(fun)
            elif fun.kind == "user":
                fd = fun.fd
                if (self.mask is not None or self.return_declared) and fd.globals:
                    cond_mask |= 1 << 31
                param_mask = 0
                for idx, (p, a) in enumerate(zip(fd.params, args)):
                    if symbol_run_class(p.sym, arg_mask) == V:
                        param_mask |= 1 << idx
                    if p.qual in ("inout", "out"):
                        if self.mask is not None or self.return_declared:
                            cond_mask |= 1 << idx
                        if (~arg_mask & param_mask & (1 << idx)) != 0:
                            if adapt_mask == 0:
                                w("{ " if top_level else "({ ")
                            self.show_type(p.ty)
                            w(" _arg%d_ = " % idx)
                            self.show_expr(a)
                            w("; ")
                            adapt_mask |= 1 << idx
                if adapt_mask != 0 and not (fd.ret.kind == "void" and fd.ret.array is None) and not top_level:
                    w("auto _ret_ = ")
                    has_ret = True
                self.show_sym(fun)
                dep_key = (fun, (param_mask | (1 << 31)) if cond_mask != 0 else param_mask)
                if dep_key not in self.deps:
                    self.deps.append(dep_key)
            elif fun.kind == "struct":
                self.show_sym(fun)
                if arg_mask == 0:
                    w("_scalar")
            else:
                raise ValueError("bad identifier to function call")
        w("{{" if array_constructor else "(")
        for idx, a in enumerate(args):
            if idx:
                w(", ")
            if adapt_mask & (1 << idx):
                w("_arg%d_" % idx)
            else:
                self.show_expr(a)
        if cond_mask != 0:
            if args:
                w(", ")
            if self.mask is not None:
                if self.return_declared:
                    w("(")
                    self.show_expr(self.mask)
                    w(")&ret_mask")
                else:
                    self.show_expr(self.mask)
            elif self.return_declared:
                w("ret_mask")
            else:
                w("~0")
        w("}}" if array_constructor else ")")
        if adapt_mask != 0:
            w("; ")
            for idx, a in enumerate(args):
                if adapt_mask & (1 << idx):
                    self.show_lval(a)
                    w(" = force_scalar(_arg%d_); " % idx)
            if has_ret:
                w("_ret_; })")
            else:
                w("}" if top_level else "})")

    @staticmethod
    def texel_fetch_offset(sampler_e, uv_e, off_e):
        if sampler_e.kind == "var" and uv_e.kind == "var" and off_e.kind == "call" and off_e.ctor is None \
                and off_e.fun.name == "ivec2" and len(off_e.args) == 2 \
                and off_e.args[0].kind == "int" and off_e.args[1].kind == "int":
            return (sampler_e.sym, uv_e.sym, off_e.args[0].value, off_e.args[1].value)
        return None

    # ---- declarations (lib.rs:2290-2600) ----------------------------------------------
    def define_texel_fetch_ptr(self, base, sampler, o):
        self.show_indent()
        self.w(f"auto {sampler.name}_{base.name}_fetch = texelFetchPtr({sampler.name}, {base.name}, "
               f"{o[0]}, {o[1]}, {o[2]}, {o[3]});\n")

    def show_sym_decl_name(self, sym):
        if sym.kind == "global":
            if sym.storage == "const":
                self.w("static constexpr ")
            self.w(sym.name)
        elif sym.kind == "local":
            if sym.storage == "const":
                self.w("const ")
            self.w(sym.name)
        else:
            raise ValueError(sym)

    def show_single_declaration(self, d):
        sym = d["sym"]
        w = self.w
        if sym.kind == "global":
            st = sym.storage
            is_uniform = st == "uniform" or (isinstance(st, tuple) and st[0] == "sampler")
            if not self.is_frag:
                if is_uniform or (st == "out" and sym.run_class == S):
                    w("// ")
            else:
                if isinstance(st, tuple) and st[0] == "fragcolor":
                    fragcolor = ("gl_FragColor", "gl_SecondaryFragColor")[st[1]]
                    w(f"#define {sym.name} {fragcolor}\n")
                    self.show_indent()
                    w("// ")
                elif st == "out":
                    w(f"#define {sym.name} gl_FragColor\n")
                    self.show_indent()
                    w("// ")
                elif is_uniform or (st == "in" and sym.run_class == S):
                    w("// ")
        old = self.is_scalar
        self.is_scalar = symbol_run_class(sym, self.vector_mask) == S
        self.show_type(d["ty"])
        w(" ")
        self.show_sym_decl_name(sym)
        self.is_scalar = old
        if d["init"] is not None:
            w(" = ")
            self.show_expr(d["init"])

    def show_declaration(self, d):
        self.show_indent()
        self.show_single_declaration(d)
        for sym, init in d["tail"]:
            self.w(", ")
            self.w(sym.name)
            if init is not None:
                self.w(" = ")
                self.show_expr(init)
        self.w(";\n")
        base = d["sym"]
        if base.kind == "local":
            while True:
                for i, (sampler, b, offsets) in enumerate(self.texel_fetches):
                    if b is base:
                        self.texel_fetches.pop(i)
                        self.define_texel_fetch_ptr(base, sampler, offsets)
                        break
                else:
                    break

    def show_function_prototype(self, fd):
        self.with_scalar(not self.return_vector, lambda: self.show_type(fd.ret))
        w = self.w
        w(" ")
        w(fd.name)
        w("(")
        for i, p in enumerate(fd.params):
            if i:
                w(", ")
            self.with_scalar(symbol_run_class(p.sym, self.vector_mask) == S, lambda: self.show_type(p.ty))
            if p.qual in ("out", "inout"):
                w("&")
            w(" ")
            w(p.name)
        if self.vector_mask & (1 << 31):
            if fd.params:
                w(", ")
            w("I32 _cond_mask_")
        w(")")

    def has_conditional_return(self, body):
        b = self.push_buffer()
        self.show_compound(body)
        self.pop_buffer(b)
        r = self.return_declared
        self.return_declared = False
        return r

    def show_function_definition(self, fd, vector_mask):
        w = self.w
        if fd.name == "main":
            w("ALWAYS_INLINE ")
        self.show_function_prototype(fd)
        w(" ")
        self.return_type = fd.ret
        if vector_mask & (1 << 31):
            self.mask = E("condmask", Type("bool"))
        self.show_indent()
        w("{\n")
        self.indent += 1
        is_void = fd.ret.kind == "void" and fd.ret.array is None
        if self.has_conditional_return(fd.body):
            self.show_indent()
            w("I32" if self.return_vector else "int32_t")
            w(" ret_mask = ")
            if self.mask is not None:
                self.show_expr(self.mask)
            else:
                w("~0")
            w(";\n")
            self.show_indent()
            if not is_void:
                self.with_scalar(not self.return_vector, lambda: self.show_type(fd.ret))
                w(" ret;\n")
        if fd.name in ("swgl_drawSpanRGBA8", "swgl_drawSpanR8"):
            needs_undo = []
            for g in fd.globals:
                if g.kind == "global" and g.storage == "in" and g.run_class == V:
                    if not needs_undo:
                        w("struct _Undo_ {\nSelf* self;\n")
                    self.show_type(g.ty)
                    w(f" {g.name};\n")
                    needs_undo.append(g.name)
            if needs_undo:
                w("explicit _Undo_(Self* self) : self(self)")
                for n in needs_undo:
                    w(f", {n}(self->{n})")
                w(" {}\n")
                w("~_Undo_() {\n")
                for n in needs_undo:
                    w(f"self->{n} = {n};\n")
                w("}} _undo_(this);\n")
        self.texel_fetches = []
        for (sampler, base), offsets in fd.texel_fetches.items():
            self.add_used_global(sampler)
            if base.kind == "global":
                self.add_used_global(base)
                self.define_texel_fetch_ptr(base, sampler, offsets)
            elif base.kind == "local":
                if fd.has_parameter(base):
                    self.define_texel_fetch_ptr(base, sampler, offsets)
                else:
                    self.texel_fetches.append((sampler, base, offsets))
            else:
                raise ValueError(base)
        for st in fd.body[1]:
            self.show_statement(st)
        if self.return_declared:
            self.show_indent()
            w("return;\n" if is_void else "return ret;\n")
        self.indent -= 1
        self.show_indent()
        w("}\n")
        self.return_type = None
        self.return_declared = False
        self.mask = None

    # ---- statements (lib.rs:2640-3330) ------------------------------------------------
    def show_compound(self, c):
        self.show_indent()
        self.w("{\n")
        self.indent += 1
        for st in c[1]:
            self.show_statement(st)
        self.indent -= 1
        self.show_indent()
        self.w("}\n")

    def show_statement(self, s):
        k = s[0]
        w = self.w
        if k == "compound":
            self.show_compound(s)
        elif k == "decl":
            self.show_declaration(s[1])
        elif k == "expr":
            self.show_indent()
            if s[1] is not None:
                self.show_expr(s[1], True)
            w(";\n")
        elif k == "if":
            self.show_selection(s[1], s[2], s[3])
        elif k == "switch":
            self.show_switch(s)
        elif k == "while":
            self.show_indent()
            w("while (")
            self.show_expr(s[1])
            w(") ")
            self.show_statement(s[2])
        elif k == "do":
            self.show_indent()
            w("do ")
            self.show_statement(s[1])
            w(" while (")
            self.show_expr(s[2])
            w(");\n")
        elif k == "for":
            self.show_indent()
            w("for (")
            init = s[1]
            if init[0] == "decl":
                self.show_declaration(init[1])
            elif init[1] is not None:
                self.show_expr(init[1])
            if s[2] is not None:
                self.show_expr(s[2])
            w("; ")
            if s[3] is not None:
                self.show_expr(s[3])
            w(") ")
            self.show_statement(s[4])
        elif k == "return":
            self.show_return(s[1])
        elif k == "break":
            self.show_indent()
            w("break;\n")
        elif k == "continue":
            self.show_indent()
            w("continue;\n")
        elif k == "discard":
            self.show_indent()
            self.uses_discard = True
            if self.mask is not None:
                w("swgl_IsPixelDiscarded |= (")
                self.show_expr(self.mask)
                w(")")
                if self.return_declared:
                    w("&ret_mask")
                w(";\n")
            else:
                w("swgl_IsPixelDiscarded = true;\n")
        else:
            raise ValueError(k)

    def show_selection(self, cond, body, else_stmt):
        w = self.w
        self.show_indent()
        if self.return_declared or self.expr_run_class(cond) != S:
            if self.mask is None or else_stmt is not None:
                self.cond_index += 1
                cond_index = self.cond_index
                w("auto _c%d_ = " % cond_index)
                self.show_expr(cond)
                w(";\n")
                mask = E("cond", Type("bool"), index=cond_index, e=cond)
            else:
                cond_index = 0
                mask = cond
            previous = self.mask
            if previous is not None:
                both = E("binary", Type("bool"), op="&", lhs=previous, rhs=mask)
                self.cond_index += 1
                nested = self.cond_index
                self.show_indent()
                w("auto _c%d_ = " % nested)
                self.show_expr(both)
                w(";\n")
                self.mask = E("cond", Type("bool"), index=nested, e=both)
            else:
                self.mask = mask
            self.show_statement(body)
            self.mask = previous
            if else_stmt is not None:
                inverted = E("unary", Type("bool"), op="~", e=mask)
                previous = self.mask
                if previous is not None:
                    both = E("binary", Type("bool"), op="&", lhs=previous, rhs=inverted)
                    self.show_indent()
                    w("_c%d_ = " % cond_index)
                    self.show_expr(both)
                    w(";\n")
                    self.mask = E("cond", Type("bool"), index=cond_index, e=both)
                else:
                    self.mask = inverted
                self.show_statement(else_stmt)
                self.mask = previous
        else:
            w("if (")
            self.show_expr(cond)
            w(") {\n")
            self.indent += 1
            self.show_statement(body)
            self.indent -= 1
            self.show_indent()
            if else_stmt is not None:
                w("} else ")
                self.show_statement(else_stmt)
            else:
                w("}\n")

    def show_switch(self, s):
        w = self.w
        head, cases = s[1], s[2]
        if self.expr_run_class(head) != S:
            raise NotImplementedError("switch on a per-lane value (lib.rs:3083-3092 lowers it to ifs)")
        self.show_indent()
        w("switch (")
        self.show_expr(head)
        w(") {\n")
        self.indent += 1
        for case in cases:
            self.show_indent()
            if case["label"] is not None:
                w("case ")
                self.show_expr(case["label"])
                w(":\n")
            else:
                w("default:\n")
            self.indent += 1
            has_decl = any(st[0] == "decl" for st in case["stmts"])
            if has_decl:
                self.show_indent()
                w("{\n")
                self.indent += 1
            for st in case["stmts"]:
                self.show_statement(st)
            if has_decl:
                self.show_indent()
                w("}\n")
                self.indent -= 1
            self.indent -= 1
        self.indent -= 1
        self.show_indent()
        w("}\n")

    def use_return_mask(self):
        return self.mask is not None and self.mask.kind != "condmask"

    def show_return(self, e):
        w = self.w
        self.show_indent()
        rmt = "I32" if self.return_vector else "int32_t"
        if e is not None:
            if self.use_return_mask():
                if self.return_declared:
                    w(f"ret = if_then_else(ret_mask & {rmt}(")
                    self.show_expr(self.mask)
                    w("), ")
                    self.show_expr(e)
                    w(", ret);\n")
                else:
                    w("ret = ")
                    self.show_expr(e)
                    w(";\n")
                self.show_indent()
                if self.return_declared:
                    w(f"ret_mask &= ~{rmt}(")
                else:
                    w(f"ret_mask = ~{rmt}(")
                self.show_expr(self.mask)
                w(");\n")
                self.return_declared = True
            else:
                if self.return_declared:
                    w("ret = if_then_else(ret_mask, ")
                    self.show_expr(e)
                    w(", ret);\n")
                else:
                    w("return ")
                    self.show_expr(e)
                    w(";\n")
        else:
            if self.use_return_mask():
                self.show_indent()
                if self.return_declared:
                    w(f"ret_mask &= ~{rmt}(")
                else:
                    w(f"ret_mask = ~{rmt}(")
                self.show_expr(self.mask)
                w(");\n")
                self.return_declared = True
            else:
                w("return;\n")

    # ---- function instantiation (lib.rs:3490-3540) -----------------------------------------
    def show_cxx_function_definition(self, sym, vector_mask):
        if sym.kind != "user":
            return
        fd, run_class = sym.fd, sym.run_class
        self.vector_mask = vector_mask
        if vector_mask & (1 << 31):
            self.return_vector = True
        elif run_class == S:
            self.return_vector = False
        elif is_dep(run_class):
            self.return_vector = (run_class[1] & vector_mask) != 0
        else:
            self.return_vector = True
        key = (sym, vector_mask)
        state = self.functions.get(key)
        if state is True:
            return
        if state is False:
            self.show_function_prototype(fd)
            self.functions[key] = True
            return
        self.functions[key] = False
        b = self.push_buffer()
        self.show_function_definition(fd, vector_mask)
        deps = self.deps
        self.deps = []
        for (dsym, dmask) in deps:
            self.show_cxx_function_definition(dsym, dmask)
        self.flush_buffer()
        self.pop_buffer(b)
        self.functions[key] = True

    def show_translation_unit(self, tu):
        self.flush_buffer()
        for ed in tu:
            if ed[0] == "decl":
                self.show_declaration(ed[1])
            elif ed[0] == "structdef":
                self.show_indent()
                self.show_struct(ed[1])
                self.w(";\n")
            self.flush_buffer()
        for name in ("main", "swgl_drawSpanRGBA8", "swgl_drawSpanR8"):
            sym = self.hir.lookup(name)
            if sym is not None:
                self.show_cxx_function_definition(sym, 0)
                self.flush_buffer()


# ---- per-stage plumbing (lib.rs:120-743, 3560-3680) -----------------------------------------
def is_sampler_storage(st):
    return isinstance(st, tuple) and st[0] == "sampler"


def build_uniform_indices(indices, st):
    for sym in st.used_globals:
        if sym.kind == "global" and (sym.storage == "uniform" or is_sampler_storage(sym.storage)):
            if sym.name not in indices:
                indices[sym.name] = (len(indices) + 1, sym.ty.kind, sym.storage)


def sorted_uniforms(indices):
    return sorted(indices.items())  # BTreeMap iteration order


def write_common_globals(o, attribs, outputs, uniforms):
    w = o.w
    w(f"struct {o.name}_common {{\n")
    # write_program_samplers
    w("struct Samplers {\n")
    for name, (_, tk, storage) in sorted_uniforms(uniforms):
        if tk in ("sampler2D", "sampler2DRect", "isampler2D"):
            suffix = storage[1] if is_sampler_storage(storage) and storage[1] else ""
            w(f" {tk}{suffix}_impl {name}_impl;\n")
            w(f" int {name}_slot;\n")
    w(" bool set_slot(int index, int value) {\n")
    w("  switch (index) {\n")
    for name, (index, tk, _) in sorted_uniforms(uniforms):
        if tk in ("sampler2D", "sampler2DRect", "isampler2D"):
            w(f"  case {index}:\n")
            w(f"   {name}_slot = value;\n")
            w("   return true;\n")
    w("  }\n")
    w("  return false;\n")
    w(" }\n")
    w("} samplers;\n")
    # write_bind_attrib_location
    w("struct AttribLocations {\n")
    for sym in attribs:
        w(f" int {sym.name} = NULL_ATTRIB;\n")
    w(" void bind_loc(const char* name, int index) {\n")
    for sym in attribs:
        w(f"  if (strcmp(\"{sym.name}\", name) == 0) {{ {sym.name} = index; return; }}\n")
    w(" }\n")
    w(" int get_loc(const char* name) const {\n")
    for sym in attribs:
        w(f"  if (strcmp(\"{sym.name}\", name) == 0) {{ return {sym.name} != NULL_ATTRIB ? {sym.name} : -1; }}\n")
    w("  return -1;\n")
    w(" }\n")
    w("} attrib_locations;\n")
    old = o.is_scalar
    o.is_scalar = True
    for sym in outputs:
        if sym.storage == "out" and sym.run_class == S:
            o.show_type(sym.ty)
            w(f" {sym.name};\n")
    for name, (_, tk, storage) in sorted_uniforms(uniforms):
        if is_sampler_storage(storage):
            w(f"{cxx_primitive_type_name(tk)}{storage[1] or ''} {name};\n")
        else:
            o.show_type_kind(Type(tk))
            w(f" {name};\n")
    o.is_scalar = old
    # write_bind_textures
    w("void bind_textures() {\n")
    for name, (_, tk, storage) in sorted_uniforms(uniforms):
        if is_sampler_storage(storage):
            if tk in ("sampler2D", "sampler2DRect"):
                w(f" {name} = lookup_sampler(&samplers.{name}_impl, samplers.{name}_slot);\n")
            elif tk == "isampler2D":
                w(f" {name} = lookup_isampler(&samplers.{name}_impl, samplers.{name}_slot);\n")
    w("}\n")
    w("};\n")


def write_set_uniforms(o, uniforms):
    w = o.w
    w("static void set_uniform_1i(VertexShaderImpl* impl, int index, int value) {\n")
    w(" Self* self = (Self*)impl;\n")
    w(" if (self->samplers.set_slot(index, value)) return;\n")
    w(" switch (index) {\n")
    for name, (index, tk, _) in sorted_uniforms(uniforms):
        w(f" case {index}:\n")
        if tk == "int":
            w(f"  self->{name} = int32_t(value);\n")
        else:
            w(f"  assert(0); // {name}\n")
        w("  break;\n")
    w(" }\n")
    w("}\n")
    w("static void set_uniform_4fv(VertexShaderImpl* impl, int index, const float *value) {\n")
    w(" Self* self = (Self*)impl;\n")
    w(" switch (index) {\n")
    for name, (index, tk, _) in sorted_uniforms(uniforms):
        w(f" case {index}:\n")
        if tk == "vec4":
            w(f"  self->{name} = vec4_scalar::load_from_ptr(value);\n")
        else:
            w(f"  assert(0); // {name}\n")
        w("  break;\n")
    w(" }\n")
    w("}\n")
    w("static void set_uniform_matrix4fv(VertexShaderImpl* impl, int index, const float *value) {\n")
    w(" Self* self = (Self*)impl;\n")
    w(" switch (index) {\n")
    for name, (index, tk, _) in sorted_uniforms(uniforms):
        w(f" case {index}:\n")
        if tk == "mat4":
            w(f"  self->{name} = mat4_scalar::load_from_ptr(value);\n")
        else:
            w(f"  assert(0); // {name}\n")
        w("  break;\n")
    w(" }\n")
    w("}\n")


def write_load_attribs(o, attribs):
    w = o.w
    w("static void load_attribs(VertexShaderImpl* impl, VertexAttrib *attribs, "
      "uint32_t start, int instance, int count) {Self* self = (Self*)impl;\n")
    for sym in attribs:
        func = "load_flat_attrib" if sym.run_class == S else "load_attrib"
        w(f" {func}(self->{sym.name}, attribs[self->attrib_locations.{sym.name}], start, instance, count);\n")
    w("}\n")


def write_store_outputs(o, outputs):
    w = o.w
    old = o.is_scalar
    o.is_scalar = True
    w("public:\nstruct InterpOutputs {\n")
    if o.hir.used_clip_dist != 0:
        w(" Float swgl_ClipDistance;\n")
    for sym in outputs:
        if sym.run_class != S:
            o.show_type(sym.ty)
            w(f" {sym.name};\n")
    w("};\nprivate:\n")
    o.is_scalar = old
    w("ALWAYS_INLINE void store_interp_outputs(char* dest_ptr, size_t stride) {\n")
    w("  for(int n = 0; n < 4; n++) {\n")
    w("    auto* dest = reinterpret_cast<InterpOutputs*>(dest_ptr);\n")
    if o.hir.used_clip_dist != 0:
        for i, comp in enumerate("xyzw"):
            if o.hir.used_clip_dist & (1 << i):
                w(f"    dest->swgl_ClipDistance.{comp} = get_nth(gl_ClipDistance[{i}], n);\n")
            else:
                w(f"    dest->swgl_ClipDistance.{comp} = 0.0f;\n")
    for sym in outputs:
        if sym.run_class != S:
            w(f"    dest->{sym.name} = get_nth({sym.name}, n);\n")
    w("    dest_ptr += stride;\n")
    w("  }\n")
    w("}\n")


def write_read_inputs(o, inputs):
    w = o.w
    w(f"typedef {o.name}_vert::InterpOutputs InterpInputs;\n")
    w("InterpInputs interp_step;\n")
    varyings = [s for s in inputs if s.run_class != S]
    if varyings:
        w("struct InterpPerspective {\n")
        for sym in varyings:
            o.show_type(sym.ty)
            w(f" {sym.name};\n")
        w("};\n")
        w("InterpPerspective interp_perspective;\n")
    head = ("FragmentShaderImpl* impl, const void* init_, const void* step_) {Self* self = (Self*)impl;"
            "const InterpInputs* init = (const InterpInputs*)init_;"
            "const InterpInputs* step = (const InterpInputs*)step_;\n")
    w("static void read_interp_inputs(" + head)
    for sym in varyings:
        n = sym.name
        w(f"  self->{n} = init_interp(init->{n}, step->{n});\n")
        w(f"  self->interp_step.{n} = step->{n} * 4.0f;\n")
    w("}\n")
    used_fragcoord = o.used_fragcoord
    if varyings or (used_fragcoord & (4 | 8)) != 0:
        o.use_perspective = True
    if o.use_perspective:
        w("static void read_perspective_inputs(" + head)
        if varyings:
            w("  Float w = 1.0f / self->gl_FragCoord.w;\n")
        for sym in varyings:
            n = sym.name
            w(f"  self->interp_perspective.{n} = init_interp(init->{n}, step->{n});\n")
            w(f"  self->{n} = self->interp_perspective.{n} * w;\n")
            w(f"  self->interp_step.{n} = step->{n} * 4.0f;\n")
        w("}\n")
    w("ALWAYS_INLINE void step_interp_inputs(int steps = 4) {\n")
    if used_fragcoord & 1:
        w("  step_fragcoord(steps);\n")
    if inputs:
        w("  float chunks = steps * 0.25f;\n")
    for sym in varyings:
        w(f"  {sym.name} += interp_step.{sym.name} * chunks;\n")
    w("}\n")
    if o.use_perspective:
        w("ALWAYS_INLINE void step_perspective_inputs(int steps = 4) {\n")
        if used_fragcoord & 1:
            w("  step_fragcoord(steps);\n")
        w("  step_perspective(steps);\n")
        if inputs:
            w("  float chunks = steps * 0.25f;\n")
        if varyings:
            w("  Float w = 1.0f / gl_FragCoord.w;\n")
        for sym in varyings:
            n = sym.name
            w(f"  interp_perspective.{n} += interp_step.{n} * chunks;\n")
            w(f"  {n} = w * interp_perspective.{n};\n")
        w("}\n")


def write_abi(o):
    w = o.w
    has_rgba8 = o.hir.lookup("swgl_drawSpanRGBA8") is not None
    has_r8 = o.hir.lookup("swgl_drawSpanR8") is not None
    if o.is_frag:
        w("static void run(FragmentShaderImpl* impl) {\n")
        w(" Self* self = (Self*)impl;\n")
        if o.uses_discard:
            w(" self->swgl_IsPixelDiscarded = false;\n")
        w(" self->main();\n")
        w(" self->step_interp_inputs();\n")
        w("}\n")
        w("static void skip(FragmentShaderImpl* impl, int steps) {\n")
        w(" Self* self = (Self*)impl;\n")
        w(" self->step_interp_inputs(steps);\n")
        w("}\n")
        if o.use_perspective:
            w("static void run_perspective(FragmentShaderImpl* impl) {\n")
            w(" Self* self = (Self*)impl;\n")
            if o.uses_discard:
                w(" self->swgl_IsPixelDiscarded = false;\n")
            w(" self->main();\n")
            w(" self->step_perspective_inputs();\n")
            w("}\n")
            w("static void skip_perspective(FragmentShaderImpl* impl, int steps) {\n")
            w(" Self* self = (Self*)impl;\n")
            w(" self->step_perspective_inputs(steps);\n")
            w("}\n")
        if has_rgba8:
            w("static int draw_span_RGBA8(FragmentShaderImpl* impl) {\n")
            w(" Self* self = (Self*)impl; DISPATCH_DRAW_SPAN(self, RGBA8); }\n")
        if has_r8:
            w("static int draw_span_R8(FragmentShaderImpl* impl) {\n")
            w(" Self* self = (Self*)impl; DISPATCH_DRAW_SPAN(self, R8); }\n")
        w(f"public:\n{o.name}_frag() {{\n")
        w(" init_span_func = &read_interp_inputs;\n")
        w(" run_func = &run;\n")
        w(" skip_func = &skip;\n")
        if has_rgba8:
            w(" draw_span_RGBA8_func = &draw_span_RGBA8;\n")
        if has_r8:
            w(" draw_span_R8_func = &draw_span_R8;\n")
        if o.uses_discard:
            w(" enable_discard();\n")
        if o.use_perspective:
            w(" enable_perspective();\n")
            w(" init_span_w_func = &read_perspective_inputs;\n")
            w(" run_w_func = &run_perspective;\n")
            w(" skip_w_func = &skip_perspective;\n")
        else:
            w(" init_span_w_func = &read_interp_inputs;\n")
            w(" run_w_func = &run;\n")
            w(" skip_w_func = &skip;\n")
    else:
        w("static void run(VertexShaderImpl* impl, char* interps, size_t interp_stride) {\n")
        w(" Self* self = (Self*)impl;\n")
        w(" self->main();\n")
        w(" self->store_interp_outputs(interps, interp_stride);\n")
        w("}\n")
        w("static void init_batch(VertexShaderImpl* impl) {\n")
        w(" Self* self = (Self*)impl; self->bind_textures(); }\n")
        w(f"public:\n{o.name}_vert() {{\n")
        w(" set_uniform_1i_func = &set_uniform_1i;\n")
        w(" set_uniform_4fv_func = &set_uniform_4fv;\n")
        w(" set_uniform_matrix4fv_func = &set_uniform_matrix4fv;\n")
        w(" init_batch_func = &init_batch;\n")
        w(" load_attribs_func = &load_attribs;\n")
        w(" run_primitive_func = &run;\n")
        if o.hir.used_clip_dist != 0:
            w(" enable_clip_distance();\n")
    w("}\n")


def parse_stage(src):
    from glsl_parse import parse
    lo = Lower()
    tu = lo.translation_unit(parse(src))
    Infer(lo.st).run(tu)
    return lo.st, tu


def translate_stage(name, st, tu, is_frag, uniform_indices):
    uniforms, inputs, outputs = [], [], []
    for ed in tu:
        if ed[0] != "decl":
            continue
        sym = ed[1]["sym"]
        if sym.kind == "global" and sym in st.used_globals:
            if sym.storage == "uniform" or is_sampler_storage(sym.storage):
                uniforms.append(sym)
            elif sym.storage == "in":
                inputs.append(sym)
            elif sym.storage == "out" or (isinstance(sym.storage, tuple) and sym.storage[0] == "fragcolor"):
                outputs.append(sym)
    o = Out(st, name, is_frag)
    w = o.w
    part_name = name + ("_frag" if is_frag else "_vert")
    if not is_frag:
        write_common_globals(o, inputs, outputs, uniform_indices)
        w(f"struct {name}_vert : VertexShaderImpl, {name}_common {{\nprivate:\n")
    else:
        w(f"struct {name}_frag : FragmentShaderImpl, {name}_vert {{\nprivate:\n")
    w(f"typedef {part_name} Self;\n")
    o.show_translation_unit(tu)
    pruned_inputs = [s for s in inputs if s in o.used_globals]
    if not is_frag:
        write_set_uniforms(o, uniform_indices)
        write_load_attribs(o, pruned_inputs)
        write_store_outputs(o, outputs)
    else:
        write_read_inputs(o, pruned_inputs)
    write_abi(o)
    w("};\n\n")
    if is_frag:
        w(f"struct {name}_program : ProgramImpl, {name}_frag {{\n")
        w("int get_uniform(const char *name) const override {\n")
        for uname, (index, _, _) in sorted_uniforms(uniform_indices):
            w(f" if (strcmp(\"{uname}\", name) == 0) {{ return {index}; }}\n")
        w(" return -1;\n")
        w("}\n")
        w("void bind_attrib(const char* name, int index) override {\n")
        w(" attrib_locations.bind_loc(name, index);\n}\n")
        w("int get_attrib(const char* name) const override {\n")
        w(" return attrib_locations.get_loc(name);\n}\n")
        w("size_t interpolants_size() const override { return sizeof(InterpOutputs); }\n")
        w("VertexShaderImpl* get_vertex_shader() override {\n")
        w(" return this;\n}\n")
        w("FragmentShaderImpl* get_fragment_shader() override {\n")
        w(" return this;\n}\n")
        w(f"const char* get_name() const override {{ return \"{name}\"; }}\n")
        w(f"static ProgramImpl* loader() {{ return new {name}_program; }}\n")
        w("};\n\n")
    # define_global_consts
    for ed in tu:
        if ed[0] == "decl":
            d = ed[1]
            sym = d["sym"]
            if sym.kind == "global" and sym.storage == "const":
                old = o.is_scalar
                o.is_scalar = symbol_run_class(sym, o.vector_mask) == S
                o.show_type(d["ty"])
                w(f" constexpr {part_name}::{sym.name};\n")
                o.is_scalar = old
    o.flush_buffer()
    return "".join(o.output)


def translate(name, vs_src, fs_src):
    """glsl_to_cxx::translate (lib.rs:45-100): vertex then fragment stage of one key."""
    vs_state, vs_tu = parse_stage(vs_src)
    fs_state, fs_tu = parse_stage(fs_src)
    uniform_indices = {}
    build_uniform_indices(uniform_indices, vs_state)
    build_uniform_indices(uniform_indices, fs_state)
    out = translate_stage(name, vs_state, vs_tu, False, uniform_indices)
    out += "\n"
    out += translate_stage(name, fs_state, fs_tu, True, uniform_indices)
    return out
