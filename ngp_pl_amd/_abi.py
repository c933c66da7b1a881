"""A small reader of include/ngp_hip.h: the prototypes a C compiler sees (comments stripped first), as data.

Used by the tests to hold the hand-written ctypes table of `_lib.py` to the header (arity and every argument's class), and to
generate the C translation unit that takes the address of every declared entry point with the header's own prototype and links it
against libngp_hip.so.  Round 3's header had one declaration swallowed by a comment: a regular expression over the raw text still
"found" it, a compiler did not -- hence comments go first, exactly as translation phase 3 does it.
"""
import ctypes as C
import os
import re

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "ngp_hip.h")        # the drop-in boundary
# library-internal entry points (cross-translation-unit launchers, white-box test hooks): exported, declared apart, same conventions
INTERNAL_HEADER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "ngp_internal.h")

_SCALARS = {
    "int": C.c_int, "int32_t": C.c_int32, "uint32_t": C.c_uint32, "int64_t": C.c_int64, "uint64_t": C.c_uint64,
    "float": C.c_float, "double": C.c_double, "size_t": C.c_size_t, "long long": C.c_longlong, "unsigned": C.c_uint,
}
_POINTER_TYPEDEFS = {"ngp_stream_t"}          # typedef void* ngp_stream_t


def strip_comments(text):
    """/* ... */ and // ... replaced by one space each (C11 5.1.1.2 phase 3); string literals do not occur in this header."""
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _strip_blocks(text):
    """Drops the bodies of struct definitions (their members end in ';' too) and preprocessor lines."""
    text = re.sub(r"^\s*#[^\n]*(\\\n[^\n]*)*", " ", text, flags=re.M)
    # the `extern "C" {` ... `}` wrapper (each half sits under its own #ifdef __cplusplus) is not a block to drop
    if 'extern "C" {' in text:
        text = text.replace('extern "C" {', " ", 1)
        last = text.rindex("}")
        text = text[:last] + " " + text[last + 1:]
    out, depth = [], 0
    for ch in text:
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
            if depth < 0:
                raise ValueError("unbalanced braces in the header")
        elif depth == 0:
            out.append(ch)
    return "".join(out)


class Param:
    def __init__(self, ctype, name):
        self.ctype, self.name = ctype, name
        self.is_pointer = "*" in ctype or ctype.replace("const", "").strip() in _POINTER_TYPEDEFS

    @property
    def scalar(self):
        """ctypes class of a by-value argument (None for pointers)."""
        if self.is_pointer:
            return None
        return _SCALARS[self.ctype.replace("const", "").strip()]

    def __repr__(self):
        return "%s %s" % (self.ctype, self.name)


class Proto:
    def __init__(self, ret, name, params):
        self.ret, self.name, self.params = ret, name, params

    def c_pointer_decl(self, var):
        """`ret (*var)(params)`: the header's prototype as a function-pointer declarator."""
        args = ", ".join(p.ctype for p in self.params) or "void"
        return "%s (*%s)(%s)" % (self.ret, var, args)


def parse(path=HEADER):
    """name -> Proto for every function declared in the header, as a compiler would see them."""
    text = open(path).read()
    text = re.sub(r'^\s*#include[^\n]*', " ", text, flags=re.M)
    body = _strip_blocks(strip_comments(text))
    # `extern "C"` wrapper braces were removed together with their (unbalanced across #ifdef) partners: what is left between
    # semicolons is typedefs and prototypes
    protos = {}
    for stmt in body.split(";"):
        stmt = " ".join(stmt.split())
        m = re.match(r"^(?:extern \"C\" )?((?:const )?[\w ]+?\**) ?\b(ngp_\w+) ?\((.*)\)$", stmt)
        if not m or stmt.startswith("typedef"):
            continue
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                pm = re.match(r"^(.*?)(\w+)$", a)
                ctype, pname = pm.group(1).strip(), pm.group(2)
                if not ctype:                       # unnamed parameter
                    ctype, pname = pname, ""
                params.append(Param(ctype, pname))
        if name in protos:
            raise ValueError("%s declared twice in %s" % (name, path))
        protos[name] = Proto(ret, name, params)
    return protos


def parse_all():
    """Public and internal prototypes together (what libngp_hip.so exports); a name may only be declared in one of the two."""
    pub, internal = parse(HEADER), parse(INTERNAL_HEADER)
    both = set(pub) & set(internal)
    if both:
        raise ValueError("declared in both headers: %s" % sorted(both))
    return {**pub, **internal}


def ctypes_agrees(argtypes, proto):
    """None if a ctypes argtypes list matches the prototype, else a description of the first disagreement."""
    if len(argtypes) != len(proto.params):
        return "%s: %d ctypes arguments, the header declares %d" % (proto.name, len(argtypes), len(proto.params))
    for i, (t, p) in enumerate(zip(argtypes, proto.params)):
        t_is_ptr = t in (C.c_void_p, C.c_char_p) or hasattr(t, "contents") or issubclass(t, C._Pointer)
        if p.is_pointer != t_is_ptr:
            return "%s argument %d (%r): header %s, ctypes %s" % (proto.name, i, p, "pointer" if p.is_pointer else "scalar", t.__name__)
        if not p.is_pointer:
            want = p.scalar
            if C.sizeof(want) != C.sizeof(t) or (want in (C.c_float, C.c_double)) != (t in (C.c_float, C.c_double)):
                return "%s argument %d (%r): ctypes %s" % (proto.name, i, p, t.__name__)
    return None
