"""Tiny C-header -> ctypes.Structure translator.

The POD model (csrc/cm_model.h: ``cm_model_t``) is shared verbatim between the
host model compiler, the HIP kernels and the test oracle.  Instead of keeping a
hand-written Python mirror in sync, the ctypes layout is derived from the header
itself.  Supports exactly what those headers use: ``#define NAME int``,
``enum { A = 1, ... }`` (ignored), and ``typedef struct tag { ... } name;`` with
scalar / 1-D / 2-D array members of int, unsigned, double, float, uint64_t or a
previously parsed struct.
"""
import ctypes
import re

_BASE = {
    "bool": ctypes.c_bool,
    "short": ctypes.c_short,
    "ushort": ctypes.c_ushort,
    "uchar": ctypes.c_ubyte,
    "uint": ctypes.c_uint,
    "DiagnosticCodes": ctypes.c_short,
    "int": ctypes.c_int,
    "unsigned": ctypes.c_uint,
    "double": ctypes.c_double,
    "float": ctypes.c_float,
    "uint64_t": ctypes.c_uint64,
}


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def parse_defines(text, known=None):
    macros = dict(known or {})
    for m in re.finditer(r"^\s*#define\s+(\w+)\s+(\d+)\s*$", _strip_comments(text), flags=re.M):
        macros[m.group(1)] = int(m.group(2))
    return macros


def parse_structs(text, macros, known_types=None):
    """Returns {typedef_name: ctypes.Structure subclass} for every typedef struct in text."""
    types = dict(_BASE)
    types.update(known_types or {})
    out = {}
    body_re = re.compile(r"typedef\s+struct\s+\w*\s*\{(.*?)\}\s*(\w+)\s*;", flags=re.S)
    for m in body_re.finditer(_strip_comments(text)):
        body, name = m.group(1), m.group(2)
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            decl = decl.replace("unsigned short", "ushort").replace("unsigned char", "uchar").replace("unsigned int", "uint")
            tname, rest = decl.split(" ", 1)
            if tname not in types:
                raise ValueError("unknown type %r in struct %s" % (tname, name))
            base = types[tname]
            for d in rest.split(","):
                d = d.strip()
                mm = re.match(r"^(\w+)((?:\[\w+\])*)$", d)
                if not mm:
                    raise ValueError("cannot parse declarator %r in struct %s" % (d, name))
                dims = [macros[x] if x in macros else int(x) for x in re.findall(r"\[(\w+)\]", mm.group(2))]
                t = base
                for n in reversed(dims):
                    t = t * n
                fields.append((mm.group(1), t))
        cls = type(name, (ctypes.Structure,), {"_fields_": fields})
        types[name] = cls
        out[name] = cls
    return out
