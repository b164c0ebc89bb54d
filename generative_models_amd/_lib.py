"""ctypes binding of libgmk.so, generated from include/gmk.h at import time.

The prototypes are parsed from the header the library was compiled against, so the Python side can never
drift from the C ABI.  There is no fallback: if the shared library is missing or a symbol is absent the
import fails loudly (the product path must not silently run on anything else).
"""
import ctypes
import os
import re

import torch  # noqa: F401  (FIRST: libgmk.so must bind to the HIP runtime torch has loaded - two libamdhip64 copies in one process
#                      do not share a device: a library loaded before torch reports "no ROCm-capable device" at its first launch)

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
HEADER = os.path.join(_ROOT, "include", "gmk.h")
# GMK_LIBGMK: another build of the SAME library (a diagnostic variant build); still the HIP path
LIBPATH = os.environ.get("GMK_LIBGMK") or os.path.join(_PKG, "libgmk.so")

_CTYPES = {
    "int": ctypes.c_int,
    "int64_t": ctypes.c_int64,
    "uint64_t": ctypes.c_uint64,
    "float": ctypes.c_float,
}


def _ctype(decl):
    decl = decl.strip()
    if "*" in decl:
        return ctypes.c_char_p if decl.replace(" ", "") == "constchar*" else ctypes.c_void_p
    base = decl.replace("const", "").split()[0]
    return _CTYPES[base]


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes], [argnames])} for every prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"^\s*((?:const\s+)?\w+\s*\*?)\s*(gmk_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.M | re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argtypes, argnames = [], []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)(\w+)$", a)
                argtypes.append(_ctype(mm.group(1)))
                argnames.append(mm.group(2))
        protos[name] = (_ctype(ret), argtypes, argnames)
    return protos


class GmkError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIBPATH):
        raise ImportError(
            f"{LIBPATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            f"or make -C generative_models_amd/csrc). There is no CPU fallback.")
    lib = ctypes.CDLL(LIBPATH)
    protos = parse_header()
    for name, (ret, argtypes, _) in protos.items():
        fn = getattr(lib, name)     # AttributeError if the library does not export a declared symbol
        fn.restype = ret
        fn.argtypes = argtypes
    return lib, protos


lib, PROTOS = _load()


def check(rc, what=""):
    if rc != 0:
        msg = lib.gmk_last_error()
        raise GmkError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
