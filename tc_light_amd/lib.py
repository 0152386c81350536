"""ctypes binding of libtclight_hip.so (the C ABI declared in include/tclight_hip.h).

There is no CPU or PyTorch fallback: if the shared library is missing or a call fails the
caller gets an exception.  Signatures are parsed from the header so it stays the single
source of truth for the boundary.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(ROOT, "include", "tclight_hip.h")
LIB_PATH = os.environ.get("TCL_LIB_PATH") or os.path.join(_HERE, "libtclight_hip.so")     # TCL_LIB_PATH: A/B runs of two builds (tools/ab)

_ERR = {1: "TCL_EINVAL (bad argument / unsupported shape)", 2: "TCL_ELAUNCH (HIP error)"}


def _ctype(decl):
    decl = decl.strip()
    if decl.startswith("const char*") or decl.startswith("const char *"):
        return ctypes.c_char_p
    if "*" in decl or decl.startswith("hipStream_t"):
        return ctypes.c_void_p
    if decl.startswith("size_t"):
        return ctypes.c_size_t
    if decl.startswith("int64_t") or decl.startswith("long"):
        return ctypes.c_int64
    if decl.startswith("float"):
        return ctypes.c_float
    if decl.startswith("int") or decl.startswith("unsigned"):
        return ctypes.c_int
    raise ValueError(f"unsupported C type in header: {decl!r}")


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for ret, name, args in re.findall(r"\b(int|size_t)\s+(tcl_\w+)\s*\(([^)]*)\)\s*;", src):
        out[name] = (ctypes.c_int if ret == "int" else ctypes.c_size_t, [_ctype(a) for a in args.split(",") if a.strip() and a.strip() != "void"])
    return out


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). tc_light_amd has no fallback path.")
        self._dll = ctypes.CDLL(LIB_PATH)
        self._sig = parse_header()
        for name, (res, args) in self._sig.items():
            fn = getattr(self._dll, name)   # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args

    def __getattr__(self, name):
        sig = self.__dict__["_sig"].get(name)
        if sig is None:
            raise AttributeError(name)
        fn = getattr(self._dll, name)
        res = sig[0]

        def call(*a):
            conv = [x.data_ptr() if isinstance(x, torch.Tensor) else (x.encode() if isinstance(x, str) else x) for x in a]
            r = fn(*conv)
            if res is ctypes.c_int and r != 0:
                raise RuntimeError(f"{name} failed: {_ERR.get(r, r)}")
            return r
        call.__name__ = name
        setattr(self, name, call)
        return call


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib


def stream():
    return torch.cuda.current_stream().cuda_stream


def check(t, dtype, *, dev=True):
    if t is None:
        return None
    if t.dtype != dtype or not t.is_contiguous() or (dev and not t.is_cuda):
        raise ValueError(f"expected contiguous {'device ' if dev else ''}{dtype} tensor, got {t.dtype} "
                         f"contiguous={t.is_contiguous()} device={t.device}")
    return t
