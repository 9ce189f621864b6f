"""ctypes binding of librscotr.so (the C ABI declared in include/rscotr.h).

The product path has NO fallback: if the shared object is missing or a call fails, a
RuntimeError is raised. Signatures are parsed from include/rscotr.h so the header stays the
single source of truth for the boundary.
"""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RSCOTR_LIB") or os.path.join(HERE, "librscotr.so")  # RSCOTR_LIB: A/B builds (scripts/)
HEADER = os.path.join(HERE, "..", "include", "rscotr.h")
_TRACE = os.environ.get('RSCOTR_TRACE_CALLS') == '1'

_CTYPES = {
    "int": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float,
    "double": ctypes.c_double, "void": None, "size_t": ctypes.c_size_t,
    "uint64_t": ctypes.c_uint64, "uint32_t": ctypes.c_uint32, "int32_t": ctypes.c_int32,
}


def header_abi_version(path=HEADER):
    with open(path) as fh:
        m = re.search(r"^#define\s+RSCOTR_ABI_VERSION\s+(\d+)", fh.read(), flags=re.M)
    if m is None:
        raise RuntimeError("include/rscotr.h does not define RSCOTR_ABI_VERSION")
    return int(m.group(1))


def parse_header(path=HEADER):
    """Return {name: (restype, [argtypes])} for every function declared in the header."""
    with open(path) as fh:
        src = fh.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(rscotr_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        out[name] = (_ctype(ret), [] if args in ("", "void") else [_ctype(_strip_name(a)) for a in args.split(",")])
    return out


def _strip_name(arg):
    arg = arg.strip()
    if "*" in arg:
        return arg[: arg.rindex("*") + 1]
    return " ".join(arg.split()[:-1])


def _ctype(t):
    t = t.replace("const", "").strip()
    if t.endswith("*"):
        base = t[:-1].strip()
        if base == "char":
            return ctypes.c_char_p
        if base == "unsigned char":
            return ctypes.c_void_p
        return ctypes.c_void_p
    return _CTYPES[t]


class _Lib:
    def __init__(self):
        self._dll = None
        self._sigs = None

    def load(self):
        if self._dll is not None:
            return self._dll
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"librscotr.so not found at {LIB_PATH}: run `python -m rscotr_amd.build` "
                "(there is no CPU or PyTorch fallback for the HIP path)")
        dll = ctypes.CDLL(LIB_PATH)
        self._sigs = parse_header()
        dll.rscotr_version.restype = ctypes.c_int
        built, want = dll.rscotr_version(), header_abi_version()
        if built != want:  # (a stale build, or an A/B library picked by RSCOTR_LIB that predates an argument-list change)
            raise RuntimeError(f"{LIB_PATH} was built for ABI revision {built}, include/rscotr.h declares {want}: "
                               "rebuild it (`python -m rscotr_amd.build --force`)")
        for name, (ret, args) in self._sigs.items():
            fn = getattr(dll, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = ret
            fn.argtypes = args
        self._dll = dll
        return dll

    def call(self, name, *args):
        dll = self.load()
        if _TRACE:  # RSCOTR_TRACE_CALLS=1 (with AMD_SERIALIZE_KERNEL=3): the last line on stderr names a faulting launch
            import sys
            print(f'[rscotr] {name}{args}', file=sys.stderr, flush=True)
        rc = getattr(dll, name)(*args)
        if rc != 0:
            msg = dll.rscotr_last_error()
            raise RuntimeError(f"{name} failed ({rc}): {msg.decode() if msg else ''}")

    def __getattr__(self, name):
        if name.startswith("rscotr_"):
            return getattr(self.load(), name)
        raise AttributeError(name)


lib = _Lib()
