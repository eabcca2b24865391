"""ctypes loader for libdeeprl_amd.so (the C-ABI of include/deeprl_amd.h).

The prototypes are parsed from the public header itself, so the header is the single source of
truth for the boundary.  There is NO fallback: if the shared library is missing or a call fails,
an exception is raised -- the product path never routes around the HIP kernels.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "deeprl_amd.h")
# DEEPRL_AMD_LIB: measurement builds of the SAME sources (tools/phase_trace.py loads libdeeprl_amd_trace.so); still a HIP
# library with the full C ABI -- there is no non-HIP implementation to point this at.
LIBRARY = os.environ.get("DEEPRL_AMD_LIB") or os.path.join(_HERE, "lib", "libdeeprl_amd.so")

_SCALARS = {
    "int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64,
    "float": ctypes.c_float, "double": ctypes.c_double,
}


class DraError(RuntimeError):
    pass


def parse_header(path=HEADER):
    """Returns {name: [ctype, ...]} for every `int dra_*(...)` prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(dra_\w+)\s*\(([^)]*)\)\s*;", src):
        name, args = m.group(1), m.group(2).strip()
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    types.append(ctypes.c_void_p)
                    continue
                toks = [t for t in a.replace("const", " ").split() if t]
                base = toks[0]
                if base not in _SCALARS:
                    raise DraError("unknown C type in header prototype %s: %r" % (name, a))
                types.append(_SCALARS[base])
        protos[name] = types
    return protos


class _Lib:
    def __init__(self):
        self._dll = None
        self.protos = parse_header()

    def _load(self):
        if self._dll is None:
            if not os.path.isfile(LIBRARY):
                raise DraError(
                    "%s not found: build it with `make -C deeprl_amd/csrc` (or __graft_entry__.build()). "
                    "deeprl_amd has no CPU / eager fallback." % LIBRARY)
            dll = ctypes.CDLL(LIBRARY)
            for name, argtypes in self.protos.items():
                try:
                    fn = getattr(dll, name)
                except AttributeError:
                    raise DraError("%s does not export %s declared in %s" % (LIBRARY, name, HEADER))
                fn.argtypes = argtypes
                fn.restype = ctypes.c_int
            self._dll = dll
        return self._dll

    def __getattr__(self, name):
        if not name.startswith("dra_"):
            raise AttributeError(name)
        fn = getattr(self._load(), name)

        def call(*args):
            rc = fn(*args)
            if rc != 0:
                raise DraError("%s failed with code %d%s" % (name, rc, _describe(rc)))
            return rc

        call.raw = fn
        setattr(self, name, call)
        return call


def _describe(rc):
    if rc == -22:
        return " (invalid argument)"
    if rc == -12:
        return " (out of host memory)"
    if rc == -110:
        return " (a bounded device-side wait gave up -- late_step's arrival slots or the actor's in-launch hand-over: results invalid)"
    if rc == 100:
        return " (hipErrorNoDevice: deeprl_amd needs an MI355X; there is no CPU path)"
    if rc == 2:
        return " (hipErrorOutOfMemory)"
    return " (hipError_t)" if rc > 0 else ""


lib = _Lib()


def ptr(t):
    """Device (or pinned-host) pointer of a torch tensor, or None."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def ptr_array(tensors):
    """C array of pointers (`const T* const*`) built from a list of tensors (None allowed)."""
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


_torch = None


def stream_ptr(stream=None):
    """hipStream_t of `stream` (default: torch's current stream on the current device).  The raw-stream
    query is ~20x cheaper than torch.cuda.current_stream(), which builds a Stream object per call -- the
    generic agent path asks ~50 times per agent step."""
    global _torch
    if _torch is None:
        import torch
        _torch = torch
    if stream is not None:
        return ctypes.c_void_p(stream.cuda_stream)
    return ctypes.c_void_p(_torch._C._cuda_getCurrentRawStream(_torch.cuda.current_device()))
