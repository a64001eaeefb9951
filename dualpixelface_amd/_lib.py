"""ctypes binding of libdpf_hip.so (the C ABI declared in include/dpf_hip.h).

The prototypes are parsed from the header itself so the binding cannot drift from the ABI.  There is
no fallback: if the library is missing or a call fails, this raises -- the product never computes on
a CPU path (the CPU restatement lives in oracle/ and is test infrastructure only).
"""
import ctypes
import os
import re

# Load order matters: the PyTorch-ROCm wheel bundles its own libamdhip64.so.7; importing torch first makes the dynamic
# loader bind libdpf_hip.so to that same HIP runtime (two runtimes in one process cannot share streams or pointers).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DPF_LIB_PATH') or os.path.join(_HERE, 'libdpf_hip.so')   # override: A/B builds of the kernels
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'dpf_hip.h')

_CTYPES = {
    'int': ctypes.c_int,
    'unsigned': ctypes.c_uint,
    'long long': ctypes.c_longlong,
    'float': ctypes.c_float,
    'double': ctypes.c_double,
}

ERRORS = {-1: 'DPF_ERR_INVALID_ARG', -2: 'DPF_ERR_LAUNCH', -3: 'DPF_ERR_UNSUPPORTED'}


class DpfError(RuntimeError):
    pass


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [(argtype, argname), ...])} for every dpf_* prototype of the header."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    protos = {}
    for m in re.finditer(r'\b(int|long long)\s+(dpf_\w+)\s*\(([^)]*)\)\s*;', src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        parsed = []
        for a in args.split(','):
            a = ' '.join(a.split())
            if not a or a == 'void':
                continue
            if '*' in a:
                parsed.append((ctypes.c_void_p, a.split('*')[-1].strip()))
            else:
                toks = a.split(' ')
                tname = ' '.join(t for t in toks[:-1] if t != 'const')
                parsed.append((_CTYPES[tname], toks[-1]))
        protos[name] = (_CTYPES[ret], parsed)
    return protos


class _Lib(object):
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise DpfError('libdpf_hip.so not found at %s -- build it with `make -C dualpixelface_amd/csrc` '
                           '(or __graft_entry__.build()); there is no CPU fallback' % LIB_PATH)
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        for name, (ret, args) in self.protos.items():
            fn = getattr(self.cdll, name)     # AttributeError if the library lacks a declared symbol
            fn.restype = ret
            fn.argtypes = [t for t, _ in args]

    def call(self, name, *args):
        rc = getattr(self.cdll, name)(*args)
        if self.protos[name][0] is ctypes.c_int and rc != 0:
            raise DpfError('%s failed: %s' % (name, ERRORS.get(rc, rc)))
        return rc


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
