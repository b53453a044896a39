"""ctypes binding of libaa_hip.so (C ABI declared in include/aa_hip.h and include/aa_hip_f32.h).

The prototypes are parsed from the header itself so the header stays the single source of truth.
There is NO CPU fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('AA_HIP_LIB') or os.path.join(HERE, 'libaa_hip.so')     # AA_HIP_LIB: another build of the SAME ABI (same-box A/B runs)
HEADER = os.path.join(os.path.dirname(HERE), 'include', 'aa_hip.h')
HEADER_F32 = os.path.join(os.path.dirname(HERE), 'include', 'aa_hip_f32.h')   # fp32 parity-mode twins

_SCALARS = {
    'int': ctypes.c_int,
    'long': ctypes.c_long,
    'float': ctypes.c_float,
    'int64_t': ctypes.c_int64,
}


def parse_header(path: str = HEADER) -> dict[str, tuple[str, list[tuple[str, str]]]]:
    """Return {name: (return_type, [(ctype_string, arg_name), ...])} for every prototype."""
    text = open(path).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    protos = {}
    for m in re.finditer(r'^\s*(const char\*|int)\s+(aa_\w+)\s*\(([^;]*?)\)\s*;', text, flags=re.M | re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        arglist = []
        args = ' '.join(args.split())
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                am = re.match(r'^(.*?)(\w+)$', a)
                arglist.append((am.group(1).strip(), am.group(2)))
        protos[name] = (ret, arglist)
    return protos


def _ctype(t: str):
    if '*' in t:
        return ctypes.c_void_p
    t = t.replace('const', '').strip()
    return _SCALARS[t]


class AAHipError(RuntimeError):
    pass


class _Lib:
    def __init__(self) -> None:
        self._dll = None
        self.protos = {**parse_header(HEADER), **parse_header(HEADER_F32)}

    def load(self):
        if self._dll is not None:
            return self._dll
        if not os.path.exists(LIB_PATH):
            raise AAHipError(
                f'{LIB_PATH} not found: build it with `python -m align_anything_amd.build` '
                '(hipcc --offload-arch=gfx950). There is no CPU fallback for the hot path.'
            )
        # In a torch process the HIP runtime torch ships must be the one this library binds to (same
        # SONAME): load torch first so both share one runtime / device context.  The library itself has no
        # torch dependency.
        try:
            import torch  # noqa: F401
        except ImportError:  # pure-ctypes host: the system ROCm runtime is used
            pass
        dll = ctypes.CDLL(LIB_PATH)
        for name, (ret, args) in self.protos.items():
            fn = getattr(dll, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = ctypes.c_char_p if ret.startswith('const char') else ctypes.c_int
            fn.argtypes = [_ctype(t) for t, _ in args]
        self._dll = dll
        return dll

    def call(self, name: str, *args) -> None:
        dll = self.load()
        rc = getattr(dll, name)(*args)
        if rc != 0:
            msg = dll.aa_last_error()
            raise AAHipError(f'{name} failed (rc={rc}): {msg.decode() if msg else "?"}')


LIB = _Lib()


def call(name: str, *args) -> None:
    LIB.call(name, *args)


def exported_symbols() -> list[str]:
    return sorted(LIB.protos)
