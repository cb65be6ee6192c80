"""ctypes binding of libprx_hip.so (the C ABI declared in include/prx.h).

The header is the single source of truth: function prototypes are parsed from
it and turned into ctypes signatures, so the Python side can never drift from
the ABI.  There is NO fallback: if the shared library is missing or a call
fails, a `PrxError` is raised.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
HEADER_PATH = os.path.join(_ROOT, "include", "prx.h")
LIB_PATH = os.path.join(_HERE, "csrc", "libprx_hip.so")


class PrxError(RuntimeError):
    pass


class GemmArgs(ctypes.Structure):
    """mirror of `prx_gemm_args` (include/prx.h)"""
    _fields_ = [
        ("A", ctypes.c_void_p), ("a_is_f32", ctypes.c_int), ("a_mode", ctypes.c_int), ("lda", ctypes.c_int),
        ("B", ctypes.c_void_p), ("ldb", ctypes.c_int),
        ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int),
        ("H", ctypes.c_int), ("W", ctypes.c_int), ("Cin", ctypes.c_int), ("up", ctypes.c_int),
        ("alpha", ctypes.c_float),
        ("bias_n", ctypes.c_void_p), ("bias_m", ctypes.c_void_p),
        ("aux", ctypes.c_void_p), ("ldaux", ctypes.c_int),
        ("resid", ctypes.c_void_p), ("ldr", ctypes.c_int),
        ("act", ctypes.c_int),
        ("out_f32", ctypes.c_void_p), ("ldc_f32", ctypes.c_int),
        ("out_bf16", ctypes.c_void_p), ("out_bf16_pre", ctypes.c_void_p), ("ldc_bf16", ctypes.c_int),
        ("f32", ctypes.c_int), ("row16", ctypes.c_int), ("ctx", ctypes.c_void_p),
    ]

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        if not self.ctx:               # tests / tools time and tune their bare GEMM calls through one context of their own
            try:
                self.ctx = tool_ctx()
            except PrxError:
                pass


PREC_BF16, PREC_F32, PREC_F16 = 0, 1, 2
# the product default: IEEE-half operands -- the reference's own GPU arithmetic for the CLIP towers (slip.py:175), at the
# bf16 MFMA rate with 8x finer operand rounding (include/prx.h PRX_PREC_F16)
DEFAULT_PRECISION = "fp16"


def precision_code(precision) -> int:
    """'fp16' | 'bf16' | 'f32' (or the PRX_PREC_* integers) -> PRX_PREC_*; None -> the product default"""
    if precision is None:
        precision = DEFAULT_PRECISION
    if isinstance(precision, str):
        precision = precision.lower()
    if precision in (PREC_F16, "fp16", "f16", "half", "float16"):
        return PREC_F16
    if precision in (PREC_BF16, "bf16", "bfloat16"):
        return PREC_BF16
    if precision in (PREC_F32, "f32", "fp32", "float32"):
        return PREC_F32
    raise ValueError(f"unknown precision {precision!r} (want 'fp16', 'bf16' or 'f32')")


def split_precision(precision):
    """(drawer precision, perceptor precision) of a session-level precision.  "ref" is the REFERENCE's own arithmetic mix on a
    GPU: the VQGAN decoder in fp32 (taming's VQModel stays fp32, vqgan.py:124-140) and the CLIP towers in IEEE half
    (clip.load keeps fp16 weights / activations, slip.py:175) -- here: exact-f32 MFMA decoder + fp16-operand towers."""
    if isinstance(precision, str) and precision.lower() in ("ref", "reference", "mixed"):
        return "f32", "fp16"
    return precision, precision


def precision_name(precision) -> str:
    return {PREC_F16: "fp16", PREC_BF16: "bf16", PREC_F32: "f32"}[precision_code(precision)]


_SCALARS = {
    "int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double,
    "long long": ctypes.c_longlong, "size_t": ctypes.c_size_t, "uint64_t": ctypes.c_uint64,
    "int64_t": ctypes.c_int64, "uint32_t": ctypes.c_uint32, "prx_stream_t": ctypes.c_void_p,
}


def parse_header(path: str = HEADER_PATH) -> Dict[str, Tuple[object, List[object]]]:
    """Return {name: (restype, [argtypes])} for every `prx_*` prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"(?:^|[;}\n])\s*((?:const\s+)?[\w ]+?[\s\*]+)(prx_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "*" in ret:
            restype = ctypes.c_char_p if "char" in ret else ctypes.c_void_p
        elif ret == "void":
            restype = None
        else:
            restype = _SCALARS[ret]
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                    continue
                toks = a.replace("const", " ").split()
                ty = " ".join(toks[:-1]) if len(toks) > 1 else toks[0]
                argtypes.append(_SCALARS[ty])
        protos[name] = (restype, argtypes)
    return protos


_lib = None
_protos = None
_tool_ctx = None


def tool_ctx():
    """A `prx_gemm_ctx` owned by this Python process for bare `prx_k_gemm` calls (tests, tools/gemm_*.py): tile overrides
    and per-launch timing live in a context, never in the library (include/prx.h).  Runner handles have their own."""
    global _tool_ctx
    if _tool_ctx is None:
        _tool_ctx = load().prx_gemm_ctx_create()
    return _tool_ctx


def load() -> ctypes.CDLL:
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PrxError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or `make -C pixray_amd/csrc`). There is no CPU fallback for the hot path.")
    # PyTorch bundles its own HIP/HSA runtime (SONAME libamdhip64.so.7, same as /opt/rocm's).  Import torch FIRST so
    # that the dynamic loader binds this library to the runtime torch already loaded: one process, one runtime,
    # and the hipStream_t handed over from torch belongs to the runtime that launches our kernels.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    _protos = parse_header()
    for name, (restype, argtypes) in _protos.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise PrxError(f"libprx_hip.so does not export {name} declared in include/prx.h") from e
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error() -> str:
    lib = load()
    msg = lib.prx_last_error()
    return msg.decode() if msg else ""


def _ptr(x):
    """torch tensor / int / None -> raw device address"""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, ctypes.c_void_p):
        return x.value
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    if isinstance(x, (ctypes.Structure,)):
        return ctypes.addressof(x)
    raise TypeError(f"cannot pass {type(x)} as a pointer")


def call(name: str, *args):
    """Call `name` from the C ABI; tensors are passed as device pointers. Raises PrxError on rc != 0."""
    lib = load()
    restype, argtypes = _protos[name]
    if len(args) != len(argtypes):
        raise TypeError(f"{name} expects {len(argtypes)} arguments, got {len(args)}")
    conv = []
    for a, t in zip(args, argtypes):
        conv.append(_ptr(a) if t is ctypes.c_void_p else a)
    rc = getattr(lib, name)(*conv)
    if restype is ctypes.c_int and rc != 0:
        raise PrxError(f"{name} failed (rc={rc}): {last_error()}")
    return rc


def device_available() -> bool:
    """is a ROCm device visible?  (One function, so that the CPU emulation of the kernels -- tests/_emu.py, test
    infrastructure -- can stand in for a device; the product has no CPU path.)"""
    import torch
    return torch.cuda.is_available()


def require_device() -> None:
    load()      # fail loudly if the HIP extension is missing
    if not device_available():
        raise PrxError("no ROCm device visible: the hot path has no CPU fallback")


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
