"""TEST INFRASTRUCTURE: run the product's HIP sources on the CPU through tools/hipemu (host build of the same .hip files
against a stand-in <hip/hip_runtime.h>: every work-item a fiber, MFMA / LDS-DMA / wave shuffles emulated with the hardware's
register layouts).  `enable()` points `pixray_amd._lib` at libprx_emu.so and lifts the "tensors must be on a ROCm device"
guards, so that the SAME Python wrappers (ops.py autograd Functions, runner handles, C-ABI calls) drive the emulated kernels on
CPU tensors.  Nothing here is used by the product, and nothing here says anything about speed."""
import contextlib
import os
import subprocess

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tools", "hipemu")
EMU_LIB = os.path.join(EMU_DIR, "libprx_emu.so")


def build() -> str:
    if os.environ.get("HIPEMU_LIB"):           # a checking build made by hand (tools/hipemu/Makefile, SAN=...)
        return os.environ["HIPEMU_LIB"]
    subprocess.run(["make", "-C", EMU_DIR, "-j", str(os.cpu_count() or 4)], check=True, stdout=subprocess.DEVNULL)
    return EMU_LIB


@contextlib.contextmanager
def enable():
    from pixray_amd import _lib, ops
    emu_lib = build()
    saved = dict(avail_fn=_lib.device_available, path=_lib.LIB_PATH, lib=_lib._lib, protos=_lib._protos, ctx=_lib._tool_ctx, stream=_lib.current_stream,
                 need=ops._need_cuda, warr=ops._weight_array, sync=torch.cuda.synchronize, avail=torch.cuda.is_available)
    _lib.LIB_PATH, _lib._lib, _lib._protos, _lib._tool_ctx = emu_lib, None, None, None
    _lib.current_stream = lambda: 0
    _lib.device_available = lambda: True
    ops._need_cuda = lambda *ts: None

    def weight_array(tensors):
        import ctypes
        arr = (ctypes.c_void_p * len(tensors))()
        for i, t in enumerate(tensors):
            assert t.dtype == torch.float32 and t.is_contiguous()
            arr[i] = t.data_ptr()
        return arr
    ops._weight_array = weight_array
    torch.cuda.synchronize = lambda *a, **k: None
    try:
        yield _lib.load()
    finally:
        _lib.LIB_PATH, _lib._lib, _lib._protos, _lib._tool_ctx = saved["path"], saved["lib"], saved["protos"], saved["ctx"]
        _lib.current_stream, ops._need_cuda, ops._weight_array = saved["stream"], saved["need"], saved["warr"]
        _lib.device_available = saved["avail_fn"]
        torch.cuda.synchronize = saved["sync"]
