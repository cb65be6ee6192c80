"""Host side of the C-ABI exchange step (`prx_comm`, include/prx.h): the one-shot direct-write all-reduce of the sharded
iteration.  The IPC window handles are exchanged ONCE, at construction, through whatever process group
the caller already has (gloo or RCCL); the per-step collective itself never touches torch.distributed.

    comm = OneShotComm(group, rank, world, max_bytes=4 << 20)
    comm.all_reduce_sum_(g)          # in place, on the current stream; g: contiguous fp32 CUDA tensor

The Session uses it for the image-gradient all-reduce when it is handed one (`Session(..., comm=...)`, or
`PRX_ONESHOT_ALLREDUCE=1` with api.build_*); the default remains torch.distributed's all_reduce (RCCL).
"""
from __future__ import annotations

import ctypes

import torch

from ._lib import PrxError, call, current_stream, load


class OneShotComm:
    def __init__(self, group, rank: int, world: int, max_bytes: int = 4 << 20):
        import torch.distributed as dist
        load()
        self.rank, self.world, self.max_bytes = int(rank), int(world), int(max_bytes)
        h = ctypes.c_void_p()
        call("prx_comm_create", ctypes.addressof(h), self.rank, self.world, self.max_bytes)
        self.handle = h
        nb = load().prx_comm_handle_bytes()
        blob = ctypes.create_string_buffer(nb)
        call("prx_comm_export", self.handle, ctypes.addressof(blob))
        if self.world > 1:
            blobs = [None] * self.world
            dist.all_gather_object(blobs, bytes(blob.raw), group=group)
            allb = ctypes.create_string_buffer(b"".join(blobs), nb * self.world)
            call("prx_comm_connect", self.handle, ctypes.addressof(allb))
            dist.barrier(group=group)          # nobody writes into a window that has not been mapped everywhere yet
        self._pad = None

    def all_reduce_sum_(self, t: torch.Tensor) -> torch.Tensor:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise PrxError("OneShotComm.all_reduce_sum_: contiguous fp32 CUDA tensor expected")
        n = t.numel()
        if n % 4 == 0 and t.data_ptr() % 16 == 0:
            call("prx_allreduce_grad", self.handle, t, n, current_stream())
            return t
        # scalar-sized vectors (the min / max renormalisation sums): through a 16-byte-aligned, zero-padded staging buffer
        m = (n + 3) // 4 * 4
        if self._pad is None or self._pad.numel() < m:
            self._pad = torch.zeros(max(m, 16), device=t.device, dtype=torch.float32)
        self._pad[:m].zero_()
        self._pad[:n].copy_(t.reshape(-1))
        call("prx_allreduce_grad", self.handle, self._pad, m, current_stream())
        t.copy_(self._pad[:n].reshape(t.shape))
        return t

    def status(self) -> int:
        return int(load().prx_comm_status(self.handle))

    def close(self):
        if self.handle:
            load().prx_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
