"""Host side of the C-ABI exchange step (`prx_comm`, include/prx.h): the one-shot direct-write all-reduce of the sharded
iteration.  The IPC window handles are exchanged ONCE, at construction, through whatever process group
the caller already has (gloo or RCCL); the per-step collective itself never touches torch.distributed.

    comm = OneShotComm(group, rank, world, max_bytes=4 << 20)
    comm.all_reduce_sum_(g)          # in place, on the current stream; g: contiguous fp32 CUDA tensor
    comm.all_reduce_(mm, "max")      # the {-min, max} pair; comm.all_reduce_(acc64, "sum") the four fp64 renormalisation sums

The Session uses it for all three collectives of the sharded iteration when it is handed one (`Session(..., comm=...)`, or
`PRX_ONESHOT_ALLREDUCE=1` with api.build_*): after start-up such a run does not touch torch.distributed again.  The default
remains torch.distributed's all_reduce (RCCL).
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
        self._pad = {}

    SUM_F32, MAX_F32, SUM_F64 = 0, 1, 2          # include/prx.h PRX_COMM_*

    def all_reduce_(self, t: torch.Tensor, op: str = "sum") -> torch.Tensor:
        """in place over the ranks, on the current stream: fp32 sum / max, or fp64 sum (the three collectives of the sharded
        iteration: dL/d(image), the {-min, max} pair of the batch-global renormalisation, its four backward sums)"""
        if not (t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.float64)):
            raise PrxError("OneShotComm.all_reduce_: contiguous fp32 / fp64 CUDA tensor expected")
        if t.dtype == torch.float64:
            if op != "sum":
                raise PrxError("OneShotComm.all_reduce_: fp64 tensors are summed only")
            code, words = self.SUM_F64, 2 * t.numel()
        else:
            if op not in ("sum", "max"):
                raise PrxError(f"OneShotComm.all_reduce_: unknown op {op!r}")
            code, words = (self.SUM_F32 if op == "sum" else self.MAX_F32), t.numel()
        if words % 4 == 0 and t.data_ptr() % 16 == 0:
            call("prx_allreduce", self.handle, t, words, code, current_stream())
            return t
        # scalar-sized vectors: through a 16-byte-aligned staging buffer padded with the operation's neutral element
        m = (words + 3) // 4 * 4
        per = 2 if t.dtype == torch.float64 else 1
        key = t.dtype
        pad = self._pad.get(key) if isinstance(self._pad, dict) else None
        if pad is None or pad.numel() * per < m:
            if not isinstance(self._pad, dict):
                self._pad = {}
            pad = self._pad[key] = torch.zeros(max(m // per, 16), device=t.device, dtype=t.dtype)
        n = t.numel()
        pad[:m // per].fill_(float("-inf") if op == "max" else 0.0)
        pad[:n].copy_(t.reshape(-1))
        call("prx_allreduce", self.handle, pad, m, code, current_stream())
        t.copy_(pad[:n].reshape(t.shape))
        return t

    def all_reduce_sum_(self, t: torch.Tensor) -> torch.Tensor:
        return self.all_reduce_(t, "sum")

    def check(self):
        """raise when a wait of an earlier call timed out (its result was poisoned with NaN); synchronises the device"""
        st = self.status()
        if st:
            raise PrxError(f"one-shot all-reduce: rank {self.rank} gave up waiting for rank {st - 1}'s data (prx_comm_status = {st})")

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
