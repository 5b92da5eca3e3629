"""Torch-free collectives of the path: thin wrapper over libm324's m324_comm_* entry points (RCCL over xGMI).

The product path of this package issues its collectives through torch.distributed (backend "nccl" = RCCL), because the
reference's callers own that process group (setup.py:134-140).  This module is for hosts that bind libm324 without
torch.distributed: the same gradient all-reduce / k|v all-gather, taking raw device pointers and a HIP stream.

    uid = Communicator.unique_id()            # rank 0; ship the hex string to the other ranks
    comm = Communicator(uid, rank, world)     # every rank, after selecting its device
    comm.all_reduce(tensor, average=True); comm.all_gather(send, recv); comm.close()
"""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as L
from .ops import code_of


class Communicator:
    def __init__(self, unique_id_hex: str, rank: int, world: int):
        self._h = C.c_void_p()
        L.check(L.load().m324_comm_init(C.byref(self._h), unique_id_hex.encode(), rank, world), "m324_comm_init")
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id() -> str:
        buf = C.create_string_buffer(257)
        L.check(L.load().m324_comm_unique_id(buf, 257), "m324_comm_unique_id")
        return buf.value.decode()

    def all_reduce(self, t: torch.Tensor, average: bool = False) -> torch.Tensor:
        if not (t.is_cuda and t.is_contiguous()):
            raise L.M324Error("all_reduce: contiguous HIP tensor required")
        L.check(L.load().m324_comm_allreduce(self._h, t.data_ptr(), t.numel(), code_of(t.dtype), int(average),
                                             torch.cuda.current_stream().cuda_stream), "m324_comm_allreduce")
        return t

    def all_gather(self, send: torch.Tensor, recv: torch.Tensor) -> torch.Tensor:
        if not (send.is_cuda and send.is_contiguous() and recv.is_cuda and recv.is_contiguous()) or \
                recv.numel() != self.world * send.numel() or recv.dtype != send.dtype:
            raise L.M324Error("all_gather: recv must be a contiguous HIP tensor of world x send elements")
        L.check(L.load().m324_comm_allgather(self._h, send.data_ptr(), recv.data_ptr(), send.numel(), code_of(send.dtype),
                                             torch.cuda.current_stream().cuda_stream), "m324_comm_allgather")
        return recv

    def close(self) -> None:
        if self._h:
            L.check(L.load().m324_comm_destroy(self._h), "m324_comm_destroy")
            self._h = C.c_void_p()
