"""ctypes binding of libv3d_comm.so (include/v3d_comm.h): the frame-axis exchanges of a frame-sharded evaluation over RCCL behind a C ABI.

The Python host itself uses torch.distributed (v3d_amd/dist.py::FrameShard: the same message schedule as P2P ops on torch tensors); this
binding exists for the tests and as the reference of what a non-Python host binds.  Verified on hardware with ONE rank only (one GPU per box)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libv3d_comm.so")
ABI_VERSION = 1
ID_BYTES = 128
c_vp, c_i32, c_i64 = C.c_void_p, C.c_int32, C.c_int64

SIGNATURES = {
    "v3d_comm_abi_version": (c_i32, []),
    "v3d_comm_last_error": (C.c_char_p, []),
    "v3d_comm_unique_id": (c_i32, [c_vp]),
    "v3d_comm_init": (c_i32, [c_vp, c_i32, c_i32, C.POINTER(c_vp)]),
    "v3d_comm_destroy": (c_i32, [c_vp]),
    "v3d_comm_frame_range": (c_i32, [c_i32, c_i32, c_i32, C.POINTER(c_i32), C.POINTER(c_i32)]),
    "v3d_comm_allgather_frames": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i64, c_vp]),
    "v3d_comm_exchange_halo_and_sums": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "v3d_comm_selftest": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "v3d_comm_debug_schedule": (c_i32, [c_i32, c_i32, c_i32, c_i32, c_i64, c_i32, c_i64, c_i64, C.POINTER(c_i64), c_i32]),
}


def load_library(path: Optional[str] = None):
    lib = C.CDLL(path or LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    v = lib.v3d_comm_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError(f"libv3d_comm.so ABI version {v} != expected {ABI_VERSION}; rebuild")
    return lib


def debug_schedule(lib, kind: int, rank: int, world: int, *, have_buf: bool = True, B: int = 1, T_global: int = 18, frame_bytes: int = 0, nsums: int = 0):
    """The message list the library would issue (no GPU needed): [(send, peer, buffer, offset, bytes), ...] - see include/v3d_comm.h."""
    cap = 4096
    out = (c_i64 * (5 * cap))()
    n = lib.v3d_comm_debug_schedule(kind, rank, world, int(have_buf), B, T_global, frame_bytes, nsums, out, cap)
    if n < 0:
        raise ValueError(lib.v3d_comm_last_error().decode())
    return [tuple(int(out[5 * i + j]) for j in range(5)) for i in range(n)]


def frame_range(lib, T_global: int, world: int, rank: int):
    t0, tl = c_i32(), c_i32()
    if lib.v3d_comm_frame_range(T_global, world, rank, C.byref(t0), C.byref(tl)) != 0:
        raise ValueError(lib.v3d_comm_last_error().decode())
    return t0.value, tl.value


class Comm:
    """One rank's communicator.  unique_id: the 128 bytes rank 0 got from `Comm.unique_id(lib)`, handed over by the host's own channel."""

    def __init__(self, unique_id: bytes, rank: int, world: int, lib=None):
        self.lib = lib or load_library()
        self.rank, self.world = rank, world
        h = c_vp()
        buf = C.create_string_buffer(unique_id, ID_BYTES)
        self._check(self.lib.v3d_comm_init(C.cast(buf, c_vp), rank, world, C.byref(h)), "v3d_comm_init")
        self.handle = h

    @staticmethod
    def unique_id(lib) -> bytes:
        buf = C.create_string_buffer(ID_BYTES)
        if lib.v3d_comm_unique_id(C.cast(buf, c_vp)) != 0:
            raise RuntimeError(lib.v3d_comm_last_error().decode())
        return buf.raw

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed (rc={rc}): {self.lib.v3d_comm_last_error().decode()}")

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def allgather_frames(self, local: torch.Tensor, T_global: int) -> torch.Tensor:
        """local [B, T_local, ...] contiguous -> [B, T_global, ...]"""
        assert local.is_contiguous()
        B = local.shape[0]
        frame_bytes = local[0, 0].numel() * local.element_size()
        out = torch.empty((B, T_global) + tuple(local.shape[2:]), dtype=local.dtype, device=local.device)
        self._check(self.lib.v3d_comm_allgather_frames(self.handle, local.data_ptr(), out.data_ptr(), B, T_global, frame_bytes, self._stream()),
                    "v3d_comm_allgather_frames")
        return out

    def exchange_halo_and_sums(self, buf: Optional[torch.Tensor], B: int, T_global: int, frame_bytes: int, sums: Optional[torch.Tensor]):
        total = allsums = None
        if sums is not None:
            assert sums.dtype == torch.float64 and sums.is_contiguous()
            allsums = torch.empty((self.world,) + tuple(sums.shape), dtype=torch.float64, device=sums.device)
            total = torch.empty_like(sums)
        self._check(self.lib.v3d_comm_exchange_halo_and_sums(self.handle, None if buf is None else buf.data_ptr(), B, T_global, frame_bytes,
                                                             None if sums is None else sums.data_ptr(), None if sums is None else allsums.data_ptr(),
                                                             None if sums is None else total.data_ptr(), 0 if sums is None else sums.numel(), self._stream()),
                    "v3d_comm_exchange_halo_and_sums")
        return total

    def selftest(self, src: torch.Tensor) -> torch.Tensor:
        dst = torch.empty_like(src)
        self._check(self.lib.v3d_comm_selftest(self.handle, src.data_ptr(), dst.data_ptr(), src.numel() * src.element_size(), self._stream()), "v3d_comm_selftest")
        return dst

    def destroy(self):
        if self.handle:
            self._check(self.lib.v3d_comm_destroy(self.handle), "v3d_comm_destroy")
            self.handle = None
