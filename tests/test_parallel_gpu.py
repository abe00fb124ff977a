"""k22_comm_broadcast_weights on a real RCCL communicator (one rank: the one GPU of the test box).  The world-size-2 behaviour
of the sharded path is covered on CPU by tests/test_parallel_cpu.py (gloo); the driver's multi-GPU bench exercises RCCL over xGMI."""
import ctypes as C

import pytest
import torch

from kandinsky2_amd import _lib

pytestmark = pytest.mark.gpu


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


@pytest.mark.timeout(180)
def test_comm_broadcast_weights_on_a_one_rank_rccl_communicator():
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")                      # context + the RCCL copy PyTorch links
    try:
        rccl = C.CDLL("librccl.so")
    except OSError:
        rccl = C.CDLL("librccl.so.1")
    uid = _UniqueId()
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        arena = torch.arange(0, 3 << 20, dtype=torch.int32, device="cuda").view(torch.uint8)
        want = arena.clone()
        L = _lib.lib()
        _lib.check(L.k22_comm_broadcast_weights(arena.data_ptr(), arena.numel(), 0, comm, _lib.current_stream()))
        torch.cuda.synchronize()
        assert torch.equal(arena, want)
        assert L.k22_comm_broadcast_weights(None, 16, 0, comm, None) != 0      # null arena is rejected, not dereferenced
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
