"""Ghost-row exchange straight on RCCL (ncclSend / ncclRecv in one group, on the caller's stream).

`torch.distributed`'s "nccl" backend IS RCCL, but `batch_isend_irecv` costs ~65 us of host time
per exchange and runs its kernel on a stream of the process group's own choosing -- measured here
on the hardware queue of the compute stream, where the exchange then waits behind the interior
rows it was meant to overlap.  This module calls the same librccl.so (the copy bundled with
PyTorch, already loaded by it) through ctypes: the exchange kernel goes on the stream the band
engine asks for (`Core.comm_stream()`, measured to run beside the compute stream) and posting it
costs a few ctypes calls.  `torch.distributed` is still what bootstraps the communicator (it
carries the 128-byte unique id from rank 0 to the others) and what `bench.py` uses for barriers.

`RcclP2P` offers the four names `BandRunner` uses of torch.distributed (`P2POp`, `isend`, `irecv`,
`batch_isend_irecv`), so it drops in as the runner's `dist`.
"""
import ctypes as C
import os


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]       # ncclUniqueId, rccl.h


def uid_to_bytes(uid):
    """all 128 bytes (the id has NUL bytes inside: a c_char array read as an attribute stops at one)"""
    return C.string_at(C.addressof(uid), C.sizeof(uid))


def uid_from_bytes(raw):
    if len(raw) != C.sizeof(_UniqueId):
        raise ValueError("ncclUniqueId is %d bytes, got %d" % (C.sizeof(_UniqueId), len(raw)))
    uid = _UniqueId()
    C.memmove(C.addressof(uid), raw, len(raw))
    return uid


_NCCL_CHAR = 0                                       # ncclInt8 / ncclChar: the buffers go as bytes


def _load():
    import torch
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    lib = C.CDLL(path)
    lib.ncclGetErrorString.restype = C.c_char_p
    lib.ncclGetErrorString.argtypes = [C.c_int]
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    lib.ncclSend.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclRecv.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return lib


class RcclError(RuntimeError):
    pass


class RcclP2P:
    """A communicator of `world` ranks for point-to-point ghost-row exchanges.

    bootstrap: an initialised torch.distributed module (any backend) used once, to broadcast the
    unique id; None for a single rank (a band that is its own ring neighbour: tests, tools)."""

    isend, irecv = "isend", "irecv"

    class P2POp:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    class _Done:
        def wait(self):                 # stream-ordered: whoever waits for the stream has the data
            pass

    def __init__(self, bootstrap, rank, world, uid_bytes=None):
        """uid_bytes: the 128-byte ncclUniqueId if the caller has distributed it already (bench.py does,
        so that a rank that fails before the broadcast cannot leave the others waiting in it)"""
        import torch
        self.torch, self.rank, self.world = torch, rank, world
        self.lib = _load()
        if uid_bytes is not None:
            uid = uid_from_bytes(uid_bytes)
        else:
            uid = _UniqueId()
            if rank == 0:
                self._check(self.lib.ncclGetUniqueId(C.byref(uid)))
            if world > 1:
                box = [uid_to_bytes(uid) if rank == 0 else None]
                bootstrap.broadcast_object_list(box, src=0)
                uid = uid_from_bytes(box[0])
        self.comm = C.c_void_p()
        self._check(self.lib.ncclCommInitRank(C.byref(self.comm), world, uid, rank))

    @staticmethod
    def new_unique_id():
        """-> 128 bytes (rank 0 calls this; the caller carries them to the other ranks)"""
        lib = _load()
        uid = _UniqueId()
        rc = lib.ncclGetUniqueId(C.byref(uid))
        if rc != 0:
            raise RcclError("rccl: %s" % lib.ncclGetErrorString(rc).decode())
        return uid_to_bytes(uid)

    def _check(self, rc):
        if rc != 0:
            raise RcclError("rccl: %s" % self.lib.ncclGetErrorString(rc).decode())

    def batch_isend_irecv(self, ops):
        """all of `ops` as one RCCL group on the CURRENT torch stream; order within the list is the
        matching order per peer, as with torch.distributed"""
        stream = C.c_void_p(self.torch.cuda.current_stream().cuda_stream)
        lib, comm = self.lib, self.comm
        self._check(lib.ncclGroupStart())
        try:
            for o in ops:
                t = o.tensor
                fn = lib.ncclSend if o.op == "isend" else lib.ncclRecv
                self._check(fn(C.c_void_p(t.data_ptr()), t.numel() * t.element_size(), _NCCL_CHAR, o.peer, comm, stream))
        finally:
            rc = lib.ncclGroupEnd()        # a failed send must not leave the group open
        self._check(rc)
        return [self._Done()]

    def entry_points(self):
        """addresses of ncclSend, ncclRecv, ncclGroupStart, ncclGroupEnd in the loaded librccl.so, for
        gcm_set_exchange (the C library posts the exchange itself: Core.band_run)"""
        lib = self.lib
        return tuple(C.cast(f, C.c_void_p).value for f in (lib.ncclSend, lib.ncclRecv, lib.ncclGroupStart, lib.ncclGroupEnd))

    def self_check(self):
        """one ring shift of a small tensor (to the south neighbour, from the north one) on the
        current stream; raises if what arrives is not the north neighbour's rank"""
        torch = self.torch
        north, south = (self.rank - 1) % self.world, (self.rank + 1) % self.world
        out = torch.full((64,), float(self.rank), dtype=torch.float64, device="cuda")
        got = torch.full((64,), -1.0, dtype=torch.float64, device="cuda")
        self.batch_isend_irecv([self.P2POp("isend", out, south), self.P2POp("irecv", got, north)])
        torch.cuda.current_stream().synchronize()
        if not bool((got == float(north)).all()):
            raise RcclError("rccl self-check: rank %d received %r from rank %d" % (self.rank, got[:2].tolist(), north))

    def count(self):
        """ncclCommCount: the number of ranks this communicator really has (bench.py prints it in the N > 1 line)"""
        n = C.c_int(-1)
        self.lib.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        self._check(self.lib.ncclCommCount(self.comm, C.byref(n)))
        return n.value

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()
