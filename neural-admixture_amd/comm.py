"""Transports for the collectives of nadm_step (include/nadm.h, nadm_comm_t).

The step itself -- launches AND collectives -- is queued by ONE C call; what differs between deployments is only who moves the
bytes:

* ``rccl_comm``      a communicator of the library's own over RCCL / xGMI (ncclCommInitRank; the 128-byte id travels through
                     torch.distributed's store).  The production transport: the reference gets its communicator from
                     ``init_process_group("nccl")`` + DistributedDataParallel (src/utils.py:88-93, neural_admixture.py:315-319).
* ``torch_comm``     the same three operations on top of a torch.distributed group of ANY backend, as host callbacks -- for
                     process groups without RCCL (gloo: the world-2 tests that share one GPU).  Blocking, not for measurement.
* ``emulated_comm``  rank 0 of ``world`` ranks whose collectives do nothing: the per-rank cost of a world-rank step on ONE GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch

from ._lib import lib, check, CommStruct, COMM_SLICES_FN, COMM_DESTROY_FN


class Comm:
    """Owner of a ``nadm_comm_t``; ``handle`` is what nadm_plan_desc_t.comm takes.  Must outlive every plan built on it."""

    def __init__(self, handle, rank: int, world: int, kind: str, owned: bool = True, keep=None):
        self.handle, self.rank, self.world, self.kind = handle, int(rank), int(world), kind
        self._owned, self._keep = owned, keep

    def count_ranks(self, device: torch.device) -> int:
        """Sum of ones over the communicator: the number of ranks that really take part (a connectivity check)."""
        t = torch.ones(4, dtype=torch.float32, device=device)
        tr = getattr(self, "transport", None)
        if tr is not None:
            tr.buffers.append(t)
        c = self.handle.contents
        st = torch.cuda.current_stream().cuda_stream if device.type == "cuda" else 0
        check(c.all_reduce(c.ctx, t.data_ptr(), 4, st), "comm all_reduce")
        if device.type == "cuda":
            torch.cuda.synchronize()
        if tr is not None:
            tr.buffers.pop()
        return int(round(float(t[0].item())))

    def close(self) -> None:
        if self._owned and self.handle is not None:
            lib.nadm_comm_free(self.handle)
        self.handle = None

    def __del__(self):
        # An RCCL communicator is only destroyed by an explicit close(): ncclCommDestroy is a collective-ish call that must not run
        # from a finalizer (interpreter shutdown, another rank already gone); leaving it to process exit is harmless
        if self.kind != "rccl":
            try:
                self.close()
            except Exception:
                pass


def loaded_librccl() -> Optional[bytes]:
    """Path of the librccl.so already mapped into this process (torch's own copy), so that the library's communicator and
    torch's share one RCCL; None: let the loader search."""
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "librccl" in line:
                    return line.split(None, 5)[-1].strip().encode()
    except OSError:
        pass
    return None


def rccl_comm(rank: int, world: int, group=None, device: Optional[torch.device] = None) -> Comm:
    """One RCCL communicator over the ranks of ``group`` (default group; world = 1 needs no torch.distributed), for the GPU
    ``device`` (default: the current one) -- ncclCommInitRank binds the communicator to the HIP device that is current."""
    if device is not None and device.type == "cuda":
        with torch.cuda.device(device):
            return rccl_comm(rank, world, group)
    path = loaded_librccl()
    uid = (C.c_char * 128)()
    if rank == 0:
        check(lib.nadm_comm_rccl_unique_id(path, uid), "comm_rccl_unique_id")
    if world > 1:
        import torch.distributed as dist
        box = [bytes(uid.raw) if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        uid = (C.c_char * 128).from_buffer_copy(box[0])
    out = C.POINTER(CommStruct)()
    check(lib.nadm_comm_rccl(path, uid, rank, world, C.byref(out)), "comm_rccl")
    return Comm(out, rank, world, "rccl")


def emulated_comm(world: int) -> Comm:
    out = C.POINTER(CommStruct)()
    check(lib.nadm_comm_emulated(world, C.byref(out)), "comm_emulated")
    return Comm(out, 0, world, "emulated")


class _TorchTransport:
    """The three collectives on registered torch tensors, called back from inside nadm_step with raw pointers."""

    def __init__(self, rank: int, world: int, group):
        self.rank, self.world, self.group = rank, world, group
        self.buffers: List[torch.Tensor] = []
        self.error: Optional[BaseException] = None

    def view(self, ptr: int, n: int) -> torch.Tensor:
        for t in self.buffers:
            off = ptr - t.data_ptr()
            if 0 <= off and off + 4 * n <= t.numel() * t.element_size():
                return t.view(-1)[off // 4: off // 4 + n]
        raise RuntimeError("collective on a buffer that was not registered with the transport")

    def _on(self, stream: Optional[int], fn) -> int:
        try:
            dev = self.buffers[0].device
            if dev.type == "cuda":
                with torch.cuda.stream(torch.cuda.ExternalStream(stream or 0, device=dev)):
                    fn()
            else:
                fn()
            return 0
        except BaseException as e:                       # never let an exception cross the C frame
            self.error = e
            return 1

    def reduce_scatter(self, _ctx, buf, sl, stream) -> int:
        import torch.distributed as dist
        # (a sum over the whole buffer leaves this rank's slice with the sum, which is all the contract asks for)
        return self._on(stream, lambda: dist.all_reduce(self.view(buf, sl * self.world), group=self.group))

    def all_gather(self, _ctx, buf, sl, stream) -> int:
        import torch.distributed as dist

        def go():
            full = self.view(buf, sl * self.world)
            mine = full[self.rank * sl:(self.rank + 1) * sl].clone()
            dist.all_gather_into_tensor(full, mine, group=self.group)
        return self._on(stream, go)

    def all_reduce(self, _ctx, buf, n, stream) -> int:
        import torch.distributed as dist
        return self._on(stream, lambda: dist.all_reduce(self.view(buf, n), group=self.group))


def torch_comm(rank: int, world: int, group=None) -> Comm:
    """Collectives through torch.distributed (any backend).  Register every buffer the step communicates with
    ``comm.transport.buffers.append(tensor)`` (Engine does)."""
    tr = _TorchTransport(rank, world, group)
    cbs = (COMM_SLICES_FN(tr.reduce_scatter), COMM_SLICES_FN(tr.all_gather), COMM_SLICES_FN(tr.all_reduce))
    st = CommStruct(rank, world, None, cbs[0], cbs[1], cbs[2], COMM_DESTROY_FN())
    c = Comm(C.pointer(st), rank, world, "torch", owned=False, keep=(st, cbs, tr))
    c.transport = tr
    return c


def make_comm(device: torch.device, rank: int, world: int, group=None) -> Comm:
    """The transport for this process: RCCL when the ranks sit on different GPUs behind an nccl process group (or there is one
    rank), torch.distributed callbacks otherwise.  Should the library's own communicator fail to come up on ANY rank (librccl not
    loadable, ncclCommInitRank refused), every rank falls back to the callbacks over the process group together -- slower (the
    reduce-scatter becomes an all-reduce, every collective a host call) but running; ``Comm.kind`` says which one is in use."""
    import torch.distributed as dist
    if world == 1:
        return rccl_comm(0, 1, device=device) if device.type == "cuda" else torch_comm(0, 1, group)
    if device.type == "cuda" and dist.get_backend(group) == "nccl":
        comm, err = None, None
        try:
            comm = rccl_comm(rank, world, group, device=device)
        except Exception as e:                                  # noqa: BLE001 -- whatever it was, the ranks must agree on what happens next
            err = e
        ok = torch.tensor([0.0 if comm is None else 1.0], device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if float(ok.item()) == 1.0:
            return comm
        if comm is not None:
            comm.close()
        import sys
        print(f"[neural_admixture_amd] the step's own RCCL communicator is not available on every rank ({err!r}): "
              "falling back to torch.distributed callbacks", file=sys.stderr)
    return torch_comm(rank, world, group)
