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
import os
from typing import List, Optional

import torch

from ._lib import lib, check, CommStruct, COMM_SLICES_FN, COMM_DESTROY_FN, COMM_CHECK_FN


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
    torch's share one RCCL; None: let the loader search.  Only ``librccl.so`` / ``librccl.so.<version>`` count -- a plugin such
    as librccl-net.so has none of the symbols."""
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                fields = line.split(None, 5)
                if len(fields) < 6:
                    continue
                path = fields[5].strip()
                base = os.path.basename(path)
                if base == "librccl.so" or base.startswith("librccl.so."):
                    return path.encode()
    except OSError:
        pass
    return None


class CommUnavailable(RuntimeError):
    """The library's own communicator cannot come up, and every rank knows it BEFORE any of them entered ncclCommInitRank: falling
    back to another transport is safe (make_comm does)."""


class CommInitFailed(RuntimeError):
    """ncclCommInitRank failed or timed out on some rank.  Every rank raises this (the same list of ranks in the message); a rank
    whose call never returned has a helper thread stuck inside the library -- tear the process down, do not fall back."""


def init_timeout_s() -> float:
    """Deadline of ncclCommInitRank's watchdog: NADM_COMM_TIMEOUT_S, default 120 s (first contact over xGMI takes seconds)."""
    try:
        return float(os.environ.get("NADM_COMM_TIMEOUT_S", "120"))
    except ValueError:
        return 120.0


def rccl_comm(rank: int, world: int, group=None, device: Optional[torch.device] = None, librccl: Optional[bytes] = None,
              timeout_s: Optional[float] = None) -> Comm:
    """One RCCL communicator over the ranks of ``group`` (default group; world = 1 needs no torch.distributed), for the GPU
    ``device`` (default: the current one) -- ncclCommInitRank binds the communicator to the HIP device that is current.

    ncclCommInitRank is itself a collective, so first contact is a little protocol in which no rank ever waits for a peer that has
    already given up (the reference tears everything down and re-raises on the master when a rank fails, src/main.py:119-133):
      1. every rank loads the library and resolves its symbols (``nadm_comm_rccl_probe``); rank 0 draws the unique id
      2. the id is broadcast -- by every rank, whatever happened in 1 (a rank 0 that failed sends None)
      3. the ranks exchange what they have; unless ALL are ready every rank raises ``CommUnavailable`` -- nobody has entered the
         collective call yet
      4. ncclCommInitRank under a watchdog (``timeout_s``; csrc/nadm_step.hip): a peer that dies now costs the others the timeout,
         not forever
      5. the ranks exchange the outcome; unless ALL succeeded every rank aborts what it got and raises ``CommInitFailed``
    ``librccl``: another library with the same seven entry points (tests: a stub whose ncclCommInitRank fails or hangs)."""
    if device is not None and device.type == "cuda":
        with torch.cuda.device(device):
            return rccl_comm(rank, world, group, None, librccl, timeout_s)
    path = librccl if librccl is not None else loaded_librccl()
    timeout_ms = int(1000 * (init_timeout_s() if timeout_s is None else timeout_s))
    uid = (C.c_char * 128)()
    err = None
    if lib.nadm_comm_rccl_probe(path):
        err = (lib.nadm_last_error() or b"").decode()
    elif rank == 0 and lib.nadm_comm_rccl_unique_id(path, uid):
        err = (lib.nadm_last_error() or b"").decode()
    if world > 1:
        import torch.distributed as dist
        box = [bytes(uid.raw) if (rank == 0 and err is None) else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        states = [None] * world
        dist.all_gather_object(states, err, group=group)
        bad = {r: e for r, e in enumerate(states) if e is not None}
        if bad or box[0] is None:
            raise CommUnavailable(f"RCCL transport not available on every rank (rank: reason) {bad}")
        uid = (C.c_char * 128).from_buffer_copy(box[0])
    elif err is not None:
        raise CommUnavailable(err)
    out = C.POINTER(CommStruct)()
    rc = lib.nadm_comm_rccl(path, uid, rank, world, timeout_ms, C.byref(out))
    msg = (lib.nadm_last_error() or b"").decode() if rc else None
    if world > 1:
        import torch.distributed as dist
        states = [None] * world
        dist.all_gather_object(states, msg, group=group)
        bad = {r: e for r, e in enumerate(states) if e is not None}
        if bad:
            if rc == 0:
                lib.nadm_comm_abort(out)                 # (ncclCommAbort: its peers are gone or never arrived)
            raise CommInitFailed(f"ncclCommInitRank did not complete on every rank (rank: reason) {bad}")
    elif rc:
        raise CommInitFailed(msg)
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

    @staticmethod
    def torch_stream(stream: Optional[int], dev: torch.device) -> "torch.cuda.Stream":
        """The torch stream object for a raw HIP stream handle.  Handle 0 is the device's default stream and must be named as such:
        ``ExternalStream(0)`` does NOT wrap the null stream -- torch reads a zero pointer as "no pointer given" and hands out a fresh
        pool stream, on which a collective is ordered against nothing the step launched (r05: the reduce-scatter of message B read
        gradients pass 3 had not finished and the optimizer ran before the sums were back, at 4+ ranks sharing one GPU)."""
        return torch.cuda.ExternalStream(stream, device=dev) if stream else torch.cuda.default_stream(dev)

    def _on(self, stream: Optional[int], fn) -> int:
        try:
            dev = self.buffers[0].device
            if dev.type == "cuda":
                with torch.cuda.stream(self.torch_stream(stream, dev)):
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
    st = CommStruct(rank, world, None, cbs[0], cbs[1], cbs[2], COMM_DESTROY_FN(), COMM_CHECK_FN())
    c = Comm(C.pointer(st), rank, world, "torch", owned=False, keep=(st, cbs, tr))
    c.transport = tr
    return c


def make_comm(device: torch.device, rank: int, world: int, group=None) -> Comm:
    """The transport for this process: RCCL when the ranks sit on different GPUs behind an nccl process group (or there is one
    rank), torch.distributed callbacks otherwise.  Should the library's own communicator be unavailable on ANY rank (librccl not
    loadable, a symbol missing, no unique id), every rank learns so before anyone enters ncclCommInitRank (``rccl_comm``) and all fall
    back to the callbacks over the process group together -- slower (the reduce-scatter becomes an all-reduce, every collective a host
    call) but running; ``Comm.kind`` says which one is in use.  A failure INSIDE ncclCommInitRank is not papered over: every rank
    raises ``CommInitFailed``.  Call it again for a second communicator (message A, ``Engine(comm_a=...)``)."""
    import torch.distributed as dist
    if world == 1:
        return rccl_comm(0, 1, device=device) if device.type == "cuda" else torch_comm(0, 1, group)
    if device.type == "cuda" and dist.get_backend(group) == "nccl":
        try:
            return rccl_comm(rank, world, group, device=device)
        except CommUnavailable as e:
            import sys
            print(f"[neural_admixture_amd] the step's own RCCL communicator is not available ({e}): "
                  "falling back to torch.distributed callbacks", file=sys.stderr)
    return torch_comm(rank, world, group)
