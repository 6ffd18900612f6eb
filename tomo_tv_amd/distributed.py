"""Slab sharding of the tilt axis across ranks (one process per GPU, ``torch.distributed``; backend "nccl" is RCCL).

The projector loop (FP, BP, SIRT, SART, clamp, momentum) is independent per x-slice because every slice shares one
system matrix, so ranks own contiguous slabs and exchange nothing there.  Only two things cross ranks:

* scalar partial sums (residual, step norms, TV, ||grad TV||^2): one all-reduce of float64 elements each
  (reference: ``MPI_Allreduce`` at tomofusion/cpu/utils/mpi_ctvlib.cpp:307,323,362,455,547);
* x-slice planes per neighbour before every 3-D TV stencil pass (reference ring exchange:
  mpi_ctvlib.cpp:400-422; done here before EVERY pass so results equal the single-process ones).

Volumes and sinograms are gathered (``get_recon`` and friends) with ``all_gather`` / ``gather`` of plain tensors
padded to the largest slab -- no pickling, and ``dst=`` keeps the assembled array on one rank only.
"""
from dataclasses import dataclass

import numpy as np


def slab_partition(nslice, world, rank):
    """Contiguous slab of ``rank``: ``(first, count)``.

    ``count = nslice//world (+1 for rank < nslice%world)``; fixes the off-by-formula start of the reference
    (mpi_ctvlib.cpp:48-54, ``first_slice += rank % nproc``)."""
    base, rem = divmod(nslice, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


@dataclass
class SlabComm:
    """Thin wrapper over a ``torch.distributed`` process group for slab engines."""
    group: object = None
    # issue the real collectives even in a group of ONE rank (a single-GPU box can then exercise the RCCL calls, tensor
    # devices and stream ordering of the product's N > 1 path: tests/nccl_world1_script.py)
    force: bool = False

    def __post_init__(self):
        import torch.distributed as dist
        self.dist = dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (launch with torchrun / init_process_group)")
        self.rank = dist.get_rank(self.group)
        self.world = dist.get_world_size(self.group)

    @property
    def prev(self):
        return (self.rank - 1) % self.world

    @property
    def next(self):
        return (self.rank + 1) % self.world

    def global_rank(self, r):
        return r if self.group is None else self.dist.get_global_rank(self.group, r)

    def native_ok(self):
        """True when the engine may run its collectives itself: RCCL from ``libtomo_hip.so`` on the engine's stream
        (``tomo_comm_*``: one ncclGroup per communication round, no second stream, no Python between the kernels) instead of
        ``torch.distributed`` calls.  ``TOMO_NATIVE_COMM=0`` keeps the torch path (both give the same bits)."""
        import os
        return self.on_device() and os.environ.get("TOMO_NATIVE_COMM", "1") != "0"

    def on_device(self):
        """True when the collectives need device tensors (RCCL), False for host tensors (gloo)."""
        return self.dist.get_backend(self.group) == "nccl"

    def allreduce_sum(self, t):
        """In-place sum of tensor ``t`` over ranks (device tensor for RCCL, CPU tensor for gloo)."""
        if self.world > 1 or self.force:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    def allreduce_max(self, t):
        if self.world > 1 or self.force:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return t

    def allreduce_min(self, t):
        if self.world > 1 or self.force:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return t

    def broadcast(self, t, src):
        if self.world > 1 or self.force:
            self.dist.broadcast(t, src=self.global_rank(src), group=self.group)
        return t

    def barrier(self):
        if self.world > 1 or self.force:
            self.dist.barrier(group=self.group)

    def exchange_planes(self, first_planes, last_planes, halo_lo, halo_hi):
        """Ring exchange: my last plane(s) become ``next``'s lo halo, my first plane(s) become ``prev``'s hi halo.
        The tensors may hold several planes; ``halo_lo`` must match the neighbour's ``last_planes`` in size and
        ``halo_hi`` its ``first_planes``."""
        d = self.dist
        if self.world == 1 and not self.force:
            halo_lo.copy_(last_planes)
            halo_hi.copy_(first_planes)
            return
        nxt, prv = self.global_rank(self.next), self.global_rank(self.prev)
        # tags keep the two messages apart when prev == next (world 2); RCCL ignores tags but keeps post order
        ops = [d.P2POp(d.isend, last_planes, nxt, self.group, 1), d.P2POp(d.isend, first_planes, prv, self.group, 2),
               d.P2POp(d.irecv, halo_lo, prv, self.group, 1), d.P2POp(d.irecv, halo_hi, nxt, self.group, 2)]
        for req in d.batch_isend_irecv(ops):
            req.wait()

    def allreduce_with_planes(self, t, first_planes, last_planes, recv_lo, recv_hi):
        """One communication round: sum ``t`` over the ranks AND ring-exchange the planes (as ``exchange_planes``), issued
        together so that the two do not wait for each other (TV descent: sum g^2 and the gradient's boundary slices)."""
        d = self.dist
        if self.world == 1 and not self.force:
            recv_lo.copy_(last_planes)
            recv_hi.copy_(first_planes)
            return t
        work = d.all_reduce(t, op=d.ReduceOp.SUM, group=self.group, async_op=True)
        nxt, prv = self.global_rank(self.next), self.global_rank(self.prev)
        ops = [d.P2POp(d.isend, last_planes, nxt, self.group, 1), d.P2POp(d.isend, first_planes, prv, self.group, 2),
               d.P2POp(d.irecv, recv_lo, prv, self.group, 1), d.P2POp(d.irecv, recv_hi, nxt, self.group, 2)]
        reqs = d.batch_isend_irecv(ops)
        work.wait()
        for req in reqs:
            req.wait()
        return t

    def gather_slabs(self, local, counts, device=None, dst=None):
        """Concatenate the ranks' slabs along axis 0.  ``local``: this rank's (counts[rank], ...) float32 array.
        ``dst=None``: every rank returns the whole array (all_gather); ``dst=r``: only rank r does, the others
        return None (gather).  Uneven slabs are padded to the largest one for the collective."""
        import torch
        if self.world == 1 and not self.force:
            return local
        maxc = max(counts)
        pad = np.zeros((maxc,) + local.shape[1:], np.float32)
        pad[:local.shape[0]] = local
        dev = device if (self.on_device() and device is not None) else "cpu"
        mine = torch.from_numpy(pad).to(dev)
        if dst is None:
            out = torch.empty((self.world * maxc,) + local.shape[1:], dtype=torch.float32, device=dev)
            self.dist.all_gather_into_tensor(out, mine, group=self.group)
            parts = out.cpu().numpy().reshape((self.world, maxc) + local.shape[1:])
        else:
            bufs = [torch.empty_like(mine) for _ in range(self.world)] if self.rank == dst else None
            self.dist.gather(mine, bufs, dst=self.global_rank(dst), group=self.group)
            if self.rank != dst:
                return None
            parts = [b.cpu().numpy() for b in bufs]
        return np.concatenate([parts[r][:counts[r]] for r in range(self.world)], axis=0)

    def all_gather_ints(self, value):
        """One integer per rank, as a list (device ids and the like)."""
        import torch
        if self.world == 1 and not self.force:
            return [int(value)]
        dev = torch.device("cuda", torch.cuda.current_device()) if self.on_device() else "cpu"
        out = torch.zeros(self.world, dtype=torch.int64, device=dev)
        self.dist.all_gather_into_tensor(out, torch.tensor([int(value)], dtype=torch.int64, device=dev), group=self.group)
        return [int(v) for v in out.cpu()]
