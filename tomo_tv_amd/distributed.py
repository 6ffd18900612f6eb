"""Slab sharding of the tilt axis across ranks (one process per GPU, ``torch.distributed``; backend "nccl" is RCCL).

The projector loop (FP, BP, SIRT, SART, clamp, momentum) is independent per x-slice because every slice shares one
system matrix, so ranks own contiguous slabs and exchange nothing there.  Only two things cross ranks:

* scalar partial sums (residual, step norms, TV, ||grad TV||^2): one all-reduce of a float64 element each
  (reference: ``MPI_Allreduce`` at tomofusion/cpu/utils/mpi_ctvlib.cpp:307,323,362,455,547);
* one x-slice plane per neighbour before every 3-D TV stencil pass (reference ring exchange:
  mpi_ctvlib.cpp:400-422; done here before EVERY pass so results equal the single-process ones).
"""
from dataclasses import dataclass


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

    def allreduce_sum(self, t):
        """In-place sum of tensor ``t`` over ranks (device tensor for RCCL, CPU tensor for gloo)."""
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    def allreduce_max(self, t):
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return t

    def broadcast(self, t, src):
        if self.world > 1:
            self.dist.broadcast(t, src=self.global_rank(src), group=self.group)
        return t

    def barrier(self):
        if self.world > 1:
            self.dist.barrier(group=self.group)

    def exchange_planes(self, first_plane, last_plane, halo_lo, halo_hi):
        """Ring exchange: my last plane becomes ``next``'s lo halo, my first plane becomes ``prev``'s hi halo."""
        d = self.dist
        if self.world == 1:
            halo_lo.copy_(last_plane)
            halo_hi.copy_(first_plane)
            return
        nxt, prv = self.global_rank(self.next), self.global_rank(self.prev)
        # tags keep the two messages apart when prev == next (world 2); RCCL ignores tags but keeps post order
        ops = [d.P2POp(d.isend, last_plane, nxt, self.group, 1), d.P2POp(d.isend, first_plane, prv, self.group, 2),
               d.P2POp(d.irecv, halo_lo, prv, self.group, 1), d.P2POp(d.irecv, halo_hi, nxt, self.group, 2)]
        for req in d.batch_isend_irecv(ops):
            req.wait()
