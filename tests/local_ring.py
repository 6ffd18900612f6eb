"""In-process stand-in for a ``torch.distributed`` group: N slab engines on ONE GPU, one Python thread per rank.

The GPU boxes the tests run on have a single device, and RCCL refuses two ranks on one device, so the slab-sharded
composition of ``tomo_tv_amd/engine.py`` (which scalars are all-reduced, which planes are exchanged and when, the
interior-slab code paths of the REAL kernels: non-wrapping halos, ``is_first = is_last = 0``) is driven here by
threads: rank r's engine calls the same ``SlabComm`` methods, implemented with a barrier and device-to-device copies.
All engines run on torch's current (default) stream of the device, so the order in which the threads issue work is the
order the device executes it; a barrier therefore also orders one thread's kernels before another thread's copies.
Test infrastructure only: nothing in ``tomo_tv_amd`` imports it.
"""
import threading

import numpy as np
import torch


class ThreadRing:
    def __init__(self, world):
        self.world = world
        self.bar = threading.Barrier(world)
        self.slots = [None] * world

    def comm(self, rank):
        return RingComm(self, rank)

    def run(self, fn):
        """Run ``fn(comm)`` in ``world`` threads; returns the list of results, re-raises the first exception."""
        out, err = [None] * self.world, [None] * self.world

        def body(r):
            try:
                out[r] = fn(self.comm(r))
            except BaseException as e:  # noqa: BLE001 -- reported to the caller below
                err[r] = e
                self.bar.abort()
        ths = [threading.Thread(target=body, args=(r,)) for r in range(self.world)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        real = [e for e in err if e is not None and not isinstance(e, threading.BrokenBarrierError)]
        if real or any(err):
            raise (real or [e for e in err if e is not None])[0]
        return out


class RingComm:
    """The methods of ``tomo_tv_amd.distributed.SlabComm`` that the engines use."""
    group = None

    def __init__(self, ring, rank):
        self.ring, self.rank, self.world = ring, rank, ring.world

    @property
    def prev(self):
        return (self.rank - 1) % self.world

    @property
    def next(self):
        return (self.rank + 1) % self.world

    def on_device(self):
        return True

    def _all(self, mine):
        """Deposit ``mine``, return everybody's deposits once all have arrived."""
        self.ring.slots[self.rank] = mine
        self.ring.bar.wait()
        return list(self.ring.slots)

    def _reduce(self, t, op):
        if self.world == 1:
            return t
        tot = op(torch.stack(self._all(t)), 0)
        self.ring.bar.wait()            # everybody has formed its total from the un-modified inputs
        t.copy_(tot)
        self.ring.bar.wait()
        return t

    def allreduce_sum(self, t):
        return self._reduce(t, torch.sum)

    def allreduce_max(self, t):
        return self._reduce(t, lambda s, d: torch.max(s, d).values)

    def broadcast(self, t, src):
        if self.world > 1:
            parts = self._all(t)
            if self.rank != src:
                t.copy_(parts[src])
            self.ring.bar.wait()
        return t

    def barrier(self):
        if self.world > 1:
            self.ring.bar.wait()

    def exchange_planes(self, first_planes, last_planes, halo_lo, halo_hi):
        parts = self._all((first_planes, last_planes))
        halo_lo.copy_(parts[self.prev][1])
        halo_hi.copy_(parts[self.next][0])
        self.ring.bar.wait()            # nobody overwrites its send planes before the neighbours have copied them

    def allreduce_with_planes(self, t, first_planes, last_planes, recv_lo, recv_hi):
        self.allreduce_sum(t)
        self.exchange_planes(first_planes, last_planes, recv_lo, recv_hi)
        return t

    def gather_slabs(self, local, counts, device=None, dst=None):
        parts = self._all(local)
        out = np.concatenate(parts, axis=0) if (dst is None or dst == self.rank) else None
        self.ring.bar.wait()
        return out

    def all_gather_ints(self, value):
        parts = self._all(int(value))
        self.ring.bar.wait()
        return parts
