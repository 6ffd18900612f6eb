"""Several GPUs driven from ONE plain Python process.

The reference picks its multi-GPU engine by itself whenever more than one device is visible -- ``tomofusion/__init__.py:21-34``
(``determine_gpu_config``), ``gpu/reconstructor.py:23-33`` -- and that engine spreads the slices over the devices from inside one
process (an OpenMP team, one thread per GPU: ``multigpuengine.cpp:140-193``).  Here the same plain ``python script.py`` gets one
slab engine per device, each driven by its own persistent host thread (ctypes drops the GIL during every C call, and HIP's
current device is per thread), composed exactly like the ranks of a ``torchrun`` job: the engines are the ordinary slab engines of
``engine.py`` / ``chemistry.py`` with a communicator whose "ranks" are threads.  The hot collectives (halo rings, the all-reduce of
||grad TV||^2, the scalar sums of an iteration) are the library's own RCCL groups on each engine's stream (``tomo_comm_*``: every
thread joins one communicator with ``ncclCommInitRank``, one rank per device); the cold ones (assembling a volume, broadcasting a
slice, a max over ranks) go through host memory and a thread barrier.  The facade forwards every method call to all engines at
once and returns rank 0's answer (scalars are all-reduced, getters assemble the whole array on every rank).
"""
import os
import queue
import threading

import numpy as np


class InProcWorld:
    """``world`` persistent worker threads, one per slab engine, plus what their communicators share."""

    def __init__(self, world):
        self.world = int(world)
        self.bar = threading.Barrier(self.world)
        self.slots = [None] * self.world
        self._jobs = [queue.Queue() for _ in range(self.world)]
        self._done = queue.Queue()
        self._threads = [threading.Thread(target=self._loop, args=(r,), daemon=True, name=f"tomo-gpu-{r}") for r in range(self.world)]
        for t in self._threads:
            t.start()

    def _loop(self, rank):
        while True:
            job = self._jobs[rank].get()
            if job is None:
                return
            seq, fn = job
            try:
                self._done.put((seq, rank, fn(rank), None))
            except BaseException as e:  # noqa: BLE001 -- re-raised on the calling thread
                self.bar.abort()        # the other ranks may be waiting for this one inside a collective
                self._done.put((seq, rank, None, e))

    def run(self, fn):
        """``fn(rank)`` on every worker at once; the list of results; the first real exception is re-raised.

        Once one rank has failed the others get ``TOMO_INPROC_GRACE`` seconds (default 60) to come back: a rank that sits in a native
        RCCL group or a stream synchronisation waiting for the failed one is not freed by the Python barrier's abort, and a facade
        call must not hang on it forever.  If they do not return, the world is marked dead (every later call raises at once) and
        the original error is raised with a note: the process should end (its worker threads are daemons)."""
        if getattr(self, "dead", None):
            raise RuntimeError(f"in-process multi-GPU world is dead: {self.dead}")
        # every job carries a sequence number: results of a call that was interrupted on this thread (KeyboardInterrupt inside the
        # wait below) arrive late and must not be taken for this call's
        self._seq = seq = getattr(self, "_seq", 0) + 1
        for q in self._jobs:
            q.put((seq, fn))
        out, err = [None] * self.world, [None] * self.world
        got, failed = 0, False
        grace = float(os.environ.get("TOMO_INPROC_GRACE", "60"))
        while got < self.world:
            try:
                s, r, v, e = self._done.get(timeout=grace if failed else None)
                if s != seq:
                    continue            # a stale entry of an interrupted call
            except queue.Empty:
                first = [x for x in err if x is not None][0]
                self.dead = f"rank(s) {[i for i in range(self.world) if err[i] is None and out[i] is None]} did not return within {grace:.0f} s after: {first!r}"
                raise RuntimeError(f"in-process multi-GPU call failed and left ranks waiting ({self.dead}); the world cannot be used any more") from first
            out[r], err[r] = v, e
            failed = failed or e is not None
            got += 1
        if any(e is not None for e in err):
            self.bar.reset()
            real = [e for e in err if e is not None and not isinstance(e, threading.BrokenBarrierError)]
            raise (real or [e for e in err if e is not None])[0]
        return out

    def comm(self, rank):
        return InProcComm(self, rank)

    def close(self):
        if getattr(self, "closed", False):
            return
        self.closed = True
        for q in self._jobs:
            q.put(None)


class InProcComm:
    """The ``SlabComm`` interface (``distributed.py``) between the threads of an ``InProcWorld``: a barrier and copies.  Tensors may
    live on different devices: ``copy_`` / ``to`` order the copy behind both devices' current streams, which is where the engines
    run (``_SlabBackend.enable_torch``)."""
    group = None
    force = False

    def __init__(self, world, rank):
        self.w, self.rank, self.world = world, int(rank), world.world

    @property
    def prev(self):
        return (self.rank - 1) % self.world

    @property
    def next(self):
        return (self.rank + 1) % self.world

    def on_device(self):
        return True

    def native_ok(self):
        """RCCL from the library, one rank per DEVICE: not when two slab engines share a device (RCCL refuses that; the tests that
        put several slabs on one GPU run the barrier-and-copy collectives instead), not when switched off."""
        return not getattr(self.w, "shared_device", False) and os.environ.get("TOMO_NATIVE_COMM", "1") != "0"

    def _all(self, mine):
        """Deposit ``mine``, return everybody's deposits once all have arrived."""
        self.w.slots[self.rank] = mine
        self.w.bar.wait()
        return list(self.w.slots)

    def _reduce(self, t, op):
        if self.world == 1:
            return t
        import torch
        parts = [p.to(t.device) for p in self._all(t)]
        tot = op(torch.stack(parts), 0)
        self.w.bar.wait()               # everybody has formed its total from the un-modified inputs
        t.copy_(tot)
        self.w.bar.wait()
        return t

    def allreduce_sum(self, t):
        import torch
        return self._reduce(t, torch.sum)

    def allreduce_max(self, t):
        import torch
        return self._reduce(t, lambda s, d: torch.max(s, d).values)

    def allreduce_min(self, t):
        import torch
        return self._reduce(t, lambda s, d: torch.min(s, d).values)

    def broadcast(self, t, src):
        if self.world > 1:
            parts = self._all(t)
            if self.rank != src:
                t.copy_(parts[src])
            self.w.bar.wait()
        return t

    def barrier(self):
        if self.world > 1:
            self.w.bar.wait()

    def exchange_planes(self, first_planes, last_planes, halo_lo, halo_hi):
        parts = self._all((first_planes, last_planes))
        halo_lo.copy_(parts[self.prev][1])
        halo_hi.copy_(parts[self.next][0])
        self.w.bar.wait()               # nobody overwrites its send planes before the neighbours have copied them

    def allreduce_with_planes(self, t, first_planes, last_planes, recv_lo, recv_hi):
        self.allreduce_sum(t)
        self.exchange_planes(first_planes, last_planes, recv_lo, recv_hi)
        return t

    def gather_slabs(self, local, counts, device=None, dst=None):
        parts = self._all(local)
        out = np.concatenate(parts, axis=0) if (dst is None or dst == self.rank) else None
        self.w.bar.wait()
        return out

    def all_gather_ints(self, value):
        parts = self._all(int(value))
        self.w.bar.wait()
        return parts


def visible_devices():
    from . import _lib
    return list(range(_lib.device_count()))


def process_group_world():
    """Ranks of the ``torch.distributed`` job this process is part of (1: a plain process)."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size()
    except ImportError:
        pass
    return 1


def process_group_initialized():
    """True inside a ``torch.distributed`` job of any size (a one-rank job still wants the sharded class with its collectives)."""
    try:
        import torch.distributed as dist
        return bool(dist.is_available() and dist.is_initialized())
    except ImportError:
        return False


class InProcessMultiGPU:
    """K slab engines on K devices behind the method table of ONE engine.

    ``make(comm, device)`` builds the slab engine of one rank (called on that rank's thread, so that everything the engine does on
    its device happens on the thread that owns it).  Every method call is forwarded to all engines concurrently -- each call is
    collective over the in-process world exactly as it is over a ``torchrun`` job -- and rank 0's result is returned; plain
    attributes read from rank 0 and are written to every rank (``tomo.tv_eps = 1e-8``)."""

    def __init__(self, make, devices):
        devices = [int(d) for d in devices]
        if len(devices) < 1:
            raise ValueError("no device")
        object.__setattr__(self, "_devices", devices)
        object.__setattr__(self, "_world", InProcWorld(len(devices)))
        self._world.shared_device = len(set(devices)) < len(devices)
        object.__setattr__(self, "_engines", self._world.run(lambda r: make(self._world.comm(r), devices[r])))
        import weakref
        weakref.finalize(self, InProcWorld.close, self._world)       # the worker threads end with the facade (and at interpreter exit)

    def __getattr__(self, name):
        engines = object.__getattribute__(self, "_engines")
        attr = getattr(engines[0], name)
        if not callable(attr):
            return attr
        world = object.__getattribute__(self, "_world")

        def call(*args, **kw):
            return world.run(lambda r: getattr(engines[r], name)(*args, **kw))[0]
        call.__name__ = name
        return call

    def __setattr__(self, name, value):
        if name.startswith("_"):
            object.__setattr__(self, name, value)
            return
        for e in self._engines:
            setattr(e, name, value)

    def __dir__(self):
        return sorted(set(dir(type(self)) + dir(self._engines[0])))

    # the reference's own multi-GPU queries (multigpuengine.cpp:385-421)
    def get_gpu_ids(self):
        return list(self._devices)

    def is_multi_gpu_enabled(self):
        return len(self._devices) > 1

    def print_gpu_usage(self):
        e = self._engines
        print(f"{len(e)} GPUs in one process, one slab engine each: " +
              ", ".join(f"GPU {d}: slices {x.first}..{x.first + x.nloc - 1}" for d, x in zip(self._devices, e)))

    def close(self):
        w = self._world
        try:
            w.run(lambda r: self._engines[r].close() if hasattr(self._engines[r], "close") else None)
        finally:
            w.close()
