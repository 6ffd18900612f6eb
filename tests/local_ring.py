"""N slab engines on ONE GPU, one thread per rank: the product's in-process world (tomo_tv_amd/inprocess.py) with every rank on
the same device.

The GPU boxes the tests run on have a single device, and RCCL refuses two ranks on one device, so the slab-sharded composition
of ``tomo_tv_amd/engine.py`` (which scalars are all-reduced, which planes are exchanged and when, the interior-slab code paths of
the REAL kernels: non-wrapping halos, ``is_first = is_last = 0``) is driven by the product's own thread world -- the one a plain
process uses to spread a volume over several GPUs -- with its barrier-and-copy collectives (``InProcComm``; with one rank per
DEVICE the hot collectives are RCCL groups instead).  All engines run on torch's current (default) stream of the device, so the
order in which the threads issue work is the order the device executes it.
"""
from tomo_tv_amd.inprocess import InProcWorld


class ThreadRing:
    def __init__(self, world):
        self.world = world
        self.w = InProcWorld(world)
        self.w.shared_device = True              # several ranks on one device: no RCCL communicator

    def comm(self, rank):
        return self.w.comm(rank)

    def run(self, fn):
        """Run ``fn(comm)`` on the ``world`` rank threads; returns the list of results, re-raises the first exception."""
        try:
            return self.w.run(lambda r: fn(self.w.comm(r)))
        finally:
            self.w.close()
