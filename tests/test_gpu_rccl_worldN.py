"""The slab-sharded path over RCCL with MORE THAN ONE rank -- activates by itself on a box with >= 2 GPUs.

The test boxes of this pool have one GPU, so until now nothing with world > 1 had ever run over RCCL (gloo world 2/3 on CPU and a
world-1 RCCL group were the cover).  This test starts 2 and (with >= 3 devices) 3 fresh ranks, one per GPU -- child processes
started by a parent that issues no HIP call of its own (counting devices does not initialise the GPU on this image), never a
re-exec of a process that has -- and runs tests/rccl_worldN_script.py in each."""
import os
import socket
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _visible_devices():
    import torch
    return torch.cuda.device_count()           # does not initialise the GPU (no context is created)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_world(world, timeout=900):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "rccl_worldN_script.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    deadline = time.time() + timeout
    failed = None
    while any(p.poll() is None for p in procs):
        if time.time() > deadline or any(p.poll() not in (None, 0) for p in procs):
            failed = "timeout" if time.time() > deadline else "a rank failed"
            for p in procs:                    # plain children: stop the ranks that would wait in a collective
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.1)
    outs = [p.communicate(timeout=30) for p in procs]
    return failed, [p.returncode for p in procs], outs


def test_rccl_world_script_on_one_rank(gpu):
    """The same script with ONE rank (collectives forced): keeps the script itself exercised on the single-GPU boxes."""
    failed, codes, outs = run_world(1)
    assert failed is None and codes == [0] and "RCCL_WORLD1_OK" in outs[0][0], outs[0][0][-1500:] + outs[0][1][-3000:]


@pytest.mark.parametrize("world", [2, 3])
def test_rccl_world_n_equals_single_engine(world):
    n = _visible_devices()
    if n < world:
        pytest.skip(f"needs {world} GPUs, {n} visible (the single-GPU boxes of this pool: covered by gloo world 2/3 on CPU, "
                    "the thread ring on one GPU and the world-1 RCCL group)")
    failed, codes, outs = run_world(world)
    tail = "\n".join(f"--- rank {r}: rc {c}\n{o[-1500:]}\n{e[-3000:]}" for r, (c, (o, e)) in enumerate(zip(codes, outs)))
    assert failed is None and all(c == 0 for c in codes), tail
    assert f"RCCL_WORLD{world}_OK" in outs[0][0], tail
