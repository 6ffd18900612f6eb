import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_rccl_world1_plumbing_matches_single_engine(gpu):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, os.path.join(HERE, "nccl_world1_script.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0 and "NCCL_WORLD1_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def test_c_host_shards_through_the_native_communicator(gpu, tmp_path):
    """tests/native/comm_host.c: a plain C program (no Python, no torch) drives the slab-sharded calls of include/tomo_hip.h
    (tomo_comm_init / tomo_comm_tv_gd / tomo_comm_read_scalars ...) on a one-rank RCCL communicator and must reproduce the
    single-slab calls bit for bit.  (mpi_ctvlib.cpp:400-422,455,547 is the host it stands in for.)"""
    root = os.path.dirname(HERE)
    exe = str(tmp_path / "comm_host")
    libdir = os.path.join(root, "tomo_tv_amd")
    subprocess.run(["gcc", "-O2", "-std=gnu11", os.path.join(HERE, "native", "comm_host.c"), "-o", exe, "-L", libdir, "-ltomo_hip",
                    "-lm", "-Wl,-rpath," + libdir], check=True)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "COMM_HOST_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def test_c_host_two_ranks_on_two_gpus(gpu, tmp_path):
    """The same C program as rank 0 and rank 1 of a two-GPU communicator (id passed through a file): activates by itself on a box
    with >= 2 GPUs, skips on the single-GPU boxes of this pool."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    root = os.path.dirname(HERE)
    exe, idf = str(tmp_path / "comm_host"), str(tmp_path / "comm.id")
    libdir = os.path.join(root, "tomo_tv_amd")
    subprocess.run(["gcc", "-O2", "-std=gnu11", os.path.join(HERE, "native", "comm_host.c"), "-o", exe, "-L", libdir, "-ltomo_hip",
                    "-lm", "-Wl,-rpath," + libdir], check=True)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([exe, "2", str(r), idf, str(r)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in (0, 1)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("two-rank C host timed out")
    assert all(p.returncode == 0 for p in procs) and "COMM_HOST_OK" in outs[0][0], str(outs)[-3000:]


def test_one_device_engine_answers_the_multi_gpu_queries(gpu, capsys):
    """``multigpuengine(...)`` / ``multigpufusion(...)`` return the plain one-device classes where one device is left; the
    reference class's own queries (multigpuengine.cpp:385-421) must still work on what comes back."""
    import numpy as np
    from tomo_tv_amd.chemistry import multigpufusion
    from tomo_tv_amd.engine import multigpuengine
    ang = np.deg2rad(np.linspace(-60, 60, 5))
    t = multigpuengine(4, 32, ang, devices=[0])
    assert t.get_gpu_ids() == [0] and t.is_multi_gpu_enabled() is False
    t.print_gpu_usage()
    m = multigpufusion(4, 32, 2, ang, ang, devices=[0])
    assert m.get_gpu_ids() == [0] and m.is_multi_gpu_enabled() is False
    m.print_gpu_usage()
    assert "device 0" in capsys.readouterr().out
