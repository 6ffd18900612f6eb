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
