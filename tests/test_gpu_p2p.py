"""Peer-to-peer sort-last exchange through the C ABI (rdr_composite_*): three processes share one GPU,
open each other's buffers through HIP IPC handles, and every rank must end up with the composited keys
of its rows and the whole gathered image, bit-identical to the single-GPU frame."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_one_gpu_composite_to_the_single_gpu_frame(tmp_path, world):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "p2p_worker.py"), str(k), str(world), str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for k in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
