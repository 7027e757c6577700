"""Peer-to-peer sort-last exchange through the C ABI (rdr_composite_*): three processes share one GPU,
open each other's buffers through HIP IPC handles, and every rank must end up with the composited keys
of its rows and the whole gathered image, bit-identical to the single-GPU frame."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run_ranks(tmp_path, world, **extra_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
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
    return outs


@pytest.mark.parametrize("world,split", [(2, "interleaved"), (3, "interleaved"), (2, "contiguous"), (3, "contiguous")])
def test_ranks_sharing_one_gpu_composite_to_the_single_gpu_frame(tmp_path, world, split):
    _run_ranks(tmp_path, world, P2P_SPLIT=split)


@pytest.mark.parametrize("size,env", [("250x141", {}), ("251x139", {}), ("256x144", {"P2P_SKYBOX": "1"}),
                                      ("256x144", {"RDR_COMPOSITE_TILE_FLAGS": "0"}), ("256x144", {"P2P_OCCLUSION": "1"}),
                                      ("251x139", {"P2P_SPLIT": "contiguous"}), ("256x144", {"P2P_REPLICATE": "2.0"}),
                                      ("640x360", {"P2P_REPLICATE": "8.0", "P2P_OCCLUSION": "1"})])
def test_tile_skipping_at_awkward_sizes_with_skybox_and_switched_off(tmp_path, size, env):
    """The exchange moves 32x32-pixel tiles and skips those no rank touched: widths that are not a multiple of four
    (or odd), heights that cut tiles, a skybox (no pixel is clear colour: every tile travels), the flags off, and
    two-phase occlusion culling (the flags gather over both phases of a frame)."""
    _run_ranks(tmp_path, 2, P2P_SIZE=size, **env)


def test_a_peer_that_never_arrives_is_an_error_not_a_hang(tmp_path):
    """The waits between ranks are bounded: rank 1 connects but renders nothing; rank 0's exchange gives up after
    about five seconds, reports RDR_FRAME_PEER_TIMEOUT and fails its next exchange call."""
    outs = _run_ranks(tmp_path, 2, P2P_MODE="timeout")
    assert "rank 0: waited" in outs[0]


def test_ranks_out_of_step_meet_again_after_a_reconnect(tmp_path):
    """A rank that sat out a frame leaves the ranks' exchange counters apart.  The rank that waited gets an error, its
    connection stays broken (no silent reads of half-written keys), and after disconnect + export + connect on both
    ranks the counters are aligned and frames composite exactly again."""
    outs = _run_ranks(tmp_path, 2, P2P_MODE="reconnect")
    assert all("ok" in o for o in outs), outs
