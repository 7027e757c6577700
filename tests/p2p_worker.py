"""One rank of the peer-to-peer sort-last test (tests/test_gpu_p2p.py): ``p2p_worker.py rank world dir``.
Ranks are separate processes (they may share one GPU); handles travel through files in ``dir``."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from renderer_amd import Renderer, RendererConfig, synth  # noqa: E402
from renderer_amd.distributed import band_row_index, instance_range, row_range  # noqa: E402
from renderer_amd.scene import Camera  # noqa: E402

W, H = (int(v) for v in os.environ.get("P2P_SIZE", "256x144").split("x"))


def publish(path, data):
    with open(path + ".tmp", "wb") as f:
        f.write(data)
    os.rename(path + ".tmp", path)


def wait_for(paths, seconds=60.0):
    t0 = time.time()
    while not all(os.path.exists(p) for p in paths):
        if time.time() - t0 > seconds:
            raise SystemExit(f"timed out waiting for {[p for p in paths if not os.path.exists(p)]}")
        time.sleep(0.01)


def main():
    rank, world, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    device = int(os.environ.get("P2P_DEVICE", "0"))
    scene = synth.make_field(60, 3, lods=1, alpha_fraction=0.3, with_env=True, prepared=True)
    conf = RendererConfig(cull_enabled=os.environ.get("P2P_OCCLUSION") == "1", skybox=os.environ.get("P2P_SKYBOX") == "1")
    r = Renderer(W, H, conf, device=device)
    r.set_scene(scene)
    if os.environ.get("P2P_SPLIT", "interleaved") == "interleaved":  # chunks of meshlets and bands of rows dealt round-robin
        r.set_partition_interleaved(64, rank, world)
        r.set_shade_bands(32, rank, world)
        r.set_partition_replication(float(os.environ.get("P2P_REPLICATE", "0")))  # > 0: near draws are taken by every rank
        mine = band_row_index(H, 32, rank, world).numpy()
    else:
        r.set_partition(*instance_range(scene.meshlet_visibility_count, rank, world))
        lo, hi = row_range(H, rank, world)
        r.set_shade_rows(lo, hi)
        mine = np.arange(lo, hi)
    publish(os.path.join(d, f"handle_{rank}.bin"), r.composite_export())
    names = [os.path.join(d, f"handle_{k}.bin") for k in range(world)]
    wait_for(names)
    r.composite_connect(rank, world, [open(n, "rb").read() for n in names])

    if os.environ.get("P2P_MODE") == "timeout":
        # rank 0 runs a frame, rank 1 never does: rank 0's wait must give up by itself and say so
        from renderer_amd import RendererError, abi
        if rank == 0:
            t0 = time.time()
            r.render_visbuffer()
            r.composite_exchange()
            flags = r.frame_status()  # waits for the stream: returns once the bounded wait has given up
            waited = time.time() - t0
            ok = bool(flags & abi.FRAME_PEER_TIMEOUT) and 3.0 < waited < 30.0
            try:
                r.composite_exchange()
                ok = False
            except RendererError as e:
                ok = ok and "peer rank" in str(e)
            print(f"rank 0: waited {waited:.1f} s, flags {flags:#x}, {'ok' if ok else 'MISMATCH'}", flush=True)
            publish(os.path.join(d, "timeout_seen"), b"1")
        else:
            wait_for([os.path.join(d, "timeout_seen")], 90.0)
            ok = True
        publish(os.path.join(d, f"done_{rank}"), b"1")
        wait_for([os.path.join(d, f"done_{k}") for k in range(world)])
        try:
            r.composite_disconnect()
        except RendererError:
            pass
        r.close()
        raise SystemExit(0 if ok else 1)

    if os.environ.get("P2P_MODE") == "reconnect":
        # Rank 1 sits out one frame (as a rank whose frame failed would): rank 0's exchange gives up after its bounded
        # wait, its connection is broken from then on, and both ranks disconnect, export and connect again.  The two
        # renderers have then run different numbers of exchanges; the connect aligns them, and the frames that follow
        # must composite exactly.
        from renderer_amd import RendererError, abi
        ok = True

        def frame():
            r.render_visbuffer()
            r.composite_exchange()
            r.shade()
            r.composite_gather()
            r.sync()

        frame()  # both ranks
        if rank == 0:
            try:
                frame()  # alone: the wait for rank 1 times out
                ok = bool(r.frame_status() & abi.FRAME_PEER_TIMEOUT)
                try:
                    r.composite_exchange()
                    ok = False
                except RendererError as e:
                    ok = ok and ("peer rank" in str(e) or "out of step" in str(e))
                try:
                    r.render_visbuffer()
                    r.composite_exchange()  # still broken after the error was reported once
                    ok = False
                except RendererError as e:
                    ok = ok and "out of step" in str(e)
            except RendererError as e:
                ok = "peer rank" in str(e)
        publish(os.path.join(d, f"phase1_{rank}"), b"1")
        wait_for([os.path.join(d, f"phase1_{k}") for k in range(world)], 90.0)
        try:
            r.composite_disconnect()
        except RendererError:
            pass
        publish(os.path.join(d, f"handle2_{rank}.bin"), r.composite_export())
        names = [os.path.join(d, f"handle2_{k}.bin") for k in range(world)]
        wait_for(names)
        r.composite_connect(rank, world, [open(n, "rb").read() for n in names])
        whole = Renderer(W, H, conf, device=device)
        whole.set_scene(scene)
        whole.render()
        vis, color = whole.read_visbuffer_u64(), whole.read_color()
        for k in range(3):
            frame()
            gv, gc = r.read_visbuffer_u64(), r.read_color()
            if not (np.array_equal(gv[mine], vis[mine]) and np.array_equal(gc, color)):
                print(f"rank {rank}: frame {k} after the reconnect differs", flush=True)
                ok = False
            publish(os.path.join(d, f"re_{k}_{rank}"), b"1")
            wait_for([os.path.join(d, f"re_{k}_{q}") for q in range(world)])
        r.composite_disconnect()
        r.close()
        whole.close()
        print(f"rank {rank}: {'ok' if ok else 'MISMATCH'}", flush=True)
        raise SystemExit(0 if ok else 1)

    whole = Renderer(W, H, conf, device=device)  # the single-GPU answer, in this process
    whole.set_scene(scene)
    base = scene.camera
    cams = [base, Camera(tuple(np.add(base.position, (3.0, 1.0, -2.0))), base.direction),
            Camera(tuple(np.add(base.position, (-4.0, 2.0, 5.0))), tuple(np.add(base.direction, (0.1, 0.05, 0.0)))),
            # looking away: an empty frame (every tile the last frame covered must be cleared in every peer's image), then back
            Camera(base.position, tuple(-np.asarray(base.direction))), base]
    frames = []
    for cam in cams:  # frames back to back, no host synchronisation inside a frame
        r.render_visbuffer(cam)
        r.composite_exchange()
        r.shade()
        r.composite_gather()
        r.sync()
        frames.append((r.read_visbuffer_u64().copy(), r.read_color().copy()))
        # every rank must have read this frame before any rank renders over its buffers: the exchange orders the
        # device side by itself, this barrier is only for the host readbacks above
        publish(os.path.join(d, f"read_{len(frames)}_{rank}"), b"1")
        wait_for([os.path.join(d, f"read_{len(frames)}_{k}") for k in range(world)])
    ok = True
    for k, cam in enumerate(cams):
        whole.render(cam)
        vis, color = whole.read_visbuffer_u64(), whole.read_color()
        gv, gc = frames[k]
        if not np.array_equal(gv[mine], vis[mine]):
            print(f"rank {rank} frame {k}: {int((gv[mine] != vis[mine]).sum())} composited keys differ", flush=True)
            ok = False
        if not np.array_equal(gc, color):
            print(f"rank {rank} frame {k}: {int((gc != color).any(axis=2).sum())} pixels of the gathered image differ", flush=True)
            ok = False
    publish(os.path.join(d, f"done_{rank}"), b"1")
    wait_for([os.path.join(d, f"done_{k}") for k in range(world)])
    r.composite_disconnect()
    r.close()
    whole.close()
    print(f"rank {rank}: {'ok' if ok else 'MISMATCH'}", flush=True)
    raise SystemExit(0 if ok else 1)


if __name__ == "__main__":
    main()
