#!/usr/bin/env python3
"""Benchmark of the per-frame hot path: cull -> raster to the 64-bit visibility buffer ->
(sort-last composite) -> deferred PBR shade -> sRGB store.

  python bench.py --gpus N --steps K --warmup W

A step is one frame with the scene already resident in HBM.  The workload is the instanced
icosphere field at 3840x2160 ("Field-S": 1 250 instances x 20 480 triangles = 25.6 M triangles
per GPU), the stand-in for the Sponza / Bistro configurations whose assets do not ship with the
reference.  With N > 1 there is one process per GPU: started by torch.distributed.run, or, when
WORLD_SIZE is not set, by this script itself, which then spawns its N ranks as fresh child
processes before anything touches a GPU.  Every rank holds the whole scene and rasterises its share
of the meshlet instances (chunks of 64 meshlets dealt round-robin, ``--split interleaved``, or a
contiguous N-th); the full-frame visibility buffers are composited with a 64-bit MIN (peer to peer
over xGMI, or an RCCL reduce-scatter), every rank shades its rows (32-row bands dealt round-robin, or
a contiguous strip) and the RGBA8 rows are gathered.  ``--scaling strong`` (the default) renders the same field at every N, ``weak``
an N-times larger one.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def algorithmic_bytes(st, width, height, draws):
    """Bytes each kernel must move per frame, from the frame's own counters: struct sizes are
    the reference's GPU layouts (16 B draw, 64 B transform, 160 B mesh, 32 B meshlet, 32 B cull
    command, 16 B visible-meshlet record, 4 B vertex id + 12 B position, 3 B per triangle,
    8 B per visibility key, 64 B vertex, 4 B RGBA8, 4 B covered-pixel list entry + 4 B key word,
    4 B instance -> draw, 16 B per-frame draw record, 12 B triangle -> vertices entry)."""
    px = width * height
    cov = st["pixels_covered"]
    return {
        # the clear kernel proper only zeroes list heads and tile counts: the 8 B per pixel of the visibility
        # buffer are stored by extra workgroups of the meshlet cull launch and are charged there
        "clear": (width // 64 + 1) * (height // 64 + 1) * 4 + 13000,
        "draw_cull": draws * (16 + 64 + 160) + st["commands"] * 32,
        "meshlet_cull": st["commands"] * 32 + st["meshlets_tested"] * 32 + st["meshlets_visible"] * 16 + px * 8,
        "raster": st["meshlets_visible"] * (16 + 32) + st["vertices_transformed"] * (4 + 12)
                  + st["triangles_tested"] * 3 + st["fragments"] * 8,
        # classification: every key read once, a list entry per covered pixel, the clear colour elsewhere
        "classify": px * 8 + cov * 8 + (px - cov) * 4,
        # shadow pass: list entry + key word + draw id + draw record + vertex ids + 3 positions per covered pixel,
        # 1 B result; BVH nodes are shared (L2)
        "shadow": cov * (4 + 4 + 4 + 16 + 12 + 3 * 12 + 1),
        "shade": cov * (4 + 4 + 4 + 16 + 12 + 1 + 3 * 64 + 4),
    }


def cpu_baseline(scene, width, height, frames):
    """The CPU oracle (a restatement of the reference path, not lavapipe) on this host's cores."""
    from oracle import orc
    from renderer_amd import RendererConfig
    o = orc.OracleRenderer(width, height, RendererConfig())
    o.set_scene(scene)
    o.render()  # warm-up (page-in, thread pool)
    t0 = time.perf_counter()
    for _ in range(frames):
        o.render()
    dt = (time.perf_counter() - t0) / frames
    return dt, orc.num_threads()


def secondary_series(W, H, frames, instances, subdivisions):
    """Shorter runs of the other workloads the measurement plan names, after the timed region."""
    import torch
    from renderer_amd import Renderer, RendererConfig, synth
    cases = [
        ("field, lod on (simplified chain of up to 8 levels)", lambda: synth.make_field(instances, subdivisions, lods=8, with_env=True, prepared=True),
         RendererConfig(lod_enabled=True, cull_enabled=False, collect_stats=True)),
        ("field, two-phase occlusion culling on", lambda: synth.make_field(instances, subdivisions, lods=1, with_env=True, prepared=True),
         RendererConfig(lod_enabled=False, cull_enabled=True, collect_stats=True)),
        ("field, 50% of draws alpha-masked (checkerboard albedo, cutoff 0.5)",
         lambda: synth.make_field(instances, subdivisions, lods=1, alpha_fraction=0.5, with_env=True, prepared=True),
         RendererConfig(lod_enabled=False, cull_enabled=False, collect_stats=True)),
        ("atrium (interior stand-in for Sponza: every pixel covered, textured, large clipped triangles), skybox on",
         lambda: synth.make_atrium(columns=10, spheres=60, sphere_subdiv=5),
         RendererConfig(cull_enabled=False, collect_stats=True, skybox=True)),
        ("Field-XL: 50 000 instances (1.02 G triangles submitted) at 7680x4320, the largest configuration, on one GPU",
         lambda: synth.make_field(50000, subdivisions, lods=1, with_env=True, prepared=True),
         RendererConfig(lod_enabled=False, cull_enabled=False, collect_stats=True)),
    ]
    out = []
    for name, make, conf in cases:
        scene = make()
        w, h = (7680, 4320) if name.startswith("Field-XL") else (W, H)
        r = Renderer(w, h, conf)
        r.set_scene(scene)
        for _ in range(5):
            r.render()
        r.sync()
        t0 = time.perf_counter()
        for _ in range(frames):
            r.render()
        r.sync()
        dt = (time.perf_counter() - t0) / frames
        st = r.stats()
        out.append({"workload": name, "width": w, "height": h, "frames_per_s": 1.0 / dt, "ms_per_frame": dt * 1e3,
                    "Mtri_per_s_submitted": st["triangles_submitted"] / dt / 1e6,
                    "Mtri_per_s_rasterised": st["triangles_rasterised"] / dt / 1e6,
                    "fragments": st["fragments"], "pixels_covered": st["pixels_covered"]})
        r.close()
    out.append(camera_path_series(W, H, instances, subdivisions))
    # throughput with two frames in flight, as the reference's frame loop keeps up to three
    # (renderer/src/renderer.rs:919-922): two renderers take turns, each on its own stream
    scene = synth.make_field(instances, subdivisions, lods=1, with_env=True, prepared=True)
    conf = RendererConfig(lod_enabled=False, cull_enabled=False, collect_stats=True)
    pair = [Renderer(W, H, conf), Renderer(W, H, conf)]
    for r in pair:
        r.set_scene(scene)
    for k in range(10):
        pair[k & 1].render()
    for r in pair:
        r.sync()
    t0 = time.perf_counter()
    for k in range(2 * frames):
        pair[k & 1].render()
    for r in pair:
        r.sync()
    dt = (time.perf_counter() - t0) / (2 * frames)
    st = pair[0].stats()
    out.append({"workload": "field (the headline workload), two frames in flight on two streams", "frames_per_s": 1.0 / dt,
                "ms_per_frame": dt * 1e3, "Mtri_per_s_submitted": scene.triangle_count() / dt / 1e6,
                "Mtri_per_s_rasterised": st["triangles_rasterised"] / dt / 1e6,
                "fragments": st["fragments"], "pixels_covered": st["pixels_covered"]})
    for r in pair:
        r.close()
    torch.cuda.synchronize()
    return out


def camera_path(positions=120):
    """The camera poses of the moving-camera series as (position, direction) pairs: one closed loop over the field, a
    sideways sway of 30 units with a rise and fall of 5, a dolly of 15 units towards the field and a yaw of +-0.25 rad."""
    import math
    norm = math.sqrt(1.0 + 0.35 * 0.35)
    poses = []
    for k in range(positions):
        a = 2.0 * math.pi * k / positions
        yaw = 0.25 * math.sin(a)
        poses.append(((30.0 * math.sin(a), 25.0 + 5.0 * math.sin(2.0 * a), -60.0 + 7.5 * (1.0 - math.cos(a))),
                      (math.sin(yaw) / norm, 0.35 / norm, -math.cos(yaw) / norm)))
    return poses


def camera_path_series(W, H, instances, subdivisions, positions=120):
    """The field seen from a camera that moves: one closed loop of `positions` camera poses (a sideways sway of 30
    units with a rise and fall of 5, a dolly of 15 units towards the field and a yaw of +-0.25 rad), two-phase occlusion
    culling on, so every frame's first phase draws what the frame before it saw.  The first loop is the warm-up, the
    second gives the mean (wall clock over the loop, frames queued back to back like the other series), and a third,
    on a renderer that brackets every frame's kernels with events and is read back after every frame, the spread over
    the poses: median, minimum and maximum of a frame's summed kernel time (the brackets add about 0.03 ms to a frame)."""
    from renderer_amd import Renderer, RendererConfig, synth
    from renderer_amd.scene import Camera
    scene = synth.make_field(instances, subdivisions, lods=1, with_env=True, prepared=True)
    path = [Camera(pos, direction) for pos, direction in camera_path(positions)]
    r = Renderer(W, H, RendererConfig(lod_enabled=False, cull_enabled=True, collect_stats=True))
    r.set_scene(scene)
    for cam in path:
        r.render(cam)
    r.sync()
    t0 = time.perf_counter()
    for cam in path:
        r.render(cam)
    r.sync()
    mean = (time.perf_counter() - t0) / positions * 1e3
    st = r.stats()
    r.close()
    r = Renderer(W, H, RendererConfig(lod_enabled=False, cull_enabled=True, collect_stats=True, time_kernels=1))
    r.set_scene(scene)
    for cam in path[-5:]:
        r.render(cam)
    r.stats()
    per_frame = []
    for cam in path:
        r.render(cam)
        per_frame.append(sum(r.stats()["kernel_ms"].values()))
    r.close()
    per_frame.sort()
    return {"workload": "field, camera on a closed path of %d poses, two-phase occlusion culling on" % positions,
            "width": W, "height": H, "frames_per_s": 1e3 / mean, "ms_per_frame": mean,
            "kernel_ms_per_frame_median": per_frame[positions // 2], "kernel_ms_per_frame_min": per_frame[0],
            "kernel_ms_per_frame_max": per_frame[-1],
            "Mtri_per_s_submitted": scene.triangle_count() / mean / 1e3,
            "Mtri_per_s_rasterised": st["triangles_rasterised"] / mean / 1e3,
            "fragments": st["fragments"], "pixels_covered": st["pixels_covered"]}


def spawn_ranks(n):
    """Start the N ranks of a multi-GPU run as fresh child processes of this (GPU-free) parent:
    same command line, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment.  Rank 0's
    stdout (the JSON line) is this process's stdout; the exit code is non-zero if any rank's is."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if rank == 0 else sys.stderr))
    codes = []
    failed_at = None
    while len(codes) < n:  # a rank that dies leaves the others waiting in a collective: end them too
        for p in procs:
            if p.poll() is not None and p.returncode != 0 and failed_at is None:
                failed_at = time.time()
        if all(p.poll() is not None for p in procs):
            break
        if failed_at is not None and time.time() - failed_at > 20:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.2)
    codes = [p.wait() for p in procs]
    return next((c for c in codes if c != 0), 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--instances", type=int, default=1250, help="instances of the field (per GPU with --scaling weak)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1: strong = the same scene at every N (the fixed-scene configurations), weak = N times the instances")
    ap.add_argument("--subdivisions", type=int, default=5)
    ap.add_argument("--cpu-frames", type=int, default=40, help="CPU-baseline frames, about 10 s of CPU work (0 = skip)")
    ap.add_argument("--secondary-frames", type=int, default=50,
                    help="single GPU only: frames for the secondary series (LOD on, alpha-masked, atrium); 0 = skip")
    ap.add_argument("--composite", choices=("auto", "p2p", "reduce_scatter", "all_reduce"), default="auto",
                    help="exchange step of the sort-last composite (auto: the library's peer-to-peer exchange over xGMI when "
                         "the ranks can open each other's buffers and a probe frame matches the collective, else RCCL reduce-scatter)")
    ap.add_argument("--split", choices=("interleaved", "contiguous"), default="interleaved",
                    help="multi-GPU only: how meshlet instances and rows are dealt to the ranks")
    ap.add_argument("--frames-in-flight", type=int, default=int(os.environ.get("RDR_FRAMES_IN_FLIGHT", "2")),
                    help="multi-GPU only: renderers taking turns so that one frame's exchange overlaps the next frame's raster")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="gloo: rehearsal of the multi-rank path with the ranks sharing the visible GPUs and the exchange "
                         "staged through the host (checks the plumbing; not a measurement)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="run the composite / gather collectives even with one rank (rehearsal)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher and never touches a GPU
        raise SystemExit(spawn_ranks(args.gpus))

    # stdout carries exactly one JSON line: native libraries write to file descriptor 1 on their own (RCCL's
    # version banner, gloo's connection messages), so everything but that line is sent to stderr
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    from renderer_amd import Renderer, RendererConfig, synth

    if os.environ.get("RDR_BENCH_WATCHDOG"):  # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["RDR_BENCH_WATCHDOG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    multi = world > 1 or args.force_collectives
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    W, H = args.width, args.height
    # the sphere goes through the library's mesh preparation, as every mesh of the reference goes through append_mesh
    total_instances = args.instances * world if args.scaling == "weak" else args.instances
    scene = synth.make_field(total_instances, args.subdivisions, lods=1, with_env=True, prepared=True)
    tris_per_frame = scene.triangle_count()
    # occlusion culling (cull_enabled) never changes the image; on this scene nothing hides behind anything, so the
    # second phase and the depth pyramid would only cost time (series "occlusion culling on" below)
    # kernel durations come from event brackets inside the timed region; bracketing costs about 33 us per frame (8 % of
    # this frame), so only every n-th frame is bracketed and the others run as a caller without timers runs them:
    # every 8th, or more often when the run is short, so that at least five frames are bracketed
    bracket_every = min(8, max(1, args.steps // 5))
    conf = RendererConfig(lod_enabled=False, cull_enabled=False, collect_stats=True, time_kernels=bracket_every)
    def make_renderer():
        rr = Renderer(W, H, conf, device=local_rank)
        rr.set_scene(scene)
        return rr

    frame = None
    if multi:
        from renderer_amd.distributed import SortLastPipeline
        frame = SortLastPipeline(make_renderer, scene.meshlet_visibility_count, H, rank, world,
                                 depth=args.frames_in_flight, composite=args.composite, split=args.split)
        r = frame.renderers()[0]
    else:
        r = make_renderer()
        r.set_stream(torch.cuda.current_stream().cuda_stream)

    def step():
        if frame is not None:
            frame.step()  # raster own share -> u64-min exchange -> shade own rows -> all-gather, on the slot's stream
        else:
            r.render()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    r.stats()  # drop the warm-up frames from the kernel timers
    if frame is not None:
        frame.time_composite()  # device time stamps around the exchange and the gather of the timed frames
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = r.stats()  # counters of the last frame + mean kernel times over the timed frames
    composite_ms = frame.composite_ms() if frame is not None else None

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        fps = args.steps / elapsed
        kms = st["kernel_ms"]
        ab = algorithmic_bytes(st, W, H, len(scene.mesh_draws))
        dom = max(("raster", "shade", "shadow", "classify", "meshlet_cull", "clear", "draw_cull"), key=lambda k: kms[k])
        if dom == "raster":
            ab_dom, t_dom = ab["raster"], kms["raster"]
        else:
            ab_dom, t_dom = ab[dom], kms[dom]
        achieved = ab_dom / (t_dom * 1e-3) / 1e9 if t_dom > 0 else 0.0
        # HBM traffic and vector-issue figures of the dominant kernel are NOT measured by this run (counters need
        # rocprofv3 passes of their own): they are read from the committed record of profiles/refresh_r3.sh on the
        # same workload, and the line says so (traffic_source)
        traffic = None
        issue = None
        raster_issue = None
        traffic_source = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("workload") == [args.instances, args.subdivisions, W, H] and world == 1:
                    traffic = rec.get("kernels", {}).get(dom)
                    issue = rec.get("issue", {}).get(dom)
                    raster_issue = rec.get("issue", {}).get("raster")
                    traffic_source = "profiles/pmc_traffic.json (rocprofv3 --pmc passes of profiles/refresh_r3.sh, committed; not this run)"
            except Exception:
                traffic = None
        out = {
            "metric": "Mtri/s",
            "value": tris_per_frame * fps / 1e6,
            "unit": "Mtri/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "frames_per_s": fps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32+u64",
            "data": "synthetic",
            "config": {
                "workload": f"field: {total_instances} instances x icosphere-{args.subdivisions} "
                            f"({tris_per_frame / 1e6:.1f} Mtri submitted per frame, meshlets from the library's mesh preparation), "
                            f"{W}x{H}, lod off, occlusion culling off; "
                            "stand-in for the Sponza/Bistro 4K configs (assets not shipped)",
                "instances": total_instances, "width": W, "height": H,
                "ranks": dist.get_world_size() if multi else 1, "backend": (dist.get_backend() if multi else None),
                "parallelism": "single GPU" if frame is None else
                               f"sort-last x{world} ({args.split} shares), u64-min {frame.composite.replace('_', '-')} + row-split shade + RGBA8 row gather, "
                               f"{len(frame.slots)} frames in flight" + ("" if args.backend == "nccl" else " [gloo rehearsal]"),
            },
            "kernel_ms": {k: round(v, 4) for k, v in kms.items() if v > 0},
            # the brackets tile the frame (each spans its group of launches, dependency latency included) and are taken on
            # every n-th frame, which they lengthen by about 33 us: the difference to ms_per_step is that and the host's gaps
            "kernel_ms_frames": st["frames_timed"], "kernel_ms_every": bracket_every,
            "kernel_ms_sum": round(sum(kms.values()), 4),
            "frame_minus_kernels_ms": round(ms - sum(kms.values()), 4),
            "launches_per_frame": {"all_opaque_no_occlusion": 8, "with_large_triangles": 10, "with_alpha_tested_draws": 12,
                                   "note": "dependent kernel launches of one frame on this path (draw cull, meshlet cull, raster, tile path, "
                                           "classify, shadow, shadow walk, shade; the tile path is one launch after a frame without large or "
                                           "mid-sized triangles, else tile set-up, mid + fill, tile raster; alpha-tested draws add two)"},
            "frame_counters": {k: st[k] for k in ("draws_visible", "meshlets_tested", "meshlets_visible",
                                                  "triangles_submitted", "triangles_tested", "triangles_rasterised",
                                                  "fragments", "pixels_covered", "vertices_transformed")},
            "roofline": {"kernel": "k_" + dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         # the contract's bound is HBM or MFMA; of the two this path can only meet HBM, but none of the three
                         # large kernels is limited by bandwidth (valu_issue: the SIMD cycles the launch had per vector
                         # instruction it issued, the busy share that is at the measured costs of an instruction, the lanes active)
                         "limited_by": {"raster": "vector instruction issue (valu_issue: SIMD cycles per wave64 vector instruction against a cost of 2.3-2.8 "
                                                  "cycles with register operands and 4.0-4.5 with a scalar operand, 64-bit, packed or cross-lane, "
                                                  "profiles/valu_rate.hip; four waves per SIMD run it as fast as five), then the rate of scattered 64-bit atomics",
                                        "shade": "vector instruction issue (valu_issue), then the vector memory path",
                                        "shadow": "vector instruction issue, then dependent-load latency of the tree walk"}.get(dom, "memory"),
                         "algorithmic_bytes": ab_dom, "kernel_ms": t_dom,
                         "frame_algorithmic_bytes": sum(ab.values()),
                         "frame_frac": sum(ab.values()) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         # not memory-bound: what it is bound by instead (PMC passes of profiles/pmc_passes.sh)
                         "valu_issue": issue},
        }
        if composite_ms is not None:  # device time of the exchange steps (rank 0, mean per frame; overlapped with the other slot's raster)
            out["composite_ms"] = {"keys": round(composite_ms[0], 4), "rows": round(composite_ms[1], 4), "form": frame.composite}
        raster_s = (kms["raster"] + kms["big_raster"]) * 1e-3
        if raster_s > 0:  # the rasteriser's own roofline is the L2 atomic rate (per-lane and per-wave scans issue one per fragment)
            out["raster_atomics"] = {"fragments": st["fragments"], "per_s": st["fragments"] / raster_s,
                                     "bytes_per_s": st["fragments"] * 8 / raster_s,
                                     "note": "64-bit atomicMin issued per fragment by k_raster / k_mid_raster; the tile path merges without atomics"}
            try:
                # the ceiling, measured now on this GPU: a kernel that does nothing but issue the same atomic in the per-lane
                # scan's address pattern (64 boxes of 2 x 2 pixels scattered over 32 x 32 pixels per wave; profiles/atomic_peak.hip)
                sys.path.insert(0, os.path.join(ROOT, "profiles"))
                import atomic_peak
                torch.cuda.synchronize()
                peak = atomic_peak.rates(local_rank, repeats=3, full=False)["box_2x2_region32"] * 1e9
                if peak > 0:
                    out["raster_atomics"].update({"peak": peak, "frac": st["fragments"] / raster_s / peak, "unit": "atomics/s",
                                                  "peak_source": "profiles/atomic_peak.hip, pattern box_2x2_region32, measured in this run"})
                    # its two throughput limits side by side: the share of the atomic ceiling, and the share of its cycles the
                    # SIMDs spent issuing vector instructions (from the committed PMC record).  The second is the one the
                    # kernel is at: a build without the atomic runs no faster (profiles/r3/raster_without_atomics.txt)
                    out["raster_atomics"]["valu_issue_frac"] = [(raster_issue or {}).get("valu_busy_lo"), (raster_issue or {}).get("valu_busy_hi")]
                    out["raster_atomics"]["limits"] = "vector instruction issue first, the atomic rate second"
            except Exception as e:  # the helper library is optional
                out["raster_atomics"]["peak_error"] = str(e)
        if world == 1 and frame is None and args.secondary_frames > 0:
            out["secondary"] = secondary_series(W, H, args.secondary_frames, args.instances, args.subdivisions)
        if world == 1 and args.cpu_frames > 0:
            dt, threads = cpu_baseline(scene, W, H, args.cpu_frames)
            out["cpu_baseline"] = {"value": tris_per_frame / dt / 1e6, "unit": "Mtri/s", "cores": threads,
                                   "kind": "port", "frames_per_s": 1.0 / dt,
                                   "sample": f"{args.cpu_frames} full frames of the same workload "
                                             "(CPU restatement of the reference path, OpenMP; not lavapipe)"}
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if frame is not None:
        frame.close()
    else:
        r.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
