#!/usr/bin/env python3
"""How evenly a sort-last split deals the work out, measured on ONE GPU: for N in {2, 4, 8} every rank's share is
rendered alone (no exchange) and timed with the library's own event brackets.

  geometry(k)  the frame's cull + raster kernels with rank k's share of the meshlet instances
  shading(k)   classify + shadow + shade of rank k's rows of the complete (composited) visibility buffer

The slowest rank sets the frame at N GPUs, so max / mean over the ranks is what strong scaling loses to the
split before any exchange cost.  Both splits of renderer_amd.distributed are measured: contiguous N-ths, and
chunks of meshlets / bands of rows dealt round-robin.

  python profiles/rank_alone.py [--frames 20] [--scenes field,atrium,field_xl] [--out gpurun_out/rank_alone.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from renderer_amd import Renderer, RendererConfig, synth  # noqa: E402
from renderer_amd.distributed import DEFAULT_BAND, DEFAULT_CHUNK, DEFAULT_REPLICATE_PIXELS, instance_range, row_range  # noqa: E402

GEOMETRY = ("clear", "draw_cull", "meshlet_cull", "raster", "big_raster")
SHADING = ("classify", "shadow", "shade")


def scenes(names):
    out = []
    if "field" in names:
        out.append(("field (1250 x icosphere-5), 3840x2160", lambda: synth.make_field(1250, 5, lods=1, with_env=True, prepared=True),
                    3840, 2160, RendererConfig(lod_enabled=False, cull_enabled=False, collect_stats=True, time_kernels=1)))
    if "atrium" in names:
        out.append(("atrium, 3840x2160", lambda: synth.make_atrium(columns=10, spheres=60, sphere_subdiv=5),
                    3840, 2160, RendererConfig(cull_enabled=False, collect_stats=True, time_kernels=1, skybox=True)))
    if "atrium8k" in names:
        out.append(("atrium, 7680x4320", lambda: synth.make_atrium(columns=10, spheres=60, sphere_subdiv=5),
                    7680, 4320, RendererConfig(cull_enabled=False, collect_stats=True, time_kernels=1, skybox=True)))
    if "field_xl" in names:
        out.append(("Field-XL (50 000 x icosphere-5), 7680x4320", lambda: synth.make_field(50000, 5, lods=1, with_env=True, prepared=True),
                    7680, 4320, RendererConfig(lod_enabled=False, cull_enabled=False, collect_stats=True, time_kernels=1)))
    return out


def timed(r, frames, geometry_only):
    for _ in range(3):
        r.render_visbuffer() if geometry_only else r.render()
    r.sync()
    r.stats()
    for _ in range(frames):
        r.render_visbuffer() if geometry_only else r.render()
    r.sync()
    return r.stats()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--scenes", default="field,atrium,field_xl")
    ap.add_argument("--worlds", default="2,4,8")
    ap.add_argument("--chunk", type=int, default=DEFAULT_CHUNK)
    ap.add_argument("--band", type=int, default=DEFAULT_BAND)
    ap.add_argument("--replicate", type=float, default=DEFAULT_REPLICATE_PIXELS,
                    help="interleaved split: draws whose triangles average more pixels than this are drawn by every rank on its own rows (0 = off)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "rank_alone.json"))
    args = ap.parse_args()
    results = []
    for name, make, w, h, conf in scenes(args.scenes.split(",")):
        scene = make()
        n = scene.meshlet_visibility_count
        r = Renderer(w, h, conf)
        r.set_scene(scene)
        st = timed(r, args.frames, False)
        whole = {"geometry_ms": sum(st["kernel_ms"][k] for k in GEOMETRY), "shading_ms": sum(st["kernel_ms"][k] for k in SHADING)}
        print(f"== {name}: one GPU geometry {whole['geometry_ms']:.3f} ms, shading {whole['shading_ms']:.3f} ms", flush=True)
        for world in (int(v) for v in args.worlds.split(",")):
            for split in ("contiguous", "interleaved"):
                rows = []
                for rank in range(world):
                    # geometry of the rank's share
                    if split == "interleaved":
                        r.set_partition(0, n)
                        r.set_partition_interleaved(args.chunk, rank, world)
                        r.set_shade_bands(args.band, rank, world)  # draws every rank takes are rasterised on the rank's rows
                        r.set_partition_replication(args.replicate)
                    else:
                        r.set_partition_interleaved(DEFAULT_CHUNK, 0, 1)
                        r.set_partition_replication(0.0)
                        r.set_partition(*instance_range(n, rank, world))
                        r.set_shade_rows(0, h)
                    g = timed(r, args.frames, True)
                    # shading of the rank's rows of the complete frame
                    r.set_partition_interleaved(DEFAULT_CHUNK, 0, 1)
                    r.set_partition_replication(0.0)
                    r.set_partition(0, n)
                    if split == "interleaved":
                        r.set_shade_bands(args.band, rank, world)
                    else:
                        r.set_shade_rows(*row_range(h, rank, world))
                    s = timed(r, args.frames, False)
                    rows.append({"rank": rank,
                                 "geometry_ms": sum(g["kernel_ms"][k] for k in GEOMETRY), "raster_ms": g["kernel_ms"]["raster"] + g["kernel_ms"]["big_raster"],
                                 "shading_ms": sum(s["kernel_ms"][k] for k in SHADING),
                                 "fragments": g["fragments"], "meshlets_visible": g["meshlets_visible"], "pixels_covered": s["pixels_covered"]})
                summary = {}
                for key in ("geometry_ms", "raster_ms", "shading_ms", "fragments", "pixels_covered"):
                    vals = [row[key] for row in rows]
                    mean = sum(vals) / len(vals)
                    summary[key] = {"max": max(vals), "mean": mean, "max_over_mean": max(vals) / mean if mean else None}
                frame = [row["geometry_ms"] + row["shading_ms"] for row in rows]
                summary["frame_ms"] = {"max": max(frame), "mean": sum(frame) / len(frame), "max_over_mean": max(frame) / (sum(frame) / len(frame))}
                results.append({"scene": name, "world": world, "split": split, "chunk": args.chunk, "band": args.band, "replicate_pixels": args.replicate if split == "interleaved" else 0.0,
                                "one_gpu": whole, "ranks": rows, "summary": summary})
                print(f"N={world} {split:11s} geometry max/mean {summary['geometry_ms']['max_over_mean']:.3f} (max {summary['geometry_ms']['max']:.3f} ms)  "
                      f"shading max/mean {summary['shading_ms']['max_over_mean']:.3f} (max {summary['shading_ms']['max']:.3f} ms)  "
                      f"frame max {summary['frame_ms']['max']:.3f} ms  fragments max/mean {summary['fragments']['max_over_mean']:.3f}  "
                      f"covered px max/mean {summary['pixels_covered']['max_over_mean']:.3f}", flush=True)
        r.close()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
