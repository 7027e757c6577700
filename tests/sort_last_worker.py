"""One rank of a sort-last job whose ranks share one GPU (gloo carries device tensors; RCCL
refuses two ranks on one device).  Started by test_gpu_pipeline.py as
``python sort_last_worker.py RANK WORLD PORT OUT_DIR``; writes the colour images this rank ends
up holding, one per frame."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H = 320, 180
SUN = (0.3, 0.8, -0.4)
CAMERAS = [((0.0, 1.0, 30.0), (0.0, 0.0, 1.0)), ((-8.0, 6.0, 0.0), (-0.7, 0.3, 0.6)), ((3.0, 0.2, -10.0), (0.3, -0.5, -0.8))]


def main(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from renderer_amd import Renderer, RendererConfig, synth
    from renderer_amd.distributed import SortLastPipeline
    from renderer_amd.scene import Camera

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = synth.make_atrium()

    def make():
        r = Renderer(W, H, RendererConfig())
        r.set_scene(scene)
        return r

    pipe = SortLastPipeline(make, scene.meshlet_visibility_count, H, rank, world, depth=2)
    frames = [pipe.step(Camera(*c), SUN) for c in CAMERAS[:2]]
    torch.cuda.synchronize()
    images = [f.color.cpu().numpy().copy() for f in frames]
    last = pipe.step(Camera(*CAMERAS[2]), SUN)  # reuses the first slot
    torch.cuda.synchronize()
    images.append(last.color.cpu().numpy().copy())
    for k, img in enumerate(images):
        np.save(os.path.join(out_dir, f"color{rank}_{k}.npy"), img)
    pipe.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
