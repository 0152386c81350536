"""Entry point with the reference's surface (run.py:8-32): python run.py --config Y | -i video -p prompt [-n neg] [--multi_axis].

load_config -> seed_everything -> init_iclight -> Generator -> relit frames + config.yaml with the reference's metric keys
(total_time, sec_per_frame, max_memory_allocated, total_number_of_frames; generate.py:607-618).
Multi-GPU: torchrun --nproc-per-node N run.py ... shards frames over the ranks (tc_light_amd/parallel.py).
"""
import os
import random
import sys

import numpy as np
import torch

from tc_light_amd.config_utils import load_config, save_config
from tc_light_amd.dataparser import VideoDataParser, get_frame_ids
from tc_light_amd.generate import Generator
from tc_light_amd.model_utils import init_iclight
from tc_light_amd.parallel import Dist
from tc_light_amd.text import encode_prompt_pair


def seed_everything(seed):
    torch.manual_seed(seed); torch.cuda.manual_seed_all(seed); random.seed(seed); np.random.seed(seed)


def main(argv=None):
    config = load_config(argv)
    seed_everything(config.seed)
    if config.sd_version != "iclight":
        raise NotImplementedError("tc_light_amd implements the IC-Light path (sd_version: iclight); see invert.py")
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", 1), ("RANK", 0), ("LOCAL_RANK", 0)))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:
        torch.distributed.init_process_group("nccl", device_id=dev)
    pipe, scheduler, config.model_key = init_iclight(dev, config.get("models"), seed=config.seed)
    config.max_memory_allocated, config.total_time = 0, 0
    parser = VideoDataParser(config.data, dev)
    g = config.generation
    frame_ids = get_frame_ids(g.frame_range, parser.n_frames, g.frame_ids)
    config.total_number_of_frames = len(frame_ids)
    d = Dist(rank, world)
    lo, hi = d.range(len(frame_ids))
    frames_all = parser.load_video(frame_ids)
    flows = parser.load_flow_cache(frame_ids)
    if flows is None and config.post_opt.apply_opt:           # no cache: estimate with MemFlowNet (video_dataparser.py:63-110)
        from tc_light_amd.memflow import MemFlowEngine
        from tc_light_amd.model_utils import load_memflow_state
        flows = parser.estimate_and_cache_flow(frames_all, frame_ids, MemFlowEngine(load_memflow_state((config.get("models") or {}).get("memflow")), dev),
                                               save_flow=(rank == 0))
    cfg = dict(g); cfg.update(config.post_opt); cfg["seed"] = config.seed
    rmbg = background = None
    if g.get("background_cond"):                               # generate.py:68-69, 147-167
        from tc_light_amd.model_utils import load_rmbg_state
        from tc_light_amd.rmbg import RMBGEngine
        rmbg = RMBGEngine(load_rmbg_state((config.get("models") or {}).get("rmbg")), dev)
        background = parser.load_video(path=g.background_image_path)
    gen = Generator(pipe.unet, pipe.vae, cfg, dist=d, scheduler=scheduler, rmbg=rmbg)
    for name, prompt in g.prompt.items():
        conds = encode_prompt_pair(prompt, g.negative_prompt, dev, config.get("models", {}).get("text_encoder"))
        conds_t = encode_prompt_pair(g.prompt_t, g.negative_prompt_t, dev, config.get("models", {}).get("text_encoder"))
        masks = inv = k = past = None
        if config.post_opt.apply_opt:
            from tc_light_amd.flow_ids import soft_masks_and_ids
            fut, past = flows
            masks, inv, k = soft_masks_and_ids(frames_all, fut, past, alpha=parser.alpha)
        out, info = gen(frames_all[lo:hi], conds, conds_t, past, masks, inv, n_total=len(frame_ids), k=k, background=background)
        if rank == 0:
            config.total_time += info["total_time"]
            config.sec_per_frame = config.total_time / len(frame_ids)
            config.max_memory_allocated = max(config.max_memory_allocated, info["max_memory_allocated"])
            path = os.path.join(g.output_path, f"lmr_{g.local_merge_ratio}_gmr_{g.global_merge_ratio}_alpha_t_{g.alpha_t}_opt_{name}")
            save_config(config, path, gene=True)
            np.save(os.path.join(path, "output.npy"), (out.clamp(0, 1) * 255).byte().permute(0, 2, 3, 1).cpu().numpy())
            print(f"[INFO] {len(frame_ids)} frames in {info['total_time']:.1f} s ({1 / config.sec_per_frame:.3f} frames/s) -> {path}")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])
