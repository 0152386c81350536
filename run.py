"""Entry point with the reference's surface (run.py:8-32): python run.py --config Y | -i video -p prompt [-n neg] [--multi_axis].

load_config -> seed_everything -> init_iclight -> Generator -> output.mp4 (frames / .npy when the image has no encoder), output_gt, loss
curves and config.yaml with the reference's metric keys (total_time, sec_per_frame, max_memory_allocated, total_number_of_frames;
generate.py:607-630).
Multi-GPU: torchrun --nproc-per-node N run.py ... shards frames over the ranks (tc_light_amd/parallel.py).
"""
import os
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL between processes needs dmabuf IPC on this driver (set before HIP starts)
import random
import sys

import numpy as np
import torch

from tc_light_amd.config_utils import load_config, save_config
from tc_light_amd.dataparser import VideoDataParser, get_frame_ids, save_loss_curve, save_video
from tc_light_amd.generate import Generator
from tc_light_amd.model_utils import allow_random, init_iclight
from tc_light_amd.parallel import Dist
from tc_light_amd.text import encode_prompt_pair


def seed_everything(seed):
    torch.manual_seed(seed); torch.cuda.manual_seed_all(seed); random.seed(seed); np.random.seed(seed)


def main(argv=None):
    config = load_config(argv)
    seed_everything(config.seed)
    if config.sd_version != "iclight":
        raise NotImplementedError("tc_light_amd implements the IC-Light path (sd_version: iclight); see invert.py")
    if config.generation.prompt is None:          # checked before models, video and flow estimation are paid for
        raise NotImplementedError("generation.prompt is null: the Cosmos/Pixtral prompt up-sampler (generate.py:538-549) is outside this "
                                  "engine -- give a prompt (-p ... or generation.prompt)")
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", 1), ("RANK", 0), ("LOCAL_RANK", 0)))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", device_id=dev)
    models = config.get("models") or {}
    ok_random = allow_random(models)
    pipe, scheduler, config.model_key = init_iclight(dev, models, seed=config.seed)
    config.max_memory_allocated, config.total_time = 0, 0
    parser = VideoDataParser(config.data, dev)
    g = config.generation
    frame_ids = get_frame_ids(g.frame_range, parser.n_frames, g.frame_ids)
    config.total_number_of_frames = len(frame_ids)
    d = Dist(rank, world)
    lo, hi = d.range(len(frame_ids))
    frames_all = parser.load_video(frame_ids)
    flows = None
    if config.post_opt.apply_opt:
        # rank 0 owns the flow cache (read, or estimated with MemFlowNet and written: video_dataparser.py:63-110); the other ranks
        # receive the tensors.  A failure on rank 0 (e.g. missing MemFlow weights) is announced before the payload so that every
        # rank raises instead of waiting in the broadcast until the collective times out.
        err = None
        if rank == 0:
            try:
                flows = parser.load_flow_cache(frame_ids)
                if flows is None:
                    from tc_light_amd.memflow import MemFlowEngine
                    from tc_light_amd.model_utils import load_memflow_state
                    flows = parser.estimate_and_cache_flow(frames_all, frame_ids, MemFlowEngine(load_memflow_state(models.get("memflow"), allow=ok_random), dev))
            except Exception as e:            # noqa: BLE001 - re-raised below on every rank
                err = e
        if world > 1:
            ok = torch.tensor([0 if err is not None else 1], device=dev)
            torch.distributed.broadcast(ok, src=0)
            if int(ok.item()) == 0:
                torch.distributed.destroy_process_group()
                raise err if err is not None else RuntimeError("rank 0 failed to load / estimate the optical flow (see its traceback)")
            flows = flows if rank == 0 else tuple(torch.empty(len(frame_ids), 2, parser.h, parser.w, device=dev) for _ in range(2))
            for t in flows:
                torch.distributed.broadcast(t, src=0)
        elif err is not None:
            raise err
    cfg = dict(g); cfg.update(config.post_opt); cfg["seed"] = config.seed
    rmbg = background = None
    if g.get("background_cond"):                               # generate.py:68-69, 147-167
        from tc_light_amd.model_utils import load_rmbg_state
        from tc_light_amd.rmbg import RMBGEngine
        rmbg = RMBGEngine(load_rmbg_state(models.get("rmbg"), allow=ok_random), dev)
        background = parser.load_video(path=g.background_image_path)
        if background.shape[0] == len(frame_ids) and world > 1:     # a per-frame background video: this rank's block of it
            background = background[lo:hi]
    gen = Generator(pipe.unet, pipe.vae, cfg, dist=d, scheduler=scheduler, rmbg=rmbg)
    for name, prompt in g.prompt.items():
        conds = encode_prompt_pair(prompt, g.negative_prompt, dev, models.get("text_encoder"), allow_random=ok_random)
        conds_t = encode_prompt_pair(g.prompt_t, g.negative_prompt_t, dev, models.get("text_encoder"), allow_random=ok_random)
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
            # generate.py:613-630: save_name, config.yaml, output.mp4 (+ frames), output_gt.mp4, loss curves
            path = os.path.join(g.output_path, f"lmr_{g.local_merge_ratio}_gmr_{g.global_merge_ratio}_alpha_t_{g.alpha_t}_opt_{name}")
            save_config(config, path, gene=True)
            out = out.clamp(0, 1)
            np.save(os.path.join(path, "output.npy"), (out * 255).byte().permute(0, 2, 3, 1).cpu().numpy())    # first: survives any encoder failure
            if config.post_opt.apply_opt:
                save_loss_curve(info["losses_exposure"], path, "loss_exposure")
                if info["losses_unique"] is not None:
                    save_loss_curve(info["losses_unique"], path, "loss_unique_tensor")
            save_video(out, path, save_frame=bool(g.get("save_frame", False)), fps=parser.fps, gif=False)
            gt_path = os.path.join(path, "gt")
            if not os.path.exists(gt_path) or len(os.listdir(gt_path)) != len(frame_ids):
                save_video(frames_all, path, save_frame=False, post_fix="_gt", fps=parser.fps, gif=False)
            print(f"[INFO] {len(frame_ids)} frames in {info['total_time']:.1f} s ({1 / config.sec_per_frame:.3f} frames/s) -> {path}")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])
