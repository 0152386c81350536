"""VideoDataParser (reference: utils/dataparsers/video_dataparser.py:12-156) -- frames, cached optical flow, soft masks, track ids.

Frames: .npy / .pt tensors [N,3,H,W] or [N,H,W,3] (uint8 or float), a directory of such per-frame files, or a video container when
torchvision.io / cv2 is importable (neither is in the target image).  Flow: the reference's on-disk cache format
`<video>_{past,future}_flow_memflow/%04d.pt` (one [1,2,H,W] tensor per frame, used when the file count matches, :112-132).  Flow
ESTIMATION: `estimate_and_cache_flow` runs the MemFlowNet engine (tc_light_amd/memflow.py) when no cache exists and writes the same files.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F


def _to_nchw01(a):
    t = torch.as_tensor(np.asarray(a)) if not isinstance(a, torch.Tensor) else a
    if t.dim() == 4 and t.shape[-1] == 3:
        t = t.permute(0, 3, 1, 2)
    t = t.float()
    return t / 255.0 if t.max() > 1.5 else t


def process_frames(frames, h, w):
    """utils/VidToMe/utils.py:147-179: resize so the short side covers, then centre-crop to (h, w)."""
    fh, fw = frames.shape[-2:]
    s = max(h / fh, w / fw)
    nh, nw = max(h, round(fh * s)), max(w, round(fw * s))
    frames = F.interpolate(frames, size=(nh, nw), mode="bilinear", antialias=True, align_corners=False)
    t, l = (nh - h) // 2, (nw - w) // 2
    return frames[..., t:t + h, l:l + w]


class VideoDataParser:
    def __init__(self, data_config, device):
        self.rgb_path = data_config.get("rgb_path")
        self.h, self.w = int(data_config["height"]), int(data_config["width"])
        self.fps = data_config.get("fps", 25)
        self.alpha = data_config.get("alpha", 0.5)
        self.device = device
        self.unq_inv = None
        self._all = None

    def _read(self, path):
        if os.path.isdir(path):
            fs = sorted(f for f in os.listdir(path) if f.endswith((".npy", ".pt")))
            return torch.cat([_to_nchw01(np.load(os.path.join(path, f)) if f.endswith(".npy") else torch.load(os.path.join(path, f)))
                              .reshape(-1, *_to_nchw01(np.load(os.path.join(path, f)) if f.endswith(".npy") else torch.load(os.path.join(path, f))).shape[-3:]) for f in fs])
        if path.endswith(".npy"):
            return _to_nchw01(np.load(path))
        if path.endswith(".pt"):
            return _to_nchw01(torch.load(path))
        try:
            import torchvision.io as tvio
            return _to_nchw01(tvio.read_video(path, pts_unit="sec", output_format="TCHW")[0])
        except ImportError:
            pass
        try:
            import cv2
            cap, out = cv2.VideoCapture(path), []
            while True:
                ok, fr = cap.read()
                if not ok:
                    break
                out.append(torch.from_numpy(fr[..., ::-1].copy()))
            return _to_nchw01(torch.stack(out))
        except ImportError:
            raise RuntimeError(f"no video decoder (torchvision.io / cv2) in this environment: convert {path} to a [N,H,W,3] .npy first")

    @property
    def n_frames(self):
        if self._all is None:
            self._all = self._read(self.rgb_path)
        return self._all.shape[0]

    def load_video(self, frame_ids=None, path=None):
        fr = self._read(path) if path is not None else (self._all if self._all is not None else self._read(self.rgb_path))
        if path is None:
            self._all = fr
        if frame_ids is not None:
            fr = fr[list(frame_ids)]
        return process_frames(fr, self.h, self.w).to(self.device)

    def _flow_dir(self, kind):
        """create_folder (video_dataparser.py:126-131): <video>_<name> beside a file, <dir>/<name> inside a frame directory."""
        name = f"{kind}_flow_memflow"
        if os.path.isdir(self.rgb_path):
            return os.path.join(self.rgb_path, name)
        return os.path.splitext(self.rgb_path)[0] + "_" + name

    def load_flow_cache(self, frame_ids):
        """-> (future_flows, past_flows) [N,2,H,W] from the reference's .pt cache (files named by frame id, used when the file count matches
        the number of frames, video_dataparser.py:112-116), or None when absent."""
        out = []
        for kind in ("future", "past"):
            d = self._flow_dir(kind)
            if not os.path.isdir(d) or len(os.listdir(d)) != len(frame_ids):
                return None
            out.append(torch.cat([torch.load(os.path.join(d, f"{fid:04d}.pt")).reshape(1, 2, self.h, self.w) for fid in frame_ids]))
        return out[0].to(self.device), out[1].to(self.device)

    def estimate_and_cache_flow(self, frames, frame_ids, engine, save_flow=True):
        """load_flow for flow_model 'memflow' (video_dataparser.py:63-110): frames [N,3,h,w] in [0,1] (already processed to the working size)
        -> (future_flows, past_flows) [N,2,h,w]; saved per frame as [1,2,h,w] tensors under <video>_{future,past}_flow_memflow/%04d.pt."""
        from .memflow import estimate_flows
        fut, past = estimate_flows(engine, frames)
        if save_flow and self.rgb_path:
            for kind, fl in (("future", fut), ("past", past)):
                d = self._flow_dir(kind)
                os.makedirs(d, exist_ok=True)
                for i, fid in enumerate(frame_ids):
                    torch.save(fl[i:i + 1].cpu(), os.path.join(d, f"{fid:04d}.pt"))
        return fut, past


def get_frame_ids(frame_range, n_frames, frame_ids=None):
    """utils/VidToMe/utils.py:330-346."""
    if frame_ids is not None:
        return list(frame_ids)
    start, end, step = frame_range
    end = n_frames if end is None or end < 0 else min(end, n_frames)
    return list(range(start, end, step))
