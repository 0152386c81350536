"""VideoDataParser + video I/O (reference: utils/dataparsers/video_dataparser.py:12-156, utils/VidToMe/utils.py:83-179).

Frames come from (in the order the reference's `load_video` tests them, utils/VidToMe/utils.py:118-139):
  * `.mp4` / `.avi` containers -- decoded with torchvision.io or cv2 when one of them is importable (neither is in the target image: the
    error then says how to convert; `.npy` / `.pt` tensors [N,3,H,W] or [N,H,W,3] are accepted as the decoder-free stand-in);
  * `.gif` -- PIL `ImageSequence`;
  * a directory of `.jpg/.png/.JPG/.PNG` frames (FRAME_EXT, utils.py:16) read with PIL.  The reference's `load_image` maps such frames to
    `x*255/127 - 1` and `VideoDataParser.load_video` maps them back with `(x+1)*127/255` (utils.py:76-80, video_dataparser.py:37-38);
    the round trip is reproduced literally so the f32 rounding is the reference's.
Flow: the reference's on-disk cache format `<video>_{past,future}_flow_memflow/%04d.pt` (one [1,2,H,W] tensor per frame, used when the
file count matches, :112-132).  `estimate_and_cache_flow` runs the MemFlowNet engine (tc_light_amd/memflow.py) when no cache exists.
Output: `save_video` / `save_frames` (utils.py:147-187): `output{post_fix}.mp4` through torchvision/cv2 when an encoder exists, else the
frames as PNGs under `frames{post_fix}/` plus `output{post_fix}.npy` (uint8 [N,H,W,3]) so `evaluate.py`-style consumers still find data.
"""
import os
from glob import glob

import numpy as np
import torch
import torch.nn.functional as F

FRAME_EXT = [".jpg", ".png", ".JPG", ".PNG"]            # utils/VidToMe/utils.py:16


def _to_nchw01(a):
    """array / tensor [N,3,H,W] or [N,H,W,3] -> float32 NCHW in [0,1]; integer dtypes are 8-bit pixel values (scaled by 1/255), floating
    dtypes are taken as already in [0,1] (the scaling follows the dtype, never the value range: a dark uint8 clip stays a uint8 clip)."""
    t = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))
    if t.dim() == 3:
        t = t[None]
    if t.dim() == 4 and t.shape[-1] == 3 and t.shape[1] != 3:
        t = t.permute(0, 3, 1, 2)
    if t.dtype.is_floating_point:
        return t.float()
    return t.float() / 255.0


def _torch_load(path):
    return torch.load(path, map_location="cpu", weights_only=True)


def load_image(image_path):
    """utils/VidToMe/utils.py:76-80: RGB frame -> [1,3,H,W] in [-1, 1.008]."""
    from PIL import Image
    img = np.asarray(Image.open(image_path).convert("RGB"), dtype=np.uint8)
    t = torch.from_numpy(img.copy()).permute(2, 0, 1).float().div(255.0)            # T.ToTensor()
    return (t * 255.0 / 127.0 - 1.0).unsqueeze(0)


def glob_frame_paths(video_path):
    paths = []
    for ext in FRAME_EXT:
        paths += glob(os.path.join(video_path, f"*{ext}"))
    return sorted(paths)


def process_frames(frames, h, w):
    """utils/VidToMe/utils.py:83-105: T.Resize so the short side covers (bilinear, antialiased on tensors), then centre-crop to (h, w)."""
    fh, fw = frames.shape[-2:]
    s = max(w / fw, h / fh)
    nh, nw = int(round(fh * s)), int(round(fw * s))
    if (nh, nw) != (fh, fw):
        frames = F.interpolate(frames, size=(nh, nw), mode="bilinear", antialias=True, align_corners=False)
    t, l = int(round((nh - h) / 2.0)), int(round((nw - w) / 2.0))                  # torchvision center_crop
    return frames[..., t:t + h, l:l + w]


def read_frames(path):
    """-> float32 [N,3,H,W]; containers / gifs / tensors in [0,1], frame directories in the reference's [-1,1] convention."""
    if os.path.isdir(path):
        fps = glob_frame_paths(path)
        if fps:
            return torch.cat([load_image(p) for p in fps])
        fs = sorted(f for f in os.listdir(path) if f.endswith((".npy", ".pt")))
        if not fs:
            raise FileNotFoundError(f"no frames ({'/'.join(FRAME_EXT)} or .npy/.pt) under {path}")
        return torch.cat([_to_nchw01(np.load(os.path.join(path, f)) if f.endswith(".npy") else _torch_load(os.path.join(path, f))) for f in fs])
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    if path.endswith(".npy"):
        return _to_nchw01(np.load(path))
    if path.endswith(".pt"):
        return _to_nchw01(_torch_load(path))
    if path.endswith(".gif"):
        from PIL import Image, ImageSequence
        return torch.stack([torch.from_numpy(np.asarray(fr.convert("RGB")).copy()).permute(2, 0, 1).float() / 255.0
                            for fr in ImageSequence.Iterator(Image.open(path))])
    if path.lower().endswith(".avi"):                        # Motion-JPEG AVI (what save_video writes here): own reader, no codec library
        fr = read_mjpeg_avi(path)
        if fr is not None:
            return _to_nchw01(fr)
    try:
        import torchvision.io as tvio
        return tvio.read_video(path, pts_unit="sec", output_format="TCHW")[0].float() / 255.0
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
        raise RuntimeError(f"no video decoder (torchvision.io / cv2) in this environment: convert {path} to a frame directory "
                           f"({'/'.join(FRAME_EXT)}) or a [N,H,W,3] uint8 .npy first")


# ---- Motion-JPEG in an AVI container, written / read with nothing but PIL: the reference's save_video needs an H.264 encoder behind torchvision
# (utils/VidToMe/utils.py:147-166) and this image has none; a relight must still come out as ONE playable file.  RIFF layout: hdrl (avih + one
# 'vids' / 'MJPG' stream), movi ('00dc' chunks, one JPEG per frame), idx1.
def write_mjpeg_avi(dst, frames_u8, fps=30, quality=95):
    """frames_u8: uint8 [N,H,W,3] (numpy or tensor)."""
    import io
    import struct
    from PIL import Image
    fr = frames_u8.numpy() if isinstance(frames_u8, torch.Tensor) else np.asarray(frames_u8)
    n, h, w = fr.shape[:3]
    jpgs = []
    for f in fr:
        b = io.BytesIO()
        Image.fromarray(f).save(b, format="JPEG", quality=quality, subsampling=0)
        jpgs.append(b.getvalue())
    chunk = lambda cc, data: cc + struct.pack("<I", len(data)) + data + (b"\0" if len(data) & 1 else b"")
    lst = lambda kind, data: b"LIST" + struct.pack("<I", len(data) + 4) + kind + data
    # one time base for both headers: dwRate / dwScale = fps to 1/1000 (29.97 stays 29.97; NTSC rates become exactly 30000/1001 etc.)
    scale, rate = 1000, max(int(round(float(fps) * 1000)), 1)
    for num in (24000, 30000, 60000, 120000):
        if abs(float(fps) - num / 1001.0) < 5e-3:
            scale, rate = 1001, num
    us = int(round(1e6 * scale / rate))
    biggest = max(len(j) for j in jpgs)
    avih = struct.pack("<14I", us, biggest * max(int(round(fps)), 1), 0, 0x10, n, 0, 1, biggest, w, h, 0, 0, 0, 0)
    strh = b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIII4H", 0, 0, 0, 0, scale, rate, 0, n, biggest, 0xFFFFFFFF, 0, 0, 0, w, h)
    strf = struct.pack("<IiiHH4sIiiII", 40, w, h, 1, 24, b"MJPG", w * h * 3, 0, 0, 0, 0)
    hdrl = lst(b"hdrl", chunk(b"avih", avih) + lst(b"strl", chunk(b"strh", strh) + chunk(b"strf", strf)))
    movi, idx, off = b"", b"", 4
    for j in jpgs:
        c = chunk(b"00dc", j)
        idx += b"00dc" + struct.pack("<III", 0x10, off, len(j))
        movi += c
        off += len(c)
    body = b"AVI " + hdrl + lst(b"movi", movi) + chunk(b"idx1", idx)
    with open(dst, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def read_mjpeg_avi(path):
    """-> uint8 tensor [N,H,W,3], or None when the file is not a RIFF AVI whose video stream is (M)JPEG."""
    import io
    import struct
    from PIL import Image
    data = open(path, "rb").read()
    if data[:4] != b"RIFF" or data[8:12] != b"AVI ":
        return None
    out = []

    def walk(lo, hi):
        while lo + 8 <= hi:
            cc, sz = data[lo:lo + 4], struct.unpack("<I", data[lo + 4:lo + 8])[0]
            if cc == b"LIST":
                walk(lo + 12, lo + 8 + sz)
            elif cc[2:] in (b"dc", b"db") and sz > 2 and data[lo + 8:lo + 10] == b"\xff\xd8":
                out.append(torch.from_numpy(np.asarray(Image.open(io.BytesIO(data[lo + 8:lo + 8 + sz])).convert("RGB")).copy()))
            lo += 8 + sz + (sz & 1)
    walk(12, len(data))
    return torch.stack(out) if out else None


def save_frames(frames, path, ext="png", frame_ids=None):
    """utils/VidToMe/utils.py:182-187: frames [N,3,H,W] in [0,1] -> path/%04d.ext"""
    from PIL import Image
    os.makedirs(path, exist_ok=True)
    ids = list(frame_ids) if frame_ids is not None else list(range(len(frames)))
    u8 = (frames.clamp(0, 1) * 255).to(torch.uint8).permute(0, 2, 3, 1).cpu().numpy()
    for i, fr in zip(ids, u8):
        Image.fromarray(fr).save(os.path.join(path, "{:04}.{}".format(i, ext)))


def save_video(frames, path, frame_ids=None, save_frame=False, gif=True, post_fix="", fps=30):
    """utils/VidToMe/utils.py:147-179.  Returns the path of what was written.  Encoders are probed in the reference's order (torchvision
    write_video libx264 crf 23 / medium for .mp4, imageio for .gif, PIL as the gif fallback); with none importable the uint8 frames are
    written as `output{post_fix}.npy` and as PNGs so nothing is lost."""
    os.makedirs(path, exist_ok=True)
    ids = list(frame_ids) if frame_ids is not None else list(range(len(frames)))
    frames = frames[ids]
    proc = (frames.permute(0, 2, 3, 1) * 255).to(torch.uint8).cpu()
    out = None

    def _mp4_torchvision(dst):
        from torchvision.io import write_video
        write_video(dst, proc, fps=fps, video_codec="libx264", options={"crf": "23", "preset": "medium"})

    def _mp4_cv2(dst):
        import cv2
        wr = cv2.VideoWriter(dst, cv2.VideoWriter_fourcc(*"mp4v"), fps, (proc.shape[2], proc.shape[1]))
        if not wr.isOpened():
            raise RuntimeError("cv2.VideoWriter could not open the output")
        for fr in proc.numpy():
            wr.write(fr[..., ::-1])
        wr.release()

    def _gif_imageio(dst):
        import imageio
        imageio.mimsave(dst, [f.numpy() for f in proc], "GIF", fps=fps, loop=0)

    def _gif_pil(dst):
        from PIL import Image
        ims = [Image.fromarray(f.numpy()) for f in proc]
        ims[0].save(dst, save_all=True, append_images=ims[1:], duration=int(1000 / max(fps, 1)), loop=0)

    # an encoder that is missing OR fails (a torchvision / PyAV build without libx264, a cv2 writer error) must not lose a finished
    # relight: every failure falls through to the next writer and finally to .npy + PNG frames
    dst = os.path.join(path, f"output{post_fix}.gif" if gif else f"output{post_fix}.mp4")
    for writer in ((_gif_imageio, _gif_pil) if gif else (_mp4_torchvision, _mp4_cv2)):
        try:
            writer(dst)
            out = dst
            break
        except Exception as e:            # noqa: BLE001 - any encoder failure
            if not isinstance(e, ImportError):
                print(f"[WARN] {writer.__name__[1:]} failed ({type(e).__name__}: {e}); trying the next writer")
            if os.path.exists(dst):
                os.remove(dst)
    if out is None and not gif:                        # no H.264 encoder: one playable file all the same (Motion-JPEG AVI through PIL)
        try:
            out = os.path.join(path, f"output{post_fix}.avi")
            write_mjpeg_avi(out, proc, fps=fps)
            print(f"[INFO] no H.264 encoder (torchvision / cv2) in this environment: wrote Motion-JPEG {out} instead of output{post_fix}.mp4")
        except Exception as e:            # noqa: BLE001
            print(f"[WARN] mjpeg_avi failed ({type(e).__name__}: {e})")
            if os.path.exists(out):
                os.remove(out)
            out = None
    if out is None:                                    # no encoder at all: keep the data, say so
        out = os.path.join(path, f"output{post_fix}.npy")
        np.save(out, proc.numpy())
        save_frames(frames, os.path.join(path, f"frames{post_fix}"), frame_ids=ids)
        print(f"[INFO] no working video encoder (torchvision / cv2 / imageio / PIL): wrote {out} and PNG frames instead of output{post_fix}.mp4")
    else:
        print(f"[INFO] save video to {out}")
    if save_frame:
        save_frames(frames, os.path.join(path, f"frames{post_fix}"), frame_ids=ids)
    return out


def save_loss_curve(loss_list, output_path, title="Loss Curve"):
    """utils/VidToMe/utils.py:189-197 (matplotlib when importable; the raw values are always kept beside the plot)."""
    vals = [float(v) for v in (loss_list.detach().cpu().tolist() if isinstance(loss_list, torch.Tensor) else loss_list)]
    np.save(os.path.join(output_path, f"{title}.npy"), np.asarray(vals, np.float32))
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        plt.plot(vals); plt.xlabel("Iteration"); plt.ylabel("Loss"); plt.title(title); plt.grid()
        plt.savefig(os.path.join(output_path, f"{title}.png")); plt.close()
    except Exception:
        pass


class VideoDataParser:
    def __init__(self, data_config, device):
        self.rgb_path = data_config.get("rgb_path")
        self.h, self.w = int(data_config["height"]), int(data_config["width"])
        self.fps = data_config.get("fps", 30)                    # video_dataparser.py:15
        self.alpha = data_config.get("alpha", 0.5)
        self.flow_model = data_config.get("flow_model", "memflow")
        self.device = device
        self.unq_inv = None
        self._all = None

    @property
    def n_frames(self):
        if self._all is None:
            self._all = read_frames(self.rgb_path)
        return self._all.shape[0]

    def load_video(self, frame_ids=None, path=None):
        """video_dataparser.py:34-41: frames [N,3,h,w] f32 in [0,1] on the device, resized + centre-cropped to the working size."""
        fr = read_frames(path) if path is not None else (self._all if self._all is not None else read_frames(self.rgb_path))
        if path is None:
            self._all = fr
        if frame_ids is not None:
            fr = fr[list(frame_ids)]
        fr = process_frames(fr, self.h, self.w)
        if fr.min() < 0:                                         # frame directories arrive in [-1, 1] (load_image)
            fr = (fr + 1.0) * 127.0 / 255.0
        return fr.to(self.device)

    def _flow_dir(self, kind):
        """create_folder (video_dataparser.py:126-131): <video>_<name> beside a file, <dir>/<name> inside a frame directory."""
        name = f"{kind}_flow_{self.flow_model}"
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
            out.append(torch.cat([_torch_load(os.path.join(d, f"{fid:04d}.pt")).reshape(1, 2, self.h, self.w) for fid in frame_ids]))
        return out[0].to(self.device), out[1].to(self.device)

    def estimate_and_cache_flow(self, frames, frame_ids, engine, save_flow=True):
        """load_flow for flow_model 'memflow' (video_dataparser.py:63-110): frames [N,3,h,w] in [0,1] (already processed to the working size)
        -> (future_flows, past_flows) [N,2,h,w]; saved per frame as [1,2,h,w] tensors under <video>_{future,past}_flow_memflow/%04d.pt
        (each file is written under a temporary name and renamed, so a reader never sees a partial file)."""
        from .memflow import estimate_flows
        fut, past = estimate_flows(engine, frames)
        if save_flow and self.rgb_path:
            for kind, fl in (("future", fut), ("past", past)):
                d = self._flow_dir(kind)
                os.makedirs(d, exist_ok=True)
                for i, fid in enumerate(frame_ids):
                    dst = os.path.join(d, f"{fid:04d}.pt")
                    torch.save(fl[i:i + 1].cpu(), dst + ".tmp")
                    os.replace(dst + ".tmp", dst)
        return fut, past


def get_frame_ids(frame_range, n_frames, frame_ids=None):
    """utils/VidToMe/utils.py:330-346: [start, end, step] with end == -1 (or beyond the clip) meaning the clip length; sorted ids."""
    if frame_ids is None:
        fr = list(frame_range)
        if len(fr) > 1 and (fr[1] is None or fr[1] == -1):
            fr[1] = n_frames
        if len(fr) > 1 and fr[1] > n_frames:
            print(f"[WARNING] end frame {fr[1]} has been adjusted to number of frames {n_frames}.")
            fr[1] = n_frames
        frame_ids = list(range(*fr))
    frame_ids = sorted(frame_ids)
    shown = frame_ids if len(frame_ids) <= 4 else frame_ids[:2] + ["..."] + frame_ids[-2:]
    print("[INFO] frame indexes: ", " ".join(str(i) for i in shown))
    return frame_ids
