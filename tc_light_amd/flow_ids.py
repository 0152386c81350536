"""Stage-2 input producer, host side (reference: VideoDataParser.load_data, utils/dataparsers/video_dataparser.py:43-61,134-156)."""
import torch

from .lib import lib, stream


def get_soft_mask_bwds(org_images, flows, past_flows, alpha=0.1, beta=1e2, diff_threshold=0.1):
    """utils/flow_utils.py:40-54 -> [N,1,H,W] f32."""
    n, _, h, w = org_images.shape
    img, fw, pf = (t.float().contiguous() for t in (org_images, flows, past_flows))
    mask = torch.empty(n, 1, h, w, device=img.device)
    lib().tcl_soft_mask_bwds(img, fw, pf, n, h, w, float(alpha), float(beta), float(img.max().item() * diff_threshold), mask, stream())
    return mask


def get_flowid(frames, flows, mask_bwds, rgb_threshold=0.01):
    """utils/flow_utils.py:56-93 -> (ids int32 [N,H,W], number of ids K)."""
    n, _, h, w = frames.shape
    fr, fw, mk = (t.float().contiguous() for t in (frames, flows, mask_bwds))
    ids = torch.empty(n, h, w, dtype=torch.int32, device=fr.device)
    last = torch.zeros(1, dtype=torch.int32, device=fr.device)
    ws = torch.empty(lib().tcl_flowid_workspace_bytes(h, w), dtype=torch.uint8, device=fr.device)
    lib().tcl_flowid(fr, fw, mk, n, h, w, float(fr.max().item() * rgb_threshold), ids, last, ws, stream())
    return ids, int(last.item())


def soft_masks_and_ids(frames, future_flows, past_flows, alpha=0.5):
    """load_data (video_dataparser.py:43-61): masks, then track ids; voxelization(voxel_size=None) is the identity on dense ids."""
    masks = get_soft_mask_bwds(frames, future_flows, past_flows, alpha=alpha)
    ids, k = get_flowid(frames, future_flows, masks)
    return masks, ids.reshape(-1), k
