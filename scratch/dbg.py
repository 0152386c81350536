import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from oracle import vidtome as OV
from tc_light_amd.vidtome import VidToMe
g = np.random.default_rng(3)
N, C = 345, 320
tome = VidToMe("cuda")
bank=None
for F, randf, coin in [(4, 2, 0.9), (3, 0, 0.7), (4, 1, 0.1), (1, -1, 0.3)]:
    base = g.standard_normal((1, N, C)).astype(np.float32)
    x = torch.from_numpy(base + 0.3 * g.standard_normal((2 * F, N, C)).astype(np.float32)).half()
    tome.draws = [(randf, coin)]
    tome.begin_forward(F, (15, 23))
    tome.trace=[]
    merged, unm, T = tome.compute_merge("blk", x.cuda(), F, N, C)
    r = OV.compute_merge(x.float(), F, bank, randf, coin, emulate_f16=True)
    um = unm.cpu().long() if unm is not None else torch.arange(F*N)
    d = (merged.cpu().float() != r["merged"]).any(-1)
    print(F, T, (um==r["unm"]).float().mean().item(), d.sum(1).tolist(), d[0].nonzero().flatten()[:10].tolist())
    tr = tome.trace[0]
    if 'gather' in tr and tr['gather'] is not None:
        print(' mrg1 eq', (tr['gather'].cpu().long()==r['gather']).float().mean().item())
    bank = tome.banks["blk"].cpu().float()
