"""Fit the GEMM configuration cost model to a sweep produced by tools/micro/tune_gemm.py and report the
regret of model-chosen configurations against the per-shape best.  usage: fit_gemm_model.py gpurun_out/tune_gemm.json"""
import json, math, sys
import numpy as np
from scipy.optimize import least_squares

CFG = {1: (128, 128, 3), 2: (64, 128, 4), 3: (128, 64, 4), 4: (64, 64, 4), 11: (256, 128, 2), 5: (256, 320, 1), 6: (128, 320, 1), 7: (256, 256, 1), 8: (128, 256, 1)}
ORDER = [1, 2, 3, 4, 11, 5, 6, 7, 8]
KB = 32


def shape(key):
    if key[0] == 'g':
        M, N, K, act = map(int, key[1:5]); return M, N, K, act
    B, H, W, ci, co, stride = map(int, key[1:7]); up = key[7]
    if up != 'None':
        hu, wu = eval(up)
    else:
        hu, wu = H, W
    ho, wo = (hu - 1) // stride + 1, (wu - 1) // stride + 1
    return B * ho * wo, co, 9 * ci, 0


def predict(p, cfg, sp, M, N, K):
    """p: dict cfg -> (tstep_us, alpha, t0_us); global fin0, finbw"""
    bm, bn, occ = CFG[cfg]
    tstep, alpha, t0 = p[cfg]
    tiles = math.ceil(M / bm) * math.ceil(N / bn)
    nk = K // KB
    sp = max(1, min(sp, nk // 4)) if sp > 1 else 1
    nkp = math.ceil(nk / sp); sp = math.ceil(nk / nkp)
    blocks = tiles * sp
    S = 256 * occ
    if blocks >= S:
        t = math.ceil(blocks / S) * nkp * tstep
    else:
        b = math.ceil(blocks / 256)
        t = nkp * tstep * max(b / occ, alpha)
    t += t0
    if sp > 1:
        t += p['fin0'] + M * N * (4.0 * sp + 4) / p['finbw']
    return t


def unpack(x):
    p = {}
    for i, c in enumerate(ORDER):
        p[c] = (x[3 * i], x[3 * i + 1], x[3 * i + 2])
    p['fin0'] = x[-2]; p['finbw'] = x[-1] * 1e6
    return p


def main():
    d = json.load(open(sys.argv[1]))
    rows = []
    for e in d:
        M, N, K, act = shape(e['key'])
        if act == 2:
            continue
        for k, v in e['alls'].items():
            c, s = map(int, k.split('_'))
            rows.append((c, s, M, N, K, v))
    x0 = []
    for c in ORDER:
        x0 += [1.0, 0.5, 5.0]
    x0 += [3.0, 3.0]

    def resid(x):
        p = unpack(x)
        return [math.log(predict(p, c, s, M, N, K) / v) for c, s, M, N, K, v in rows]
    lo = [0.05, 0.05, 0.0] * len(ORDER) + [0.0, 0.2]
    hi = [10.0, 1.0, 40.0] * len(ORDER) + [30.0, 20.0]
    r = least_squares(resid, x0, bounds=(lo, hi), loss='soft_l1', f_scale=0.2)
    p = unpack(r.x)
    print("rms log err", float(np.sqrt(np.mean(np.square(r.fun)))))
    for c in ORDER:
        print(f"cfg {c:2d} {CFG[c]}: tstep {p[c][0]:.3f} us  alpha {p[c][1]:.2f}  t0 {p[c][2]:.1f} us")
    print("fin0", p['fin0'], "finbw (B/us)", p['finbw'])
    # regret
    tot_best = tot_model = tot_def = 0.0
    worst = []
    for e in d:
        M, N, K, act = shape(e['key'])
        if act == 2 or not e['alls']:
            tot_best += e['n'] * e['best'][0]; tot_model += e['n'] * e['default']; tot_def += e['n'] * e['default']; continue
        cands = [tuple(map(int, k.split('_'))) for k in e['alls']]
        ch = min(cands, key=lambda cs: predict(p, cs[0], cs[1], M, N, K))
        tm = e['alls'][f"{ch[0]}_{ch[1]}"]
        tot_best += e['n'] * e['best'][0]; tot_model += e['n'] * tm; tot_def += e['n'] * e['default']
        worst.append((e['n'] * (tm - e['best'][0]), e['key'], ch, tm, e['best']))
    print(f"default {tot_def/1e3:.1f} ms  model {tot_model/1e3:.1f} ms  best {tot_best/1e3:.1f} ms (both passes)")
    for w in sorted(worst, reverse=True)[:12]:
        print("  regret %.0f us: %s chose %s %.1f best %s" % (w[0], w[1], w[2], w[3], w[4]))
    print("constants for gemm.hip:")
    print("{" + ", ".join(f"{{{c}, {p[c][0]:.3f}f, {p[c][1]:.2f}f, {p[c][2]:.1f}f}}" for c in ORDER) + "}", f"fin0 {p['fin0']:.2f} finbw {p['finbw']:.0f}")


main()
