"""Round 5, second pass: the memory-level-parallelism rewrites of the HBM-bound kernels (k_gn_stats / k_gn_apply / k_layernorm / k_tome_normalize: several
16-byte loads in flight per thread instead of one) must not change a bit.  Loads TWO builds of the library (argv[1] = base, argv[2] = new), runs both on
the same seeded inputs -- shapes of the UNet levels, the VAE, ragged row counts, concat inputs -- compares every output byte, and times both.
    python tools/ab/cmp_elem_libs.py tc_light_amd/libtclight_hip_base.so tc_light_amd/libtclight_hip.so"""
import ctypes
import sys
import time

import torch

sys.path.insert(0, ".")
from tc_light_amd import lib as libmod   # noqa: E402

H = torch.float16


def load(path):
    libmod.LIB_PATH = path
    return libmod._Lib()


def st():
    return torch.cuda.current_stream().cuda_stream


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    A, B = load(sys.argv[1]), load(sys.argv[2])
    g = torch.Generator(device="cuda").manual_seed(5)
    bad = 0
    print("== GroupNorm (B, HW, C1, C2, silu, raw): us base -> new, GB/s new")
    for Bn, HW, C1, C2, silu, raw in [(50, 14400, 320, 0, 1, 0), (50, 14400, 320, 320, 1, 1), (50, 14400, 640, 320, 1, 1), (50, 3600, 640, 0, 1, 0),
                                       (50, 3600, 1280, 640, 1, 1), (50, 900, 1280, 0, 1, 0), (50, 900, 1280, 1280, 1, 1), (50, 225, 1280, 0, 0, 0),
                                       (3, 690, 320, 0, 0, 0), (3, 691, 1280, 640, 1, 0), (2, 921600, 128, 0, 1, 0), (2, 230400, 256, 0, 1, 0),
                                       (2, 14400, 512, 0, 0, 0), (1, 7, 320, 0, 1, 0)]:
        C = C1 + C2
        x1 = (torch.randn(Bn, HW, C1, device="cuda", generator=g) * 2 + 0.5).to(H)
        x2 = torch.randn(Bn, HW, C2, device="cuda", generator=g).to(H) if C2 else None
        ga, be = torch.randn(C, device="cuda", generator=g).to(H), torch.randn(C, device="cuda", generator=g).to(H)
        outs, us = [], []
        for L in (A, B):
            ws = torch.zeros(int(L.tcl_groupnorm_workspace_bytes(Bn, C)), dtype=torch.uint8, device="cuda")
            y = torch.zeros(Bn, HW, C, device="cuda", dtype=H)
            yr = torch.zeros(Bn, HW, C, device="cuda", dtype=H) if raw else None
            def run():
                if raw:
                    L.tcl_groupnorm_concat_f16(x1, C1, x2 if C2 else 0, C2, ga, be, y, yr, Bn, HW, 32, 1e-5, silu, ws, st())
                else:
                    L.tcl_groupnorm_f16(x1, C1, x2 if C2 else 0, C2, ga, be, y, Bn, HW, 32, 1e-5, silu, ws, st())
            us.append(timed(run))
            outs.append((y.clone(), yr.clone() if raw else None))
        same = torch.equal(outs[0][0], outs[1][0]) and (not raw or torch.equal(outs[0][1], outs[1][1]))
        bad += not same
        nbytes = Bn * HW * C * 2 * (3 + raw)
        print(f"  {(Bn, HW, C1, C2, silu, raw)}: {us[0]:8.1f} -> {us[1]:8.1f} us  {nbytes / us[1] * 1e-3:7.0f} GB/s  {'same bits' if same else 'DIFFERENT'}")
    print("== LayerNorm / LayerNorm+metric / tome_normalize (rows, C): us base -> new, GB/s new")
    for rows, C in [(1728000, 320), (432000, 640), (108000, 1280), (999, 320), (1001, 640), (13, 1280), (95040, 320), (7, 2048)]:
        x = (torch.randn(rows, C, device="cuda", generator=g) * 3).to(H)
        ga, be = torch.randn(C, device="cuda", generator=g).to(H), torch.randn(C, device="cuda", generator=g).to(H)
        res = []
        for L in (A, B):
            y, y2, m2, m = (torch.zeros_like(x) for _ in range(4))
            t1 = timed(lambda: L.tcl_layernorm_f16(x, ga, be, y, rows, C, 1e-5, st()))
            t2 = timed(lambda: L.tcl_layernorm_metric_f16(x, ga, be, y2, m2, rows, C, 1e-5, st()))
            t3 = timed(lambda: L.tcl_tome_normalize_f16(y, m, rows, C, st()))
            res.append((y, y2, m2, m, t1, t2, t3))
        same = all(torch.equal(res[0][i], res[1][i]) for i in range(4)) and torch.equal(res[1][2], res[1][3]) and torch.equal(res[1][0], res[1][1])
        bad += not same
        nb = rows * C * 2
        print(f"  {(rows, C)}: ln {res[0][4]:7.1f} -> {res[1][4]:7.1f} ({2 * nb / res[1][4] * 1e-3:5.0f} GB/s)  ln+metric {res[0][5]:7.1f} -> {res[1][5]:7.1f} "
              f"({3 * nb / res[1][5] * 1e-3:5.0f} GB/s)  normalize {res[0][6]:7.1f} -> {res[1][6]:7.1f} ({2 * nb / res[1][6] * 1e-3:5.0f} GB/s)  "
              f"{'same bits' if same else 'DIFFERENT'}")
    print("ALL SAME BITS" if not bad else f"{bad} CASES DIFFER")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
