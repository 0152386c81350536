#!/bin/bash
# round 5: head_dim 80 attention, 4 waves per block (3 blocks per CU) vs 8 waves per block (TCL_FLASH80=8: half the LDS-DMA bytes per query)
for r in 1 2; do for v in 4 8; do echo "== TCL_FLASH80=$v (round $r)"; TCL_FLASH80=$v python tools/micro/bench_attn.py 2>&1 | grep "d=80"; TCL_FLASH80=$v python - <<'PY'
import sys, os; sys.path.insert(0, os.getcwd())
import torch
from tc_light_amd.lib import lib
L = lib(); H = torch.float16
st = lambda: torch.cuda.current_stream().cuda_stream
for B, T in ((2, 11880), (2, 7920), (4, 11880)):
    d, Hh = 80, 8; C = Hh * d
    q, k, v = (torch.randn(B, T, C, device="cuda").to(H) for _ in range(3)); o = torch.empty_like(q)
    wq = torch.empty(L.tcl_attention_q_bytes(B, Hh, T, d), dtype=torch.uint8, device="cuda"); wkv = torch.empty(L.tcl_attention_kv_bytes(B, Hh, T, d), dtype=torch.uint8, device="cuda")
    f = lambda: L.tcl_attention_f16(q, C, T * C, k, C, T * C, v, C, T * C, o, C, T * C, B, Hh, T, T, d, d ** -0.5, 1, 1, wq, wkv, st())
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(200): f()
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 200
    print(f"  sustained d=80 B={B} T={T}: {ms*1e3:8.1f} us {4.0*B*Hh*T*T*d/ms/1e9:7.1f} TF/s (incl. pack)")
PY
done; done
TCL_FLASH80=8 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -x -q -m gpu -k "attention" 2>&1 | tail -3
