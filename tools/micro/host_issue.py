"""Is the denoise loop bound by the host or by the GPU?  Runs bench.py's pass on 60 frames / 4 steps (extra argv is appended, e.g.
--height 360 --width 640) and prints how long the host took to ISSUE ddim_sample and when the GPU was done."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); os.chdir(sys.path[0])
sys.argv = ['bench.py', '--frames', '60', '--steps', '4', '--warmup', '1', '--no_cpu_baseline', '--no_extras', '--epochs', '4', '--epochs_exposure', '2'] + sys.argv[1:]
import torch
import bench
from tc_light_amd import generate
orig = generate.Generator.ddim_sample
def patched(self, *a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig(self, *a, **k)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"[ddim] host issue {t1 - t0:.2f} s, until GPU done {t2 - t0:.2f} s", flush=True)
    return r
generate.Generator.ddim_sample = patched
bench.main()
