#!/bin/bash
# round 6, GPU run 16: the plain default bench twice more (box-to-box / run-to-run spread of the headline on the final code)
O=gpurun_out/profiles_r6; mkdir -p $O
for i in b c; do timeout 1500 python bench.py --no_cpu_baseline > $O/bench_final_run_$i.json 2> /dev/null; python -c "
import json
r=json.loads([l for l in open('$O/bench_final_run_$i.json') if l.startswith('{')][-1]); print('run $i', round(r['value'],4), r['phase_seconds'], 'frac', round(r['roofline']['frac'],4), 'alone', round(r['roofline']['alone']['achieved']))"; done
