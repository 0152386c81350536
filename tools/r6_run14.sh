#!/bin/bash
set -x
O=gpurun_out/profiles_r6; mkdir -p $O
timeout 900 python tools/micro/parity_budget.py 2 > $O/parity_budget.txt 2>&1; cat $O/parity_budget.txt | grep -v Warning
