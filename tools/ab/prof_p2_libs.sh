#!/bin/bash
# per-kernel times of stage 2 alone (300 x 1280 x 720; $1 = track reuse: 0.02 = the bench's codebook regime) for several library builds ($2.. = suffixes of
# tc_light_amd/libtclight_hip<suffix>.so; "" = the tree's), outputs digested: every build must print the same digest
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
reuse=${1:-0.02}; shift
for l in "$@"; do
  p=$PWD/tc_light_amd/libtclight_hip$l.so
  rm -rf /tmp/prof$l
  P2_DIGEST=1 P2_STAGES=2 TCL_LIB_PATH=$p timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof$l -o p2 --output-format csv -- python tools/micro/bench_p2.py 300 720 1280 48 $reuse 2>&1 | grep "^stage"
  f=$(find /tmp/prof$l -name "*kernel_stats.csv" | head -1)
  echo "== lib '$l'"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:12]:
    print(f'{r["Name"][:60]:60s} {int(r["Calls"]):7d} {float(r["TotalDurationNs"])/1e6:10.2f} ms  avg {float(r["AverageNs"])/1e3:9.2f} us')
PY
done
