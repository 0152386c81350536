#!/bin/bash
# GPU box (via gpurun): round-5 evidence -> gpurun_out/profiles_r5/
#   1. kernel-trace stats of the default bench (the metric's configuration), timed pass only
#   2. HBM-side bytes per head_dim-40 attention call in the pass + SQ counters of the flash kernel alone
#   3. HBM-side bytes per stage-2 iteration, both codebook regimes (tools/collect_path2_traffic.sh)
bash $GRAFT_REPO_ROOT/tools/collect_profiles.sh r5
bash $GRAFT_REPO_ROOT/tools/collect_path2_traffic.sh r5
