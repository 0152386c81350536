#!/bin/bash
# SIMD issue microbenchmarks (instruction costs, matrix/vector overlap, the flash tile's instruction mix) -> stdout
set -e
cd "$(dirname "$0")/../.."
mkdir -p scratch
for b in valu_rates overlap flash_mix; do
  [ -x scratch/$b ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/micro/$b.hip -o scratch/$b
  echo "### tools/micro/$b.hip"; ./scratch/$b
done
