#!/bin/bash
# builds tools/_bin/gw_micro_<mask> for the masks given (default: a standard set); cross-compiles for gfx950
set -e
cd "$(dirname "$0")"
mkdir -p _bin
for m in ${@:-0 1 2 3 4 8 9 11}; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -Wno-pass-failed -mllvm -amdgpu-mfma-vgpr-form=1 -DGW_DBG=$m -x hip gw_micro.cpp -o _bin/gw_micro_$m &
done
wait
ls -la _bin
