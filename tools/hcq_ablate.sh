#!/bin/bash
# Timing variants of csrc/hcq_conv.hip with phases switched off (HCQ_DBG bits, see the source): builds
# tools/_bin/libseld_hcqdbg<N>.so from the variant object + the regular objects of the other files.
#   tools/hcq_ablate.sh 1 2 4 8 3 7 15     then     SELD_HIP_LIB=tools/_bin/libseld_hcqdbg1.so python tools/hcq_check.py cnn1
set -e
cd "$(dirname "$0")/.."
CS=sound-event-localization-and-detection_amd/csrc
mkdir -p tools/_bin
OTHERS=$(ls $CS/*.o | grep -v hcq_conv.o)
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wno-pass-failed -mllvm -amdgpu-mfma-vgpr-form=1 -DHCQ_DBG=$n \
      -c $CS/hcq_conv.hip -o tools/_bin/hcq_conv_dbg$n.o &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libseld_hcqdbg$n.so tools/_bin/hcq_conv_dbg$n.o $OTHERS
done
ls -la tools/_bin/*.so
