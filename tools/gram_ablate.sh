#!/bin/bash
# Ablation builds of the Gram kernel (results are wrong on purpose; timing only).  Run here on the CPU:
#   tools/gram_ablate.sh 1 2 3 7      -> tools/_abl/libquake_hip_abl<mask>.so
# then on the GPU:  python tools/gram_ablate.py --rows 16384 --qubits 12 0 1 2 3 7
set -e
cd "$(dirname "$0")/.."
python -c "from quake_amd import build; build.build_library(verbose=False)"
mkdir -p tools/_abl
for m in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Iquake_amd/csrc -DQK_GRAM_ABLATE=$m \
      -x hip -c quake_amd/csrc/qk_gram.hip -o tools/_abl/qk_gram_abl$m.o &
done
wait
for m in "$@"; do
  objs=$(ls quake_amd/csrc/build/*.o | grep -v qk_gram.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs tools/_abl/qk_gram_abl$m.o -o tools/_abl/libquake_hip_abl$m.so
done
ls -la tools/_abl/*.so
