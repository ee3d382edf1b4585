#!/bin/bash
# Phase-timing build of the LDS simulator: tools/sim_trace.sh  ->  tools/_abl/libquake_hip_simtrace.so
set -e
cd "$(dirname "$0")/.."
python -c "from quake_amd import build; build.build_library(verbose=False)"
mkdir -p tools/_abl
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Iquake_amd/csrc -DQK_SIM_TRACE -x hip -c quake_amd/csrc/qk_sim.hip -o tools/_abl/qk_sim_trace.o
objs=$(ls quake_amd/csrc/build/*.o | grep -v qk_sim.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs tools/_abl/qk_sim_trace.o -o tools/_abl/libquake_hip_simtrace.so
ls -la tools/_abl/libquake_hip_simtrace.so
