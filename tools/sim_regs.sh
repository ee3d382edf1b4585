#!/bin/bash
# register / scratch use of the LDS simulator's kernels (compile only): tools/sim_regs.sh
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Iquake_amd/csrc -Rpass-analysis=kernel-resource-usage \
  -x hip -c quake_amd/csrc/qk_sim.hip -o /tmp/qk_sim_regs.o 2>&1 | grep -E "Function Name|VGPRs:|SGPRs:|ScratchSize|Occupancy" \
  | sed -e 's/.*remark: [^ ]* *//' -e 's/ \[-Rpass.*//' | paste - - - - -
