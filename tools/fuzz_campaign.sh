#!/bin/bash
# Randomised parity campaign over the simulator's paths and the planner's forced choices (tools/fuzz_gpu.py per setting).
set -o pipefail
O=gpurun_out/fuzz_campaign.log
: > $O
run() { echo "== $*" >> $O; env "$@" timeout -k 10 400 python -u tools/fuzz_gpu.py --circuits 400 --grams ${GRAMS:-0} --seed $((SEED + ${SEED_OFFSET:-0})) 2>&1 | grep -v amdgpu.ids | tail -3 >> $O || exit 1; }
if [ -z "$FUZZ_SECOND_HALF" ]; then
SEED=101 GRAMS=200 run QK_NOOP=1
SEED=102 run QK_SIM_PATH=reg
SEED=103 run QK_SIM_PATH=reg QK_REG_RB=3
SEED=104 run QK_SIM_PATH=reg QK_REG_RB=4 QK_REG_QTAB=1
SEED=105 run QK_SIM_PATH=reg QK_REG_OVERLAP=2 QK_REG_LV=12
fi
SEED=106 run QK_SIM_PATH=lds
SEED=107 run QK_SIM_PATH=wave
SEED=108 run QK_SIM_PATH=reg QK_RDM_REG=1
SEED=109 run QK_SIM_PATH=reg QK_REG_MERGE=0 QK_REG_QTAB=0
echo done >> $O
