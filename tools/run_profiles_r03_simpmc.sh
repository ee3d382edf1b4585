#!/bin/bash
# Round-3 profiles, the simulator's counter passes only (same commands and file names as in run_profiles_r03.sh).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
for q in 8 10 12 16; do
  rm -rf $O/sim$q
  echo "== sim$q" >> $O/progress.log
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/sim$q -o p -- python3 tools/sim_bench.py --qubits $q --gib 1 --maps zz_r1 zz_r2 genome_g8 > $O/sim$q.json 2> $O/sim$q.err || exit 1
done
