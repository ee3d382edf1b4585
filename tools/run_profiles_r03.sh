#!/bin/bash
# Round-3 profiles: the default bench under rocprofv3 --kernel-trace --stats, PMC passes (one counter block per
# run, no trace flags beside --kernel-trace) and the secondary benches.  Everything lands under gpurun_out/r03/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
rm -rf $O && mkdir -p $O
step() { echo "== $1" >> $O/progress.log; }
step kt;   timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err || exit 1
tail -c 600 $O/bench.json
step pmcA; timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmcA -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-path --test-rows 0 > $O/pmcA.json 2> $O/pmcA.err || exit 1
step pmcB; timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmcB -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-path --test-rows 0 > $O/pmcB.json 2> $O/pmcB.err || exit 1
step pmcC; timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE -d $O/pmcC -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-path --test-rows 0 > $O/pmcC.json 2> $O/pmcC.err || exit 1
# simulator: register-resident kernel (8, 10, 12 qubits) and LDS kernel (16 qubits), issue / wait / LDS counters
for q in 8 10 12 16; do
  step sim$q
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/sim$q -o p -- python3 tools/sim_bench.py --qubits $q --gib 1 --maps zz_r1 zz_r2 genome_g8 > $O/sim$q.json 2> $O/sim$q.err || exit 1
done
step pqkG; timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmcG -o p -- python3 tools/config_bench.py 4 > $O/pmcG.json 2> $O/pmcG.err || exit 1
step pqkH; timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/pmcH -o p -- python3 tools/config_bench.py 4 > $O/pmcH.json 2> $O/pmcH.err || exit 1
step configs; timeout -k 10 300 python3 tools/config_bench.py > $O/configs.json 2> $O/configs.err || exit 1
step simbench; timeout -k 10 300 python3 tools/sim_bench.py --qubits 16 14 13 12 11 10 9 8 7 6 --gib 4 --maps h_only z_r2 zz_r1 zz_r2 zz_full_r2 custom1_r1 custom1_r2 genome_g8 > $O/sim_bench.json 2> $O/sim_bench.err || exit 1
step simbench_noreg; QK_SIM_PATH=noreg timeout -k 10 300 python3 tools/sim_bench.py --qubits 13 12 11 10 9 8 7 6 --gib 4 --maps h_only z_r2 zz_r1 zz_r2 zz_full_r2 custom1_r1 custom1_r2 genome_g8 > $O/sim_bench_noreg.json 2> $O/sim_bench_noreg.err || exit 1
step simtrace; timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/sim_kt -o kt -- python3 tools/sim_bench.py --qubits 12 8 --gib 4 --maps zz_r2 genome_g8 > /dev/null 2> $O/sim_kt.err || exit 1
step ga; timeout -k 10 100 python3 tools/ga_bench.py > $O/ga_bench.json 2> $O/ga_bench.err || exit 1
step ga_trace; timeout -k 10 100 rocprofv3 --kernel-trace --stats -d $O/ga_kt -o kt -- python3 tools/ga_bench.py > /dev/null 2> $O/ga_kt.err || exit 1
step latency; timeout -k 10 200 python3 tools/latency_bench.py > $O/latency.json 2> $O/latency.err || exit 1
step pqk_genome; timeout -k 10 200 python3 bench.py --workload pqk --pqk-map genome --steps 5 > $O/pqk_genome.json 2> $O/pqk_genome.err || exit 1
step pqk_zz; timeout -k 10 200 python3 bench.py --workload pqk --steps 5 > $O/pqk_zz.json 2> $O/pqk_zz.err || exit 1
step done
cat $O/configs.json
