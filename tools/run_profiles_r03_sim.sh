#!/bin/bash
# Round-3 profiles, simulator part only (after a change that touches nothing else): same steps and file names as
# run_profiles_r03.sh, plus a kernel trace of the two-pass circuits at 16 qubits.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
rm -rf $O/sim_kt $O/sim16_kt
step() { echo "== $1" >> $O/progress.log; }
step simbench; timeout -k 10 300 python3 tools/sim_bench.py --qubits 16 14 13 12 11 10 9 8 7 6 --gib 4 --maps h_only z_r2 zz_r1 zz_r2 zz_full_r2 custom1_r1 custom1_r2 genome_g8 > $O/sim_bench.json 2> $O/sim_bench.err || exit 1
step simbench_noreg; QK_SIM_PATH=noreg timeout -k 10 300 python3 tools/sim_bench.py --qubits 13 12 11 10 9 8 7 6 --gib 4 --maps h_only z_r2 zz_r1 zz_r2 zz_full_r2 custom1_r1 custom1_r2 genome_g8 > $O/sim_bench_noreg.json 2> $O/sim_bench_noreg.err || exit 1
step simtrace; timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/sim_kt -o kt -- python3 tools/sim_bench.py --qubits 12 8 --gib 4 --maps zz_r2 genome_g8 > /dev/null 2> $O/sim_kt.err || exit 1
step sim16trace; timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/sim16_kt -o kt -- python3 tools/sim_bench.py --qubits 16 --gib 4 --maps zz_r2 > $O/sim16.json 2> $O/sim16_kt.err || exit 1
step latency; timeout -k 10 200 python3 tools/latency_bench.py > $O/latency.json 2> $O/latency.err || exit 1
step done
