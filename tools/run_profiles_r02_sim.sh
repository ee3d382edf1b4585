set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02
mkdir -p $O
for q in 8 10 12 16; do
  rm -rf $O/sim$q
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/sim$q -o p -- python3 tools/sim_bench.py --qubits $q --gib 1 --maps zz_r1 zz_r2 > $O/sim$q.json 2> $O/sim$q.err || exit 1
done
timeout -k 10 300 python3 tools/sim_bench.py --qubits 16 13 12 11 10 8 --gib 4 --maps h_only zz_r1 zz_r2 genome_g8 > $O/sim_bench.json 2> $O/sim_bench.err || exit 1
QK_SIM_PATH=lds timeout -k 10 300 python3 tools/sim_bench.py --qubits 10 8 --gib 4 --maps h_only zz_r1 zz_r2 genome_g8 > $O/sim_bench_lds.json 2> $O/sim_bench_lds.err || exit 1
timeout -k 10 300 python3 tools/config_bench.py > $O/configs.json 2> $O/configs.err || exit 1
echo partial done
