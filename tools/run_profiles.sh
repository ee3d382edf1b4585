set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r01
rm -rf $O && mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --test-rows 0 > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmcA -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmcA.json 2> $O/pmcA.err
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmcB -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmcB.json 2> $O/pmcB.err
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE -d $O/pmcC -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmcC.json 2> $O/pmcC.err
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmcD -o p -- python3 tools/sim_bench.py --qubits 16 --maps zz_r1 zz_r2 > $O/pmcD.json 2> $O/pmcD.err
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/pmcE -o p -- python3 tools/sim_bench.py --qubits 16 --maps zz_r1 zz_r2 > $O/pmcE.json 2> $O/pmcE.err
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/pmcF -o p -- python3 tools/sim_bench.py --qubits 16 --maps zz_r1 zz_r2 > $O/pmcF.json 2> $O/pmcF.err
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmcG -o p -- python3 tools/config_bench.py 4 > $O/pmcG.json 2> $O/pmcG.err
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/pmcH -o p -- python3 tools/config_bench.py 4 > $O/pmcH.json 2> $O/pmcH.err
timeout -k 10 300 python3 tools/config_bench.py > $O/configs.json 2> $O/configs.err
cat $O/configs.json
