#!/bin/bash
# PMC passes (issue / wait / LDS) over one simulator configuration: tools/pmc_sim2.sh <qubits> <map> <outdir> [gib]
set -o pipefail
n=$1; map=$2; out=gpurun_out/$3; gib=${4:-4}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $R/$out/p1 -o p1 --output-format csv -- python3 $R/tools/sim_bench.py --qubits $n --gib $gib --maps $map > $R/$out/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM --kernel-trace -d $R/$out/p2 -o p2 --output-format csv -- python3 $R/tools/sim_bench.py --qubits $n --gib $gib --maps $map > $R/$out/p2.log 2>&1 || exit 1
cd $R
python3 - <<PY
import csv,glob,collections
for p in ("p1","p2"):
    for f in glob.glob("$out/%s/**/*counter_collection.csv"%p, recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:70]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        for k,v in acc.items():
            if "sim" in k: print(p,k,len(n[k]),{a:int(b) for a,b in v.items()})
PY
