#!/bin/bash
# Instruction-fetch side of the LDS simulator: tools/pmc_sim_icache.sh <qubits> <map> <outdir> [gib]
set -o pipefail
n=$1; map=$2; out=gpurun_out/$3; gib=${4:-1}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU --kernel-trace -d $R/$out/p1 -o p1 --output-format csv -- python3 $R/tools/sim_bench.py --qubits $n --gib $gib --maps $map > $R/$out/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d $R/$out/p2 -o p2 --output-format csv -- python3 $R/tools/sim_bench.py --qubits $n --gib $gib --maps $map > $R/$out/p2.log 2>&1 || exit 1
cd $R
python3 - <<PY
import csv,glob,collections
for p in ("p1","p2"):
    for f in glob.glob("$out/%s/**/*counter_collection.csv"%p, recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:50]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        for k,v in acc.items():
            if "sim_pass" in k: print(p,k,len(n[k]),{a:int(b/len(n[k])) for a,b in v.items()})
PY
