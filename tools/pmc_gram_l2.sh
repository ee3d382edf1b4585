#!/bin/bash
# L2 hit / miss / request counts of the Gram kernel variants: tools/pmc_gram_l2.sh <rows> <qubits> <outdir> "<counters>"
set -o pipefail
N=$1; q=$2; out=gpurun_out/$3; ctr=${4:-"TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V="dma3m_64x64_w4_g64 ring5_64x64_w4"
timeout -s KILL 250 rocprofv3 --pmc $ctr --kernel-trace -d $R/$out/pL -o p --output-format csv -- python3 $R/tools/gram_ab.py --rows $N --qubits $q --rounds 1 $V > $R/$out/l.log 2>&1 || { tail -5 $R/$out/l.log; exit 1; }
cd $R
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$out/pL/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "gram" not in r["Kernel_Name"]: continue
        acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, {a: "%.4g" % b[-1] for a, b in v.items()})
PY
