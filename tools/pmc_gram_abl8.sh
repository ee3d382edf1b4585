#!/bin/bash
# clock and time of the Gram kernel with and without L2-miss traffic (ablation build 8: every DMA reads stage 0)
set -o pipefail
N=$1; q=$2; out=gpurun_out/$3
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in 0 8; do
  timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES TCC_MISS_sum --kernel-trace -d $R/$out/p$m -o p --output-format csv -- python3 $R/tools/gram_ablate.py --rows $N --qubits $q --rounds 2 --child $m > $R/$out/l$m.log 2>&1 || { tail -5 $R/$out/l$m.log; exit 1; }
  python3 - <<PY
import csv, glob, collections
for f in glob.glob("$R/$out/p$m/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gram" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("mask $m", {a: "%.4g" % b[-1] for a, b in acc.items()})
PY
  grep ablate_mask $R/$out/l$m.log
done
