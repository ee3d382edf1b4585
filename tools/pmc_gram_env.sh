#!/bin/bash
# L2 misses, clock and time of the default Gram kernel under different environment settings, one process each:
# tools/pmc_gram_env.sh <rows> <qubits> <outdir> "VAR=a" "VAR=b" ...
set -o pipefail
N=$1; q=$2; out=gpurun_out/$3; shift 3
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for setting in "$@"; do
  i=$((i+1))
  export "$setting"
  timeout -s KILL 200 rocprofv3 --pmc TCC_MISS_sum TCC_HIT_sum GRBM_GUI_ACTIVE --kernel-trace -d $R/$out/p$i -o p --output-format csv -- python3 $R/tools/gram_ab.py --rows $N --qubits $q --rounds 2 auto > $R/$out/l$i.log 2>&1 || { tail -5 $R/$out/l$i.log; exit 1; }
  python3 - <<PY
import csv, glob, collections
for f in glob.glob("$R/$out/p$i/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gram" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("$setting", {a: "%.4g" % b[-1] for a, b in acc.items()})
PY
  grep variant $R/$out/l$i.log
done
