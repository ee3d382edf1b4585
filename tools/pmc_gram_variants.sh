#!/bin/bash
# L2-miss traffic and clock of the Gram kernel variants in one process: tools/pmc_gram_variants.sh <rows> <qubits> <outdir>
set -o pipefail
N=$1; q=$2; out=gpurun_out/$3
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V="dma3m_64x64_w4_g64 ring3_64x64_w4 ring5_64x64_w4"
timeout -s KILL 250 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$out/pA -o p --output-format csv -- python3 $R/tools/gram_ab.py --rows $N --qubits $q --rounds 1 $V > $R/$out/a.log 2>&1 || exit 1
timeout -s KILL 250 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $R/$out/pB -o p --output-format csv -- python3 $R/tools/gram_ab.py --rows $N --qubits $q --rounds 1 $V > $R/$out/b.log 2>&1 || exit 1
cd $R
python3 - <<PY
import csv, glob, collections
for p in ("pA", "pB"):
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % p, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "gram" not in r["Kernel_Name"]: continue
            acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(p, k, {a: ["%.4g" % x for x in b] for a, b in v.items()})
PY
grep variant $out/a.log
