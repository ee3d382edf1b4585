#!/bin/bash
# Where the cycles of a short-row Gram launch go: MFMA pipe busy, instruction mix and wave states.
# tools/pmc_gram_issue.sh <rows> <qubits> <outdir> [variant ...]
set -o pipefail
N=$1; q=$2; out=gpurun_out/$3; shift 3
V=${@:-ring3_64x64_w4}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $R/$out/pA -o p --output-format csv -- python3 $R/tools/gram_ab.py --rows $N --qubits $q --rounds 1 $V > $R/$out/a.log 2>&1 || { tail -5 $R/$out/a.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --kernel-trace -d $R/$out/pB -o p --output-format csv -- python3 $R/tools/gram_ab.py --rows $N --qubits $q --rounds 1 $V > $R/$out/b.log 2>&1 || { tail -5 $R/$out/b.log; exit 1; }
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
            print(p, k, {a: "%.5g" % b[-1] for a, b in v.items()})
PY
grep variant $out/a.log
