#!/bin/bash
# instructions per sample of each circuit of tools/wave_cost.py: tools/pmc_wave_cost.sh <qubits> <outdir>
set -o pipefail
n=$1; out=gpurun_out/$2
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace -d $R/$out/p -o p --output-format csv -- python3 $R/tools/wave_cost.py --qubits $n > $R/$out/ladder.log 2>&1 || exit 1
cd $R
python3 - <<PY
import csv, glob, collections, json
names = [json.loads(l)["circuit"] for l in open("$out/ladder.log") if l.startswith("{")]
ms = [json.loads(l)["ms"] for l in open("$out/ladder.log") if l.startswith("{")]
for f in glob.glob("$out/p/**/*counter_collection.csv", recursive=True):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "wave_sim" not in r["Kernel_Name"]: continue
        acc.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    d = [acc[k] for k in sorted(acc)]
    # two launches per circuit (warm + timed): take the second
    for i, nm in enumerate(names):
        v = d[2 * i + 1]
        print(nm, "ms", ms[i], "VALU/sample", round(v["SQ_INSTS_VALU"] / 65536), "SALU/sample", round(v["SQ_INSTS_SALU"] / 65536), "LDS/sample", round(v["SQ_INSTS_LDS"] / 65536))
PY
