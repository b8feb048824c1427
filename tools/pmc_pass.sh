#!/bin/bash
# One rocprofv3 counter pass over a short bench run; usage: tools/pmc_pass.sh <tag> <counter> [<counter> ...]
# Writes gpurun_out/pmc_<tag>/ and prints per-kernel sums.  Each pass is its own process (rocprofv3 cannot split passes).
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-h2d --no-other-configs > gpurun_out/pmc_$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob(f"gpurun_out/pmc_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k, {c: f"{v / n[(k, c)]:.4g}/launch" for c, v in acc[k].items()})
PY
