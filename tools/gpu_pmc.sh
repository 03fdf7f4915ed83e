#!/bin/bash
# usage: tools/gpu_pmc.sh "<COUNTER ...>" <kernel-substring> <command...> : per-kernel mean of PMC counters (rocprofv3 csv)
CNT="$1"; KSUB="$2"; shift 2
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_tmp
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout 240 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT -- "$@" > $OUT/run.log 2>&1)
python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, sys, collections
out, ksub = sys.argv[1], sys.argv[2]
files = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
if not files:
    print("no counter csv; files:", glob.glob(out + "/**/*", recursive=True)[:20]); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if ksub in k:
            acc[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r.get("End_Timestamp") and r.get("Start_Timestamp"):
                dur[k[:90]].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3)
for k, d in acc.items():
    print(k)
    if dur[k]:
        print("   %-28s n=%d mean=%.4g" % ("duration_us", len(dur[k]), sum(dur[k]) / len(dur[k])))
    for c, v in sorted(d.items()):
        print("   %-28s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
