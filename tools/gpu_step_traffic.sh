#!/bin/bash
# Whole-step HBM traffic: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes: FETCH_SIZE takes 3 of the 4 TCC
# slots) over `bench.py --steps 2 --warmup 1` (3 identical train steps), summed over EVERY kernel dispatch and divided by 3.
# FETCH_SIZE counts 64 B per 128-B request on gfx950 (MI355X_MICROARCH.md "HBM"): doubled.  Output: gpurun_out/step_traffic.{txt,json}
export TMPDIR=/tmp
mkdir -p gpurun_out
for CNT in FETCH_SIZE WRITE_SIZE; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_step_$CNT
  rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && timeout 900 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-infer --no-fp32 --no-free-run > $OUT/run.log 2>&1)
done
python3 - <<'PY'
import csv, glob, json, os, re, collections
root = os.environ.get("GRAFT_REPO_ROOT", ".")
tot = {}
per = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    s = 0.0
    for f in glob.glob(root + "/gpurun_out/pmc_step_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c: continue
            v = float(r["Counter_Value"]); s += v
            k = re.sub(r"\(.*", "", r["Kernel_Name"])[:110]
            per[k][c] += v
            if c == "FETCH_SIZE": cnt[k] += 1
    tot[c] = s
steps = 3.0
fetch_b = tot["FETCH_SIZE"] * 1024 * 2 / steps      # KB, x2 gfx950 correction
write_b = tot["WRITE_SIZE"] * 1024 / steps
lines = ["# whole-step HBM traffic: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over bench.py --steps 2 --warmup 1 (3 steps), all kernels, per step",
         "# FETCH_SIZE (KB) x2 (64 B counted per 128-B request on gfx950), WRITE_SIZE (KB) as is",
         "fetch_bytes_per_step %.4e" % fetch_b, "write_bytes_per_step %.4e" % write_b, "traffic_bytes_per_step %.4e" % (fetch_b + write_b), "",
         "%-112s %7s %12s %12s" % ("kernel", "calls", "fetch MB/st", "write MB/st")]
rows = sorted(per.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"] * 2 + kv[1]["WRITE_SIZE"]))
for k, d in rows[:60]:
    lines.append("%-112s %7d %12.1f %12.1f" % (k, cnt[k], d["FETCH_SIZE"] * 2 * 1024 / steps / 1e6, d["WRITE_SIZE"] * 1024 / steps / 1e6))
open(root + "/gpurun_out/step_traffic.txt", "w").write("\n".join(lines) + "\n")
json.dump({"fetch_bytes": fetch_b, "write_bytes": write_b, "traffic_bytes": fetch_b + write_b,
           "per_kernel": {k: {"calls_3_steps": cnt[k], "fetch_bytes_per_step": d["FETCH_SIZE"] * 2 * 1024 / steps, "write_bytes_per_step": d["WRITE_SIZE"] * 1024 / steps} for k, d in rows[:60]}},
          open(root + "/gpurun_out/step_traffic.json", "w"), indent=1)
print("\n".join(lines[:48]))
PY
rm -rf gpurun_out/pmc_step_FETCH_SIZE gpurun_out/pmc_step_WRITE_SIZE
