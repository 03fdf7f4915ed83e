#!/usr/bin/env python3
"""profiles/pmc_traffic.json <- gpurun_out/step_traffic.json (tools/gpu_step_traffic.sh): the step's HBM traffic and the per-launch traffic of the
dominant kernel (conv_wide_kernel<3, 0, false, 0> = bench.py's "conv_wide_kernel<bf16,BN=256,KS=3,MODE=0>").  Usage: python tools/update_pmc_json.py <round tag>"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
st = json.load(open(os.path.join(ROOT, "gpurun_out", "step_traffic.json")))
p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
d = json.load(open(p))
d["step"] = {"fetch_bytes": st["fetch_bytes"], "write_bytes": st["write_bytes"], "traffic_bytes": st["traffic_bytes"],
             "source": "%s: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --steps 2 --warmup 1, all kernel dispatches / 3 steps "
                       "(tools/gpu_step_traffic.sh, profiles/%s_step_traffic.txt); FETCH_SIZE x2 per the gfx950 correction" % (tag, tag)}
NAMES = {"void uegan::conv_wide_kernel<3, 0, false": "conv_wide_kernel<bf16,BN=256,KS=3,MODE=0>",
         "void uegan::conv_tall_kernel<4, 0, false, false": "conv_tall_kernel<bf16,BN=128,KS=3,MODE=0>",      # (both tile heights: <..., 2> and <..., 4>)
         "void uegan::conv_tall_kernel<4, 1, true, false": "conv_tall_kernel<bf16,BN=128,KS=3,MODE=1>"}
agg = {}
for k, v in st["per_kernel"].items():
    for pre, name in NAMES.items():
        if k.startswith(pre):
            a = agg.setdefault(name, [0, 0.0, 0.0])
            a[0] += v["calls_3_steps"]; a[1] += v["fetch_bytes_per_step"] * 3; a[2] += v["write_bytes_per_step"] * 3
for name, (n, f, w) in agg.items():
    d[name] = {"fetch_bytes": f / n, "write_bytes": w / n, "traffic_bytes": (f + w) / n, "launches_averaged": n,
               "round": "%s (profiles/%s_step_traffic.txt): %d launches per step" % (tag, tag, n // 3)}
json.dump(d, open(p, "w"), indent=1)
print(json.dumps({k: v for k, v in d.items() if k == "step" or "conv_tall" in k}, indent=1))
