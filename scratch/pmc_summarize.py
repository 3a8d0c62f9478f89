#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same command.
usage: pmc_summarize.py <fetch_dir> <write_dir> <tag>   -> profiles/<tag>_pmc_per_kernel.csv + <tag>_pmc_traffic.json
bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md: gfx950 FETCH_SIZE counts 64 B per 128-B request)."""
import csv, glob, json, os, sys
from collections import defaultdict

def load(d, counter):
    acc, cnt = defaultdict(float), defaultdict(int)
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:                      # gpurun_out accumulates earlier runs: newest file only
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"]
            acc[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return acc, cnt

fetch, fc = load(sys.argv[1], "FETCH_SIZE")
write, wc = load(sys.argv[2], "WRITE_SIZE")
tag = sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for k in sorted(fetch, key=lambda k: -fetch[k]):
    n = fc[k]
    b = (2 * fetch[k] + write.get(k, 0.0)) * 1024 / n
    rows.append((k, n, fetch[k] / n, write.get(k, 0.0) / max(1, wc.get(k, 1)), b))
with open(os.path.join(root, "profiles", f"{tag}_pmc_per_kernel.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches", "FETCH_SIZE_per_launch", "WRITE_SIZE_per_launch", "hbm_bytes_per_launch"])
    for r in rows:
        w.writerow([r[0][:160], r[1], f"{r[2]:.1f}", f"{r[3]:.1f}", f"{r[4]:.0f}"])
def pick(sub):
    for r in rows:
        if sub in r[0]:
            return r[4], r[1]
    return None, 0
att, n_att = pick("decode_attention")
qa, n_qa = pick("decode_qa_kernel")
if n_qa:                                        # q|k|v + attention as one launch: the per-layer count comes from it
    n_att = n_qa
lm, _ = pick("lm_head_kernel")
sys.path.insert(0, root)
import bench                                    # kernel_source_stamp(): the traffic figure is tied to the kernels it was measured on
gemv = sum(r[4] * r[1] for r in rows if "decode_gemv2_kernel" in r[0])
n_layers_steps = n_att
out = {"kernel_source_stamp": bench.kernel_source_stamp(), "gemv_source_stamp": bench.kernel_source_stamp(bench.GEMV_SOURCES), "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `bench.py --steps 1 --warmup 0 --decode-tokens 16 "
                 "--no-cpu-baseline --no-extras`; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts 64 B per 128-B request)",
       "decode_attention_bytes": att, "decode_attention_launches": n_att,
       "fused_qa": bool(n_qa), "qa_source_stamp": bench.kernel_source_stamp(bench.QA_SOURCES), "decode_qa_bytes": qa,
       # the fused launch streams the same K / V rows plus the q|k|v weights (4096 x 1024 bf16)
       "decode_qa_algorithmic_bytes_at_that_context": 32 * 2 * 8 * 128 * 2 * (406 + 7.0) + 4096 * 1024 * 2,
       "layer_gemv_group_bytes": gemv / n_att if n_att else None, "lm_head_bytes": lm,
       # 16-token run, 32 rows x 30 s: contexts 406 .. 406+14 over the 15 decode steps -> mean 413; K+V rows of 8 kv heads x 128 x 2 B
       "decode_attention_algorithmic_bytes_at_that_context": 32 * 2 * 8 * 128 * 2 * (406 + 7.0)}
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
