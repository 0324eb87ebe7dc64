#!/usr/bin/env python3
"""HBM bytes per launch from two rocprofv3 counter passes of the bench command (tools/pmc_bench.sh NAME "FETCH_SIZE" /
"WRITE_SIZE"): `tools/pmc_traffic.py gpurun_out/pmc_FETCH gpurun_out/pmc_WRITE profiles/rNN` writes
profiles/rNN_pmc_hbm_traffic.txt (per kernel) and profiles/rNN_pmc_hbm_traffic.json (what bench.py reads).
FETCH_SIZE / WRITE_SIZE count KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of wide coalesced reads at 64 B
(MI355X_MICROARCH.md, HBM): the read side is doubled."""
import collections
import csv
import glob
import json
import sys


def load(d, name):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
prefix = sys.argv[3]
shape = [int(v) for v in sys.argv[4:7]] if len(sys.argv) >= 7 else [512, 2048, 2048]
rows = {}
lines = [f"{'kernel':70s} {'calls':>5s} {'fetch GB':>9s} {'fetch x2':>9s} {'write GB':>9s} {'moved GB':>9s}"]
for k in sorted(fetch, key=lambda k: -sum(fetch[k])):
    if "bh::" not in k:
        continue
    fv = sum(fetch[k]) / len(fetch[k]) * 1024 / 1e9
    wv = sum(write.get(k, [0])) / max(1, len(write.get(k, [0]))) * 1024 / 1e9
    rows[k] = (len(fetch[k]), 2 * fv, wv)
    lines.append(f"{k[:70]:70s} {len(fetch[k]):5d} {fv:9.3f} {2 * fv:9.3f} {wv:9.3f} {2 * fv + wv:9.3f}")


def moved(sub, count=1):
    hit = [v for k, v in rows.items() if sub in k]
    assert len(hit) == 1, (sub, [k for k in rows if sub in k])
    return count * (hit[0][1] + hit[0][2]) * 1e9


# one Richardson-Lucy iteration: 2 x (Y fwd, Z x OTF, Y inv) + fused ratio X + fused update X.  Which kernels those are
# depends on the build's defaults (register-stage or LDS-stepped column passes; real or complex transfer function): take the
# per-launch bytes of whatever ran, weighted by how often it ran per iteration (calls / iterations).
def per_iteration(sub):
    hit = [(k, v) for k, v in rows.items() if sub in k]
    return sum(v[1] + v[2] for _, v in hit) * 1e9 / max(1, len(hit)), sum(v[0] for _, v in hit)


n_it = rows[[k for k in rows if "xw_kernel<10, 4>" in k][0]][0]  # one fused-ratio launch per iteration
it = moved("xw_kernel<10, 4>") + moved("xw_kernel<10, 5>")
col = {k: v for k, v in rows.items() if "col_pass_kernel<" in k or "colw_kernel<" in k or "colz_kernel<" in k}
for k, v in col.items():
    per_it = round(v[0] / n_it)          # launches of this kernel per iteration (the OTF build adds a stray call or two)
    it += per_it * (v[1] + v[2]) * 1e9
# deskew + overhang fill as bh_deskew launches them: the resampling kernel plus every kernel of csrc/deskew_rows.inc and csrc/fill.hip
# that ran beside it (the conditional mask pipeline moves nothing when the one-pass path took the volume), per deskew call
dk_rows = {k: v for k, v in rows.items() if "deskew_pers_kernel<" in k or "deskew_kernel<" in k}
dk_key = max(dk_rows, key=lambda k: dk_rows[k][1] + dk_rows[k][2])  # the resampling kernel that moved the volume (the conditional one moved nothing)
n_dk = dk_rows[dk_key][0]
pair = 0.0
for k, v in rows.items():
    if any(t in k for t in ("deskew_pers_kernel<", "deskew_kernel<", "rows::", "dilate_", "shell_kernel", "apply_fill_kernel", "finalize_kernel",
                            "mask0_kernel")):
        pair += v[0] / n_dk * (v[1] + v[2]) * 1e9
rec = {"shape": shape, "unit": "bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, KiB counters x1024)", "rl_iteration": it, "deskew_pair": pair,
       "deskew_kernel": (dk_rows[dk_key][1] + dk_rows[dk_key][2]) * 1e9,
       "per_kernel_gb": {k[:60]: round(v[1] + v[2], 3) for k, v in rows.items()},
       "source": f"{prefix}_pmc_hbm_traffic.txt: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of "
                 "`python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-ops`, summed over the 8 passes of one "
                 "iteration.  The x2 of FETCH_SIZE is calibrated for 16-B-per-lane reads; the real-OTF Z pass reads its multiplier "
                 "with 8 B per lane, which the counter probably tallies exactly, so its fetch (and the iteration total, by up to "
                 "2 x 4.4 GB at config 2) may be overstated"}
open(prefix + "_pmc_hbm_traffic.txt", "w").write("\n".join(lines) + "\n")
json.dump(rec, open(prefix + "_pmc_hbm_traffic.json", "w"), indent=1)
print("\n".join(lines[:14]))
print("rl_iteration GB:", it / 1e9, " deskew kernel GB:", rec["deskew_kernel"] / 1e9, " deskew + fill pair GB:", pair / 1e9)
