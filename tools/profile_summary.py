#!/usr/bin/env python3
"""Condense gpurun_out/<tag>/ (tools/profile_round.sh) into the files kept under profiles/:
  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --steps 3 --warmup 1`
  <tag>_pmc.txt            mean counter values per kernel of the separate --pmc passes
  <tag>_bench.json         the default bench line of the same build
  traffic_latest.json      HBM bytes per launch of K1 + the issue counters of its pass A, stamped with the commit
usage: tools/profile_summary.py gpurun_out/<tag> [commit]"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

out = sys.argv[1].rstrip("/")
tag = os.path.basename(out)
root = os.path.dirname(os.path.dirname(os.path.abspath(out)))
prof = os.path.join(root, "profiles")
commit = sys.argv[2] if len(sys.argv) > 2 else None
if commit is None:
    try:
        commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        commit = "unknown (run on the GPU box: pass the commit as the second argument)"

agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
lines = []
for k in sorted(agg):
    if not any(s in k for s in ("count_", "stats_", "em_", "bg_", "mirror")):
        continue
    lines.append(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        lines.append("   %-28s mean=%.5g  n=%d" % (c, sum(v) / len(v), len(v)))
open(os.path.join(prof, tag + "_pmc.txt"), "w").write("\n".join(lines) + "\n")

stats = glob.glob(os.path.join(out, "stats", "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(prof, tag + "_kernel_stats.csv"))
if os.path.exists(os.path.join(out, "bench.json")):
    shutil.copy(os.path.join(out, "bench.json"), os.path.join(prof, tag + "_bench.json"))


def mean(k, c):
    for name in agg:
        if name.startswith(k):
            v = agg[name].get(c)
            if v:
                return sum(v) / len(v)
    return None


def kernel_ms(prefix):
    if not stats:
        return None
    for row in csv.DictReader(open(stats[0])):
        if prefix in row["Name"]:
            return float(row["AverageNs"]) * 1e-6
    return None


KIB = 1024.0
sc_f, sc_w = mean("pengk::count_scatter_kernel", "FETCH_SIZE"), mean("pengk::count_scatter_kernel", "WRITE_SIZE")
hi_f, hi_w = mean("pengk::count_hist_kernel", "FETCH_SIZE"), mean("pengk::count_hist_kernel", "WRITE_SIZE")
ga_f, ga_w = mean("pengk::count_gather_kernel", "FETCH_SIZE"), mean("pengk::count_gather_kernel", "WRITE_SIZE")
if None not in (sc_f, sc_w, hi_f, hi_w, ga_f, ga_w):
    t = {
        "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/profile_round.sh), bench.py at its default size "
                "(10M x 200 bp, W=10, both strands), per launch. Counters are KiB. FETCH_SIZE of the 16-B-per-lane streaming read in "
                "count_hist_kernel is doubled per MI355X_MICROARCH.md (gfx950 reports 1/2); count_scatter_kernel reads 4 B per "
                "lane (uncalibrated width): raw value kept, doubled value in *_if_doubled.",
        "commit": commit, "config": "BASELINE configs[2], 10M x 200 bp, W=10, BOTH", "profile": tag,
        "count_scatter_kernel": {"fetch_bytes_raw": sc_f * KIB, "fetch_bytes_if_doubled": 2 * sc_f * KIB, "write_bytes": sc_w * KIB},
        "count_hist_kernel": {"fetch_bytes_raw": hi_f * KIB, "fetch_bytes_corrected": 2 * hi_f * KIB, "write_bytes": hi_w * KIB},
        "count_gather_kernel": {"fetch_bytes_raw": ga_f * KIB, "write_bytes": ga_w * KIB},
        "count_kernel_hbm_bytes_per_launch": (sc_f + sc_w + 2 * hi_f + hi_w + ga_f + ga_w) * KIB,
        "algorithmic_bytes_per_launch": 584194304,
    }
    valu, salu, lds = (mean("pengk::count_scatter_kernel", c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"))
    busy, act_valu = mean("pengk::count_scatter_kernel", "SQ_BUSY_CYCLES"), mean("pengk::count_scatter_kernel", "SQ_ACTIVE_INST_VALU")
    ms = kernel_ms("count_scatter_kernel")
    if None not in (valu, salu, lds, busy, act_valu, ms):
        n_simd, clk = 1024, 2.4e9
        peak = n_simd * clk / 4.0  # one wave-instruction per SIMD every 4 cycles (tools/ubench/valu_rate2.hip: 4.2-4.6 measured)
        cycles = busy / 32.0       # SQ_BUSY_CYCLES is summed over the 32 shader engines
        t["issue"] = {
            "kernel": "count_scatter_kernel<10,both> (pass A of K1)", "bound": "instruction issue (VALU)",
            "achieved": valu / (ms * 1e-3), "peak": peak, "unit": "VALU wave-instructions/s", "frac": valu / (ms * 1e-3) / peak,
            "valu_busy_frac": act_valu * 4.0 / (n_simd * cycles),
            "per_launch": {"valu_wave_instructions": valu, "salu_instructions": salu, "lds_instructions": lds,
                           "kernel_ms": ms, "busy_cycles_per_shader_engine": cycles,
                           "wave_steps_of_64_windows": 1.91e9 / 64, "valu_per_wave_step": valu / (1.91e9 / 64),
                           "all_instructions_per_wave_step": (valu + salu + lds) / (1.91e9 / 64)},
            "source": "rocprofv3 --pmc SQ_* passes at commit %s (%s), not this run" % (commit, tag),
            "note": "the byte roofline cannot describe this kernel: it moves 14x its algorithmic bytes and still uses a fifth of the HBM "
                    "rate; what it is near is the rate at which 4 waves per SIMD (LDS-limited occupancy) can issue instructions",
        }
    json.dump(t, open(os.path.join(prof, "traffic_latest.json"), "w"), indent=1)
print("\n".join(lines[:60]))
