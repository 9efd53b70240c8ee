#!/usr/bin/env python3
"""Condense gpurun_out/<tag>/ (tools/profile_round.sh) into the files kept under profiles/:
  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --steps 3 --warmup 1`
  <tag>_pmc.txt            mean counter values per kernel of the separate --pmc passes
  <tag>_bench.json         the default bench line of the same build
  traffic_by_config.json   HBM bytes per launch of K1 + the issue counters of its scan, one entry per configuration
                           (W / size / strand mode), each stamped with the commit it was taken at
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

stats = sorted(glob.glob(os.path.join(out, "stats", "*", "*kernel_stats.csv")), key=os.path.getmtime, reverse=True)  # newest run first
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
# ---- HBM traffic of K1 per launch, for the configuration this profile was taken at --------------------------------------
# K1 = every pengk::count_* kernel of a pengk_count_bg launch plus bg_finish_fused_kernel.  FETCH_SIZE / WRITE_SIZE are
# KiB; per MI355X_MICROARCH.md (HBM section) gfx950 reports HALF the bytes of a 16-B-per-lane streaming read, so the
# kernels that read their keys that way (count_hist_kernel, count_rescatter12_kernel) have FETCH_SIZE doubled; the
# scans read 4 / 8 B per lane (uncalibrated width): raw value kept.
bench = None
if os.path.exists(os.path.join(out, "bench.json")):
    try:
        bench = json.loads([l for l in open(os.path.join(out, "bench.json")) if l.startswith("{")][-1])
    except Exception:
        bench = None
WIDE_READERS = ("pengk::count_hist_kernel", "pengk::count_rescatter12_kernel")
k1 = {}
total = 0.0
complete = True
for name in sorted(agg):
    if not (name.startswith("pengk::count_") or name.startswith("pengk::bg_finish_fused")):
        continue
    f, w = agg[name].get("FETCH_SIZE"), agg[name].get("WRITE_SIZE")
    if not f or not w:
        complete = False
        continue
    f, w = sum(f) / len(f), sum(w) / len(w)
    fc = 2 * f if name.startswith(WIDE_READERS) else f
    k1[name] = {"fetch_bytes_raw": f * KIB, "fetch_bytes_corrected": fc * KIB, "write_bytes": w * KIB}
    total += (fc + w) * KIB
if k1 and complete and bench:
    cfg = bench["config"]
    W, nseq, L, strand = cfg["W"], cfg["n_seq_per_gpu"], cfg["seq_len"], cfg["strand"]
    key = "W%d_n%d_L%d_%s" % (W, nseq, L, strand)
    t = {
        "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/profile_round.sh), per pengk_count_bg launch; "
                "FETCH_SIZE doubled for the 16-B-per-lane streaming readers per MI355X_MICROARCH.md (gfx950 reports 1/2)",
        "commit": commit, "config": cfg["workload"].split(";")[0], "profile": tag, "kernels": k1,
        "count_kernel_hbm_bytes_per_launch": total,
        "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
    }
    scan = next((n for n in agg if n.startswith("pengk::count_scatter_kernel") or n.startswith("pengk::count_scatter12_kernel")), None)
    if scan:
        valu, salu, lds = (mean(scan, c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"))
        busy, act_valu = mean(scan, "SQ_BUSY_CYCLES"), mean(scan, "SQ_ACTIVE_INST_VALU")
        ms = kernel_ms(scan.split("::")[1].split("<")[0])
        if None not in (valu, salu, lds, busy, act_valu, ms):
            n_simd, clk = 1024, 2.4e9
            peak4 = n_simd * clk / 4.0  # measured class rate of this kernel's instructions (tools/ubench/valu_rate2.hip)
            peak2 = n_simd * clk / 2.0  # the guide's nominal SIMD rate (MI355X_MICROARCH.md: 2 cycles per wave64 instruction)
            cycles = busy / 32.0        # SQ_BUSY_CYCLES is summed over the 32 shader engines
            steps = cfg["ltot_global"] / 64.0
            t["issue"] = {
                "kernel": scan.replace("pengk::", "") + " (the scan of K1)", "bound": "instruction issue (VALU)",
                "achieved": valu / (ms * 1e-3), "peak": peak4, "unit": "VALU wave-instructions/s", "frac": valu / (ms * 1e-3) / peak4,
                "peak_nominal_2_cycle": peak2, "frac_of_nominal_2_cycle_rate": valu / (ms * 1e-3) / peak2,
                "valu_busy_frac": act_valu * 4.0 / (n_simd * cycles),
                "per_launch": {"valu_wave_instructions": valu, "salu_instructions": salu, "lds_instructions": lds,
                               "kernel_ms": ms, "busy_cycles_per_shader_engine": cycles,
                               "wave_steps_of_64_windows": steps, "valu_per_wave_step": valu / steps,
                               "all_instructions_per_wave_step": (valu + salu + lds) / steps},
                "source": "rocprofv3 --pmc SQ_* passes at commit %s (%s), not this run" % (commit, tag),
                "note": "`peak` prices a wave-instruction at 4 cycles per SIMD -- the rate tools/ubench/valu_rate2 MEASURES on this chip for "
                        "the classes this kernel is made of (v_bfe / v_min / v_cmpx / v_mad_u32_u24, every VOP3 / SDWA / DPP form: 4.2-4.6 "
                        "cycles); the guide's nominal SIMD rate is 2 cycles (MI355X_MICROARCH.md), `frac_of_nominal_2_cycle_rate` is against "
                        "that.  The byte roofline cannot describe this kernel: it moves many times its algorithmic bytes and still uses "
                        "a fraction of the HBM rate",
            }
    path = os.path.join(prof, "traffic_by_config.json")
    allc = json.load(open(path)) if os.path.exists(path) else {}
    allc[key] = t
    # K2+K3: the sweep kernel of the same profile (a thread reads 4 B of count per lane, uncalibrated width: raw FETCH_SIZE)
    for name in sorted(agg):  # (stats_pair_kernel -- both strands from W = 12 on, the kernel the step runs -- comes second and wins)
        if name.startswith("pengk::stats_kernel<") or name.startswith("pengk::stats_pair_kernel<"):
            f, w = agg[name].get("FETCH_SIZE"), agg[name].get("WRITE_SIZE")
            if f and w:
                f, w = sum(f) / len(f) * KIB, sum(w) / len(w) * KIB
                Ws = int(name.split("<")[1].split(">")[0])
                allc["sweep_W%d" % Ws] = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of %s per launch (raw values)" % name.replace("pengk::", ""), "commit": commit,
                                          "profile": tag, "fetch_bytes_raw": f, "write_bytes": w, "hbm_bytes_per_launch": f + w,
                                          "algorithmic_bytes_per_launch": 28 * 4 ** Ws}
    json.dump(allc, open(path, "w"), indent=1)
    print("traffic entry", key, "%.3f GB per launch" % (total / 1e9))
print("\n".join(lines[:60]))
