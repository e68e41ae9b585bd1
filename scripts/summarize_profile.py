#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (scripts/collect_profile.sh) into the committed summaries under profiles/:
  profiles/<name>_kernel_stats.csv   rocprofv3 --stats per-kernel table (verbatim)
  profiles/<name>_summary.md         the same as a table + step total
  profiles/<name>_pmc_traffic.json   HBM bytes per launch per kernel from the FETCH_SIZE / WRITE_SIZE passes,
                                     hbm = (2*FETCH_SIZE + WRITE_SIZE) KiB (MI355X_MICROARCH.md: gfx950 FETCH_SIZE
                                     tallies wide streaming reads at half their bytes)
usage: python scripts/summarize_profile.py <tag> <name> "<title>"
"""
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return re.sub(r"\s+", " ", name)


def counters(path, counter):
    out = {}
    f = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        return out
    for row in csv.DictReader(open(f[0])):
        if row["Counter_Name"] != counter:
            continue
        k = re.sub(r"\(.*$", "", short(row["Kernel_Name"]).replace("void ", ""))
        d = out.setdefault(k, [0, 0.0])
        d[0] += 1
        d[1] += float(row["Counter_Value"])
    return out


def main():
    tag, name, title = sys.argv[1], sys.argv[2], sys.argv[3]
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)[0]
    dst = os.path.join(ROOT, "profiles", name + "_kernel_stats.csv")
    shutil.copy(stats, dst)
    rows = list(csv.DictReader(open(stats)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    steps = 5   # --steps 3 --warmup 1 + the instrumented step
    with open(os.path.join(ROOT, "profiles", name + "_summary.md"), "w") as f:
        f.write("# %s\n\n" % title)
        f.write("Command (GPU box): `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 "
                "--warmup 1 --no-cpu-baseline`\n(%d steps in the trace).  HBM traffic per kernel: `%s_pmc_traffic.json` "
                "(separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes).\n\n" % (steps, name))
        f.write("| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for r in rows:
            if float(r["Percentage"]) < 0.15:
                continue
            f.write("| `%s` | %s | %.3f | %.1f | %.1f |\n" % (short(r["Name"])[:100], r["Calls"],
                                                          float(r["TotalDurationNs"]) / 1e6,
                                                          float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
        f.write("\nTotal kernel time %.1f ms over %d steps = %.1f ms/step.\n" % (tot / 1e6, steps, tot / 1e6 / steps))
    fe, wr = counters(os.path.join(src, "fetch"), "FETCH_SIZE"), counters(os.path.join(src, "write"), "WRITE_SIZE")
    ker = {}
    for k in sorted(set(fe) | set(wr), key=lambda k: -(2 * fe.get(k, [0, 0])[1] + wr.get(k, [0, 0])[1])):
        n = max(fe.get(k, [0, 0])[0], wr.get(k, [0, 0])[0])
        fk = fe.get(k, [0, 0.0])[1] / max(n, 1)
        wk = wr.get(k, [0, 0.0])[1] / max(n, 1)
        ker[k] = {"launches": n, "FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk,
                  "hbm_bytes_per_launch_corrected": (2 * fk + wk) * 1024}
    per_step = sum(v["hbm_bytes_per_launch_corrected"] * v["launches"] for v in ker.values()) / 3  # 2 steps + instrumented
    import hashlib
    h = hashlib.sha1()
    for fn in sorted(glob.glob(os.path.join(ROOT, "tinyrecurrentunet_amd", "csrc", "*.h*")) +
                     glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(open(fn, "rb").read())
    json.dump({"csrc_hash": h.hexdigest()[:12],
               "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 1 "
                       "--warmup 1` (3 steps incl. the instrumented one); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per "
                       "MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts half of wide streaming reads)",
               "hbm_bytes_per_step": per_step, "kernels": ker},
              open(os.path.join(ROOT, "profiles", name + "_pmc_traffic.json"), "w"), indent=1)
    # matrix-pipe / wait fractions per kernel from the SQ pass (one more rocprofv3 --pmc run):
    #   mfma_pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES)   (32 SIMDs behind one SQ counter instance)
    #   wave_parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES, wave_issue_stalled = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
    sq = {}
    names = ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
             "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT")
    cs = {nm: counters(os.path.join(src, "sq"), nm) for nm in names}
    for k in cs["SQ_BUSY_CYCLES"]:
        v = {nm: cs[nm].get(k, [0, 0.0])[1] for nm in names}
        if v["SQ_BUSY_CYCLES"] <= 0 or v["SQ_WAVE_CYCLES"] <= 0:
            continue
        sq[k] = {"launches": cs["SQ_BUSY_CYCLES"][k][0],
                 "mfma_pipe_busy": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (32.0 * v["SQ_BUSY_CYCLES"]), 4),
                 "wave_parked": round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 4),
                 "wave_issue_stalled": round(v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 4),
                 "wave_issuing": round(v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"], 4),
                 "lds_bank_conflict_cycles": v["SQ_LDS_BANK_CONFLICT"]}
    if sq:
        json.dump({"note": "rocprofv3 --pmc SQ_* pass over `bench.py --steps 1 --warmup 1`; see scripts/summarize_profile.py "
                           "for the ratios", "kernels": dict(sorted(sq.items(), key=lambda kv: -kv[1]["mfma_pipe_busy"]))},
                  open(os.path.join(ROOT, "profiles", name + "_pmc_sq.json"), "w"), indent=1)
    print("step total %.1f ms, HBM %.1f GB/step" % (tot / 1e6 / steps, per_step / 1e9))


if __name__ == "__main__":
    main()
