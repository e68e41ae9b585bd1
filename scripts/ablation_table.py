#!/usr/bin/env python3
"""profiles/<name>_ablation.md from gpurun_out/<tag>_ablation.jsonl (scripts/dbg/ablation.sh) and the per-kernel HBM traffic
of the round's committed profiles (profiles/<f32 name>_pmc_traffic.json, <bf16 name>_pmc_traffic.json):
BASELINE.json configs[4] -- "multi-resolution STFT loss on/off + PCEN feature on/off (isolates rFFT-kernel HBM fraction)".
usage: python scripts/ablation_table.py <tag> <out name> <f32 profile name> <bf16 profile name>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FFT_KERNELS = ("stft_features_kernel", "pcen_scan_kernel", "pcen_pow_kernel", "mask_istft_frames_kernel", "ola_kernel",
               "mask_istft_bwd_kernel", "stft_loss_fwd_kernel", "stft_loss_fwdgrad_kernel", "stft_bwd_kernel",
               "ola_gather_kernel", "loss_grad_gather_kernel", "loss_finalize_kernel", "l1_grad_kernel", "reduce_cols_kernel")


def main():
    tag, name, pf32, pbf16 = sys.argv[1:5]
    rows = [json.loads(l) for l in open(os.path.join(ROOT, "gpurun_out", tag + "_ablation.jsonl")) if l.strip()]
    with open(os.path.join(ROOT, "profiles", name + "_ablation.md"), "w") as f:
        f.write("# BASELINE.json configs[4]: MR-STFT loss on/off x PCEN on/off (1x MI355X, 64 x 4 s per GPU, final code of the round)\n\n")
        f.write("`bash scripts/dbg/ablation.sh %s` (bench.py --steps 20 --warmup 3 per line; same box for all eight lines).\n\n" % tag)
        f.write("| dtype | MR-STFT loss | PCEN (C_in) | ms / step | frames/s | delta vs full |\n|---|---|---|---:|---:|---:|\n")
        full = {}
        for r in rows:
            if "error" in r:
                f.write("| %s | | | failed | | |\n" % r["error"])
                continue
            w = r["config"]["workload"]
            stft = "off" if "WITHOUT MR-STFT" in w else "on"
            pcen = "on (4)" if "C_in=4" in w else "off (3)"
            key = r["dtype"]
            if stft == "on" and pcen.startswith("on"):
                full[key] = r["ms_per_step"]
            d = r["ms_per_step"] - full.get(key, r["ms_per_step"])
            f.write("| %s | %s | %s | %.3f | %.0f | %+.3f ms |\n" % (key, stft, pcen, r["ms_per_step"], r["value"], d))
        for dt, pn in (("f32", pf32), ("bf16", pbf16)):
            fn = os.path.join(ROOT, "profiles", pn + "_pmc_traffic.json")
            if not os.path.exists(fn):
                continue
            doc = json.load(open(fn))
            tot = doc["hbm_bytes_per_step"]
            f.write("\n## rFFT-class kernels of the %s step: HBM bytes per step (PMC, 2 x FETCH_SIZE + WRITE_SIZE; `%s_pmc_traffic.json`)\n\n" % (dt, pn))
            f.write("| kernel | launches / step | MB / launch | MB / step | of the step's HBM bytes |\n|---|---:|---:|---:|---:|\n")
            acc = 0.0
            for k, v in doc["kernels"].items():
                if not k.startswith(FFT_KERNELS):
                    continue
                per_step = v["hbm_bytes_per_launch_corrected"] * v["launches"] / 3.0
                acc += per_step
                f.write("| `%s` | %.1f | %.1f | %.1f | %.2f %% |\n" % (k, v["launches"] / 3.0, v["hbm_bytes_per_launch_corrected"] / 1e6,
                                                                       per_step / 1e6, 100 * per_step / tot))
            f.write("| **all FFT front end / back end / loss kernels** | | | **%.1f** | **%.2f %%** of %.1f GB |\n" % (
                acc / 1e6, 100 * acc / tot, tot / 1e9))
    print(open(os.path.join(ROOT, "profiles", name + "_ablation.md")).read())


if __name__ == "__main__":
    main()
