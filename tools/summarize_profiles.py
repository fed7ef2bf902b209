#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/collect_profiles.sh into the small files kept under profiles/.

    python tools/summarize_profiles.py <prof dir> <out dir> <tag>

  <tag>_kernel_stats_topk8_256.csv   rocprofv3 --kernel-trace --stats of the default bench command (copied as is)
  <tag>_kernel_stats_dense_256.csv   the same for the dense adaptive regime, <tag>_kernel_stats_train.csv for bench --train
  <tag>_traffic.json                 HBM bytes per launch per kernel = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE counts
                                     128-B requests at 64 B on gfx950: MI355X_MICROARCH.md, HBM section), mean over launches
  <tag>_pmc_sq.json                  SQ counters per kernel (mean per launch) + MFMA utilisation =
                                     SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES summed over the 4 SIMDs of a CU ...) -- see note
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def find(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return f[0] if f else None


def counters(d):
    """-> {kernel name: {counter: mean value per launch}}"""
    f = find(d, "*counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(list))
    if not f:
        return {}
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def short(name):
    for key, out in (("screen_ring_kernel<1", "screen_ring_kernel<1>"), ("screen_ring_kernel<0", "screen_ring_kernel<0>")):
        if key in name:
            return out
    for key in ("screen_kernel<1", "screen_kernel<0", "project16_kernel", "refine_kernel", "conv_pair16_kernel", "gather_rows_kernel",
                "aggregate_direct_kernel", "aggregate_fold_kernel", "fold_kernel", "ovf_attend_kernel", "ovf_scores_aggregate_kernel", "ovf_combine_kernel", "screen_theta_kernel", "dense_attend_kernel", "gemm32_kernel", "gemm16s_kernel",
                "score_select_kernel", "edge_softmax_topk_kernel", "thr_bias_kernel"):
        if key in name:
            return key + (">" if key.endswith(("<1", "<0")) else "")
    return None


def main():
    src, dst, tag = sys.argv[1:4]
    os.makedirs(dst, exist_ok=True)
    for sub, name in (("stats", "topk8_256"), ("dense", "dense_256"), ("md8", "adaptive_mean_degree_8_256"), ("train", "train"),
                      ("train_adaptive", "train_adaptive"), ("c512", "topk8_512"), ("c1024", "adaptive_topk16_1024")):
        f = find(os.path.join(src, sub), "*kernel_stats.csv")
        if f:
            shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats_{name}.csv"))
    fetch, write = counters(os.path.join(src, "fetch")), counters(os.path.join(src, "write"))
    traffic = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, bench.py default workload 256x256 top-k k=8, "
                       "--steps 20 --warmup 5); bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE counts 128-B "
                       "requests at 64 B (MI355X_MICROARCH.md, HBM section); gather_rows_kernel's launches rotate over 4 row sets "
                       "(833 MiB, past the 256 MiB Infinity Cache)"}
    split = {}
    for k in fetch:
        sk = short(k)
        if sk and "FETCH_SIZE" in fetch[k]:
            w = write.get(k, {}).get("WRITE_SIZE", 0.0)
            traffic[sk] = (2.0 * fetch[k]["FETCH_SIZE"] + w) * 1024.0
            split[sk] = {"fetch_bytes": 2.0 * fetch[k]["FETCH_SIZE"] * 1024.0, "write_bytes": w * 1024.0}
    traffic["split"] = split       # (round 6) reads and writes apart: fetch = 2 * FETCH_SIZE KiB, write = WRITE_SIZE KiB
    # (round 6) the dense regime's and the training path's kernels from their own passes (same corrections; keys carry the regime)
    for regime, keys in (("dense", ("dense_attend_kernel", "project16_kernel", "screen_ring_kernel<1>")), ("train", ("gemm16s_kernel", "gemm32_kernel"))):
        fe, wr = counters(os.path.join(src, regime + "_fetch")), counters(os.path.join(src, regime + "_write"))
        for k in fe:
            sk = short(k)
            if sk in keys and "FETCH_SIZE" in fe[k]:
                w = wr.get(k, {}).get("WRITE_SIZE", 0.0)
                key = f"{regime}:{sk}"
                if key in split:        # several template instances of one kernel: keep the larger (the dominant launch shape)
                    if 2.0 * fe[k]["FETCH_SIZE"] * 1024.0 + w * 1024.0 <= traffic[key]:
                        continue
                traffic[key] = (2.0 * fe[k]["FETCH_SIZE"] + w) * 1024.0
                split[key] = {"fetch_bytes": 2.0 * fe[k]["FETCH_SIZE"] * 1024.0, "write_bytes": w * 1024.0, "kernel": k[:120]}
    json.dump(traffic, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
    durations = {}
    f = find(os.path.join(src, "stats"), "*kernel_stats.csv")
    if f:
        for row in csv.DictReader(open(f)):
            sk = short(row["Name"])
            if sk:
                durations[sk] = float(row["AverageNs"]) / 1e3
    sq = counters(os.path.join(src, "sq"))
    sq2 = counters(os.path.join(src, "sq2"))
    out = {"note": "mean per launch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES counts "
                   "cycles (= 32 x the number of v_mfma_f32_32x32x16 issued, summed over all SIMDs); GRBM_GUI_ACTIVE is summed over the 8 "
                   "XCDs (clock_ghz = GRBM_GUI_ACTIVE / 8 / kernel duration); mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x "
                   "GRBM_GUI_ACTIVE / 8) = fraction of the cycles the matrix pipes are busy AT THE CLOCK THE KERNEL RAN AT; "
                   "lds_util = SQ_LDS_IDX_ACTIVE / (256 CUs x GRBM_GUI_ACTIVE / 8).  CAUTION: the shader clock read inside the "
                   "kernels (s_memtime against s_memrealtime: profiles/r03_screen_ring_clock_and_factors.log, bench.py roofline.sustained) is "
                   "1.8-1.9 GHz under these streams, LOWER than clock_ghz here: mfma_util / lds_util are understated by that ratio"}
    for k in sq:
        sk = short(k)
        if not sk:
            continue
        e = dict(sq[k])
        e.update(sq2.get(k, {}))
        if e.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in e:
            cyc = e["GRBM_GUI_ACTIVE"] / 8.0
            e["mfma_util"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
            if "SQ_LDS_IDX_ACTIVE" in e:
                e["lds_util"] = e["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc)
            if sk in durations:
                e["avg_duration_us"] = durations[sk]
                if durations[sk] >= 40.0:                  # (GRBM_GUI_ACTIVE of a short kernel is dominated by its ramp)
                    e["clock_ghz"] = cyc / (durations[sk] * 1e3)
                else:
                    e.pop("mfma_util", None); e.pop("lds_util", None)
        out[sk] = e
    json.dump(out, open(os.path.join(dst, f"{tag}_pmc_sq.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
