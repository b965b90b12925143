#!/usr/bin/env python3
"""Digest rocprofv3 CSV output (tools/profile_round.sh) into a short text summary:
per-kernel time statistics from the kernel trace, and per-kernel means of every PMC counter.
HBM traffic: FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 64 B per
128-B request for wide coalesced streams (MI355X_MICROARCH.md, HBM section) -- the raw value and
the x2-corrected value are both printed; the walk's 128-B row gathers are uncalibrated, so the
raw number is a lower bound and the corrected one an upper bound of the HBM read bytes."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    for key, idx in (("walk_reg_kernel", 3), ("walk_fast_kernel", 2)):
        if key in name:
            # template argument `idx` = RETRY (the persistent second pass over handed-over queries);
            # walk_reg_kernel<METRIC, STEPS, OFF32, RETRY, R>, walk_fast_kernel<METRIC, STEPS, RETRY>
            args = name[name.find("<") + 1:name.rfind(">")].replace(" ", "").split(",")
            retry = len(args) > idx and args[idx] in ("true", "1")
            regs = "/R%s" % args[4] if key == "walk_reg_kernel" and len(args) > 4 and args[4] != "1" else ""
            return key + regs + ("/retry" if retry else "")
    for key in ("walk_hot2_kernel", "walk_hot3_kernel", "walk_hot4_kernel", "walk_bitmap_reg_kernel", "walk_bitmap_kernel",
                "mlp_narrow_kernel", "gd_prune_kernel", "knn_scan_kernel"):
        if key in name:
            return key
    if "walk_hotN_kernel" in name:
        return "walk_hotN_kernel<" + name[name.find("<") + 1:name.find(">")] + ">"
    for key in ("walk_hot_kernel", "walk_general_kernel", "rerank_pair_kernel", "rerank_kernel", "mlp_layer_mfma_kernel", "mlp_layer_vec_kernel", "mlp_layer_kernel",
                "normalize_kernel", "fill_u32_kernel"):
        if key in name:
            return key
    return name[:60]


def find(root, pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


def kernel_trace(root):
    rows = []
    for f in find(root, "*kernel_trace.csv"):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    return rows


def main():
    out = sys.argv[1]
    res = {}
    print("== kernel trace (", out, ") ==")
    stats = defaultdict(list)
    for r in kernel_trace(os.path.join(out, "trace")):
        stats[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("%-24s %8s %12s %12s %12s %12s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us"))
    for k, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        print("%-24s %8d %12.1f %12.2f %12.2f %12.2f" % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v)))
        res[k] = dict(calls=len(v), avg_us=sum(v) / len(v))
    # the timed region's launches of the dominant kernel = the last 10 walk_fast dispatches at ef=64
    for sub in sorted(glob.glob(os.path.join(out, "pmc_*"))):
        if not os.path.isdir(sub):
            continue
        files = find(sub, "*counter_collection.csv")
        acc = defaultdict(lambda: defaultdict(list))
        for f in files:
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("\n== counters:", os.path.basename(sub), "(mean per dispatch over the last 10 dispatches) ==")
        for k in sorted(acc):
            line = []
            for c, vals in sorted(acc[k].items()):
                tail = vals[-10:]
                m = sum(tail) / len(tail)
                line.append("%s=%.4g" % (c, m))
                res.setdefault(k, {})[c] = m
            print("%-24s %s" % (k, "  ".join(line)))
    dom = next((k for k in ("walk_hot_kernel", "walk_reg_kernel", "walk_fast_kernel") if k in res), "walk_fast_kernel")
    w = res.get(dom, {})
    if "FETCH_SIZE" in w:
        fetch_kib, write_kib = w["FETCH_SIZE"], w.get("WRITE_SIZE", 0.0)
        raw = (fetch_kib + write_kib) * 1024
        corr = (2 * fetch_kib + write_kib) * 1024
        print("\n%s HBM bytes per launch: raw %.4g B (FETCH+WRITE), fetch-x2-corrected %.4g B" % (dom, raw, corr))
        res["dominant_kernel"] = dom
        res["walk_fast_hbm_bytes_per_launch_raw"] = raw
        res["walk_fast_hbm_bytes_per_launch"] = corr
    json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
