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
    if "walk_coop_kernel" in name:   # walk_coop_kernel<STEPS, LATE>: the two-wavefront walk (round 6)
        return "walk_coop_kernel"
    if "walk_reg_big_kernel" in name:  # walk_reg_big_kernel<METRIC, STEPS, OFF32, RETRY, AUX>
        args = name[name.find("<") + 1:name.rfind(">")].replace(" ", "").split(",")
        return "walk_reg_big_kernel" + ("/retry" if len(args) > 3 and args[3] in ("true", "1") else "")
    for key in ("walk_hot_spec_kernel", "walk_reg_wide_kernel"):
        if key in name:
            return key + (name[name.find("<"):name.find(">") + 1].replace(" ", "") if "<" in name and "wide" in key else "")
    for key in ("walk_hot_dot_big_kernel", "walk_hot_dot_kernel", "walk_hotw_big_kernel", "walk_hotw2_kernel", "walk_hotw_kernel"):
        if key in name:  # the dot-metric and wide-row hot instances (template arguments kept: R, WIDE)
            return key + (name[name.find("<"):name.find(">") + 1].replace(" ", "") if "<" in name and "dot" in key else "")
    for key in ("walk_bitmap_big_kernel", "walk_hot_big_kernel", "walk_hot2_kernel", "mlp_net_kernel", "mlp_fused_kernel", "walk_bitmap_reg_kernel", "walk_bitmap_kernel",
                "mlp_narrow_kernel", "gd_prune_kernel", "knn_scan_kernel"):
        if key in name:
            return key
    if "walk_hotN_kernel" in name:
        return "walk_hotN_kernel<" + name[name.find("<") + 1:name.find(">")] + ">"
    for key in ("walk_hot_kernel", "walk_general_kernel", "rerank_pair_kernel", "rerank_kernel", "mlp_mfma_net_kernel", "mlp_mfma_pack_kernel", "mlp_slab_kernel", "mlp_layer_mfma_kernel", "mlp_layer_vec_kernel", "mlp_layer_kernel",
                "normalize_kernel", "fill_u32_kernel", "order_hist_kernel", "order_scan_kernel", "order_scatter_kernel"):
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
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--last", type=int, default=10, help="dispatches per kernel that belong to the timed region (= bench --steps)")
    ap.add_argument("--config", default=None, help="bench configuration: also writes an entry of profiles/counters_latest.json")
    ap.add_argument("--tag", default="r00")
    a = ap.parse_args()
    out, last = a.out, a.last
    res = {}
    print("== kernel trace (", out, ") ==")
    stats = defaultdict(list)
    rows = kernel_trace(os.path.join(out, "trace"))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        stats[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("%-28s %6s %11s %10s %10s %10s | timed launches only (last %d): %8s %8s %8s" % (
        "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", last, "avg_us", "min_us", "max_us"))
    ours = lambda k: any(x in k for x in ("walk_", "mlp_", "rerank", "normalize", "knn_scan", "gd_prune", "fill_u32", "order_", "gbnns"))
    for k, v in sorted(stats.items(), key=lambda kv: -sum(kv[1][-last:])):
        if not ours(k):
            continue
        t = v[-last:]
        print("%-28s %6d %11.1f %10.2f %10.2f %10.2f | %39s %8.2f %8.2f %8.2f" % (
            k, len(v), sum(v), sum(v) / len(v), min(v), max(v), "", sum(t) / len(t), min(t), max(t)))
        res[k] = dict(calls=len(v), avg_us=sum(v) / len(v), timed_avg_us=sum(t) / len(t), timed_total_us=sum(t))
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
        print("\n== counters:", os.path.basename(sub), "(mean per dispatch over the last %d dispatches) ==" % last)
        for k in sorted(acc):
            if not any(x in k for x in ("walk_", "mlp_", "rerank", "normalize", "knn_scan", "gd_prune", "gbnns")):
                continue
            line = []
            for c, vals in sorted(acc[k].items()):
                tail = vals[-last:]
                m = sum(tail) / len(tail)
                line.append("%s=%.4g" % (c, m))
                res.setdefault(k, {})[c] = m
            print("%-24s %s" % (k, "  ".join(line)))
    walks = [k for k in res if k.startswith("walk_") and "general" not in k and "timed_total_us" in res[k]]
    dom = max(walks, key=lambda k: res[k]["timed_total_us"]) if walks else "walk_fast_kernel"
    try:  # the kernel bench.py's own line names (gbnns_profile.walk_kernel): other efs' walks of the recall sweep may be longer
        line = [l for l in open(os.path.join(out, "bench_plain.json")) if l.startswith("{")][-1]
        named = json.loads(line)["roofline"]["kernel"].split(" (")[0]
        if short(named) in res:
            dom = short(named)
    except Exception:
        pass
    w = res.get(dom, {})
    print("\ndominant kernel of the timed region:", dom, "-- %.2f us per timed launch under the profiler" % w.get("timed_avg_us", 0.0))
    if "FETCH_SIZE" in w:
        fetch_kib, write_kib = w["FETCH_SIZE"], w.get("WRITE_SIZE", 0.0)
        raw = (fetch_kib + write_kib) * 1024
        corr = (2 * fetch_kib + write_kib) * 1024
        print("\n%s HBM bytes per launch: raw %.4g B (FETCH+WRITE), fetch-x2-corrected %.4g B" % (dom, raw, corr))
        res["dominant_kernel"] = dom
        res["walk_fast_hbm_bytes_per_launch_raw"] = raw
        res["walk_fast_hbm_bytes_per_launch"] = corr
    if "SQ_INSTS_VALU" in w:
        # second roofline: a VALU instruction holds its SIMD's issue port for 4 cycles (MI355X_MICROARCH.md); 1024 SIMDs
        cyc = w.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # the counter is summed over the 8 XCDs
        waves = max(w.get("SQ_WAVES", 1.0), 1.0)
        print("%s VALU issue: %.4g instructions per launch = %.0f per wavefront; kernel %.4g cycles (GRBM_GUI_ACTIVE / 8)"
              % (dom, w["SQ_INSTS_VALU"], w["SQ_INSTS_VALU"] / waves, cyc))
        if cyc > 0:
            print("  VALU issue-port occupancy = insts x 4 / (1024 SIMDs x cycles) = %.3f" % (w["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cyc)))
        if w.get("SQ_WAVE_CYCLES"):
            print("  per-wavefront life: active VALU %.3f, any instruction %.3f, waiting (s_waitcnt) %.3f, issue stall %.3f of SQ_WAVE_CYCLES"
                  % (w.get("SQ_ACTIVE_INST_VALU", 0) / w["SQ_WAVE_CYCLES"], w.get("SQ_ACTIVE_INST_ANY", 0) / w["SQ_WAVE_CYCLES"],
                     w.get("SQ_WAIT_ANY", 0) / w["SQ_WAVE_CYCLES"], w.get("SQ_WAIT_INST_ANY", 0) / w["SQ_WAVE_CYCLES"]))
    json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
    if a.config:
        ef = None
        try:
            ef = json.loads([l for l in open(os.path.join(out, "bench_plain.json")) if l.startswith("{")][-1])["config"]["ef"]
        except Exception:
            pass
        if ef is not None and "FETCH_SIZE" in w:
            path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "counters_latest.json")
            try:
                allc = json.load(open(path))
            except Exception:
                allc = {}
            allc["%s:ef%d" % (a.config, ef)] = {
                "source": "profiles/%s_%s_ef%d_summary.txt (rocprofv3 --pmc, separate passes, bench.py --config %s --ef %d --steps %d --no-extras --serial)"
                          % (a.tag, a.config, ef, a.config, ef, last),
                "kernel": dom, "kernel_us_profiled": w.get("timed_avg_us"),
                "FETCH_SIZE_KiB": w["FETCH_SIZE"], "WRITE_SIZE_KiB": w.get("WRITE_SIZE", 0.0),
                "hbm_bytes_raw": res["walk_fast_hbm_bytes_per_launch_raw"], "hbm_bytes_corrected": res["walk_fast_hbm_bytes_per_launch"],
                "valu_insts": w.get("SQ_INSTS_VALU"), "gpu_cycles": (w.get("GRBM_GUI_ACTIVE", 0.0) / 8.0) or None,
                "waves": w.get("SQ_WAVES"),
            }
            json.dump(allc, open(os.path.join(out, "counters_entry.json"), "w"), indent=1)
            print("\ncounters entry written to", os.path.join(out, "counters_entry.json"), "(merge into profiles/counters_latest.json)")


if __name__ == "__main__":
    main()
