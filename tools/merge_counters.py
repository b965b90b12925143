#!/usr/bin/env python3
"""Copies the digests of tools/profile_round.sh runs (gpurun_out/prof_<tag>_<config>_ef<ef>/summary.txt) into
profiles/<tag>_<config>_ef<real ef>_summary.txt and merges each run's entry of counters_entry.json
(tools/digest_profile.py) into profiles/counters_latest.json.  python tools/merge_counters.py r03"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
path = os.path.join(ROOT, "profiles", "counters_latest.json")
allc = json.load(open(path))
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "prof_%s_*" % tag))):
    name = os.path.basename(d)[len("prof_") + len(tag) + 1:]          # <config>_ef<ef>
    cfg = name.rsplit("_ef", 1)[0]
    summ = os.path.join(d, "summary.txt")
    if not os.path.exists(summ) or os.path.getsize(summ) < 200:
        print("skipped (no digest):", d)
        continue
    try:
        line = [l for l in open(os.path.join(d, "bench_plain.json")) if l.startswith("{")][-1]
        ef = json.loads(line)["config"]["ef"]                          # ef 0 on the command line = the configuration's own / gate ef
    except Exception:
        ef = int(name.rsplit("_ef", 1)[1])
    key = "%s:ef%d" % (cfg, ef)
    ent = os.path.join(d, "counters_entry.json")
    if os.path.exists(ent):
        e = json.load(open(ent))
        if key in e:
            allc[key] = e[key]
    shutil.copy(summ, os.path.join(ROOT, "profiles", "%s_%s_ef%d_summary.txt" % (tag, cfg, ef)))
    print("merged", os.path.basename(d), "->", key)
json.dump(allc, open(path, "w"), indent=1)
