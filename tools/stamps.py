#!/usr/bin/env python3
"""Diagnostic: per-segment cycle shares of the walk hop loop (needs a `make STAMPS=1` build)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gbnns_dim_red_amd as g
from gbnns_dim_red_amd import synth
# CONFIG=<bench.py configuration> (default sift): same synthetic recipe as the bench
import bench
cfg = bench.CONFIGS[os.environ.get("CONFIG", "sift")]
kw = dict(n=cfg["n"], nq=cfg["nq"], d=cfg["d"], d_low=cfg["d_low"], d_hidden=cfg["d_hidden"], seed=1234)
if cfg.get("unit_norm"):
    kw["unit_norm"] = True
ds = synth.make_dataset(device="cuda:0", cache_dir=os.environ.get("GBNNS_CACHE", "/tmp/gbnns_cache"), **kw)
ix = ds.index(metric=g.METRIC_NEG_DOT if cfg.get("negdot") else g.METRIC_L2)
lib = g.load_library()
lib.gbnns_debug_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
names = ["select", "row wait", "prefetch issue", "visited (hash)", "gather+dist (hot instance: + visited)", "inserts", "TOTAL wave life"]
NQ = int(os.environ.get("NQ", len(ds.queries)))
qsub = ds.queries[:NQ].contiguous()
for ef in (int(a) for a in (sys.argv[1:] or ["64"])):
    for _ in range(3):
        r = ix.search(qsub, ef, want=("hops",))
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 32)()
    lib.gbnns_debug_read_stamps(ix._h, buf)
    r = ix.search(qsub, ef, want=("hops",))
    torch.cuda.synchronize()
    lib.gbnns_debug_read_stamps(ix._h, buf)
    hops = r["hops"].double().sum().item()
    tot = buf[6]
    print(f"nq={NQ} ef={ef}: hops/query {hops/NQ:.1f}, wave life {tot/NQ:.0f} cycles/query, {tot/hops:.0f} cycles/hop")
    for i in range(6):
        print(f"   {names[i]:16s} {buf[i]/hops:8.0f} cycles/hop  {100.0*buf[i]/tot:5.1f} %")
    h = [buf[8 + i] for i in range(13)]
    lab = ["0", "1", "2", "3", "4", "5-8", "9-16", "17+"]
    print("   survivors/hop histogram: " + "  ".join(f"{lab[i]}:{100.0*h[i]/max(sum(h[:8]),1):.1f}%" for i in range(8)))
    if buf[30]:
        print(f"   loop latch (insert end -> next select) {buf[30]/hops:.0f} cycles/hop")
    if buf[7] or buf[31]:
        print(f"   node = runner-up prediction (prefetch 1) {buf[7]/hops:.3f}/hop, = closest new survivor (prefetch 2) {buf[31]/hops:.3f}/hop")
    if any(buf[21:30]):
        print(f"   two-list structure: flush {buf[21]/hops:.0f} cycles/hop ({buf[24]/hops:.3f} flushes/hop, {buf[21]/max(buf[24],1):.0f} cycles each), "
              f"refresh_cache {buf[22]/hops:.0f} cycles/hop ({buf[25]/hops:.3f}/hop, {buf[22]/max(buf[25],1):.0f} each), eviction step {buf[23]/hops:.0f} cycles/hop "
              f"({buf[28]/hops:.3f} batch inserts/hop); base expansions {buf[26]/hops:.3f}/hop, sequential fallbacks {buf[27]/hops:.4f}/hop, slow selects {buf[29]/hops:.4f}/hop")
    print(f"   merges {h[8]/hops:.3f}/hop  merge fallbacks {h[9]/hops:.4f}/hop  sequential offers {h[10]/hops:.3f}/hop  "
          f"fast selects {h[11]/hops:.3f}/hop  probe iterations {h[12]/hops:.3f}/hop")
