#!/usr/bin/env python3
"""Diagnostic: per-segment cycle shares of the walk hop loop (needs a `make STAMPS=1` build)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gbnns_dim_red_amd as g
from gbnns_dim_red_amd import synth
ds = synth.make_dataset(device="cuda:0", cache_dir=os.environ.get("GBNNS_CACHE", "/tmp/gbnns_cache"))
ix = ds.index()
lib = g.load_library()
lib.gbnns_debug_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
names = ["select", "row wait", "prefetch issue", "visited (hash)", "gather+dist", "inserts", "TOTAL wave life"]
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
    print(f"   merges {h[8]/hops:.3f}/hop  merge fallbacks {h[9]/hops:.4f}/hop  sequential offers {h[10]/hops:.3f}/hop  "
          f"fast selects {h[11]/hops:.3f}/hop  probe iterations {h[12]/hops:.3f}/hop")
