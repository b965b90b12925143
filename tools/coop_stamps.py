#!/usr/bin/env python3
"""Diagnostic (needs a `make STAMPS=1` build in place of the library): where the two wavefronts of walk_coop_kernel spend a hop --
keeper: select / waiting for the scout / claim / insert; scout: waiting for the keeper / expansions; how often the prepared
expansion was the right node.  CONFIG=<bench.py configuration> (default gist), argv = beams."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gbnns_dim_red_amd as g
from gbnns_dim_red_amd import synth
import bench
cfg = bench.CONFIGS[os.environ.get("CONFIG", "gist")]
kw = dict(n=cfg["n"], nq=cfg["nq"], d=cfg["d"], d_low=cfg["d_low"], d_hidden=cfg["d_hidden"], seed=1234)
ds = synth.make_dataset(device="cuda:0", cache_dir=os.environ.get("GBNNS_CACHE", "/tmp/gbnns_cache"), **kw)
ix = ds.index()
WAVES = int(os.environ.get("WAVES", "2"))   # 2: keeper + scout, 3: keeper + claimer + ranger
ix.knob("coop", WAVES - 1)
lib = g.load_library()
lib.gbnns_debug_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
q = ds.queries
for ef in (int(a) for a in (sys.argv[1:] or [str(cfg["ef"])])):
    for _ in range(3):
        r = ix.search(q, ef, want=("hops",))
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 32)()
    lib.gbnns_debug_read_stamps(ix._h, buf)
    r = ix.search(q, ef, want=("hops",))
    torch.cuda.synchronize()
    lib.gbnns_debug_read_stamps(ix._h, buf)
    hops = r["hops"].double().sum().item()
    nq = len(q)
    print(f"ef={ef}: hops/query {hops/nq:.1f}; keeper walk life {buf[6]/nq:.0f} cycles/query = {buf[6]/hops:.0f} cycles/hop")
    for i, nm in enumerate(("select", "waiting for the scout (barriers 1 + 2)", "read + claim", "insert")):
        print(f"   keeper {nm:42s} {buf[i]/hops:8.0f} cycles/hop")
    print(f"   keeper set-up {buf[4]/nq:.0f} cycles/query; longest walk {buf[7]} cycles; whole life (set-up, walk, outputs, re-rank) mean {buf[5]/nq:.0f} cycles")
    if WAVES == 3 or os.environ.get("SPREAD"):
        t0 = 2**62 - buf[26]
        print(f"   keepers started over {(buf[27] - t0) / 100:.1f} us, the last one ended {(buf[28] - t0) / 100:.1f} us after the first one started (s_memrealtime); longest life {buf[29]} cycles; most workgroups alive at once {buf[31]} (left at the end: {buf[30]})")
    if WAVES == 3:
        for base, who in ((8, "claimer"), (16, "ranger ")):
            for i, nm in enumerate(("waiting for a post", "expansion of a node not prepared", "expansion ahead", "waiting for the other half (before the post)",
                                    "waiting for the other half (after the post)")):
                print(f"   {who} {nm:46s} {buf[base + i]/hops:8.0f} cycles/hop")
            print(f"   {who} prepared expansion was the node {buf[base + 5]/hops:.3f}/hop, was not {buf[base + 6]/hops:.3f}/hop; predicted without the closest new id "
                  f"{buf[base + 7]/hops:.3f}/hop; guess = closest new id {buf[25 if base == 8 else 24]/hops:.3f}/hop")
        continue
    print(f"   scout  waiting for the keeper (barrier 1)         {buf[16]/hops:8.0f} cycles/hop")
    print(f"   scout  expansion of a node not prepared            {buf[17]/hops:8.0f} cycles/hop ({buf[17]/max(buf[20],1):.0f} each)")
    print(f"   scout  expansion ahead (+ guess)                   {buf[18]/hops:8.0f} cycles/hop")
    print(f"   prepared expansion was the node: {buf[19]/hops:.3f}/hop, was not: {buf[20]/hops:.3f}/hop; no guess {buf[21]/hops:.3f}/hop; guess = closest new id {buf[22]/hops:.3f}/hop; prepared expansions aborted {buf[23]/hops:.4f}/hop")
    print(f"   scout  inside its expansions: adjacency words {buf[24]/hops:.0f}, claims {buf[25]/hops:.0f}, rows + distances {buf[26]/hops:.0f}, closest new id + its adjacency request {buf[27]/hops:.0f} cycles/hop")
    print(f"   scout  iterations 0 .. 3 began at {buf[28]/nq:.0f} / {buf[29]/nq:.0f} / {buf[30]/nq:.0f} / {buf[31]/nq:.0f} cycles after its loop started (mean over queries)")
