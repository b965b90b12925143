#!/usr/bin/env python3
"""Probe (not a test): a stream of host batches through GBNNS_MEM_HOST + GBNNS_FLAG_DEFER_JOIN (page-locked buffers),
for a timeline under `rocprofv3 --kernel-trace --memory-copy-trace`.  python tools/host_flight_probe.py DEPTH [BATCHES]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gbnns_dim_red_amd as g
from gbnns_dim_red_amd import synth

def bind_to_gpu_node(which):
    """Restricts this process to the CPUs of the GPU's NUMA node ("local") or of another node ("remote"): page-locked
    buffers allocated afterwards come from that node's memory (first touch)."""
    p = torch.cuda.get_device_properties(0)
    bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
    nodes = sorted(int(d[4:]) for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit())
    if which == "remote":
        node = [x for x in nodes if x != node][0]
    cpus = set()
    for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    os.sched_setaffinity(0, cpus & os.sched_getaffinity(0))
    print("bound to NUMA node", node, "(GPU", bdf, ")", flush=True)


if os.environ.get("GBNNS_PROBE_NODE"):
    bind_to_gpu_node(os.environ["GBNNS_PROBE_NODE"])
depth = int(sys.argv[1])
count = int(sys.argv[2]) if len(sys.argv) > 2 else 48
g.load_library()
os.makedirs("/tmp/gbnns_cache", exist_ok=True)
ds = synth.make_dataset(n=1_000_000, nq=10_000, d=128, d_low=32, d_hidden=256, seed=1234, cache_dir="/tmp/gbnns_cache", device="cuda:0")
ix = ds.index()
nq = ds.queries.shape[0]
sets = 2 * depth
hq = [ds.queries.cpu().pin_memory()] + [synth.more_queries(ds, nq, b).cpu().pin_memory() for b in range(1, 4)]
outs = [{} for _ in range(sets)]


tc, tw = [], []


def run(n):
    for i in range(n):
        t0 = time.perf_counter()
        ix.search(hq[i % 4], 64, want=(), out=outs[i % sets], flags=g.FLAG_DEFER_JOIN, defer_depth=depth)
        t1 = time.perf_counter()
        ix.wait(sets - 1)
        tc.append((t1 - t0) * 1e6)
        tw.append((time.perf_counter() - t1) * 1e6)
    ix.wait(0)


run(24)
tc.clear(); tw.clear()
t = time.perf_counter()
run(count)
dt = time.perf_counter() - t
print("depth %d: %.2f M q/s (%.3f ms per batch)" % (depth, count * nq / dt / 1e6, dt / count * 1e3))
print("call us:", " ".join("%.0f" % x for x in tc))
print("wait us:", " ".join("%.0f" % x for x in tw))
ix.join()
torch.cuda.synchronize()
