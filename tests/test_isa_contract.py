"""The arithmetic contract, checked on the shipped code object (CPU test: no GPU needed).

DESIGN.md section 2: every distance the walk / re-rank / kNN / graph-pruning kernels compute is a chain of
separately rounded IEEE binary32 multiplies and adds in the reference's order (support_func.h:107-163) --
a fused multiply-add anywhere in them would change bits and, through compare-driven control flow, answers.
The compiler flags (-ffp-contract=off) promise that; this test verifies it on the gfx950 machine code inside
lib/libgbnns_hip.so: it unbundles the device code objects, disassembles them and fails on any floating-point
fused / multiply-accumulate instruction in those kernels.  It also pins one scheduling property of the
hand-laid-out walk_hot kernels that cost 10 % when it broke (tools/check_isa.sh): the row gather of a hop
must not wait for the adjacency prefetch issued just before it (no `s_waitcnt vmcnt(0)` in between).
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gbnns_dim_red_amd", "lib", "libgbnns_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

# kernels under the bit-exact distance contract (mangled-name substrings)
CONTRACT = ("walk_", "rerank_", "knn_scan", "gd_prune")
# floating-point instructions that fuse or chain a multiply with an add (integer v_mad_u32_u24 etc. are fine)
FUSED = re.compile(r"^v_(fma_|fmac_|fmaak_|fmamk_|mac_f|mad_f|madak_f|madmk_f|pk_fma_|fma_mix|mad_mix|dot\d)")


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    """{mangled kernel name: [instruction lines]} of every gfx950 code object in the library."""
    if not os.path.exists(LIB):
        pytest.fail(LIB + " is missing: run __graft_entry__.build() first")
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump of the ROCm toolchain not found")
    work = tmp_path_factory.mktemp("isa")
    lib = shutil.copy(LIB, work)  # llvm-objdump --offloading writes the bundles next to its input
    subprocess.run([OBJDUMP, "--offloading", lib], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = {}
    objs = [f for f in os.listdir(work) if f.endswith("gfx950")]
    assert objs, "no gfx950 code object inside the library"
    for f in objs:
        text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", os.path.join(work, f)], check=True,
                              capture_output=True, text=True).stdout
        name = None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                name = m.group(1)
                out.setdefault(name, [])
                continue
            if name and line.startswith("\t"):
                out[name].append(line.strip().split("//")[0].strip())
    return out


def test_no_fused_multiply_add_in_the_distance_kernels(kernels):
    checked = 0
    bad = []
    for name, insts in kernels.items():
        if not any(k in name for k in CONTRACT):
            continue
        checked += 1
        for ins in insts:
            if FUSED.match(ins):
                bad.append((name[:90], ins))
    assert checked >= 20, "expected the walk / re-rank / kNN / pruning kernels in the code objects, found %d" % checked
    assert not bad, "fused multiply-add inside a bit-exact kernel: %s" % bad[:5]


def test_distance_kernels_use_separate_mul_and_add(kernels):
    """Sanity of the check itself: the kernels do contain the separately rounded forms."""
    hot = [v for k, v in kernels.items() if "walk_hot_kernel" in k]
    assert hot, "walk_hot_kernel not found"
    text = "\n".join(hot[0])
    assert "v_pk_mul_f32" in text and "v_pk_add_f32" in text


def test_projection_kernels_fma_only_in_div_sqrt_expansion(kernels):
    """The MLP layers (support_func.h:624-633) are under the same contract; the correctly rounded divide / sqrt of
    normalizeVector (:636-642) legitimately expand to fma sequences -- only the NORM variants may contain any."""
    for name, insts in kernels.items():
        if "mlp_" not in name or "mlp_mfma_" in name:   # (mlp_mfma_*: the opt-in throughput option and its weight packer -- not under the contract)
            continue
        n_fma = sum(1 for i in insts if FUSED.match(i))
        if n_fma:
            # NORM = last template argument true (mlp_layer*_kernel<RELU, NORM>, mlp_narrow_kernel<RELU, NORM>); the one-launch
            # kernel (mlp_net_kernel) always ends with the normalise step
            assert re.search(r"Lb[01]ELb1E", name) or "mlp_net_kernel" in name, \
                "fma in a projection kernel without the normalise step: " + name
            assert n_fma < 64, (name, n_fma)  # a div + a sqrt expansion, not a dot-product loop
    # the matrix cores serve the throughput option only (mlp_mfma_net_kernel / mlp_layer_mfma_kernel, GBNNS_FLAG_MFMA_PROJECTION)
    for name, insts in kernels.items():
        if "mlp_" in name:
            if "mlp_mfma_pack_kernel" in name:
                continue
            assert any(i.startswith("v_mfma") for i in insts) == ("mlp_layer_mfma_kernel" in name or "mlp_mfma_net_kernel" in name), name
    net = [v for k, v in kernels.items() if "mlp_net_kernel" in k]
    assert net, "mlp_net_kernel not found"
    for insts in net:  # its products and sums are the separately rounded packed forms, its folds the row swaps
        text = "\n".join(insts)
        assert "v_pk_mul_f32" in text and "v_pk_add_f32" in text and "v_permlane16_swap_b32" in text and "v_permlane32_swap_b32" in text


def test_hot_kernel_gather_does_not_wait_for_the_prefetch(kernels):
    hot = {k: v for k, v in kernels.items() if re.search(r"walk_hot(2|_big)?_kernel", k)}
    assert len({k.split("walk_hot")[1][:4] for k in hot}) == 3  # the ef <= 64, ef <= 128 and ef <= 512 instances
    seen = set()
    for name, insts in hot.items():
        key = tuple(insts[:50])
        if key in seen:  # the same kernel is present in every compilation unit's code object
            continue
        seen.add(key)
        # hot_expand's block: four row loads into the fixed registers v[40:55], offsets 0 / 16 / 32 / 48
        gathers = [i for i, s in enumerate(insts[:-1]) if s.startswith("global_load_dwordx4 v[40:43]") and "offset" not in s
                   and insts[i + 1].startswith("global_load_dwordx4 v[44:47]") and "offset:16" in insts[i + 1]]
        assert gathers, name
        for gi in gathers:
            # (the ef <= 64 instances test the visited set between the prefetch and the gather since round 4: up to two
            # bucket forms of ~50 instructions each)
            back = [j for j in range(max(0, gi - 200), gi) if re.match(r"global_load_dword\s", insts[j])]
            assert back, "no adjacency prefetch in front of the gather in " + name
            between = insts[back[-1] + 1:gi]
            assert not any(re.match(r"s_waitcnt vmcnt\(0\)", s) for s in between), \
                "the gather waits for the adjacency prefetch (vmcnt(0)) in " + name


READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def test_hot_kernel_register_budgets(tmp_path):
    """The hand-laid-out walk kernels claim fixed VGPRs (v36 .. v63; csrc/walk_hot.hip, hot_expand) beside the compiler's own.
    Their occupancy rests on BOTH register files (measured on the hardware, tools/ubench/occupancy_census.hip: a
    wavefront's scalar registers are handed out as ceil16(sgpr_count) + 16 of 800 per SIMD, so <= 80 -> 8 wavefronts per
    SIMD, <= 96 -> 7, above -> 6; vector registers: <= 64 -> 8, <= 72 -> 7, <= 80 -> 6, <= 96 -> 5) and on nothing
    spilling.  Rounds 1-3 shipped these kernels at 106 scalar registers -- an inline-asm clobber of s98 / s99 -- i.e. at
    6 wavefronts per SIMD whatever the vector-register count said.  Only the one-register instances (ef <= 64) are held
    to the 8-wavefront class: the others are bounded by their LDS share (<= 22 wavefronts per CU) well before.  This test reads the kernel descriptors' metadata
    of the shipped code objects and fails when an instance leaves its class."""
    if not os.path.exists(READELF) or not os.path.exists(OBJDUMP):
        pytest.skip("ROCm llvm tools not found")
    lib = shutil.copy(LIB, tmp_path)
    subprocess.run([OBJDUMP, "--offloading", lib], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    meta = {}
    for f in os.listdir(tmp_path):
        if not f.endswith("gfx950"):
            continue
        text = subprocess.run([READELF, "--notes", os.path.join(tmp_path, f)], check=True, capture_output=True, text=True).stdout
        for block in text.split("  - .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", block)
            if not name:
                continue
            meta[name.group(1)] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, block).group(1))
                                   for k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "vgpr_spill_count")
                                   if re.search(r"\.%s:\s+(\d+)" % k, block)}
    # (substring of the mangled name, VGPR ceiling, SGPR ceiling)
    budgets = [("15walk_hot_kernelE", 64, 80), ("20walk_hot_spec_kernelE", 64, 80), ("16walk_hot2_kernelE", 72, 112), ("19walk_hot_big_kernelE", 96, 112),
               # the instances for adjacency rows of 33 .. 64 slots (second expansion pass)
               ("16walk_hotw_kernelE", 64, 80), ("17walk_hotw2_kernelE", 72, 112), ("20walk_hotw_big_kernelE", 96, 112),
               # the negative-dot metric on the same shapes
               ("19walk_hot_dot_kernelILi1E", 64, 80), ("19walk_hot_dot_kernelILi2ELb0E", 72, 112), ("23walk_hot_dot_big_kernelI", 96, 112),
               # (its two-pass instance with the stash: 73 registers = six wavefronts per SIMD, 24 per CU -- the two-register lists' LDS allows 21)
               ("19walk_hot_dot_kernelILi2ELb1E", 80, 112),
               # the generic hop over 192-byte rows at ef <= 64 (the reference's deep 96 -> 48 shape): query in LDS, 6 wavefronts per SIMD
               ("20walk_reg_wide_kernelILi12E", 80, 96),
               # the pair-form list instances over 384-byte rows (PLAIN walks over deep vectors at ef <= 128): three wavefronts per SIMD
               ("15walk_reg_kernelILi0ELi24E", 136, 112)]
    for sub, cap, scap in budgets:
        hits = {k: v for k, v in meta.items() if sub in k}
        assert hits, sub
        for k, v in hits.items():
            assert v["vgpr_count"] <= cap, (k, v)
            assert v["sgpr_count"] <= scap, (k, v)
            assert v["private_segment_fixed_size"] == 0 and v.get("vgpr_spill_count", 0) == 0, (k, v)
    # the one-launch projection: two wavefronts per SIMD (<= 256 registers), nothing spilled (a spill inside its k loop is
    # what the first builds of it lost 20 % to)
    hits = {k: v for k, v in meta.items() if "mlp_net_kernel" in k}
    assert hits
    for k, v in hits.items():
        assert v["vgpr_count"] <= 256 and v["private_segment_fixed_size"] == 0 and v.get("vgpr_spill_count", 0) == 0, (k, v)
    # the slab kernel for single layers: the same two wavefronts per SIMD in its wide shape (8 wavefronts x 8 A queries, A <= 5),
    # and at least as many in the narrow ones (<= 192 registers; 140 in the 1 000 x 64-neuron shape), nothing spilled
    hits = {k: v for k, v in meta.items() if "mlp_slab_kernel" in k}
    assert len(hits) >= 12, sorted(hits)
    for k, v in hits.items():
        assert v["private_segment_fixed_size"] == 0 and v.get("vgpr_spill_count", 0) == 0, (k, v)
        assert v["vgpr_count"] <= (256 if "mlp_slab_kernelILi8E" in k else 192), (k, v)
    # no bit-exact distance kernel may spill to scratch at all (a spill in a latency chain is a hidden HBM round trip),
    # and the generic two-list / bitmap walks stay within 3 wavefronts per SIMD (<= 168 registers)
    for k, v in meta.items():
        if any(c in k for c in ("walk_", "rerank_")):
            assert v["private_segment_fixed_size"] == 0 and v.get("vgpr_spill_count", 0) == 0, (k, v)
            if "walk_general" not in k and "walk_fast" not in k:
                # (the pair form over 384- / 512- / 576-byte rows -- Li24E / Li32E / Li36E -- holds 2 x 12 .. 18 sixteen-byte steps of row and
                # query per lane: two wavefronts per SIMD by design, one for the auxiliary-graph instances)
                wide = re.search(r"walk_(reg|bitmap)_big_kernelILi0ELi(24|32|36)E", k) is not None
                # (the two-wavefront walk for small batches -- walk_coop.hip -- runs at most two wavefronts per SIMD by design: workgroups of
                # two wavefronts, at most four of them per CU; its re-rank keeps 24 sixteen-byte loads in flight per lane)
                wide = wide or "walk_coop_kernel" in k
                assert v["vgpr_count"] <= (264 if wide else 176), (k, v)
