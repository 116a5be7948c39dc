"""CPU-side build check (no GPU): register / scratch budget of the traversal kernels, read from the code-object metadata of
the built library.  Scratch is a PERFORMANCE cliff, not a correctness one: round 2 re-ran the instrumented twins and the whole
GPU suite on builds that spill 76-180 bytes per lane (profiles/r02_twin_scratch_check.txt) with byte-identical results, but
forcing the metric kernel to spill costs 12-48 %.  A compiler bump or a few more live registers in the
metric path must therefore not go unnoticed."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_metadata import kernel_metadata  # noqa: E402

LIB = os.path.join(ROOT, "embree-compressed_amd", "lib", "libembree3.so")


def test_no_scratch_in_the_plain_traversal_kernels():
    md = kernel_metadata(LIB)
    plain = {n: r for n, r in md.items() if re.match(r"trace_(pool_)?kernel<", n)}
    assert len(plain) >= 200, "expected every leaf policy x robust x occluded x counted x vec instantiation"
    # the instrumented twins (4th template argument true) may use what they like; every kernel a product call can reach
    # (COUNT == false) must not spill, with one documented exception (grid-mode cBVH walk at C = 3: 12 bytes, 2 registers)
    # (the C = 3 cBVH walks sit at the 168-VGPR limit of three waves per SIMD: the allocator parks one or two cold tuples - e.g. values
    # that are live across a blob visit but not used in it - in at most 32 bytes; test_metric_kernel_register_budget bounds the
    # number of scratch instructions)
    allowed = {f"trace_kernel<CbvhLeaf<{m}, 3, {f}>, true, false, false, {v}>": 32 for m in (0, 1, 2) for v in ("true", "false") for f in ("true", "false")}
    bad = []
    for n, r in plain.items():
        args = [a.strip() for a in n[n.index("<") + 1 : n.rindex(">")].rsplit(",", 4)]
        counted = args[-2] == "true"
        if counted:
            continue
        if r["scratch"] > allowed.get(n, 0):  # (SGPR spills go to VGPR lanes, not to memory: the C = 4, 5 walks have 2-19 of them)
            bad.append((n, r["vgpr"], r["scratch"], r["sgpr_spills"]))
    assert not bad, "traversal kernels with scratch: %r" % bad


def test_metric_kernel_register_budget():
    metric = "trace_kernel<CbvhLeaf<1, 3, true>, true, false, false, true>"
    lane = "trace_kernel<CbvhLeaf<1, 3, false>, true, false, false, true>"
    md = kernel_metadata(LIB, disassemble=(metric, lane))
    # BASELINE.json metric path: cbvh.leaf, C = 3, closest hit, 16-byte aligned records, incoherent rays -> quad form:
    # FOUR waves per SIMD (<= 128 VGPRs, no scratch) and four workgroups per CU (LDS)
    r = md[metric]
    assert r["scratch"] == 0 and r["vgpr"] <= 128 and 4 * r["lds"] <= 160 * 1024, r
    # the one-ray-per-lane form (coherent batches): three waves per SIMD, at most a cold tuple in scratch
    r = md[lane]
    assert r["scratch"] <= 32 and r["vgpr"] <= 168 and r["scratch_ops"] <= 4, r
    g = md["trace_kernel<GridCellLeaf, true, false, false, true>"]
    assert g["scratch"] == 0 and g["vgpr"] <= 128, g  # eager path: four waves per SIMD
