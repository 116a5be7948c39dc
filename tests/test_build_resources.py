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
    allowed = {"trace_kernel<CbvhLeaf<2, 3>, true, false, false, false>": 12, "trace_kernel<CbvhLeaf<2, 3>, true, false, false, true>": 12}
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
    md = kernel_metadata(LIB)
    # BASELINE.json metric path: cbvh.leaf, C = 3, closest hit, 16-byte aligned records
    r = md["trace_kernel<CbvhLeaf<1, 3>, true, false, false, true>"]
    assert r["scratch"] == 0 and r["vgpr"] <= 168, r  # three waves per SIMD
    g = md["trace_kernel<GridCellLeaf, true, false, false, true>"]
    assert g["scratch"] == 0 and g["vgpr"] <= 128, g  # eager path: four waves per SIMD
