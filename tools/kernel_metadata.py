"""Per-kernel resource metadata (VGPRs, SGPRs, scratch, spills, LDS) of every gfx950 kernel inside a built shared library.
Reads the code objects out of the .hip_fatbin section (no GPU needed).  Used by tests/test_build_resources.py and by hand:
  python3 tools/kernel_metadata.py embree-compressed_amd/lib/libembree3.so [regex]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [o.replace("rtamd::dev::", "").replace("(rtamd::LaunchParams)", "").replace("void ", "") for o in out[: len(names)]]


def kernel_metadata(lib_path, disassemble=()):
    """-> {demangled kernel name: {vgpr, sgpr, scratch, sgpr_spills, vgpr_spills, lds, max_wg}}; for the kernels whose demangled
    names are listed in `disassemble` also "scratch_ops": number of scratch_load / scratch_store instructions in the ISA."""
    res = {}
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fatbin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib_path, os.path.join(td, "copy.so")])
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, s in enumerate(starts):  # one bundle per .hip translation unit
            part = os.path.join(td, "bundle%d" % i)
            open(part, "wb").write(blob[s : starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(td, "co%d.elf" % i)
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                   "--input=" + part, "--output=" + co])
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            cur = {}
            entries = []
            for line in notes.split("\n"):
                m = re.match(r"\s*-?\s*\.(\w+):\s+(\S+)\s*$", line)
                if not m:
                    continue
                k, val = m.group(1), m.group(2)
                if k in ("group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count",
                         "max_flat_workgroup_size", "agpr_count"):
                    cur[k] = int(val)
                elif k == "name" and val.startswith("_Z"):
                    cur["name"] = val
                elif k == "wavefront_size":  # last key of a kernel's (alphabetically sorted) metadata map
                    entries.append(cur)
                    cur = {}
            names = _demangle([e.get("name", "?") for e in entries])
            for e, n in zip(entries, names):
                ops = None
                if n in disassemble:
                    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--disassemble-symbols=" + e["name"], co], capture_output=True, text=True).stdout
                    ops = len(re.findall(r"\bscratch_(?:load|store)", dis))
                res[n] = dict(scratch_ops=ops, vgpr=e.get("vgpr_count", 0), agpr=e.get("agpr_count", 0), sgpr=e.get("sgpr_count", 0), scratch=e.get("private_segment_fixed_size", 0),
                              sgpr_spills=e.get("sgpr_spill_count", 0), vgpr_spills=e.get("vgpr_spill_count", 0), lds=e.get("group_segment_fixed_size", 0),
                              max_wg=e.get("max_flat_workgroup_size", 0))
    return res


if __name__ == "__main__":
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    for n, r in sorted(kernel_metadata(sys.argv[1]).items()):
        if re.search(pat, n):
            print("%-90s vgpr %3d sgpr %3d scratch %4d spills s%d/v%d lds %d" % (n, r["vgpr"], r["sgpr"], r["scratch"], r["sgpr_spills"], r["vgpr_spills"], r["lds"]))
