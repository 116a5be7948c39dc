"""Static instruction statistics of one kernel of a built library (development aid): total instructions, s_nop, VALU / SALU / memory split.
    python3 tools/isa_stats.py <libembree3.so> "<substring of the demangled kernel name>" [--dump out.s]"""
import os, re, subprocess, sys, tempfile, collections
LLVM = "/opt/rocm/lib/llvm/bin"
lib, pat = sys.argv[1], sys.argv[2]
with tempfile.TemporaryDirectory() as td:
    fat = os.path.join(td, "fat")
    subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(td, "copy.so")])
    blob = open(fat, "rb").read()
    M = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(M), blob)]
    for i, s in enumerate(starts):
        part = os.path.join(td, "b%d" % i)
        open(part, "wb").write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = os.path.join(td, "co%d.elf" % i)
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + part, "--output=" + co])
        txt = subprocess.run([LLVM + "/llvm-objdump", "-d", "--demangle", co], capture_output=True, text=True).stdout
        cur, body = None, collections.defaultdict(list)
        for line in txt.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
            if m:
                cur = m.group(1)
            elif cur and line.startswith("\t"):
                body[cur].append(line.strip().split()[0])
        for k, ops in body.items():
            if pat in k:
                c = collections.Counter(ops)
                valu = sum(v for o, v in c.items() if o.startswith("v_"))
                salu = sum(v for o, v in c.items() if o.startswith("s_") and not o.startswith("s_nop") and not o.startswith("s_waitcnt"))
                mem = sum(v for o, v in c.items() if o.startswith(("global_", "ds_", "flat_", "scratch_", "buffer_", "s_load")))
                print("%s\n  total %d  VALU %d  SALU %d  memory/LDS %d  s_nop %d  s_waitcnt %d" % (k.replace("rtamd::dev::", "")[:140], len(ops), valu, salu, mem, c["s_nop"], c["s_waitcnt"]))
                if "--dump" in sys.argv:
                    open(sys.argv[sys.argv.index("--dump") + 1], "w").write(txt)
