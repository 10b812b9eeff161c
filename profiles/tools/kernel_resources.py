#!/usr/bin/env python3
"""Register / LDS / scratch figures of the gfx950 kernels in an object or library built by nextgenmap_amd/build.py: unbundles the
clang offload bundle(s) inside the file and prints the AMDGPU metadata notes of the kernels whose (demangled) names contain a pattern.
  python profiles/tools/kernel_resources.py nextgenmap_amd/build/mapper.o cs_canon heavy2"""
import os, re, shutil, struct, subprocess, sys, tempfile
path, pats = sys.argv[1], sys.argv[2:]
data = open(path, "rb").read()
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
seen = set()
for m in re.finditer(re.escape(MAGIC), data):
    base = m.start()
    n = struct.unpack_from("<Q", data, base + 24)[0]
    off = base + 32
    for _ in range(n):
        o, sz, tl = struct.unpack_from("<QQQ", data, off)
        triple = data[off + 24:off + 24 + tl].decode()
        off += 24 + tl
        if "gfx950" not in triple or sz == 0:
            continue
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(data[base + o:base + o + sz])
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
        os.unlink(f.name)
        # kernel entries of amdhsa.kernels: blocks that start with "  - .agpr_count:" (keys are sorted)
        blocks = re.split(r"\n\s+- (?=\.agpr_count:)", out)
        for b in blocks[1:]:
            kv = {}
            for line in b.splitlines():
                mm = re.match(r"\s*(\.[a-z_]+):\s+(\S.*)$", line)
                if mm and not line.startswith("        "):   # (argument entries are nested deeper)
                    kv.setdefault(mm.group(1), mm.group(2).strip())
            name = kv.get(".name")
            if not name or name in seen:
                continue
            dem = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip() if filt else name
            if pats and not any(p in dem or p in name for p in pats):
                continue
            seen.add(name)
            print("%s\n    vgpr %s (spilled %s)  sgpr %s (spilled %s)  agpr %s  scratch %s B  static LDS %s B  max workgroup %s" % (
                dem[:160], kv.get(".vgpr_count"), kv.get(".vgpr_spill_count"), kv.get(".sgpr_count"), kv.get(".sgpr_spill_count"), kv.get(".agpr_count"),
                kv.get(".private_segment_fixed_size"), kv.get(".group_segment_fixed_size"), kv.get(".max_flat_workgroup_size")))
