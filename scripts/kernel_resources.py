#!/usr/bin/env python3
"""registers / scratch / spills of every kernel in an object or the library (no GPU needed):
scripts/kernel_resources.py <file.o|.so> [regex on the demangled name]"""
import re
import subprocess
import sys
import tempfile

L = "/opt/rocm/lib/llvm/bin/"
with tempfile.TemporaryDirectory() as t:
    subprocess.run([L + "llvm-objcopy", f"--dump-section=.hip_fatbin={t}/fat.bin", sys.argv[1]], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([L + "clang-offload-bundler", "--type=o", f"--input={t}/fat.bin", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--output={t}/dev.co", "--unbundle"], check=True)
    notes = subprocess.run([L + "llvm-readelf", "--notes", f"{t}/dev.co"], capture_output=True, text=True).stdout
rows, cur = [], None
for line in notes.splitlines():
    m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2).strip()
    if k == "name" and v.startswith("_Z"):
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None and k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "vgpr_spill_count", "group_segment_fixed_size", "agpr_count"):
        cur.setdefault(k, v)
dem = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for r, d in zip(rows, dem):
    d = d.replace("psk::", "").replace("void ", "").split("(")[0]
    if pat and not re.search(pat, d):
        continue
    print("vgpr=%4s agpr=%3s scratch=%5s spill=%3s  %s" % (r.get("vgpr_count", "?"), r.get("agpr_count", "0"), r.get("private_segment_fixed_size", "?"),
                                                        r.get("vgpr_spill_count", "0"), d[:170]))
