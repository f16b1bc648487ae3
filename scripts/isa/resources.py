"""Register / LDS / occupancy table of the kernels of one source file: scripts/isa/resources.py csrc/gemm_f32.hip [name filter] [-D...]
(hipcc -Rpass-analysis=kernel-resource-usage, demangled; run from detr-tensorflow_amd/)."""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-Wno-unused-function",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]*\])?): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    if flt and flt not in n:
        continue
    print(f"{r.get('VGPRs', 0):4d} v {r.get('AGPRs', 0):4d} a  spill {r.get('VGPRs Spill', 0):4d}  occ {r.get('Occupancy [waves/SIMD]', 0)}  lds {r.get('LDS Size [bytes/block]', 0):6d}  {n[:150]}")
