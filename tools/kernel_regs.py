"""Register / LDS / occupancy report of the compiled kernels:  python tools/kernel_regs.py [name-substring] [extra hipcc flags...]"""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
       "-Rpass-analysis=kernel-resource-usage", "-c", "solver_f32.hip" if "--double" not in sys.argv else "solver_f64.hip", "-o", "/tmp/kernel_regs.o"] + sys.argv[2:]
out = subprocess.run(cmd, cwd=os.path.join(root, "bdd_amd", "csrc"), capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    dem = subprocess.run(["/usr/bin/c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    if pat in dem:
        print(f"{dem[:90]:90s} VGPR {r.get('VGPRs')} AGPR {r.get('AGPRs')} spill {r.get('VGPRs Spill')} occ {r.get('Occupancy [waves/SIMD]')} LDS {r.get('LDS Size [bytes/block]')}")
