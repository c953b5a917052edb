"""Register allocation and spills of every kernel of a .hip source (device assembly metadata).  usage: python tools/kernel_regs.py gemm attention ..."""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for f in sys.argv[1:]:
    out = f"/tmp/{f}.s"
    extra = {"vision": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}.get(f, [])          # the Makefile's per-file flags (FLAGS_<file>)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", *extra, "-o", out, os.path.join(ROOT, "blim_amd", "csrc", f + ".hip")],
                   check=True, stderr=subprocess.DEVNULL)
    s = open(out).read()
    for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.vgpr_spill_count:\s+(\d+)', s, re.S):
        name, blk = m.group(1), m.group(2)
        vg = re.search(r'\.vgpr_count:\s+(\d+)', blk); sg = re.search(r'\.sgpr_spill_count:\s+(\d+)', blk)
        try:
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        except FileNotFoundError:
            pass
        print(f"{name[:78]:80s} vgpr {vg.group(1) if vg else '?':>4s}  vgpr spills {m.group(3):>4s}  sgpr spills {sg.group(1) if sg else '?'}")
