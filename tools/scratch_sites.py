"""Development aid: where a kernel's scratch (spill) instructions sit relative to its MFMA main loop.
    python tools/scratch_sites.py gemm.hip [name-substring]
Compiles csrc/<file> to ISA (device only; no GPU needed) and lists, per kernel with scratch traffic, how many scratch loads / stores lie before the
first MFMA, between the first and the last one (main loop + anything the compiler sank into it), and after."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "blim_amd", "csrc")


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "gemm.hip"
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    out = f"/tmp/{os.path.splitext(src)[0]}.s"
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out]
    if src == "vision.hip":
        cmd[1:1] = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
    subprocess.run(cmd, check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    starts = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^(_Z\w+):", l)] if m]
    names = [n for _, n in starts]
    try:
        dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    except OSError:
        dem = names
    for k, (i0, name) in enumerate(starts):
        i1 = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
        body = lines[i0:i1]
        if sub and sub not in dem[k]:
            continue
        sc = [(n, l.strip()) for n, l in enumerate(body) if re.search(r"\bscratch_(load|store)", l)]
        if not sc:
            continue
        mf = [n for n, l in enumerate(body) if "v_mfma" in l]
        a, b = (mf[0], mf[-1]) if mf else (0, 0)
        pre = sum(1 for n, _ in sc if n < a); mid = sum(1 for n, _ in sc if a <= n <= b); post = sum(1 for n, _ in sc if n > b)
        # loop structure: labels between first and last mfma
        print(f"{dem[k][:70]:70s} scratch ops {len(sc):3d}: before first MFMA {pre}, between first and last MFMA {mid}, after {post}")
        if "-v" in sys.argv:
            for n, l in sc:
                print(f"      line {n:6d} ({'pre' if n < a else 'mid' if n <= b else 'post'}): {l}")


if __name__ == "__main__":
    main()
