"""Where a wave's time goes inside one K-step of the compensated GEMM's SECOND pass (gemm.hip phase 2; instrumented + ablation libraries: `make -C blim_amd/csrc ablate_p2`).

Every wave, per step: [quadrants A + B: 16 MFMAs, 6 refills from the step's own tile, 3 LDS-DMA] [vmcnt: my LDS-DMA pieces of the next tile] [barrier] [quadrant C: 8 MFMAs,
2 refills + scales from the next tile, 2 LDS-DMA] [quadrant D: 8 MFMAs, 4 refills, 2 LDS-DMA].  Numbers are shader-clock cycles (s_memtime) summed over the pass's K loop of each tile,
averaged over tiles; one line per ablation build:
0 = as shipped, 1 = no MFMAs, 2 = no fragment refills, 3 = no LDS-DMA after the first tiles, 4 / 5 = the W operand / both operands moved and read with the byte count of an
e2m1 (fp4) image (round 6: an upper bound on what a mixed-format pass could gain) (1 - 5: timing only, wrong results).

    python tools/gemm_waits_lo6.py            # all four builds, each in its own process
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
names = ("qa+qb", "vmcnt", "barrier", "qc", "qd")
LABEL = {0: "as shipped", 1: "no MFMAs", 2: "no fragment refills", 3: "no LDS-DMA", 4: "W operand with an e2m1 image's bytes (17 of 25 KiB)", 5: "both operands with e2m1 bytes", 6: "vmcnt waits one tile looser (is it LDS-DMA latency?)", 7: "a barrier every other step only (is it the rendezvous?)"}


def one(v):
    os.environ["BLIM_LIB_PATH"] = os.path.join(ROOT, "tools", "bin", f"libblim_hip_ablate_p2_{v}.so")
    import numpy as np, torch
    from blim_amd import engine as eng
    lib = eng.load_library()
    for (M, N, K) in ((32768, 37888, 3584), (32768, 3584, 18944)):
        a = torch.empty((M, 2 * K), dtype=torch.float16, device="cuda"); w = torch.empty((N, K), dtype=torch.float16, device="cuda")
        a.normal_(); a[:, K:] *= 2.0 ** -11; w.normal_(std=0.02)
        nwg = (M // 256) * (N // 256)
        st = torch.zeros((nwg * 3, 8), dtype=torch.int64, device="cuda")
        eng.gemm_f16_lo6(a, w); torch.cuda.synchronize()
        lib.blim_debug_gemm_stamps(st.data_ptr())
        eng.gemm_f16_lo6(a, w); torch.cuda.synchronize()
        lib.blim_debug_gemm_stamps(None)
        s = st.cpu().numpy().astype(np.float64)
        loop_us = (s[:nwg, 2] - s[:nwg, 1]).mean() * 0.01
        wt = s[nwg:].reshape(nwg, 2, 8)[:, :, :5]
        line = f"[{v}: {LABEL[v]}] {M}x{N}x{K}: both passes {loop_us:.1f} us per tile ({K // 64} 16-bit + {K // 128} e2m3 K-steps)"
        for g in range(2):
            tot = wt[:, g].sum(axis=1).mean()
            parts = wt[:, g].mean(axis=0)
            line += f"\n   waves {4 * g}-{4 * g + 3}: {tot / (K // 128):.0f} cycles per e2m3 K-step | " + "  ".join(f"{n} {v_ / (K // 128):.0f}" for n, v_ in zip(names, parts))
        print(line, flush=True)
        del a, w


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one(int(sys.argv[1]))
    else:
        for v in [int(x) for x in os.environ.get("P2_VARIANTS", "0,1,2,3,4,5,6").split(",")]:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), str(v)], capture_output=True, text=True)
            print(r.stdout if r.returncode == 0 else f"[{v}] failed: {r.stderr[-800:]}", flush=True)
