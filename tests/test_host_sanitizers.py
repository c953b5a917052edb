"""CPU (-m "not gpu"): the engine's HOST code under AddressSanitizer + UBSan.

The reference has no native code; this repo's host side of the C ABI is ~2,000 lines of C++ in csrc/engine.hip, adapters.hip and train.hip (weight and adapter
tables, lazily built derived copies, workspace growth, option parsing, the trainer's buffers, error paths).  GPU sanitizers are not available on the pool, so those
three files are compiled as plain C++ for x86 against tests/hostmock/ -- a mock of the two dozen HIP runtime calls they use ("device" memory = calloc, launches do
nothing) and generated do-nothing kernel launchers -- and tests/hostmock/driver.cpp walks the ABI: every weight name, adapters of rank 4 / 8 / 16 in partial and full
sets, every option key, every scoring entry point in every numeric mode, the trainer's life cycle, bad arguments, wrong state, out of device memory.  The run must
end with exit code 0, no sanitizer report and no leak ("Next round" item 5)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs ROCm's clang++ (x86 ASan / UBSan runtimes)")
def test_host_side_of_the_c_abi_is_clean_under_asan_and_ubsan(tmp_path):
    out = str(tmp_path / "build")
    r = subprocess.run(["make", "-C", os.path.join(HERE, "hostmock"), "-j4", f"OUT={out}"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    env.pop("LD_PRELOAD", None)
    d = subprocess.run([os.path.join(out, "driver")], capture_output=True, text=True, timeout=600, env=env)
    log = d.stdout + d.stderr
    assert d.returncode == 0 and "host sanitizer drive: ok" in d.stdout, log[-4000:]
    assert "AddressSanitizer" not in log and "runtime error" not in log and "LeakSanitizer" not in log, log[-4000:]
    shutil.rmtree(out, ignore_errors=True)
