"""Runs tests/native/abi_parity.bin on the MI355X: a g++ host program with no Python/PyTorch in the process that drives
libbsvd_hip.so through include/bsvd_hip.h (device memory from the HIP runtime) and checks four layer cases -- temporal
fusion with halos, stride 2, PixelShuffle + skip, a split-fp16 head -> conv -> tail chain, the Winograd form -- against the plain-C
double-accumulating oracle conv (oracle/conv_ref.c).  Built by __graft_entry__.build()."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "native", "abi_parity.bin")


@pytest.mark.gpu
def test_c_abi_consumer_without_torch():
    subprocess.check_call(["make", "-C", os.path.dirname(BIN), "-s"])      # (no-op when the binary is newer than the header / library)
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi_parity: all cases ok" in r.stdout
    assert r.stdout.count(" ok") >= 8 and "Winograd F(2,3)" in r.stdout


def test_native_consumer_builds_and_links_only_the_abi():
    """CPU-side: the consumer compiles against the public header alone and its undefined bsvd_* symbols are exactly
    entry points the header declares (no reach into library internals, no torch)."""
    subprocess.check_call(["make", "-C", os.path.dirname(BIN), "-s"])
    out = subprocess.run(["nm", "-D", "--undefined-only", BIN], capture_output=True, text=True).stdout
    used = sorted({l.split()[-1].split("@")[0] for l in out.splitlines() if " bsvd_" in l})
    import re
    hdr = open(os.path.join(ROOT, "include", "bsvd_hip.h")).read()
    declared = set(re.findall(r"\b(bsvd_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)))
    assert used and set(used) <= declared, (used, declared)
    libs = subprocess.run(["ldd", BIN], capture_output=True, text=True).stdout
    assert "libbsvd_hip.so" in libs and "torch" not in libs and "python" not in libs.lower()
