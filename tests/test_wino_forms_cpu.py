"""The Winograd forms of bsvd_amd/csrc/wino_forms.h on the CPU (g++, no HIP): the matrices satisfy the defining identity
AT [(G g) . (BT d)] = 3-tap correlation, and the hand-factored input / output transforms the kernel uses equal the matrix forms."""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_wino_forms_identity_and_factoring():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "wino_forms_test")
        subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "bsvd_amd", "csrc"),
                        os.path.join(HERE, "native", "wino_forms_test.cpp"), "-o", exe], check=True)
        r = subprocess.run([exe], capture_output=True, text=True)
    print(r.stdout)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
    assert "F(2,3)" in r.stdout and "F(4,3)" in r.stdout and "F(6,3)" in r.stdout


def test_wino_eligibility_depends_on_the_layer_only():
    """every schedule (clip / stream / sharded / MIMO) must run the same arithmetic per layer: the choice may not look at the clip"""
    sys.path.insert(0, ROOT)
    from bsvd_amd.engine import wino_eligible
    from bsvd_amd.netspec import make_netspec
    net = make_netspec([64, 128, 256], 64, 4, 3, "relu6", 64, False)
    picked = [sp.name for sp in net.layers if wino_eligible(sp, "f16x3")]
    assert len(picked) == 20 and all(sp.cin_pad >= 128 and sp.stride == 1 for sp in net.layers if wino_eligible(sp, "f16x3"))
    assert not any(wino_eligible(sp, "fp32") for sp in net.layers)
    assert len([1 for sp in net.layers if wino_eligible(sp, "f16x3", 256)]) == 10
