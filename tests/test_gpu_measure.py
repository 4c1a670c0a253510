"""The MEASUREMENT build of the library (tools/build_measure.sh, -DBSVD_MEASURE): the kernel variants that are not in the product
 stay parity-tested -- in their own library, in their own process.  `measure` marker: these tests build / load
build/measure/libbsvd_hip*.so (prebuilt by __graft_entry__.build(); built here with hipcc when missing, ~1 min each)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.measure]
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _measure_lib(extra="", suffix=""):
    path = os.path.join(ROOT, "build", "measure", "libbsvd_hip%s.so" % suffix)
    srcs = [os.path.join(ROOT, "bsvd_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "bsvd_amd", "csrc"))
            if f.endswith((".hip", ".h"))] + [os.path.join(ROOT, "include", "bsvd_hip.h")]
    if not os.path.exists(path) or any(os.path.getmtime(f) > os.path.getmtime(path) for f in srcs):
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "build_measure.sh"), extra, suffix], timeout=1500)
    return path


def _run(lib, *args):
    env = dict(os.environ, BSVD_HIP_LIB=lib, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "measure_driver.py")] + list(args), env=env, capture_output=True, text=True,
                       timeout=1500)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


def test_measurement_variants_vs_oracle_and_bit_identical_within_a_form():
    out = _run(_measure_lib(), "forms")
    assert "MEASURE FORMS OK" in out


def test_mixasm_resplit_is_bit_identical_to_the_compiler_form(tmp_path):
    """the inline-asm re-split of the Winograd transform (v_fma_mixlo / mixhi from the packed hi pair, one asm block) against the
    plain C++ form the compiler schedules itself (-DBSVD_WX_MIXASM=0): same bits on F(2,3) and F(6,3), pinned to the ROCm in this image."""
    a, b = str(tmp_path / "asm.json"), str(tmp_path / "plain.json")
    _run(_measure_lib(), "digest", a)
    _run(_measure_lib("-DBSVD_WX_MIXASM=0", "_nomixasm"), "digest", b)
    da, db = json.load(open(a)), json.load(open(b))
    assert da and da == db, (da, db)
