"""The committed fixtures are exactly what the committed recipe produces from the real reference: when /root/reference is
present (the build container), tests/golden/make_golden.py is run end to end into a temporary directory -- in a
subprocess, because the recipe patches torch.Tensor.cuda / torch.zeros to keep the reference on the CPU -- and every
array is compared bit for bit with tests/golden/*.npz.  Skipped where the reference does not exist (the GPU box)."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
REF = os.environ.get("BSVD_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "Experimental_root")), reason="reference checkout not present")
def test_recipe_runs_end_to_end_and_reproduces_every_fixture(tmp_path):
    code = ("import sys; sys.dont_write_bytecode = True; sys.path.insert(0, %r); import make_golden as mg; mg.main(%r)"
            % (GOLDEN, str(tmp_path)))
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    made = sorted(os.path.basename(f) for f in glob.glob(os.path.join(str(tmp_path), "*.npz")))
    have = sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "*.npz")))
    assert made == have, (set(made) ^ set(have))
    for name in made:
        a, b = np.load(os.path.join(str(tmp_path), name)), np.load(os.path.join(GOLDEN, name))
        assert sorted(a.files) == sorted(b.files), name
        for k in a.files:
            assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype and a[k].tobytes() == b[k].tobytes(), (name, k)
    # the recipe must not leave bytecode in the read-only reference tree
    assert not glob.glob(os.path.join(REF, "Experimental_root", "archs", "__pycache__"))
