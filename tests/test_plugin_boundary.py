"""The drop-in boundary: bsvd_amd next to the reference plug-in in ONE BasicSR registry.

BasicSR's registry asserts on duplicate names (/root/reference/BasicSR/basicsr/utils/registry.py:38-41) and the reference's
import-time scans register BSVD / TSN / DenoisingModel / ValFolderDataset (Experimental_root/archs/__init__.py:5-9,
models/__init__.py:5-9).  These tests load that registry -- the reference's REAL registry.py by path where /root/reference
exists, a strict stand-in elsewhere -- import the reference plug-in before AND after ``bsvd_amd`` and check the documented
recipe: nothing asserts in either order, the engine answers to ``<name>_MI355X`` at once and to the
stock names after ``bsvd_amd.install(replace=True)``.  Container only: the reference's real ``DenoisingModel.test`` and
``DenoisingModel.validation`` drive ``bsvd_amd.BSVD`` (CPU oracle executor patched in) and agree with the reference net.
Every scenario runs in a fresh interpreter (tests/plugin_boundary_driver.py)."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("BSVD_REFERENCE", "/root/reference")
HAVE_REF = os.path.isdir(os.path.join(REF, "Experimental_root"))
needs_ref = pytest.mark.skipif(not HAVE_REF, reason="reference checkout not present (the GPU box)")


def drive(*args):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "plugin_boundary_driver.py"), *args], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


NAMES = ("BSVD", "TSN", "DenoisingModel", "ValFolderDataset")


def check_registry(res, dataset_in_reference):
    ref_names = NAMES if dataset_in_reference else NAMES[:3]
    for snap in ("after_import", "after_install_keep", "after_uninstall"):
        for n in NAMES:
            assert res[snap][n + "_MI355X"] == "bsvd_amd", (snap, n)
        for n in ref_names:                   # importing bsvd_amd / install(replace=False) never displace the reference
            assert res[snap][n] not in (None, "bsvd_amd"), (snap, n, res[snap][n])
    if not dataset_in_reference:              # DALI-less box: the reference never registered its dataset -> free name
        assert res["after_import"]["ValFolderDataset"] is None
        assert res["install_keep"]["dataset"]["ValFolderDataset"] == "engine"
        assert res["after_uninstall"]["ValFolderDataset"] is None
    assert res["install_keep"]["arch"] == {"BSVD": "reference", "TSN": "reference"}
    assert res["install_keep"]["model"] == {"DenoisingModel": "reference"}
    for kind, names in (("arch", ("BSVD", "TSN")), ("model", ("DenoisingModel",)), ("dataset", ("ValFolderDataset",))):
        assert res["install_replace"][kind] == {n: "engine" for n in names}
    for n in NAMES:
        assert res["after_install_replace"][n] == "bsvd_amd"
    assert res["built_arch"] == ["bsvd_amd.arch", "BSVD", True, 16]
    assert res["model_cls"] and res["dataset_cls"]


@pytest.mark.parametrize("order", ["ref_first", "engine_first"])
def test_both_import_orders_with_a_strict_registry(order):
    check_registry(drive("registry", "stub", order), dataset_in_reference=True)


@needs_ref
@pytest.mark.parametrize("order", ["ref_first", "engine_first"])
def test_both_import_orders_with_the_reference_registry_and_plugin(order):
    res = drive("registry", "real", order)
    check_registry(res, dataset_in_reference=False)
    assert res["after_import"]["BSVD"] == "ref_bsvd_arch"           # the reference's own class (bsvd_arch.py:440)
    assert res["after_import"]["DenoisingModel"] == "Experimental_root"


@needs_ref
def test_reference_denoising_model_test_drives_the_engine_class():
    """denoising_model.py:170-190 -> validation_seq_infer.py:33-100 -> :10-31 -> net(noisyframe, noise_map=...)"""
    res = drive("drive_test")
    assert res["engine_cls"] == "bsvd_amd.arch" and res["ref_cls"] == "ref_bsvd_arch"
    assert res["shape"] == [1, 5, 3, 30, 50] and res["same_shape"]
    assert res["ref_range"][0] >= 0.0 and res["ref_range"][1] <= 1.0
    assert res["max_abs"] < 1e-4, res["max_abs"]


@needs_ref
def test_validation_matches_the_reference_validation(tmp_path):
    """the calls of basicsr.test_pipeline (BasicSR/basicsr/test.py:26-41) against both model classes: same totals, same log
    lines (denoising_model.py:353-359), same per-folder CSVs (:335-345)"""
    res = drive("drive_validation", str(tmp_path))
    r, e = res["reference"], res["engine"]
    assert r["model"].startswith("Experimental_root") and r["net"] == "ref_bsvd_arch"
    assert e["model"] == "bsvd_amd.denoise" and e["net"] == "bsvd_amd.arch"
    assert set(r["total"]) == set(e["total"]) == {"psnr", "psnr_float"}
    for k in r["total"]:
        assert abs(r["total"][k] - e["total"][k]) < 1e-3, (k, r["total"], e["total"])
    assert len(r["log"]) == len(e["log"]) == 2                      # the reference logs after every folder (:322)
    assert r["log"][0] == e["log"][0]
    assert sorted(r["csv"]) == sorted(e["csv"]) == ["bus.csv", "car.csv"]
    for name in r["csv"]:
        ra, ea = r["csv"][name].splitlines(), e["csv"][name].splitlines()
        assert ra[0] == ea[0] and len(ra) == len(ea)
        for x, y in zip(ra[1:], ea[1:]):
            xs, ys = x.split(","), y.split(",")
            assert xs[0] == ys[0]
            assert all(abs(float(a) - float(b)) < 1e-3 for a, b in zip(xs[1:], ys[1:])), (x, y)
