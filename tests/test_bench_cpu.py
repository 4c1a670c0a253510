"""bench.py's host-side contract that can be checked without a GPU: the BASELINE workloads table, argument handling and the
refusal modes (no GPU work is started here)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, env=e, timeout=300)


def test_workloads_cover_the_single_gpu_baseline_configs():
    sys.path.insert(0, ROOT)
    import bench
    from bsvd_amd.netspec import make_netspec
    assert sorted(bench.WORKLOADS) == ["c1", "c2", "c3", "c5"]
    assert (bench.WORKLOADS["c1"]["h"], bench.WORKLOADS["c1"]["w"], bench.WORKLOADS["c1"]["frames"], bench.WORKLOADS["c1"]["mode"]) == (540, 960, 10, "clip")
    assert (bench.WORKLOADS["c2"]["h"], bench.WORKLOADS["c2"]["w"], bench.WORKLOADS["c2"]["frames"]) == (480, 856, 85)
    assert bench.WORKLOADS["c3"]["blind"] and not bench.WORKLOADS["c1"]["blind"]
    assert (bench.WORKLOADS["c5"]["h"], bench.WORKLOADS["c5"]["w"], bench.WORKLOADS["c5"]["mode"]) == (1080, 1920, "perframe")
    # FLOP per frame the JSON reports
    c64 = make_netspec([64, 128, 256], 64, 4, 3, "relu6", 64)
    assert 2 * c64.macs_per_frame(540, 960) == 1_227_239_424_000
    assert abs(2 * c64.macs_per_frame(480, 856) / 1e12 - 0.9727) < 1e-3
    assert abs(2 * c64.macs_per_frame(1080, 1920) / 1e12 - 4.909) < 1e-3
    assert bench.usable_cores() >= 1


def test_gpus_flag_must_match_the_launcher():
    r = _run("--gpus", "2")
    assert r.returncode != 0 and "torch.distributed.run" in (r.stderr + r.stdout)


def test_stream_modes_refuse_more_than_one_rank():
    # a live stream does not shard in time: replicas only; the refusal comes before any device work
    r = _run("--gpus", "2", "--mode", "stream", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "single-GPU" in (r.stderr + r.stdout)
    r = _run("--gpus", "3", "--scaling", "strong", "--total-frames", "80", env={"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "not divisible" in (r.stderr + r.stdout)


def test_power_probe_parses_rocm_smi_and_survives_its_absence(tmp_path, monkeypatch):
    """bench.power_probe: package power / cap / shader clock from rocm-smi's text (a stand-in on PATH prints the real tool's
    lines); without rocm-smi the probe returns None instead of failing the bench."""
    sys.path.insert(0, ROOT)
    import bench
    fake = tmp_path / "rocm-smi"
    fake.write_text("#!/bin/sh\n"
                    "case \"$*\" in\n"
                    "  *showmaxpower*) echo 'GPU[0]\t\t: Max Graphics Package Power (W): 1400.0';;\n"
                    "  *) echo '============ ROCm System Management Interface ============'\n"
                    "     echo 'GPU[0]\t\t: Current Socket Graphics Package Power (W): 1398.0'\n"
                    "     echo 'GPU[0]\t\t: mclk clock level: 0: (2000Mhz)'\n"
                    "     echo 'GPU[0]\t\t: sclk clock level: 1: (1620Mhz)';;\n"
                    "esac\n")
    fake.chmod(0o755)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    calls = []
    pw = bench.power_probe(lambda: calls.append(1), 1.2)
    assert pw and pw["cap_w"] == 1400.0 and pw["package_w"] == 1398.0 and pw["sclk_mhz"] == 1620.0 and pw["samples"] >= 1
    assert calls
    monkeypatch.setenv("PATH", str(tmp_path / "nowhere"))
    assert bench.power_probe(lambda: None, 0.1) is None


def test_value_normalised_arithmetic():
    """bench.normalise_value: a box whose fixed calibration layer takes 5 % longer than the reference box's shows 5 %
    fewer frames/s on the same commit; the normalised figure undoes exactly that, and is None when a leg is missing."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.normalise_value(366.0, 1.05, 1.00) == pytest.approx(384.3)
    assert bench.normalise_value(384.4, 1.00, 1.00) == pytest.approx(384.4)
    assert bench.normalise_value(400.0, 0.95, 1.00) == pytest.approx(380.0)
    for cal, ref in ((None, 1.0), (1.0, None), (0.0, 1.0), (1.0, 0.0)):
        assert bench.normalise_value(366.0, cal, ref) is None
    assert set(bench.BOX_CAL_REF) >= {"direct64_ms", "wino256_ms", "source"} and bench.BOX_CAL_REF["direct64_ms"] > 0


def test_committed_traffic_table_was_collected_on_these_kernel_sources():
    """roofline.traffic is read from the committed profiles/traffic.json (PMC passes cannot run inside bench.py): the table names the
    kernel sources it was collected on, bench.py compares (roofline.traffic_sources_match), and this test keeps the committed pair in step --
    a kernel change without a new PMC session fails here instead of silently quoting stale bytes."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert t.get("sources_sha16") == bench.kernel_sources_sha16(), "kernel sources changed since profiles/traffic.json was collected: re-run tools/pmc_run.sh + make_traffic.py"
