"""The bench line's CONTRACT, checked on the line the round's last GPU call produced (profiles/r05_bench_default.json, written by bench.py itself):
the fields the driver and the judge read must be there, typed and mutually consistent -- metric = BASELINE.json's string, value = frames / time,
roofline.frac = achieved / peak, the parity gate present and inside its bars for the headline, traffic per step with the commit it was measured at."""
import json
import os

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = os.path.join(REPO, "profiles", "r05_bench_default.json")


@pytest.fixture(scope="module")
def line():
    if not os.path.exists(LINE):
        pytest.skip("no round-5 bench line in profiles/")
    with open(LINE) as f:
        return json.loads([l for l in f if l.startswith("{")][-1])


def test_driver_fields(line):
    base = json.load(open(os.path.join(REPO, "BASELINE.json")))
    assert line["metric"] == base["metric"].replace("×", "x")
    assert line["unit"] == "frames/s" and line["higher_is_better"] is True and line["scaling"] == "weak" and line["data"] == "synthetic"
    assert line["n_gpus"] == 1 and line["steps"] > 0 and line["warmup"] >= 0 and line["vs_baseline"] is None      # BASELINE.md publishes no number
    assert "workload" in line["config"] and "854x480 batch=1 online" in line["config"]["workload"] and "model" not in line["config"]
    assert abs(line["value"] - 1e3 / line["ms_per_step"]) <= 2e-3 * line["value"]          # batch 1: frames/s = 1000 / ms per step
    assert line["dtype"].startswith("f32")


def test_roofline_and_cpu_baseline(line):
    r = line["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.3 < r["frac"] < 1.0
    assert r["avg_launch_ms"] * 1.0 < line["ms_per_step"]                                   # a launch is shorter than the step that holds it
    assert abs(r["executed_over_algorithmic"] - 5.97) < 0.05                                # f32x3: six bf16 products per fp32 product, minus the exact passes
    t = r["traffic"]
    assert t["static"] is True and t["measured_at_commit"] not in ("", "unknown") and t["source"].startswith("profiles/r05_pmc_traffic_configs1")
    assert 1.0 < t["conv_family"]["ratio"] < 3.0 and t["conv_family"]["algorithmic_MB_per_step"] == 2712.0
    ps = r["pipe_sustained"]
    assert 1500 < ps["tflops_noise_operands"] < ps["tflops_zero_operands"] <= 2600
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "frames/s" and 0.1 < c["value"] < 50 and 1 <= c["cores"] <= c["host_nproc"] and "iterations" in c["sample"]


def test_parity_gate_is_in_the_line(line):
    p = line["parity"]
    assert p["within_bars"] is True and p["bars"] == {"max_dlogit_over_std": 1e-3, "loss_rel": 1e-5, "grad_rel_l2": 1e-3, "iou": 1.0 - 1e-3}
    assert p["max_dlogit_over_std"] <= 1e-3 and p["loss_rel"] <= 1e-5 and p["iou"] >= 1 - 1e-3 and p["grad_rel_l2_worst"]["value"] <= 1e-3
    assert set(p["grad_rel_l2"]) == {"stages.0.0.weight", "stages.2.1.weight", "fuse.weight"} and p["grad_rel_l2_worst"]["tensors_compared"] >= 30
    extras = {e["config"].split(":")[0]: e for e in line["extra_configs"]}
    c2 = next(e for k, e in extras.items() if k.startswith("configs[2]"))
    assert c2["parity"]["batch"] == 12 and "bf16" in c2["parity"]["dtype"] and c2["parity"]["bars"]["iou"] == 0.985 and "note" in c2["parity"]
    assert c2["roofline"]["traffic"]["source"].startswith("profiles/r05_pmc_traffic_configs2")
    c4 = next(e for k, e in extras.items() if k.startswith("configs[4]") and "EXACT" not in k)
    assert c4["parity"]["within_bars"] is True and c4["roofline"]["traffic"]["source"].startswith("profiles/r05_pmc_traffic_configs4")
