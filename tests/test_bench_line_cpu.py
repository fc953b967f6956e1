"""The bench line the driver parses: compact, < 4 KB, carries `roofline` + `cpu_baseline`.

Round 5's line was 21.5 KB (four more `extras` blocks) and BENCH_r05.parsed came back null; the line is now built by
bench.compact_line from the full result, which goes to gpurun_out/bench_full.json + stderr instead."""
import glob
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORDED = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[3-9]", "bench_h8192_chain*.json")))
RECORDED = [p for p in RECORDED if "pmc" not in p and "kernel_durations" not in p]


def _module_path_of(out):
    """what bench.main() attaches as roofline.module_path, rebuilt for recorded lines that lack it"""
    ex = out.get("extras") or {}
    mp = dict((out.get("roofline") or {}).get("module_path") or {})
    for key, name in (("single_launch_per_layer", "h8192"), ("h4096", "h4096")):
        e = ex.get(key) or {}
        if "us_per_launch" in e and name not in mp:
            mp[name] = {"us_per_layer": e["us_per_launch"], "GBps": e["GBps"], "frac": e["frac_of_8TBps"], "kernel": e["kernel"]}
    return mp


@pytest.mark.parametrize("path", RECORDED, ids=[os.path.relpath(p, ROOT) for p in RECORDED])
def test_compact_line_of_recorded_results(path):
    out = json.load(open(path))
    out.setdefault("roofline", {})["module_path"] = _module_path_of(out)
    out["full"] = "gpurun_out/bench_full.json"
    line = bench.compact_line(out)
    assert "\n" not in line
    assert len(line) < bench.COMPACT_LIMIT, len(line)
    c = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in c, k
    assert c["value"] == out["value"] and c["ms_per_step"] == out["ms_per_step"]
    assert c["unit"] == "GB/s" and c["dtype"] == "f16" and c["data"] == "synthetic" and c["vs_baseline"] is None
    assert "workload" in c["config"] and "model" not in c["config"]
    r = c["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    if "cpu_baseline" in out:
        cb = c["cpu_baseline"]
        assert set(cb) == {"value", "unit", "cores", "kind", "sample"} and cb["kind"] in ("port", "reference")
    assert "extras" not in c and "soak" not in c


def test_compact_line_stays_small_whatever_the_extras_hold():
    big = {"metric": "m", "value": 1.0, "unit": "GB/s", "n_gpus": 1, "steps": 2, "warmup": 1, "ms_per_step": 0.2,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
           "config": {"workload": "w" * 5000, "arithmetic": "reference " + "x" * 3000, "parallelism": "p" * 900, "kernel": "k"},
           "roofline": {"bound": "hbm", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 1 / 8000.0, "traffic": None,
                        "note": "n" * 9000,
                        "module_path": {f"row{i}": {"us_per_layer": 1.0, "frac": 0.1, "kernel": "z" * 200} for i in range(80)}},
           "cpu_baseline": {"value": 0.01, "unit": "GB/s", "cores": 8, "kind": "port", "sample": "s" * 4000},
           "extras": {f"e{i}": {"what": "y" * 500} for i in range(60)}}
    line = bench.compact_line(big)
    assert len(line) < bench.COMPACT_LIMIT
    c = json.loads(line)
    assert c["roofline"]["frac"] == 1 / 8000.0 and c["cpu_baseline"]["kind"] == "port"


def test_tp_row_line_is_compact():
    out = {"metric": "m", "value": 900.0, "unit": "GB/s", "n_gpus": 8, "steps": 20, "warmup": 5, "ms_per_step": 3.0,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
           "config": {"workload": "Llama-3-70B shaped decoder layers x 20", "mode": "tp_row", "decoder_layers": 20, "hipgraph": True,
                      "arithmetic": "reference roundings per weight", "parallelism": "tp8 row-parallel"},
           "roofline": {"bound": "hbm", "achieved": 112.5, "peak": 8000.0, "unit": "GB/s", "frac": 112.5 / 8000, "traffic": None},
           "tp_row": {"us_per_decoder_layer": 150.0, "parity_rel_err_vs_cpu_oracle": 2e-4, "parity_per_projection": {"q": 1e-4} },
           "weak_scaling": {"what": "8 x rings", "value": 20000.0, "unit": "GB/s", "us_per_launch": 6.5, "scaling": "weak"}}
    c = json.loads(bench.compact_line(out))
    assert c["n_gpus"] == 8 and c["scaling"] == "strong" and c["weak_scaling"]["value"] == 20000.0
    assert c["tp_row"]["us_per_decoder_layer"] == 150.0
