"""bench.py prints ONE compact headline object as the LAST line of stdout (round 5's full object had grown to 24.7 KB on
one line and the driver's record came back `parsed: null`).  This runs the formatter on a canned full result -- round 5's
own (profiles/r05z_bench_full.json) -- and checks size, validity and the keys the driver's record held in rounds 1-4."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_compact_line_is_small_valid_and_complete():
    full = json.load(open(os.path.join(ROOT, "profiles", "r05z_bench_full.json")))
    assert len(json.dumps(full)) > 20000          # (the canned object is the one that broke the record)
    full["full"] = "bench_full.json"
    line = _bench().compact_line(full)
    assert "\n" not in line and len(line) < 4096, len(line)
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in d, k
    assert d["vs_baseline"] is None and d["dtype"] == "u8" and d["higher_is_better"] is True
    for k in ("workload", "codec", "units_per_gpu", "unit_bytes", "compressed_bytes_per_gpu", "decompressed_bytes_per_gpu"):
        assert k in d["config"], k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel", "kernel_ms", "algorithmic_bytes_per_launch",
              "per_kernel_ms", "dominant_kernel", "dominant_kernel_frac"):
        assert k in d["roofline"], k
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-4
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert set(d["per_codec_summary"]) == set(full["per_codec"])
    for name, (ms, gibs, frac) in d["per_codec_summary"].items():
        assert abs(ms - full["per_codec"][name]["ms_per_step"]) < 0.01 and 0 < frac < 1
    # the headline figures are the full object's
    assert abs(d["value"] - full["value"]) < 1e-3 and abs(d["ms_per_step"] - full["ms_per_step"]) < 1e-3


def test_compact_line_without_the_optional_parts():
    """N > 1 ranks / --no-cpu-baseline / --no-per-codec runs carry no cpu_baseline, per_codec or latency line."""
    full = json.load(open(os.path.join(ROOT, "profiles", "r05z_bench_full.json")))
    for k in ("cpu_baseline", "cpu_baseline_all_cores", "cpu_context", "per_codec", "config1_latency", "archive_paths", "lz4_streamed"):
        full.pop(k, None)
    full["roofline"].pop("per_kernel_ms")
    d = json.loads(_bench().compact_line(full))
    assert "cpu_baseline" not in d and "per_codec_summary" not in d and d["roofline"]["frac"] > 0
