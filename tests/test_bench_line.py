"""bench.py's one stdout line must stay parsable by the driver: round 4's 21.7 KB line came back as `parsed: null`.
The line builder is run on canned numbers (round 4's full report, a leg added on top) -- no GPU, no engine call."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("zs_bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _canned():
    return json.load(open(os.path.join(ROOT, "profiles", "r04_bench_final_build.json")))


def test_line_is_short_and_round_trips():
    m = _bench()
    full = _canned()
    assert len(json.dumps(full)) > 20000  # the report that broke the driver's parser
    # more legs than any run has: the line's size must not follow the report's
    for i in range(40):
        full["secondary"]["extra_leg_%d" % i] = dict(full["secondary"]["english64_L1"])
    text = m.compact_line(full)
    assert len(text) < 4096 and "\n" not in text
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] == full["value"] and line["config"]["workload"].startswith("english64")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    legs = line["secondary_summary"]["legs"]
    assert legs["english64_L1"] == [1375.01, 0.000262, 73.05]
    assert legs["inflate1g"][0] == full["secondary"]["inflate1g"]["value"]
    assert "secondary" not in line


def test_emit_report_prints_the_compact_line_last(tmp_path, capsys, monkeypatch):
    m = _bench()
    monkeypatch.setattr(m, "ROOT", str(tmp_path))
    m.emit_report(_canned())
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1 and len(out[0]) < 4096
    assert json.loads(out[0])["secondary_summary"]["full_report"] == "bench_secondary.json"
    full = json.load(open(tmp_path / "bench_secondary.json"))
    assert "corpus_L6" in full["secondary"] and "files" in full["secondary"]["corpus_L6"]


def test_traffic_names_its_source():
    m = _bench()
    b, src = m.pmc_traffic("zs_match_kernel", "traffic_english64_L6")
    assert b and src.startswith("profiles/") and src.endswith("pmc_traffic_english64_L6.json")
    r = m.roofline({"match": 2.7, "links": 0.3}, 94534690, b, None, src)
    assert r["traffic_source"] == src and r["scope"] == "kernel"
    r = m.roofline_with_pmc({"match": 2.7, "links": 0.3}, 94534690, "traffic_english64_L6")
    assert r["kernel"] == "zs_match_kernel" and r["traffic_source"] == src and r["traffic"] == b
    # the rounds of the chunk form are a stage of many launches, and say so
    b, src = m.pmc_traffic("zs_fast_sweep_kernel", "fast64_L1")
    r = m.roofline({"fast_sweep": 46.6, "links": 0.3}, 97620422, b, None, src)
    assert r["scope"] == "stage" and r["launches"] and r["traffic"] > 97620422 and r["traffic_source"] == src
