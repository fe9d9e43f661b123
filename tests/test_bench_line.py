"""bench.py's stdout contract: ONE compact JSON line below 4 KB that round-trips through json.loads and carries the
driver's keys + roofline + cpu_baseline + parity; everything else goes to the side file.  Driven here from a full record of an earlier
run (profiles/r04last_bench_default.json: the 20 KB line the driver could not keep) -- no GPU needed."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_line", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_contract_line_is_small_and_complete(tmp_path):
    bench = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r04last_bench_default.json")))
    assert len(json.dumps(full)) > 16000   # the record that went unparsed
    path = bench.write_detail(full, str(tmp_path / "bench_detail.json"))
    line = bench.compact_line(full, path)
    assert "\n" not in line and len(line.encode()) < 4096, len(line)
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "parity"):
        assert k in d, k
    assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"] and d["vs_baseline"] is None
    assert d["config"]["baseline_config"] == 2 and d["config"]["workload"].startswith("BASELINE configs[1]")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "launches", "algorithmic_bytes_per_launch"):
        assert k in d["roofline"], k
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-4
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["parity"] == {"parity_checked_loci": 8000, "mismatches": 0}
    assert set(d["configs"]) == {"3", "4", "5"}
    for leg in d["configs"].values():
        assert leg["value"] > 0 and "kernel" in leg["roofline"] and "mismatches" in leg["parity"]
    assert d["e2e"]["pipeline_loci_per_s"] == full["e2e"]["pipeline_loci_per_s"]
    # no prose beyond the workload / sample labels
    assert not any(k in line for k in ("value_is", "launch_is", "algorithmic_bytes_model", '"note"'))
    # the side file holds the whole record
    assert json.load(open(path)) == full


def test_contract_line_says_what_binds():
    # next to the nominal SURVEY 8(d) fraction the line carries, for the headline kernel and for every leg, what actually
    # moves (frac_io_only), what the HBM counters saw as a fraction of peak (frac_traffic, where a PMC profile is committed) and the
    # measured bound; driven from this round's full record (profiles/r06f_bench_detail.json)
    bench = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r06f_bench_detail.json")))
    line = bench.compact_line(full)
    assert len(line.encode()) < 4096, len(line)
    d = json.loads(line)
    r = d["roofline"]
    for k in ("frac", "frac_io_only", "frac_traffic", "valu_issue_frac", "bound_measured", "traffic", "bound"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["bound_measured"] == "valu-issue"
    assert r["frac_io_only"] < r["frac_traffic"] < 0.05 < r["frac"]   # 0.2 % / 0.6 % of peak actually moved; the nominal pricing says 0.8
    assert abs(r["frac_traffic"] - r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9 / r["peak"]) < 1e-4
    for k, leg in d["configs"].items():
        assert "frac_io_only" in leg["roofline"] and "bound_measured" in leg["roofline"], k
    assert "frac_traffic" in d["configs"]["4"]["roofline"]
    e = d["e2e"]
    for k in ("ingest_loci_per_s", "ingest_loci_per_s_host", "pipeline_loci_per_s", "pipeline_loci_per_s_host_ingest", "gpu_loci_per_s", "write_loci_per_s"):
        assert e[k] > 0, k
    assert e["ingest_loci_per_s"] > 2 * e["ingest_loci_per_s_host"] and e["pipeline_loci_per_s"] > 2 * e["pipeline_loci_per_s_host_ingest"] and e["pipeline_vcf_identical"]


def test_contract_line_without_legs_or_baseline():
    bench = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r04last_bench_default.json")))
    for k in ("configs", "e2e", "cpu_baseline", "parity", "cpu_baseline_all_cores"):
        full.pop(k)
    full["multi_gpu_digest_check"] = {"ranks": 2, "shards_recomputed_on_another_gpu": 2, "digest_mismatches": 0}
    d = json.loads(bench.compact_line(full))
    assert d["multi_gpu_digest_check"]["digest_mismatches"] == 0 and "configs" not in d and d["config"]["loci_per_gpu"] == 10000
