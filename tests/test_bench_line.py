"""The one stdout line of bench.py stays small enough for the driver to parse (a 20 KB line was recorded
as parsed = null).  Built from a committed full record (profiles/r04_final_bench.json, the line that broke the parse)
and from a worst-case record with every figure at full float precision."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "rccl_ranks_seen")
ROOF = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "valu_busy_profiled", "profile")
CPU = ("value", "unit", "cores", "kind", "sample", "cpu_model", "outputs_match", "outputs_compared")
COMPOSITE = ("bls12381_pairings_per_s", "bls12381_pair_checks_per_s", "bls12381_g1_msm_2p20_s", "bls12381_g1_msm_2p20_s_affine",
             "bls12381_g1_msm_2p20_s_checked", "bls12381_g1_commit_2p20_s", "bn256_pairings_per_s")


def _full_record():
    return json.load(open(os.path.join(ROOT, "profiles", "r04_final_bench.json")))


def _check(line):
    s = json.dumps(line, separators=(",", ":"))
    assert len(s) < bench.LINE_LIMIT, len(s)
    assert "\n" not in s
    for k in REQUIRED:
        assert k in line, k
    for k in ROOF:
        assert k in line["roofline"], k
    for k in CPU:
        assert k in line["cpu_baseline"], k
    assert len(line["cpu_baseline"]) <= 10
    assert all(not isinstance(v, (dict, list)) for v in line["cpu_baseline"].values())
    assert all(not isinstance(v, (dict, list)) for v in line["roofline"].values())
    for k in COMPOSITE:
        assert isinstance(line[k], float), k
    for k in ("bls12381_pairings_per_s", "bls12381_g1_msm_2p20_s", "bn256_pairings_per_s", "bls12381_g1_commit_2p20_s"):
        assert 0 < line[k + "_frac"] < 1, k
    assert line["config"]["workload"].startswith("Ed25519 batched fixed-base + var-base")
    return s


def test_line_from_the_record_that_broke_the_parse():
    full = _full_record()
    assert len(json.dumps(full)) > 4 * bench.LINE_LIMIT  # the full record is what no longer fits
    s = _check(bench.compact_line(full))
    back = json.loads(s)
    assert back["value"] == full["value"] and back["ms_per_step"] == full["ms_per_step"]
    assert abs(back["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-3


def test_line_worst_case_precision_and_missing_legs():
    full = _full_record()

    def widen(o):
        if isinstance(o, dict):
            return {k: widen(v) for k, v in o.items()}
        if isinstance(o, float):
            return o * 1.0000001234567891
        return o

    _check(bench.compact_line(widen(full)))
    # legs skipped by flags (--no-other / --no-cpu-baseline, N > 1): still a valid, small line with the headline keys
    for drop in (("other_workloads",), ("cpu_baseline",), ("other_workloads", "cpu_baseline")):
        rec = {k: v for k, v in full.items() if k not in drop}
        line = bench.compact_line(rec)
        assert len(json.dumps(line)) < bench.LINE_LIMIT
        assert line["value"] == full["value"] and line["roofline"]["frac"] is not None


def test_emit_prints_one_line_and_writes_detail(tmp_path, capsys, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(_full_record())
    out = capsys.readouterr().out
    assert out.count("\n") == 1 and len(out) < bench.LINE_LIMIT
    assert json.loads(out)["detail_file"] == "bench_detail.json"
    assert "other_workloads" in json.load(open(tmp_path / "bench_detail.json"))


def test_an_oversized_line_loses_optional_keys_not_the_run():
    """emit() asserted the length after the whole benchmark had run.  A record whose optional parts are too long
    (error strings, a long CPU model) still yields a parseable line with the contract keys, marked `truncated`."""
    full = _full_record()
    line = bench.compact_line(full)
    line["other_workloads_error"] = "x" * 3000
    line["cpu_baseline"]["cpu_model"] = "y" * 1500
    s = bench.fit_line(line)
    back = json.loads(s)
    assert len(s) < bench.LINE_LIMIT and back["truncated"] is True
    for k in REQUIRED:
        assert k in back, k
    assert back["value"] == full["value"] and back["roofline"]["frac"] is not None
    assert "truncated" not in json.loads(bench.fit_line(bench.compact_line(full)))


def test_full_batch_digest_checks_reach_the_line():
    full = _full_record()
    ok = {"outputs_match": True, "outputs_compared": 1 << 16, "cpu_per_s": 9000.0}
    full.setdefault("cpu_baseline", {}).setdefault("other_workloads", {})["full_batch_digests"] = {
        "bls12381": {"pair": ok, "g1_mul": ok, "g2_mul": dict(ok, outputs_match=False)},
        "bn256": {"error": "boom"},
        "bls12381_g1_msm_2p20": {"outputs_match": True, "outputs_compared": 1 << 20, "cpu_seconds": 3.5}}
    c = bench.compact_line(full)["checks"]
    assert c["bls12381_pair_full"] == 1 << 16 and c["bls12381_g1_mul_full"] == 1 << 16 and c["bls12381_g2_mul_full"] is False
    assert c["msm_2p20_full"] == 1 << 20 and c["full_error"].startswith("bn256: boom")
