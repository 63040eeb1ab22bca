"""bench.py end to end on the GPU at a tiny size: one JSON line with the contract's keys, the roofline and cpu_baseline
objects, every leg present and parity-checked where the leg carries a check."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_legs_at_a_tiny_size():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batches-per-step", "8",
                        "--cpu-seconds", "1", "--pgpr-users", "8", "--pgpr-steps", "1", "--sharded-steps", "2"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["unit"] == "triples/s" and d["value"] > 1e8 and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert 0.2 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["configs1_launch"]["bound"] == "l2" and r["configs1_launch"]["frac"] <= 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "triples/s"
    for leg in ("sharded", "tucker", "reference_layout", "configs0", "evaluation", "pgpr", "cafe"):
        assert leg in d and "error" not in d[leg], (leg, d[leg].get("error"))
    assert d["sharded"]["parity"] and abs(d["sharded"]["parity"]["sharded_loss"] - d["sharded"]["parity"]["plain_loss"]) < 1e-5
    assert d["tucker"]["parity"]["loss_relation_sets"] == d["tucker"]["parity"]["loss_per_row_product"]
    assert d["pgpr"]["roofline"]["frac"] > 0 and d["pgpr"]["parity"]["missing_all_at_ties"] is True
    # BASELINE-size parity, inside this suite (not only in the driver's bench line):
    # configs[1] -- every score of a full 2048 x 257 RotatE d=200 launch of the timed region against oracle/kge_ref.c, its loss
    # against float64, every sampled negative against the used sets
    par = d["parity"]
    assert par["triples"] == 2048 * 257 and par["max_rel_score_err"] < 1e-4
    assert par["one_launch_scores_equal_plain_scorer"] is True and par["loss"]["rel_err"] < 1e-4
    assert par["negatives"]["drawn"] == 2048 * 256 and par["negatives"]["in_item_range"] and par["negatives"]["in_used_set"] == 0
    # configs[2] -- the one-user 64/64/64 leaf proof on the ml-1m-shaped graph (~250 k leaves), rule of tests/test_gpu_pgpr.py
    pg = d["pgpr"]["parity"]
    assert pg["users"] == 1 and pg["oracle_leaves"] > 200_000 and pg["device_leaves"] >= pg["common"] > 200_000
    assert pg["safe_leaves"] <= pg["common"]  # every safe oracle leaf was emitted
    # CAFE -- the users of a 1 024-user call the oracle walked: same items
    cf = d["cafe"]["parity"]
    assert cf["users"] >= 1 and cf["common"] >= 0.9 * cf["oracle_items"] > 0 and cf["max_rel_score_diff"] < 1e-4
    # the other scorers' HBM launches (>= 4 GB tables): present and sane
    oth = d["roofline"]["other_scorers"]
    for tag in ("transe_d64", "transe_d200", "distmult_d200", "complex_d256"):
        assert oth[tag]["bound"] == "hbm" and 0.3 < oth[tag]["frac"] <= 1.0 and oth[tag]["table_GB"] >= 4.0
    ev = d["evaluation"]
    for tag in ("ml1m_2985_items", "yelp2018_45539_items"):
        assert ev[tag]["parity"]["columns_equal_where_scores_differ"] is True and ev[tag]["topk_product_form_ms"] > 0
