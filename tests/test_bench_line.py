"""The contract line of bench.py stays small enough for a record that keeps only the tail of stdout (round 3's 18.8 KB
line lost BASELINE configs 3-5 there), and still carries every config's roofline and cpu_baseline."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _records():
    # full records: round 3's line was the full record itself; from round 4 on bench.py writes it to bench_detail.json
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "r03_bench_all.json")) + glob.glob(os.path.join(ROOT, "profiles", "r0[4-9]_bench_detail*.json")))


@pytest.mark.parametrize("path", _records(), ids=os.path.basename)
def test_contract_line_is_compact_and_complete(path):
    import bench
    with open(path) as fh:
        full = json.load(fh)
    line = bench.compact_line(full)
    assert len(line) < bench.LINE_LIMIT
    assert "\n" not in line
    assert '"cqt"' in line[-2000:]
    out = json.loads(line)
    assert list(out)[-1] == "configs"
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(out["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    for kind in ("istft", "mel", "mfcc", "mdct", "imdct", "cqt"):
        e = out["configs"][kind]
        assert e["roofline"]["frac"] > 0 and e["roofline"]["kernel"].startswith("k_")
        assert e["cpu_baseline"]["cores"] >= 1 and e["parity"]["within_tolerance"] is True
    if os.path.basename(path) >= "r05":
        # verdict r4 item 3: SURVEY 8(d)'s roofs -- mel / mfcc priced on HBM with the two arithmetic pipes beside it (never summed over one
        # peak), cqt on the f32 vector pipe with both flop counts; the untimed pre-warm launches are in the line
        assert isinstance(out["prewarm_launches"], int)
        for kind in ("mel", "mfcc"):
            r = out["configs"][kind]["roofline"]
            assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
            assert 0 < r["mfma_frac"] < 1 and 0 < r["valu_frac"] < 1
            assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        r = out["configs"]["cqt"]["roofline"]
        assert r["bound"] == "valu" and r["unit"] == "TFLOP/s" and 0 < r["valu_frac_real_input"] < r["frac"] < 1 and 0 < r["hbm_frac"] < 0.1


def test_line_sheds_optional_keys_before_it_outgrows_the_limit():
    import bench
    with open(_records()[-1]) as fh:
        full = json.load(fh)
    full["extras"] = {f"extra{i}": dict(next(iter(full["extras"].values()))) for i in range(120)}
    line = bench.compact_line(full)
    assert len(line) < bench.LINE_LIMIT and '"cqt"' in line[-2000:]
