"""bench.py's contract line (CPU): whatever the full record holds, what goes to stdout is ONE strict-JSON line under 4 KB
carrying every key the driver parses (round 4's 22 KB line outgrew the driver's capture and the round's
headline went unmeasured).  Fed with the full records of earlier rounds kept under profiles/ and with a bloated one."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline")


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def strict(text):
    def refuse(name):
        raise ValueError(name)
    return json.loads(text, parse_constant=refuse)


@pytest.mark.parametrize("record", ["r04_final_bench_256.json", "r04_final_bench_512.json", "r04_bench_rccl_loopback_256.json",
                                    "r03_final_bench_256.json"])
def test_contract_line_of_a_full_record(bench, record):
    full = json.loads(open(os.path.join(ROOT, "profiles", record)).read())
    assert len(json.dumps(full)) > 5000  # these are the long lines of rounds 3 and 4
    c = bench.contract_line(full, "gpurun_out/bench_full_n1.json")
    text = json.dumps(c, separators=(",", ":"), allow_nan=False)
    assert len(text.encode()) < bench.CONTRACT_LIMIT == 4096
    d = strict(text)
    for key in CONTRACT_KEYS:
        assert key in d, key
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in d["roofline"], key
    assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"] and d["roofline"]["frac"] == full["roofline"]["frac"]
    assert d["config"]["workload"] == full["config"]["workload"] and d["dtype"] == "f32" and d["data"] == "synthetic"
    if full.get("cpu_baseline"):
        assert d["cpu_baseline"]["value"] == full["cpu_baseline"]["value"] and d["cpu_baseline"]["cores"] == 1
        assert d["cpu_baseline"]["kind"] == "port" and "sample" in d["cpu_baseline"] and d["cpu_baseline"]["all_cores"]["cores"] >= 1


def test_contract_line_stays_short_and_strict_whatever_the_record_holds(bench):
    full = json.loads(open(os.path.join(ROOT, "profiles", "r04_final_bench_256.json")).read())
    # a record that grew: forty more progressive cases per size, prose everywhere, numbers that are not JSON
    for side in ("256", "512"):
        for k in range(40):
            full["progressive"][side][f"another_case_{k}"] = {"ms": 0.1 + k, "frac": 0.5, "what": "x" * 300}
    full["host_load"] = {f"{s}_{t}": full["host_load"]["256"] for s in (64, 128, 256, 512, 1024) for t in ("a", "b", "c")}
    full["pipeline_plain"]["ms_fill"] = float("inf")
    full["roofline"]["traffic"] = float("nan")
    full["roofline_raymarch"]["frac"] = float("-inf")
    full["watchdog"] = "stuck " * 200
    c = bench.contract_line(full, "/somewhere/" + "long/" * 20 + "bench_full_n1.json")
    text = json.dumps(c, separators=(",", ":"), allow_nan=False)
    assert len(text.encode()) < 4096
    d = strict(text)
    for key in CONTRACT_KEYS + ("cpu_baseline",):
        assert key in d, key
    assert d["roofline"]["traffic"] is None and "progressive" not in d  # dropped: the least important block goes first
    assert len(d["watchdog"]) <= 160
    # and the full record itself is written as strict JSON too (emit() maps non-finite numbers to null)
    assert strict(json.dumps(bench._finite(full), allow_nan=False))["pipeline_plain"]["ms_fill"] is None
