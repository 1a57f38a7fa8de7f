"""The bench line's contract, checked on the lines kept under profiles/ (written by bench.py on the
GPU box): every key the driver and the reviewers read is there, with the right type, and the derived
figures are consistent with each other."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_single_gpu_line_carries_the_contract():
    d = _line("r03_bench_n1.json")
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], typ), key
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["unit"] == "edges/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # achieved = algorithmic bytes of one launch / its average duration, traffic = counter bytes of one launch
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["avg_ms"] * 1e-3) / 1e9) < 0.01 * r["achieved"]
    assert r["traffic"] is None or (isinstance(r["traffic"], int) and 0.3 * r["alg_bytes_per_launch"] < r["traffic"] < 1.5 * r["alg_bytes_per_launch"])
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "edges/s" and c["sample"]
    # value = entries / step time; the entry points' HIP-event times fit inside the step
    entries = int(d["config"]["workload"].split(" nodes, ")[1].split(" edge entries")[0])
    assert abs(d["value"] - entries / (d["ms_per_step"] * 1e-3)) < 1e-5 * d["value"]
    assert sum(d["kernels_ms"].values()) <= d["ms_per_step"] * 1.001


def test_multi_gpu_rehearsal_line_names_what_travelled():
    """The line of the driver's N > 1 command (rehearsed with two gloo ranks on one GPU): its value is the fastest
    fully timed exchange-mode leg, every leg is reported, and configs[4]'s per-rank share is timed in the same run."""
    d = _line("r03_rehearsal_gloo_2ranks_one_gpu.json")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["backend"] == "gloo" and d["config"]["name"] == "products"
    x = d["dist"]
    assert x["halo"] == "exchange" and x["world_size"] == 2 and x["n_halo"] > 0 and "exchange_error" not in x
    assert x["bytes_sent_per_step"] > 0 and x["exchange_ms"] > 0 and x["exchanges_per_step"] >= 2
    legs = x["legs"]
    assert {"exchange_sequential", "exchange_pipeline", "replicate"} <= set(legs) and all("error" not in v for v in legs.values())
    best = min(v["ms_per_step"] for k, v in legs.items() if k.startswith("exchange") and "trial" not in k)
    assert abs(d["ms_per_step"] - best) < 1e-3 and x["overlap"] in ("off", "pipeline", "two_phase")
    assert legs["exchange_sequential"]["bytes_sent_per_step"] == legs["exchange_pipeline"]["bytes_sent_per_step"]
    assert d["replicate_mode"]["bytes_sent_per_step"] == 0 and d["replicate_mode"]["value"] > 0
    c4 = d["configs4"]
    assert "12500000 nodes" in c4["workload"] and c4["world_size"] == 2 and "error" not in c4
    assert c4["denominators_one_gpu"]["whole_graph_one_gpu_ms"] == 280.3 and c4["denominators_one_gpu"]["share_alone_ms"] == 36.1
    assert c4["exchange"]["value"] > 0 and c4["replicate"]["value"] > 0 and c4["exchange"]["vs_whole_graph_on_one_gpu"] is None     # a ratio only at 8 ranks
    assert all("error" not in v for v in c4["legs"].values())
