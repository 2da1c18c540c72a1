"""bench.py's output contract (no GPU): the LAST stdout line is one compact JSON object the driver can parse from
an 8 KB tail -- round 2 printed 24 KB and the record came back `parsed: null`."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def canned():
    alg = bench.ALG_BYTES_PER_DECISION * bench.BATCH
    rf = bench.roofline_entry("ev::k_eval_sorted", 0.0481234567, alg, 0.0581234567, 118.2e6,
                              {"avg_ms": "HIP events ...", "traffic": "profiles/r03_v1_uniform_fixed_pmc.json"})
    one = {"value": 17.123456789e9, "unit": "decisions/s", "ms_per_step": 0.06123456, "allowed_fraction": 0.18123,
           "whole_step_frac": 0.0771234, "roofline": rf, "detail": {"stages": {"x": list(range(1000))}}}
    return {
        "metric": "GCRA decisions/sec, 10M keys", "value": 18.04e9, "unit": "decisions/s", "n_gpus": 1, "steps": 20, "warmup": 5,
        "ms_per_step": 0.0581234567, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64",
        "data": "synthetic",
        "config": {"workload": "configs[1]: 10000000 pre-hashed keys SoA on 1 GPU, uniform request stream, batch=1048576, "
                               "params (100,1000/3600s), q=1, one timestamp per batch, decisions only",
                   "keys_per_gpu": 10_000_000, "batch": 1 << 20, "stream": "uniform", "resident_state": "fixed", "pipelined": True},
        "allowed_fraction": 0.999, "roofline": rf,
        "zipf_stream": dict(one), "wide_layout": dict(one), "general_uniform": dict(one), "general_zipf": dict(one),
        "per_key_plans": {f"{st}_{pl}": dict(one, plans=pl) for st in ("uniform", "zipf") for pl in ("tiers4", "tiers1000")},
        "verified": True, "verified_legs": 9, "verify_failed": [],
        "string_keys": {"key_%d": dict(one, launches_per_batch=6.0), "ascii_32_64": dict(one, launches_per_batch=6.0),
                        "workload": "a long description " * 20},
        "cpu_baseline": {"value": 1.81e6, "unit": "decisions/s", "cores": 1, "kind": "port", "sample": "first 8 batches " * 30,
                         "all_cores": {"value": 14.1e6, "cores": 16, "machine_cores": 256, "note": "x" * 500},
                         "reference_shape": {"value": 11.2e6, "unit": "decisions/s", "sample": "y" * 500,
                                             "published_by_reference": {"value": 12_500_000}}},
    }


REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def test_compact_line_is_small_and_complete():
    line = bench.compact_line(canned())
    assert "\n" not in line
    assert len(line) < 4096, len(line)
    d = json.loads(line)
    for k in REQUIRED:
        assert k in d, k
    for k in ("workload", "keys_per_gpu", "batch", "stream", "resident_state"):
        assert k in d["config"], k
    r = d["roofline"]
    for k in ("bound", "kernel", "avg_ms", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "whole_step_frac",
              "source"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert 0.0 < r["frac"] <= 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
    assert r["avg_ms"] <= d["ms_per_step"]                       # the kernel is part of the step
    assert abs(r["traffic_over_algorithmic"] - 118.2e6 / (36.125 * (1 << 20))) < 1e-3
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["all_cores"]["cores"] == 16 and cb["reference_shape"]["value"] == 11.2e6
    for k in ("zipf_stream", "wide_layout", "general_zipf"):
        assert set(("value", "whole_step_frac")) <= set(d[k]), k
    assert set(d["string_keys"]) == {"key_%d", "ascii_32_64"}
    assert "stages" not in line and "traffic_source" not in line  # the per-stage tables live in the detail file


def test_a_fraction_above_one_is_never_printed():
    alg = bench.ALG_BYTES_PER_DECISION * bench.BATCH
    r = bench.roofline_entry("bp::k_scatter", 0.0003, alg, 0.06, None, None)  # a launch that left at once
    assert r["frac"] is None and r["achieved"] is None and "invalid" in r
    res = canned()
    res["roofline"] = r
    d = json.loads(bench.compact_line(res))
    assert d["roofline"]["frac"] is None


def test_a_kernel_longer_than_its_step_is_flagged():
    alg = bench.ALG_BYTES_PER_DECISION * bench.BATCH
    r = bench.roofline_entry("rs::k_onesweep", 0.1085, alg, 0.0615, None, None)  # round 2's headline entry
    assert "invalid" in r


def test_oversized_results_shed_optional_parts_not_the_contract():
    res = canned()
    res["per_gpu"] = [{"rank": i, "share_of_traffic": 0.125, "decisions_per_s": 1e9, "junk": "z" * 300} for i in range(8)]
    res["config"]["workload"] = "w" * 2500
    line = bench.compact_line(res)
    assert len(line) <= 4096
    d = json.loads(line)
    for k in REQUIRED:
        assert k in d, k


def test_stage_table_gives_no_rate_to_partial_launches():
    st = {"bucket_scatter": {"kernel": "bp::k_scatter", "launches_per_batch": 0.05, "avg_ms": 0.02, "per_batch_ms": 0.001},
          "eval": {"kernel": "ev::k_eval_sorted", "launches_per_batch": 1.0, "avg_ms": 0.04, "per_batch_ms": 0.04}}
    out = bench.stage_table(st, 37.9e6, "uniform", "fixed")
    assert "achieved_GBs" not in out["bucket_scatter"] and "note" in out["bucket_scatter"]
    assert out["eval"]["achieved_GBs"] > 0


def test_compact_line_carries_the_oracle_verdict_and_the_per_key_plan_legs():
    d = json.loads(bench.compact_line(canned()))
    assert d["verified"] is True and d["verified_legs"] == 9 and d["verify_failed"] == []
    pk = d["per_key_plans"]
    assert set(pk) == {"uniform_tiers4", "uniform_tiers1000", "zipf_tiers4", "zipf_tiers1000"}
    for v in pk.values():
        assert v["value"] > 0 and v["ms_per_step"] > 0 and 0 < v["kernel_frac"] <= 1


def test_plan_tiers_are_valid_distinct_and_spread():
    import numpy as np
    from oracle import oracle as O
    for kind, T in (("tiers4", 4), ("tiers1000", 1000)):
        tiers, tier_of = bench.plan_tiers(kind, 200_000)
        assert tiers.shape == (T, 3) and len({tuple(r) for r in tiers.tolist()}) == T
        assert tuple(tiers[0]) == (100, 1000, 3600)
        assert tier_of.max() == T - 1 and np.bincount(tier_of, minlength=T).min() > 0
        for b, c, p in tiers.tolist():
            st, ei, dvt = O.derive(b, c, p, 0)
            assert st == 0 and b >= 2 and 0 < ei <= dvt < 2**60   # what TC_CFG_FIXED_PARAMS asks of a plan (tc::fixed_plan_ok)


def test_verify_run_accepts_the_oracle_and_rejects_a_flipped_decision():
    """the bench's checker itself, fed with an 'engine' that is the oracle (and then with one wrong byte / one wrong TAT)"""
    import numpy as np
    from oracle import oracle as O
    from throttlecrab_amd import workload as W
    keys, batch, nb = 5000, 4096, 6
    z = W.Zipf(keys)
    host = [z.slots(batch, start=i * batch) for i in range(4)]          # (the stream wraps: batch i uses host[i % 4])
    tiers, tier_of = bench.plan_tiers("tiers4", keys)
    per_slot = tiers[tier_of]
    orc = O.DenseOracle(keys)
    ring, allowed = {}, 0
    for i in range(nb):
        pr = per_slot[host[i % 4]]
        r = orc.batch_slots(host[i % 4], pr[:, 0], pr[:, 1], pr[:, 2], 1, W.T0_NS + i * 1_000_000)
        allowed += int(r.allowed.sum())
        if i >= nb - 3:
            ring[i] = r.allowed.copy()
    tat, exp, occ = orc.dump()
    exp = np.where(occ, exp, 0).astype(np.uint64)
    snap = {"batches": nb, "selfcheck": 0, "state": (tat, exp), "ring": ring,
            "counters": {"allowed": allowed, "denied": nb * batch - allowed, "errors": 0, "total": nb * batch}}
    v = bench.verify_run(snap, host, 0, per_slot, keys, batch, False)
    assert v["ok"] and v["batches_compared_bytewise"] == 3 and 0 < v["oracle_allowed"] < nb * batch
    bad = dict(snap, ring={k: a.copy() for k, a in ring.items()})
    bad["ring"][nb - 1][17] ^= 1
    assert not bench.verify_run(bad, host, 0, per_slot, keys, batch, False)["ok"]
    t2 = tat.copy()
    t2[int(np.flatnonzero(occ)[0])] += 1
    assert not bench.verify_run(dict(snap, state=(t2, exp)), host, 0, per_slot, keys, batch, False)["ok"]
    assert not bench.verify_run(dict(snap, counters=dict(snap["counters"], allowed=allowed - 1)), host, 0, per_slot, keys, batch, False)["ok"]


def test_gpus_flag_launches_the_ranks_itself():
    """`--gpus N` was parsed and ignored (ranks came from torchrun's environment only, so a plain
    `python bench.py --gpus 8` ran one rank and said n_gpus: 1).  Now: a plain start with N > 1 launches the N ranks
    (torch.distributed.run on 127.0.0.1), under torchrun the flag must agree with WORLD_SIZE, and the line's n_gpus is what the
    collective reached.  The plan is a pure function: checked here without a GPU (the real launch: tests/test_gpu_sharding.py)."""
    import argparse
    import subprocess

    def plan(gpus, argv, env):
        return bench.launch_plan(argparse.Namespace(gpus=gpus), argv, env)

    kind, cmd = plan(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], {})
    assert kind == "launch"
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert plan(1, ["--gpus", "1"], {}) == ("run", 1) and plan(None, [], {}) == ("run", 1)
    assert plan(8, ["--gpus", "8"], {"WORLD_SIZE": "8"}) == ("run", 8)       # the driver's torchrun line
    assert plan(None, [], {"WORLD_SIZE": "4"}) == ("run", 4)
    assert plan(8, [], {"WORLD_SIZE": "1"})[0] == "error" and plan(0, [], {})[0] == "error"
    assert plan(2, [], {"MASTER_PORT": "31234"})[1][cmd.index("--master-port") + 1] == "31234"
    # ... and through the command line (no GPU is touched by --plan-only)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plan-only", "--steps", "2"], capture_output=True,
                         text=True, timeout=120, env={k: v for k, v in os.environ.items() if k != "WORLD_SIZE"})
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["action"] == "launch" and "--nproc-per-node=2" in d["cmd"] and "--plan-only" not in d["cmd"]
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, WORLD_SIZE="2"))
    assert bad.returncode != 0 and "torchrun started 2" in bad.stderr
