"""N > 1 path with the real engine: two processes (both on cuda:0 -- the GPU box has one device), each handed the
GLOBAL stream, each keeping the requests it owns with the device partition kernel (tc_route_batch) and
deciding them on its own engine; the counter blocks and the top-denied blocks are all-gathered (gloo here, RCCL
in bench.py).  The union of the shards must equal one sequential pass of the oracle over the whole stream."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import throttlecrab_amd as t
    from throttlecrab_amd import sharded, workload as W
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    n_keys, n, nb = 20000, 60000, 4  # keys per shard; requests per global batch
    eng = t.Engine(n_keys, n, track_denied=True)
    eng.use_torch_stream()
    allowed_global = np.zeros(nb * n, np.int64)
    for b in range(nb):
        gids = W.Zipf(world * n_keys).slots(n, start=b * n)   # global key ids, the same stream on every rank
        d = torch.from_numpy(gids.astype(np.int32)).cuda()
        slots, pos, counts = eng.route_batch(d, world, only=rank, want_pos=True)   # the device partition kernel
        torch.cuda.synchronize()
        mine = int(counts[rank].item())
        res = eng.rate_limit_batch_slots(slots[:mine].contiguous(), max_burst=5, count_per_period=50, period=60, quantity=1,
                                         now_ns=W.T0_NS + b * 10**8, want=("allowed",), inputs_ready=True)
        torch.cuda.synchronize()
        allowed_global[b * n + pos[:mine].cpu().numpy()] = res.allowed.cpu().numpy()
    c = eng.counters()
    block = torch.tensor([c[k] for k in ("total", "allowed", "denied", "errors", "swept", "batches", "keys_inserted", "live_slots")],
                         dtype=torch.int64)
    per_rank, totals = sharded.all_gather_counters(block, dist, world)
    top = torch.from_numpy(sharded.pack_top_denied(eng.top_denied(sharded.TOPK), rank, world, n_keys))
    gathered = torch.zeros(world * sharded.TOPK, 2, dtype=torch.int64)
    dist.all_gather_into_tensor(gathered, top)
    tt = torch.from_numpy(allowed_global)
    dist.all_reduce(tt)
    if rank == 0:
        q.put((totals, per_rank.tolist(), tt.numpy(), sharded.merge_top_denied(gathered.numpy(), 10)))
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


def test_two_engines_two_processes_match_single_pass():
    import torch.multiprocessing as mp
    from oracle import oracle as O
    from throttlecrab_amd import workload as W
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    totals, per_rank, allowed, top = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n_keys, n, nb = 20000, 60000, 4
    orc = O.DenseOracle(2 * n_keys)  # one pass, keyed by the global id
    ref, denied = [], np.zeros(2 * n_keys, np.int64)
    for b in range(nb):
        gids = W.Zipf(2 * n_keys).slots(n, start=b * n)
        r = orc.batch_slots(gids, 5, 50, 60, 1, W.T0_NS + b * 10**8)
        ref.append(r.allowed.astype(np.int64))
        np.add.at(denied, gids[r.allowed == 0], 1)
    ref = np.concatenate(ref)
    assert top == sorted(((int(g), int(c)) for g, c in enumerate(denied) if c), key=lambda t: (-t[1], t[0]))[:10]
    assert np.array_equal(allowed, ref)
    assert totals["total"] == nb * n and totals["allowed"] == int(ref.sum())
    assert per_rank[0][0] + per_rank[1][0] == nb * n and min(per_rank[0][0], per_rank[1][0]) > 0.2 * nb * n


def _bench_line(args, env_extra, timeout=600):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("route", ["replicate", "exchange"])
def test_bench_gpus_flag_launches_that_many_ranks(route):
    """a plain `python bench.py --gpus 2` (no torchrun around it) must run TWO ranks and say so.  The box has one
    GPU: TC_BENCH_ONE_DEVICE=1 puts both ranks on it with gloo for the metrics -- the launcher, the process group, the router,
    the per-GPU table and the summary are the ones a real node runs."""
    d = _bench_line(["--gpus", "2", "--steps", "3", "--warmup", "2", "--no-cpu", "--keys", "200000", "--batch", "65536", "--route", route],
                    {"TC_BENCH_ONE_DEVICE": "1", "MASTER_PORT": str(_free_port())})
    assert d["n_gpus"] == 2 and len(d["per_gpu"]) == 2 and d["scaling"] == "weak" and d["steps"] == 3
    assert d["config"]["batch"] == 2 * 65536 and d["value"] > 0
    assert abs(sum(g["share_of_traffic"] for g in d["per_gpu"]) - 1.0) < 1e-6 and min(g["share_of_traffic"] for g in d["per_gpu"]) > 0.3


def test_bench_gpus_1_prints_the_single_gpu_line():
    d = _bench_line(["--gpus", "1", "--steps", "3", "--warmup", "2", "--no-cpu", "--no-also", "--keys", "200000", "--batch", "65536"], {})
    assert d["n_gpus"] == 1 and "per_gpu" not in d and d["config"]["batch"] == 65536 and d["verified"] is True
    for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
