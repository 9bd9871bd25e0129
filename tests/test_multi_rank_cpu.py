"""The N > 1 path on CPU: two gloo ranks, each owning a row shard of the continuous batch and stepping its own
engine; per step the ranks all-gather their decoder outputs (min_llm_inference_amd/sharding.py -- the same code
bench.py uses over RCCL).  Checks: the gathered token matrix of every step equals what the two shards produce
when run in one process, and per-item outputs equal an unsharded run (rows are independent)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

B_TOTAL, S, D, V, N_ITEMS, WORLD = 6, 64, 32, 1024, 14, 2


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from engine_sim import CpuEngine, make_items, make_model
    from min_llm_inference_amd.sharding import TokenGather, shard_bounds, shard_items
    model = make_model(71, V, S, D)
    items = make_items(72, N_ITEMS, 1, 24)
    lo, hi = shard_bounds(B_TOTAL, rank, world)
    eng = CpuEngine(oracle, model, shard_items(items, rank, world), hi - lo, S)
    gather = TokenGather(hi - lo, world, torch.device("cpu"))
    steps = []
    while True:
        local = torch.from_numpy(eng.step() if not eng.done() else np.full((hi - lo,), -1, np.int32))
        gather.buffer().copy_(local)   # the engine's decoder output for this step
        gather()                       # asynchronous all-gather; overlaps the next step on GPUs
        gather.wait()
        steps.append(gather.latest().clone().numpy())
        flag = torch.tensor([0 if eng.done() else 1])
        dist.all_reduce(flag)  # keep stepping until every rank is done (lockstep, as bench.py does)
        if flag.item() == 0:
            break
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), steps=np.stack(steps),
             ids=np.array(sorted(eng.finished)), **{f"item{k}": v for k, v in eng.finished.items()})
    dist.destroy_process_group()


def test_two_rank_row_sharding_gloo(tmp_path):
    import oracle
    from engine_sim import CpuEngine, make_items, make_model, run_cpu_engine
    from min_llm_inference_amd.sharding import shard_bounds, shard_items
    oracle.lib()  # build before forking
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(WORLD, port, str(tmp_path)), nprocs=WORLD, join=True)
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(WORLD)]
    assert (r[0]["steps"] == r[1]["steps"]).all()          # every rank holds the same gathered tokens

    # single-process replay of the two shards
    model = make_model(71, V, S, D)
    items = make_items(72, N_ITEMS, 1, 24)
    engines = []
    for k in range(WORLD):
        lo, hi = shard_bounds(B_TOTAL, k, WORLD)
        engines.append(CpuEngine(oracle, model, shard_items(items, k, WORLD), hi - lo, S))
    for step in r[0]["steps"]:
        expect = np.concatenate([e.step() if not e.done() else np.full((e.B,), -1, np.int32) for e in engines])
        assert (step == expect).all()
    assert all(e.done() for e in engines)

    # sharding does not change any item's output
    whole, _ = run_cpu_engine(oracle, model, items, B_TOTAL, S)
    got = {}
    for k in range(WORLD):
        for item_id in r[k]["ids"]:
            got[int(item_id)] = r[k][f"item{int(item_id)}"]
    assert sorted(got) == sorted(whole)
    for item_id in whole:
        assert (got[item_id] == whole[item_id]).all()


def test_shard_bounds_cover_all_rows():
    from min_llm_inference_amd.sharding import shard_bounds
    for n, w in ((8192, 8), (1024, 3), (5, 8), (1, 1)):
        spans = [shard_bounds(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
