"""N>1 path on CPU: world_size-2 gloo ranks shard the packs, build their partial states (the
oracle stands in for the device kernels, which need a GPU), all-gather them with the product's
dist plumbing and merge with the product's knox_agg_combine; the result must equal the oracle
over the whole table."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import knoxdb_amd as K
    from knoxdb_amd import dist as kdist
    from oracle import oracle as O
    from test_abi_surface import _state_from_oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pack, n_packs = 1024, 10
    first, last = kdist.shard_range(n_packs, rank, world)
    # every rank generates only its own packs (counter-based generator: no data exchange)
    x = O.generate(O.I64, 0, 2, first * pack, (last - first) * pack, lo=-(1 << 40), span=1 << 41)
    bits, cnt = O.cmp(O.I64, O.RG, x, -(1 << 38), 1 << 38)
    st = _state_from_oracle(O.I64, K.BlockType.Int64, x, bits, first * pack)
    merged, total = kdist.allgather_states([st], cnt)
    # the preallocated form bench.py uses every step must give the same merge (twice: buffers are reused)
    g = kdist.StateGather(1)
    for _ in range(2):
        m2, t2 = g([st], cnt)
        assert t2 == total and bytes(m2[0]) == bytes(merged[0])
    # the north star's literal collective for the order-free integer aggregates: ONE all-reduce of {sum, count}
    # (two 32-bit halves of the wrapping sum + the count) -- must equal the all-gather + combine bit for bit
    red = kdist.SumCountReduce()
    for _ in range(2):
        gsum, gcnt = red(int(st.isum), cnt)
        assert gcnt == total and gsum == int(merged[0].isum), (gsum, int(merged[0].isum))
    # the split form bench.py keeps in flight behind the next scan: two exchanges outstanding, read in order
    h0 = red.start(int(st.isum), cnt)
    h1 = red.start(7 + rank, 3)
    assert red.finish(h0) == (int(merged[0].isum), total)
    assert red.finish(h1) == (sum(7 + r for r in range(world)), 3 * world)
    # ... also when the partial sums sit at the wrap-around edge
    edge = (1 << 64) - 1 - rank
    esum, ecnt = red(edge, 1 << 40)
    assert esum == sum((1 << 64) - 1 - r for r in range(world)) % (1 << 64) and ecnt == world << 40
    res = {f: K.agg_value(merged[0], f) for f in (K.ReducerFunc.Sum, K.ReducerFunc.Min, K.ReducerFunc.Max,
                                                   K.ReducerFunc.First, K.ReducerFunc.Last)}
    q.put((rank, total, {int(k): (int(v[0]), v[1], v[2]) for k, v in res.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_pack_sharded_aggregate_matches_whole_table():
    from oracle import oracle as O
    import knoxdb_amd as K
    world, port = 2, 29500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    x = O.generate(O.I64, 0, 2, 0, 10 * 1024, lo=-(1 << 40), span=1 << 41)
    bits, cnt = O.cmp(O.I64, O.RG, x, -(1 << 38), 1 << 38)
    want = {int(K.ReducerFunc.Sum): O.reduce(O.I64, O.R_SUM, x, bits), int(K.ReducerFunc.Min): O.reduce(O.I64, O.R_MIN, x, bits),
            int(K.ReducerFunc.Max): O.reduce(O.I64, O.R_MAX, x, bits), int(K.ReducerFunc.First): O.reduce(O.I64, O.R_FIRST, x, bits),
            int(K.ReducerFunc.Last): O.reduce(O.I64, O.R_LAST, x, bits)}
    for rank, total, res in outs:
        assert total == cnt
        for f, (v, seen, row) in res.items():
            assert (v, seen) == (int(want[f][0]), want[f][1]), (rank, f)
            if f != int(K.ReducerFunc.Sum):
                assert row == want[f][2], (rank, f)


def test_shard_range_partitions_exactly():
    from knoxdb_amd.dist import shard_range
    for n in (1, 7, 8, 1024, 1025):
        for w in (1, 2, 3, 8):
            rs = [shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n and all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))


def test_bench_gpus_flag_launches_that_many_ranks():
    """`bench.py --gpus 2` with no launcher around it starts two rank processes itself (fresh interpreters, the parent
    never touches the GPU); with the gloo rehearsal backend they bring the process group up even on a box without a GPU,
    then refuse to run: there is no CPU fallback."""
    import subprocess
    env = dict(os.environ, KNOX_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--rows-log2", "20", "--no-cpu-baseline"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert "[rank 0] process group up: backend gloo, 2 rank(s)" in r.stderr, r.stderr[-2000:]
    assert "[rank 1] process group up: backend gloo, 2 rank(s)" in r.stderr, r.stderr[-2000:]
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "no HIP device visible" in r.stderr and r.stdout.strip() == ""


def test_bench_abi_collective_reaches_the_id_exchange_in_every_rank():
    """`bench.py --gpus 2 --collective abi`: the per-step exchange is the library's own RCCL communicator (knox_comm_*, what
    a Go host calls).  Rank 0 makes the ncclUniqueId, the process group carries its 128 bytes to every rank, then each rank
    calls knox_comm_init on its device.  Here (gloo rehearsal, possibly no GPU) both children must get as far as holding
    the id; on a box without a device they then refuse to run (no CPU fallback), with one they bring the communicator up."""
    import subprocess
    env = dict(os.environ, KNOX_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--collective", "abi", "--steps", "1", "--warmup", "1",
                        "--rows-log2", "20", "--no-cpu-baseline", "--no-configs"], env=env, capture_output=True, text=True, timeout=300)
    for rank in (0, 1):
        assert "[rank %d] abi collective: unique id exchanged (128 bytes)" % rank in r.stderr, r.stderr[-2000:]
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "no HIP device visible" in r.stderr and r.stdout.strip() == ""
