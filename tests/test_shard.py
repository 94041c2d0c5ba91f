"""Multi-rank sharding logic on CPU (gloo, world_size 2): the N>1 path of bench.py and of any
batch front-end.  No GPU and no codec involved: images are represented by their pixel counts and the
"encode" of a rank is the oracle, so this also checks that a sharded job reproduces the unsharded
result image for image."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from seqoia_amd import shard


def test_contiguous_blocks_cover_everything():
    for n in (0, 1, 7, 1024, 65536):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                seen += list(shard.contiguous(n, world, r))
            assert seen == list(range(n))
            sizes = [len(shard.contiguous(n, world, r)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_lpt_balances_suite_shaped_sizes():
    # SURVEY.md Appendix C suite-shaped size table (pixels per image), replicated
    rng = np.random.default_rng(0)
    sizes = [64 * 64] * 226 + [512 * 512] * 214 + [36000] * 1010 + [640000] * 619 + [1800000] * 187 + [8100000] * 14
    rng.shuffle(sizes)
    parts = shard.lpt(sizes, 8)
    assert sorted(i for p in parts for i in p) == list(range(len(sizes)))
    loads = [sum(sizes[i] for i in p) for p in parts]
    assert max(loads) / (sum(loads) / 8) < 1.02  # within 2 % of perfect balance


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as po

    shapes = [(64, 64, 3), (100, 37, 4), (33, 5, 4), (128, 16, 3), (17, 17, 4), (256, 8, 4), (5, 5, 3)]
    mine = shard.lpt([w * h for w, h, _ in shapes], world)[rank]
    lens = []
    for i in mine:
        w, h, ch = shapes[i]
        lens.append(len(po.restatement().encode(po.synth(w, h, ch, "mixed", 3, i), w, h, ch, 0)))
    allv = shard.gather_lengths(mine, lens, len(shapes))
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the timing reduction bench.py does
    if rank == 0:
        q.put((allv, float(t.item())))
    dist.destroy_process_group()


def test_two_ranks_gloo_reproduce_single_rank():
    from oracle import pyoracle as po

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    allv, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shapes = [(64, 64, 3), (100, 37, 4), (33, 5, 4), (128, 16, 3), (17, 17, 4), (256, 8, 4), (5, 5, 3)]
    want = [len(po.restatement().encode(po.synth(w, h, ch, "mixed", 3, i), w, h, ch, 0)) for i, (w, h, ch) in enumerate(shapes)]
    assert allv == want
    assert tmax == 2.0


def _bench(*argv):
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("n", [2, 3])
def test_bench_gpus_flag_starts_that_many_ranks(n):
    """`python bench.py --gpus N` must itself start N ranks (RANK / WORLD_SIZE / MASTER_* in their environment) and
    relay rank 0's line; --dry-run makes the ranks rendezvous over gloo and report their shards without a GPU."""
    out = _bench("--gpus", str(n), "--dry-run")
    assert out["n_gpus"] == n and out["scaling"] == "weak" and out["images_rank0"] == 1024 and out["images_total"] == 1024 * n
    out = _bench("--gpus", str(n), "--dry-run", "--scaling", "strong")
    assert out["n_gpus"] == n and out["scaling"] == "strong" and out["images_total"] == 1024
    assert out["images_rank0"] == len(shard.contiguous(1024, n, 0))


def test_launcher_stops_the_other_ranks_when_one_dies():
    """A rank that dies before the rendezvous must not leave the others waiting for it until some time-out: the launcher
    polls all children, terminates the rest and exits non-zero at once."""
    import subprocess
    import sys
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SQOA_BENCH_FAULT_RANK"] = "1"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--dry-run"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "(1, 3)" in r.stderr, r.stderr
    assert time.time() - t0 < 60


def test_bench_under_an_external_launcher_is_a_rank():
    """Under torch.distributed.run the environment is already there: bench.py must not start ranks of its own."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--config", "c5"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # one line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"] == "c5" and out["scaling"] == "strong" and out["images_total"] == 65536


def test_c5_suite_table_and_lpt_shards():
    """BASELINE.json configs[4]: the suite-shaped table replicated to 65 536 images and
    LPT-sharded over 8 ranks on w*h: every image exactly once, pixel loads within 1 % of each other."""
    from seqoia_amd import suite

    rep = suite.replica()
    assert len(rep) == 2871
    px = sum(w * h for _, w, h, _, _ in rep)
    assert abs(px - 1.32e9) / 1.32e9 < 0.01  # bench1-noaverage.txt:118-123: about 1.32 Gpx
    imgs = suite.images()
    assert len(imgs) == 65536 and [im.index for im in imgs] == list(range(65536))
    assert {im.channels for im in imgs} == {3, 4} and max(im.pixels for im in imgs) == 3840 * 2160 and min(im.pixels for im in imgs) == 64 * 64
    parts = [suite.shard(65536, 8, r) for r in range(8)]
    assert sorted(im.index for p in parts for im in p) == list(range(65536))
    loads = [sum(im.pixels for im in p) for p in parts]
    assert max(loads) / min(loads) < 1.01
