"""N > 1 path on CPU: world_size-2 gloo run of the sharding + count all-reduce logic that bench.py uses on RCCL.  Each rank
generates its own shard of the synthetic workload and verifies it (the C oracle stands in for the GPU kernel here)."""
import importlib
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

shard = importlib.import_module("ps-signature-and-el-passo_amd.shard")


def test_shard_ranges_cover_exactly():
    for n in (0, 1, 7, 64, 65536, 1000003):
        for world in (1, 2, 3, 8):
            seen = 0
            for r in range(world):
                s, c = shard.shard_range(n, r, world)
                assert s == seen
                seen += c
            assert seen == n


def _worker(rank, world, port, n_total, out):
    import ctypes
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    from elp_testlib import OracleBackedCtx, oracle
    synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = oracle()
    ctx = OracleBackedCtx()
    A, H = 3, 2
    wl = synth.Workload(ctx, A)
    key = ctx.key_handle()

    def verify_fn(start, count):
        recs, mask, expect = wl.verify_id_batch(count, H, first_item=start, corrupt_every=3, corrupt_at=1)
        rsz = len(recs) // max(count, 1)
        flags = np.array([L.elpo_verify_id(key, recs[i * rsz:(i + 1) * rsz], mask, 1, b"hello", 5) for i in range(count)], dtype=np.uint8)
        assert list(flags) == list(expect)
        return flags, int(flags.sum())

    start, flags, local, total = shard.verify_sharded(verify_fn, n_total, rank, world, dist)
    out.put((rank, start, len(flags), local, total))
    dist.destroy_process_group()


def test_two_rank_gloo_count_reduce():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctxmp = mp.get_context("spawn")
    q = ctxmp.Queue()
    n_total = 7
    procs = [ctxmp.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # items 0..6, corrupted where n % 3 == 1 -> items 1, 4 rejected -> 5 accepted
    assert [r[1] for r in res] == [0, 4] and [r[2] for r in res] == [4, 3]
    assert sum(r[3] for r in res) == 5
    assert all(r[4] == 5 for r in res)


def _run_bench(*argv, timeout=600):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), capture_output=True, text=True, timeout=timeout, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus_flag_starts_the_ranks_dry_run():
    """`python bench.py --gpus 2` (the driver's command shape, no torchrun around it) must itself start 2 ranks, rendezvous on
    127.0.0.1, shard the batch, all-reduce the count and have rank 0 print ONE line with n_gpus = 2 -- here on gloo (--dry-run)."""
    r, out = _run_bench("--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-3000:]
    assert out is not None and out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["dry_run"] is True
    assert out["reduced_count"] == 2 * 65536 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert len([ln for ln in r.stdout.splitlines() if ln.startswith("{")]) == 1
    r, out = _run_bench("--gpus", "2", "--dry-run", "--config", "5")
    assert r.returncode == 0 and out["config"]["attrs"] == 16 and out["config"]["batch_per_gpu"] == 131072 and out["reduced_count"] == 2 * 131072


def test_bench_without_gpu_fails_loudly_after_spawning():
    """Without --dry-run there is no CPU path: the spawned ranks must fail (no GPU in this container), and so must the launcher."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    r, out = _run_bench("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0 and out is None
    assert "starting 2 ranks" in r.stderr


def test_dry_run_prints_the_per_rank_memory_budget():
    """`bench.py --gpus 8 --config 5 --dry-run` (what the driver can run before an 8-GPU node exists): every rank's budget -- 28.5 GB of W = 20 tables for the
    16-attribute key + launch workspace + its 131 072-proof shard -- is computed from the table geometry and fits the 288 GB of one MI355X ten times over."""
    import importlib
    shard = importlib.import_module("ps-signature-and-el-passo_amd.shard")
    b5 = shard.rank_memory_budget("bn254", 16, 4, 20, 131072)
    assert b5["tables_bytes"] == 13 * (1 << 19) * (22 * 72 + 18 * 144) and 28.4e9 < b5["tables_bytes"] < 28.5e9 and b5["fits"] and b5["fraction_of_hbm"] < 0.11
    b4 = shard.rank_memory_budget("bn254", 8, 4, 20, 65536)
    assert abs(b4["tables_bytes"] / 2**30 - 15.54) < 0.01            # the 15.5 GiB bench.py reports from elp_key_table_bytes
    assert b4["records_bytes"] == 65536 * 800
    bls = shard.rank_memory_budget("bls12_381", 8, 4, 20, 65536)
    assert abs(bls["tables_bytes"] / 2**30 - 24.17) < 0.01           # tools/probes/verify_probe.py: table_GiB=24.17
