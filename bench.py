#!/usr/bin/env python3
"""bench.py -- EL PASSO RP verifications/s (el_passo_verify_id, 8 attributes, 4 hidden) on N MI355X.

One "step" = one pass of the hot path (the fused verify_id kernel) over one batch of synthetic proofs per GPU (records already
resident in HBM), followed by the RCCL count all-reduce.  Shards are independent (weak scaling).
  --config 4 (default): 65 536 proofs per GPU, 8 attributes, 4 hidden   (BASELINE.json configs[3], the metric's configuration)
  --config 5          : 131 072 proofs per GPU, 16 attributes, 4 hidden (BASELINE.json configs[4]: 2^20 proofs over 8 GPUs)
Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` and `cpu_baseline` objects.

Launching: under torchrun (RANK / LOCAL_RANK / WORLD_SIZE set) every process is one rank.  Started plainly as
`python bench.py --gpus N` with N > 1 it starts N ranks itself -- as a CHILD `python -m torch.distributed.run` process, before
anything here touches the GPU -- and exits with that child's code.  `--dry-run` exercises the same launch / rendezvous / shard /
count-reduce / report path on CPU (gloo, no kernel, value = null): it is what tests/test_dist_gloo.py runs in the GPU-less container.
"""
import argparse
import ctypes
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "ps-signature-and-el-passo_amd"
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec


CONFIGS = {4: {"batch": 65536, "attrs": 8, "hidden": 4}, 5: {"batch": 131072, "attrs": 16, "hidden": 4}}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=4, choices=sorted(CONFIGS), help="BASELINE.json configuration (4 = the metric's, 5 = 16 attributes, 131 072 per rank)")
    ap.add_argument("--batch", type=int, default=0, help="proofs per GPU per step (default: the configuration's)")
    ap.add_argument("--attrs", type=int, default=0)
    ap.add_argument("--hidden", type=int, default=0)
    ap.add_argument("--dry-run", action="store_true", help="CPU / gloo run of the launch, shard, count-reduce and report path; no kernel, value = null")
    ap.add_argument("--window", type=int, default=-1, help="fixed-base window bits of the key tables (library default 8; 20 = 16 GiB of signed-digit tables "
                    "for the 8-attribute key, 13 table additions per scalar instead of 16 at W = 16: profiles/r02_window_sweep.json); reported in `config` and `key_tables`")
    ap.add_argument("--curve", default="bn254", choices=["bn254", "bls12_381"], help="curve of the headline run")
    ap.add_argument("--no-second-curve", action="store_true", help="skip the secondary BLS12-381 measurement at N=1")
    ap.add_argument("--only", default="", help="comma-separated secondary sections to run beside the headline (pcie, w16, config5, aggregated, secondary, host_api, bls); default: all")
    ap.add_argument("--child-section", default="", help="(internal) run ONE secondary section in this process and print its JSON: aggregated | host_api")
    ap.add_argument("--headline-only", action="store_true", help="only the headline workload (profiling runs: every k_verify_id launch has the headline size)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="items timed on the CPU oracle (0 disables, -1 = max(4096, 256 x cores))")
    args = ap.parse_args(argv)
    if args.window < 0:      # default table width: 20 bits (15.5 GiB) for the configuration the metric is quoted on, 16 bits (2.5 GiB at 16 attributes) for config 5:
        args.window = 16 if args.config == 5 else 20      # an RP with several IdP keys cannot afford 28.5 GB per key and rank; --window 20 opts in
    cfg = CONFIGS[args.config]
    args.batch = args.batch or cfg["batch"]
    args.attrs = args.attrs or cfg["attrs"]
    args.hidden = args.hidden or cfg["hidden"]
    if args.headline_only:
        args.no_second_curve, args.cpu_sample = True, 0
    args.only = set(x for x in args.only.split(",") if x)
    if args.only and "bls" not in args.only:
        args.no_second_curve = True
    return args


def launch_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a child torchrun (this process has not touched the GPU and
    never will) and hand back its exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print("bench.py: starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr)
    return subprocess.call(cmd, env=env)


def dry_run(args, rank, world):
    """The N > 1 control path without a GPU: gloo rendezvous, contiguous shards, barrier-bracketed timed loop, count all-reduce (SUM),
    max-over-ranks time, one JSON line from rank 0.  No verification runs here (there is no CPU fallback of the kernel): each rank
    contributes the size of its shard as its "accepted" count, so the reduced total must equal world x batch."""
    import torch
    import torch.distributed as dist
    shard = importlib.import_module(PKG + ".shard")
    if world > 1:
        dist.init_process_group(backend="gloo")
    first, count = shard.shard_range(world * args.batch, rank, world)
    cnt = torch.zeros(1, dtype=torch.int64)

    def step():
        cnt.fill_(count)
        if world > 1:
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ranks = torch.ones(1, dtype=torch.int64)
    if world > 1:
        tdt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
        dt = float(tdt.item())
        dist.all_reduce(ranks, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({"metric": "EL PASSO credential verifications/sec (8 attrs)", "value": None, "unit": "verifications/s",
                          "n_gpus": world, "ranks_seen": int(ranks.item()), "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "u32", "data": "synthetic", "dry_run": True, "backend": "gloo" if world > 1 else "none",
                          "config": workload_config(args, world), "reduced_count": int(cnt.item()),
                          "shard": {"first": first, "count": count},
                          "per_rank_memory_budget": shard.rank_memory_budget(args.curve, args.attrs, args.hidden, args.window, args.batch)}))
    if world > 1:
        dist.destroy_process_group()
    return 0 if int(cnt.item()) == world * args.batch and int(ranks.item()) == world else 4


def workload_config(args, world):
    A, H, B = args.attrs, args.hidden, args.batch
    return {"workload": "BASELINE.json config %d: batch of %d EL PASSO RP el_passo_verify_id per GPU, %d attributes with %d hidden, id-retrieval, "
                        "curve %s" % (args.config, B, A, H, "BN254 (the reference's actual mcl default; golden-vector pinned)" if args.curve == "bn254"
                                      else "BLS12-381 (north-star curve; pinned by vectors of the reference's own wasm run on this curve, tests/golden/bls12_381_*.json)"),
            "baseline_config": args.config, "batch_per_gpu": B, "attrs": A, "hidden": H, "curve": args.curve.upper(), "window_bits": args.window or 8,
            "parallelism": "independent shards x%d + RCCL count all-reduce" % world}


def main():
    args = parse_args()
    in_group = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus > 1 and not in_group:
        sys.exit(launch_ranks(args))
    if in_group and int(os.environ["WORLD_SIZE"]) != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%s: launch one rank per GPU" % (args.gpus, os.environ["WORLD_SIZE"]), file=sys.stderr)
        sys.exit(2)
    if args.dry_run:
        sys.exit(dry_run(args, int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))))
    if args.child_section:
        sys.exit(child_section(args))

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks_seen = 1
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=dev)      # "nccl" is RCCL on ROCm
        rs = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(rs, op=dist.ReduceOp.SUM)
        ranks_seen = int(rs.item())

    pkg = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")
    shard = importlib.import_module(PKG + ".shard")
    curve_id = pkg.CURVE_BN254 if args.curve == "bn254" else pkg.CURVE_BLS12_381
    ctx = pkg.Context(curve_id, local_rank)
    A, H, B = args.attrs, args.hidden, args.batch
    t_setup = time.time()
    wl = synth.Workload(ctx, A, seed=20211, window_bits=args.window)
    first, count = shard.shard_range(world * B, rank, world)          # weak scaling: B items per rank
    recs, mask, expect = wl.verify_id_batch(count, H, first_item=first, with_retrieval=True)
    t_setup = time.time() - t_setup
    rsz = len(recs) // B
    host = np.frombuffer(recs, dtype=np.uint8)
    d_rec = torch.from_numpy(host.copy()).to(dev)
    d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
    d_flags = torch.zeros(B, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    kern_events = []      # (start, end) HIP events around every timed launch, on the stream the kernel is launched on (= torch's current stream)

    def step(timed=False):
        d_cnt.zero_()
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        ctx._chk(ctx.lib.elp_verify_id_batch_dev(ctx.h, stream, B, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad),
                                                 d_flags.data_ptr(), d_cnt.data_ptr()))
        if timed:
            e1.record()
            kern_events.append((e0, e1))
        if world > 1:
            dist.all_reduce(d_cnt, op=dist.ReduceOp.SUM)     # the only collective: accepted-count over xGMI

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(timed=True)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tdt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
        dt = float(tdt.item())
    total_accepted = int(d_cnt.item())
    flags = d_flags.cpu().numpy()
    shard_ok = bool((flags == expect).all())
    exp_total = torch.tensor([int(expect.sum())], dtype=torch.int64, device=dev)
    ok_all = torch.tensor([1 if shard_ok else 0], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(exp_total, op=dist.ReduceOp.SUM)
        dist.all_reduce(ok_all, op=dist.ReduceOp.MIN)
    parity_ok = bool(ok_all.item()) and total_accepted == int(exp_total.item())

    # dominant-kernel duration: average over the launches of the timed region itself, HIP events on the launch stream
    if kern_events:
        kern_ms = sum(a.elapsed_time(b) for a, b in kern_events) / len(kern_events)
    else:
        ms = ctypes.c_float()
        ctx._chk(ctx.lib.elp_time_verify_id_dev(ctx.h, stream, 1, B, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None,
                                                len(wl.ad), d_flags.data_ptr(), d_cnt.data_ptr(), ctypes.byref(ms)))
        kern_ms = float(ms.value)
    algo_bytes_per_item = rsz + 4                   # affine inputs + 4-byte verdict (804 B at A=8, BN254)
    achieved = B * algo_bytes_per_item / (kern_ms * 1e-3) / 1e9
    traffic, traffic_source = None, None
    tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")     # per-launch HBM bytes from the round's rocprofv3 --pmc passes (not measured in this run)
    if os.path.exists(tf) and args.config == 4 and args.curve == "bn254":
        try:
            tj = json.load(open(tf))
            traffic = tj.get("k_verify_id_bytes_per_launch")
            traffic_source = "profiles/hbm_traffic.json (%s): FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE of the round's separate rocprofv3 --pmc passes over the same command; not re-measured in this run" % tj.get("tag", "round PMC pass")
        except Exception:
            traffic = None

    out = None
    if rank == 0:
        value = world * B * args.steps / dt
        out = {
            "metric": "EL PASSO credential verifications/sec (8 attrs)",
            "value": value, "unit": "verifications/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": workload_config(args, world), "ranks_seen": ranks_seen,
            "parity_ok": parity_ok, "accepted": total_accepted, "expected_accepted": int(exp_total.item()),
            "setup_s": t_setup,
            "key_tables": {"window_bits": args.window or 8, "table_bytes": ctx.key_table_bytes(), "table_GiB": ctx.key_table_bytes() / 2.0**30,
                           "set_pubkey_ms": wl.t_set_pubkey_ms, "set_rp_and_secret_ms": wl.t_set_params_ms,
                           "note": "signed-digit fixed-base window tables of the key's 14 G1 and 10 G2 bases, per key and per GPU; built on the GPU (k_window_bases, k_table_fill)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source, "kernel": "k_verify_id_staged (k_verify_id with coalesced record loads, ELP_OPT_COALESCED_RECORDS default)", "kernel_ms": kern_ms, "algorithmic_bytes_per_item": algo_bytes_per_item,
                         "note": "integer-VALU bound path: see valu_bound"},
        }
        out["valu_bound"] = valu_bound(ctx, "verify_id" if (args.curve == "bn254" and A == 8 and H == 4) else None, args.window, B, kern_ms, 162)

    want = lambda sec: not args.only or sec in args.only      # noqa: E731
    if rank == 0 and world == 1 and not args.headline_only and want("pcie"):     # profiling runs: every verification dispatch of the process is a timed-region dispatch
        # PCIe-inclusive rate (host buffers in, flags out: elp_verify_id_batch, the path PSVerifier::el_passo_verify_id_batch takes; pinned
        # staging, chunks copied and verified on several streams) -- reported, never the headline value.  One warm-up call, median of 7.
        ts = []
        for it in range(8):
            t1 = time.perf_counter()
            fl2, cnt2 = ctx.verify_id_batch(recs, mask, True, wl.ad)
            ts.append(time.perf_counter() - t1)
        ts = sorted(ts[1:])
        out["pcie_inclusive_value"] = B / ts[len(ts) // 2]
        out["pcie_inclusive"] = {"value": B / ts[len(ts) // 2], "unit": "verifications/s", "median_ms": ts[len(ts) // 2] * 1e3, "min_ms": ts[0] * 1e3,
                                 "max_ms": ts[-1] * 1e3, "calls": len(ts), "parity_ok": bool((fl2 == expect).all()) and cnt2 == int(expect.sum()),
                                 "note": "host records in, verdicts out through elp_verify_id_batch (includes the Python wrapper's buffer handling)"}
        if args.cpu_sample != 0 and args.curve == "bn254":
            ncore = usable_cores()
            samp = args.cpu_sample if args.cpu_sample > 0 else max(4096, 256 * ncore)
            out["cpu_baseline"] = cpu_baseline(wl, ctx, recs, rsz, mask, flags, min(samp, B))
    if rank == 0 and world == 1 and args.curve == "bn254" and not args.no_second_curve:
        try:
            out["bls12_381"] = second_curve(pkg, synth, local_rank, dev, A, H, B, args.window)
        except Exception as e:  # pragma: no cover
            out["bls12_381"] = {"error": str(e)}
    if rank == 0 and world == 1 and args.curve == "bn254" and args.config == 4 and not args.headline_only and (args.window or 8) != 16 and want("w16"):
        try:   # the same workload on the 16-bit tables (1/16 of the memory): what the wide tables buy
            out["w16"] = other_config(pkg, synth, local_rank, dev, 4, 16)
            out["value_w16"] = out["w16"]["value"]              # the same workload on 1.2 GiB of tables (1/13 of the headline's): what a key-per-IdP deployment can afford
            out["ms_per_step_w16"] = out["w16"]["kernel_ms"]
        except Exception as e:  # pragma: no cover
            out["w16"] = {"error": str(e)}
    if rank == 0 and world == 1 and args.curve == "bn254" and args.config == 4 and not args.headline_only and want("config5"):
        try:   # BASELINE.json config 5 at N = 1: one rank's share (131 072 proofs, 16 attributes) on this GPU
            out["config5_rank_share"] = other_config(pkg, synth, local_rank, dev, 5, 16)        # W = 16: 2.5 GiB per rank (W = 20: 28.5 GB, opt-in via --config 5 --window 20)
        except Exception as e:  # pragma: no cover
            out["config5_rank_share"] = {"error": str(e)}
    if rank == 0 and world == 1 and not args.headline_only and want("aggregated"):
        out["aggregated"] = run_child_section(args, "aggregated")
    if rank == 0 and world == 1 and args.curve == "bn254" and (not args.no_second_curve or "secondary" in args.only) and want("secondary"):
        try:
            out["secondary"] = secondary_workloads(pkg, synth, local_rank, dev, args.window)
        except Exception as e:  # pragma: no cover
            out["secondary"] = {"error": str(e)}
    if rank == 0 and world == 1 and args.curve == "bn254" and args.config == 4 and not args.headline_only and want("host_api"):
        # the reference-API path: std::vector<IdProof> / wire messages through the C++ PSVerifier (key set-up excluded and reported)
        out["host_api"] = run_child_section(args, "host_api")
    if rank == 0:
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    if not parity_ok:
        sys.exit(3)


def run_child_section(args, name):
    """Sections that use streams of their own (two side streams for the pipelined aggregated batches, the C++ verifier's context streams) run in a PROCESS of their
    own: private-memory (scratch) blocks are per hardware queue, and in a process whose default stream has already run 2^20-item launches of the large-frame kernels
    every further queue makes the runtime reclaim blocks between queues -- seconds per event (profiles/r04_scratch_stall.md).  The child builds its own context and
    workload (same seed, same window width) and prints the section's JSON."""
    cmd = [sys.executable, os.path.abspath(__file__), "--child-section", name, "--config", str(args.config), "--window", str(args.window), "--curve", args.curve]
    # Under a profiler (rocprofv3 preloads its tool library into every child) the child's 2^20-item dispatches of OTHER kernels would land in the same counter
    # directory, distort the per-kernel averages and -- serialised under --pmc -- run into the time-out: profile the headline with --headline-only, and skip the
    # child sections when a profiler preload is detected anyway (round-4 advisor finding).
    if any(os.environ.get(k) for k in ("ROCPROFILER_LIBRARY_PATH", "ROCPROF_OUTPUT_PATH", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD")) or \
            "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return {"skipped": "profiler environment detected: child sections are not run under rocprofv3 (use --headline-only / --only for profiling passes)"}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        line = [x for x in r.stdout.strip().splitlines() if x.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": "child section %s failed (rc %d): %s" % (name, r.returncode, r.stderr[-400:])}
        res = json.loads(line[-1])
        res["process"] = "child process of bench.py (own context and streams; see run_child_section)"
        return res
    except Exception as e:  # pragma: no cover
        return {"error": str(e)}


def child_section(args):
    import torch
    pkg = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = pkg.Context(pkg.CURVE_BN254 if args.curve == "bn254" else pkg.CURVE_BLS12_381, 0)
    A, H, B = args.attrs, args.hidden, args.batch
    if args.child_section == "aggregated":
        wl = synth.Workload(ctx, A, seed=20211, window_bits=args.window)
        res = aggregated_section(ctx, wl, synth, dev, (1 << H) - 1, H, B)
    elif args.child_section == "host_api":
        wl = synth.Workload(ctx, A, seed=20211, window_bits=8)          # the C++ verifier builds its own tables at args.window; this context only synthesises the proofs
        recs, mask, expect = wl.verify_id_batch(B, H, with_retrieval=True)
        res = host_api(pkg, wl, recs, B, A, H, 0, expect, args.window, 0)
    else:
        res = {"error": "unknown section"}
    print(json.dumps(res))
    ctx.close()
    return 0


def aggregated_section(ctx, wl, synth, dev, mask, H, B):
    """Aggregated (random-linear-combination) verification beside the per-item kernel -- reported, never `value`.  All measurements run on DISTINCT proofs (2^20 of
    them, synthesised on the device by the batch prover from 65 536 credentials presented 16 times with fresh randomness: synth.distinct_proofs_dev), not on tiled
    copies of one batch.  Single calls at 65 536 / 262 144 / 1 048 576 items, and the SUSTAINED rate of 65 536-item batches pipelined over two streams of the one
    context (batch i on stream A, batch i + 1 on stream B; each stream has its own workspaces): the serial tail of a batch -- Fp12 product, Pippenger sum of the
    sig2's, one Miller loop + final exponentiation, ~2.5 ms on a handful of workgroups -- runs beside the per-item kernel of the next batch."""
    import numpy as np
    import torch
    main = torch.cuda.current_stream()
    NT = 1 << 20
    t0 = time.perf_counter()
    d_all, amask, exp_all = wl.distinct_proofs_dev(NT, H, B, dev, main.cuda_stream)
    assert amask == mask
    t_gen = time.perf_counter() - t0
    rsz = d_all.numel() // NT
    d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
    seed_buf = np.frombuffer(bytes((7 * i + 1) & 0xFF for i in range(32)), dtype=np.uint8).copy()
    fl = torch.zeros(NT, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(4, dtype=torch.int64, device=dev)
    res = {"note": "elp_verify_id_batch_aggregated_dev: per-item NIZK + one Miller loop, Pippenger MSM of the sig2's, one final exponentiation per batch; exact "
                   "per-item fallback when the batch equation fails.  Distinct proofs throughout (not tiled copies).",
           "distinct_proofs": NT, "proof_synthesis_s": t_gen}

    def call(agg, stream, first, n, cslot=0):
        rp, fp, cp = d_all.data_ptr() + first * rsz, fl.data_ptr() + first, cnt.data_ptr() + 8 * cslot
        if agg:
            ctx._chk(ctx.lib.elp_verify_id_batch_aggregated_dev(ctx.h, stream, n, rp, mask, 1, d_ad.data_ptr(), None, len(wl.ad), seed_buf.ctypes.data, fp, cp))
        else:
            ctx._chk(ctx.lib.elp_verify_id_batch_dev(ctx.h, stream, n, rp, mask, 1, d_ad.data_ptr(), None, len(wl.ad), fp, cp))

    # the multi-stream measurements come FIRST, while the process has not yet run a 2^20-item launch on its default stream (see run_child_section)
    # sustained: the 16 distinct batches of 65 536, twice over, alternating between two streams (aggregated) / back to back on one stream (per item)
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=-1)      # the second one from the other priority pool: surely another hardware queue
    nbat = NT // B
    # Exactly two side streams.  What overlaps is the TAIL of one batch with the tail of the other (a per-item kernel holds every SIMD's registers, so a tail cannot run
    # beside it: profiles/r04_agg_two_stream_timeline.txt, r06_agg_two_stream_timeline.txt), and only if the two streams sit on two hardware queues.  More streams do not help and run into the runtime's
    # scratch reclaim between queues -- 80-295 ms per batch measured with three and four (profiles/r04_scratch_stall.md).
    for label, agg, streams in (("aggregated_two_streams", True, (sa, sb)), ("aggregated_one_stream", True, (sa,)), ("per_item_one_stream", False, (sa,))):
        for s_ in streams:
            call(agg, s_.cuda_stream, 0, B)             # warm-up: this stream's workspaces
        torch.cuda.synchronize()
        fl.zero_()
        cnt.zero_()
        torch.cuda.synchronize()                         # the side streams do not wait for the default stream
        t0 = time.perf_counter()
        for rep in range(2):
            for b in range(nbat):
                s_ = streams[b % len(streams)]
                call(agg, s_.cuda_stream, b * B, B, cslot=b % len(streams))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ok = bool((fl.cpu().numpy() == exp_all).all()) and int(cnt.sum().item()) == 2 * int(exp_all.sum())
        res["sustained_%d_%s" % (B, label)] = {"value": 2 * NT / dt, "unit": "verifications/s", "ms_per_batch": dt / (2 * nbat) * 1e3, "batches": 2 * nbat,
                                                "streams": len(streams), "parity_ok": ok}
    # fallback in the middle of the pipeline: one batch carries a forged signature that passes its NIZK half (sig2 of another item) -- its batch equation fails,
    # the per-item fallback of THAT batch decides; the batches around it stay on the fast path; every verdict stays exact
    G1 = ctx.G1
    d_bad = d_all[5 * B * rsz:6 * B * rsz].clone()
    d_bad[7 * rsz + G1:7 * rsz + 2 * G1] = d_all[(5 * B + 8) * rsz + G1:(5 * B + 8) * rsz + 2 * G1]      # item 7 of the batch gets item 8's sig2
    exp_bad = exp_all[5 * B:6 * B].copy()
    exp_bad[7] = 0
    fl.zero_()
    cnt.zero_()
    flb = torch.zeros(B, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(8):
        s_ = (sa, sb)[b % 2]
        if b == 3:
            ctx._chk(ctx.lib.elp_verify_id_batch_aggregated_dev(ctx.h, s_.cuda_stream, B, d_bad.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad),
                                                                seed_buf.ctypes.data, flb.data_ptr(), cnt.data_ptr() + 16))
        else:
            call(True, s_.cuda_stream, b * B, B, cslot=b % 2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    okf = bool((flb.cpu().numpy() == exp_bad).all()) and int(cnt[2].item()) == int(exp_bad.sum())
    for b in range(8):
        if b != 3:
            okf = okf and bool((fl[b * B:(b + 1) * B].cpu().numpy() == exp_all[b * B:(b + 1) * B]).all())
    res["fallback_mid_pipeline"] = {"parity_ok": okf, "ms_for_8_batches": dt * 1e3, "note": "batch 3 of 8 fails its batch equation (one swapped sig2) and is decided per item"}
    # single calls, on the first side stream: every large-frame launch of this process stays on the two hardware queues that already hold scratch blocks
    for nb in (B, 4 * B, NT):
        for agg in (True, False):
            call(agg, sa.cuda_stream, 0, nb)          # warm-up (workspaces)
            torch.cuda.synchronize()
            fl.zero_()
            cnt.zero_()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(sa)
            for _ in range(2):
                call(agg, sa.cuda_stream, 0, nb)
            e1.record(sa)
            torch.cuda.synchronize()
            ms_b = e0.elapsed_time(e1) / 2
            ok = bool((fl[:nb].cpu().numpy() == exp_all[:nb]).all()) and int(cnt[0].item()) == 2 * int(exp_all[:nb].sum())
            res["batch_%d_%s" % (nb, "aggregated" if agg else "per_item")] = {"value": nb / (ms_b * 1e-3), "ms_per_batch": ms_b, "parity_ok": ok}
    return res


def valu_frac(ctx, ops_key, window, items, kern_ms):
    """valu_bound of a secondary workload: counted Montgomery-product equivalents per item (profiles/op_counts.json[ops_key]) x items / kernel time against the
    fp_mul micro-benchmark peak of this lease (BN254 limbs)."""
    try:
        ops = json.load(open(os.path.join(ROOT, "profiles", "op_counts.json"))).get(ops_key, {})
        key = "W%d" % (window or 8)
        if key not in ops:
            return {"note": "no op count for %s at %s" % (ops_key, key)}
        fm = ctypes.c_float()
        lanes, iters = 256 * 4 * 64 * 8, 1000
        ctx._chk(ctx.lib.elp_bench_fp_mul(ctx.h, lanes, iters, ctypes.byref(fm)))
        peak = lanes * iters * 2 / (fm.value * 1e-3)
        per_item = ops[key]["fp_mul_equivalents"]
        ach = per_item * items / (kern_ms * 1e-3)
        return {"fp_mul_equivalents_per_item": per_item, "achieved": ach, "fp_mul_peak_per_s": peak, "frac": ach / peak, "unit": "modmul/s",
                "op_count_source": "profiles/op_counts.json[%s][%s] (counted)" % (ops_key, key)}
    except Exception as e:  # pragma: no cover
        return {"error": str(e)}


def valu_bound(ctx, ops_key, window, B, kern_ms, macs_per_mul):
    """Secondary ceiling (the binding one: the path is integer-VALU bound, not HBM bound): Montgomery products/s of the kernel against the
    fp_mul micro-benchmark of the same limb code at full occupancy, + the instruction-issue reading of the round's PMC pass."""
    try:
        fm = ctypes.c_float()
        lanes, iters = 256 * 4 * 64 * 8, 1000
        ctx._chk(ctx.lib.elp_bench_fp_mul(ctx.h, lanes, iters, ctypes.byref(fm)))
        peak = lanes * iters * 2 / (fm.value * 1e-3)
        vb = {"fp_mul_peak_per_s": peak, "unit": "modmul/s",
              "note": "peak = Montgomery products/s of the fp_mul micro-benchmark (same limb code, 8 waves/SIMD; %d multiply-adds each); achieved = "
                      "multiply-adds per verification / %d (profiles/op_counts.json: COUNTED on the host twin of the kernel code, sparse tables at this "
                      "window width) x verifications/s of the kernel" % (macs_per_mul, macs_per_mul)}
        oc = os.path.join(ROOT, "profiles", "op_counts.json")
        if ops_key and os.path.exists(oc):
            ops = json.load(open(oc)).get(ops_key, {})
            key = "W%d" % (window or 8)
            if key in ops:
                per_item = ops[key]["fp_mul_equivalents"]
                ach = per_item * B / (kern_ms * 1e-3)
                vb.update({"fp_mul_equivalents_per_verification": per_item, "multiply_adds_per_verification": ops[key]["multiply_adds"],
                           "op_count_source": "profiles/op_counts.json[%s][%s]%s" % (ops_key, key, " (extrapolated)" if ops[key].get("extrapolated") else " (counted)"),
                           "achieved": ach, "frac": ach / peak})
        import glob
        import re
        cands = sorted((q for q in glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")) if re.match(r"r\d+_summary\.json$", os.path.basename(q))),
                       key=lambda q: int(re.match(r"r(\d+)_", os.path.basename(q)).group(1)))
        pj = cands[-1] if cands else ""                   # the newest round's PMC summary
        if ops_key == "verify_id" and pj:
            try:
                pm = json.load(open(pj))
                iv, waves = pm["pmc_per_launch"]["SQ_INSTS_VALU"], pm["pmc_per_launch"]["SQ_WAVES"]
                vb["issue_model"] = {"valu_wave_instructions_per_launch": iv, "waves": waves, "source": "profiles/%s (round PMC pass, W = %s)" % (os.path.basename(pj), pm.get("window")),
                                     "ns_per_valu_instruction_per_wave": kern_ms * 1e6 / (iv / waves),
                                     "note": "one resident wave per SIMD: the wave issues one vector instruction every ~5 cycles whatever its type "
                                             "(profiles/r01_ubench_valu.log), so kernel time = instructions per wave x issue interval; see DESIGN.md section 5"}
            except Exception:
                pass
        return vb
    except Exception as e:  # pragma: no cover
        return {"error": str(e)}


def second_curve(pkg, synth, local_rank, dev, A, H, B, window):
    """Same workload on the BLS12-381 instantiation (12-word field, M-type twist): a secondary, shorter measurement."""
    import numpy as np
    import torch
    ctx = pkg.Context(pkg.CURVE_BLS12_381, local_rank)
    wl = synth.Workload(ctx, A, seed=20211, window_bits=window)
    recs, mask, expect = wl.verify_id_batch(B, H, with_retrieval=True)
    d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
    d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
    d_flags = torch.zeros(B, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    ms = ctypes.c_float()
    for reps in (1, 3):      # warm-up launch, then 3 timed launches (HIP events)
        d_cnt.zero_()
        ctx._chk(ctx.lib.elp_time_verify_id_dev(ctx.h, stream, reps, B, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad),
                                                d_flags.data_ptr(), d_cnt.data_ptr(), ctypes.byref(ms)))
    torch.cuda.synchronize()
    flags = d_flags.cpu().numpy()
    rsz = len(recs) // B
    ach = B * (rsz + 4) / (ms.value * 1e-3) / 1e9
    res = {"value": B / (ms.value * 1e-3), "unit": "verifications/s", "batch": B, "kernel_ms": float(ms.value),
           "parity_ok": bool((flags == expect).all()) and int(d_cnt.item()) == 3 * int(expect.sum()),
           "algorithmic_bytes_per_item": rsz + 4,
           "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                        "kernel": "k_verify_id_paired<Paired<BLS12_381>>", "kernel_ms": float(ms.value)},
           "valu_bound": valu_bound(ctx, "verify_id_bls12_381" if (A == 8 and H == 4) else None, window, B, float(ms.value), 392),
           "note": "BLS12-381 instantiation (14 limbs of 28 bits).  Parity: the reference's own wasm run on this curve (oracle/wasm_curve.js) produced "
                   "tests/golden/bls12_381_*.json; model, C oracle, host build of the device code and the GPU reproduce every verdict "
                   "(tests/test_oracle_bls_golden.py, tests/test_gpu_bls_golden.py); hashAndMapToG1 is mcl's (SHA-512 setHashOf, SvdW, cofactor)"}
    tf = os.path.join(ROOT, "profiles", "hbm_traffic_bls12_381.json")      # per-launch HBM bytes of the round's separate --pmc passes over this kernel
    if os.path.exists(tf):
        try:
            tj = json.load(open(tf))
            res["roofline"]["traffic"] = tj.get("bytes_per_launch")
            res["roofline"]["traffic_source"] = "profiles/hbm_traffic_bls12_381.json (%s): FETCH_SIZE x2 + WRITE_SIZE of separate rocprofv3 --pmc passes; not re-measured in this run" % tj.get("tag")
        except Exception:
            pass
    try:     # CPU baseline of this curve: the C oracle's BLS12-381 build on a bounded sample, verdicts compared with the GPU's
        ncore = usable_cores()
        res["cpu_baseline"] = cpu_baseline(wl, ctx, recs, rsz, mask, flags, min(B, max(768, 24 * ncore)), bls=True)
        res["cpu_baseline"]["kind"] = "port"
        res["cpu_baseline"]["note"] = "the C oracle's BLS12-381 build (reference structure), pinned to the reference's wasm run on this curve by tests/test_oracle_bls_golden.py"
    except Exception as e:  # pragma: no cover
        res["cpu_baseline"] = {"error": str(e)}
    try:     # aggregated verification of the same batch (round 6: main kernel on lane pairs, k_verify_id_agg_paired); the last of four calls timed with HIP events
        d_afl = torch.zeros(B, dtype=torch.uint8, device=dev)
        agg_ms = []
        for _ in range(4):
            d_cnt.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx._chk(ctx.lib.elp_verify_id_batch_aggregated_dev(ctx.h, stream, B, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad), None, d_afl.data_ptr(),
                                                                d_cnt.data_ptr()))
            e1.record()
            torch.cuda.synchronize()
            agg_ms.append(e0.elapsed_time(e1))
        am = min(agg_ms[1:])
        res["aggregated_65536" if B == 65536 else "aggregated_%d" % B] = {
            "value": B / (am * 1e-3), "unit": "verifications/s", "ms_per_batch": am, "per_item_ms_per_batch": float(ms.value),
            "parity_ok": bool((d_afl.cpu().numpy() == expect).all()) and int(d_cnt.item()) == int(expect.sum()),
            "kernels": "k_verify_id_agg_paired (two lanes per item) -> k_fp12_reduce16 -> Pippenger -> k_agg_final_coop; exact per-item fallback inside the call"}
    except Exception as e:  # pragma: no cover
        res["aggregated"] = {"error": str(e)}
    ctx.close()
    try:     # BASELINE config 2's shape on this curve: 4 096 PS verifications, A = 3 -- the row-of-16 pairing check (round 6) against the interpreter
        ctx = pkg.Context(pkg.CURVE_BLS12_381, local_rank)
        wl3 = synth.Workload(ctx, 3, seed=20211, window_bits=16)
        n = 4096
        precs, pexpect = wl3.ps_verify_batch(n)
        d_prec = torch.from_numpy(np.frombuffer(precs, dtype=np.uint8).copy()).to(dev)
        d_pfl = torch.zeros(n, dtype=torch.uint8, device=dev)
        out = {}
        for mode, tag in ((1, "row_of_16"), (0, "interpreter")):
            ctx.set_pair16(mode)
            f = lambda: ctx._chk(ctx.lib.elp_ps_verify_batch_dev(ctx.h, stream, n, d_prec.data_ptr(), 3, d_pfl.data_ptr(), d_cnt.data_ptr()))
            f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                f()
            e1.record()
            torch.cuda.synchronize()
            out["ms_" + tag] = e0.elapsed_time(e1) / 3
            out["parity_ok"] = out.get("parity_ok", True) and bool((d_pfl.cpu().numpy() == pexpect).all())
        out["value"] = n / (out["ms_row_of_16"] * 1e-3)
        out["unit"] = "verifications/s"
        res["ps_verify_4096x3attrs"] = out
        ctx.close()
    except Exception as e:  # pragma: no cover
        res["ps_verify_4096x3attrs"] = {"error": str(e)}
    return res


def other_config(pkg, synth, local_rank, dev, config, window):
    """Another BASELINE.json configuration on this GPU (BN254), device-resident records, HIP-event timing of the kernel."""
    import numpy as np
    import torch
    cfg = CONFIGS[config]
    A, H, B = cfg["attrs"], cfg["hidden"], cfg["batch"]
    ctx = pkg.Context(pkg.CURVE_BN254, local_rank)
    wl = synth.Workload(ctx, A, seed=20211, window_bits=window)
    recs, mask, expect = wl.verify_id_batch(B, H, with_retrieval=True)
    d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
    d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
    d_flags = torch.zeros(B, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    ms = ctypes.c_float()
    for reps in (1, 3):
        d_cnt.zero_()
        ctx._chk(ctx.lib.elp_time_verify_id_dev(ctx.h, stream, reps, B, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad),
                                                d_flags.data_ptr(), d_cnt.data_ptr(), ctypes.byref(ms)))
    torch.cuda.synchronize()
    flags = d_flags.cpu().numpy()
    rsz = len(recs) // B
    ach = B * (rsz + 4) / (ms.value * 1e-3) / 1e9
    res = {"workload": "BASELINE.json config %d, one rank's share: %d el_passo_verify_id, %d attributes with %d hidden, id-retrieval, BN254" % (config, B, A, H),
           "value": B / (ms.value * 1e-3), "unit": "verifications/s", "batch": B, "kernel_ms": float(ms.value), "window_bits": window,
           "table_bytes": ctx.key_table_bytes(), "set_pubkey_ms": wl.t_set_pubkey_ms,
           "parity_ok": bool((flags == expect).all()) and int(d_cnt.item()) == 3 * int(expect.sum()),
           "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_item": rsz + 4}}
    ctx.close()
    return res


def secondary_workloads(pkg, synth, local_rank, dev, window):
    """BASELINE.json configs 2 and 3 (parity-test cases, reported for orientation): 4096 PS verifications with 3 attributes and
    65 536 IdP issuances with 8 attributes (4 hidden), device-resident records, torch events on the launch stream."""
    import numpy as np
    import torch
    res = {}
    stream = torch.cuda.current_stream().cuda_stream

    def timed(fn, reps=3):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    ctx = pkg.Context(pkg.CURVE_BN254, local_rank)
    wl = synth.Workload(ctx, 3, seed=20211, window_bits=window)
    n = 4096
    recs, expect = wl.ps_verify_batch(n)
    d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
    d_fl = torch.zeros(n, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    ms = timed(lambda: ctx._chk(ctx.lib.elp_ps_verify_batch_dev(ctx.h, stream, n, d_rec.data_ptr(), 3, d_fl.data_ptr(), d_cnt.data_ptr())))
    res["ps_verify_4096x3attrs"] = {"value": n / (ms * 1e-3), "unit": "verifications/s", "kernel_ms": ms,
                                    "parity_ok": bool((d_fl.cpu().numpy() == expect).all()),
                                    "path": "k_ps_k_coop (K on 8 lanes per item) + k_pair16 (round 6: the pairing check with one item per 16-lane row of a wave, every Fp12-level "
                                            "operation one inner product per lane over operands in LDS; ELP_OPT_PAIR16, csrc/elpasso_pair16.h, profiles/r06_pair16.md); "
                                            "round 5 ran the 32-lane-pair interpreter here: 2.77-2.84 ms",
                                    "valu_bound": valu_frac(ctx, "ps_verify", window, n, ms),
                                    "roofline": {"bound": "hbm", "achieved": n * 228 / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                 "frac": n * 228 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_item": 228}}
    # PS verification between the interpreter's range and the full-chip kernels: the four-lanes-per-item pairing check (ELP_OPT_PAIR4, round 5) on / off
    nm = 16384
    mrecs, mexpect = wl.ps_verify_batch(nm)
    d_mrec = torch.from_numpy(np.frombuffer(mrecs, dtype=np.uint8).copy()).to(dev)
    d_mfl = torch.zeros(nm, dtype=torch.uint8, device=dev)
    mid = {}
    mid_ok = True
    for mode in (0, 1):
        ctx.set_pair4(mode)
        for m in (8192, 12288, 16384):
            d_mfl.zero_()
            mms = timed(lambda: ctx._chk(ctx.lib.elp_ps_verify_batch_dev(ctx.h, stream, m, d_mrec.data_ptr(), 3, d_mfl.data_ptr(), d_cnt.data_ptr())))
            mid["ps_verify_n%d_%s_ms" % (m, "four_lanes" if mode else "two_lanes")] = mms
            mid_ok = mid_ok and bool((d_mfl.cpu().numpy()[:m] == mexpect[:m]).all())      # every (mode, size) run is checked, not only the last
    mid["parity_ok"] = mid_ok
    res["mid_batches_ps_verify"] = mid
    # small batches and lone items, cooperative kernels on / off (ELP_OPT_COOP_PAIRING): latency, not throughput
    lat = {}
    ctx.set_pair4(0)          # this section compares the interpreter with the per-lane kernels; the four-lane path has the sections mid_batches_*
    for coop, row16, tag in ((1, 1, "row_of_16"), (1, 0, "interpreter"), (0, 0, "per_lane")):      # row_of_16 = the default since round 6 (4 ... 4 096 items)
        ctx.set_coop_pairing(coop)
        ctx.set_pair16(row16)
        for m in (4096, 1, 64, 1024, 4096):       # the first entry warms tables and TLBs for this mode (its time is overwritten by the last)
            ms = timed(lambda: ctx._chk(ctx.lib.elp_ps_verify_batch_dev(ctx.h, stream, m, d_rec.data_ptr(), 3, d_fl.data_ptr(), d_cnt.data_ptr())))
            lat["ps_verify_n%d_%s_ms" % (m, tag)] = ms
    ctx.set_coop_pairing(1)
    ctx.set_pair16(1)
    ctx.close()
    ctx = pkg.Context(pkg.CURVE_BN254, local_rank)
    ctx.set_pair4(0)
    wl = synth.Workload(ctx, 8, seed=20211, window_bits=window)
    nl = 8192
    vrecs, vmask, vexpect = wl.verify_id_batch(nl, 4, with_retrieval=True)
    d_vrec = torch.from_numpy(np.frombuffer(vrecs, dtype=np.uint8).copy()).to(dev)
    d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
    d_fl = torch.zeros(nl, dtype=torch.uint8, device=dev)
    for coop in (1, 0):
        ctx.set_coop_pairing(coop)
        for m in (8192, 1, 64, 1024, 4096, 8192):
            ms = timed(lambda: ctx._chk(ctx.lib.elp_verify_id_batch_dev(ctx.h, stream, m, d_vrec.data_ptr(), vmask, 1, d_ad.data_ptr(), None, len(wl.ad),
                                                                         d_fl.data_ptr(), d_cnt.data_ptr())))
            lat["verify_id_n%d_%s_ms" % (m, "cooperative" if coop else "per_lane")] = ms
    lat["parity_ok"] = bool((d_fl.cpu().numpy()[:nl] == vexpect).all())
    ctx.set_coop_pairing(1)
    # mid-size batches (round 5; VERDICT r4 "what's missing" #2): NIZK half in the job kernels + the pairing check on four lanes per item (k_vid_mid, ELP_OPT_PAIR4)
    # against the two-lane kernels' flat round
    nmid = 16384
    mvrecs, mvmask, mvexpect = wl.verify_id_batch(nmid, 4, with_retrieval=True)
    d_mvrec = torch.from_numpy(np.frombuffer(mvrecs, dtype=np.uint8).copy()).to(dev)
    d_mvfl = torch.zeros(nmid, dtype=torch.uint8, device=dev)
    midv = {}
    midv_ok = True
    for mode in (0, 1):
        ctx.set_pair4(mode)
        for m in (9217, 12288, 16384):
            d_mvfl.zero_()
            mms = timed(lambda: ctx._chk(ctx.lib.elp_verify_id_batch_dev(ctx.h, stream, m, d_mvrec.data_ptr(), mvmask, 1, d_ad.data_ptr(), None, len(wl.ad),
                                                                          d_mvfl.data_ptr(), d_cnt.data_ptr())))
            midv["verify_id_n%d_%s_ms" % (m, "four_lanes" if mode else "two_lanes")] = mms
            midv_ok = midv_ok and bool((d_mvfl.cpu().numpy()[:m] == mvexpect[:m]).all())
    midv["parity_ok"] = midv_ok
    res["mid_batches_verify_id"] = midv
    if os.environ.get("ELP_BENCH_SMALL_OVERLAP"):      # experiment: the two-stream form of the same calls (ELP_OPT_STREAM_OVERLAP) inside this process
        ctx.set_coop_pairing(1)
        ctx.set_stream_overlap(1)
        for m in (4096, 64, 1024, 4096):
            t0 = time.perf_counter()
            ms = timed(lambda: ctx._chk(ctx.lib.elp_verify_id_batch_dev(ctx.h, stream, m, d_vrec.data_ptr(), vmask, 1, d_ad.data_ptr(), None, len(wl.ad),
                                                                         d_fl.data_ptr(), d_cnt.data_ptr())))
            lat["verify_id_n%d_overlap_ms" % m] = ms
            lat["verify_id_n%d_overlap_wall_ms_4_calls" % m] = (time.perf_counter() - t0) * 1e3
        ctx.set_stream_overlap(0)
    res["small_batches"] = lat
    ctx.close()
    ctx = pkg.Context(pkg.CURVE_BN254, local_rank)
    wl = synth.Workload(ctx, 8, seed=20211, window_bits=window)
    n = 65536
    recs, mask, expect = wl.provide_id_batch(n, 4)
    d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
    d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
    d_fl = torch.zeros(n, dtype=torch.uint8, device=dev)
    d_sig = torch.zeros(n * 2 * ctx.G1, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    ms = timed(lambda: ctx._chk(ctx.lib.elp_provide_id_batch_dev(ctx.h, stream, n, d_rec.data_ptr(), mask, d_ad.data_ptr(), None,
                                                                  len(wl.ad), d_sig.data_ptr(), d_fl.data_ptr(), d_cnt.data_ptr())))
    res["provide_id_65536x8attrs"] = {"value": n / (ms * 1e-3), "unit": "issuances/s", "kernel_ms": ms,
                                      "parity_ok": bool((d_fl.cpu().numpy() == expect).all()),
                                      "valu_bound": valu_frac(ctx, "provide_id", window, n, ms),
                                      "roofline": {"bound": "hbm", "achieved": n * 548 / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                   "frac": n * 548 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_item": 548}}
    # wire ingest: the same kind of proofs as undecoded IdProof messages -- T-L-V parse, point decompression
    # and attribute hashing run inside the kernel
    nw = 65536
    vrecs, vmask, vexpect = wl.verify_id_batch(nw, 4, with_retrieval=True)
    msgs, moff = wl.wire_messages(vrecs, nw, 4, with_retrieval=True)
    d_msg = torch.from_numpy(np.frombuffer(msgs, dtype=np.uint8).copy()).to(dev)
    d_off = torch.from_numpy(moff.view(np.int32).copy()).to(dev)
    d_wfl = torch.zeros(nw, dtype=torch.uint8, device=dev)
    ms = timed(lambda: ctx._chk(ctx.lib.elp_verify_id_wire_batch_dev(ctx.h, stream, nw, d_msg.data_ptr(), d_off.data_ptr(), 1, d_ad.data_ptr(),
                                                                      None, len(wl.ad), d_wfl.data_ptr(), d_cnt.data_ptr())))
    res["verify_id_wire_65536x8attrs"] = {"value": nw / (ms * 1e-3), "unit": "verifications/s", "kernel_ms": ms,
                                          "bytes_per_message": len(msgs) / nw, "parity_ok": bool((d_wfl.cpu().numpy() == vexpect).all())}
    # lone messages and small / mid-size batches of messages (round 5): decoded into records on the GPU (k_wire_decode), then the record path of that size
    # (ELP_OPT_WIRE_DECODE = 1, the default) against the fused wire kernels' full round (0)
    wsm = {}
    for mode in (0, 1):
        ctx.set_wire_decode(mode)
        d_wfl.zero_()
        for m in (16384, 1, 1024, 16384):      # the first entry warms this mode
            wms = timed(lambda: ctx._chk(ctx.lib.elp_verify_id_wire_batch_dev(ctx.h, stream, m, d_msg.data_ptr(), d_off.data_ptr(), 1, d_ad.data_ptr(),
                                                                               None, len(wl.ad), d_wfl.data_ptr(), d_cnt.data_ptr())))
            wsm["verify_id_wire_n%d_%s_ms" % (m, "decoded_to_records" if mode else "fused_wire_kernel")] = wms
        wsm["parity_ok"] = wsm.get("parity_ok", True) and bool((d_wfl.cpu().numpy()[:16384] == vexpect[:16384]).all())
    res["wire_small_batches"] = wsm
    # user side: batch prover, its output fed straight to the batch verifier
    recs, mask = wl.prove_id_batch(n, 4, with_retrieval=True)
    d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
    osz = ctx.lib.elp_verify_id_record_size(ctx.curve, 8, 4, 1)
    d_out = torch.zeros(n * osz, dtype=torch.uint8, device=dev)
    ms = timed(lambda: ctx._chk(ctx.lib.elp_prove_id_batch_dev(ctx.h, stream, n, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None,
                                                                len(wl.ad), d_out.data_ptr(), d_fl.data_ptr(), d_cnt.data_ptr())))
    produced = int(d_fl.sum().item())
    d_cnt.zero_()
    ctx._chk(ctx.lib.elp_verify_id_batch_dev(ctx.h, stream, n, d_out.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad),
                                             d_fl.data_ptr(), d_cnt.data_ptr()))
    torch.cuda.synchronize()
    res["prove_id_65536x8attrs"] = {"value": n / (ms * 1e-3), "unit": "proofs/s", "kernel_ms": ms,
                                    "parity_ok": produced == n and int(d_cnt.item()) == n,
                                    "note": "every proof of the batch prover is accepted by the batch verifier"}
    # the general single-output multi-scalar multiplication (north_star's "Pippenger bucket MSM"; elp_g1_msm_dev, round 6): points and 256-bit scalars resident in HBM
    try:
        res.update(msm_lines(ctx, wl, dev, stream, timed))
    except Exception as e:  # pragma: no cover
        res["msm_g1"] = {"error": str(e)}
    ctx.close()
    return res


def msm_lines(ctx, wl, dev, stream, timed):
    """secondary.msm_g1_<n>: sum_i k_i P_i over n = 65 536 and 1 048 576 DISTINCT points of G1 with full-width scalars, device buffers, the caller's workspace.
    Checked through linearity, a size-independent property: MSM(P, k) + MSM(P, k') == MSM(P, k + k' mod r) on the full size, and against the sum of n elp_g1_mul
    results at 4 096 points."""
    import numpy as np
    import torch
    out = {}
    R = 0x2523648240000001BA344D8000000007FF9F800000000010A10000000000000D
    G1 = ctx.G1
    nmax = 1 << 20
    rng = np.random.default_rng(20216)
    # distinct points: P_i = s_i g with 64-bit s_i (made on the GPU in slices through elp_g1_mul)
    pts = bytearray()
    for lo in range(0, nmax, 1 << 16):
        ks = np.zeros((1 << 16, 32), dtype=np.uint8)
        ks[:, :8] = rng.integers(1, 256, size=(1 << 16, 8), dtype=np.uint8)
        pts += ctx.g1_mul(bytes(wl.g) * (1 << 16), ks.tobytes())
    k1 = rng.integers(0, 256, size=(nmax, 32), dtype=np.uint8)
    k1[:, 31] &= 0x1F                                   # < 2^253 < r
    k2 = rng.integers(0, 256, size=(nmax, 32), dtype=np.uint8)
    k2[:, 31] &= 0x1F
    ksum = bytearray()                                  # k1 + k2 mod r
    for i0 in range(0, nmax, 1 << 14):
        for row1, row2 in zip(k1[i0:i0 + (1 << 14)], k2[i0:i0 + (1 << 14)]):
            ksum += ((int.from_bytes(row1.tobytes(), "little") + int.from_bytes(row2.tobytes(), "little")) % R).to_bytes(32, "little")
    d_pts = torch.from_numpy(np.frombuffer(bytes(pts), dtype=np.uint8).copy()).to(dev)
    d_k = [torch.from_numpy(x.reshape(-1).copy()).to(dev) for x in (k1, k2)] + [torch.from_numpy(np.frombuffer(bytes(ksum), dtype=np.uint8).copy()).to(dev)]
    d_out = torch.zeros(3 * G1, dtype=torch.uint8, device=dev)
    fm = ctypes.c_float()
    lanes, iters = 256 * 4 * 64 * 8, 1000
    ctx._chk(ctx.lib.elp_bench_fp_mul(ctx.h, lanes, iters, ctypes.byref(fm)))
    peak = lanes * iters * 2 / (fm.value * 1e-3)
    for n in (1 << 16, 1 << 20):
        wsb = ctx.lib.elp_msm_workspace_bytes(ctx.curve, 1, n)
        d_ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
        ms = timed(lambda: ctx._chk(ctx.lib.elp_g1_msm_dev(ctx.h, stream, n, d_pts.data_ptr(), d_k[0].data_ptr(), d_ws.data_ptr(), d_out.data_ptr())))
        for t in range(3):
            ctx._chk(ctx.lib.elp_g1_msm_dev(ctx.h, stream, n, d_pts.data_ptr(), d_k[t].data_ptr(), d_ws.data_ptr(), d_out.data_ptr() + t * G1))
        torch.cuda.synchronize()
        o = d_out.cpu().numpy().tobytes()
        lin = ctx.g1_add(o[:G1], o[G1:2 * G1]) == o[2 * G1:] and o[:G1] != bytes(G1)
        # 32 windows of 8 bits: one mixed addition (11 field products: 7 M + 4 S) per point and window is the bucket phase; the reductions are O(windows x buckets)
        per_point = 34 * 11      # 2 x 17 byte-windows of the split scalars
        ach = per_point * n / (ms * 1e-3)
        out["msm_g1_%d" % n] = {"value": n / (ms * 1e-3), "unit": "points/s", "ms": ms, "linearity_ok": bool(lin),
                                "algorithmic_bytes_per_point": G1 + 32, "hbm_GBps": n * (G1 + 32) / (ms * 1e-3) / 1e9,
                                "valu_bound": {"fp_mul_equivalents_per_point": per_point, "achieved": ach, "fp_mul_peak_per_s": peak, "frac": ach / peak, "unit": "modmul/s",
                                               "note": "bucket phase only (2 x 17 windows of the GLV-split scalars x one mixed addition of 11 field products per point); bucket reductions excluded"},
                                "kernels": "k_msm_prepare -> k_msm_buckets (LDS histogram + counting sort, one workgroup per window and slice of at most 8 192 points) -> k_msm_combine (slices, above 8 per window) -> k_msm_reduce -> k_msm_final_glv<17>; scalars split as +-k1 +- k2 lam first (k_msm_split_scalars), buckets dealt to the lanes by size"}
    # small case against the plain sum of scalar multiples
    n = 4096
    ref = ctx.g1_mul(bytes(pts[:n * G1]), k1[:n].tobytes())
    acc = ref
    while len(acc) > G1:                               # pairwise tree of batched additions
        h = len(acc) // 2
        acc = ctx.g1_add(acc[:h], acc[h:])
    d_ws = torch.zeros(ctx.lib.elp_msm_workspace_bytes(ctx.curve, 1, n), dtype=torch.uint8, device=dev)
    ctx._chk(ctx.lib.elp_g1_msm_dev(ctx.h, stream, n, d_pts.data_ptr(), d_k[0].data_ptr(), d_ws.data_ptr(), d_out.data_ptr()))
    torch.cuda.synchronize()
    out["msm_g1_parity_4096_vs_sum_of_multiples"] = bool(d_out.cpu().numpy().tobytes()[:G1] == acc)
    return out


def host_api(pkg, wl, recs, B, A, H, first, expect, window, device):
    """B el_passo_verify_id proofs through the C++ protocol classes of csrc/host (the reference's API surface, src/ps-verifier.h:18-49):
    IdProof objects -> PSVerifier::el_passo_verify_id_batch (records packed by host threads, attributes hashed on the host, one
    elp_verify_id_batch per context) and wire messages -> el_passo_verify_id_wire_batch / _wire_packed (T-L-V parse, decompression and
    attribute hashing on the GPU).  Times are host wall-clock per call (PCIe included), key set-up reported separately; a second run
    shards the batch over two contexts on the same GPU (the multi-device dispatcher: one host thread + one stream per context)."""
    import numpy as np
    b = importlib.import_module(PKG + ".build")
    L = ctypes.CDLL(b.HOST_LIB)
    L.elph_last_error.restype = ctypes.c_char_p
    msgs, moff = wl.wire_messages(recs, B, H, first_item=first, with_retrieval=True)
    moff = np.ascontiguousarray(moff, dtype=np.uint32)
    res = {"note": "C++ PSVerifier over the C-ABI (csrc/host): objects = std::vector<IdProof> -> el_passo_verify_id_batch; wire = std::vector<PSBuffer> "
                   "-> el_passo_verify_id_wire_batch; wire_packed = one contiguous buffer + offsets; host wall-clock per call, PCIe included, best / median of 5"}
    # the same window width in both rows (round 3 compared W = 20 with W = 16): what differs is the dispatcher alone
    for tag, W, nctx in (("one_context", window, 1), ("two_contexts_one_gpu", window, 2)):
        outv = (ctypes.c_double * 10)()
        acc = (ctypes.c_uint64 * 3)()
        flags = np.zeros(B, dtype=np.uint8)
        rc = L.elph_bench_verify_id_pipelined(ctypes.c_int(A), ctypes.c_int(H), wl.g, wl.gg, wl.XX, wl.Yi, wl.YYi, wl.apk, wl.g, wl.h, wl.service, wl.ad,
                                    recs, ctypes.c_size_t(B), ctypes.c_uint64(first), msgs, moff.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(W),
                                    ctypes.c_int(nctx), ctypes.c_int(device), ctypes.c_int(5), outv, acc, flags.ctypes.data_as(ctypes.c_void_p))
        if rc != 0:
            res[tag] = {"error": (L.elph_last_error() or b"").decode()}
            continue
        ok = bool((flags == expect).all()) and int(acc[0]) == int(acc[1]) == int(acc[2]) == int(expect.sum())
        res[tag] = {"window_bits": W, "contexts": nctx, "setup_s": outv[0], "build_objects_s": outv[1], "parity_ok": ok,
                    "objects": {"value": B / (outv[2] * 1e-3), "unit": "verifications/s", "best_ms": outv[2], "median_ms": outv[3]},
                    "wire": {"value": B / (outv[4] * 1e-3), "unit": "verifications/s", "best_ms": outv[4], "median_ms": outv[5], "bytes_per_message": len(msgs) / B},
                    "wire_packed": {"value": B / (outv[6] * 1e-3), "unit": "verifications/s", "best_ms": outv[6], "median_ms": outv[7]},
                    "objects_pipelined": {"value": B / (outv[8] * 1e-3), "unit": "verifications/s", "ms_per_batch_sustained": outv[8], "ms_one_batch_alone": outv[9],
                                          "note": "PSVerifier::el_passo_verify_id_submit / _collect, two batches in flight: the host packs batch i + 1 and its records cross "
                                                  "PCIe while the GPU verifies batch i (elp_verify_id_batch_submit / _wait); with several contexts every context keeps its "
                                                  "own slots and the shards of a batch are submitted side by side (round 5)"}}
    return res


def usable_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return n


def cpu_baseline(wl, ctx, recs, rsz, mask, gpu_flags, sample, bls=False):
    """The C oracle (reference-structure restatement, oracle/elp_oracle.c; bls: its BLS12-381 build) timed on the host cores over the
    first `sample` items of the same workload; its verdicts are also compared with the GPU's."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from elp_testlib import oracle, oracle_bls
    L = oracle_bls() if bls else oracle()
    A = wl.A
    g1 = wl.g + wl.Yi + ctx.hash_to_g1([wl.service]) + wl.g + wl.apk + wl.h + wl.X
    g2 = wl.gg + wl.XX + wl.YYi
    key = ctypes.c_void_p(L.elpo_key_new(A, g1, g2))
    cores = usable_cores()
    fl = np.zeros(sample, dtype=np.uint8)
    t0 = time.perf_counter()
    acc = L.elpo_verify_id_batch(key, sample, recs[:sample * rsz], rsz, mask, 1, wl.ad, len(wl.ad), fl.ctypes.data, cores)
    dt = time.perf_counter() - t0
    # single-thread rate on a smaller slice
    s1 = min(sample, 96)
    t0 = time.perf_counter()
    L.elpo_verify_id_batch(key, s1, recs[:s1 * rsz], rsz, mask, 1, wl.ad, len(wl.ad), None, 1)
    dt1 = time.perf_counter() - t0
    return {"value": sample / dt, "unit": "verifications/s", "cores": cores, "kind": "port",
            "sample": "first %d items of the same batch, C oracle in reference structure (one scalar-mult per term, two full "
                      "pairings), OpenMP over items" % sample,
            "single_thread_value": s1 / dt1, "agrees_with_gpu": bool((fl == gpu_flags[:sample]).all()), "accepted": int(acc)}


if __name__ == "__main__":
    main()
