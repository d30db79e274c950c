#!/usr/bin/env python3
"""Headline benchmark: speaker-embeddings/sec for one VAE training pass (forward + ELBO + backward
[+ RCCL gradient all-reduce when N > 1]) at BASELINE.json's config 2: d_in=256, z=64, batch=1024 per
GPU, fp32, synthetic N(0,1) embeddings already resident in HBM.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W          (N > 1, one rank per GPU)

Prints ONE JSON line (rank 0).  `value` = whole-job embeddings/s; weak scaling (1024 rows per GPU).
`roofline`: algorithmic FLOPs of the step (7,143,424 FLOP/embedding at cfg2) over the
step duration measured with HIP events on the launch stream, against the fp32 MFMA peak.
`cpu_baseline`: the CPU oracle (a port of the reference's math, oracle/psvae_oracle.py) timed on
this box's host cores over a bounded sample of the same workload (N = 1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

H = 512
PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md chip table
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA (spec figure without 2:1 sparsity)


def flop_per_embedding(D: int, Z: int) -> int:
    fwd = 2 * (D * H + H * H + H * Z) + (Z * H + H * H + H * D)
    bwd = 2 * fwd - 2 * D * H
    return 2 * (fwd + bwd)


def cpu_baseline(D, Z, B, steps=40, warmup=3):
    """oracle fwd+ELBO+bwd on the host cores (checker code used as the timed CPU port)."""
    from oracle import psvae_oracle as O
    # the GPU box gives one job a 16-core share of a 256-thread host: more threads only thrash
    ncores = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(ncores)
    p = O.init_params(D, Z, 0)
    torch.manual_seed(0)
    x = torch.randn(B, D)
    eps = torch.randn(B, Z)
    for _ in range(warmup):
        O.train_step(p, x, eps, 1.0)
    t0 = time.perf_counter()
    n = 0
    while n < steps and time.perf_counter() - t0 < 20.0:
        O.train_step(p, x, eps, 1.0)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(B * n / dt, 1), "unit": "speaker-embeddings/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{n} steps of fwd+ELBO+bwd at B={B}, D={D}, Z={Z} (oracle, torch-CPU fp32)"}


def bench_decode(args, world, rank, dev):
    """cfg5: decode-only sampling (inference.py:22-25): z ~ N(0, I) from the device Philox stream, x_hat =
    decoder(z).  Rows are split evenly over the ranks; no collective."""
    from pseudo_speaker_vae_amd import ops
    from pseudo_speaker_vae_amd.model import VAEModel
    D, Z = args.input_dim, args.latent_dim
    n = args.samples // world
    torch.manual_seed(0)
    m = VAEModel(D, Z)
    dec = [p.detach().to(dev) for p in m.ordered_params()[12:]]
    out = torch.empty(n, D, device=dev)
    z = torch.empty(n, Z, device=dev)
    scratch = torch.empty(1, dtype=torch.uint8, device=dev)

    def step(i):
        from pseudo_speaker_vae_amd import _ffi
        _ffi.check(_ffi.lib().psvae_philox_normal_f32(z.data_ptr(), n * Z, 1234 + rank, i * ((n * Z + 3) // 4),
                                                      torch.cuda.current_stream().cuda_stream), "philox")
        if bf16:
            ops.decode_bf16(z, dec, D, out=out, scratch=step.scratch)
        else:
            ops.decode(z, dec, D, out=out, scratch=step.scratch)
    from pseudo_speaker_vae_amd import _ffi
    bf16 = args.dtype == "bf16"
    nbytes = _ffi.lib().psvae_decode_bf16_scratch_bytes(n, D, Z) if bf16 else _ffi.lib().psvae_decode_scratch_bytes(n)
    step.scratch = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)

    def fence():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize(dev)
    steps, warm = max(1, min(args.steps, 20)), max(1, min(args.warmup, 3))
    for i in range(warm):
        step(i)
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(steps):
        step(warm + i)
    ev1.record()
    fence()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([wall, dev_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, dev_ms = t.tolist()
    if rank == 0:
        fpe = 2 * (Z * H + H * H + H * D)  # 851,968 FLOP per decoded embedding at cfg5
        achieved = fpe * n / (dev_ms / 1e3 / steps) / 1e12
        print(json.dumps({
            "metric": "speaker-embeddings/sec (decode)", "value": round(n * world * steps / wall, 1),
            "unit": "speaker-embeddings/s", "n_gpus": world, "steps": steps, "warmup": warm,
            "ms_per_step": round(wall / steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"cfg5: decode-only sampling, {n * world} rows per step ({n}/GPU), d_in={D} z={Z}, "
                                   "Philox z on device + 3-layer decoder"
                                   + (" (bf16 operands / fp32 accumulate)" if bf16 else ""), "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3),
                         "peak": PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / (PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS), 4),
                         "traffic": None, "flop_per_embedding": fpe}}), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=1024, help="rows per GPU")
    ap.add_argument("--input-dim", type=int, default=256)
    ap.add_argument("--latent-dim", type=int, default=64)
    ap.add_argument("--mode", choices=["auto", "graph", "eager"], default="auto")
    ap.add_argument("--with-adam", action="store_true", help="also run the fused Adam step inside the timed region")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="bf16 = config 3 arithmetic (bf16 MFMA operands, fp32 accumulate and master weights)")
    ap.add_argument("--cond", action="store_true", help="config 3: one-hot gender+age conditioning (needs --dtype bf16)")
    ap.add_argument("--workload", choices=["train", "decode"], default="train",
                    help="train = cfg2 (headline); decode = cfg5: decode-only sampling, --samples rows in total")
    ap.add_argument("--samples", type=int, default=1_000_000, help="decode workload: rows over all GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tune", action="store_true", help="skip the launch-plan autotuning (heuristic configs)")
    ap.add_argument("--plan", default="", help="use this launch plan (text form printed as config.launch_plan)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    from pseudo_speaker_vae_amd import build
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    use_dist = world > 1 or os.environ.get("PSVAE_BENCH_FORCE_DIST") == "1"  # 1-rank RCCL smoke of the DDP path
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)
        pg = dist.group.WORLD
    if rank == 0:
        build.build(verbose=False)  # no-op when the shipped .so matches the sources (content hash)
    if use_dist:
        dist.barrier()              # nobody loads the library before rank 0 has (re)built it
    if args.workload == "decode":
        return bench_decode(args, world, rank, dev)

    from pseudo_speaker_vae_amd.engine import TrainEngine
    D, Z, B = args.input_dim, args.latent_dim, args.batch
    eng = TrainEngine(D, Z, B, device=dev, seed=0, beta=1.0, process_group=pg, dtype=args.dtype,
                      conditional=args.cond)
    g = torch.Generator().manual_seed(1234 + rank)  # each rank its own shard of synthetic rows
    x = torch.randn(B, D, generator=g).to(dev)
    if args.cond:
        eng.set_labels(torch.randint(0, 3, (B,), generator=g), torch.randint(0, 3, (B,), generator=g))

    plan = "untuned"
    if args.plan:
        plan = eng.tune(x, plan=args.plan)  # a plan tuned by an earlier run (profiling runs reuse the bench's)
    elif not args.no_tune:
        plan = eng.tune(x)  # launch-plan autotuning, once per shape, outside the timed region
    mode = args.mode
    if mode == "auto":
        mode = "graph"
    if mode == "graph" and pg is not None and os.environ.get("PSVAE_DDP_OVERLAP") == "1":
        mode = "graph-ddp-overlap"  # two captured halves around two RCCL all-reduces (decoder bucket overlapped)
        eng.capture_ddp(x)

        def step():
            eng.replay_ddp()
            if args.with_adam:
                eng.optimizer_step()
    elif mode == "graph":
        in_graph = os.environ.get("PSVAE_DDP_IN_GRAPH", "0")  # 0 (default): the all-reduce is issued after the graph -- the
        # plain, everywhere-proven RCCL usage; 1: captured inside the graph (36 us/step less at 1 rank, but multi-rank
        # capture could not be rehearsed on the 1-GPU box, so it is opt-in); 2: decoder bucket as a parallel graph branch
        if pg is not None and in_graph in ("1", "2"):
            try:
                eng.capture_with_collective(x, overlap=(in_graph == "2"))
                mode = "graph(all-reduce inside" + (", decoder bucket overlapped)" if in_graph == "2" else ")")
            except Exception as e:  # the collective could not be captured: keep it outside the graph
                if rank == 0:
                    print(f"[bench] in-graph all-reduce unavailable ({type(e).__name__}: {e}); falling back", file=sys.stderr)
                eng._collective_in_graph = False
                mode = "graph+allreduce"
                eng.capture(x)
        else:
            if pg is not None:
                mode = "graph+allreduce"  # one captured step, then ONE RCCL all-reduce of the flat gradient buffer
            eng.capture(x)

        def step():
            eng.replay()
            if args.with_adam:
                eng.optimizer_step()
    else:
        def step():
            eng.forward_backward(x)
            if args.with_adam:
                eng.optimizer_step()

    def fence():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    fence()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([wall, dev_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, dev_ms = t.tolist()
    loss = eng.loss3.detach().cpu().tolist()

    if rank == 0:
        traffic = None  # L2-fill bytes per step from the committed PMC passes (profiles/, cfg2 shape only)
        pmc = os.path.join(REPO, "profiles", "r01_cfg2_pmc.json")
        if (D, Z, B) == (256, 64, 1024) and args.dtype == "f32" and os.path.exists(pmc):
            traffic = json.load(open(pmc))["hbm_bytes_per_step"]
        fpe = flop_per_embedding(D, Z) + (2 * 3 * 2 * 6 * H if args.cond else 0)  # +6 one-hot columns x3, fwd+wgrad
        peak = PEAK_FP32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
        step_s_dev = dev_ms / 1e3 / args.steps
        achieved = fpe * B / step_s_dev / 1e12  # per GPU, over the HIP-event duration of the step
        out = {
            "metric": "speaker-embeddings/sec (fwd+bwd)",
            "value": round(B * world * args.steps / wall, 1),
            "unit": "speaker-embeddings/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{'cfg3' if args.dtype == 'bf16' else 'cfg2'}: "
                                   f"{'conditional (one-hot gender+age) ' if args.cond else ''}"
                                   f"VAE fwd+ELBO+bwd{'+Adam' if args.with_adam else ''}"
                                   f"{'+RCCL grad all-reduce' if world > 1 else ''}, d_in={D} z={Z} hidden=512, "
                                   f"batch={B}/GPU, global_batch={B * world}, "
                                   f"{'fp32' if args.dtype == 'f32' else 'bf16 operands / fp32 accumulate'}",
                       "launch": mode, "parallelism": f"dp{world}", "launch_plan": plan},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                         "flop_per_embedding": fpe, "device_ms_per_step": round(step_s_dev * 1e3, 5),
                         # what `achieved` divides by: HIP-event time of the WHOLE step (every launch of it, not only
                         # the GEMM kernel -- the committed rocprofv3 summary has the per-kernel averages)
                         "scope": "whole step: all launches of psvae_train_step (11 grouped-GEMM + weight transpose + "
                                  "bottleneck kernel at fp32), HIP events on the launch stream"},
            "loss": [round(v, 6) for v in loss],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(D, Z, B)
        print(json.dumps(out), flush=True)
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
