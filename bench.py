#!/usr/bin/env python3
"""Headline benchmark: speaker-embeddings/sec for one VAE training pass (forward + ELBO + backward
[+ RCCL gradient all-reduce when N > 1]) at BASELINE.json's config 2: d_in=256, z=64, batch=1024 per
GPU, fp32, synthetic N(0,1) embeddings already resident in HBM.

    python bench.py [--gpus N --steps K --warmup W]

N > 1 runs one rank per GPU over RCCL.  Two ways to start it, same result:
  * `python bench.py --gpus N ...` by itself: this process never touches the GPU; it starts N fresh child
    ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1), waits for them, relays rank
    0's JSON line and exits non-zero if any rank failed (or if the box has fewer than N GPUs);
  * `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py --gpus N ...`: the ranks are already there (WORLD_SIZE is set), each runs its share.

Prints ONE JSON line (rank 0).  `value` = whole-job embeddings/s; weak scaling (1024 rows per GPU).
`roofline`: algorithmic FLOPs of the step (7,143,424 FLOP/embedding at cfg2) over the
step duration measured with HIP events on the launch stream, against the fp32 MFMA peak.
`cpu_baseline`: the CPU oracle (a port of the reference's math, oracle/psvae_oracle.py) timed on
this box's host cores over a bounded sample of the same workload (N = 1, rank 0 only).

Other workloads (not the headline): --workload decode (cfg5 decode-only sampling), --workload langevin (cfg5
conditional sampling: the classifier-guided Langevin chain of inference.py:72-103 + one decode).
--stub: CPU rehearsal of the launcher / barrier / max-over-ranks aggregation with gloo and a fake step (tests only;
its `value` means nothing).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

H = 512
PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md chip table
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA (spec figure without 2:1 sparsity)
PEAK_FP32_VALU_TFLOPS = 157.3   # fp32 vector peak (same table): the Langevin chain is per-lane VALU work
PEAK_HBM_GBS = 8000.0


def flop_per_embedding(D: int, Z: int) -> int:
    fwd = 2 * (D * H + H * H + H * Z) + (Z * H + H * H + H * D)
    bwd = 2 * fwd - 2 * D * H
    return 2 * (fwd + bwd)


def config_label(D, Z, B, dtype, cond) -> str:
    """BASELINE.json config name of a training shape, or "custom" (the label never claims a config the shape is not)"""
    if (D, Z, B, dtype, cond) == (256, 64, 1024, "f32", False):
        return "cfg2"
    if (D, Z, B, dtype, cond) == (256, 64, 4096, "bf16", True):
        return "cfg3"
    if (D, Z, B, dtype, cond) == (512, 128, 1024, "f32", False):
        return "cfg4 (per-GPU shape)"
    return "custom"


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


# --------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` with no ranks around yet
# --------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_children(args) -> int:
    """Start args.gpus fresh rank processes of this script and relay rank 0's JSON line.  Runs BEFORE anything in this
    process touches the GPU (device_count() does not initialise it on this image), so no process that has used the
    GPU ever forks or execs."""
    n = args.gpus
    if not args.stub:
        have = torch.cuda.device_count()
        if have < n:
            print(f"[bench] --gpus {n} needs {n} GPUs on this node, found {have}: nothing was run", file=sys.stderr)
            return 2
        from pseudo_speaker_vae_amd import build
        build.build(verbose=False)  # once, here: the ranks find a library that matches the sources
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), PSVAE_BENCH_CHILD="1")
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    deadline = time.time() + args.launch_timeout
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"[bench] rank {r} exited with code {code}", file=sys.stderr)
        if pending and (rc != 0 or time.time() > deadline):
            if rc == 0:
                rc = 124
                print(f"[bench] ranks {sorted(pending)} still running after {args.launch_timeout} s", file=sys.stderr)
            for r in pending:  # a rank died or hung: the others would wait in a collective forever
                procs[r].terminate()
            t_kill = time.time() + 10
            for r in pending:
                try:
                    procs[r].wait(timeout=max(0.1, t_kill - time.time()))
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            break
        if pending:
            time.sleep(0.05)
    out0 = procs[0].stdout.read() if procs[0].stdout else ""
    line = None
    for ln in out0.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if rc == 0 and line is None:
        print("[bench] rank 0 printed no result line", file=sys.stderr)
        rc = 1
    if rc == 0:
        print(line, flush=True)
    return rc


# --------------------------------------------------------------------------------------------------------------
# one rank
# --------------------------------------------------------------------------------------------------------------
class _StdoutToStderr:
    """RCCL prints a version banner on STDOUT when its first communicator comes up; the result line must be the only
    thing on stdout, so file descriptor 1 points at stderr while the process group is initialised and warmed up"""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


class Rank:
    """process-group plumbing of one rank: init, fence (barrier + device sync), max over ranks, teardown"""

    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.stub = args.stub
        self.dev = torch.device("cpu") if args.stub else torch.device("cuda", self.local_rank)
        self.pg = None
        self.use_dist = self.world > 1 or os.environ.get("PSVAE_BENCH_FORCE_DIST") == "1"  # 1-rank RCCL smoke
        if not self.stub:
            torch.cuda.set_device(self.local_rank)
        if self.use_dist:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            with _StdoutToStderr():
                if self.stub:
                    dist.init_process_group("gloo")
                else:
                    dist.init_process_group("nccl", device_id=self.dev)
                self.pg = dist.group.WORLD
                dist.barrier()  # brings the communicator up (and its banner out) now
                if not self.stub:
                    torch.cuda.synchronize(self.dev)

    def barrier(self):
        if self.use_dist:
            import torch.distributed as dist
            dist.barrier()

    def fence(self):
        self.barrier()
        if not self.stub:
            torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, *vals):
        if self.world == 1:
            return list(vals)
        import torch.distributed as dist
        t = torch.tensor(vals, dtype=torch.float64, device=self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.tolist()

    def ranks_seen(self):
        """how many ranks really took part: counted THROUGH the process group (a sum of ones), not read from the environment"""
        if not self.use_dist:
            return 1
        import torch.distributed as dist
        t = torch.ones(1, dtype=torch.float64, device=self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t.item())

    def close(self):
        if self.use_dist:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()


def timed(rk: Rank, step, steps: int, warmup: int):
    """W untimed steps, then EXACTLY `steps` between two fences; returns (wall s, device ms), max over ranks"""
    for i in range(warmup):
        step(i)
    rk.fence()
    if rk.stub:
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i)
        rk.fence()
        wall = time.perf_counter() - t0
        dev_ms = wall * 1e3
    else:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for i in range(steps):
            step(warmup + i)
        ev1.record()
        rk.fence()
        wall = time.perf_counter() - t0
        dev_ms = ev0.elapsed_time(ev1)
    wall, dev_ms = rk.max_over_ranks(wall, dev_ms)
    return wall, dev_ms


def bench_stub(args, rk: Rank):
    """the launcher / aggregation path on CPU with gloo: each rank "trains" a fake flat gradient and averages it"""
    import torch.distributed as dist
    B = args.batch
    grad = torch.full((1024,), float(rk.rank + 1))

    if os.environ.get("PSVAE_BENCH_STUB_FAIL_RANK") == str(rk.rank):  # test hook: this rank dies before the collective
        os._exit(3)

    def step(i):
        grad.fill_(float(rk.rank + 1))
        time.sleep(0.001 * (1 + rk.rank))  # ranks of unequal speed: the max over ranks must win
        if rk.world > 1:
            dist.all_reduce(grad, op=dist.ReduceOp.SUM)
            grad.mul_(1.0 / rk.world)
    wall, dev_ms = timed(rk, step, args.steps, args.warmup)
    ranks_seen = rk.ranks_seen()
    if rk.rank == 0:
        print(json.dumps({
            "metric": "speaker-embeddings/sec (fwd+bwd)", "value": round(B * rk.world * args.steps / wall, 1),
            "unit": "speaker-embeddings/s", "n_gpus": rk.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "STUB (CPU/gloo rehearsal of the launcher and aggregation path; not a measurement)",
                       "parallelism": f"dp{rk.world}", "allreduce": "gloo sum + scale", "ranks_seen": ranks_seen},
            "stub_grad_mean": float(grad[0])}), flush=True)


def bench_decode(args, rk: Rank, emit: bool = True):
    """cfg5: decode-only sampling (inference.py:22-25): z ~ N(0, I) from the device Philox stream, x_hat =
    decoder(z).  Rows are split evenly over the ranks; no collective."""
    from pseudo_speaker_vae_amd import _ffi, ops
    from pseudo_speaker_vae_amd.model import VAEModel
    dev, world, rank = rk.dev, rk.world, rk.rank
    D, Z = args.input_dim, args.latent_dim
    n = args.samples // world
    torch.manual_seed(0)
    m = VAEModel(D, Z)
    dec = [p.detach().to(dev) for p in m.ordered_params()[12:]]
    out = torch.empty(n, D, device=dev)
    z = torch.empty(n, Z, device=dev)
    bf16 = args.dtype == "bf16"
    nbytes = _ffi.lib().psvae_decode_bf16_scratch_bytes(n, D, Z) if bf16 else _ffi.lib().psvae_decode_scratch_bytes(n)
    scratch = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)

    separate = os.environ.get("PSVAE_DECODE_SEPARATE_DRAW") == "1"  # comparison: draw launch + decode (round 2's form)

    def step(i):
        if not separate:  # z drawn inside the fused decode kernel (psvae_sample_decode_*): one call
            ops.sample_decode(n, Z, dec, D, 1234 + rank, i * ((n * Z + 3) // 4), bf16=bf16, out=out, scratch=scratch)
            return
        _ffi.check(_ffi.lib().psvae_philox_normal_f32(z.data_ptr(), n * Z, 1234 + rank, i * ((n * Z + 3) // 4),
                                                      torch.cuda.current_stream().cuda_stream), "philox")
        if bf16:
            ops.decode_bf16(z, dec, D, out=out, scratch=scratch)
        else:
            ops.decode(z, dec, D, out=out, scratch=scratch)
    steps, warm = max(1, min(args.steps, 20)), max(1, min(args.warmup, 3))
    wall, dev_ms = timed(rk, step, steps, warm)
    if rank == 0:
        fpe = 2 * (Z * H + H * H + H * D)  # 851,968 FLOP per decoded embedding at cfg5
        achieved = fpe * n / (dev_ms / 1e3 / steps) / 1e12
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
        res = {
            "metric": "speaker-embeddings/sec (decode)", "value": round(n * world * steps / wall, 1),
            "unit": "speaker-embeddings/s", "n_gpus": world, "steps": steps, "warmup": warm,
            "ms_per_step": round(wall / steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"cfg5: decode-only sampling, {n * world} rows per step ({n}/GPU), d_in={D} z={Z}, "
                                   + ("Philox z on device (separate launch) + 3-layer decoder" if separate else
                                      "z drawn from Philox inside the fused 3-layer decode kernel")
                                   + (" (bf16 operands / fp32 accumulate)" if bf16 else ""), "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": None, "flop_per_embedding": fpe}}
        if emit:
            print(json.dumps(res), flush=True)
        return res
    return None


def bench_langevin(args, rk: Rank):
    """cfg5, conditional leg (inference.py:29-110): z0 ~ N(0, I); `--langevin-steps` steps of
    z += 0.5 s^2 grad[log p(y|z) + log p(z)] + s w noise with a LatentClassifier head (Z -> hidden -> C), all inside
    ONE kernel launch (one wavefront per 16 latent rows, on the matrix cores), then one decode of the final z.  Rows
    are split over the ranks."""
    from pseudo_speaker_vae_amd import _ffi, ops
    from pseudo_speaker_vae_amd.latent_classifier import LatentClassifier
    from pseudo_speaker_vae_amd.model import VAEModel
    dev, world, rank = rk.dev, rk.world, rk.rank
    D, Z, T, hidden, C_ = args.input_dim, args.latent_dim, args.langevin_steps, 128, 3
    n = args.samples // world
    torch.manual_seed(0)
    m = VAEModel(D, Z)
    clf = LatentClassifier(Z, C_, num_layers=2, hidden_dim=hidden)
    dec = [p.detach().to(dev) for p in m.ordered_params()[12:]]
    W1, b1, W2, b2 = [p.detach().to(dev).contiguous() for p in clf.parameters()]
    z = torch.empty(n, Z, device=dev)
    out = torch.empty(n, D, device=dev)
    scratch = torch.empty(int(_ffi.lib().psvae_decode_scratch_bytes(n)), dtype=torch.uint8, device=dev)
    nblk = (n * Z + 3) // 4
    L = _ffi.lib()

    def step(i):
        s = torch.cuda.current_stream().cuda_stream
        base = (i * (T + 1)) * nblk
        _ffi.check(L.psvae_philox_normal_f32(z.data_ptr(), n * Z, 99 + rank, base, s), "philox")
        _ffi.check(L.psvae_langevin_f32(n, Z, hidden, C_, W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr(),
                                        z.data_ptr(), 1, 0.05, 1.0, T, None, 99 + rank, base + nblk, None, s), "langevin")
        ops.decode(z, dec, D, out=out, scratch=scratch)
    steps, warm = max(1, min(args.steps, 5)), max(1, min(args.warmup, 1))
    wall, dev_ms = timed(rk, step, steps, warm)
    if rank == 0:
        # per row and Langevin step: classifier forward + backward 2*(2*Z*hidden + 2*hidden*C) MAC-flops, prior, update
        fpr = 2 * 2 * (Z * hidden + hidden * C_) + 6 * Z
        flops = (fpr * T + 2 * (Z * H + H * H + H * D)) * n
        achieved = flops / (dev_ms / 1e3 / steps) / 1e12
        print(json.dumps({
            "metric": "speaker-embeddings/sec (conditional sampling)", "value": round(n * world * steps / wall, 1),
            "unit": "speaker-embeddings/s", "n_gpus": world, "steps": steps, "warmup": warm,
            "ms_per_step": round(wall / steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg5 conditional: {n * world} rows ({n}/GPU) x {T} Langevin steps "
                                   f"(LatentClassifier {Z}->{hidden}->{C_}, target class 1) + one decode, d_in={D} z={Z}",
                       "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_FP32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
                         "flop_per_row_step": fpr,
                         "note": "the chain keeps z in registers for all steps (HBM traffic: one read + one write of z per "
                                 "launch); 94 % of its FLOPs are the classifier's 64->128 layer forward and backward, on "
                                 "fp32 MFMA (one wavefront per 16 rows, four per SIMD); the rest is Philox / softmax / update VALU work"}}), flush=True)


def bench_api_step(args, rk: Rank):
    """The drop-in API path, as the reference's training script drives it through Lightning
    (/root/reference/ps_vae/lightning.py:67-131, training.py:84): optimizer.zero_grad(); out = PseudoSpeakerVAE.
    training_step((x, y), i); out["loss"].backward()  [; optimizer.step() with --with-adam: torch.optim.Adam, what
    configure_optimizers returns].  Same shape and metric as the headline (cfg2), but every step goes through the
    nn.Module / autograd surface instead of TrainEngine's replay loop; from the third step on the fused step runs as one
    captured HIP graph (pseudo_speaker_vae_amd/api_graph.py; PSVAE_API_GRAPH=0 keeps it eager)."""
    from pseudo_speaker_vae_amd.lightning import PseudoSpeakerVAE
    if rk.world != 1:
        raise SystemExit("[bench] --workload api-step is a one-GPU measurement (the N-GPU job is the train workload)")
    dev = rk.dev
    D, Z, B = args.input_dim, args.latent_dim, args.batch
    torch.manual_seed(0)
    mod = PseudoSpeakerVAE(model=dict(input_dim=D, latent_dim=Z), optimizer=dict(lr=1e-3), scheduler=dict(T_max=10)).to(dev)
    opt = mod.configure_optimizers()["optimizer"]
    plan = mod.model.tune_from_committed_plan(B)
    g = torch.Generator().manual_seed(1234)
    nbatch = 8  # the loop walks a few resident batches: every step copies a different x into the step's input
    xs = [torch.randn(B, D, generator=g).to(dev) for _ in range(nbatch)]
    y = torch.zeros(B, dtype=torch.long, device=dev)

    def step(i):
        opt.zero_grad(set_to_none=True)
        out = mod.training_step((xs[i % nbatch], y), i)
        out["loss"].backward()
        if args.with_adam:
            opt.step()
    prewarm(step, args, rk)
    wall, dev_ms = timed(rk, step, args.steps, args.warmup)
    graphs = mod.model._scratch_cache.graphs
    loss = [float(mod.logged[k]) for k in ("train_loss", "train_recon_loss", "train_kl_loss")]
    fpe = flop_per_embedding(D, Z)
    step_s_dev = dev_ms / 1e3 / args.steps
    achieved = fpe * B / step_s_dev / 1e12
    slot = next(iter(graphs.slots.values()), None)
    print(json.dumps({
        "metric": "speaker-embeddings/sec (fwd+bwd)", "value": round(B * args.steps / wall, 1),
        "unit": "speaker-embeddings/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{config_label(D, Z, B, 'f32', False)} through the drop-in API: optimizer.zero_grad + "
                               f"PseudoSpeakerVAE.training_step + loss.backward(){' + optimizer.step (' + type(opt).__name__ + ')' if args.with_adam else ''}"
                               f" per step (lightning.py:67-131), d_in={D} z={Z} hidden=512, batch={B}, fp32",
                   "launch": "one captured HIP graph per step behind the autograd Function" if graphs.served
                   else "eager launches from the autograd Function",
                   "graph_steps_served": graphs.served, "launches_per_step": slot.launches if slot else None,
                   "launch_plan": plan, "n_gpus": 1, "parallelism": "dp1", "prewarm_ms": args.prewarm_ms},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None, "flop_per_embedding": fpe,
                     "device_ms_per_step": round(step_s_dev * 1e3, 5),
                     "scope": "whole API step incl. the batch copy, the loss copy and the gradient scale launch, "
                              "HIP events on the launch stream"},
        "loss": [round(v, 6) for v in loss]}), flush=True)


def prewarm(step, args, rk: Rank):
    """untimed, BEFORE the W warm-up steps: run the step for --prewarm-ms of wall time so that the timed region sees the
    clocks and caches of a training run in progress, not of a GPU that was idle a millisecond ago (a 20-step run from
    idle measured 5 % slower than the same loop after 40 ms of work).  Reported as config.prewarm_ms; 0 switches it off."""
    if args.prewarm_ms <= 0 or rk.stub:
        return
    t0 = time.perf_counter()
    i = 0
    while (time.perf_counter() - t0) * 1e3 < args.prewarm_ms:
        for _ in range(50):
            step(i)
            i += 1
        torch.cuda.synchronize(rk.dev)


def committed_traffic(D, Z, B, dtype, plan, cond=False):
    """HBM-side bytes per step from the committed rocprofv3 PMC passes -- reported with its provenance, and only when
    the committed passes were taken on this shape; never presented as measured in this run"""
    if (D, Z, B, dtype, cond) == (256, 64, 4096, "bf16", True):
        names = ("r03_cfg3_final_pmc.json",)
    elif (D, Z, B, dtype, cond) == (256, 64, 1024, "f32", False):
        names = ("r03_cfg2_pmc.json", "r02_cfg2_pmc.json", "r01_cfg2_pmc.json")
    else:
        return None, None
    for name in names:
        path = os.path.join(REPO, "profiles", name)
        if os.path.exists(path):
            try:
                rec = json.load(open(path))
            except (OSError, ValueError):
                continue
            same_plan = rec.get("launch_plan") in (None, plan)
            return rec.get("hbm_bytes_per_step"), f"profiles/{name} (separate rocprofv3 --pmc passes of `python bench.py ...`" \
                + ("" if same_plan else "; taken under a DIFFERENT launch plan than this run's") + ")"
    return None, None


def run_rank(args) -> int:
    rk = Rank(args)
    world, rank, dev = rk.world, rk.rank, rk.dev
    if world != args.gpus:
        args.gpus = world
    if args.stub:
        bench_stub(args, rk)
        rk.close()
        return 0
    from pseudo_speaker_vae_amd import build
    if rank == 0 and os.environ.get("PSVAE_BENCH_CHILD") != "1":
        build.build(verbose=False)  # no-op when the shipped .so matches the sources (content hash)
    rk.barrier()                    # nobody loads the library before rank 0 has (re)built it
    if args.workload == "decode":
        bench_decode(args, rk)
        rk.close()
        return 0
    if args.workload == "langevin":
        bench_langevin(args, rk)
        rk.close()
        return 0
    if args.workload == "api-step":
        bench_api_step(args, rk)
        rk.close()
        return 0

    out = measure_train(args, rk)
    if rank == 0 and out is not None:
        if secondary_wanted(args, world):
            out["secondary"] = measure_secondary(args, rk)
        print(json.dumps(out), flush=True)
    rk.close()
    return 0


def secondary_wanted(args, world) -> bool:
    """the default headline run (one GPU, cfg2 as BASELINE.json names it) also carries short measurements of the other
    single-GPU configurations in its JSON line, so that they are timed by whoever runs the headline"""
    return (world == 1 and not args.no_secondary and args.workload == "train" and args.dtype == "f32" and not args.cond
            and (args.input_dim, args.latent_dim, args.batch) == (256, 64, 1024) and not args.gather and not args.with_adam
            and args.mode == "auto" and not args.plan and not args.tune)


def measure_secondary(args, rk):
    """cfg3 (bf16, conditional, B = 4096), cfg4's per-GPU shape (fp32, d = 512, z = 128, B = 1024) and cfg5's decode-only
    sampling (1 M rows, fp32 and bf16): same code paths as `--dtype bf16 --cond --batch 4096`, `--input-dim 512
    --latent-dim 128`, `--workload decode [--dtype bf16]`, fewer steps.  A failure is recorded, never raised."""
    import copy
    res = {}

    def brief(d):
        return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "dtype": d["dtype"],
                "workload": d["config"]["workload"], "roofline_frac": d["roofline"]["frac"],
                "roofline_peak": d["roofline"]["peak"], "traffic": d["roofline"].get("traffic"),
                **({"launches_per_step": d["config"]["launches_per_step"], "weight_copies": d["config"]["weight_copies"]}
                   if "launches_per_step" in d["config"] else {})}

    def run(name, fn, **over):
        a = copy.copy(args)
        a.no_cpu_baseline = True
        for k, v in over.items():
            setattr(a, k, v)
        try:
            res[name] = brief(fn(a))
        except Exception as e:  # noqa: BLE001 -- the headline line must not depend on a secondary measurement
            res[name] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.synchronize(rk.dev)
        torch.cuda.empty_cache()

    run("cfg3", lambda a: measure_train(a, rk), dtype="bf16", cond=True, batch=4096, steps=100, warmup=20)
    run("cfg4_per_gpu_shape", lambda a: measure_train(a, rk), input_dim=512, latent_dim=128, steps=100, warmup=20)
    run("cfg5_decode_f32", lambda a: bench_decode(a, rk, emit=False), workload="decode", steps=3, warmup=1)
    run("cfg5_decode_bf16", lambda a: bench_decode(a, rk, emit=False), workload="decode", dtype="bf16", steps=5, warmup=1)
    return res


def measure_train(args, rk):
    """one training-pass measurement (the headline's code path for any shape / dtype): the result dict on rank 0"""
    world, rank, dev = rk.world, rk.rank, rk.dev
    from pseudo_speaker_vae_amd.engine import TrainEngine
    D, Z, B = args.input_dim, args.latent_dim, args.batch
    eng = TrainEngine(D, Z, B, device=dev, seed=0, beta=1.0, process_group=rk.pg, dtype=args.dtype,
                      conditional=args.cond, direct_allreduce=(args.allreduce == "direct") if rk.pg is not None else False)
    g = torch.Generator().manual_seed(1234 + rank)  # each rank its own shard of synthetic rows
    x = torch.randn(B, D, generator=g).to(dev)
    if args.cond:
        eng.set_labels(torch.randint(0, 3, (B,), generator=g), torch.randint(0, 3, (B,), generator=g))

    gather = None
    if args.gather:
        # batch assembly inside the timed step (ps_vae/data/cv.py:58-76 loads one file per sample; here the dataset is one
        # [N, D] table in HBM and a batch is an index gather, pseudo_speaker_vae_amd/data.py): a CommonVoice-sized table
        # (674,068 rows, plots/dataset_info_train.json) of synthetic rows, a shuffled epoch permutation on the device,
        # every step gathers ITS rows into the step's input buffer with psvae_gather_rows_f32 before the pass runs
        from pseudo_speaker_vae_amd.data import PackedEmbeddingTable
        n_rows = args.table_rows
        tab = PackedEmbeddingTable(torch.randn(n_rows, D, device=dev, generator=torch.Generator(device=dev).manual_seed(7 + rank)), device=dev)
        perm = torch.randperm(n_rows, device=dev, generator=torch.Generator(device=dev).manual_seed(11 + rank))
        nb = n_rows // B
        def gather(i, out):
            o = (i % nb) * B
            tab.batch(perm[o:o + B], out=out)

    if args.plan:
        plan = eng.tune(x, plan=args.plan)       # an explicit plan (text form printed as config.launch_plan)
    elif args.tune:
        plan = eng.tune(x, force=True)           # measure now (a few seconds, outside the timed region)
    else:
        plan = eng.tune(x)                       # the committed plan for this GPU model and shape, else heuristics
    mode = args.mode
    allreduce = "none"
    calibrated = None
    t_over = None
    direct = rk.pg is not None and eng.averager is not None and eng.averager.direct is not None
    in_graph_env = os.environ.get("PSVAE_DDP_IN_GRAPH", "0") in ("1", "2") or os.environ.get("PSVAE_DDP_OVERLAP") == "1"
    if mode == "auto" and not args.with_adam and not in_graph_env:
        # One GPU: the step can be issued as ONE C call that enqueues its launches on the stream ("eager") or replayed
        # as a captured HIP graph.  Both run the same kernels; a graph replay carries a fixed bubble of a few us per
        # launch of the graph, the eager form needs a host that stays ahead of the GPU.  Measure both for a moment
        # (untimed, before the warm-up steps) and take the faster one; config.launch says which, config.calibration
        # holds both figures.
        # N > 1: the same two forms, each followed by ONE exchange of the flat gradient buffer; every rank measures both
        # (the exchanges are collective), the decision is taken on the maximum over the ranks so that all ranks agree.
        eng.capture(x)
        def _time(fn, ms=120.0):
            torch.cuda.synchronize(dev)
            for _ in range(20):
                fn()
            torch.cuda.synchronize(dev)
            t0, n = time.perf_counter(), 0
            while (time.perf_counter() - t0) * 1e3 < ms:
                for _ in range(100):
                    fn()
                n += 100
                torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / n * 1e6
        if rk.pg is not None:
            def _time(fn, reps=300):  # (fixed counts: every rank must issue the same number of collectives)
                rk.fence()
                for _ in range(20):
                    fn()
                rk.fence()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                rk.fence()
                return (time.perf_counter() - t0) / reps * 1e6
        if gather is not None and rk.pg is None:
            t_graph = _time(lambda: (gather(0, eng.static_x), eng.replay()))
            t_eager = _time(lambda: (gather(0, x), eng.forward_backward(x)))
        elif rk.pg is not None:
            t_graph = _time(eng.replay)
            t_eager = _time(lambda: eng.step_then_allreduce(x))
            t_graph, t_eager = rk.max_over_ranks(t_graph, t_eager)
            # third form (fp32 under RCCL, more than one rank): the decoder bucket's all-reduce on a side stream under the
            # encoder backward (TrainEngine.step_overlapped) -- pays ~25 us of host work for hiding part of the exchange
            if world > 1 and args.dtype == "f32" and not direct and eng.averager.backend == "nccl" and gather is None \
                    and os.environ.get("PSVAE_DDP_NO_OVERLAP_FORM") != "1":
                t_over = rk.max_over_ranks(_time(lambda: eng.step_overlapped(x)), 0.0)[0]
        else:
            t_graph = _time(eng.replay)
            t_eager = _time(lambda: eng.forward_backward(x))
        calibrated = {"graph_us": round(t_graph, 2), "eager_us": round(t_eager, 2)}
        mode = "eager" if t_eager < t_graph else "graph"
        if t_over is not None:
            calibrated["overlapped_us"] = round(t_over, 2)
            if t_over < min(t_eager, t_graph):
                mode = "overlapped"
    if mode == "auto":
        mode = "graph"
    if mode == "overlapped":
        allreduce = "2 buckets: the decoder's on a side stream under the encoder backward, the encoders' behind the step"

        def step(i):
            eng.step_overlapped(x)
            if args.with_adam:
                eng.optimizer_step()
    elif mode == "graph" and rk.pg is not None and os.environ.get("PSVAE_DDP_OVERLAP") == "1":
        mode = "graph-ddp-overlap"  # two captured halves around two RCCL all-reduces (decoder bucket overlapped)
        allreduce = "2 buckets, eager between two graphs"
        eng.capture_ddp(x)

        def step(i):
            eng.replay_ddp()
            if args.with_adam:
                eng.optimizer_step()
    elif mode == "graph":
        in_graph = os.environ.get("PSVAE_DDP_IN_GRAPH", "0")  # 0 (default): the all-reduce is issued after the graph -- the
        # plain, everywhere-proven RCCL usage; 1: captured inside the graph; 2: decoder bucket as a parallel graph branch
        if rk.pg is not None and in_graph in ("1", "2"):
            try:
                eng.capture_with_collective(x, overlap=(in_graph == "2"))
                mode = "graph(all-reduce inside" + (", decoder bucket overlapped)" if in_graph == "2" else ")")
                allreduce = "captured in the step's graph"
            except Exception as e:  # the collective could not be captured: keep it outside the graph
                if rank == 0:
                    print(f"[bench] in-graph all-reduce unavailable ({type(e).__name__}: {e}); falling back", file=sys.stderr)
                eng._collective_in_graph = False
                mode = "graph+allreduce"
                allreduce = "one exchange of the flat gradient buffer after the graph"
                eng.capture(x)
        else:
            if rk.pg is not None:
                mode = "graph+allreduce"  # one captured step, then ONE RCCL all-reduce of the flat gradient buffer
                allreduce = "one exchange of the flat gradient buffer after the graph"
            if eng._graph is None or args.with_adam:
                eng.capture(x, with_adam=args.with_adam and rk.pg is None)

        adam_in_graph = args.with_adam and rk.pg is None

        def step(i):
            if gather is not None:
                gather(i, eng.static_x)
            eng.replay()
            if args.with_adam and not adam_in_graph:
                eng.optimizer_step()
    else:
        if rk.pg is not None and args.mode == "eager":   # asked for explicitly: the phase-split form with two buckets
            allreduce = "2 buckets (decoder first), async under the encoder backward"
        elif rk.pg is not None:
            mode = "stream+allreduce"
            allreduce = "one exchange of the flat gradient buffer behind the step on the same stream"
        ddp_stream = rk.pg is not None and args.mode != "eager"

        def step(i):
            if gather is not None:
                gather(i, x)
            if ddp_stream:
                eng.step_then_allreduce(x)
            else:
                eng.forward_backward(x)
            if args.with_adam:
                eng.optimizer_step()

    prewarm(step, args, rk)
    wall, dev_ms = timed(rk, step, args.steps, args.warmup)
    loss = eng.loss3.detach().cpu().tolist()
    ranks_seen = rk.ranks_seen()

    if rank == 0:
        traffic, traffic_source = committed_traffic(D, Z, B, args.dtype, plan, args.cond)
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
            "config": {"workload": f"{config_label(D, Z, B, args.dtype, args.cond)}: "
                                   f"{'conditional (one-hot gender+age) ' if args.cond else ''}"
                                   f"{'batch gather from a ' + str(args.table_rows) + '-row packed table in HBM + ' if args.gather else ''}"
                                   f"VAE fwd+ELBO+bwd{'+Adam' if args.with_adam else ''}"
                                   f"{'+RCCL grad all-reduce' if world > 1 else ''}, d_in={D} z={Z} hidden=512, "
                                   f"batch={B}/GPU, global_batch={B * world}, "
                                   f"{'fp32' if args.dtype == 'f32' else 'bf16 operands / fp32 accumulate'}",
                       "launch": {"overlapped": "stream, two C calls around the decoder bucket's all-reduce (side stream)",
                                  "eager": "stream (one C call enqueues the step's launches)",
                                  "graph": "graph (captured HIP graph, one replay per step)"}.get(mode, mode),
                       "n_gpus": world, "ranks_seen": ranks_seen, "parallelism": f"dp{world}",
                       "allreduce": allreduce if rk.pg is None else allreduce + (
                           " [direct: one-shot reduce-scatter + all-gather over IPC-mapped buffers, stream-ordered "
                           "interprocess events (csrc/ipc_allreduce.hip)]" if direct else " [RCCL ncclAvg]"),
                       "batch_source": ("each step's rows gathered on the device from a shuffled packed table "
                                        "(psvae_gather_rows_f32, inside the timed step)") if args.gather else
                                       "one static batch resident in HBM (batch assembly is outside the timed step; --gather times it)",
                       "weight_copies": ("k-major fp32 copies kept current by the optimizer step (outside fwd+bwd)" if args.dtype == "f32" else
                                          "bf16 packs rebuilt inside every pass (PSVAE_BF16_PACK_IN_STEP=1)" if getattr(eng, "_pack_in_step", False) else
                                          "bf16 packs kept current by the optimizer step (outside fwd+bwd; PSVAE_BF16_PACK_IN_STEP=1 "
                                          "rebuilds them inside every pass)"),
                       "launch_plan": plan, "launches_per_step": eng.launches_per_step(),
                       "prewarm_ms": args.prewarm_ms, "calibration": calibrated},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                         "traffic_source": traffic_source,
                         "flop_per_embedding": fpe, "device_ms_per_step": round(step_s_dev * 1e3, 5),
                         # what `achieved` divides by: HIP-event time of the WHOLE step (every launch of it, not only
                         # the GEMM kernel -- the committed rocprofv3 summary has the per-kernel averages)
                         "scope": "whole step: every launch of the training pass, HIP events on the launch stream"},
            "loss": [round(v, 6) for v in loss],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(D, Z, B)
        return out
    return None


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=1024, help="rows per GPU")
    ap.add_argument("--input-dim", type=int, default=256)
    ap.add_argument("--latent-dim", type=int, default=64)
    ap.add_argument("--mode", choices=["auto", "graph", "eager", "overlapped"], default="auto")
    ap.add_argument("--with-adam", action="store_true", help="also run the fused Adam step inside the timed region")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="bf16 = config 3 arithmetic (bf16 MFMA operands, fp32 accumulate and master weights)")
    ap.add_argument("--cond", action="store_true", help="config 3: one-hot gender+age conditioning")
    ap.add_argument("--workload", choices=["train", "decode", "langevin", "api-step"], default="train",
                    help="train = cfg2 (headline); decode = cfg5 decode-only sampling; langevin = cfg5 conditional "
                         "sampling (Langevin chain + decode); --samples rows in total; api-step = the train shape "
                         "through PseudoSpeakerVAE.training_step + loss.backward() (the drop-in API path)")
    ap.add_argument("--allreduce", choices=["rccl", "direct"], default=os.environ.get("PSVAE_BENCH_ALLREDUCE", "rccl"),
                    help="N > 1: gradient exchange -- RCCL's all-reduce (default), or the direct one-shot reduce-scatter + "
                         "all-gather over IPC-mapped buffers (csrc/ipc_allreduce.hip)")
    ap.add_argument("--gather", action="store_true",
                    help="train workload: assemble each step's batch from a packed embedding table in HBM inside the "
                         "timed step (a CommonVoice-sized table of synthetic rows, shuffled epoch order)")
    ap.add_argument("--table-rows", type=int, default=674068, help="--gather: rows of the packed table")
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="untimed: run the step for this long before the W warm-up steps (clocks / caches of a run in "
                         "progress); 0 = off")
    ap.add_argument("--samples", type=int, default=1_000_000, help="decode / langevin workloads: rows over all GPUs")
    ap.add_argument("--langevin-steps", type=int, default=100, help="Langevin steps per row (inference.py:34 default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="default headline run on one GPU: do not append the short cfg3 / cfg4-shape / cfg5 measurements "
                         "(`secondary` in the JSON line)")
    ap.add_argument("--tune", action="store_true", help="measure a launch plan now instead of using the committed one")
    ap.add_argument("--no-tune", action="store_true", help="(kept for older command lines: tuning is opt-in now)")
    ap.add_argument("--plan", default="", help="use this launch plan (text form printed as config.launch_plan)")
    ap.add_argument("--stub", action="store_true", help="CPU/gloo rehearsal of the launcher and aggregation (tests)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="self-launched ranks: give up after this many s")
    return ap.parse_args(argv)


def main() -> int:
    args = parse_args()
    if args.gpus < 1:
        print("[bench] --gpus must be >= 1", file=sys.stderr)
        return 2
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_children(args)  # this process stays off the GPU
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
