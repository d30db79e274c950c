"""TrainEngine: the native training-step driver for the hot path.

One process per GPU.  Parameters, gradients and Adam state live in flat fp32 buffers (the 18 tensors
in the reference's creation order, model.py:14-36, each 256-B aligned), so that
  * the HIP kernels write gradients straight into the all-reduce payload,
  * data-parallel averaging (the reference's DDPStrategy, training.py:78) is ONE RCCL all-reduce of the flat buffer
    after the captured step (or two bucketed ones, decoder first, overlapped with the encoder backward in the eager
    mode); the three loss scalars ride in the payload (replaces Lightning's per-key `sync_dist` all-reduces,
    lightning.py:127-129),
  * Adam (lightning.py:205) is one fused kernel over the flat buffers; its step counter and learning rate live on the
    device, so the optimizer can be captured in the step's HIP graph and CosineAnnealingLR (lightning.py:206-213) is
    one small device write per epoch (`scheduler_step`),
  * the optimizer also maintains the k-major ([in, out]) copies of the weight matrices the forward GEMMs read
    (`flat_params_t`): the training pass itself then needs no per-step weight transpose.
On one GPU the whole step (forward + ELBO + backward [+ Adam]) is captured into a HIP graph.
"""
from __future__ import annotations

import math
import os
import struct
from typing import List, Optional

import torch

from . import _ffi, ops
from .ddp import GradAverager, broadcast_flat
from .model import VAEModel

ALIGN = 64  # floats (256 B)


def _pad(n: int) -> int:
    return (n + ALIGN - 1) // ALIGN * ALIGN


class FlatLayout:
    """[encoder_mu x6 | encoder_sigma x6 | loss3 | decoder x6] -- decoder last = first bucket to reduce."""

    def __init__(self, shapes: List[torch.Size], extra: Optional[List[torch.Size]] = None):
        """extra: additional tensors (the conditional tables) placed at the end of the encoder bucket"""
        assert len(shapes) == _ffi.NPARAM
        self.shapes = list(shapes)
        self.extra_shapes = list(extra or [])
        self.offsets, self.extra_offsets = [], []
        o = 0
        for i, s in enumerate(shapes):
            if i == 12:
                for e in self.extra_shapes:
                    self.extra_offsets.append(o)
                    o += _pad(e.numel())
                self.loss_offset = o
                o += ALIGN
                self.enc_end = o
            self.offsets.append(o)
            o += _pad(s.numel())
        self.total = o

    def views(self, flat: torch.Tensor) -> List[torch.Tensor]:
        return [flat[o:o + s.numel()].view(s) for o, s in zip(self.offsets, self.shapes)]

    def kmajor_views(self, flat: torch.Tensor) -> List[torch.Tensor]:
        """the same offsets read as the k-major copies: weight k is [in, out]; bias slots alias the (unused) range"""
        out = []
        for o, s in zip(self.offsets, self.shapes):
            v = flat[o:o + s.numel()]
            out.append(v.view(s[1], s[0]) if len(s) == 2 else v.view(s))
        return out

    def extra_views(self, flat: torch.Tensor) -> List[torch.Tensor]:
        return [flat[o:o + s.numel()].view(s) for o, s in zip(self.extra_offsets, self.extra_shapes)]

    def adam_layout(self, param, param_kmajor, grad, exp_avg, exp_avg_sq) -> "_ffi.AdamDev":
        L = _ffi.AdamDev()
        L.param, L.grad, L.exp_avg, L.exp_avg_sq = (_ffi.ptr(t) for t in (param, grad, exp_avg, exp_avg_sq))
        L.param_kmajor = _ffi.ptr(param_kmajor) if param_kmajor is not None else None
        nv = 0
        for i, (o, s) in enumerate(zip(self.offsets, self.shapes)):
            if len(s) == 2:
                k = i // 2
                L.mat_off[k], L.mat_rows[k], L.mat_cols[k] = o, s[0], s[1]
            else:
                L.vec_off[nv], L.vec_len[nv] = o, s.numel()
                nv += 1
        for o, s in zip(self.extra_offsets, self.extra_shapes):
            L.vec_off[nv], L.vec_len[nv] = o, s.numel()
            nv += 1
        L.n_vec = nv
        return L


class TrainEngine:
    def __init__(self, input_dim: int, latent_dim: int, batch: int, device="cuda:0", seed: int = 0,
                 beta: float = 1.0, normalize_decoder: bool = False, use_cos_loss: bool = False,
                 lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 process_group=None, state_dict=None, dtype: str = "f32", conditional: bool = False,
                 shard: Optional[tuple] = None, scheduler: Optional[dict] = None, direct_allreduce: Optional[bool] = None):
        """dtype "bf16" = BASELINE config 3 arithmetic (bf16 MFMA operands, fp32 accumulate / master weights /
        gradients); conditional = one-hot gender+age gather tables, either dtype;
        scheduler = CosineAnnealingLR keyword arguments {"T_max": .., "eta_min": ..} (lightning.py:206)."""
        if dtype not in ("f32", "bf16"):
            raise ValueError(dtype)
        self.dtype, self.conditional = dtype, conditional
        self.D, self.Z, self.B = input_dim, latent_dim, batch
        self.device = torch.device(device)
        self.beta = float(beta)
        self.flags = (_ffi.NORMALIZE_DECODER if normalize_decoder else 0) | (_ffi.COS_LOSS if use_cos_loss else 0)
        self.base_lr, self.betas, self.adam_eps, self.weight_decay = float(lr), betas, eps, weight_decay
        self.scheduler = dict(scheduler) if scheduler else None
        self.epoch = 0
        self.pg = process_group
        self.world = 1
        self.rank = 0
        if process_group is not None:
            import torch.distributed as dist
            self.world = dist.get_world_size(process_group)
            self.rank = dist.get_rank(process_group)
        if shard is not None:  # (rank, world) of the epsilon stream without a process group (tests)
            self.rank, self._eps_world = int(shard[0]), int(shard[1])
        else:
            self._eps_world = self.world

        # default nn.Linear init in the reference's creation order under `seed` (CPU generator), so
        # every rank starts from identical parameters without a broadcast (Coll-4 of SURVEY.md 2.2)
        torch.manual_seed(seed)
        cpu_model = VAEModel(input_dim, latent_dim, normalize_decoder)
        if state_dict is not None:
            cpu_model.load_state_dict(state_dict)
        src = [p.detach() for p in cpu_model.ordered_params()]
        self.keys = [k for k, _ in cpu_model.named_parameters()]
        extra = [torch.Size((_ffi.COND_DIM, _ffi.HIDDEN))] * 3 if conditional else None
        self.layout = FlatLayout([p.shape for p in src], extra)
        n = self.layout.total
        z = lambda: torch.zeros(n, dtype=torch.float32, device=self.device)
        self.flat_params, self.flat_grads, self.exp_avg, self.exp_avg_sq = z(), z(), z(), z()
        self.params = self.layout.views(self.flat_params)
        self.grads = self.layout.views(self.flat_grads)
        for dst, s in zip(self.params, src):
            dst.copy_(s)
        self.loss3 = self.flat_grads[self.layout.loss_offset:self.layout.loss_offset + 3]
        # bucket 0 = encoders + loss scalars, bucket 1 = decoder (reduced first: its grads are ready first)
        if direct_allreduce is None:  # PSVAE_DDP_DIRECT=1: the one-shot exchange over IPC buffers instead of RCCL's all-reduce
            direct_allreduce = os.environ.get("PSVAE_DDP_DIRECT") == "1"
        self.averager = GradAverager(self.flat_grads, [0, self.layout.enc_end, n], process_group, direct=direct_allreduce) \
            if process_group is not None else None
        self.cond_tabs = self.layout.extra_views(self.flat_params)
        self.cond_gtabs = self.layout.extra_views(self.flat_grads)
        for t in self.cond_tabs:  # nn.Linear-like init of the extra one-hot columns (same seed on every rank)
            bound = 1.0 / (input_dim + _ffi.COND_DIM) ** 0.5
            t.copy_(torch.empty(t.shape).uniform_(-bound, bound))
        if process_group is not None:
            broadcast_flat(self.flat_params, 0, process_group)
        self.gender = self.age = self._cond = None
        self.flat_params_t = self.params_t = None
        if dtype == "bf16":
            if self.flags != 0 and not (batch >= 64 and input_dim in (128, 256, 512) and latent_dim in (64, 128)):
                raise ValueError("the bf16 path has the loss variants (normalised decoder output, cosine loss) on its "
                                 "row-chain shapes only: batch >= 64, input_dim in {128, 256, 512}, latent_dim in {64, 128}")
            self.ws = ops.Bf16Workspace(batch, input_dim, latent_dim, self.device)
            self.outs = (self.ws.x_hat, self.ws.mu, self.ws.logvar)
            # bf16 working copies of the weights (the row chains' packs, inside the workspace): made here, refreshed by
            # optimizer_step -- the step itself does not rebuild them (PSVAE_BF16_PACK_IN_STEP=1: it does, one launch more)
            self._pack_in_step = os.environ.get("PSVAE_BF16_PACK_IN_STEP", "0") == "1"
            self._packed_version = None
            self.refresh_packs()
        else:
            self.ws = ops.Workspace(batch, input_dim, latent_dim, self.device, normalize_decoder)
            self.outs = self.ws.new_outputs()
            # k-major copies of the weight matrices, same flat layout; made once here, then kept current by Adam
            self.flat_params_t = z()
            self.params_t = self.layout.kmajor_views(self.flat_params_t)
            self.refresh_kmajor()
        # Adam: {lr, step} live on the device (graph-safe); one fused launch over the flat buffers
        self.adam_state = torch.zeros(24, dtype=torch.uint8, device=self.device)
        self._write_lr(self.base_lr)
        self._adam_layout = self.layout.adam_layout(self.flat_params, self.flat_params_t, self.flat_grads,
                                                    self.exp_avg, self.exp_avg_sq)
        self.step_count = 0
        # device-resident Philox state {seed, offset}: advanced by the forward itself (graph-safe).  Data-parallel
        # ranks share the seed and interleave their counter ranges by GLOBAL row index: rank r starts
        # r * nblk blocks in, and after each step every rank skips the other ranks' blocks -- W ranks of B/W rows
        # draw exactly the epsilon one rank with B rows would.
        self._nblk = (batch * latent_dim + 3) // 4
        self.rng_state = torch.tensor([seed, self.rank * self._nblk], dtype=torch.int64, device=self.device)
        self._rng_skip = (self._eps_world - 1) * self._nblk
        self._graph = None
        self._adam_in_graph = False
        self._collective_in_graph = False
        self.static_x = None

    # ------------------------------------------------------------------------------------------
    def state_dict(self):
        return {k: p.detach().clone() for k, p in zip(self.keys, self.params)}

    def refresh_kmajor(self):
        """re-make the k-major weight copies from flat_params (after loading / overwriting parameters by hand; the
        engine's own optimizer keeps them current)"""
        if self.params_t is not None:
            ops.transpose_params(self.ws, self.params, self.params_t)

    def refresh_packs(self):
        """bf16 engine: re-make the bf16 working copies of the weights from flat_params.  optimizer_step does it; writes
        to the parameters through torch ops are noticed by their version counter and trigger it before the next step."""
        if self.dtype == "bf16" and not self._pack_in_step:
            ops.pack_params_bf16(self.ws, self.params)
            self._packed_version = self.flat_params._version

    def _packs_flag(self) -> int:
        if self.dtype != "bf16" or self._pack_in_step:
            return 0
        if self.flat_params._version != self._packed_version:  # parameters written by hand since the last pack
            self.refresh_packs()
        return _ffi.BF16_PACKS_CURRENT

    def _step_kwargs(self):
        return dict(params_kmajor=self.params_t, cond=self._cond, flags=self.flags)

    def launches_per_step(self) -> Optional[int]:
        """kernel launches of one forward+ELBO+backward pass as this engine issues it (measured: the library counts
        the launches of the calling thread's last psvae_train_step_f32 / bf16 pass)."""
        n = ops.last_step_launches() if self.dtype == "f32" else ops.last_step_launches_bf16()
        return n if n > 0 else None

    def forward_backward(self, x: torch.Tensor, eps: Optional[torch.Tensor] = None, _local_only: bool = False):
        self._forward_backward(x, eps, _local_only)
        if eps is None:
            self._skip_rng()

    def step_then_allreduce(self, x: torch.Tensor):
        """data-parallel step without a graph: ONE C call enqueues the whole local pass on the stream, then ONE exchange of
        the flat gradient buffer (RCCL's all-reduce, or the direct IPC exchange) behind it on the same stream"""
        self._forward_backward(x, None, _local_only=True)
        self._skip_rng()
        if self.averager is not None:
            self.averager.reduce_all()
            self.averager.wait()

    def step_overlapped(self, x: torch.Tensor):
        """data-parallel step (fp32, RCCL) with the decoder bucket's exchange UNDER the encoder backward: two C calls enqueue
        [forward + ELBO + decoder backward] and [encoder backward] on the stream; between them the decoder bucket's all-reduce
        is issued on a side stream that waits for an event behind the first half; the encoder bucket (with the loss scalars)
        follows the second half on the main stream, which then waits for the side stream.  Both all-reduces in the blocking
        form (the stream waits, not the host: reduce_all's note).  Same gradients as step_then_allreduce; which of the two is
        faster depends on the ranks' exchange time against ~25 us of extra host work -- `bench.py` measures both."""
        import torch.distributed as dist
        if self.dtype != "f32" or self.averager is None or self.averager.backend != "nccl" or self.averager.direct is not None:
            raise _ffi.PsvaeError("step_overlapped: the fp32 pass under an RCCL process group")
        if self.conditional and self._cond is None:
            raise _ffi.PsvaeError("conditional engine: call set_labels(gender, age) for this batch first")
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
            self._ev_half = torch.cuda.Event()
            self._ev_comm = torch.cuda.Event()
        main = torch.cuda.current_stream(self.device)
        kw = dict(beta=self.beta, eps=None, rng_state=self.rng_state, **self._step_kwargs())
        ops.train_step(self.ws, x, self.params, self.grads, self.outs, self.loss3, phases=_ffi.PHASE_DECODER, **kw)
        self._ev_half.record(main)
        with torch.cuda.stream(self._side):
            self._side.wait_event(self._ev_half)
            dist.all_reduce(self.averager.buckets[1], op=dist.ReduceOp.AVG, group=self.averager.pg)
            self._ev_comm.record(self._side)
        ops.train_step(self.ws, x, self.params, self.grads, self.outs, self.loss3, phases=_ffi.PHASE_ENCODER, **kw)
        self._skip_rng()
        dist.all_reduce(self.averager.buckets[0], op=dist.ReduceOp.AVG, group=self.averager.pg)
        main.wait_event(self._ev_comm)

    def _skip_rng(self):
        """jump over the other ranks' Philox blocks (a torch op: captured steps carry it)"""
        if self._rng_skip:
            self.rng_state[1:2].add_(self._rng_skip)

    def _forward_backward(self, x: torch.Tensor, eps: Optional[torch.Tensor] = None, _local_only: bool = False):
        """forward + ELBO + backward (+ overlapped gradient all-reduce when data-parallel).
        Leaves averaged gradients in self.grads and {total, recon, kl} in self.loss3."""
        if self.conditional and self._cond is None:
            raise _ffi.PsvaeError("conditional engine: call set_labels(gender, age) for this batch first")
        if self.dtype == "bf16":
            ops.train_step_bf16(self.ws, x, self.params, self.grads, self.loss3, beta=self.beta, eps=eps,
                                rng_state=self.rng_state, cond=self._cond, flags=self.flags | self._packs_flag())
            if self.averager is not None and not _local_only:
                self.averager.reduce_all()
                self.averager.wait()
            return
        kw = dict(beta=self.beta, eps=eps, rng_state=self.rng_state, **self._step_kwargs())
        if self.averager is None or _local_only:
            ops.train_step(self.ws, x, self.params, self.grads, self.outs, self.loss3, **kw)
            return
        ops.train_step(self.ws, x, self.params, self.grads, self.outs, self.loss3, phases=_ffi.PHASE_DECODER, **kw)
        self.averager.reduce_bucket(1)             # decoder grads: runs under the encoder backward
        ops.train_step(self.ws, x, self.params, self.grads, self.outs, self.loss3, phases=_ffi.PHASE_ENCODER, **kw)
        self.averager.reduce_bucket(0)             # encoder grads + the 3 loss scalars
        self.averager.wait()

    def tune(self, x: torch.Tensor, iters: int = 10, plan: Optional[str] = None, force: bool = False) -> str:
        """Install the fused step's launch plan for this batch shape (fp32 only; a no-op otherwise) and return its text
        form.  Order: an explicit `plan`; else the entry for this GPU model and (B, D, Z) in the committed plan file
        (pseudo_speaker_vae_amd/launch_plans.json, or $PSVAE_PLAN_CACHE) -- so two runs on two boxes use the SAME
        plan; else (or with force=True) measure: psvae_tune_train_step_f32 times the whole step per candidate for a few
        seconds and keeps a candidate only if it beats the heuristic plan by a margin in the same session.
        Gradients, outputs and the loss buffer are overwritten by a measurement."""
        if self.dtype != "f32":
            return "untuned"
        if plan is not None:  # reuse a plan tuned earlier (text form) instead of measuring
            ops.set_plan(self.ws, plan)
            return ops.describe_plan(self.ws)
        if self.flags != 0 or self.conditional:
            # the tuner drives the default-loss unconditional step; the variants share that shape's plan
            return ops.load_committed_plan(self.ws) or "untuned"
        return ops.tune_or_load_plan(self.ws, x, self.params, self.grads, self.outs, self.loss3, iters=iters,
                                     force=force, params_kmajor=self.params_t)

    def set_labels(self, gender: torch.Tensor, age: torch.Tensor):
        """conditional engines: integer labels in {-1,0,1,2} for the current batch (labels.py maps)"""
        assert self.conditional
        if self.gender is None:
            self.gender = torch.empty(self.B, dtype=torch.int32, device=self.device)
            self.age = torch.empty(self.B, dtype=torch.int32, device=self.device)
            self._cond = _ffi.make_cond(self.cond_tabs, self.cond_gtabs, self.gender, self.age)
        self.gender.copy_(gender.to(torch.int32))  # static buffers: a captured step sees the new labels
        self.age.copy_(age.to(torch.int32))

    # ------------------------------------------------------------------------------------------
    def _write_lr(self, lr: float):
        """one 8-byte host-to-device write into the Adam state (the captured optimizer reads lr from there)"""
        self.lr = float(lr)
        b = torch.frombuffer(bytearray(struct.pack("<d", self.lr)), dtype=torch.uint8)
        self.adam_state[:8].copy_(b)

    def scheduler_step(self, epoch: Optional[int] = None) -> float:
        """CosineAnnealingLR stepped once per epoch (lightning.py:206-213, 'interval': 'epoch').  Stepped without an
        argument it follows torch's RECURSIVE form (torch.optim.lr_scheduler.CosineAnnealingLR.get_lr), so the rate
        equals torch's at every epoch -- past T_max too (T_max comes from hparams['scheduler'] and may be smaller than
        max_epochs), where the recursion and the closed form part ways; with an explicit `epoch` it jumps there by the
        closed form eta_min + (base_lr - eta_min) (1 + cos(pi t / T_max)) / 2, as torch's `step(epoch)` does.
        Without a scheduler the rate stays base_lr."""
        if not self.scheduler:
            self.epoch = self.epoch + 1 if epoch is None else int(epoch)
            return self.lr
        T, eta_min = self.scheduler["T_max"], self.scheduler.get("eta_min", 0.0)
        if epoch is not None:
            self.epoch = int(epoch)
            lr = eta_min + (self.base_lr - eta_min) * (1 + math.cos(math.pi * self.epoch / T)) / 2
        else:
            self.epoch += 1
            t, cur = self.epoch, self.lr
            if (t - 1 - T) % (2 * T) == 0:
                lr = cur + (self.base_lr - eta_min) * (1 - math.cos(math.pi / T)) / 2
            else:
                lr = (1 + math.cos(math.pi * t / T)) / (1 + math.cos(math.pi * (t - 1) / T)) * (cur - eta_min) + eta_min
        self._write_lr(lr)
        return self.lr

    def optimizer_step(self, lr: Optional[float] = None):
        """fused Adam over the flat buffers (lightning.py:205), k-major weight copies included; the 1/world gradient
        scale is already applied by the averaging all-reduce."""
        if lr is not None and lr != self.lr:
            self._write_lr(lr)
        self.step_count += 1
        ops.adam_step_dev(self._adam_layout, self.adam_state, self.betas, self.adam_eps, self.weight_decay)
        self.refresh_packs()  # (bf16: the working copies follow the master weights; a no-op for fp32)

    def adam_steps_done(self) -> int:
        """the device-side step counter (a host sync)"""
        return struct.unpack("<q", self.adam_state[8:16].cpu().numpy().tobytes())[0]

    # ------------------------------------------------------------------------------------------
    def _persistent_state(self):
        """everything a step or an optimizer step changes and a later step reads"""
        return [t for t in (self.flat_params, self.flat_params_t, self.exp_avg, self.exp_avg_sq, self.adam_state,
                            self.rng_state) if t is not None]

    def _warm(self, body, n=2):
        """Run `body` n times outside capture (hipFuncSetAttribute, lazy module load, communicator setup) WITHOUT
        leaving a trace: parameters, k-major copies, Adam moments, the device-side {lr, step} and the Philox state are
        snapshotted before and restored after, so capturing never trains the model, never advances bias correction and
        never consumes epsilon -- the first replay is step 1 from the state the caller captured in.  Gradients, outputs
        and the loss scalars are a step's outputs and hold the warm-up's values until the first replay."""
        torch.cuda.synchronize(self.device)
        state = self._persistent_state()
        saved = [t.clone() for t in state]
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(n):
                body()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        for t, s in zip(state, saved):
            t.copy_(s)
        torch.cuda.synchronize(self.device)

    def capture(self, x: torch.Tensor, with_adam: bool = False):
        """Capture forward + ELBO + backward (and, with_adam, the optimizer step: single GPU only -- with a process
        group the all-reduce sits between them) into one HIP graph.  The captured step reads its batch from a static
        buffer (`self.static_x`) and its epsilon stream from the device-resident Philox state, which the graph itself
        advances, so every replay draws fresh noise; Adam's step counter is device-resident too.  Capturing leaves
        the model, the optimizer state and the Philox state as they were (`_warm`): N replays == N eager iterations
        from the same starting point."""
        if with_adam and self.averager is not None:
            raise _ffi.PsvaeError("capture(with_adam=True) is the single-GPU form: the all-reduce comes between")
        self.static_x = x.detach().clone()

        def body():
            self.forward_backward(self.static_x, _local_only=True)
            if with_adam:
                ops.adam_step_dev(self._adam_layout, self.adam_state, self.betas, self.adam_eps, self.weight_decay)
        self.refresh_packs()  # (bf16: the working copies follow the master weights; a no-op for fp32)
        self._warm(body)
        self._packs_flag()  # (bf16: repack now, after the warm-up state was restored -- not inside the capture)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
        self._graph = g
        self._adam_in_graph = with_adam
        self._collective_in_graph = False
        return g

    def replay(self):
        """one captured step; with a process group: followed by ONE all-reduce of the flat gradient buffer
        (the collective itself stays outside the graph)"""
        self._packs_flag()  # (bf16: parameters overwritten by hand since the last pack -> repack, outside the graph)
        self._graph.replay()
        if self.averager is not None and not self._collective_in_graph:
            self.averager.reduce_all()
            self.averager.wait()

    def capture_with_collective(self, x: torch.Tensor, overlap: bool = True):
        """Data-parallel step as ONE HIP graph, RCCL all-reduce(s) included (RCCL kernels are capturable): no
        host-side stream hand-offs per step.  overlap=True: the decoder bucket's all-reduce is a parallel branch
        of the graph under the encoder backward; else one all-reduce at the end.
        Raises if the collective cannot be captured -- callers fall back to capture() + replay()."""
        import torch.distributed as dist
        if self.averager is None or self.averager.backend != "nccl":
            raise _ffi.PsvaeError("capture_with_collective needs an RCCL (nccl) process group")
        self.static_x = x.detach().clone()
        split = overlap and self.dtype == "f32"
        kw = dict(beta=self.beta, rng_state=self.rng_state, **(self._step_kwargs() if self.dtype == "f32" else {}))

        def body():
            if split:
                ops.train_step(self.ws, self.static_x, self.params, self.grads, self.outs, self.loss3,
                               phases=_ffi.PHASE_DECODER, **kw)
                self.averager.reduce_bucket(1)
                ops.train_step(self.ws, self.static_x, self.params, self.grads, self.outs, self.loss3,
                               phases=_ffi.PHASE_ENCODER, **kw)
                self._skip_rng()
                self.averager.reduce_bucket(0)
            else:
                self.forward_backward(self.static_x, _local_only=True)
                self.averager.reduce_all()
            self.averager.wait()

        self._warm(body, 3)
        self._packs_flag()  # (bf16: repack now, after the warm-up state was restored -- not inside the capture)
        dist.barrier(group=self.averager.pg)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
        self._graph = g
        self._collective_in_graph = True
        self._adam_in_graph = False
        return g

    def capture_ddp(self, x: torch.Tensor):
        """Data-parallel variant with the collectives eager: capture the step's two halves -- [forward + ELBO + decoder
        backward] and [encoder backward] -- and replay them around the collectives: graph A, all-reduce(decoder
        bucket) on RCCL's stream, graph B underneath it, all-reduce(encoder bucket + loss scalars), wait."""
        if self.averager is None or self.dtype != "f32":
            raise _ffi.PsvaeError("capture_ddp needs a process group and the fp32 path")
        self.static_x = x.detach().clone()
        kw = dict(beta=self.beta, rng_state=self.rng_state, **self._step_kwargs())

        def warm():
            ops.train_step(self.ws, self.static_x, self.params, self.grads, self.outs, self.loss3, **kw)
            self._skip_rng()
        self._warm(warm)
        self._packs_flag()  # (bf16: repack now, after the warm-up state was restored -- not inside the capture)
        self._ga, self._gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._ga):
            ops.train_step(self.ws, self.static_x, self.params, self.grads, self.outs, self.loss3,
                           phases=_ffi.PHASE_DECODER, **kw)
        with torch.cuda.graph(self._gb, pool=self._ga.pool()):
            ops.train_step(self.ws, self.static_x, self.params, self.grads, self.outs, self.loss3,
                           phases=_ffi.PHASE_ENCODER, **kw)
            self._skip_rng()

    def replay_ddp(self):
        self._ga.replay()
        self.averager.reduce_bucket(1)
        self._gb.replay()
        self.averager.reduce_bucket(0)
        self.averager.wait()
