"""TrainEngine: the native training-step driver for the hot path.

One process per GPU.  Parameters, gradients and Adam state live in flat fp32 buffers (the 18 tensors
in the reference's creation order, model.py:14-36, each 256-B aligned), so that
  * the HIP kernels write gradients straight into the all-reduce payload,
  * data-parallel averaging (the reference's DDPStrategy, training.py:78) is TWO RCCL all-reduces
    per step: the decoder bucket is issued as soon as backward phase 1 has been enqueued and runs
    on RCCL's stream underneath the encoder backward (phase 2); the encoder bucket carries the three
    loss scalars too (replaces Lightning's per-key `sync_dist` all-reduces, lightning.py:127-129),
  * Adam (lightning.py:205) is one fused kernel over the flat buffer with the 1/world scale folded in.
On one GPU the whole step (forward + ELBO + backward [+ Adam]) can be captured into a HIP graph.
"""
from __future__ import annotations

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

    def extra_views(self, flat: torch.Tensor) -> List[torch.Tensor]:
        return [flat[o:o + s.numel()].view(s) for o, s in zip(self.extra_offsets, self.extra_shapes)]


class TrainEngine:
    def __init__(self, input_dim: int, latent_dim: int, batch: int, device="cuda:0", seed: int = 0,
                 beta: float = 1.0, normalize_decoder: bool = False, use_cos_loss: bool = False,
                 lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 process_group=None, state_dict=None, dtype: str = "f32", conditional: bool = False,
                 shard: Optional[tuple] = None):
        """dtype "bf16" = BASELINE config 3 arithmetic (bf16 MFMA operands, fp32 accumulate / master weights /
        gradients); conditional = one-hot gender+age gather tables."""
        if dtype not in ("f32", "bf16"):
            raise ValueError(dtype)
        if conditional and dtype != "bf16":
            raise ValueError("the conditional engine path is wired for dtype='bf16' (config 3); "
                             "fp32 conditioning is available through ConditionalVAEModel")
        self.dtype, self.conditional = dtype, conditional
        self.D, self.Z, self.B = input_dim, latent_dim, batch
        self.device = torch.device(device)
        self.beta = float(beta)
        self.flags = (_ffi.NORMALIZE_DECODER if normalize_decoder else 0) | (_ffi.COS_LOSS if use_cos_loss else 0)
        self.lr, self.betas, self.adam_eps, self.weight_decay = lr, betas, eps, weight_decay
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
        self.flat_params = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.flat_grads = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.params = self.layout.views(self.flat_params)
        self.grads = self.layout.views(self.flat_grads)
        for dst, s in zip(self.params, src):
            dst.copy_(s)
        self.loss3 = self.flat_grads[self.layout.loss_offset:self.layout.loss_offset + 3]
        # bucket 0 = encoders + loss scalars, bucket 1 = decoder (reduced first: its grads are ready first)
        self.averager = GradAverager(self.flat_grads, [0, self.layout.enc_end, n], process_group) \
            if process_group is not None else None
        if process_group is not None:
            broadcast_flat(self.flat_params, 0, process_group)
        self.cond_tabs = self.layout.extra_views(self.flat_params)
        self.cond_gtabs = self.layout.extra_views(self.flat_grads)
        for t in self.cond_tabs:  # nn.Linear-like init of the extra one-hot columns (same seed on every rank)
            bound = 1.0 / (input_dim + _ffi.COND_DIM) ** 0.5
            t.copy_(torch.empty(t.shape).uniform_(-bound, bound))
        self.gender = self.age = self._cond = None
        if dtype == "bf16":
            if self.flags != 0:
                raise ValueError("the bf16 path implements the default loss (MSE/10 + beta*KL)")
            self.ws = ops.Bf16Workspace(batch, input_dim, latent_dim, self.device)
            self.outs = (self.ws.x_hat, self.ws.mu, self.ws.logvar)
        else:
            self.ws = ops.Workspace(batch, input_dim, latent_dim, self.device, normalize_decoder)
            self.outs = self.ws.new_outputs()
        self.step_count = 0
        # device-resident Philox state {seed, offset}: advanced by the forward itself (graph-safe).  Data-parallel
        # ranks share the seed and interleave their counter ranges by GLOBAL row index: rank r starts
        # r * nblk blocks in, and after each step every rank skips the other ranks' blocks -- W ranks of B/W rows
        # draw exactly the epsilon one rank with B rows would.
        self._nblk = (batch * latent_dim + 3) // 4
        self.rng_state = torch.tensor([seed, self.rank * self._nblk], dtype=torch.int64, device=self.device)
        self._rng_skip = (self._eps_world - 1) * self._nblk
        self._graph = None
        self.static_x = None

    # ------------------------------------------------------------------------------------------
    def state_dict(self):
        return {k: p.detach().clone() for k, p in zip(self.keys, self.params)}

    def forward_backward(self, x: torch.Tensor, eps: Optional[torch.Tensor] = None, _local_only: bool = False):
        self._forward_backward(x, eps, _local_only)
        if eps is None:
            self._skip_rng()

    def _skip_rng(self):
        """jump over the other ranks' Philox blocks (a torch op: captured steps carry it)"""
        if self._rng_skip:
            self.rng_state[1:2].add_(self._rng_skip)

    def _forward_backward(self, x: torch.Tensor, eps: Optional[torch.Tensor] = None, _local_only: bool = False):
        """forward + ELBO + backward (+ overlapped gradient all-reduce when data-parallel).
        Leaves averaged gradients in self.grads and {total, recon, kl} in self.loss3."""
        if self.dtype == "bf16":
            ops.train_step_bf16(self.ws, x, self.params, self.grads, self.loss3, beta=self.beta, eps=eps,
                                rng_state=self.rng_state, cond=self._cond)
            if self.averager is not None and not _local_only:
                self.averager.reduce_all()
                self.averager.wait()
            return
        if self.flags == 0:  # default loss: the fused 13-launch sequence
            kw = dict(beta=self.beta, eps=eps, rng_state=self.rng_state)
            if self.averager is None or _local_only:
                ops.train_step(self.ws, x, self.params, self.grads, self.outs, self.loss3, **kw)
                return
            ops.train_step(self.ws, x, self.params, self.grads, self.outs, self.loss3, phases=_ffi.PHASE_DECODER, **kw)
            self.averager.reduce_bucket(1)             # decoder grads: runs under the encoder backward
            ops.train_step(self.ws, x, self.params, self.grads, self.outs, self.loss3, phases=_ffi.PHASE_ENCODER, **kw)
            self.averager.reduce_bucket(0)             # encoder grads + the 3 loss scalars
            self.averager.wait()
            return
        xh, mu, lv = ops.vae_forward(self.ws, x, self.params, eps, 0, 0, self.flags, outs=self.outs,
                                     rng_state=self.rng_state)
        ops.elbo(self.ws, x, xh, mu, lv, self.beta, self.flags, self.loss3)
        if self.averager is None or _local_only:
            ops.vae_backward(self.ws, x, self.params, xh, mu, lv, self.grads, elbo_grad=True, beta=self.beta,
                             flags=self.flags)
            return
        ops.vae_backward(self.ws, x, self.params, xh, mu, lv, self.grads, elbo_grad=True, beta=self.beta,
                         flags=self.flags, phases=_ffi.PHASE_DECODER)
        self.averager.reduce_bucket(1)                 # decoder grads: runs under the encoder backward
        ops.vae_backward(self.ws, x, self.params, xh, mu, lv, self.grads, elbo_grad=True, beta=self.beta,
                         flags=self.flags, phases=_ffi.PHASE_ENCODER)
        self.averager.reduce_bucket(0)                 # encoder grads + the 3 loss scalars
        self.averager.wait()

    def tune(self, x: torch.Tensor, iters: int = 10, plan: Optional[str] = None) -> str:
        """Autotune the fused step's launch plan for this batch shape (fp32, default loss, unconditional only;
        a no-op otherwise).  Takes a few seconds once (or no time with a plan cache file, $PSVAE_PLAN_CACHE); later
        steps -- eager or captured -- use the plan.  Gradients,
        outputs and the loss buffer are overwritten.  Returns the plan's text form."""
        if self.dtype != "f32" or self.flags != 0 or self.conditional:
            return "untuned"
        if plan is not None:  # reuse a plan tuned earlier (text form) instead of measuring
            ops.set_plan(self.ws, plan)
            return ops.describe_plan(self.ws)
        return ops.tune_or_load_plan(self.ws, x, self.params, self.grads, self.outs, self.loss3, iters=iters)

    def set_labels(self, gender: torch.Tensor, age: torch.Tensor):
        """conditional engines: integer labels in {-1,0,1,2} for the current batch (labels.py maps)"""
        assert self.conditional
        self.gender = gender.to(device=self.device, dtype=torch.int32).contiguous()
        self.age = age.to(device=self.device, dtype=torch.int32).contiguous()
        self._cond = _ffi.make_cond(self.cond_tabs, self.cond_gtabs, self.gender, self.age)

    def optimizer_step(self, lr: Optional[float] = None):
        """fused Adam over the flat buffers (lightning.py:205)."""
        self.step_count += 1
        ops.adam_step(self.flat_params, self.flat_grads, self.exp_avg, self.exp_avg_sq, self.step_count,
                      self.lr if lr is None else lr, self.betas, self.adam_eps, self.weight_decay)
        # the loss slot inside flat_grads is "trained" too but belongs to no parameter: harmless

    # ------------------------------------------------------------------------------------------
    def capture(self, x: torch.Tensor):
        """Capture forward + ELBO + backward into one HIP graph (single GPU; launch-bound otherwise:
        13-19 kernels of a few microseconds each).  The captured step reads its batch from a static
        buffer (`self.static_x`) and its epsilon stream from the device-resident Philox state, which
        the graph itself advances, so every replay draws fresh noise."""
        self.static_x = x.detach().clone()
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(2):  # warm-up outside capture (hipFuncSetAttribute, lazy module load)
                self.forward_backward(self.static_x, _local_only=True)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.forward_backward(self.static_x, _local_only=True)
        self._graph = g
        return g

    def replay(self):
        """one captured step; with a process group: followed by ONE all-reduce of the flat gradient buffer
        (the collective itself stays outside the graph)"""
        self._graph.replay()
        if self.averager is not None and not getattr(self, "_collective_in_graph", False):
            self.averager.reduce_all()
            self.averager.wait()

    def capture_with_collective(self, x: torch.Tensor, overlap: bool = True):
        """Data-parallel step as ONE HIP graph, RCCL all-reduce(s) included (RCCL kernels are capturable): no
        host-side stream hand-offs per step.  overlap=True: the decoder bucket's all-reduce is a parallel branch
        of the graph under the encoder backward (default loss only); else one all-reduce at the end.
        Raises if the collective cannot be captured -- callers fall back to capture() + replay()."""
        import torch.distributed as dist
        if self.averager is None or self.averager.backend != "nccl":
            raise _ffi.PsvaeError("capture_with_collective needs an RCCL (nccl) process group")
        self.static_x = x.detach().clone()
        split = overlap and self.dtype == "f32" and self.flags == 0
        kw = dict(beta=self.beta, rng_state=self.rng_state)

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

        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(3):  # warm-up outside capture: communicator setup, lazy module load
                body()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        dist.barrier(group=self.averager.pg)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
        self._graph = g
        self._collective_in_graph = True
        return g

    def capture_ddp(self, x: torch.Tensor):
        """Data-parallel variant: the step cannot be ONE graph (the RCCL all-reduces stay eager), so capture
        its two halves -- [forward + ELBO + decoder backward] and [encoder backward] -- and replay them around
        the collectives: graph A, all-reduce(decoder bucket) on RCCL's stream, graph B underneath it,
        all-reduce(encoder bucket + loss scalars), wait.  Host cost per step: two graph launches + two
        collective calls instead of 13 kernel launches."""
        if self.averager is None or self.flags != 0:
            raise _ffi.PsvaeError("capture_ddp needs a process group and the default loss")
        self.static_x = x.detach().clone()
        kw = dict(beta=self.beta, rng_state=self.rng_state)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(2):
                ops.train_step(self.ws, self.static_x, self.params, self.grads, self.outs, self.loss3, **kw)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
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
