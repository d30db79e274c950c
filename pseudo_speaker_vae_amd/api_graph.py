"""Graph-backed fused step for the drop-in API path (`PseudoSpeakerVAE.training_step` + `loss.backward()`,
/root/reference/ps_vae/lightning.py:67-131 as Lightning drives it).

Launched eagerly from Python the fused step is host-bound: 11 kernel launches at ~5 us of host time each plus per-call
allocations and ctypes marshalling (round 2: ~350 us per step against ~140 us of GPU work).  After a shape has been
seen `WARM_CALLS` times, `model.ELBOStepFunction` therefore runs `psvae_train_step_f32` as ONE captured HIP graph on
static buffers owned by a `GraphSlot`: per step the host copies the batch into the slot's input (one D2D copy), replays the
graph, snapshots the three loss scalars (one 12-byte copy), and -- in backward() -- scales the flat gradient buffer by the
upstream d(total) read ON THE DEVICE (one launch) and hands out fresh views of it, which autograd's AccumulateGrad adopts
as `.grad` without a copy.  Epsilon comes from a device-resident Philox state {seed, offset} that mirrors
`VAEModel._next_rng` (same seed, same offsets as the eager path: the two are interchangeable step by step) and that the
captured step itself advances.

Safety rules (each falls back to the eager path for that call, never to wrong numbers):
  * every one of the 18 parameters requires grad, lives where it lived at capture (data_ptr of each) and is fp32;
  * no parameter's `.grad` aliases the slot's gradient buffer (a caller that keeps `.grad` alive across steps --
    `zero_grad(set_to_none=False)`, gradient accumulation -- would otherwise see it overwritten by the next replay);
  * the slot's previous forward has had its backward (or its autograd graph has been freed);
  * injected epsilon, conditional tables and no-grad calls use the eager path.
`PSVAE_API_GRAPH=0` switches the mechanism off."""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch

from . import _ffi, ops

ENABLED = os.environ.get("PSVAE_API_GRAPH", "1") != "0"
WARM_CALLS = 2          # eager calls of a shape before it is captured (they also do the kernels' one-time setup)
MAX_SLOTS = 4           # distinct (shape, flags, beta) graphs kept per model


def _i64(v: int) -> int:
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


class _Pending:
    """clears the slot's `pending` mark when the autograd context that owns it goes away without a backward"""
    __slots__ = ("slot", "token")

    def __init__(self, slot, token):
        self.slot, self.token = slot, token

    def __del__(self):
        s = self.slot
        if s is not None and s.pending is self.token:
            s.pending = None


class GraphSlot:
    """static buffers + captured graph of psvae_train_step_f32 for one (B, D, Z, device, flags, beta)"""

    def __init__(self, B: int, D: int, Z: int, device, flags: int, beta: float, params: Sequence[torch.Tensor]):
        self.key = (B, D, Z, str(device), flags, beta)
        self.B, self.D, self.Z, self.flags, self.beta = B, D, Z, flags, beta
        self.ws = ops.Workspace(B, D, Z, device, bool(flags & _ffi.NORMALIZE_DECODER))
        self.acts = self.ws.acts                       # (tests read epsilon and the ReLU masks from here)
        self.x = torch.empty(B, D, dtype=torch.float32, device=device)
        self.outs = self.ws.new_outputs()
        self.loss3 = torch.empty(3, dtype=torch.float32, device=device)
        self.shapes = [p.shape for p in params]
        self.numels = [p.numel() for p in params]
        self.pads = [(n + 63) // 64 * 64 for n in self.numels]
        self.flat = torch.empty(sum(self.pads), dtype=torch.float32, device=device)
        self._gviews = self._views()                   # the graph's own gradient pointers (never handed out)
        self.param_ptrs = tuple(p.data_ptr() for p in params)
        self.lo, self.hi = self.flat.data_ptr(), self.flat.data_ptr() + self.flat.numel() * 4
        self.rng = torch.zeros(2, dtype=torch.int64, device=device)
        self._rng_host = torch.zeros(2, dtype=torch.int64).pin_memory()
        self.dev_rng = None                            # (seed, offset) the device state holds, as far as the host knows
        self.nblk = (B * Z + 3) // 4
        self.skip = 0
        self.graph = None
        self.pending = None
        self.launches = None

    def _views(self) -> List[torch.Tensor]:
        out = []
        for v, n, p, s in zip(self.flat.split_with_sizes(self.pads), self.numels, self.pads, self.shapes):
            if n != p:
                v = v[:n]
            out.append(v.view(s) if len(s) != 1 else v)
        return out

    # --------------------------------------------------------------------------------------------
    def usable(self, params: Sequence[torch.Tensor]) -> bool:
        if self.pending is not None:
            return False
        lo, hi = self.lo, self.hi
        for p, ptr in zip(params, self.param_ptrs):
            if p.data_ptr() != ptr:
                return False
            g = p.grad
            if g is not None and lo <= g.data_ptr() < hi:
                return False
        return True

    def capture(self, params: Sequence[torch.Tensor], world: int):
        """capture one step (replays do the work; capturing executes nothing)"""
        self.skip = (world - 1) * self.nblk            # the other data-parallel ranks' Philox blocks (VAEModel._next_rng)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            ops.train_step(self.ws, self.x, params, self._gviews, self.outs, self.loss3, beta=self.beta,
                           rng_state=self.rng, flags=self.flags)
            if self.skip:
                self.rng[1:2].add_(self.skip)
        self.launches = ops.last_step_launches()
        self.graph = g

    def run(self, x: torch.Tensor, seed: int, offset: int, token) -> torch.Tensor:
        """one step on batch `x` with the Philox position (seed, offset); returns a private copy of {total, recon, kl}"""
        want = (_i64(seed), _i64(offset))
        if self.dev_rng != want:                       # first replay, torch.manual_seed, or other forwards in between
            self._rng_host[0], self._rng_host[1] = want
            self.rng.copy_(self._rng_host, non_blocking=True)
            torch.cuda.current_stream().synchronize()  # the pinned staging buffer is rewritten next time (rare path)
        self.x.copy_(x)
        self.graph.replay()
        self.dev_rng = (want[0], _i64(want[1] + self.nblk + self.skip))
        self.pending = token
        return self.loss3.clone()

    def scaled_grads(self, g_total: torch.Tensor) -> List[torch.Tensor]:
        """gradients of this slot's last step times the upstream d(total) (a device scalar; no host sync), as fresh views
        of the flat buffer that nothing else references (AccumulateGrad adopts such a tensor as `.grad` as it is)"""
        self.flat.mul_(g_total)
        self.pending = None
        return self._views()


class GraphSteps:
    """per-model table of slots; decides, call by call, whether the graph path may serve it"""

    def __init__(self):
        self.slots = {}
        self.seen = {}
        self.enabled = ENABLED
        self.served = 0      # calls the graph path has served (tests, bench)

    def invalidate(self):
        self.slots.clear()
        self.seen.clear()

    def lookup(self, x: torch.Tensor, Z: int, flags: int, beta: float, params: Sequence[torch.Tensor],
               world: int) -> Optional[GraphSlot]:
        """the slot that can serve this call, capturing it on the call after the warm-up ones; else None"""
        if not self.enabled:
            return None
        B, D = x.shape
        key = (B, D, Z, x.device, flags, beta, world)
        slot = self.slots.get(key)
        if slot is None:
            n = self.seen.get(key, 0)
            if n < WARM_CALLS:
                self.seen[key] = n + 1
                return None
            if torch.cuda.is_current_stream_capturing():
                return None
            if len(self.slots) >= MAX_SLOTS:
                self.slots.pop(next(iter(self.slots)))
            slot = GraphSlot(B, D, Z, x.device, flags, beta, params)
            slot.capture(params, world)
            self.slots[key] = slot
        elif not slot.usable(params):
            if any(p.data_ptr() != q for p, q in zip(params, slot.param_ptrs)):  # the parameters moved: start over
                del self.slots[key]
                self.seen[key] = 0
            return None
        self.served += 1
        return slot
