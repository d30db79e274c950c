"""Run the cfg2 training step once per forced GEMM tile config (under rocprofv3 --kernel-trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pseudo_speaker_vae_amd import _ffi
from pseudo_speaker_vae_amd.engine import TrainEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
D = int(sys.argv[2]) if len(sys.argv) > 2 else 256
Z = int(sys.argv[3]) if len(sys.argv) > 3 else 64
eng = TrainEngine(D, Z, B, device="cuda:0")
x = torch.randn(B, D, device="cuda:0")
for tile in (1, 2, 3, 4, 5, 6, 7, 8, 9, 0):
    _ffi.lib().psvae_debug_force_tile(tile)
    for _ in range(12):
        eng.forward_backward(x)
    torch.cuda.synchronize()
