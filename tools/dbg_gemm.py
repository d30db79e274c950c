import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pseudo_speaker_vae_amd import ops
torch.manual_seed(0)
dev="cuda:0"
for (M,N,K) in ((1024,512,512),(1024,512,256),(256,64,512),(64,64,384)):
    A=torch.randn(M,K,device=dev); B=torch.randn(K,N,device=dev)
    ref=(A.double()@B.double())
    for tile in (0x11,0x12,0x21,0x22):
        C=ops.gemm(1,A,B,tile=tile)
        err=(C.double()-ref).abs()
        bad=(err>1e-2).nonzero()
        print("NN",M,N,K,hex(tile),"max err",err.max().item(),"nbad",len(bad), bad[:6].tolist())
