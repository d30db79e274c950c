import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from pseudo_speaker_vae_amd import ops, _ffi
from oracle import psvae_oracle as O
D,Z,B = 512,128,1024
p = O.init_params(D,Z,2)
torch.manual_seed(B+D)
x, eps = torch.randn(B,D), torch.randn(B,Z)
losses, grads, acts, extra = O.train_step(p,x,eps,0.8,False,False,go=0.37)
dev="cuda:0"
params=[p[k].to(dev) for k in O.PARAM_KEYS]
for tile in (0,0x11,0x21,0x12,0x22,0x44):
    _ffi.lib().psvae_debug_force_tile(tile)
    ws=ops.Workspace(B,D,Z,dev)
    xd=x.to(dev)
    xh,mu,lv=ops.vae_forward(ws,xd,params,eps.to(dev))
    g=[torch.full_like(t,float('nan')) for t in params]
    go=torch.tensor([0.37],device=dev)
    ops.vae_backward(ws,xd,params,xh,mu,lv,g,elbo_grad=True,beta=0.8,grad_loss=go)
    torch.cuda.synchronize()
    def rel(a,b): 
        a=a.double().cpu(); b=b.double()
        return float((a-b).abs().max()/b.abs().max())
    print(hex(tile), "xh",rel(xh,acts["x_hat"]), " ".join(f"{k.split('.')[0][:5]}{k.split('.')[1]}{k.split('.')[2][0]}:{rel(gg,grads[k]):.1e}" for k,gg in zip(O.PARAM_KEYS,g) if 'decoder' in k))
