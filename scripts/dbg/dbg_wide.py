import numpy as np, torch, sys
import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,R); sys.path.insert(0,R+'/tests')
from pagnerf_amd import ops, _lib as L
import test_gpu_parity as T
dev=torch.device('cuda:0')
rs=np.random.RandomState(31)
for M,N in ((4096*3,16),(32*37+5,9),(5,2),(1000,300)):
    ridx=torch.from_numpy(np.sort(rs.randint(0,N,size=M)).astype(np.int32)).to(dev)
    counts=torch.bincount(ridx.long(),minlength=N)
    pack_start=torch.cat([torch.zeros(1,dtype=torch.int64,device=dev),torch.cumsum(counts,0)])
    rop=torch.arange(N,dtype=torch.int32,device=dev)
    x8=torch.from_numpy(rs.standard_normal(size=(8,M,8)).astype(np.float32)).to(dev); x8[:,:,6:]=0; x8=x8.bfloat16().requires_grad_(True)
    Wi,bi=T._rand_mlp(rs,(48,64,64,200))
    Wig=[w.to(dev).requires_grad_(True) for w in Wi]; big=[v.to(dev).requires_grad_(True) for v in bi]
    wts=torch.rand(M,device=dev); alpha=torch.rand(N,device=dev)
    gi=torch.from_numpy(rs.standard_normal(size=(N,200)).astype(np.float32)).to(dev)
    def inst():
        o=ops.head_composite(x8,Wig,big,wts,alpha,ridx,pack_start,rop,N,in_dim=48,out_act=L.ACT_SOFTMAX,out_dtype=torch.bfloat16,x1_grouped=(24,2))
        (o*gi).sum().backward()
    res={}
    for fused in (True,False):
        ops.WGRAD_FUSED=fused
        for p in Wig+big+[x8]: p.grad=None
        inst(); torch.cuda.synchronize()
        res[fused]=[p.grad.clone().float() for p in Wig+big+[x8]]
    names=['W0','W1','W2','b0','b1','b2','dx']
    print(M,N,[ '%s %.2e'%(n,float((a-b).norm()/(b.norm()+1e-20))) for n,a,b in zip(names,res[True],res[False])])
