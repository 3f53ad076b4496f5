import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from deepmerge_amd import ops
from microbench import timeit
from deepmerge_amd.nets.ShfitScaleFormer import relative_position_index
DEV="cuda:0"; dt=torch.bfloat16
B,N,H,D=64,256,12,64
qkv=torch.randn(B,N,3,H,D,device=DEV).to(dt); nb=1575
table=torch.randn(nb,H,device=DEV)
ref=relative_position_index([4,8,8]).to(torch.int32).to(DEV)
for name,idx in (("none",None),("random",torch.randint(0,nb,(N,N),device=DEV,dtype=torch.int32)),("zeros",torch.zeros(N,N,device=DEV,dtype=torch.int32)),("relpos",ref),("distinct-per-lane",(torch.arange(N*N,device=DEV,dtype=torch.int32)%nb).reshape(N,N))):
    bias,bias_t=ops.relpos_bias_gather(table, ref, N, transposed=True)
    out,lse=ops.attention_fwd(qkv,bias,B,N,H,D,0.125); dout=torch.randn_like(out)
    t=timeit(lambda: ops.attention_bwd(qkv,bias,out,dout,lse,B,N,H,D,0.125,idx,nb if idx is not None else 0,bias_t=bias_t))
    print(f"{name:20s} bwd {t*1e6:8.1f} us",flush=True)
