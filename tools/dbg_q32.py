"""Where does the 32-row attention forward differ from an fp64 reference?  (debug aid)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from deepmerge_amd import ops
dev = "cuda:0"
B, N, H, D = int(os.environ.get("B", 8)), int(os.environ.get("N", 256)), int(os.environ.get("H", 12)), 64
rng = np.random.default_rng(0)
qkv = torch.from_numpy(rng.normal(size=(B, N, 3, H, D)).astype(np.float32)).to(torch.bfloat16)
bias = None if os.environ.get("NOBIAS") == "1" else torch.from_numpy(rng.normal(size=(H, N, N)).astype(np.float32))
q, k, v = [qkv[:, :, i].permute(0, 2, 1, 3).double() for i in range(3)]
s = (q * 0.125) @ k.transpose(-1, -2)
if bias is not None:
    s = s + bias[None].double()
p = torch.softmax(s, -1)
o_ref = (p @ v).permute(0, 2, 1, 3).reshape(B, N, H * D)
lse_ref = torch.logsumexp(s, -1)
for rep in range(int(os.environ.get("REPS", 2))):
    out, lse = ops.attention_fwd(qkv.to(dev), None if bias is None else bias.to(dev), B, N, H, D, 0.125)
    torch.cuda.synchronize()
    el = (lse.cpu().double() - lse_ref).abs()
    el[~torch.isfinite(lse.cpu().double())] = 1e9
    bad = (el > 2e-2).nonzero()
    print(f"rep {rep}: lse bad {len(bad)} of {el.numel()}")
    if len(bad):
        bb, hh_, qq = bad[:, 0], bad[:, 1], bad[:, 2]
        print("  bad by b:", np.bincount(bb.numpy(), minlength=B))
        print("  bad by h:", np.bincount(hh_.numpy(), minlength=H))
        print("  bad by q//32:", np.bincount((qq // 32).numpy(), minlength=(N + 31) // 32))
        print("  bad by q%32:", np.bincount((qq % 32).numpy(), minlength=32))
        print("  first:", bad[:10].tolist(), [float(lse[tuple(x)]) for x in bad[:5].tolist()], [float(lse_ref[tuple(x)]) for x in bad[:5].tolist()])
    eo = (out.float().cpu().double() - o_ref).abs().reshape(B, N, H, D)
    eo[~torch.isfinite(out.float().cpu()).reshape(B, N, H, D)] = 1e9
    bado = (eo.amax(-1) > 2e-2).nonzero()
    print(f"  out rows bad {len(bado)} of {B * N * H}; max err among finite {float(eo[eo < 1e8].max()):.4f}")
    if len(bado):
        print("  out bad by q//32:", np.bincount((bado[:, 1] // 32).numpy(), minlength=(N + 31) // 32), "by q%32:", np.bincount((bado[:, 1] % 32).numpy(), minlength=32))
        x = bado[0].tolist()
        print("  first bad row", x, "err per d:", [round(float(e), 3) for e in eo[x[0], x[1], x[2]]])
