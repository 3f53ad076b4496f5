import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from test_gpu_dp import _build, _batch
from deepmerge_amd.trainer import PairTrainer
left, ld, right, rd, flag = _batch(8); mv = lambda t: t.to("cuda:0")
batch = ([mv(t) for t in left], mv(ld), [mv(t) for t in right], mv(rd), mv(flag))
net = _build("fp32"); tr = PairTrainer(net, lr=1e-4); tr.step(*batch)
ref = {n: p.grad.cpu().clone() for n, p in net.named_parameters()}
# same, but with the DP hooks installed and the exchange stubbed out
net2 = _build("fp32"); tr2 = PairTrainer(net2, lr=1e-4, n_buckets=3)
tr2.bucket_slices = tr2.fp.buckets(3)
launched = []
tr2._launch_bucket = lambda bi: launched.append((bi, float(tr2.fp.grad[tr2.bucket_slices[bi]].abs().sum())))
tr2._install_bucket_hooks()
tr2.net.train(); tr2.fp.zero_grad()
fa, fb = tr2.net(*batch[:4]); loss = tr2.criterion(fa, fb, batch[4]); loss.backward()
torch.cuda.synchronize()
print("launched", launched, "remaining", tr2._remaining)
bad = sorted([(float((p.grad.cpu() - ref[n]).abs().max() / (ref[n].abs().max() + 1e-12)), n) for n, p in net2.named_parameters()], reverse=True)
print("worst", [(round(a, 4), n) for a, n in bad[:6]])
final = [float(tr2.fp.grad[sl].abs().sum()) for sl in tr2.bucket_slices]
print("bucket abs-sums at end", final)
