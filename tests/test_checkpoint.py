"""CPU: checkpoint dict layout (Train_SMT.py:325-331) and Adam-state interop with torch.optim.Adam."""
import os

import torch

from deepmerge_amd import checkpoint as ck
from deepmerge_amd.trainer import PairTrainer


class Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(5, 7)
        self.b = torch.nn.Linear(7, 3)
        self.frozen = torch.nn.Parameter(torch.ones(4), requires_grad=False)
        self.input_image_scales, self.depth, self.name = [32, 64, 128], [3, 2, 1], "S2Former_v3-3CH-3DP-SEF-321"
        self.numerics = "fp32"


def _trainer(net):
    return PairTrainer(net, criterion=lambda a, b, f: ((a - b) ** 2).mean(), adam_fn=lambda *a, **k: None)


def test_optimizer_state_round_trips_through_torch_adam(tmp_path):
    torch.manual_seed(0)
    net = Tiny()
    tr = _trainer(net)
    tr.m.normal_(); tr.v.uniform_(0.1, 1.0); tr.step_count = 17
    sd = ck.optimizer_state_dict(tr)
    ref = torch.optim.Adam(filter(lambda p: p.requires_grad, net.parameters()), lr=3e-4)
    ref.load_state_dict(sd)                               # torch accepts the layout
    back = ref.state_dict()
    assert back["param_groups"][0]["params"] == [0, 1, 2, 3] and back["param_groups"][0]["lr"] == tr.lr
    params = [p for p in net.parameters() if p.requires_grad]
    where = {id(p): o for p, o in zip(tr.fp.params, tr.fp.offsets)}
    for i, p in enumerate(params):
        o = where[id(p)]
        assert torch.equal(back["state"][i]["exp_avg"].reshape(-1), tr.m[o:o + p.numel()])
        assert torch.equal(back["state"][i]["exp_avg_sq"].reshape(-1), tr.v[o:o + p.numel()])
        assert float(back["state"][i]["step"]) == 17.0

    # a state written by torch's own Adam (only parameters that received gradients have entries) loads back
    net2 = Tiny()
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, net2.parameters()), lr=1e-4)
    net2.a(torch.randn(2, 5)).sum().backward()            # b.* never gets a gradient, like `head` upstream
    opt.step(); opt.step()
    tr2 = _trainer(net2)
    ck.load_optimizer_state_dict(tr2, opt.state_dict())
    assert tr2.step_count == 2 and tr2.lr == 1e-4
    where2 = {id(p): o for p, o in zip(tr2.fp.params, tr2.fp.offsets)}
    o = where2[id(net2.a.weight)]
    assert torch.equal(tr2.m[o:o + 35], opt.state[net2.a.weight]["exp_avg"].reshape(-1))
    o = where2[id(net2.b.weight)]
    assert float(tr2.m[o:o + 21].abs().max()) == 0.0
    # ... and writing it out again gives torch's own key set back: no entries for the parameters that never had a gradient
    again = ck.optimizer_state_dict(tr2)
    assert sorted(again["state"].keys()) == sorted(opt.state_dict()["state"].keys()) == [0, 1]
    for i in (0, 1):
        assert torch.equal(again["state"][i]["exp_avg"], opt.state_dict()["state"][i]["exp_avg"])
        assert float(again["state"][i]["step"]) == float(opt.state_dict()["state"][i]["step"]) == 2.0
    # a trainer whose moments are all zero (every gradient so far exactly zero) still keeps its step count across a save / load
    net3 = Tiny()
    tr3 = _trainer(net3)
    tr3.step_count = 5
    sd3 = ck.optimizer_state_dict(tr3)
    assert sd3["state"] == {} and sd3["param_groups"][0]["dm_step_count"] == 5
    torch.optim.Adam(filter(lambda p: p.requires_grad, net3.parameters())).load_state_dict(sd3)      # torch ignores the extra group key
    tr4 = _trainer(Tiny())
    ck.load_optimizer_state_dict(tr4, sd3)
    assert tr4.step_count == 5


def test_checkpoint_dict_keys_and_reload(tmp_path):
    torch.manual_seed(1)
    net = Tiny()
    tr = _trainer(net)
    tr.m.normal_(); tr.v.uniform_(0.1, 1.0); tr.step_count = 3
    path = os.path.join(tmp_path, "model.pth")
    state = ck.save_checkpoint(path, tr, epoch=4, elapsed=12.345)
    assert list(state.keys()) == ["net", "optimizer", "epoch", "time", "scales", "depth", "name"]     # Train_SMT.py:325-331
    assert state["time"] == 12.35 and state["scales"] == [32, 64, 128] and state["name"].startswith("S2Former_v3")
    net2 = Tiny()
    tr2 = _trainer(net2)
    meta = ck.load_checkpoint(path, net2, tr2)
    assert meta["epoch"] == 4
    for (k, v), (k2, v2) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert k == k2 and torch.equal(v, v2)
    assert tr2.step_count == 3
    for q, o in zip(tr.fp.params, tr.fp.offsets):            # (alignment padding between parameters is not state)
        n = q.numel()
        assert torch.equal(tr.m[o:o + n], tr2.m[o:o + n]) and torch.equal(tr.v[o:o + n], tr2.v[o:o + n])
    # parameters still live in the flat buffer after loading
    p = net2.a.weight
    o = {id(q): o for q, o in zip(tr2.fp.params, tr2.fp.offsets)}[id(p)]
    assert p.data_ptr() == tr2.fp.flat.data_ptr() + 4 * o
