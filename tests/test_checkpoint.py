"""Checkpoint files in the reference layout (idr_train.py:164-184): a file written the way the reference writes it loads into the drop-in
model, and save -> load round-trips model, optimizer and scheduler state."""
import os

import numpy as np
import pytest
import torch

from mvsdf_amd import checkpoint as ck
from mvsdf_amd.model.implicit_differentiable_renderer import IDRNetwork
from mvsdf_amd.utils import synth
from mvsdf_amd.utils.config import ConfigDict


def _model(seed):
    m = IDRNetwork(ConfigDict(synth.model_conf(64)))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(64, seed).items()})
    return m


def test_reference_shaped_file_loads_and_round_trips(tmp_path):
    # a ModelParameters file exactly as the reference writes it: {"epoch", "model_state_dict"} with its key layout (idr.py:70-73)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(64, 5).items()}
    assert list(sd)[:3] == ['implicit_network.lin0.bias', 'implicit_network.lin0.weight_g', 'implicit_network.lin0.weight_v']
    os.makedirs(tmp_path / ck.MODEL_SUBDIR)
    torch.save({'epoch': 123, 'model_state_dict': sd}, tmp_path / ck.MODEL_SUBDIR / '123.pth')
    m = _model(0)
    assert ck.load_checkpoints(str(tmp_path), m, checkpoint=123) == 123
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k
    # save with optimizer + scheduler, reload into fresh objects
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, [2, 4], gamma=0.1)
    for p in m.parameters():
        p.grad = torch.full_like(p, 0.01)
    for _ in range(3):
        opt.step(); sch.step()
    ck.save_checkpoints(str(tmp_path / 'out'), 7, m, opt, sch)
    for sub in (ck.MODEL_SUBDIR, ck.OPTIMIZER_SUBDIR, ck.SCHEDULER_SUBDIR):
        assert sorted(os.listdir(tmp_path / 'out' / sub)) == ['7.pth', 'latest.pth']
    assert set(torch.load(tmp_path / 'out' / ck.MODEL_SUBDIR / 'latest.pth').keys()) == {'epoch', 'model_state_dict'}
    m2 = _model(1)
    opt2 = torch.optim.Adam(m2.parameters(), lr=1e-3)
    sch2 = torch.optim.lr_scheduler.MultiStepLR(opt2, [2, 4], gamma=0.1)
    assert ck.load_checkpoints(str(tmp_path / 'out'), m2, opt2, sch2) == 7
    for (k, a), b in zip(m.state_dict().items(), m2.state_dict().values()):
        assert torch.equal(a, b), k
    assert opt2.param_groups[0]['lr'] == opt.param_groups[0]['lr'] == pytest.approx(1e-4)
    assert sch2.last_epoch == 3
    s1, s2 = opt.state_dict()['state'], opt2.state_dict()['state']
    assert all(torch.equal(s1[i]['exp_avg'], s2[i]['exp_avg']) for i in s1)


@pytest.mark.gpu
def test_flat_adam_checkpoint_round_trip_on_device(tmp_path):
    from mvsdf_amd.optim import FlatAdam
    m = _model(0).cuda()
    opt = FlatAdam(m.parameters(), lr=1e-3)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, [1], gamma=0.5)
    for _ in range(2):
        opt.zero_grad()
        for p in m.parameters():
            p.grad.fill_(0.02)
        opt.step(); sch.step()
    ck.save_checkpoints(str(tmp_path), 2, m, opt, sch)
    m2 = _model(3).cuda()
    ref_opt = torch.optim.Adam(m2.parameters(), lr=1e-3)                       # the reference's optimizer reads FlatAdam's file
    assert ck.load_checkpoints(str(tmp_path), m2, ref_opt, checkpoint='latest') == 2
    assert float(ref_opt.state_dict()['state'][0]['step']) == 2.0
    # ... and can STEP from it (its step() reads weight_decay / amsgrad from the loaded group: FlatAdam writes a complete Adam group)
    assert ref_opt.param_groups[0]['weight_decay'] == 0 and ref_opt.param_groups[0]['amsgrad'] is False
    before = [p.detach().clone() for p in m2.parameters()]
    for p in m2.parameters():
        p.grad = torch.full_like(p, 0.02)
    ref_opt.step()
    assert float(ref_opt.state_dict()['state'][0]['step']) == 3.0
    assert any(not torch.equal(a, p.detach()) for a, p in zip(before, m2.parameters()))
    ck.load_checkpoints(str(tmp_path), m2, checkpoint='latest')                # (restore the parameters for the FlatAdam round trip below)
    opt3 = FlatAdam(m2.parameters(), lr=1e-3)
    ck.load_checkpoints(str(tmp_path), m2, opt3)
    assert torch.equal(opt3.flat_m, opt.flat_m) and torch.equal(opt3.flat_p, opt.flat_p) and opt3._t == 2
