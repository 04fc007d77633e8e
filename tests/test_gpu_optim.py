"""FlatAdam (grad-norm + clip + Adam on flat buffers) vs torch.optim.Adam + clip_grad_norm_ (idr_train.py:113,289-302)."""
import copy

import pytest
import torch

from mvsdf_amd.optim import FlatAdam

pytestmark = pytest.mark.gpu


def _models():
    torch.manual_seed(3)
    a = torch.nn.Sequential(torch.nn.Linear(37, 64), torch.nn.Softplus(beta=100), torch.nn.Linear(64, 5)).cuda()
    return a, copy.deepcopy(a)


@pytest.mark.parametrize('cap', [None, 0.05, 1e6])
def test_flat_adam_matches_torch_adam(cap):
    ref, mine = _models()
    o_ref = torch.optim.Adam(ref.parameters(), lr=1e-3)
    o_mine = FlatAdam(mine.parameters(), lr=1e-3)
    s_ref = torch.optim.lr_scheduler.MultiStepLR(o_ref, [3], gamma=0.5)
    s_mine = torch.optim.lr_scheduler.MultiStepLR(o_mine, [3], gamma=0.5)       # schedulers drive param_groups like any Optimizer
    x = torch.randn(128, 37, device='cuda')
    for it in range(6):
        for m, o in ((ref, o_ref), (mine, o_mine)):
            o.zero_grad()
            (m(x) ** 2).mean().backward()
        norm_ref = torch.cat([p.grad.flatten() for p in ref.parameters()]).norm()
        if cap:
            torch.nn.utils.clip_grad_norm_(ref.parameters(), cap)
        o_ref.step()
        o_mine.step(grad_cap=cap)
        assert abs(float(o_mine.grad_norm()) - float(norm_ref)) <= 1e-5 * float(norm_ref)
        for p, q in zip(ref.parameters(), mine.parameters()):
            assert torch.allclose(p, q, rtol=0, atol=2e-6), it
            assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-9)           # clipped gradients are written back
        s_ref.step(); s_mine.step()
    assert o_mine.param_groups[0]['lr'] == o_ref.param_groups[0]['lr'] == 5e-4


def test_flat_adam_checkpoint_layout_round_trip():
    ref, mine = _models()
    o_ref = torch.optim.Adam(ref.parameters(), lr=1e-3)
    o_mine = FlatAdam(mine.parameters(), lr=1e-3)
    x = torch.randn(64, 37, device='cuda')
    for m, o in ((ref, o_ref), (mine, o_mine)):
        for _ in range(2):
            o.zero_grad(); (m(x) ** 2).mean().backward(); o.step()
    sd = o_mine.state_dict()
    assert sd['state'].keys() == o_ref.state_dict()['state'].keys()
    assert set(sd['state'][0].keys()) == {'step', 'exp_avg', 'exp_avg_sq'} and float(sd['state'][0]['step']) == 2.0
    # a torch.optim.Adam checkpoint loads into FlatAdam and continues identically
    ref2, mine2 = _models()
    for dst, src in ((ref2, ref), (mine2, ref)):
        dst.load_state_dict(src.state_dict())
    o_ref2 = torch.optim.Adam(ref2.parameters(), lr=1e-3)
    o_mine2 = FlatAdam(mine2.parameters(), lr=1e-3)
    o_ref2.load_state_dict(o_ref.state_dict())
    o_mine2.load_state_dict(o_ref.state_dict())
    for m, o in ((ref2, o_ref2), (mine2, o_mine2)):
        o.zero_grad(); (m(x) ** 2).mean().backward(); o.step()
    for p, q in zip(ref2.parameters(), mine2.parameters()):
        assert torch.allclose(p, q, rtol=0, atol=2e-6)
    # parameters still live in the flat buffer after load_state_dict of the MODEL
    mine2.load_state_dict(ref.state_dict())
    base = o_mine2.flat_p.data_ptr()
    assert all(base <= p.data_ptr() < base + 4 * o_mine2.flat_p.numel() for p in mine2.parameters())


def test_detached_grads_are_picked_up_again():
    """model.zero_grad() (set_to_none=True by default) detaches every .grad from the flat buffer: autograd then allocates fresh tensors.
    FlatAdam.step() copies them in and re-attaches instead of applying stale values; a parameter left without a gradient is treated as
    a zero gradient (torch 1.7.1's zero_grad semantics, the reference's pinned version)."""
    ref, mine = _models()
    o_ref = torch.optim.Adam(ref.parameters(), lr=1e-3)
    o_mine = FlatAdam(mine.parameters(), lr=1e-3)
    x = torch.randn(128, 37, device='cuda')
    for it in range(4):
        ref.zero_grad(set_to_none=False)
        mine.zero_grad()                                            # nn.Module.zero_grad: p.grad = None
        assert all(p.grad is None for p in mine.parameters())
        for m in (ref, mine):
            (m(x) ** 2).mean().backward()
        if it == 2:                                                 # one parameter without a gradient this step
            list(mine.parameters())[-1].grad = None
            list(ref.parameters())[-1].grad.zero_()
        o_ref.step()
        o_mine.step()
        base = o_mine.flat_g.data_ptr()
        assert all(base <= p.grad.data_ptr() < base + 4 * o_mine.flat_g.numel() for p in mine.parameters())
        for p, q in zip(ref.parameters(), mine.parameters()):
            assert torch.allclose(p, q, rtol=0, atol=2e-6), it


def test_grad_scale_is_the_division_by_world():
    """mvsdf_adam_step_scaled(grad_scale = 1/4) == dividing the gradient by 4 first (norm, clip and update)."""
    ref, mine = _models()
    o_ref = FlatAdam(ref.parameters(), lr=1e-3)
    o_mine = FlatAdam(mine.parameters(), lr=1e-3)
    x = torch.randn(128, 37, device='cuda')
    for it in range(3):
        for m, o in ((ref, o_ref), (mine, o_mine)):
            o.zero_grad()
            (m(x) ** 2).mean().backward()
        o_ref.flat_g.div_(4.0)
        o_mine._grad_scale = 0.25
        o_ref.step(grad_cap=0.01)
        o_mine.step(grad_cap=0.01)
        assert o_mine._grad_scale == 1.0
        assert torch.equal(o_ref.norm_and_coef, o_mine.norm_and_coef)
        assert torch.equal(o_ref.flat_p, o_mine.flat_p) and torch.equal(o_ref.flat_g, o_mine.flat_g)


def test_step_can_leave_zeros_and_zero_grad_then_costs_nothing_unless_somebody_wrote():
    """FlatAdam.step(zero_grad=True): the update pass leaves zeros in the gradients, the zero_grad() that opens the next iteration (idr_train.py:283) issues no
    memset -- unless ANYTHING wrote into the buffer in between (autograd's accumulation, a torch op, the gradient sink's raw-pointer launch: each moves the
    version counter the views share).  Same parameters as the plain sequence."""
    ref, mine = _models()
    o_ref, o_mine = FlatAdam(ref.parameters(), lr=1e-3), FlatAdam(mine.parameters(), lr=1e-3)
    x = torch.randn(128, 37, device='cuda')
    memsets = {'n': 0}
    orig_zero = torch.Tensor.zero_

    def counted(self):
        if self.data_ptr() == o_mine.flat_g.data_ptr():
            memsets['n'] += 1
        return orig_zero(self)
    torch.Tensor.zero_ = counted
    try:
        for it in range(5):
            o_ref.zero_grad(); (ref(x) ** 2).mean().backward(); o_ref.step(grad_cap=0.05)
            o_mine.zero_grad(); (mine(x) ** 2).mean().backward(); o_mine.step(grad_cap=0.05, zero_grad=True)
            assert float(o_mine.flat_g.abs().max()) == 0.0                          # zeros, not the clipped gradients
            for p, q in zip(ref.parameters(), mine.parameters()):
                assert torch.equal(p, q), it
        assert memsets['n'] == 1                                                    # only the very first zero_grad() (nothing had zeroed the buffer yet)
        # a write between step() and zero_grad(): the memset is back
        (mine(x) ** 2).mean().backward()                                            # autograd accumulates into the views
        assert float(o_mine.flat_g.abs().max()) > 0
        o_mine.zero_grad()
        assert memsets['n'] == 2 and float(o_mine.flat_g.abs().max()) == 0.0
        # ... and a raw-pointer write announced the way the gradient sink does it
        o_mine.step(zero_grad=True)
        from mvsdf_amd.functional import mark_sink_written
        mark_sink_written(list(mine.parameters()))                                  # (what a sink launch does after writing through raw pointers)
        o_mine.zero_grad()
        assert memsets['n'] == 3
    finally:
        torch.Tensor.zero_ = orig_zero
