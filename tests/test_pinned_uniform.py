"""utils.general.PinnedUniform delivers the reference's CPU-generator draws (idr.py:216-221, ray_tracing.py:287).  Whatever the delivery --
one call per tensor, two draws in one staging buffer, or the deferred form whose staging buffer the native step's first kernel reads -- the values
and the consumption of torch's CPU generator are those of `torch.empty(shape).uniform_(lo, hi)` in the same order."""
import pytest
import torch

from mvsdf_amd.utils.general import PinnedUniform


def _reference(seed):
    torch.manual_seed(seed)
    a = torch.empty(100).uniform_(0.0, 1.0)
    b = torch.empty(37, 3).uniform_(-1.2, 1.2)
    tail = torch.rand(4)
    return a, b, tail


@pytest.mark.gpu                                                  # (pinned staging needs the accelerator runtime)
def test_pair_delivered_to_the_cpu_equals_two_separate_draws():
    a0, b0, t0 = _reference(5)
    draw = PinnedUniform()
    torch.manual_seed(5)
    a, b = draw.pair((100,), 0.0, 1.0, (37, 3), -1.2, 1.2, 'cpu')
    assert torch.equal(a, a0) and torch.equal(b, b0) and torch.equal(torch.rand(4), t0)


@pytest.mark.gpu
def test_all_delivery_forms_give_the_same_values_and_rng_consumption():
    a0, b0, t0 = _reference(7)
    for form in ('single', 'pair', 'defer'):
        draw = PinnedUniform()
        for rep in range(3):                                      # both staging buffers, and a buffer's second use
            torch.manual_seed(7)
            if form == 'single':
                a = draw((100,), 0.0, 1.0, 'cuda')
                b = draw((37, 3), -1.2, 1.2, 'cuda')
            elif form == 'pair':
                a, b = draw.pair((100,), 0.0, 1.0, (37, 3), -1.2, 1.2, 'cuda')
            else:
                a, b, stage = draw.pair((100,), 0.0, 1.0, (37, 3), -1.2, 1.2, 'cuda', defer=True)
                assert stage.is_pinned() and stage.numel() == 100 + 111 and a.is_cuda and b.is_cuda
                assert a.data_ptr() + 400 == b.data_ptr()            # views of one device buffer, in the staging buffer's layout
                a, b = stage[:100].clone(), stage[100:].view(37, 3).clone()      # (the native step's first kernel performs this copy)
            assert torch.equal(a.cpu(), a0) and torch.equal(b.cpu(), b0), (form, rep)
            assert torch.equal(torch.rand(4), t0), (form, rep)
