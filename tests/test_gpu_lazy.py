"""IDRNetwork.lazy_unused_outputs (opt-in): the min-sdf points of the rays without a hit (ray_tracing.py:280-308) feed only `points` /
`sdf_output`, which the training loop does not read (loss.py:176-219).  Deferred evaluation must be unobservable: every output, the loss,
the gradients and the random stream are identical to the eager forward, and a read after the next forward fails loudly."""
import numpy as np
import pytest
import torch

from helpers import t
from mvsdf_amd.model.implicit_differentiable_renderer import IDRNetwork, LazyOutputs
from mvsdf_amd.model.loss import IDRLoss
from mvsdf_amd.utils import synth
from mvsdf_amd.utils.config import ConfigDict

pytestmark = pytest.mark.gpu
TP = 0.3


def _model(W=64):
    m = IDRNetwork(ConfigDict(synth.model_conf(W)))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(W, 0).items()})
    return m.cuda().train()


def _run(lazy, read_first):
    m = _model()
    m.lazy_unused_outputs = lazy
    B, P = 2, 300
    inp, gt = synth.make_batch(B, P, 2, seed=3, feat_hw=(60, 80), focal_scale=1.4)       # wide field of view: many rays miss the object
    torch.manual_seed(7)
    out = m({k: t(v) for k, v in inp.items()}, TP)
    snap = {}
    if read_first:                                         # read the deferred keys BEFORE the loss / backward ...
        snap = {k: out[k].clone() for k in ('points', 'sdf_output')}
    lo = IDRLoss()(out, {k: t(v) for k, v in gt.items()}, TP, B)
    lo['loss'].backward()
    if not read_first:                                     # ... or after them
        snap = {k: out[k].clone() for k in ('points', 'sdf_output')}
    g = torch.cat([p.grad.flatten() for p in m.parameters()])
    rest = {k: v.detach().clone() for k, v in out.items() if torch.is_tensor(v) and k not in snap}
    return m, out, snap, rest, lo['loss'].detach().clone(), g, torch.rand(4)


@pytest.mark.parametrize('read_first', [True, False])
def test_lazy_outputs_equal_eager(read_first):
    _, out_e, snap_e, rest_e, loss_e, g_e, rng_e = _run(False, read_first)
    m, out_l, snap_l, rest_l, loss_l, g_l, rng_l = _run(True, read_first)
    from mvsdf_amd.model.implicit_differentiable_renderer import PendingOutputs
    assert isinstance(out_e, dict) and (type(out_e) is dict or isinstance(out_e, PendingOutputs)) and type(out_l) is LazyOutputs   # (eager = every tracer row evaluated in the forward: a plain dict, or the deferred step's pending one)
    miss = ~out_e['network_object_mask']
    assert int(miss.sum()) > 50 and int((~miss).sum()) > 50
    assert m.last_stats['counters'][6] > 0                                  # there was a min-sdf work list to defer
    for k in snap_e:
        assert torch.equal(snap_e[k], snap_l[k]), k
    assert rest_e.keys() == rest_l.keys()
    for k in rest_e:
        assert torch.equal(rest_e[k], rest_l[k]), k
    assert torch.equal(loss_e, loss_l) and torch.equal(g_e, g_l)
    assert torch.equal(rng_e, rng_l)                                        # same draws from the CPU generator in both modes
    # the min-sdf points differ from what the sphere tracer left there: the deferred launch really ran
    with torch.no_grad():
        y = m.implicit_network(snap_l['points'][miss])[:, :1]
    assert torch.equal(snap_l['sdf_output'][miss], y)


def test_lazy_access_paths_and_expiry():
    m = _model()
    m.lazy_unused_outputs = True
    inp, _ = synth.make_batch(1, 200, 0, seed=3, with_features=False, focal_scale=1.4)
    dev_inp = {k: t(v) for k, v in inp.items()}
    torch.manual_seed(1)
    a = m(dev_inp, TP)
    assert a._pending is not None
    _ = a['rgb_values'], a.get('grad_theta'), list(a), 'points' in a, len(a)  # none of these touches the deferred keys
    assert a._pending is not None
    d = dict(a)                                                              # a plain copy reads every key: materialised
    assert a._pending is None and set(d) == set(a.keys())
    b = m(dev_inp, TP)
    c = m(dev_inp, TP)                                                       # b's deferred rows can no longer be evaluated at b's weights
    with pytest.raises(RuntimeError, match='lazy_unused_outputs'):
        b['points']
    with pytest.raises(RuntimeError, match='lazy_unused_outputs'):
        b.items()
    assert torch.isfinite(c['points']).all() and torch.isfinite(b['rgb_values']).all()
    m.eval()
    with torch.no_grad():
        e = m(dev_inp)
    assert type(e) is dict                                                   # eval never defers (plots read `points`)
