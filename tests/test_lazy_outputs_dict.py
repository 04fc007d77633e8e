"""LazyOutputs (IDRNetwork.lazy_unused_outputs): every way of observing or moving `points` / `sdf_output` sees the eager values.
CPU-only: the deferred evaluation is stood in for by a callable that rewrites the two entries in place, like the real one does."""
import copy
import pickle

import numpy as np
import pytest

from mvsdf_amd.model.implicit_differentiable_renderer import LazyOutputs


def make():
    pts, sdf = np.zeros(3), np.zeros(3)
    calls = []

    def materialize():
        calls.append(1)
        pts[:] = 1.0                                   # in place, like the min-sdf launch + sdf_forward of the real closure
        sdf[:] = 2.0
    return LazyOutputs({'points': pts, 'sdf_output': sdf, 'rgb_values': np.full(3, 7.0)}, materialize), calls


def eager(d):
    return float(d['points'][0]) == 1.0 and float(d['sdf_output'][0]) == 2.0


def test_untouched_keys_do_not_materialize():
    out, calls = make()
    assert out['rgb_values'][0] == 7.0 and 'points' in out and len(out) == 3 and list(out.keys())[0] == 'points'
    assert out.get('rgb_values') is not None and out.setdefault('rgb_values', None) is not None
    assert calls == []


@pytest.mark.parametrize('how', ['getitem', 'get', 'items', 'values', 'iter_dict', 'copy', 'pop', 'setdefault', 'popitem', 'update', 'eq', 'or', 'ror',
                                 'ior', 'copy.copy', 'deepcopy', 'pickle', 'unpack'])
def test_every_access_path_sees_eager_values(how):
    out, calls = make()
    pts = dict.__getitem__(out, 'points')              # the raw entry (what a bypassing path would have returned)
    if how == 'getitem':
        got = out['points']
    elif how == 'get':
        got = out.get('points')
    elif how == 'items':
        got = dict(out.items())['points']
    elif how == 'values':
        got = list(out.values())[0]
    elif how == 'iter_dict':
        got = dict(out)['points']
    elif how == 'copy':
        got = out.copy()['points']
    elif how == 'pop':
        got = out.pop('points')
    elif how == 'setdefault':
        got = out.setdefault('points', None)
    elif how == 'popitem':
        out.popitem()
        got = pts
    elif how == 'update':
        out.update({'extra': 1})
        got = pts
    elif how == 'eq':
        assert not (out == {'points': None})
        got = pts
    elif how == 'or':
        got = (out | {'extra': 1})['points']
    elif how == 'ror':
        got = ({'extra': 1} | out)['points']
    elif how == 'ior':
        out |= {'extra': 1}
        got = pts
    elif how == 'copy.copy':
        c = copy.copy(out)
        assert type(c) is dict
        got = c['points']
    elif how == 'deepcopy':
        got = copy.deepcopy(out)['points']
    elif how == 'pickle':
        c = pickle.loads(pickle.dumps(out))
        assert type(c) is dict
        got = c['points']
    elif how == 'unpack':
        got = {**out}['points']
    assert calls == [1] and float(got[0]) == 1.0 and float(pts[0]) == 1.0
    out._materialize()
    assert calls == [1]                                # once only


def test_expired_outputs_raise():
    out, _ = make()
    out._expire()
    for f in (lambda: out['points'], lambda: out.copy(), lambda: out == {}, lambda: pickle.dumps(out), lambda: out.update(a=1)):
        with pytest.raises(RuntimeError, match='lazy_unused_outputs'):
            f()
    assert out['rgb_values'][0] == 7.0
