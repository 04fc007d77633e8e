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


# ---- PendingOutputs (the deferred training step): the N-shaped keys and `rgb_values` are placeholders until somebody reads one of them
def make_pending():
    from mvsdf_amd.model.implicit_differentiable_renderer import PendingOutputs
    calls = []
    eager_part = {'points': np.ones(3), 'diff_surf_pts': None, 'rgb_values': None, 'sdf_output': np.ones(3), 'network_object_mask': np.ones(3, bool),
                  'object_mask': np.ones(3, bool), 'object_mask_true': np.ones(3, bool), 'grad_theta': None, 'eikonal_points_hom': None, 'eikonal_output': None,
                  'surf_indicator_output': None}
    class Rec:                                                                     # (weak-referenceable stand-in of native_step.StepRecord)
        pass
    rec = Rec()

    def fill(target):                                                              # the product's closure: fills the dict it is handed, holds no reference to it
        calls.append(1)
        dict.update(target, {k: np.full(2, 5.0) for k in PendingOutputs._LAZY})
    out = PendingOutputs(eager_part, fill, rec)
    return out, calls, rec


def test_pending_outputs_die_by_reference_count():
    """The output dict of a deferred step, its record and through it the forward block (3.6 GB in the shipped workload) must not wait for the cyclic
    collector: a closure over the dict itself did that until late round 6 (one block per step stayed allocated until the collector ran)."""
    import gc
    import weakref
    gc.collect()
    gc.disable()
    try:
        for resolve in (False, True):
            out, _, rec = make_pending()
            w_out, w_rec = weakref.ref(out), weakref.ref(rec)
            if resolve:
                out['rgb_values']
            del out, rec
            assert w_out() is None and w_rec() is None, 'resolved' if resolve else 'pending'
    finally:
        gc.enable()


def test_pending_outputs_resolve_on_the_first_read_of_an_n_shaped_key():
    out, calls, rec = make_pending()
    assert out.pending_rec() is rec
    # what IDRLoss's deferred route and the tests read: never resolves
    assert out.raw('network_object_mask').all() and out['points'][0] == 1.0 and out.get('uncerts') is None and 'rgb_values' in out and len(out) == 11
    assert list(out)[:3] == ['points', 'diff_surf_pts', 'rgb_values']             # the reference's key order (idr.py:306-322)
    assert calls == [] and out.pending_rec() is rec
    assert float(out['rgb_values'][0]) == 5.0                                      # the first read of a lazy key: one resolution for all of them
    assert calls == [1] and out.pending_rec() is None
    assert float(out['diff_surf_pts'][0]) == 5.0 and float(out.get('eikonal_output')[0]) == 5.0
    assert calls == [1]


@pytest.mark.parametrize('how', ['items', 'values', 'dict', 'copy', 'unpack'])
def test_pending_outputs_bulk_access_resolves(how):
    out, calls, _ = make_pending()
    got = {'items': lambda: dict(out.items()), 'values': lambda: dict(zip(out.keys(), out.values())), 'dict': lambda: dict(out), 'copy': lambda: out.copy(),
           'unpack': lambda: {**out}}[how]()
    assert calls == [1] and all(v is not None for v in got.values()) and float(got['grad_theta'][0]) == 5.0


def test_step_stats_look_n_up_only_when_asked_and_group_rows():
    """IDRNetwork.last_stats of a deferred step and StepRecord.group_rows (rows of eikonal_output / grad_theta for N hit rows: point groups
    [hit | eikonal | on-surface | jittered], idr.py:253-286): host logic, no GPU."""
    from types import SimpleNamespace
    from mvsdf_amd.model.implicit_differentiable_renderer import _StepStats
    from mvsdf_amd.native_step import StepRecord
    calls = []

    class Rec:
        def resolve(self):
            calls.append(1)
            return 123, 100
    st = _StepStats(Rec(), R=2048, E=1024)
    assert st['R'] == 2048 and st.get('E') == 1024 and st.get('nope', 7) == 7 and calls == []
    assert st['N'] == 123 and st.get('N') == 123 and calls == [1]                  # looked up once
    rec = StepRecord()
    rec.step = SimpleNamespace(desc=SimpleNamespace(n_eik=1024, n_ds=300))
    rec.d_mask, rec.e_mask = 0b0011, 0b1111
    assert rec.group_rows(500) == (500 + 1024, 500 + 1024 + 600)
    rec.d_mask, rec.e_mask = 0b0010, 0b0001
    assert rec.group_rows(2048) == (1024, 2048) and rec.group_rows(0) == (1024, 0)
