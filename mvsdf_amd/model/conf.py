"""Training-phase schedule and loss weights as functions of train_progress in [0, 1].

Module-level names and values are those of the reference's config module (reference code/model/conf.py:1-33) so that
`IDR_USE_ENV=1 IDR_CONF=<module>` overrides (idr.py:15-17, loss.py:12-14) are interchangeable; here they are generated from
one table: value in phase 0 (tp < 1/6), phase 1 (tp < 1/2), phase 2.
"""
phase = (1 / 6, 1 / 2)
feat_img_scale = 2


def _schedule(per_phase):
    p0, p1, p2 = per_phase
    return lambda tp: p0 if tp < phase[0] else (p1 if tp < phase[1] else p2)


_SCHEDULES = {
    # which point sets feed the depth (d_*) and eikonal (eik_*) terms
    'd_use_rt_surf': (True, True, True), 'd_use_eik': (True, True, True),
    'd_use_dsurf_on': (True, False, False), 'd_use_dsurf_jitter': (True, False, False),
    'eik_use_rt_surf': (True, True, True), 'eik_use_eik': (True, True, True),
    'eik_use_dsurf_on': (True, False, False), 'eik_use_dsurf_jitter': (True, False, False),
    # depth-loss attenuation and loss weights
    'far_att': (1, 1, 1), 'near_att': (1, 0.1, 0.01),
    'rgb_weight': (0.5, 0.5, 0.5), 'feat_weight': (0, 0.1, 0.01), 'depth_weight': (1, 1, 1),
    'smooth': (None, None, None),
}
globals().update({name: _schedule(vals) for name, vals in _SCHEDULES.items()})

# constants
eikonal_weight, surf_weight = 0.1, 0.01
far_thresh, near_thresh = 0.25, 0.1
out_thresh_perc = 1 / 8
use_mask = use_invalid = disable_rgb_grad = False
enable_feat = enable_rgb = enable_grad_cap = True


def grad_cap(tp):
    return 2 if tp < phase[1] else 0.5
