"""Training-phase schedule and loss weights as functions of train_progress in [0, 1].

Same names and values as the reference's module-level config (reference code/model/conf.py:1-33), so that
`IDR_USE_ENV=1 IDR_CONF=<module>` overrides (idr.py:15-17, loss.py:12-14) keep working with either module.
"""
feat_img_scale = 2

phase = (1 / 6, 1 / 2)          # phase 0: depth-surface sampling; phase 1: feature + surface losses; phase 2: annealed


def _by_phase(a, b, c):
    return lambda tp: a if tp < phase[0] else (b if tp < phase[1] else c)


d_use_rt_surf = _by_phase(True, True, True)
d_use_eik = _by_phase(True, True, True)
d_use_dsurf_on = _by_phase(True, False, False)
d_use_dsurf_jitter = _by_phase(True, False, False)
eik_use_rt_surf = _by_phase(True, True, True)
eik_use_eik = _by_phase(True, True, True)
eik_use_dsurf_on = _by_phase(True, False, False)
eik_use_dsurf_jitter = _by_phase(True, False, False)

disable_rgb_grad = False

use_invalid = False
use_mask = False
out_thresh_perc = 1 / 8
enable_feat = True
enable_rgb = True
far_thresh = 0.25
far_att = _by_phase(1, 1, 1)
near_thresh = 0.1
near_att = _by_phase(1, 0.1, 0.01)
smooth = lambda tp: None
rgb_weight = _by_phase(0.5, 0.5, 0.5)
surf_weight = 0.01
feat_weight = _by_phase(0, 0.1, 0.01)
depth_weight = _by_phase(1, 1, 1)
eikonal_weight = 0.1

enable_grad_cap = True
grad_cap = lambda tp: 2 if tp < phase[1] else 0.5
