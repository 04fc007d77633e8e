"""IDRLoss with the reference's signature and output dict (reference code/model/loss.py:16-219).

Feature consistency (loss.py:115-165) and the depth-carving target (loss.py:37-63 -> my_utils.carving_t2) run as HIP
kernels (csrc/loss_kernels.hip); the remaining terms are a handful of elementwise torch ops on device tensors."""
import importlib
import os

import torch
from torch import nn
from torch.nn import functional as F

from .. import functional as Fn
from .. import ops
from . import conf as _default_conf

conf = _default_conf
if os.environ.get('IDR_USE_ENV', '0') == '1' and os.environ.get('IDR_CONF', '') != '':
    print('override conf: ', os.environ.get('IDR_CONF'))
    conf = importlib.import_module(os.environ.get('IDR_CONF'))


class IDRLoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.l1_loss = nn.L1Loss(reduction='sum')

    def get_rgb_loss(self, rgb_values, rgb_gt, network_object_mask, object_mask):
        mask = network_object_mask & object_mask                                   # loss.py:21-28; a zero-hit batch gives 0 either way
        diff = (rgb_values - rgb_gt.reshape(-1, 3)).abs() * mask.unsqueeze(-1)
        return diff.sum() / float(object_mask.shape[0])

    def get_eikonal_loss(self, grad_theta):
        if grad_theta.shape[0] == 0:
            return torch.tensor(0.0, device=grad_theta.device)
        return ((grad_theta.norm(2, dim=1) - 1) ** 2).mean()                      # loss.py:30-35

    def get_depth_loss(self, eikonal_points_hom, eikonal_output, depths, cams, size, center, far_thresh, far_att, near_thresh, near_att,
                       smooth):
        if smooth is not None or conf.use_invalid:
            raise NotImplementedError('smooth / use_invalid variants are off in the reference conf (model/conf.py:17,25)')
        pts = eikonal_points_hom.detach()[0, :, :3, 0]
        B = depths.shape[0]
        dist_r, weight = ops.depth_carve(pts, depths.reshape(B, depths.shape[-2], depths.shape[-1]), cams.reshape(B, 2, 4, 4), size,
                                         center, conf.out_thresh_perc, far_thresh, float(far_att), near_thresh, float(near_att))
        # the reference rescales the (aliased) points to world coordinates in place (loss.py:38,42): keep that side effect
        eikonal_points_hom.detach()[:, :, :3, 0] = pts / 2 * size.view(1, 1, 1) + center.view(1, 1, 3)
        return ((eikonal_output.view(-1) + dist_r).abs() * weight).mean()         # L1(eikonal_output, -dist_r) * weights, loss.py:58-61

    def get_feat_loss_corr(self, diff_surf_pts, uncerts, feat, cam, feat_src, src_cams, size, center, network_object_mask, object_mask):
        if uncerts is not None:
            raise NotImplementedError('uncerts is always None in the reference (loss.py:197)')
        mask = network_object_mask & object_mask
        if diff_surf_pts.shape[0] == 0:
            return torch.tensor(0.0, device=diff_surf_pts.device)
        counts = mask.view(feat.size()[0], -1).sum(-1)
        view_start = torch.cat([torch.zeros(1, dtype=counts.dtype, device=counts.device), counts.cumsum(0)]).to(torch.int32)
        return Fn.feat_corr_loss(diff_surf_pts, view_start, feat, feat_src, cam, src_cams, size, center)

    def get_surf_loss(self, surf_indicator_output, network_object_mask, object_mask_true):
        n = surf_indicator_output.size()[0]
        N = (network_object_mask & object_mask_true).sum()
        gt = (torch.arange(n, device=surf_indicator_output.device) < N).to(surf_indicator_output.dtype)    # [1]*N + [0]*rest
        return F.binary_cross_entropy_with_logits(surf_indicator_output, gt, reduction='mean')

    def forward(self, model_outputs, ground_truth, train_progress, n_img):
        dev = model_outputs['rgb_values'].device
        rgb_gt = ground_truth['rgb'].to(dev)
        network_object_mask = model_outputs['network_object_mask']
        object_mask = model_outputs['object_mask']

        ground_truth['size'] = ground_truth['size'][:1]                            # side effects kept (loss.py:181-182)
        ground_truth['center'] = ground_truth['center'][:1]

        if conf.enable_rgb:
            rgb_loss = self.get_rgb_loss(model_outputs['rgb_values'], rgb_gt, network_object_mask, object_mask)
        else:
            rgb_loss = torch.zeros(1, device=dev)
        eikonal_loss = self.get_eikonal_loss(model_outputs['grad_theta'])
        depth_loss = self.get_depth_loss(model_outputs['eikonal_points_hom'], model_outputs['eikonal_output'], ground_truth['depths'],
                                         ground_truth['depth_cams'], ground_truth['size'], ground_truth['center'],
                                         far_thresh=conf.far_thresh, far_att=conf.far_att(train_progress),
                                         near_thresh=conf.near_thresh, near_att=conf.near_att(train_progress),
                                         smooth=conf.smooth(train_progress))
        if conf.phase[0] <= train_progress and conf.enable_feat:
            feat_loss = self.get_feat_loss_corr(model_outputs['diff_surf_pts'], model_outputs.get('uncerts'),
                                                *[ground_truth[a] for a in ['feat', 'cam', 'feat_src', 'src_cams', 'size', 'center']],
                                                network_object_mask, object_mask)
        else:
            feat_loss = torch.zeros(1, device=dev)
        if conf.phase[0] <= train_progress:
            surf_loss = self.get_surf_loss(model_outputs['surf_indicator_output'], network_object_mask, model_outputs['object_mask_true'])
        else:
            surf_loss = torch.zeros(1, device=dev)

        loss = rgb_loss * conf.rgb_weight(train_progress) + eikonal_loss * conf.eikonal_weight + surf_loss * conf.surf_weight + \
            feat_loss * conf.feat_weight(train_progress) + depth_loss * conf.depth_weight(train_progress)
        return {'loss': loss, 'rgb_loss': rgb_loss, 'eikonal_loss': eikonal_loss, 'depth_loss': depth_loss, 'feat_loss': feat_loss,
                'surf_loss': surf_loss}
