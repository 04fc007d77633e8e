"""IDRLoss with the reference's signature and output dict (reference code/model/loss.py:16-219).

Feature consistency (loss.py:115-165) and the depth-carving target (loss.py:37-63 -> my_utils.carving_t2) run as HIP
kernels (csrc/loss_kernels.hip); the remaining terms are a handful of elementwise torch ops on device tensors."""
import importlib
import os

import torch
import torch.distributed as dist
from torch import nn
from torch.nn import functional as F

from .. import functional as Fn
from .. import native_step as NS
from .. import ops
from . import conf as _default_conf

conf = _default_conf
if os.environ.get('IDR_USE_ENV', '0') == '1' and os.environ.get('IDR_CONF', '') != '':
    print('override conf: ', os.environ.get('IDR_CONF'))
    conf = importlib.import_module(os.environ.get('IDR_CONF'))


_COUNT_STREAMS = {}                                              # per device: the stream of the data-parallel count all-reduce (IDRLoss.forward)


class IDRLoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.l1_loss = nn.L1Loss(reduction='sum')
        # with torch.distributed initialised: normalise the count-based means by the counts summed over the ranks (see forward)
        self.exact_data_parallel = True
        # forward / backward as ONE C call each (mvsdf_loss_forward / mvsdf_loss_backward) instead of four calls + two autograd nodes
        self.native = os.environ.get('MVSDF_NATIVE_STEP', '1') != '0'
        self.collective_events = None                            # set to a list: (start, end) CUDA events around the 3-count all-reduce are appended (bench.py)

    @staticmethod
    def _count_stream(dev):
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        if key not in _COUNT_STREAMS:
            _COUNT_STREAMS[key] = torch.cuda.Stream(device=dev)
        return _COUNT_STREAMS[key]

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
        pts = eikonal_points_hom.detach()[0, :, :3, 0]
        B = depths.shape[0]
        dist_r, weight = ops.depth_carve(pts, depths.reshape(B, depths.shape[-2], depths.shape[-1]), cams.reshape(B, 2, 4, 4), size,
                                         center, conf.out_thresh_perc, far_thresh, float(far_att), near_thresh, float(near_att), use_invalid=bool(conf.use_invalid))
        # the reference rescales the (aliased) points to world coordinates in place (loss.py:38,42): keep that side effect
        eikonal_points_hom.detach()[:, :, :3, 0] = pts / 2 * size.view(1, 1, 1) + center.view(1, 1, 3)
        if smooth is not None:                                                     # loss.py:57-58
            el = F.smooth_l1_loss(eikonal_output.view(-1) / smooth, -dist_r / smooth, reduction='none') * smooth
        else:
            el = (eikonal_output.view(-1) + dist_r).abs()                          # L1(eikonal_output, -dist_r), loss.py:60
        return (el * weight).mean()                                                # * far / near weights * in_range, loss.py:61

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

    def _carve(self, eikonal_points_hom, depths, cams, size, center, train_progress):
        hom = eikonal_points_hom.detach()
        B = depths.shape[0]
        args = (depths.reshape(B, depths.shape[-2], depths.shape[-1]), cams.reshape(B, 2, 4, 4), size, center, conf.out_thresh_perc,
                conf.far_thresh, float(conf.far_att(train_progress)), conf.near_thresh, float(conf.near_att(train_progress)))
        if hom.is_contiguous() and hom.dtype == torch.float32 and hom.is_cuda:
            # the kernel reads the [n, 4] rows in place and writes the world-space points back: the side effect of loss.py:38,42
            return ops.depth_carve(hom.view(-1, 4), *args, world_inplace=True, use_invalid=bool(conf.use_invalid))
        pts = hom[0, :, :3, 0]
        dist_r, weight = ops.depth_carve(pts, *args, use_invalid=bool(conf.use_invalid))
        hom[:, :, :3, 0] = pts / 2 * size.view(1, 1, 1) + center.view(1, 1, 3)
        return dist_r, weight

    def forward(self, model_outputs, ground_truth, train_progress, n_img):
        """Same outputs as the reference (loss.py:176-219).  The feature term and the depth-carving target are HIP kernels; the
        remaining elementwise terms and the weighted sum are ONE more launch (csrc/loss_kernels.hip::k_loss_terms)."""
        # a DEFERRED step (IDRNetwork.deferred_step): the forward is enqueued, the host does not know the hit counts and nothing here asks for them
        rec = model_outputs.pending_rec() if (self.native and hasattr(model_outputs, 'pending_rec')) else None
        if rec is not None:
            out = self._forward_deferred(rec, model_outputs, ground_truth, train_progress)
            if out is not None:
                return out
        dev = model_outputs['rgb_values'].device
        rgb_gt = ground_truth['rgb'].to(dev)
        network_object_mask = model_outputs['network_object_mask']
        object_mask = model_outputs['object_mask']

        ground_truth['size'] = ground_truth['size'][:1]                            # side effects kept (loss.py:181-182)
        ground_truth['center'] = ground_truth['center'][:1]
        smooth = conf.smooth(train_progress)                                       # loss.py:57-58: SmoothL1 depth term (None in the shipped conf)
        # conf.enable_rgb = False (loss.py:184-187): the rgb term is zeros(1) -- no contribution to the total, no gradient: weight 0, and the scalar is replaced below
        weights = (conf.rgb_weight(train_progress) if conf.enable_rgb else 0.0, conf.eikonal_weight, conf.surf_weight, conf.feat_weight(train_progress),
                   conf.depth_weight(train_progress), float(smooth) if smooth is not None else 0.0)
        phase1 = conf.phase[0] <= train_progress
        feat_on = bool(phase1 and conf.enable_feat)
        if feat_on and model_outputs.get('uncerts') is not None:
            raise NotImplementedError('uncerts is always None in the reference (loss.py:197)')
        # Data parallel (one process per GPU, rays sharded by view, gradients averaged over ranks): the three count-normalised means
        # (eikonal over grad_theta rows, depth over eikonal_output entries, surface BCE over its logits; loss.py:34,61,173) divide by
        # the GLOBAL counts so that the rank-averaged gradient equals the single-process one -- one extra all-reduce of 3 numbers.
        inv_counts = None
        if self.exact_data_parallel and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            cnt_host = [float(model_outputs['grad_theta'].shape[0]), float(model_outputs['eikonal_output'].numel()),
                        float(model_outputs['surf_indicator_output'].numel())]
            world = float(dist.get_world_size())
            if dev.type == 'cuda':
                # The counts are host numbers (shapes) while the GPU still has the end of the forward queued (the host ran ahead after the
                # step's one wait): the collective goes on a side stream, i.e. beside that work instead of behind it, and the loss kernels
                # wait for it by an event.  (The collective library orders itself against the CURRENT stream only.)
                main = torch.cuda.current_stream(dev)
                side = self._count_stream(dev)
                with torch.cuda.stream(side):
                    cnt = torch.tensor(cnt_host, device=dev)
                    ev = None
                    if self.collective_events is not None:
                        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                        ev[0].record(side)
                    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
                    if ev is not None:
                        ev[1].record(side)
                        self.collective_events.append(ev)
                    inv_counts = world / cnt.clamp(min=1.0)
                    done = torch.cuda.Event()
                    done.record(side)
                main.wait_event(done)
                inv_counts.record_stream(main)                     # allocated on the side stream, read by the loss kernels on `main`
            else:
                cnt = torch.tensor(cnt_host, device=dev)
                dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
                inv_counts = world / cnt.clamp(min=1.0)
        if self.native:
            out = self._forward_native(model_outputs, ground_truth, rgb_gt, train_progress, weights, bool(phase1), feat_on, inv_counts)
            if out is not None:
                return self._terms(out, dev)
        # masks -> hit mask, per-view row ranges of diff_surf_pts, number of BCE positives: one launch (csrc/loss_kernels.hip::k_loss_prep)
        n_views = ground_truth['feat'].size()[0] if 'feat' in ground_truth else 1
        hit_mask, view_start, n_pos = ops.loss_prep(network_object_mask, object_mask, model_outputs['object_mask_true'], n_views)
        feat_pp = None
        pts = model_outputs['diff_surf_pts']
        if feat_on and pts.shape[0] > 0:
            feat_pp = Fn.feat_corr_terms(pts, view_start, ground_truth['feat'], ground_truth['feat_src'], ground_truth['cam'],
                                         ground_truth['src_cams'], ground_truth['size'], ground_truth['center'])
        dist_r, dweight = self._carve(model_outputs['eikonal_points_hom'], ground_truth['depths'], ground_truth['depth_cams'],
                                      ground_truth['size'], ground_truth['center'], train_progress)
        out = Fn.loss_terms(model_outputs['rgb_values'], model_outputs['grad_theta'], model_outputs['eikonal_output'],
                            model_outputs['surf_indicator_output'], feat_pp, rgb_gt, hit_mask, dist_r, dweight,
                            n_pos, weights, bool(phase1), feat_on, inv_counts)
        return self._terms(out, dev)

    @staticmethod
    def _terms(out, dev):
        """The reference's dict of six scalars (loss.py:212-219); with conf.enable_rgb off rgb_loss is the constant zeros(1) of loss.py:187."""
        return {'loss': out[0], 'rgb_loss': out[1] if conf.enable_rgb else torch.zeros(1, device=dev), 'eikonal_loss': out[2], 'depth_loss': out[3],
                'feat_loss': out[4], 'surf_loss': out[5]}

    def _forward_deferred(self, rec, mo, gt, train_progress):
        """IDRLoss.forward on a pending step: ONE C call (mvsdf_loss_forward with MvsdfLossArgs.counts_dev) whose kernels take the hit counts from the
        forward block on the device, behind one autograd node over the network's parameters whose backward is mvsdf_loss_backward + mvsdf_step_backward(N < 0).
        -> the reference's dict of six scalars, or None when an input is not a plain contiguous fp32 / mask tensor on the step's device (the caller then reads
        the outputs, which waits for the counts, and takes the classic route: same numbers)."""
        st = rec.step
        dev = st.device
        if dict.get(mo, 'uncerts') is not None:
            return None
        f32 = lambda t: t is not None and t.is_cuda and t.device == dev and t.dtype == torch.float32 and t.is_contiguous()
        rgb_gt = gt['rgb']
        if not rgb_gt.is_cuda:
            rgb_gt = rgb_gt.to(dev)
        depths, dcams = gt['depths'], gt['depth_cams']
        size1, center1 = gt['size'], gt['center']
        if size1.shape[0] != 1:
            size1 = size1[:1]
        if center1.shape[0] != 1:
            center1 = center1[:1]
        if not all(f32(t) for t in (rgb_gt, depths, dcams, size1, center1)):
            return None
        masks = []
        for k in ('network_object_mask', 'object_mask', 'object_mask_true'):
            m = mo.raw(k)
            if m.dim() != 1:
                m = m.reshape(-1)
            if not (m.is_cuda and m.dtype in (torch.bool, torch.uint8) and m.is_contiguous()):
                return None
            masks.append(m)
        d, L, f = st.desc, st.layout, rec.fwd
        R = st.R
        if masks[0].numel() != R or rgb_gt.numel() != 3 * R:
            return None
        smooth = conf.smooth(train_progress)
        weights = (conf.rgb_weight(train_progress) if conf.enable_rgb else 0.0, conf.eikonal_weight, conf.surf_weight, conf.feat_weight(train_progress),
                   conf.depth_weight(train_progress), float(smooth) if smooth is not None else 0.0)
        phase1 = conf.phase[0] <= train_progress
        feat_on = bool(phase1 and conf.enable_feat)
        a = NS.LossArgs()
        nd_max, ne_max = rec.group_rows(R)                        # upper bounds (N = R): they size the block; the kernels bound their rows by the device counts
        a.R, a.N, a.n_grad, a.n_depth, a.n_surf = R, R, ne_max, nd_max, R + d.n_eik
        base = f.data_ptr()
        a.counts_dev = base + st.counts_off
        a.n_eik, a.n_ds, a.d_mask, a.e_mask = d.n_eik, d.n_ds, rec.d_mask, rec.e_mask
        a.net_mask, a.obj_mask, a.true_mask = masks[0].data_ptr(), masks[1].data_ptr(), masks[2].data_ptr()
        a.rgb, a.rgb_gt = base + L.rgb_values, rgb_gt.data_ptr()
        a.grad_theta, a.eik_out, a.surf, a.diff_pts = base + L.grad_theta, base + L.eik_out, base + L.surf, base + L.diff_pts
        a.points_hom = base + L.points_hom                        # rescaled to world coordinates in place: the side effect of loss.py:38,42
        a.feat_on, a.surf_on = int(feat_on), int(bool(phase1))
        a.B = 1
        keep = [rgb_gt, depths, dcams, size1, center1] + masks
        if feat_on:
            feat, fsrc = gt['feat'], gt['feat_src']
            if not (feat.is_cuda and fsrc.is_cuda and feat.dtype == torch.float32 and fsrc.dtype == torch.float32 and feat.shape[1] <= 32
                    and f32(gt['cam']) and f32(gt['src_cams'])):
                return None
            a.B, a.C, a.H, a.W = feat.shape
            a.V = fsrc.shape[1]
            a.feat, a.feat_src, a.cam, a.src_cams = feat.data_ptr(), fsrc.data_ptr(), gt['cam'].data_ptr(), gt['src_cams'].data_ptr()
            for i, v in enumerate(feat.stride()):
                a.feat_strides[i] = v
            for i, v in enumerate(fsrc.stride()):
                a.src_strides[i] = v
            keep += [feat, fsrc, gt['cam'], gt['src_cams']]
        elif 'feat' in gt:
            a.B = gt['feat'].size()[0]
        if R % a.B:
            return None
        dB = depths.shape[0]
        if depths.numel() != dB * depths.shape[-2] * depths.shape[-1] or dcams.numel() != dB * 32:
            return None
        gt['size'], gt['center'] = size1, center1                 # side effects kept (loss.py:181-182)
        a.size, a.center = size1.data_ptr(), center1.data_ptr()
        a.depths, a.dB, a.dh, a.dw, a.depth_cams = depths.data_ptr(), dB, depths.shape[-2], depths.shape[-1], dcams.data_ptr()
        a.out_thresh_perc, a.far_thresh, a.near_thresh = conf.out_thresh_perc, conf.far_thresh, conf.near_thresh
        a.use_invalid = 1 if conf.use_invalid else 0
        a.far_att, a.near_att = float(conf.far_att(train_progress)), float(conf.near_att(train_progress))
        a.w_rgb, a.w_eik, a.w_surf, a.w_feat, a.w_depth, a.smooth = [float(w) for w in weights]
        if self.exact_data_parallel and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # the three count-normalised means divide by the GLOBAL counts (see forward): here they are computed on the device from {N, n_true}
            # (the ray partition left the three row counts as floats behind the int64 counts of this forward: no host round trip, no arithmetic here)
            cnt = f.f(st.counts_off + 32, (3,)).clone()
            ev = None
            if self.collective_events is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
            if ev is not None:
                ev[1].record()
                self.collective_events.append(ev)
            inv_counts = (float(dist.get_world_size()) / cnt.clamp(min=1.0)).contiguous()
            keep.append(inv_counts)
            a.inv_counts = inv_counts.data_ptr()
        return self._terms(NS.deferred_loss_forward(rec, a, keep), dev)

    def _forward_native(self, mo, gt, rgb_gt, train_progress, weights, surf_on, feat_on, inv_counts):
        """The whole forward as ONE C call (mvsdf_loss_forward: mask bookkeeping, feature consistency, depth carving with the in-place
        world rescale of eikonal_points_hom, every term and its unit gradient) behind one autograd node whose backward is one launch.
        -> the six scalars, or None when an input is not a plain contiguous fp32 / mask tensor on the GPU (the generic route takes over)."""
        rgb, gth, eo, sf, pts, hom = (mo['rgb_values'], mo['grad_theta'], mo['eikonal_output'], mo['surf_indicator_output'], mo['diff_surf_pts'],
                                      mo['eikonal_points_hom'])
        f32 = lambda t: t is not None and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        depths, dcams = gt['depths'], gt['depth_cams']
        if not (all(f32(t) for t in (rgb, gth, eo, sf, pts, hom, rgb_gt, depths, dcams, gt['size'], gt['center'])) and hom.dim() == 4 and hom.shape[2:] == (4, 1)):
            return None
        masks = []
        for m in (mo['network_object_mask'], mo['object_mask'], mo['object_mask_true']):
            m = m.reshape(-1)
            if not (m.is_cuda and m.dtype in (torch.bool, torch.uint8) and m.is_contiguous()):
                return None
            masks.append(m)
        R = masks[0].numel()
        a = NS.LossArgs()
        keep = [rgb_gt, depths, dcams, gt['size'], gt['center'], inv_counts] + masks
        a.R, a.N, a.n_grad, a.n_depth, a.n_surf = R, pts.shape[0], gth.shape[0], eo.numel(), sf.numel()
        a.net_mask, a.obj_mask, a.true_mask = masks[0].data_ptr(), masks[1].data_ptr(), masks[2].data_ptr()
        a.rgb, a.rgb_gt = rgb.data_ptr(), rgb_gt.data_ptr()
        a.grad_theta, a.eik_out, a.surf, a.diff_pts = gth.data_ptr(), eo.data_ptr(), sf.data_ptr(), pts.data_ptr()
        a.points_hom = hom.data_ptr()                             # rescaled to world coordinates in place: the side effect of loss.py:38,42
        if hom.numel() != 4 * eo.numel() or rgb.shape[0] != R or rgb_gt.numel() != 3 * R:
            return None
        a.feat_on, a.surf_on = int(feat_on), int(surf_on)
        a.B = 1
        if feat_on:
            feat, fsrc = gt['feat'], gt['feat_src']
            if not (feat.is_cuda and fsrc.is_cuda and feat.dtype == torch.float32 and fsrc.dtype == torch.float32 and feat.shape[1] <= 32
                    and f32(gt['cam']) and f32(gt['src_cams'])):
                return None
            a.B, a.C, a.H, a.W = feat.shape
            a.V = fsrc.shape[1]
            a.feat, a.feat_src, a.cam, a.src_cams = feat.data_ptr(), fsrc.data_ptr(), gt['cam'].data_ptr(), gt['src_cams'].data_ptr()
            for i, v in enumerate(feat.stride()):
                a.feat_strides[i] = v
            for i, v in enumerate(fsrc.stride()):
                a.src_strides[i] = v
            keep += [feat, fsrc, gt['cam'], gt['src_cams']]
        elif 'feat' in gt:
            a.B = gt['feat'].size()[0]                            # (the per-view row ranges are computed either way)
        if R % a.B:
            return None
        a.size, a.center = gt['size'].data_ptr(), gt['center'].data_ptr()
        dB = depths.shape[0]
        a.depths, a.dB, a.dh, a.dw, a.depth_cams = depths.data_ptr(), dB, depths.shape[-2], depths.shape[-1], dcams.data_ptr()
        if depths.numel() != dB * a.dh * a.dw or dcams.numel() != dB * 32:
            return None
        a.out_thresh_perc, a.far_thresh, a.near_thresh = conf.out_thresh_perc, conf.far_thresh, conf.near_thresh
        a.use_invalid = 1 if conf.use_invalid else 0
        a.far_att, a.near_att = float(conf.far_att(train_progress)), float(conf.near_att(train_progress))
        a.w_rgb, a.w_eik, a.w_surf, a.w_feat, a.w_depth, a.smooth = [float(w) for w in weights]
        if inv_counts is not None:
            inv_counts = inv_counts.float().contiguous()
            keep.append(inv_counts)
            a.inv_counts = inv_counts.data_ptr()
        return NS.loss_forward(a, keep, rgb, gth, eo, sf, pts)
