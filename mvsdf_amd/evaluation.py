"""Full-image rendering + PSNR with the reference's eval loop semantics (reference code/evaluation/eval.py:127-185, 239-246;
code/training/idr_train.py:221-230): the image is split into pixel chunks (utils.general.split_input), every chunk goes through
IDRNetwork.forward in eval mode (HIP tracer in its eval branch, analytic normals: no autograd graph, unlike the reference whose normals
need autograd.grad and therefore run outside no_grad), the `rgb_values` are merged back."""
import math

import numpy as np
import torch

from .utils import general as utils
from .utils.plots import lin2img


@torch.no_grad()
def render_image(model, model_input, total_pixels, n_pixels=10000):
    """-> rgb_values [B * total_pixels, 3] in [-1, 1] (1 where no surface was hit, idr.py:302).  eval.py:143-156."""
    was_training = model.training
    model.eval()
    res = []
    for s in utils.split_input(model_input, total_pixels, n_pixels=n_pixels):
        out = model(s)
        res.append({'rgb_values': out['rgb_values'].detach()})
    model.train(was_training)
    batch_size = model_input['uv'].shape[0]
    return utils.merge_output(res, total_pixels, batch_size)['rgb_values']


def calculate_psnr(img1, img2, mask):
    """eval.py:239-246: images in [0, 1]; the mean squared error is taken over the masked pixels only."""
    img1 = np.asarray(img1, dtype=np.float64)
    img2 = np.asarray(img2, dtype=np.float64)
    mse = np.mean((img1 - img2) ** 2) * (img2.shape[0] * img2.shape[1]) / mask.sum()
    if mse == 0:
        return float('inf')
    return 20 * math.log10(1.0 / math.sqrt(mse))


def evaluate_rendering(model, batches, img_res, n_pixels=10000):
    """eval.py:133-185 without the PNG / file output: `batches` yields (model_input, ground_truth) of ONE full image each
    (uv [1, H*W, 2], object_mask [1, H*W], rgb [1, H*W, 3]); -> (list of PSNRs, list of rendered images [H, W, 3] in [0, 1])."""
    total_pixels = img_res[0] * img_res[1]
    psnrs, images = [], []
    for model_input, ground_truth in batches:
        rgb_eval = render_image(model, model_input, total_pixels, n_pixels).reshape(1, total_pixels, 3)
        rgb_eval = (rgb_eval + 1.0) / 2.0
        rgb_eval = lin2img(rgb_eval, img_res).cpu().numpy()[0].transpose(1, 2, 0)
        rgb_gt = (ground_truth['rgb'].reshape(1, total_pixels, 3) + 1.0) / 2.0
        rgb_gt = lin2img(rgb_gt, img_res).cpu().numpy()[0].transpose(1, 2, 0)
        mask = lin2img(model_input['object_mask'].reshape(1, total_pixels, 1).float(), img_res).cpu().numpy()[0].transpose(1, 2, 0)
        psnrs.append(calculate_psnr(rgb_eval * mask, rgb_gt * mask, mask))
        images.append(rgb_eval)
    return psnrs, images
