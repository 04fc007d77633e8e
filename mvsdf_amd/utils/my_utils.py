"""Projection helpers with the reference's names (reference code/utils/my_utils.py:71-110, 152-165).

The hot path fuses idx_world2cam / idx_cam2img / normalize_for_grid_sample / get_in_range into csrc/loss_kernels.hip;
the tensor versions below serve the phase-0 depth-surface sampling (idr.py:226-247, a "next" row of SURVEY section 8f)
and external callers."""
import torch


def get_pixel_grids(height, width, device='cuda'):
    x = (torch.arange(width, dtype=torch.float32, device=device) + 0.5).repeat(height, 1)
    y = (torch.arange(height, dtype=torch.float32, device=device) + 0.5).repeat(width, 1).t()
    return torch.stack([x, y, torch.ones_like(x)], dim=-1).unsqueeze(-1)                     # hw31


def idx_img2cam(idx_img_homo, depth, cam):
    """nhw31, n1hw -> nhw41"""
    idx_cam = cam[:, 1:2, :3, :3].unsqueeze(1).inverse() @ idx_img_homo
    idx_cam = idx_cam / (idx_cam[..., -1:, :] + 1e-9) * depth.permute(0, 2, 3, 1).unsqueeze(4)
    return torch.cat([idx_cam, torch.ones_like(idx_cam[..., -1:, :])], dim=-2)


def idx_cam2world(idx_cam_homo, cam):
    idx_world_homo = cam[:, 0:1, ...].unsqueeze(1).inverse() @ idx_cam_homo
    return idx_world_homo / (idx_world_homo[..., -1:, :] + 1e-9)


def idx_world2cam(idx_world_homo, cam):
    idx_cam_homo = cam[:, 0:1, ...].unsqueeze(1) @ idx_world_homo
    return idx_cam_homo / (idx_cam_homo[..., -1:, :] + 1e-9)


def idx_cam2img(idx_cam_homo, cam):
    idx_cam = idx_cam_homo[..., :3, :] / (idx_cam_homo[..., 3:4, :] + 1e-9)
    idx_img_homo = cam[:, 1:2, :3, :3].unsqueeze(1) @ idx_cam
    return idx_img_homo / (idx_img_homo[..., -1:, :] + 1e-9)


def normalize_for_grid_sample(input_, grid):
    size = torch.tensor(input_.size())[2:].flip(0).to(grid.dtype).to(grid.device).view(1, 1, 1, -1)
    return (grid / size * 2 - 1).clamp(-1.1, 1.1)


def get_in_range(grid):
    ok = torch.ones_like(grid[..., 0], dtype=torch.bool)
    for dim in range(grid.size()[-1]):
        ok = ok & (grid[..., dim] <= 1) & (grid[..., dim] >= -1)
    return ok.to(grid.dtype)
