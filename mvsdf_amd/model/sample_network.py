"""Differentiable intersection point x(theta) (reference code/model/sample_network.py:10-20, IDR eq. 3)."""
import torch.nn as nn


class SampleNetwork(nn.Module):
    def forward(self, surface_output, surface_sdf_values, surface_points_grad, surface_dists, surface_cam_loc, surface_ray_dirs):
        dirs0 = surface_ray_dirs.detach()
        dot = (surface_points_grad * dirs0).sum(-1, keepdim=True)              # bmm([n,1,3],[n,3,1]) in the reference
        t_theta = surface_dists - (surface_output - surface_sdf_values) / dot
        return surface_cam_loc + t_theta * surface_ray_dirs
