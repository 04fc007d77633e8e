"""First-order differentiable ray/surface intersection (IDR eq. 3; reference code/model/sample_network.py:10-20).

At the traced point the value f equals its detached copy f0, so numerically x(theta) == cam + t * dir; what matters is the
gradient: d x / d theta = -(d f / d theta) / (grad f0 . v0) * v, i.e. SURVEY.md App. E.6's scalar upstream on the hit point's SDF value."""
import torch.nn as nn


class SampleNetwork(nn.Module):
    def forward(self, surface_output, surface_sdf_values, surface_points_grad, surface_dists, surface_cam_loc, surface_ray_dirs):
        v0 = surface_ray_dirs.detach()                       # the direction inside the dot product carries no gradient
        slope = (surface_points_grad * v0).sum(dim=-1, keepdim=True)          # grad f0 . v0
        residual = surface_output - surface_sdf_values       # == 0 in value, d/d theta = d f / d theta
        t_of_theta = surface_dists - residual / slope
        return surface_cam_loc + surface_ray_dirs * t_of_theta
