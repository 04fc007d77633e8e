"""SDF grid evaluation for mesh extraction (reference code/utils/plots.py:112-148, 203-240, 294-307).

The reference evaluates `sdf(points)` on a resolution^3 grid in 50 000-point chunks, each a PyTorch MLP forward with a device->host
copy, then runs scikit-image's marching cubes.  Here the grid points are GENERATED in chunks on the device and pushed through the
tracing-MLP kernel (`mvsdf_sdf_col0`, fp32 MFMA); only the fp32 volume comes back.  Marching cubes itself is scikit-image (a
third-party dependency of the reference, absent from this image): `get_surface_trace` raises ImportError without it.
"""
import numpy as np
import torch

from .. import ops


def get_grid_uniform(resolution, cpu=False):
    """plots.py:294-307: the [-1, 1]^3 grid, points ordered like np.meshgrid(x, y, z) (default 'xy' indexing) raveled."""
    x = np.linspace(-1.0, 1.0, resolution)
    xx, yy, zz = np.meshgrid(x, x, x)
    grid_points = torch.tensor(np.vstack([xx.ravel(), yy.ravel(), zz.ravel()]).T, dtype=torch.float)
    if not cpu:
        grid_points = grid_points.cuda()
    return {'grid_points': grid_points, 'shortest_axis_length': 2.0, 'xyz': [x, x, x], 'shortest_axis_index': 0}


def sdf_on_uniform_grid(sdf, resolution, chunk=1 << 22, device=None):
    """z[resolution^3] float32 numpy = sdf at get_grid_uniform(resolution)['grid_points'], without materialising the point list:
    chunks of grid points are built on the device from their flat index.  `sdf`: ImplicitNetwork.native_sdf() (packed weights ->
    the HIP tracing-MLP kernel) or any callable points[n,3] -> [n]."""
    n = resolution
    device = device or torch.device('cuda', torch.cuda.current_device())
    ax = torch.from_numpy(np.linspace(-1.0, 1.0, n)).to(device)                  # float64 like the reference, cast per point below
    total = n * n * n
    out = torch.empty(total, dtype=torch.float32, device=device)
    net = getattr(sdf, 'native_net', None)
    for s in range(0, total, chunk):
        idx = torch.arange(s, min(total, s + chunk), device=device)
        iz = idx % n
        ix = (idx // n) % n                                                      # np.meshgrid 'xy': xx[i, j, k] = x[j], yy = y[i], zz = z[k]
        iy = idx // (n * n)
        pts = torch.stack([ax[ix], ax[iy], ax[iz]], -1).to(torch.float32)
        out[s:s + pts.shape[0]] = ops.sdf_col0(net, pts) if net is not None else sdf(pts).reshape(-1)
    return out.cpu().numpy()


def lin2img(tensor, img_res):
    """[B, H*W, C] -> [B, C, H, W]  (plots.py:375-377)."""
    batch_size, num_samples, channels = tensor.shape
    return tensor.permute(0, 2, 1).reshape(batch_size, channels, img_res[0], img_res[1])


def surface_volume(model, resolution):
    """The SDF volume get_surface_high_res_mesh_simple marches over (plots.py:150-163): sdf = implicit_network(x)[:, 0] on the
    resolution^3 grid of get_grid_uniform, in the (y, x, z)-transposed layout marching cubes receives (plots.py:166-168).
    The reference pushes 50 000-point chunks through the autograd MLP and copies each back; here the points are generated on the device
    and evaluated by the tracing-MLP kernel (k_sdf_col0)."""
    z = sdf_on_uniform_grid(model.implicit_network.native_sdf(), resolution)
    return z.astype(np.float32).reshape(resolution, resolution, resolution).transpose([1, 0, 2])


def surface_vertex_colors(model, verts, chunk=1 << 20):
    """plots.py:179,200: surf_v = sigmoid(implicit_network(verts)[:, 1]) -> RGB (1 - surf_v, surf_v, 0) per vertex."""
    verts_t = torch.as_tensor(verts, dtype=torch.float32, device=next(model.parameters()).device)
    out = []
    with torch.no_grad():
        for v in torch.split(verts_t, chunk):
            out.append(model.implicit_network(v.contiguous())[:, 1].sigmoid())
    surf_v = torch.cat(out) if out else verts_t.new_zeros(0)
    return torch.stack([1 - surf_v, surf_v, torch.zeros_like(surf_v)], dim=-1)


def get_surface_high_res_mesh_simple(model, cams=None, resolution=100):
    """plots.py:150-205.  The grid evaluation and the vertex colours run on the HIP kernels; marching cubes / the mesh object are
    scikit-image / trimesh like in the reference (third-party, absent from this image -> ImportError)."""
    volume = surface_volume(model, resolution)
    try:
        from skimage import measure
        import trimesh
    except ImportError as e:
        raise ImportError('get_surface_high_res_mesh_simple needs scikit-image and trimesh (dependencies of the reference plots.py): %s' % e)
    x = np.linspace(-1.0, 1.0, resolution)
    mc = getattr(measure, 'marching_cubes_lewiner', None) or measure.marching_cubes
    verts, faces, normals, values = mc(volume=volume, level=0, spacing=(x[2] - x[1],) * 3)
    verts = verts + np.array([x[0], x[0], x[0]])
    color = surface_vertex_colors(model, verts)
    return trimesh.Trimesh(verts, faces, normals, vertex_colors=color.cpu().numpy())


def get_surface_trace(path, epoch, sdf, resolution=100, return_mesh=False):
    """plots.py:112-148 (marching cubes of the SDF volume).  Needs scikit-image + trimesh like the reference."""
    try:
        from skimage import measure
        import trimesh
    except ImportError as e:                                                      # third-party, not part of the hot path
        raise ImportError('get_surface_trace needs scikit-image and trimesh (dependencies of the reference plots.py): %s' % e)
    grid = get_grid_uniform(resolution, cpu=True)
    z = sdf_on_uniform_grid(sdf, resolution)
    if np.min(z) > 0 or np.max(z) < 0:
        return None
    x = grid['xyz'][0]
    verts, faces, normals, _ = measure.marching_cubes(volume=z.reshape(resolution, resolution, resolution).transpose([1, 0, 2]), level=0,
                                                      spacing=(x[2] - x[1],) * 3)
    verts = verts + np.array([x[0], x[0], x[0]])
    mesh = trimesh.Trimesh(verts, faces, normals)
    mesh.export('{0}/surface_{1}.obj'.format(path, epoch), 'obj')
    return mesh if return_mesh else None
