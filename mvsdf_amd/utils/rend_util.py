"""Ray generation helpers with the reference's names (reference code/utils/rend_util.py:48-100, 141-162), HIP backed."""
from .. import ops


def get_camera_params(uv, pose, intrinsics):
    """uv[B,P,2], pose[B,4,4], intrinsics[B,4,4] -> ray_dirs[B,P,3], cam_loc[B,3]   (rend_util.py:48-75; pose-matrix branch)."""
    if pose.shape[1] == 7:
        raise NotImplementedError('quaternion poses (train_cameras) are disabled in the reference (exp_runner.py:40) and not built')
    return ops.camera_rays(uv, pose, intrinsics)


def get_sphere_intersection(cam_loc, ray_directions, r=1.0):
    """-> sphere_intersections[B,P,2], mask_intersect[B,P]   (rend_util.py:141-162)."""
    return ops.sphere_intersection(cam_loc, ray_directions, r)
