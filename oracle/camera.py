"""Host-side camera / schedule math of the reference, restated in NumPy.

Test infrastructure only (see oracle/__init__.py).  PARITY UNPINNED.
"""
import numpy as np


def stage_of(iteration, stage_interval, max_stage):
    """Progressive-growing stage for an iteration.

    Follows updater.py:252-256 (RGBDUpdater.get_stage): first interval with
    iteration+1 <= interval gives  i-1 + (it - si[i-1]) / (si[i] - si[i-1]);
    past the table the stage saturates at max_stage - 1e-8.
    """
    for i, interval in enumerate(stage_interval):
        if iteration + 1 <= interval:
            prev = stage_interval[i - 1]
            return i - 1 + (iteration - prev) / (interval - prev)
    return max_stage - 1e-8


def parse_stage_interval(text):
    """updater.py:242 -- the YAML value is a comma string."""
    return [int(tok) for tok in str(text).split(",")]


def rotate_in_plane(mat, a1, a2, angle):
    """updater.py:26-42 (update_camera_matrices): left-multiply by a plane rotation."""
    n = len(mat)
    rot = np.zeros_like(mat)
    for d in range(4):
        rot[:, d, d] = 1
    rot[:, a1, a1] = np.cos(angle)
    rot[:, a1, a2] = -np.sin(angle)
    rot[:, a2, a1] = np.sin(angle)
    rot[:, a2, a2] = np.cos(angle)
    assert rot.shape == (n, 4, 4)
    return np.matmul(rot, mat)


def camera_matrices(thetas, order=(0, 1, 2)):
    """updater.py:45-60 (get_camera_matries): (n,6) pose -> (n,4,4) float32 cam2world.

    M0 = diag(1,1,-1,1) with M0[2,3] = 1; for i in order rotate plane
    ((i+1)%3, (i+2)%3) by theta_i; finally add the translation.
    """
    thetas = np.asarray(thetas)
    n = len(thetas)
    mat = np.zeros((n, 4, 4), dtype="float32")
    mat[:, 0, 0] = 1
    mat[:, 1, 1] = 1
    mat[:, 2, 2] = -1
    mat[:, 3, 3] = 1
    mat[:, 2, 3] = 1
    for i in order:
        mat = rotate_in_plane(mat, (i + 1) % 3, (i + 2) % 3, thetas[:, i])
    mat[:, :3, 3] = mat[:, :3, 3] + thetas[:, 3:]
    return mat


class PosePrior:
    """train_rgbd.py:192-217 (CameraParamPrior): paired poses for one batch.

    Draw order of np.random: uniform(b,6), uniform(b,6), choice(2,(b,3)).
    """

    def __init__(self, x_rotate, y_rotate, z_rotate, x_translate=0, y_translate=0, z_translate=0,
                 uniform=False):
        self.rotation_range = np.array([x_rotate, y_rotate, z_rotate])
        self.camera_param_range = np.array([x_rotate, y_rotate, z_rotate,
                                            x_translate, y_translate, z_translate])
        self.uniform = uniform

    def sample(self, batch_size):
        half = batch_size // 2
        t1 = np.random.uniform(-1, 1, size=(half, 6))
        eps = np.random.uniform(0, 0.5, size=(half, 6))
        sign = np.random.choice(2, size=(half, 3)) * 2 - 1
        limit = np.clip(1 / (self.rotation_range + 1e-8), 0, 1)
        if self.uniform:
            eps[:, :3] = eps[:, :3] * sign * limit
        else:
            full_turn = self.rotation_range == 3.1415
            eps[:, :3] = eps[:, :3] * (sign * full_turn + np.abs(sign) * (~full_turn)) * limit
        t2 = -eps * np.sign(t1) + t1
        if self.uniform:
            t2 = t2 * (-1 <= t2) * (t2 <= 1) + (-2 - t2) * (t2 < -1) + (2 - t2) * (t2 > 1)
        out = np.concatenate([t1, t2], axis=0) * self.camera_param_range[None]
        return out.astype("float32")


def theta9(thetas):
    """updater.py:317-318: [cos(rot xyz), sin(rot xyz), translation] as float32."""
    thetas = np.asarray(thetas, dtype="float32")
    return np.concatenate([np.cos(thetas[:, :3]), np.sin(thetas[:, :3]), thetas[:, 3:]], axis=1).astype("float32")
