"""Small quaternion / rotation helpers (NumPy, fp64, Hamilton convention, w first).

Host-side only: used by the model compiler.  The device code has its own
versions in csrc/.
"""
import numpy as np


def quat_mul(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
        a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0],
    ])


def quat_conj(q):
    q = np.asarray(q, dtype=np.float64)
    return np.array([q[0], -q[1], -q[2], -q[3]])


def quat_normalize(q):
    q = np.asarray(q, dtype=np.float64)
    n = np.linalg.norm(q)
    if n < 1e-15:
        return np.array([1.0, 0.0, 0.0, 0.0])
    return q / n


def quat_to_mat(q):
    w, x, y, z = np.asarray(q, dtype=np.float64)
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
    ])


def mat_to_quat(R):
    R = np.asarray(R, dtype=np.float64)
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s])
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = np.array([(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s])
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = np.array([(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s])
    q = quat_normalize(q)
    if q[0] < 0:
        q = -q
    return q


def rot_vec_quat(v, q):
    return quat_to_mat(q) @ np.asarray(v, dtype=np.float64)


def axis_angle_quat(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    s = np.sin(0.5 * angle)
    return np.array([np.cos(0.5 * angle), axis[0] * s, axis[1] * s, axis[2] * s])


def principal_axes(I):
    """Symmetric 3x3 inertia -> (moments sorted descending, quat of principal frame).

    The descending order follows MuJoCo's eig3 convention [EXT]; the frame is
    made right-handed and the quaternion sign fixed (w >= 0) so the result is
    deterministic.
    """
    w, V = np.linalg.eigh(np.asarray(I, dtype=np.float64))
    order = np.argsort(-w, kind="stable")
    w = w[order]
    V = V[:, order]
    # deterministic column signs: largest-magnitude component positive
    for k in range(3):
        j = int(np.argmax(np.abs(V[:, k])))
        if V[j, k] < 0:
            V[:, k] = -V[:, k]
    if np.linalg.det(V) < 0:
        V[:, 2] = -V[:, 2]
    return w, mat_to_quat(V)
