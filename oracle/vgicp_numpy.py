"""Independent numpy restatement of the reference's VGICP loop — the cross-check that pins the C++
oracle (oracle/vgicp_oracle.cpp), since the reference itself has no tests and cannot be built here.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED by the reference's own tests (it has none).

Deliberately written differently from the C++ oracle: dense per-correspondence Jacobians through
einsum, numpy.linalg for the 3x3 inverse and the 6x6 solve, a sorted-key table instead of a hash
map.  Agreement is therefore tolerance-based (tests use 1e-9 relative on the normal equations).

Follows: src/Registration.cpp:7-102, src/LocalMap.cpp:78-118, include/ESKF_LIO/LocalMap.hpp:63-89,
src/Utils.cpp:5-11,28-32,40-63 of the reference; Open3D PointCloud::Transform semantics
(p <- R p + t, C <- R C R^T).
"""
from __future__ import annotations

import numpy as np


def skew(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def rotation_from_vector(r):
    angle = np.linalg.norm(r)
    if angle == 0.0:
        return np.eye(3)
    k = r / angle
    K = skew(k)
    return np.eye(3) + np.sin(angle) * K + (1.0 - np.cos(angle)) * (K @ K)


def left_jacobian(r):
    angle = np.linalg.norm(r)
    if angle < 1e-6:
        return np.eye(3)
    k = r / angle
    f1 = np.sin(angle) / angle
    f2 = (1.0 - np.cos(angle)) / angle
    return f1 * np.eye(3) + (1.0 - f1) * np.outer(k, k) + f2 * skew(k)


def se3_to_SE3(xi):
    xi = np.asarray(xi, dtype=np.float64)
    T = np.eye(4)
    T[:3, :3] = rotation_from_vector(xi[3:])
    T[:3, 3] = left_jacobian(xi[3:]) @ xi[:3]
    return T


def convergence_check(step, cosine_threshold, translation_sq_threshold):
    cosine = 0.5 * (np.trace(step[:3, :3]) - 1.0)
    if cosine < cosine_threshold:
        return False
    return not (float(step[:3, 3] @ step[:3, 3]) > translation_sq_threshold)


def voxel_index(points, voxel_size):
    return np.floor(np.asarray(points) / voxel_size).astype(np.int32)


def covs_to_matrices(covs9):
    """N x 9 column-major -> N x 3 x 3 indexed [n, row, col]."""
    return np.asarray(covs9).reshape(-1, 3, 3).transpose(0, 2, 1)


class NumpyMap:
    """Voxel statistics by the reference's insertion rule, looked up through sorted packed keys."""

    def __init__(self, voxel_size, max_points_per_voxel=1):
        self.voxel_size = float(voxel_size)
        self.cap = int(max_points_per_voxel)
        self._dict = {}  # key tuple -> [count, mean(3), cov(3x3)]
        self._frozen = None

    def insert(self, points, covs9):
        C = covs_to_matrices(covs9)
        keys = voxel_index(points, self.voxel_size)
        for p, c, k in zip(np.asarray(points), C, map(tuple, keys)):
            v = self._dict.get(k)
            if v is None:
                self._dict[k] = [1, p.copy(), c.copy()]
            elif v[0] < self.cap:
                n = v[0]
                v[1] = (n * v[1] + p) / (n + 1)
                v[2] = (n * v[2] + c) / (n + 1)
                v[0] = n + 1
        self._frozen = None

    def __len__(self):
        return len(self._dict)

    @staticmethod
    def _pack(keys):
        k = np.asarray(keys, dtype=np.int64)
        return ((k[:, 0] + (1 << 20)) << 42) | ((k[:, 1] + (1 << 20)) << 21) | (k[:, 2] + (1 << 20))

    def _freeze(self):
        if self._frozen is None:
            keys = np.array(list(self._dict.keys()), dtype=np.int64).reshape(-1, 3)
            packed = self._pack(keys) if len(keys) else np.zeros(0, dtype=np.int64)
            order = np.argsort(packed)
            vals = list(self._dict.values())
            means = np.array([vals[i][1] for i in order]).reshape(-1, 3)
            covs = np.array([vals[i][2] for i in order]).reshape(-1, 3, 3)
            self._frozen = (packed[order], means, covs)
        return self._frozen

    def match(self, points):
        """-> (indices of matched points, voxel means, voxel covariances[n,r,c])."""
        packed, means, covs = self._freeze()
        q = self._pack(voxel_index(points, self.voxel_size))
        if len(packed) == 0:
            return np.zeros(0, dtype=np.int64), np.zeros((0, 3)), np.zeros((0, 3, 3))
        pos = np.clip(np.searchsorted(packed, q), 0, len(packed) - 1)
        hit = packed[pos] == q
        idx = np.flatnonzero(hit)
        return idx, means[pos[hit]], covs[pos[hit]]


def normal_equations(points, covs, map_means, map_covs):
    """Sum of J^T S^-1 J and J^T S^-1 r over correspondences; J = [I | -skew(p)], S = Cp + Cv."""
    n = points.shape[0]
    if n == 0:
        return np.zeros((6, 6)), np.zeros(6)
    J = np.zeros((n, 3, 6))
    J[:, 0, 0] = J[:, 1, 1] = J[:, 2, 2] = 1.0
    x, y, z = points[:, 0], points[:, 1], points[:, 2]
    J[:, 0, 4], J[:, 0, 5] = z, -y
    J[:, 1, 3], J[:, 1, 5] = -z, x
    J[:, 2, 3], J[:, 2, 4] = y, -x
    W = np.linalg.inv(covs + map_covs)
    JT = np.einsum("nki,nkc->nic", J, W)
    r = points - map_means
    return np.einsum("nic,ncj->ij", JT, J), np.einsum("nic,nc->i", JT, r)


def solve_step(JTJ, JTr):
    if not JTJ.any() and not JTr.any():
        return np.eye(4)  # Eigen's LDLT returns the zero vector for the zero system (K3)
    return se3_to_SE3(np.linalg.solve(JTJ, -JTr))


def align(vmap: NumpyMap, points, covs9, guess, max_iteration, translation_sq_threshold,
          cosine_threshold):
    """ICP::align with the incremental in-place transform of the reference."""
    pts = np.asarray(points, dtype=np.float64).copy()
    C = covs_to_matrices(covs9).copy()
    total = np.asarray(guess, dtype=np.float64).copy()

    def move(T):
        nonlocal pts, C
        R = T[:3, :3]
        pts = pts @ R.T + T[:3, 3]
        C = R @ C @ R.T

    move(total)
    counts, JTJs, JTrs = [], [], []
    converged = False
    for _ in range(max_iteration):
        idx, mu, cv = vmap.match(pts)
        JTJ, JTr = normal_equations(pts[idx], C[idx], mu, cv)
        counts.append(len(idx))
        JTJs.append(JTJ)
        JTrs.append(JTr)
        step = solve_step(JTJ, JTr)
        total = step @ total
        if convergence_check(step, cosine_threshold, translation_sq_threshold):
            converged = True
            break
        move(step)
    return total, np.array(counts, dtype=np.uint64), np.array(JTJs), np.array(JTrs), converged


def preprocess(points, voxel_size, knn=30):
    """voxelDownsampleAndEstimateCovariances restated with numpy (src/CloudPreprocessor.cpp:76-127):
    first point per voxel, brute-force k nearest neighbours, population covariance, numpy's SVD for
    U diag(1, 1, 1e-2) V^T. Output in ascending original index."""
    pts = np.asarray(points, dtype=np.float64)
    keys = voxel_index(pts, voxel_size)
    _, first = np.unique(keys, axis=0, return_index=True)
    kept = np.sort(first)
    k = min(knn, pts.shape[0])
    out = np.zeros((len(kept), 3, 3))
    F = np.diag([1.0, 1.0, 1e-2])
    for o, i in enumerate(kept):
        d = ((pts - pts[i]) ** 2).sum(axis=1)
        nn = np.argsort(d, kind="stable")[:k]
        if k >= 3:
            x = pts[nn]
            cov = (x[:, :, None] * x[:, None, :]).mean(axis=0) - np.outer(x.mean(axis=0), x.mean(axis=0))
        else:
            cov = np.eye(3)
        U, _, Vt = np.linalg.svd(cov)
        out[o] = U @ F @ Vt
    return pts[kept], out, kept


def _quat_matrix(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _slerp(a, b, t):
    d = float(a @ b)
    if abs(d) >= 1.0 - np.finfo(np.float64).eps:
        s0, s1 = 1.0 - t, t
    else:
        th = np.arccos(abs(d))
        s0, s1 = np.sin((1.0 - t) * th) / np.sin(th), np.sin(t * th) / np.sin(th)
    if d < 0:
        s1 = -s1
    return s0 * a + s1 * b


def deskew(points, point_time, states):
    """CloudPreprocessor::deskew restated with 4x4 matrices (src/CloudPreprocessor.cpp:25-74,
    src/Utils.cpp:65-75). -> (points, number of leading points moved) or (points, -1)."""
    pts = np.asarray(points, dtype=np.float64).copy()
    t = np.asarray(point_time, dtype=np.float64)
    st = np.asarray(states, dtype=np.float64).reshape(-1, 8)
    n, S = len(pts), len(st)
    if n == 0 or S == 0:
        return pts, 0
    t_end = t[-1]
    before = np.flatnonzero(st[:, 0] <= t_end)
    if len(before) == 0 or before[-1] + 1 >= S:
        return pts, -1
    b = int(before[-1])
    a = b + 1
    f = (t_end - st[b, 0]) / (st[a, 0] - st[b, 0] + 1e-6)
    T_end = np.eye(4)
    T_end[:3, :3] = _quat_matrix(_slerp(st[b, 4:], st[a, 4:], f))
    T_end[:3, 3] = st[b, 1:4] + f * (st[a, 1:4] - st[b, 1:4])
    T_end_inv = np.linalg.inv(T_end)
    start = end = 0
    for s in range(a + 1):
        start = end
        later = np.flatnonzero(t[start:] >= st[s, 0])       # first point at or after this state's time
        if len(later) == 0:
            continue                                          # search ran to the end: nothing moves
        end = start + int(later[0])
        if end == start:
            continue
        T = np.eye(4)
        T[:3, :3] = _quat_matrix(st[s, 4:])
        T[:3, 3] = st[s, 1:4]
        T = T_end_inv @ T
        pts[start:end] = pts[start:end] @ T[:3, :3].T + T[:3, 3]
    return pts, end
