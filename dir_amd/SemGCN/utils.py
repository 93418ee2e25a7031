"""Mirror of the two SemGCN/utils.py helpers the network uses (SemGCN/utils.py:27-43,66-71)."""
import numpy as np
import torch


def get_sketch_setting():
    return [[0, 1], [1, 2], [2, 3], [3, 4], [0, 5], [5, 6], [6, 7], [7, 8], [0, 9], [9, 10], [10, 11], [11, 12],
            [0, 13], [13, 14], [14, 15], [15, 16], [0, 17], [17, 18], [18, 19], [19, 20]]


def adj_mx_from_edges(num_pts, edges, sparse=False, eye=True):
    """symmetric, row-normalised adjacency (dense float tensor).  Only `adj > 0` matters downstream."""
    if sparse:
        raise NotImplementedError('sparse adjacency is not used on the DIR path')
    a = np.zeros((num_pts, num_pts), np.float32)
    for i, j in edges:
        a[i, j] = a[j, i] = 1.0
    if eye:
        a = a + np.eye(num_pts, dtype=np.float32)
    rs = a.sum(1, keepdims=True)
    a = np.where(rs > 0, a / np.maximum(rs, 1e-30), 0.0)
    return torch.tensor(a, dtype=torch.float)
