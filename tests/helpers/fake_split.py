"""Writes a tiny synthetic split in the layout dataset/prepare_data.py:123-166 produces (img/<idx>.jpg 256x256, anno/<idx>.pkl)."""
import os
import pickle

import numpy as np


def write_split(root, n, split='test', seed=0, size=256):
    from PIL import Image
    rng = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, split, 'img'), exist_ok=True)
    os.makedirs(os.path.join(root, split, 'anno'), exist_ok=True)
    yy, xx = np.mgrid[0:size, 0:size]
    for i in range(n):
        # smooth content (JPEG-friendly) + a little texture
        img = np.stack([127 + 100 * np.sin(xx / (7.0 + i) + c) * np.cos(yy / (11.0 + c)) for c in range(3)], -1) + rng.normal(0, 6, (size, size, 3))
        Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(os.path.join(root, split, 'img', '%d.jpg' % i), quality=92)
        a = rng.normal(0, 0.2, 3)
        th = np.linalg.norm(a) + 1e-9
        k = a / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R = (np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx).astype(np.float32)
        anno = {'inter_idx': i, 'camera': {'R': R, 't': np.array([0.01 * i, -0.02, 0.8], np.float32),
                                          'camera': np.array([[1500. + i, 0, 128.], [0, 1490., 126.], [0, 0, 1.]])},
                'mano_params': {}}
        for side in ('left', 'right'):
            b = rng.normal(0, 0.4, 3)
            tb = np.linalg.norm(b) + 1e-9
            kb = b / tb
            Kb = np.array([[0, -kb[2], kb[1]], [kb[2], 0, -kb[0]], [-kb[1], kb[0], 0]])
            anno['mano_params'][side] = {'R': (np.eye(3) + np.sin(tb) * Kb + (1 - np.cos(tb)) * Kb @ Kb).astype(np.float32)[None],
                                         'pose': rng.normal(0, 0.5, (1, 45)).astype(np.float32),
                                         'shape': rng.normal(0, 0.5, (1, 10)).astype(np.float32),
                                         'trans': (np.array([-0.06 if side == 'left' else 0.06, 0, 0]) + rng.normal(0, 0.02, 3)).astype(np.float32)[None]}
        with open(os.path.join(root, split, 'anno', '%d.pkl' % i), 'wb') as f:
            pickle.dump(anno, f)
