"""Training-set generation (reference graphlearning/trainsets.py:47-156), restated: it
defines the inputs of every parity test, so it must draw the same indices as the reference."""
import sys
import numpy as np


def generate(labels, rate=1, num_trials=1, mask=None, seed=None):
    """Random per-class label sets.  `rate`: int = labels per class; float in [0,1] =
    fraction per class; (m,C) or (m,1) array = several sub-trials.  Seeds the global
    numpy RNG and draws class by class with np.random.choice(n, size, p, replace=False)
    exactly like the reference (trainsets.py:89-90, 121-128)."""
    labels = np.asarray(labels)
    if seed is not None:
        np.random.seed(seed)
    classes = np.unique(labels)
    per_class = np.bincount(labels)
    num_classes = len(classes)
    n = len(labels)
    if type(rate) == int:
        counts = (np.ones(num_classes)[None, :] * rate).astype(int)
    elif type(rate) == float:
        counts = (rate * per_class[None, :]).astype(int)
    elif type(rate) == np.ndarray:
        kind = rate.dtype
        if rate.ndim != 2:
            sys.exit('Must provide a 2-dimensional array for rate')
        if rate.shape[1] == 1:
            rate = rate @ np.ones((1, num_classes))
        if np.issubdtype(kind, np.integer):
            counts = rate.astype(int)
        elif np.issubdtype(kind, np.floating):
            counts = (rate * per_class).astype(int)
        else:
            sys.exit('Invalid numpy array type ' + str(kind))
    else:
        sys.exit('Invalid rate type ' + str(type(rate)))
    if mask is None:
        mask = np.ones(n, dtype=bool)
    out = []
    for _ in range(num_trials):
        for row in range(counts.shape[0]):
            picked = []
            for j, c in enumerate(classes):
                p = ((labels == c) & mask).astype(float)
                p = p / np.sum(p)
                picked = picked + np.random.choice(n, size=counts[row, j], p=p, replace=False).tolist()
            out.append(np.array(picked))
    if len(out) == 1:
        out = out[0]
    return out
