"""Mask construction restated with plain Python/numpy integer arithmetic (bit-exact work).
Follows train_model.py:157-209 including quirks Q1/Q2 of SURVEY.md."""
import math
import random

import numpy as np


def crop_window(random_crop, H, W, window_len1=0.45, window_len2=0.45):
    """Row/col bounds zeroed by setCroppedMask — train_model.py:172-181 (Python-float floor)."""
    x_start = math.floor((random_crop[0] / 100) * H)
    y_start = math.floor((random_crop[1] / 100) * W)
    x_end = x_start + math.floor(window_len1 * H)
    y_end = y_start + math.floor(window_len2 * W)
    return x_start, min(x_end, H), y_start, min(y_end, W)


def masks_part1(shapes, layer_idx):
    """setMasksPart1 — train_model.py:184-194: ones for the chosen layer, zeros elsewhere."""
    return [np.ones(s, np.float32) if i == layer_idx else np.zeros(s, np.float32) for i, s in enumerate(shapes)]


def masks_part2(shapes, layer_idx, rng=random):
    """setMasksPart2 — train_model.py:197-209.  Draws the crop with two rng.randint(20, 40) calls.
    Layers below `layer_idx` get ones with a zeroed window; the chosen layer and everything above it
    end up all-zero (the `else` branch overwrites the ones set at :202 — quirk Q2)."""
    random_crop = (rng.randint(20, 40), rng.randint(20, 40))
    out = []
    for i, s in enumerate(shapes):
        if i < layer_idx:
            m = np.ones(s, np.float32)
            r0, r1, c0, c1 = crop_window(random_crop, s[2], s[3])
            m[:, :, r0:r1, c0:c1] = 0
        else:
            m = np.zeros(s, np.float32)
        out.append(m)
    return out, random_crop


def draw_iteration(rng, train1_prob, n_taps=6, fixed_layer=None):
    """The Python-RNG draws of one loop iteration — train_model.py:301,327,330,201 in order.
    Returns (train_type_is_1, layer, crop or None).  `random.choices` returns a list, so the reference's
    `train_type == 1` is always False (quirk Q1): part-2 masks are always used."""
    train_type = rng.choices([1, 2], [train1_prob, 1 - train1_prob])
    layer = rng.randint(1, n_taps - 2) if fixed_layer is None else fixed_layer
    is1 = (train_type == 1)            # always False, on purpose
    return is1, layer


def isolate_layer(arrays, selected):
    """isolate_layer — train_model.py:157-165."""
    return [a.copy() if i == selected else np.zeros_like(a) for i, a in enumerate(arrays)]
