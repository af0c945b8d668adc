"""``seed_everything`` -- the one ``tgm.util`` entry every example script calls before building its loader
(role of tgm/util/seed.py:11-25).  The device-side negative sampler draws from its own counter RNG seeded by the
hook's ``seed`` argument (``tgm_amd/hooks/negatives.py``); this seeds the three host generators torch-side code reads."""
from __future__ import annotations

import random

import numpy as np
import torch


def seed_everything(seed: int) -> None:
    for seeder in (random.seed, np.random.seed, torch.manual_seed):
        seeder(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
