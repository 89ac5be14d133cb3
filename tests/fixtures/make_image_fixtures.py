#!/usr/bin/env python
"""Freeze real INPUT images of the reference's image examples as small fixtures (data only, decoded with opt_amd.io.read_png, sampled
every 4th pixel like the examples' own `downsampleFactor`):
  cat_mask_128.npz      red channel of examples/data/cat512_mask.png            (image_warping: a pixel is solved where it is 0)
  poisson_real_112x80.npz   examples/data/poisson0.png (base), poisson1.png (pasted image, same placement rule as main.cpp:34-40:
                        top-left corner), poisson_mask.png (255 = solve)        (poisson_image_editing)
Run where /root/reference is mounted:   python tests/fixtures/make_image_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from opt_amd import io      # noqa: E402

DATA = "/root/reference/examples/data"
m = io.read_png(os.path.join(DATA, "cat512_mask.png"))[::4, ::4, 0]
np.savez_compressed(os.path.join(HERE, "cat_mask_128.npz"), mask_red=m)
base = io.read_png(os.path.join(DATA, "poisson0.png"))[::4, ::4, :3]
ins = io.read_png(os.path.join(DATA, "poisson1.png"))[::4, ::4, :3]
pm = io.read_png(os.path.join(DATA, "poisson_mask.png"))[::4, ::4, 0]
H, W = base.shape[:2]
large = np.zeros_like(base)
h, w = min(H, ins.shape[0]), min(W, ins.shape[1])
large[:h, :w] = ins[:h, :w]
mask = np.zeros((H, W), dtype=np.uint8)
mask[:min(H, pm.shape[0]), :min(W, pm.shape[1])] = pm[:H, :W]
np.savez_compressed(os.path.join(HERE, "poisson_real_%dx%d.npz" % (W, H)), base=base, inserted=large, mask=mask)
print("cat mask", m.shape, "solved fraction %.3f" % (m == 0).mean(), "| poisson", base.shape, ins.shape, pm.shape, "solved fraction %.3f" % (mask == 255).mean())
