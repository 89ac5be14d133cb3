#!/usr/bin/env python
"""Freeze the reference's ARAP example INPUT mesh as a fixture: examples/data/raptor_simplify2k.off (2000 vertices, 4036
triangles) and its landmark file raptor_simplify2k.mrk (11 markers) -> tests/fixtures/raptor2k_mesh.npz.

Data only (vertex positions, triangle indices, marker indices / targets), read with opt_amd.io.read_off / read_mrk; run in the
container that has /root/reference mounted:      python tests/fixtures/make_raptor_mesh.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from opt_amd import io      # noqa: E402

DATA = "/root/reference/examples/data"
V, F = io.read_off(os.path.join(DATA, "raptor_simplify2k.off"))
idx, pos = io.read_mrk(os.path.join(DATA, "raptor_simplify2k.mrk"))
np.savez_compressed(os.path.join(HERE, "raptor2k_mesh.npz"), vertices=V.astype(np.float32), faces=np.array(F, dtype=np.int32),
                    marker_index=idx.astype(np.int32), marker_position=pos.astype(np.float32))
print("wrote raptor2k_mesh.npz:", V.shape, len(F), "faces,", len(idx), "markers")
