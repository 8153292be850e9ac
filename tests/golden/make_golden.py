"""Regenerates tests/golden/genea140_phi_oracle.npy: the full 140 x 140 Float32 kinship
matrix of the bundled genea140 pedigree as computed by the CPU oracle (oracle/genphi_oracle.c).

The reference (Julia) cannot run in the build container, so this fixture is ORACLE-derived,
not reference-derived; the oracle itself is pinned by reference_pinned.json.  It exists so
that the GPU parity test has a committed vector that does not depend on building the oracle,
and so that a change in the oracle shows up as a diff.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

ped = O.Pedigree.from_file(os.path.join(os.path.dirname(os.path.dirname(HERE)), "genlib.jl_amd", "data", "genea140.csv"))
phi = ped.phi()
np.save(os.path.join(HERE, "genea140_phi_oracle.npy"), phi)
print("genea140:", phi.shape, float(phi.astype(np.float64).sum()))
