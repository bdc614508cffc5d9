"""Where the local-reparameterization noise comes from.

mode "philox" (default): the kernels generate the noise in registers from the counter-based
Philox stream (seed, offset) documented in DESIGN.md; nothing is materialised and the backward
regenerates it.  Each stochastic forward consumes one offset.
mode "torch": the layers draw the noise with torch.randn on the device in the reference's tape
layout (one [2, B, O] draw / sqrt 2 for complex layers, cplxmodule/cplx.py:544-550) and hand it
to the kernels -- bit-compatible with a recorded reference tape, 8-16 B/output more traffic.
"""
import torch


class _NoiseState:
    def __init__(self):
        self.mode = "philox"
        self._seed = None
        self.counter = 0

    @property
    def seed(self):
        return torch.initial_seed() if self._seed is None else self._seed

    def manual_seed(self, seed):
        self._seed, self.counter = int(seed), 0

    def set_mode(self, mode):
        if mode not in ("philox", "torch"):
            raise ValueError("noise mode must be 'philox' or 'torch'")
        self.mode = mode

    def next(self):
        """(seed, offset) for one stochastic forward pass."""
        self.counter += 1
        return self.seed & 0xFFFFFFFFFFFFFFFF, self.counter


noise = _NoiseState()
