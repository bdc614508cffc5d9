"""Where the local-reparameterization noise comes from.

mode "philox" (default): the kernels generate the noise in registers from the counter-based
Philox stream (seed, offset) documented in DESIGN.md; nothing is materialised and the backward
regenerates it.  Each stochastic forward consumes one offset.
mode "philox-device": the same stream, but (seed, offset) live in a device int64[2] tensor that a
one-thread kernel advances: nothing about the noise position is baked into kernel arguments, so
a training step captured in a hipGraph (torch.cuda.CUDAGraph) draws fresh noise on every replay.
mode "torch": the layers draw the noise with torch.randn on the device in the reference's tape
layout (one [2, B, O] draw / sqrt 2 for complex layers, cplxmodule/cplx.py:544-550) and hand it
to the kernels -- bit-compatible with a recorded reference tape, 8-16 B/output more traffic.
mode "tape": the layers consume a recorded tape (`noise.set_tape([...])`): one raw normal draw per
stochastic forward, in call order -- [2, *shape] for complex layers (divided by sqrt 2 here exactly as
cplx.randn does), [*shape] for real ones.  This is how a training trajectory captured from the
reference is replayed step for step (tests/test_gpu_trajectory.py).
"""
import math
import torch


class _NoiseState:
    def __init__(self):
        self.mode = "philox"
        self._seed = None
        self.counter = 0
        self._dev = {}

    @property
    def seed(self):
        base = torch.initial_seed() if self._seed is None else self._seed
        # data-parallel ranks draw decorrelated noise from the same user seed (dp.DataParallel)
        return (base + getattr(self, "_rank", 0) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF

    def fold_rank(self, rank):
        """Fold a data-parallel rank into the Philox key (rank 0 leaves the stream unchanged)."""
        self._rank = int(rank)
        self._dev = {}

    def manual_seed(self, seed):
        self._seed, self.counter = int(seed), 0
        self._dev = {}

    def set_mode(self, mode):
        if mode not in ("philox", "philox-device", "torch", "tape"):
            raise ValueError("noise mode must be 'philox', 'philox-device', 'torch' or 'tape'")
        self.mode = mode

    def set_tape(self, tensors):
        """Switch to "tape" mode with the given raw draws (consumed front to back)."""
        self._tape = list(tensors)
        self.mode = "tape"

    def pop_tape(self, shape, like, complex_):
        if not getattr(self, "_tape", None):
            raise RuntimeError("noise tape exhausted")
        t = self._tape.pop(0).to(device=like.device, dtype=like.dtype)
        want = ((2, *shape) if complex_ else tuple(shape))
        if tuple(t.shape) != tuple(want):
            raise RuntimeError(f"noise tape entry has shape {tuple(t.shape)}, the layer needs {tuple(want)}")
        return t / math.sqrt(2) if complex_ else t

    def next(self, device=None):
        """(seed, offset) for one stochastic forward pass; in "philox-device" mode a device
        int64[2] copy of the position (and 0) instead, the device-resident position advanced."""
        if self.mode == "philox-device":
            from ... import ops
            return ops.philox_advance(self.device_state(device)), 0
        self.counter += 1
        return self.seed & 0xFFFFFFFFFFFFFFFF, self.counter

    def device_state(self, device):
        device = torch.device(device)
        key = device.index if device.index is not None else torch.cuda.current_device()
        if key not in self._dev:
            seed = self.seed & 0xFFFFFFFFFFFFFFFF
            seed = seed - (1 << 64) if seed >= (1 << 63) else seed      # same 64 bits, as int64
            self._dev[key] = torch.tensor([seed, self.counter + 1], dtype=torch.int64,
                                          device=device)
        return self._dev[key]


noise = _NoiseState()
