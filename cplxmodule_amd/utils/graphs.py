"""hipGraph capture of whole training steps (HIP graphs instead of a tracing compiler).

A step of the hot path is a fixed sequence of 15-200 kernel launches through ctypes; at the small end (BASELINE
configs[0], configs[4]) the host cannot issue them as fast as the device retires them, and even the headline step
loses ~3 % to the gaps between its 16 launches.  `GraphedStep` records one invocation of a user function
(forward + loss + backward [+ gradient exchange + optimizer step]) into a `torch.cuda.CUDAGraph` (= hipGraph) and
replays it; the user keeps the same tensors (inputs are static buffers that `replay(*new_inputs)` copies into).

What makes a step of this package capturable:
  * noise: `noise.set_mode("philox-device")` keeps the Philox stream position in device memory and a one-thread
    kernel advances it, so every replay draws fresh noise (tests/test_gpu_graph.py);
  * no host synchronisation on the path (the KL is a device scalar, `sparsity()`'s `.item()` stays outside);
  * gradients: captured steps WRITE `.grad` (they do not accumulate into tensors of an earlier step): the capture
    runs with `.grad = None`, so every replay reproduces "zero_grad(set_to_none=True); backward()";
  * data parallel: the bucket all-reduces (RCCL kernels) are captured with the step -- the hook's side stream is
    joined back into the capturing stream (dp.BucketHook._launch), `sync_gradients()` must be called inside the step;
    the capture waits for the process group's watchdog to retire the warm-up steps' collectives and runs in
    thread-local capture mode (`_settle_collective_watchdog`);
  * no autograd graph of an EARLIER eager step may be alive when the capture starts: its AccumulateGrad nodes
    remember the stream they were created on (usually the default stream), autograd then runs them there, and work on a
    non-capturing stream in the middle of a capture makes hipStreamEndCapture crash.  `GraphedStep` therefore drops the
    fused-KL caches of `modules` (they hold the previous step's KL scalar, i.e. its graph), collects garbage, and runs
    warm-up and capture on ONE side stream; tensors of earlier steps that the caller still holds (a loss kept for
    logging) must be released by the caller.
"""
import gc

import torch


def drop_autograd_state(modules):
    """Forget the fused KL terms cached on the layers of `modules` (each holds the autograd graph of the step that
    produced it); the fusion stays armed."""
    for root in modules:
        for m in root.modules():
            if getattr(m, "_kl_cache", None) is not None:
                m._kl_cache = None
    gc.collect()


def _settle_collective_watchdog(seconds=0.3):
    """The RCCL process group's watchdog thread polls the completion events of the collectives the eager warm-up steps
    issued (every ~100 ms) until it has seen them complete; a `hipEventQuery` from that thread while THIS thread captures
    in the default (global) capture mode fails with "operation not permitted when stream is capturing" and takes the
    process down (seen once in ~10 runs of the data-parallel cfg5 capture).  The device is idle here, so every such
    event is complete: give the watchdog a few polling periods to retire them.  (Collectives issued DURING a capture are
    not handed to the watchdog.)"""
    import time

    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and "nccl" in str(dist.get_backend()):
        time.sleep(seconds)


class GraphedStep:
    """Capture `fn(*inputs)` once, replay it many times.

    fn       : callable running one full step on `inputs` (tensors; `Cplx` pairs are passed as their planes);
               it must set every `.grad` it produces from None (call `zero_grad(set_to_none=True)` first) and may
               return tensors (loss, KL, outputs), which stay valid -- and are overwritten -- across replays.
    inputs   : example tensors; their storage becomes the static input buffers.
    modules  : the modules `fn` runs (their cached autograd state is dropped before the warm-up, see above).
    warmup   : eager invocations on the capture stream before the capture (allocator warm-up, lazy initialisation,
               KL-fusion arming).  The capture records the kernels of the step as it is AT THAT MOMENT: paths that arm
               themselves on first use (the fused KL of the VD / ARD layers; the conv -> batch-norm moments epilogue,
               which a batch-norm layer requests for the NEXT step) need at least one warm-up step -- or explicit arming
               at build time (`conv.arm_conv_bn(model)`) -- to be part of a capture made with warmup=0.
    """

    def __init__(self, fn, inputs=(), modules=(), warmup=3, pool=None):
        self.fn = fn
        self.static_inputs = tuple(inputs)
        dev = self.static_inputs[0].device if self.static_inputs else torch.device("cuda", torch.cuda.current_device())
        drop_autograd_state(modules)
        self.stream = side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            out = None
            for _ in range(warmup):
                out = fn(*self.static_inputs)
            del out
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        _settle_collective_watchdog()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: an event query from ANOTHER thread (the process group's watchdog) must not invalidate the capture
        with torch.cuda.graph(self.graph, pool=pool, stream=side, capture_error_mode="thread_local"):
            self.outputs = fn(*self.static_inputs)

    def replay(self, *inputs):
        """Run the captured step; new input values (same shapes / dtypes) are copied into the static buffers."""
        if inputs:
            if len(inputs) != len(self.static_inputs):
                raise ValueError(f"expected {len(self.static_inputs)} inputs, got {len(inputs)}")
            for dst, src in zip(self.static_inputs, inputs):
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src)
        self.graph.replay()
        return self.outputs

    __call__ = replay
