"""Timings of the other BASELINE.json configs on one MI355X (JSON lines):
cfg3  CplxConv2d(64,64,3) on 256x256 + CplxBatchNorm2d(64), fwd+bwd      (--cfg3-batch, default 32)
cfg4  CplxLinearVD(2048,2048) LRT + KL, fwd+bwd, batch 2^20 / shard       (--cfg4-batch)
cfg1  CplxLinear(128,128) + LinearVD(128,128), batch 64, fp32
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cplxmodule_amd import Cplx, nn
from cplxmodule_amd.nn import relevance as rel


def timed(fn, iters, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def cfg1():
    dev = "cuda"
    a, b = nn.CplxLinear(128, 128).to(dev), rel.LinearVD(128, 128).to(dev)
    x = Cplx(torch.randn(64, 128, device=dev), torch.randn(64, 128, device=dev))
    xr = torch.randn(64, 128, device=dev)

    def step():
        for m in (a, b):
            m.zero_grad(set_to_none=True)
        y = a(x)
        z = b(xr)
        loss = (y.real ** 2).sum() + (y.imag ** 2).sum() + (z ** 2).sum() + 1e-3 * sum(rel.penalties(b))
        loss.backward()
    t = timed(step, 200, 20)
    return {"config": "cfg1 CplxLinear(128,128)+LinearVD(128,128) B=64 fp32 fwd+bwd", "ms": t * 1e3,
            "samples_per_s": 64 / t, "reference_cpu_samples_per_s": 33400}


def cfg1_graph():
    """cfg1 with the whole fwd+bwd step captured in one hipGraph (device-resident noise position)."""
    dev = "cuda"
    rel.noise.set_mode("philox-device")
    a, b = nn.CplxLinear(128, 128).to(dev), rel.LinearVD(128, 128).to(dev)
    x = Cplx(torch.randn(64, 128, device=dev), torch.randn(64, 128, device=dev))
    xr = torch.randn(64, 128, device=dev)

    def step():
        y = a(x)
        z = b(xr)
        loss = (y.real ** 2).sum() + (y.imag ** 2).sum() + (z ** 2).sum() + 1e-3 * sum(rel.penalties(b))
        loss.backward()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    for m in (a, b):
        m.zero_grad(set_to_none=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    t = timed(g.replay, 500, 20)
    rel.noise.set_mode("philox")
    return {"config": "cfg1 (one hipGraph replay per step) CplxLinear(128,128)+LinearVD(128,128) B=64 fp32",
            "ms": t * 1e3, "samples_per_s": 64 / t, "reference_cpu_samples_per_s": 33400}


def cfg3(batch, dtype, layout="channels_last"):
    """layout: how the input images are stored.  channels_last (torch.channels_last, [B, H, W, C] in memory) is what the
    bf16 kernels work on end to end; with a plain contiguous (NCHW) input the layer converts on entry and autograd
    converts the input gradient back, two extra passes each way that a layer inside a network does not pay."""
    dev = "cuda"
    conv, bn = nn.CplxConv2d(64, 64, 3).to(dev), nn.CplxBatchNorm2d(64).to(dev)
    fmt = torch.channels_last if layout == "channels_last" else torch.contiguous_format
    mk = lambda: torch.randn(batch, 64, 256, 256, device=dev).to(dtype).contiguous(memory_format=fmt).requires_grad_(True)  # noqa: E731
    x = Cplx(mk(), mk())
    times = {}

    def step():
        conv.zero_grad(set_to_none=True); bn.zero_grad(set_to_none=True)
        x.real.grad = x.imag.grad = None
        y = bn(conv(x))
        torch.autograd.backward((y.real, y.imag), (y.real.detach(), y.imag.detach()))
    t = timed(step, 10, 2)
    flop = 8.0 * batch * 64 * 254 * 254 * 64 * 9 * 3
    with torch.no_grad():
        y = conv(x)
        tb = timed(lambda: bn(y), 5, 2)
    nelem = batch * 64 * 254 * 254
    return {"config": f"cfg3 CplxConv2d(64,64,3)@256x256 + CplxBatchNorm2d, B={batch} {dtype} fwd+bwd, input {layout}",
            "ms": t * 1e3, "images_per_s": batch / t, "conv_TFLOPs_algorithmic": flop / 1e12,
            "achieved_TFLOP_s_incl_bn": flop / t / 1e12, "bn_fwd_ms": tb * 1e3,
            "bn_fwd_GBps(24B/elt fp32, 12 bf16)": (24 if dtype == torch.float32 else 12) * nelem / tb / 1e9,
            "reference_cpu_images_per_s": 3.0}


def cfg4(batch, dtype):
    dev = "cuda"
    layer = rel.CplxLinearVD(2048, 2048).to(dev)
    with torch.no_grad():
        layer.log_sigma2.uniform_(-12, 4)
    x = Cplx(torch.randn(batch, 2048, device=dev).to(dtype).requires_grad_(True),
             torch.randn(batch, 2048, device=dev).to(dtype).requires_grad_(True))
    klw = torch.tensor(1e-3, device=dev)

    def step():
        layer.zero_grad(set_to_none=True)
        x.real.grad = x.imag.grad = None
        y = layer(x)
        kl = sum(rel.penalties(layer))
        torch.autograd.backward((y.real, y.imag, kl), (y.real.detach() * 2, y.imag.detach() * 2, klw))
    t = timed(step, 3, 1)
    flop = (8 + 2) * 3.0 * batch * 2048 * 2048
    return {"config": f"cfg4 CplxLinearVD(2048,2048) LRT+KL B={batch} {dtype} fwd+bwd", "ms": t * 1e3,
            "samples_per_s": batch / t, "achieved_TFLOP_s": flop / t / 1e12,
            "reference_cpu_samples_per_s": 1878}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg3-batch", type=int, default=32)
    ap.add_argument("--cfg4-batch", type=int, default=1 << 20)
    ap.add_argument("--cfg3-layout", default=None, choices=["channels_last", "nchw"],
                    help="default: channels_last for bf16 (the kernels' layout), nchw for float32 (planar kernels)")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    torch.manual_seed(0)
    jobs = [("cfg1", cfg1), ("cfg1g", cfg1_graph), ("cfg3f", lambda: cfg3(a.cfg3_batch, torch.float32, a.cfg3_layout or "nchw")),
            ("cfg3b", lambda: cfg3(a.cfg3_batch, torch.bfloat16, a.cfg3_layout or "channels_last")),
            ("cfg4b", lambda: cfg4(a.cfg4_batch, torch.bfloat16)),
            ("cfg4f", lambda: cfg4(a.cfg4_batch // 4, torch.float32))]
    for name, fn in jobs:
        if a.only and name not in a.only.split(","):
            continue
        print(json.dumps(fn()), flush=True)
        torch.cuda.empty_cache()
