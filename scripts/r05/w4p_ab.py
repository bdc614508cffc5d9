"""Round 5: the PERSISTENT one-wave-per-SIMD GEMM (ring through the output-tile boundaries, gemm_bf16_w4.hip: PERSIST;
kernel-family bit 7) against the one-tile form of the same family and the 8-wave family -- same library, same process,
per-call flags (cplxamd_*_fl), bit-exactness first, then interleaved timing (median / min over ROUNDS x PER launches).

    python scripts/r05/w4p_ab.py          (env: ROUNDS, PER, SHAPES=bench|cfg4|small|all)
The COMPLEX persistent form is off in production (slower): run with CPLXAMD_W4P_CPLX=1 to time it.
Launches the persistent form takes: bf16 output, plain epilogue, no bias -- complex forward WITHOUT bias (c_fwd0), the
plain complex input gradient (c_dgrad), the real forward / input gradient (r_fwd, r_dgrad)."""
import os
import statistics
import sys
from ctypes import c_void_p

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from cplxmodule_amd import _lib as L  # noqa: E402

ROUNDS = int(os.environ.get("ROUNDS", "7"))
PER = int(os.environ.get("PER", "8"))
SHAPES = os.environ.get("SHAPES", "bench")
E = L.LAUNCH_EXCLUSIVE
FAMS = [("w8", L.LAUNCH_FAMILY(0) | E), ("w4", L.LAUNCH_FAMILY(0x7f) | E), ("w4p", L.LAUNCH_FAMILY(0xff) | E)]


def make(lib, B, I, O, dev="cuda"):
    torch.manual_seed(0)
    bf = torch.bfloat16
    bound = (1.0 / (2 * I)) ** 0.5
    xr, xi = (torch.randn(B, I, device=dev).to(bf) for _ in range(2))
    wr, wi = (torch.empty(O, I, device=dev).uniform_(-bound, bound).to(bf) for _ in range(2))
    gr, gi = (torch.randn(B, O, device=dev).to(bf) for _ in range(2))
    a2 = (xr.float() ** 2 + xi.float() ** 2).to(bf)
    S = torch.empty(O, I, device=dev).uniform_(-12, 4).exp().to(bf)
    gs2 = torch.randn(B, O, device=dev).to(bf)
    y = [torch.empty(B, O, device=dev, dtype=bf) for _ in range(2)]
    dx = [torch.empty(B, I, device=dev, dtype=bf) for _ in range(2)]
    p = lambda t: c_void_p(t.data_ptr())  # noqa: E731
    st = c_void_p(torch.cuda.current_stream().cuda_stream)

    def chk(rc):
        assert rc == 0, rc

    def c_fwd0(fl):
        chk(lib.cplxamd_cgemm_fl(p(xr), p(xi), I, 1, p(wr), p(wi), I, 1, None, None, None, p(y[0]), p(y[1]), O, B, O, I, 0,
                                 L.BF16, L.BF16, 0, None, 0, None, 0, fl, st))
        return y

    def c_dgrad(fl):
        chk(lib.cplxamd_cgemm_fl(p(gr), p(gi), O, 1, p(wr), p(wi), 1, I, None, None, None, p(dx[0]), p(dx[1]), I, B, I, O, 1,
                                 L.BF16, L.BF16, 0, None, 0, None, 0, fl, st))
        return dx

    def r_fwd(fl):
        chk(lib.cplxamd_rgemm_fl(p(a2), I, 1, p(S), I, 1, None, None, 0, p(y[0]), O, B, O, I, L.BF16, L.BF16, 0, None, None, 0,
                                 fl, st))
        return y[:1]

    def r_dgrad(fl):
        chk(lib.cplxamd_rgemm_fl(p(gs2), O, 1, p(S), 1, I, None, None, 0, p(dx[0]), I, B, I, O, L.BF16, L.BF16, 0, None, None, 0,
                                 fl, st))
        return dx[:1]

    plans = {"c_fwd0": (1, B, O, I, 0, 0), "c_dgrad": (1, B, I, O, 0, 1), "r_fwd": (0, B, O, I, 0, 0), "r_dgrad": (0, B, I, O, 0, 1)}
    return {"c_fwd0": (c_fwd0, 8.0), "c_dgrad": (c_dgrad, 8.0), "r_fwd": (r_fwd, 2.0), "r_dgrad": (r_dgrad, 2.0)}, plans


def run(lib, tag, B, I, O, time_it=True):
    shapes, plans = make(lib, B, I, O)
    print(f"## {tag}: B={B} I={I} O={O}", flush=True)
    bad = 0
    for s, (fn, _) in shapes.items():
        outs = []
        for n, fl in FAMS:
            for t in fn(fl):
                t.zero_()
            outs.append([t.clone() for t in fn(fl)])
        torch.cuda.synchronize()
        kinds = [lib.cplxamd_gemm_plan(*plans[s], L.BF16, 0, fl, 0) for _, fl in FAMS]
        same = all(torch.equal(a, b) for o in outs[1:] for a, b in zip(outs[0], o))
        fin = all(torch.isfinite(t.float()).all().item() for t in outs[-1])
        if not same:
            bad += 1
            for (n, _), o in zip(FAMS[1:], outs[1:]):
                nd = sum((a != b).sum().item() for a, b in zip(outs[0], o))
                print(f"   {s:8s} {n}: {nd} elements differ from w8", flush=True)
        print(f"   {s:8s} kernels {kinds} (1 one-tile 8-wave, 2 persistent 8-wave, 3 w4, 6 w4 persistent): "
              f"{'identical bits' if same else 'MISMATCH'} (finite={fin})", flush=True)
    if not time_it:
        return bad
    times = {(n, s): [] for n, _ in FAMS for s in shapes}
    for n, fl in FAMS:
        for s, (fn, _) in shapes.items():
            for _ in range(3):
                fn(fl)
    torch.cuda.synchronize()
    for r in range(ROUNDS):
        for s, (fn, _) in shapes.items():
            order = FAMS if r % 2 == 0 else FAMS[::-1]
            for n, fl in order:
                fn(fl)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(PER):
                    fn(fl)
                e1.record()
                torch.cuda.synchronize()
                times[(n, s)].append(e0.elapsed_time(e1) / PER)
    flop = B * I * O
    print(f"# {ROUNDS} interleaved rounds x {PER} launches; median ms (min ms) [TF/s at the median; 8MNK complex, 2MNK real]")
    print("family".ljust(8) + "".join(s.rjust(30) for s in shapes))
    for n, _ in FAMS:
        row = n.ljust(8)
        for s, (_, mult) in shapes.items():
            med, mn = statistics.median(times[(n, s)]), min(times[(n, s)])
            row += f"{med:.4f} ({mn:.4f}) [{mult * flop / med / 1e9:6.0f}]".rjust(30)
        print(row, flush=True)
    return bad


def main():
    lib = L.load()
    bad = 0
    if SHAPES in ("small", "all"):
        # more tiles than CUs at every K-loop exit path: K tile counts 6 k, 6 k + 2, 6 k + 4
        for (B, I, O) in ((8192, 384, 4096), (8192, 448, 4096), (8192, 512, 4096), (16384, 1024, 2304), (8192, 128, 8192)):
            bad += run(lib, f"exit paths: K tiles {I // 32} / {O // 32}", B, I, O, time_it=False)
    if SHAPES in ("bench", "all"):
        bad += run(lib, "bench step (configs[1])", 8192, 4096, 4096)
    if SHAPES in ("cfg4", "all"):
        bad += run(lib, "configs[3] at batch 2^16", 65536, 2048, 2048)
    print("RESULT", "ok" if bad == 0 else f"{bad} mismatching launches")


if __name__ == "__main__":
    main()
