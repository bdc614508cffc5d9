"""Round 5 experiment: start stagger of the one-tile w4 GEMM launches (CPLXAMD_W4_STAGGER="groups,sleeps", read per
launch): the seven launch kinds of the bench step, same process, interleaved rounds, median ms.
    python scripts/r05/stagger_ab.py "2,6" "2,10" "4,3" ...
NEGATIVE result (profiles/r05_gemm_w4_persistent.txt section 5); the CPLXAMD_W4_STAGGER hook in the launcher was removed again,
this harness is kept as the record of what was run."""
import os
import statistics
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "r04"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import w4_ab  # noqa: E402

ROUNDS, PER = int(os.environ.get("ROUNDS", "7")), int(os.environ.get("PER", "8"))
configs = [""] + sys.argv[1:]
B, I, O = (int(v) for v in os.environ.get("BIO", "8192,4096,4096").split(","))
lib = w4_ab.load(os.path.join(os.path.dirname(w4_ab.L.__file__), "libcplxamd.so"))
lib.cplxamd_gemm_set_family(0x7f)          # one-tile w4 kernels everywhere (no persistent forms)
shapes = w4_ab.make(lib, B, I, O)
ref = {}
for s, (fn, _) in shapes.items():
    os.environ["CPLXAMD_W4_STAGGER"] = ""
    ref[s] = [t.clone() for t in fn()]
    for c in configs[1:]:
        os.environ["CPLXAMD_W4_STAGGER"] = c
        assert all(torch.equal(a, b) for a, b in zip(ref[s], fn())), (s, c)
times = {(c, s): [] for c in configs for s in shapes}
for r in range(ROUNDS):
    for s, (fn, _) in shapes.items():
        for c in (configs if r % 2 == 0 else configs[::-1]):
            os.environ["CPLXAMD_W4_STAGGER"] = c
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(PER):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[(c, s)].append(e0.elapsed_time(e1) / PER)
print(f"# B={B} I={I} O={O}; {ROUNDS} interleaved rounds x {PER} launches, median ms; stagger = groups,sleeps (x s_sleep(32) ~ 2048 clocks)")
print("stagger".ljust(10) + "".join(s.rjust(14) for s in shapes) + "sum".rjust(10))
for c in configs:
    med = [statistics.median(times[(c, s)]) for s in shapes]
    print((c or "off").ljust(10) + "".join(f"{m:14.4f}" for m in med) + f"{sum(med):10.4f}")
