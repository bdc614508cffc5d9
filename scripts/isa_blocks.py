"""For every kernel of an AMDGPU assembly file: the basic blocks that contain MFMAs and their
instruction mix (a K loop split over many small blocks = run-time branches inside the loop)."""
import re
import sys
from collections import Counter


def blocks(body):
    labs = [n for n, l in enumerate(body) if re.match(r"^\.L", l)]
    bounds = [0] + labs + [len(body)]
    for a, b in zip(bounds[:-1], bounds[1:]):
        yield body[a].strip(), [x.strip() for x in body[a + 1:b]]


def mix(blk):
    c = Counter()
    for x in blk:
        if not x or x[0] in ";.":
            continue
        op = x.split()[0]
        key = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else
               "waitcnt" if op.startswith("s_waitcnt") else "barrier" if op.startswith("s_barrier") else
               "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else
               "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else op)
        c[key] += 1
    return c


s = open(sys.argv[1]).read()
for m in re.finditer(r"\n(_Z\w+):\s*; @", s):
    name = m.group(1)
    end = s.index(".Lfunc_end", m.end())
    body = s[m.end():end].split("\n")
    rows = [(lab, mix(b)) for lab, b in blocks(body)]
    rows = [(lab, c) for lab, c in rows if c["mfma"]]
    if not rows:
        continue
    total = sum(c["mfma"] for _, c in rows)
    print(f"{name[:90]}: {total} MFMAs in {len(rows)} blocks")
    for lab, c in rows:
        if c["mfma"] >= 4 and (len(sys.argv) > 2 or "Loop" in lab):
            print("    ", lab[:40], dict(c))
