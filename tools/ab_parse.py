"""Condense tools/bench_gemm.py lines of an A/B log: name, t0 / t3 / auto / nosplit microseconds."""
import re, sys
for l in open(sys.argv[1]):
    if l.startswith("=="):
        print(l.strip()); continue
    name = l[:14]
    f = dict(re.findall(r"(t ?-?\d[pr]|nosplit):\s+[\d.]+ TF \(\s*(\d+) us\)", l))
    if f:
        print(name, " ".join(f"{k.replace(' ', '')}={v}" for k, v in f.items() if k.replace(' ', '') in ("t0p", "t2p", "t3p", "t4p", "t5p", "t-1p", "nosplit")))
