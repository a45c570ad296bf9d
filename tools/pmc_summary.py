"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.

    python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json>

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  MI355X_MICROARCH.md (HBM section): on
gfx950 FETCH_SIZE counts 128-byte read requests at 64 bytes, so wide coalesced reads (16 B per lane:
every kernel here, LDS-DMA included) show exactly half their bytes -> doubled below; WRITE_SIZE is
exact for 16-byte-per-lane streaming stores.  Kernel names are reduced to the form bench.py's
plan_name() prints so its `roofline.traffic` can look its kernel up.
"""
import csv
import glob
import json
import os
import re
import sys


def short_name(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(gemm_pring_kernel)<(\d+), (\d+), \d+, \d+, \d+, \d+, (true|false), (\d+)>", name)
    if m:
        return f"{m.group(1)}<{m.group(2)},{m.group(3)},lean={int(m.group(4) == 'true')},epi={m.group(5)}>"
    m = re.match(r"(gemm_ringx_kernel)<(\d+), (\d+), \d+, \d+, \d+, \d+, \d+, (true|false)>", name)
    if m:
        return f"{m.group(1)}<{m.group(2)},{m.group(3)},lean={int(m.group(4) == 'true')}>"
    m = re.match(r"(gemm_ring_kernel)<(\d+), (\d+)", name)
    if m:
        return f"{m.group(1)}<{m.group(2)},{m.group(3)}>"
    m = re.match(r"(gemm_kernel)<(true|false), (true|false), (\d+), (\d+)", name)
    if m:
        return f"{m.group(1)}<{m.group(4)},{m.group(5)}>"
    return name.split("(")[0][:80]


def collect(d: str, counter: str):
    out = {}
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            e = out.setdefault(short_name(row["Kernel_Name"]), [0, 0.0])
            e[0] += 1
            e[1] += float(row["Counter_Value"])
    return out


def main():
    fetch = collect(sys.argv[1], "FETCH_SIZE")
    write = collect(sys.argv[2], "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        nf, kf = fetch.get(k, [0, 0.0])
        nw, kw = write.get(k, [0, 0.0])
        rd = 2.0 * kf * 1024 / nf if nf else 0.0       # gfx950 correction: x2
        wr = kw * 1024 / nw if nw else 0.0
        kernels[k] = {"launches": max(nf, nw), "hbm_read_bytes_per_launch": round(rd),
                      "hbm_write_bytes_per_launch": round(wr), "hbm_bytes_per_launch": round(rd + wr)}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over "
                         "`python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing`; "
                         "FETCH_SIZE x2 (gfx950), KiB -> bytes, averaged per launch",
               "kernels": kernels}, open(sys.argv[3], "w"), indent=1)
    top = sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]
    for k, v in top:
        print(f"{k[:70]:70s} n={v['launches']:5d} rd {v['hbm_read_bytes_per_launch'] / 1e6:8.1f} MB wr "
              f"{v['hbm_write_bytes_per_launch'] / 1e6:8.1f} MB")


if __name__ == "__main__":
    main()
