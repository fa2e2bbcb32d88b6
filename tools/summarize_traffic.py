#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected separately, never with a
trace domain).  usage: summarize_traffic.py <fetch_dir> <write_dir> <out.json> [kernel-name substring ...]
Units / corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports half the
bytes of wide coalesced (16 B / lane) streaming reads -> doubled; WRITE_SIZE is taken at face value."""
import collections, csv, glob, json, os, re, sys

fetch_dir, write_dir, out = sys.argv[1:4]
want = sys.argv[4:]


def load(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    tot, n = collections.defaultdict(float), collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(.*", "", re.sub(r"<.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))).strip()
        tot[name] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), name)
        if key not in seen:
            seen.add(key)
            n[name] += 1
    return tot, n


fe, fn = load(fetch_dir, "FETCH_SIZE")
wr, wn = load(write_dir, "WRITE_SIZE")
rep = {}
for name in sorted(fe, key=lambda k: -(2 * fe[k] + wr.get(k, 0.0))):
    if want and not any(w in name for w in want):
        continue
    n = max(1, fn[name])
    rb, wb = 2.0 * fe[name] * 1024 / n, wr.get(name, 0.0) * 1024 / max(1, wn.get(name, n))
    rep[name] = {"launches": n, "hbm_read_bytes_per_launch": rb, "hbm_write_bytes_per_launch": wb,
                 "hbm_bytes_per_launch": rb + wb, "fetch_size_raw_kib_per_launch": fe[name] / n,
                 "write_size_raw_kib_per_launch": wr.get(name, 0.0) / max(1, wn.get(name, n))}
rep = dict(list(rep.items())[:14])
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); FETCH doubled per MI355X_MICROARCH.md",
           "kernels": rep}, open(out, "w"), indent=1)
for k, v in rep.items():
    print(f"{k[:40]:40s} launches {v['launches']:5d} read {v['hbm_read_bytes_per_launch'] / 1e6:9.1f} MB  write "
          f"{v['hbm_write_bytes_per_launch'] / 1e6:9.1f} MB per launch")
