#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + optional FETCH_SIZE / WRITE_SIZE passes) into the small
summaries committed under profiles/.   usage: summarize_profile.py <stats_dir> <fetch_dir> <write_dir> <out_prefix>"""
import csv, glob, json, os, sys, collections

stats_dir, fetch_dir, write_dir, out = sys.argv[1:5]


def one(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return f[0] if f else None


rows = list(csv.DictReader(open(one(stats_dir, "*kernel_stats.csv"))))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(out + "_kernel_stats.csv", "w") as f:
    f.write("kernel,calls,total_ms,avg_us,pct\n")
    for r in rows:
        f.write(f"\"{r['Name'][:140]}\",{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},{float(r['AverageNs'])/1e3:.2f},"
                f"{100*float(r['TotalDurationNs'])/tot:.2f}\n")
gemm = [r for r in rows if "gemm_f32_kernel" in r["Name"]]
g_calls = sum(int(r["Calls"]) for r in gemm)
g_ns = sum(float(r["TotalDurationNs"]) for r in gemm)
summary = {"total_kernel_ms": tot / 1e6, "gemm_f32_kernel": {"launches": g_calls, "total_ms": g_ns / 1e6,
                                                            "avg_us_per_launch": g_ns / g_calls / 1e3,
                                                            "share_of_gpu_time": g_ns / tot}}


def pmc(d, counter):
    f = one(d, "*counter_collection.csv")
    if not f:
        return None
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and "gemm_f32_kernel" in r["Kernel_Name"]:
            tot += float(r["Counter_Value"])
            n += 1
    return tot, n


fe, wr = pmc(fetch_dir, "FETCH_SIZE"), pmc(write_dir, "WRITE_SIZE")
if fe and wr and fe[1] and wr[1]:
    # MI355X_MICROARCH.md §HBM: both counters are in KiB; on gfx950 FETCH_SIZE reports exactly half the bytes of a wide
    # coalesced (16 B/lane) streaming read -> doubled; WRITE_SIZE is exact for 16 B/lane stores (our epilogue stores are
    # 4 B/lane, "uncalibrated" per the guide: taken at face value).
    read_b = 2.0 * fe[0] * 1024 / fe[1]
    write_b = wr[0] * 1024 / wr[1]
    summary["gemm_f32_kernel"].update({"hbm_read_bytes_per_launch": read_b, "hbm_write_bytes_per_launch": write_b,
                                       "hbm_bytes_per_launch": read_b + write_b,
                                       "fetch_size_raw_kib_per_launch": fe[0] / fe[1],
                                       "write_size_raw_kib_per_launch": wr[0] / wr[1]})
json.dump(summary, open(out + "_summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
