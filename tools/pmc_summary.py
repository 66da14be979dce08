#!/usr/bin/env python
"""Turn rocprofv3 --pmc CSVs (one pass with FETCH_SIZE, one with WRITE_SIZE) into a per-kernel HBM traffic
summary (bytes per launch).

    python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

Corrections applied (gfx950): both counters are in KiB; FETCH_SIZE tallies a wide coalesced 128-byte request
as 64 bytes, so the read side is doubled for kernels whose reads are 16-byte-per-lane streams (all of the
kernels listed here); WRITE_SIZE is exact for 16-byte-per-lane stores and float atomics."""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            name = re.sub(r'\(.*$', '', r['Kernel_Name']).replace('void ', '').strip()
            acc[name].append(float(r['Counter_Value']))
    return acc


def main():
    fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, [0.0]), write.get(k, [0.0])
        rd = 2.0 * 1024.0 * sum(f) / len(f)
        wr = 1024.0 * sum(w) / len(w)
        if rd + wr < 2e7:
            continue
        out[k] = dict(launches=len(f), read_bytes_per_launch=rd, write_bytes_per_launch=wr,
                      traffic_bytes_per_launch=rd + wr)
    json.dump(dict(note='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over `bench.py --steps 2 --warmup 1`; '
                        'FETCH_SIZE doubled (gfx950 counts 128-byte requests as 64 bytes); averages over all launches '
                        'of the kernel (both sampling levels)', kernels=out), open(sys.argv[3], 'w'), indent=1)
    for k, v in out.items():
        print(f"{k:60s} {v['launches']:4d} launches  read {v['read_bytes_per_launch']/1e9:8.3f} GB  write {v['write_bytes_per_launch']/1e9:8.3f} GB")


if __name__ == '__main__':
    main()
