#!/usr/bin/env python
"""Turn rocprofv3 --pmc CSVs (per operand format: one pass with FETCH_SIZE, one with WRITE_SIZE) into a per-kernel HBM
traffic summary (bytes per launch).

    python tools/pmc_summary.py <out.json> <format> <fetch counter_collection.csv> <write counter_collection.csv> [<format> ...]

Corrections applied (gfx950): both counters are in KiB; FETCH_SIZE tallies a wide coalesced 128-byte request
as 64 bytes, so the read side is doubled for kernels whose reads are 16-byte-per-lane streams (all of the
kernels listed here); WRITE_SIZE is exact for 16-byte-per-lane stores and float atomics.  The file records a hash of
the kernel sources it was taken from (bench.kernel_source_hash): bench.py quotes these byte counts only for that tree."""
import collections
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            name = re.sub(r'\(.*$', '', r['Kernel_Name']).replace('void ', '').strip()
            acc[name].append(float(r['Counter_Value']))
    return acc


def summarise(fetch_csv, write_csv):
    fetch, write = per_kernel(fetch_csv, 'FETCH_SIZE'), per_kernel(write_csv, 'WRITE_SIZE')
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, [0.0]), write.get(k, [0.0])
        rd = 2.0 * 1024.0 * sum(f) / len(f)
        wr = 1024.0 * sum(w) / len(w)
        if rd + wr < 2e7:
            continue
        out[k] = dict(launches=len(f), read_bytes_per_launch=rd, write_bytes_per_launch=wr,
                      traffic_bytes_per_launch=rd + wr)
        print(f"  {k:60s} {len(f):4d} launches  read {rd/1e9:8.3f} GB  write {wr/1e9:8.3f} GB")
    return out


def main():
    from bench import kernel_source_hash
    formats = {}
    args = sys.argv[2:]
    for i in range(0, len(args), 3):
        print(args[i])
        formats[args[i]] = summarise(args[i + 1], args[i + 2])
    json.dump(dict(note='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over `bench.py --steps 2 --warmup 1 '
                        '--wgrad-operands <format>`; FETCH_SIZE doubled (gfx950 counts 128-byte requests as 64 bytes); averages '
                        'over all launches of the kernel (both sampling levels)',
                   rays=65536, source_hash=kernel_source_hash(), formats=formats), open(sys.argv[1], 'w'), indent=1)


if __name__ == '__main__':
    main()
