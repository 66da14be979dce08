#!/usr/bin/env python
"""Per-kernel matrix-pipe occupancy and shader clock from one rocprofv3 counter pass.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d DIR -- python3 bench.py ...
    python tools/chain_counters.py DIR/*_counter_collection.csv [out.json]

clock     = GRBM_GUI_ACTIVE / 8 / duration   (the counter is summed over the 8 XCDs; reads high on launches < 0.3 ms)
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs): share of the SIMD cycles of the
            launch in which the matrix pipe was executing (32 cycles per v_mfma_f32_32x32x16_bf16)."""
import collections
import csv
import json
import re
import sys


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    dur, cnt, seen = collections.defaultdict(float), collections.Counter(), set()
    for r in csv.DictReader(open(sys.argv[1])):
        k = re.sub(r'\(.*$', '', r['Kernel_Name']).replace('void ', '').strip()
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id'])
            cnt[k] += 1
            dur[k] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    out = {}
    for k, d in sorted(agg.items(), key=lambda kv: -dur[kv[0]]):
        if not k.startswith('mlp_') or d['GRBM_GUI_ACTIVE'] == 0:
            continue
        cyc = d['GRBM_GUI_ACTIVE'] / 8
        out[k] = dict(launches=cnt[k], avg_ms=dur[k] / cnt[k] / 1e6, clock_ghz=cyc / dur[k],
                      mfma_busy=d['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 256 * 4))
        print('%-62s %3d launches  avg %7.3f ms  clock %.2f GHz  matrix pipe busy %4.1f %%' % (
            k, cnt[k], out[k]['avg_ms'], out[k]['clock_ghz'], 100 * out[k]['mfma_busy']))
    if len(sys.argv) > 2:
        json.dump(dict(note='rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE over `bench.py --steps 2 '
                            '--warmup 1`; clock = GRBM_GUI_ACTIVE / 8 / duration, mfma_busy = MFMA busy cycles / SIMD cycles',
                       kernels=out), open(sys.argv[2], 'w'), indent=1)


if __name__ == '__main__':
    main()
