#!/usr/bin/env python
"""What one rank does at the shard sizes of a strong-scaled 65,536-ray step (65,536 / 32,768 / 16,384 / 8,192 rays per
rank = 1 / 2 / 4 / 8 GPUs), measured on ONE GPU: `bench.py --rays R` per size and operand format, collected into one
JSON file (ms per step, rays/s, per-kernel ms, rays/s relative to the 65,536-ray step).

    python tools/shard_sizes.py gpurun_out/shard_sizes.json
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out_path = sys.argv[1]
    rows = {}
    for rays in (65536, 32768, 16384, 8192):
        steps = max(10, 655360 // rays // 2)
        cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--rays', str(rays), '--steps', str(steps), '--warmup', '5',
               '--no-cpu-baseline']
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        if res.returncode != 0:
            print(res.stderr[-2000:], file=sys.stderr)
            raise SystemExit(res.returncode)
        rec = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
        for fmt, r in (('bf16', rec), ('mx', rec['extra']['mx'])):
            rows.setdefault(fmt, {})[str(rays)] = {
                'ms_per_step': r['ms_per_step'], 'rays_per_s': r['value'], 'mlp_mfma_util': r.get('mlp_mfma_util'),
                'kernel_ms_per_step': r['kernel_ms_per_step'],
                'other_ms_per_step': r['ms_per_step'] - sum(r['kernel_ms_per_step'].values())}
        print(rays, 'bf16 %.3f ms' % rec['ms_per_step'], 'mx %.3f ms' % rec['extra']['mx']['ms_per_step'], flush=True)
    for fmt in rows:
        full = rows[fmt]['65536']['rays_per_s']
        for rays in rows[fmt]:
            rows[fmt][rays]['relative_to_65536'] = rows[fmt][rays]['rays_per_s'] / full
    json.dump({'note': 'one rank of a strong-scaled 65,536-ray C3 step, measured on one MI355X: bench.py --rays R, whole step '
                       '(fwd + loss + bwd + weight gradients + AdamW + repack), no collective; other_ms_per_step = step time '
                       'minus the MLP kernels (samplers, encoders, compositing, optimizer, repack, launch gaps)',
               'formats': rows}, open(out_path, 'w'), indent=1)


if __name__ == '__main__':
    main()
