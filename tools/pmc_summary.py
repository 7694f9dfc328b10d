"""Summarise rocprofv3 --pmc CSV output (one or more passes) per kernel.

usage: python tools/pmc_summary.py <out.json> <dir> [<dir> ...]

Every <dir> is searched recursively for *counter_collection.csv.  Per kernel (short name) and
counter the per-dispatch mean is reported.  HBM bytes follow MI355X_MICROARCH.md §HBM:
FETCH_SIZE / WRITE_SIZE are in KiB, and on gfx950 FETCH_SIZE reports half the bytes of wide
(16 B/lane) coalesced reads -> `hbm_read_bytes_corrected = 2 * FETCH_SIZE * 1024`.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = n.replace('unsigned short', 'bf16')
    return n.split('(')[0][:90]


def main():
    out_json = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))     # kernel -> counter -> [sum, n]
    grids = defaultdict(set)
    for d in sys.argv[2:]:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            with open(f, newline='') as fh:
                for row in csv.DictReader(fh):
                    k = short(row['Kernel_Name'])
                    a = acc[k][row['Counter_Name']]
                    a[0] += float(row['Counter_Value'])
                    a[1] += 1
                    grids[k].add(row.get('Grid_Size', ''))
    res = {}
    for k, cs in acc.items():
        e = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
        e['_dispatches'] = max(v[1] for v in cs.values())
        if 'FETCH_SIZE' in e:
            e['hbm_read_bytes_corrected'] = 2.0 * e['FETCH_SIZE'] * 1024.0
        if 'WRITE_SIZE' in e:
            e['hbm_write_bytes'] = e['WRITE_SIZE'] * 1024.0
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in e and 'SQ_BUSY_CYCLES' in e and e['SQ_BUSY_CYCLES'] > 0:
            e['mfma_busy_over_sq_busy'] = e['SQ_VALU_MFMA_BUSY_CYCLES'] / e['SQ_BUSY_CYCLES']
        res[k] = e
    with open(out_json, 'w') as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    names = sorted({c for e in res.values() for c in e})
    print('| kernel | ' + ' | '.join(names) + ' |')
    print('|---|' + '---|' * len(names))
    for k in sorted(res, key=lambda k: -res[k].get('SQ_BUSY_CYCLES', res[k].get('FETCH_SIZE', 0))):
        print(f'| `{k}` | ' + ' | '.join(f'{res[k][c]:.4g}' if c in res[k] else '' for c in names) + ' |')


if __name__ == '__main__':
    main()
