"""The bf16 admission gate for every class that defaults (or could default) to the bf16 engine, in DETERMINISTIC mode (round-5 review, items 1 and 2).

For each class: f32 training from the seeded initial weights on seeded synthetic batches with the fixed-order filter-gradient reduction
(tools/bf16_after_training.py), the bf16-against-f32 filter-gradient comparison at initialisation and at each checkpoint, and the sha256 of the trained state
at each checkpoint.  Run twice in one process (`repeat`), the hashes and every cosine must repeat; across boxes the printed table must be the same text.

    python tools/gate_table.py [classes=ssd300,yolov3,fcos,centernet,yolov2,retinanet] [checkpoints=300,600,1000] [repeat=2] [probes=1]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bf16_after_training as T          # noqa: E402

BATCH = {'retinanet': 2}


def main():
    names = (sys.argv[1] if len(sys.argv) > 1 else 'ssd300,yolov3,fcos,centernet,yolov2,retinanet').split(',')
    cps = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else '300,600,1000').split(',')]
    repeat = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    probes = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    engine = sys.argv[5] if len(sys.argv) > 5 else 'bf16'
    out = {}
    for name in names:
        runs = []
        for rep in range(repeat):
            r = T.run(name, steps=max(cps), batch=BATCH.get(name, 4), lr=1e-3, verbose=(rep == 0), checkpoints=cps, probes=probes, engine=engine)
            runs.append(r)
        same = all(r['table'] == runs[0]['table'] and r['init'] == runs[0]['init'] for r in runs)
        out[name] = dict(init=runs[0]['init'], table={str(k): v for k, v in runs[0]['table'].items()}, reproducible=same,
                         hashes=[[r['table'][k][2] for k in cps] for r in runs])
        print(f'GATE {name}: reproducible in-process: {same}; init min/third {runs[0]["init"][0]:.4f} / {runs[0]["init"][1]:.4f}; ' +
              '; '.join(f'{k}: min {v[0]:.4f} third {v[1]:.4f} state {v[2]}' for k, v in runs[0]['table'].items()), flush=True)
    print('GATE_JSON ' + json.dumps(out))


if __name__ == '__main__':
    main()
