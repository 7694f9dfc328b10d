"""the cosine shortfalls of smoke() against the bf16 mock, sorted, under given dispatch bits of odtk_debug_set(6, ...)   python tools/smoke_gaps.py [bits ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
from odtk import ops
mock = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'ssd300_bf16_mock_small.json')))['smoke']
for arg in sys.argv[1:] or ['0']:
    bits, _, brk = arg.partition(':')          # BITS[:layer whose input-gradient filter loses a tap in the bf16 engine]
    bits = int(bits)
    ops.debug_set(6, bits)
    mt = G.smoke_metrics(break_layer=brk or None)
    ops.debug_set(6, 0)
    gaps = sorted(((mock['gradient'][k][0] - c, k, round(c, 4), mock['gradient'][k][0]) for k, (c, r) in mt['bf16_vs_f32'].items()), reverse=True)
    print('bits', bits, 'broken', brk or '-', 'loss bf16', round(mt['loss_bf16'], 4), 'f32', round(mt['loss_f32'], 4), 'worst shortfalls', [(k, round(g, 4), c, mc) for g, k, c, mc in gaps[:6]], flush=True)
