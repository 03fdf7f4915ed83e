#!/usr/bin/env python3
"""usage: tools/kcat.py kernel_stats_A.txt [kernel_stats_B.txt]: per-category kernel ms/step (3 steps per stats file)"""
import re, sys
def load(f):
    rows = []
    for l in open(f):
        m = re.match(r'^(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)%\s+([\d.]+)', l)
        if m: rows.append((m.group(1).strip(), int(m.group(2)), float(m.group(3)) / 3))
    return rows
def cat(n):
    for k, v in (('conv_wide', 'conv_wide'), ('conv_tall', 'conv_tall'), ('conv_patch', 'conv_patch'), ('conv_gemm', 'conv_gemm'), ('conv_stream', 'conv_stream'), ('conv_thin', 'conv_thin'), ('conv_s2fwd', 'conv_s2fwd'), ('conv_toep', 'conv_toep'), ('dgrad_images', 'dgrad_images'), ('wgrad_tr', 'wgrad_tr'),
                 ('wgrad_reduce', 'wgrad_other'), ('conv_wgrad', 'wgrad_other'), ('head_', 'heads'), ('at::native', 'aten/rt'), ('rocclr', 'aten/rt'),
                 ('instnorm', 'instnorm'), ('moments', 'instnorm'), ('percep', 'percep'), ('sums_final', 'percep'), ('act_bwd', 'act_bwd'),
                 ('upsample', 'elementwise'), ('maxpool', 'elementwise'), ('mul_', 'elementwise'), ('residual', 'elementwise'), ('nchw', 'elementwise'),
                 ('nhwc', 'elementwise'), ('fold_', 'elementwise'), ('sn', 'sn/adam/pack'), ('dot_kernel', 'sn/adam/pack'), ('adam', 'sn/adam/pack'), ('pack', 'sn/adam/pack')):
        if k in n: return v
    return 'other'
tabs = [load(f) for f in sys.argv[1:]]
cats = {}
for i, rows in enumerate(tabs):
    for n, k, ms in rows:
        cats.setdefault(cat(n), [0.0] * len(tabs))[i] += ms
print('%-14s' % 'category' + ''.join('%12s' % f.split('/')[-1].replace('_kernel_stats.txt', '').replace('kernel_stats_', '')[:11] for f in sys.argv[1:]))
for c, v in sorted(cats.items(), key=lambda x: -x[1][0]):
    print('%-14s' % c + ''.join('%12.2f' % x for x in v))
print('%-14s' % 'TOTAL' + ''.join('%12.2f' % sum(r[2] for r in rows) for rows in tabs))
print('%-14s' % 'launches/step' + ''.join('%12d' % (sum(r[1] for r in rows) / 3) for rows in tabs))
