#!/usr/bin/env python3
"""TEST TOOL (not product): end-to-end training parity of the HIP path vs the CPU oracle -- same split,
same initial weights, same negatives every epoch, Adam(lr=0.01), full batch (reference tip.py:14-30).

north_star criterion: AUROC within +-0.002 of the reference path.  The literal reference needs
~470 s/epoch on CPU, so -- as SURVEY.md section 8(c) prescribes -- the 100-epoch
reference value comes from the oracle (validated against the reference's own code by the golden
fixtures).  The loop itself lives in tests/parity_harness.py (the driver-run 10-epoch version of this
check is tests/test_gpu_train_parity.py).

    python tools/auroc_parity.py --epochs 100 [--relations 1097] [--mod cat] [--tag r02]
Writes one JSON line (also to profiles/<tag>_auroc_parity.json with --tag).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from parity_harness import run_parity                                   # noqa: E402
from tip_amd.data import build_data_dict                                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--epochs', type=int, default=100)
    ap.add_argument('--relations', type=int, default=None)
    ap.add_argument('--min-pairs', type=int, default=None, help='500 = the 963 relations of the published runs')
    ap.add_argument('--mod', default='cat')
    ap.add_argument('--threads', type=int, default=16)
    ap.add_argument('--tag', default=None)
    args = ap.parse_args()
    dd = build_data_dict(max_relations=args.relations, min_pairs=args.min_pairs)
    res = run_parity(dd, args.mod, args.epochs, threads=args.threads, log=lambda s: print(s, flush=True))
    out = {k: v for k, v in res.items() if k not in ('snapshots', 'rec_hip', 'rec_oracle', 'loss')}
    out['loss_first'], out['loss_last'] = res['loss'][0], res['loss'][-1]
    out['reference_published'] = {'auprc': 0.948, 'where': 'analysis/evaluation.ipynb:192-195 (R = 963, 100 epochs, model/*.py scripts)'}
    print(json.dumps(out))
    if args.tag:
        os.makedirs('profiles', exist_ok=True)
        json.dump(out, open('profiles/%s_auroc_parity.json' % args.tag, 'w'), indent=1)


if __name__ == '__main__':
    main()
