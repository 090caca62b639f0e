#!/usr/bin/env python3
"""The reference's `tip.py` training script on the MI355X-native modules: same flow, same constants,
same final `torch.save(model, ...)`.  Differences: the import line, the device (no CPU path), and by
default the epoch loop is replayed as one hipGraph (`--eager` runs the reference's loop line for line).
Run from the repo root:

    python examples/tip.py [cat|add] [epochs] [--eager] [--log=train_log.jsonl]

The loss of every epoch is printed as the reference prints it (tip.py:28); `--log=FILE` ('-' = stdout) also writes one JSON
object per epoch -- {"epoch", "loss", "ms_per_epoch", "train_edges_per_s"} -- and a final {"event": "test", "auprc", "auroc",
"ap"} record (metrics / logging).

If ./data/data_dict.pkl (written by the reference's prepare.py) exists it is used; otherwise the
bundled BioSNAP graph is split with seed 1111.
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tip_amd.layers import Setting, TIP          # reference: `from src.layers import *`   # noqa: E402

MOD = sys.argv[1] if len(sys.argv) > 1 else 'cat'
MAX_EPOCH = int(sys.argv[2]) if len(sys.argv) > 2 else 100
GRAPH = '--eager' not in sys.argv      # default: the whole step replays as one hipGraph (tip_amd/train.py)
LOG = next((a.split('=', 1)[1] for a in sys.argv if a.startswith('--log=')), None)
log_f = None if LOG is None else (sys.stdout if LOG == '-' else open(LOG, 'w'))


def log(**rec):
    if log_f is not None:
        log_f.write(json.dumps(rec) + '\n')
        log_f.flush()


device = torch.device('cuda:0')                  # no CPU path in this build

torch.manual_seed(1111)
if MOD == 'cat':
    settings = Setting(sp_rate=0.9, lr=0.01, prot_drug_dim=16, n_embed=48, n_hid1=32, n_hid2=16, num_base=32)
    model = TIP(settings, device)
else:
    settings = Setting(sp_rate=0.9, lr=0.01, prot_drug_dim=64, n_embed=64, n_hid1=32, n_hid2=16, num_base=32)
    model = TIP(settings, device, mod='add')

from tip_amd.optim import Adam                             # tipk_adam_step: the whole parameter list in one launch
optimizer = Adam(model.parameters(), lr=settings.lr)       # (torch.optim.Adam(..., capturable=GRAPH) works as well)

torch.cuda.synchronize()
t0 = time.perf_counter()
if GRAPH:
    from tip_amd.train import GraphedTrainStep
    model.train()
    step = GraphedTrainStep(model, optimizer, warmup=2)       # 2 eager epochs, then the captured graph
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    losses = [step() .clone() for e in range(MAX_EPOCH - 2)]  # no host synchronisation inside the loop
    torch.cuda.synchronize()
    per = (time.perf_counter() - t1) / max(1, MAX_EPOCH - 2)
    for e, v in enumerate(torch.stack(losses).tolist()):
        print(v)
        log(epoch=e + 2, loss=v, ms_per_epoch=per * 1e3, train_edges_per_s=model.data.dd_train_idx.shape[1] / per)
else:
    for e in range(MAX_EPOCH):                                # the reference's loop, line for line
        te = time.perf_counter()
        model.train()
        optimizer.zero_grad()
        loss = model()
        print(loss.item())
        loss.backward()
        optimizer.step()
        if log_f is not None:
            torch.cuda.synchronize()
            per = time.perf_counter() - te
            log(epoch=e, loss=loss.item(), ms_per_epoch=per * 1e3, train_edges_per_s=model.data.dd_train_idx.shape[1] / per)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('%d epochs in %.2f s (%.1f ms/epoch, %.2f M train edges/s incl. sampler, decoder, loss, Adam)'
      % (MAX_EPOCH, dt, dt / MAX_EPOCH * 1e3, model.data.dd_train_idx.shape[1] * MAX_EPOCH / dt / 1e6))

record = model.test()
if record is not None:
    auprc, auroc, ap = (float(v) for v in record.sum(axis=1) / record.shape[1])
    log(event='test', auprc=auprc, auroc=auroc, ap=ap, relations=int(record.shape[1]))

os.makedirs('saved_model', exist_ok=True)
torch.save(model, f'saved_model/tip-{model.mod}-example.pt')      # whole model, as tip.py:36 (load: weights_only=False)
