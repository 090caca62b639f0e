"""Run a few graphed training epochs (for rocprofv3 --kernel-trace)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tip_amd.layers import TIP, Setting
from tip_amd.train import GraphedTrainStep
torch.manual_seed(1111)
m = TIP(Setting(), torch.device('cuda:0'))
from tip_amd.optim import Adam
opt = (torch.optim.Adam(m.parameters(), lr=0.01, capturable=True, fused=True) if os.environ.get('TIPK_TORCH_ADAM')
       else Adam(m.parameters(), lr=0.01))
step = GraphedTrainStep(m, opt)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); print('%.3f ms/epoch' % ((time.perf_counter() - t0) / 20 * 1e3))
