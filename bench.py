#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: D-D edges aggregated / s over encoder forward + backward.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload biosnap|biosnap963|synthetic]

A "step" is one pass of the hot path over the full graph: `z = FMEncoder.forward(...)` (P-P GCN x2,
P->D mean, two R-GCN layers) followed by its backward with a fixed upstream gradient g ~ N(0,1)
(SURVEY.md section 8(d)).  Inputs (graph plans, weights, g) are resident in HBM before the timed
region; graph preprocessing is timed separately (`preprocess_s`).  value = E * K / t with E the
number of directed D-D train edges, t the max over ranks of the barrier-bracketed wall time.

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): the D-D relations are sharded
over the ranks (tip_amd/dist.py), partial aggregates and the replicated-parameter gradients are
all-reduced; total work is fixed, so "scaling" is "strong".

Extra objects on the JSON line: `roofline` (dominant kernel, HIP-event timed on its launch
stream inside the timed region) and `cpu_baseline` (the oracle's CPU port on the host cores, rank 0,
N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='biosnap', choices=['biosnap', 'biosnap963', 'synthetic', 'synthetic-small'])
    ap.add_argument('--mod', default='cat', choices=['cat', 'add'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='budget of the CPU baseline leg')
    ap.add_argument('--chunk', type=int, default=None, help='gather plan chunk (edges per work item)')
    ap.add_argument('--launch', default=None, choices=['graph', 'eager'],
                    help='graph: the step is captured once into a hipGraph and replayed (default on 1 GPU); '
                         'eager: one ctypes launch per kernel (default with collectives in the step)')
    return ap.parse_args()


def make_workload(args):
    """-> (data dict on CPU, dims dict, description)."""
    from tip_amd.data import build_data_dict, synthetic_data_dict
    if args.workload.startswith('biosnap'):
        dd = build_data_dict(min_pairs=500 if args.workload == 'biosnap963' else None)
        dims = dict(prot_drug_dim=16, n_embed=48) if args.mod == 'cat' else dict(prot_drug_dim=64, n_embed=64)
        dims.update(n_hid1=32, n_hid2=16, num_base=32)
        name = 'TIP-%s full encoder, BioSNAP (645 drugs, 19081 proteins, R=%d), 1xMI355X config' % (
            args.mod, dd['n_dd_et'])
        return dd, dims, name
    if args.workload == 'synthetic':
        dd = synthetic_data_dict(with_protein_graph=True)
        dims = dict(prot_drug_dim=64, n_embed=64, n_hid1=128, n_hid2=128, num_base=32)
    else:
        dd = synthetic_data_dict(n_drug=2000, n_rel=200, n_edges=2_000_000, with_protein_graph=True)
        dims = dict(prot_drug_dim=64, n_embed=64, n_hid1=128, n_hid2=128, num_base=32)
    name = 'synthetic scaled graph (%d drugs, R=%d, E=%d, dim=128)' % (dd['n_drug'], dd['n_dd_et'],
                                                                    dd['dd_train_idx'].shape[1])
    return dd, dims, name


def cpu_baseline(dd, dims, mod, budget_s):
    """The oracle (CPU port of the same algorithm: transform-then-gather with explicit backward)
    on this host's cores, plus the reference-shaped op sequence (PyG-CPU path: lift E x in,
    per-relation slice+mm, cat, scatter-mean, autograd backward) on a bounded relation sample."""
    from oracle import tip_oracle as O
    E = dd['dd_train_idx'].shape[1]
    R = dd['n_dd_et']
    p = O.init_params(dd['n_drug'], dd['n_prot'], R, mod=mod, seed=1111, prot_drug_dim=dims['prot_drug_dim'],
                      n_embed=dims['n_embed'], n_hid1=dims['n_hid1'], n_hid2=dims['n_hid2'],
                      num_base=dims['num_base'])
    up = torch.randn(dd['n_drug'], dims['n_hid2'], generator=torch.Generator().manual_seed(0))
    # torch's default (= all hardware threads) oversubscribes these small ops on big hosts
    threads = max(1, min(16, os.cpu_count() or 1))
    torch.set_num_threads(threads)

    def step():
        z, saved = O.fm_encoder_fwd(p, dd, mod)
        O.fm_encoder_bwd(up, p, dd, saved, mod)
    step()                                                    # warm-up (page-in, thread pool)
    t0 = time.perf_counter()
    n = 0
    while True:
        step()
        n += 1
        if time.perf_counter() - t0 > budget_s * 0.6 or n >= 20:
            break
    dt = (time.perf_counter() - t0) / n
    out = {'value': E / dt, 'unit': 'edges/s', 'cores': threads, 'kind': 'port',
           'sample': 'full workload: %d oracle encoder fwd+bwd passes over all %d edges (%.2f s each)' % (n, E, dt)}

    # reference-shaped flavour on the first relations (bounded: cost grows ~ R * E)
    try:
        r_s = min(R, 24)
        e_s = int(dd['dd_train_range'][r_s - 1, 1])
        ei = dd['dd_train_idx'][:, :e_s]
        rg = dd['dd_train_range'][:r_s]
        x = torch.randn(dd['n_drug'], 64).requires_grad_(True)
        prm = [p['rgcn1.basis'][:, :64].clone().requires_grad_(True), p['rgcn1.att'][:r_s].clone().requires_grad_(True),
               p['rgcn1.root'][:64].clone().requires_grad_(True)]
        g = torch.randn(dd['n_drug'], prm[0].shape[2])
        t0 = time.perf_counter()
        y = O.rgcn_fwd_reference_shaped(x, ei, rg, *prm)
        y.backward(g)
        dt_ref = time.perf_counter() - t0
        out['reference_shaped'] = {'value': e_s / dt_ref, 'unit': 'edges/s (one R-GCN layer fwd+bwd)', 'cores': threads,
                                   'sample': 'first %d relations, %d edges, PyG op sequence under autograd, %.2f s'
                                             % (r_s, e_s, dt_ref)}
    except Exception as exc:                                   # never let the side leg kill the bench
        out['reference_shaped'] = {'error': repr(exc)}
    return out


def pmc_traffic(enc, label, d_row):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary
    (profiles/*_pmc_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
    command, corrected as MI355X_MICROARCH.md prescribes); None if no summary matches the launch."""
    import glob
    try:
        graph = {'dd': enc.rgcn1 if d_row == enc.rgcn1.out_channels else enc.rgcn2}['dd']._cache.value \
            if '[dd.' in label else None
        if graph is None:
            return None
        if label.startswith('rel_gather'):
            bwd = '.bwd' in label
            rp = graph.rl_bwd if bwd else graph.rl_fwd
            split = 1 if (bwd or d_row <= 16) else 2          # column blocks (tipk_rel_gather.hip)
            key_prefix = 'rel_gather_kernel<%d, %s' % (d_row // split // 4, 'true' if bwd else 'false')
            grid = rp.n_wg * split * 1024
        else:
            plan = graph.fwd if '.fwd' in label else graph.bwd
            lanes = 1
            while lanes < d_row // 4:
                lanes *= 2
            waves = -(-plan.items.shape[0] // (64 // lanes))
            grid = -(-waves // 4) * 256
            key_prefix = 'gather_sum_kernel<4, %d, false>' % lanes
        for fn in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json')), reverse=True):
            for name, k in json.load(open(fn))['kernels'].items():
                if name.startswith(key_prefix) and name.endswith('grid=%d' % grid):
                    return k['hbm_bytes_per_launch']
    except Exception:
        pass
    return None


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('--gpus %d needs torch.distributed.run with %d ranks' % (args.gpus, args.gpus))
        args.gpus = world
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the product path has no CPU fallback)'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    import torch.distributed as dist
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner to the C-level stdout (seen
    # after the JSON line, when its stdio buffer is flushed): while collectives may run, fd 1 points at
    # stderr; it is restored -- after flushing C stdio -- just before the result is printed.
    stdout_fd = None
    if world > 1 or os.environ.get('TIPK_FORCE_SHARD'):
        sys.stdout.flush()
        stdout_fd = os.dup(1)
        os.dup2(2, 1)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from tip_amd import ops
    from tip_amd.data import Data
    from tip_amd.layers import FMEncoder
    if args.chunk:
        os.environ['TIPK_CHUNK'] = str(args.chunk)
    launch = args.launch or 'graph'

    dd, dims, wl_name = make_workload(args)
    E = int(dd['dd_train_idx'].shape[1])
    R = dd['n_dd_et']
    torch.manual_seed(1111)
    enc = FMEncoder(dev, dd['n_drug_feat'], R, dd['n_prot'], dd['n_prot'], dd['n_drug'], mod=args.mod, **dims).to(dev)
    sharded = world > 1 or bool(os.environ.get('TIPK_FORCE_SHARD'))       # (the env switch exercises the
    if sharded:                                                            # collective path on one rank)
        from tip_amd.dist import shard_encoder
        if not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29533')
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        for prm in enc.parameters():                           # identical replicas
            dist.broadcast(prm.data, 0)
        shard_encoder(enc, dd['dd_train_range'], rank, world)
    d = Data.from_dict({k: v for k, v in dd.items() if k != 'dd_edge_index'}).to(dev)
    g_up = torch.randn(dd['n_drug'], dims['n_hid2'], generator=torch.Generator().manual_seed(0)).to(dev)

    def step():
        for prm in enc.parameters():
            prm.grad = None
        z = enc(d.d_feat, d.dd_train_idx, d.dd_train_et, d.dd_train_range, d.d_norm, d.p_feat,
                d.pp_train_indices, d.dp_edge_index, d.dp_range_list)
        z.backward(g_up)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    t0 = time.perf_counter()
    step()                                                     # builds + caches all gather plans
    torch.cuda.synchronize()
    preprocess_s = time.perf_counter() - t0
    run = step
    if launch == 'graph':
        # the whole step (about 55 kernels, plus the RCCL all-reduces when sharded) becomes one
        # hipGraph: replay removes the per-launch host cost, which is larger than the kernels
        # themselves at BioSNAP scale.  If capture fails (e.g. a collective that cannot be captured on
        # this RCCL build) every rank falls back to eager launches together.
        ok = torch.ones(1, device=dev)
        graph = None
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            # with a process group alive the RCCL watchdog thread makes HIP calls of its own: only this
            # thread's calls may invalidate the capture
            mode = {'capture_error_mode': 'thread_local'} if dist.is_initialized() else {}
            with torch.cuda.graph(graph, **mode):
                step()
        except Exception as exc:                               # noqa: BLE001
            sys.stderr.write('graph capture failed on rank %d (%r): eager launches\n' % (rank, exc))
            ok.zero_()
            torch.cuda.synchronize()
        if dist.is_initialized() and world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) > 0:
            run = graph.replay
        else:
            launch = 'eager'
    for _ in range(args.warmup):
        run()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    fence()
    elapsed = time.perf_counter() - t0
    # per-kernel durations: HIP events on the launch stream around every launch of an eager pass of
    # the same steps, same process (events cannot be read back from inside a replayed graph)
    ops.timing_start()
    for _ in range(max(3, min(args.steps, 10))):
        step()
    fence()
    kern = ops.timing_stop()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        # dominant kernel = the sparse-aggregation launch with the largest share of device time
        gs = {k: v for k, v in kern.items() if k.startswith('gather_sum') or k.startswith('rel_gather')}
        dom = max(gs, key=lambda k: gs[k][0] * gs[k][1])
        d_row = int(dom.split('d=')[1].rstrip(']'))
        n_launch_edges = E if '[dd.' in dom else None
        if n_launch_edges is None:                             # P-P / P->D launches (not expected to dominate)
            n_launch_edges = int(dd['pp_train_indices'].shape[1]) + dd['n_prot']
        if world > 1:
            n_launch_edges = n_launch_edges // world           # rank 0's share (balanced by edges)
        # one id + one d-wide fp32 row per edge and pass (ids are 4 B in the generic
        # plans, 2 B in the relation-local ones; the figure keeps 4 B so runs stay comparable)
        alg_bytes = n_launch_edges * (4 + 4 * d_row)
        achieved = alg_bytes / (gs[dom][1] * 1e-3) / 1e9
        traffic = pmc_traffic(enc, dom, d_row)
        lds_note, lds_roof = None, None
        if dom.startswith('rel_gather'):
            lds_note = ('rows are gathered from LDS (relation-local kernel): algorithmic GB/s exceeds the HBM peak by '
                        'design; HBM only carries the ids and one coalesced read of Y (fwd) / write of dY (bwd) -- see '
                        'traffic; the bound that applies is lds_roofline')
            # the kernel's real ceiling: every gathered row is a ds_read_b128 stream out of the CU's LDS
            # (ds_read_b128: 256 B/clk/CU x 256 CUs x 2.4 GHz, MI355X_MICROARCH "LDS"); same algorithmic bytes
            lds_peak = 256 * 256 * 2.4e9 / 1e9
            lds_roof = {'bound': 'lds', 'achieved': achieved, 'peak': lds_peak, 'unit': 'GB/s', 'frac': achieved / lds_peak}
        out = {
            'metric': 'D-D edges aggregated/sec (encoder fwd+bwd)',
            'value': E * args.steps / elapsed, 'unit': 'edges/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'BioSNAP graph (bundled), random-init weights, fixed N(0,1) upstream gradient'
            if args.workload.startswith('biosnap') else 'synthetic',
            'config': {'workload': wl_name, 'mod': args.mod, 'directed_dd_edges': E, 'relations': R,
                       'parallelism': 'relation-sharded x%d' % world if world > 1 else 'single GPU',
                       'launch': 'hipGraph replay of the captured step' if launch == 'graph'
                       else 'eager (one ctypes call per kernel)'},
            'roofline': {'bound': 'hbm', 'kernel': dom, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'launch_ms': gs[dom][1], 'algorithmic_bytes_per_launch': alg_bytes, 'note': lds_note,
                         'lds_roofline': lds_roof},
            'kernels_ms': {k: {'launches': v[0], 'mean_ms': round(v[1], 5)}
                           for k, v in sorted(kern.items(), key=lambda kv: -kv[1][0] * kv[1][1])},
            'preprocess_s': preprocess_s,
            'whole_step_algorithmic_GBps': E * sum(2 * (4 + 4 * dims[k]) for k in ('n_hid1', 'n_hid2')) / (ms * 1e-3) / 1e9,
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(dd, dims, args.mod, args.cpu_seconds)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if stdout_fd is not None:
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)                         # the banner (if any) goes to stderr
        os.dup2(stdout_fd, 1)
        os.close(stdout_fd)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
