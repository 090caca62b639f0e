"""Bank-conflict model of `tipk_stream_gather`'s row reads for a StreamPlan (CPU, no GPU needed): per ds_read_b128
wave-instruction (band, step, id position) the 16-lane groups of MI355X_MICROARCH.md (LDS) are replayed on the plan's ids;
extra cycles = (largest number of DIFFERENT rows on one 16-byte bank unit of a group) - 1.
    python tools/lds_conflict_sim.py          # BioSNAP: transposed pass d = 32 / d = 16, pair cells
"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GROUP0 = (0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27)


def conflict_share(plan, chunk=4096):
    """-> (extra cycles / all LDS cycles of the row reads, reads simulated)."""
    L, S, P = plan.lanes, 64 // plan.lanes, plan.piece
    row_bytes = plan.row_bytes
    ids = (plan.ids.to(torch.int64) // plan.idx_unit).view(-1, P, S, 8)          # table rows
    cells = plan.cells.to(torch.int64) & 0xffffffff
    length = (cells >> 24) & 15                                                   # [bands, S]
    lane = torch.arange(64)
    slot_of = lane // L
    piece_of = lane % L
    grp = torch.tensor([(l // 32) * 2 + (0 if (l % 32) in GROUP0 else 1) for l in range(64)])
    base_cyc = extra = 0
    nb = ids.shape[0]
    for b0 in range(0, nb, chunk):
        idc = ids[b0:b0 + chunk]                                                  # [B, P, S, 8]
        ln = length[b0:b0 + chunk]
        B = idc.shape[0]
        act = (torch.arange(P).view(1, P, 1) < ln.view(B, 1, S))                  # [B, P, S]
        rows = idc[:, :, slot_of, :].permute(0, 1, 3, 2)                          # [B, P, 8, 64] row read by every lane
        active = act[:, :, slot_of].unsqueeze(2).expand(B, P, 8, 64)
        unit = (rows * (row_bytes // 16) + piece_of.view(1, 1, 1, 64)) % 16       # 16-byte bank unit of the lane's piece
        for g in range(4):
            sel = grp == g
            r, u, a = rows[..., sel], unit[..., sel], active[..., sel]             # [B, P, 8, 16]
            any_act = a.any(-1)
            # distinct rows per unit: count lanes whose (unit,row) pair is the first of its kind
            key = u * 100000 + r
            key = torch.where(a, key, torch.full_like(key, -1))
            srt = torch.sort(key, dim=-1).values
            first = torch.ones_like(srt, dtype=torch.bool)
            first[..., 1:] = srt[..., 1:] != srt[..., :-1]
            first &= srt >= 0
            uu = torch.where(srt >= 0, srt // 100000, torch.zeros_like(srt))
            mult = torch.zeros(srt.shape[:-1] + (16,), dtype=torch.int64).scatter_add_(-1, uu, first.long())
            worst = mult.max(-1).values
            base_cyc += int(any_act.sum())
            extra += int((worst - 1).clamp(min=0)[any_act].sum())
    return extra / max(1, base_cyc + extra), base_cyc


if __name__ == '__main__':
    from tip_amd.data import build_data_dict
    from tip_amd.plan import build_stream_plan, build_stream_plan_rows
    dd = build_data_dict()
    src, dst = dd['dd_train_idx']
    rel = dd['dd_train_et']
    N, R = dd['n_drug'], dd['n_dd_et']
    for d, lanes in ((32, 8), (16, 4)):
        sp = build_stream_plan(src, dst, rel, N, R, 256, lanes, 4, compact=True)
        print('transposed pass d = %d (L = %d): conflict share %.3f' % ((d, lanes) + conflict_share(sp)[:1]))
    keep = src <= dst
    pp = build_stream_plan_rows(src[keep] * N + dst[keep], rel[keep], N * N, R, 256, 8, 4)
    print('pair cells (L = 8): conflict share %.3f' % conflict_share(pp)[0])
