"""Diagnostic: per-tile s_memtime stamps of conv3x3_halo_kernel (GMK dev variant 99 writes them into the GN-statistics buffer)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from generative_models_amd import ops
from generative_models_amd._lib import lib
T = torch.bfloat16
B, C, S = 1024, 128, 28
ops.GN_STATS = True
VAR = 0x100 + (int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for nsrc in (1, 2):
    srcs = [torch.randn((B, S, S, C), device="cuda").to(T) for _ in range(nsrc)]
    cin = nsrc * C
    w = torch.randn((C, cin, 3, 3), device="cuda") / (cin * 9) ** 0.5
    wf = torch.empty(w.numel(), device="cuda", dtype=T); wd = torch.empty_like(wf)
    ops.pack_conv_weight(w, wf, wd)
    for _ in range(10):
        ops.conv_igemm(srcs, wf, C, 3, ops.NORMAL, (S, S))
    lib.gmk_set_dev_variant(VAR)
    r = 256 // S
    nt = (B * S + r - 1) // r
    part = torch.zeros(nt * 8 * 2 * (C // 4) * 2, device="cuda", dtype=torch.float32)
    out = torch.empty((B, S, S, C), device="cuda", dtype=T)
    EPI = os.environ.get("EPI", "plain")
    bias = torch.randn(C, device="cuda") if EPI != "plain" else None
    emb = torch.randn(B, C, device="cuda") if "emb" in EPI else None
    res = torch.randn(B, S, S, C, device="cuda").to(T) if "res" in EPI else None
    ops.check(lib.gmk_conv_igemm(ops._p(srcs[0]), ops._p(srcs[1]) if nsrc > 1 else None, C, C if nsrc > 1 else 0, B, S, S, S, S, 3,
                                 ops.NORMAL, ops._p(wf), C, 0, C, ops._p(bias), ops._p(emb), C if emb is not None else 0, ops._p(res),
                                 ops._p(out), C, ops._p(part), part.numel() * 4, ops._DT[T], ops._s()), "conv")
    torch.cuda.synchronize()
    lib.gmk_set_dev_variant(0)
    st = part.view(torch.int64)[: 256 * 16 * 4].view(256, 16, 4).cpu().double()
    ntile = 12
    st = st[:, :ntile]
    first = (st[:, :, 1] - st[:, :, 0])          # tile start -> end of phase 0
    rest = (st[:, :, 2] - st[:, :, 1])           # remaining phases
    epi = (st[:, :, 3] - st[:, :, 2])
    gap = (st[:, 1:, 0] - st[:, :-1, 3])
    whole = (st[:, 1:, 0] - st[:, :-1, 0])
    med = lambda t: float(t.median())
    print(f"{os.environ.get('EPI', 'plain'):9s} variant {VAR & 255} cin={cin}: per tile (cycles, median over 256 WGs x {ntile} tiles): phase0 {med(first):.0f}  other phases {med(rest):.0f}  "
          f"epilogue {med(epi):.0f}  gap {med(gap):.0f}  tile-to-tile {med(whole):.0f}")
    print("  tile 0 of each WG: phase0 %.0f (prologue excluded), kernel span %.0f cycles" %
          (med(first[:, 0]), float(st[:, :, 3].max() - st[:, 0, 0].min())))
    print("  phase0 by tile index:", [int(first[:, i].median()) for i in range(ntile)])
    print("  epilogue by tile index:", [int(epi[:, i].median()) for i in range(ntile)])
