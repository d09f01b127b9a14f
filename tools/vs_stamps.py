"""Per-wave phase stamps of the last k_vs_sample launch: where a launch's cycles
go.  Needs the diagnostic build of the library (the stamps cost 3 us per
launch and are compiled out otherwise):
    make -C distributions_amd/csrc stamps
    DIST_VS_STAMPS=/tmp/st.bin python tools/vs_stamps.py run [rows per sub-sweep]   # C2 probe
    python tools/vs_stamps.py /tmp/st.bin
    make -C distributions_amd/csrc          # back to the product build"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def report(path):
    a = np.fromfile(path, dtype=np.uint64).reshape(-1, 6)
    nonzero = a[:, 4] != 0
    a = a[nonzero]
    st = a[:, :5].astype(np.int64)
    hw = a[:, 5]
    simd = ((hw >> 4) & 3).astype(np.int64)
    cu = ((hw >> 8) & 15).astype(np.int64)
    sh = ((hw >> 12) & 1).astype(np.int64)
    se = ((hw >> 13) & 7).astype(np.int64)
    xcc = ((hw >> 32) & 15).astype(np.int64)
    chunks = (hw >> 40).astype(np.int64)   # chunks of 32 links the wave ran
    # s_memtime bases differ between CUs: times relative to the first wave of
    # the LAST launch on the same CU (a wave with nothing to do leaves the
    # stamps of an older launch behind)
    cukey = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    keep = np.zeros(len(st), bool)
    for c in np.unique(cukey):
        m = cukey == c
        recent = m & (st[:, 0] >= st[m, 0].max() - 100_000)
        st[recent] -= st[recent, 0].min()
        keep |= recent
    keep_rows = np.zeros(len(nonzero), bool)
    keep_rows[np.nonzero(nonzero)[0][keep]] = True
    st, hw, simd, cu, sh, se, xcc, chunks = (
        v[keep] for v in (st, hw, simd, cu, sh, se, xcc, chunks))
    key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    span = st[:, 4].max()
    print("%d waves on %d XCDs, launch spans %d cycles" % (
        len(st), len(np.unique(xcc)), span))
    names = ["setup (loads, own scores)", "vector A: sum + scan",
             "vector B: sum + scan", "write back"]
    for i, n in enumerate(names):
        d = st[:, i + 1] - st[:, i]
        print("  %-28s mean %8.0f  median %8.0f  p95 %8.0f  max %8.0f" % (
            n, d.mean(), np.median(d), np.percentile(d, 95), d.max()))
    life = st[:, 4] - st[:, 0]
    print("  %-28s mean %8.0f  median %8.0f  p95 %8.0f  max %8.0f" % (
        "wave lifetime", life.mean(), np.median(life),
        np.percentile(life, 95), life.max()))
    print("  wave start: median %d  p95 %d  max %d" % (
        np.median(st[:, 0]), np.percentile(st[:, 0], 95), st[:, 0].max()))
    print("  wave end:   p5 %d  median %d  p95 %d  max %d" % (
        np.percentile(st[:, 4], 5), np.median(st[:, 4]),
        np.percentile(st[:, 4], 95), st[:, 4].max()))
    keys, inv = np.unique(key, return_inverse=True)
    per = np.bincount(inv)
    end = np.zeros(len(keys), np.int64)
    np.maximum.at(end, inv, st[:, 4])
    work = np.bincount(inv, weights=life)
    print("  %d SIMDs seen; waves per SIMD: min %d median %d max %d" % (
        len(keys), per.min(), np.median(per), per.max()))
    print("  SIMD finish time: min %d  p5 %d  median %d  p95 %d  max %d" % (
        end.min(), np.percentile(end, 5), np.median(end),
        np.percentile(end, 95), end.max()))
    print("  sum of wave lifetimes per SIMD / span: min %.2f median %.2f max "
          "%.2f" % (work.min() / span, np.median(work) / span,
                    work.max() / span))
    if chunks.any():
        # is the span the heaviest SIMD's work?
        load = np.bincount(inv, weights=chunks)
        cukeys, cuinv = np.unique(keys // 4, return_inverse=True)
        cuload = np.bincount(cuinv, weights=load)
        cuend = np.zeros(len(cukeys), np.int64)
        np.maximum.at(cuend, cuinv, end)
        print("  chunks per wave: mean %.1f min %d max %d; per SIMD: mean %.0f "
              "min %d max %d; per CU: mean %.0f min %d max %d" % (
                  chunks.mean(), chunks.min(), chunks.max(), load.mean(),
                  load.min(), load.max(), cuload.mean(), cuload.min(),
                  cuload.max()))
        print("  corr(SIMD finish, SIMD chunks) %.2f  corr(CU finish, CU "
              "chunks) %.2f  corr(SIMD finish, its CU's chunks) %.2f" % (
                  np.corrcoef(end, load)[0, 1],
                  np.corrcoef(cuend, cuload)[0, 1],
                  np.corrcoef(end, cuload[cuinv])[0, 1]))
        slope = np.polyfit(cuload, cuend, 1)
        print("  CU finish ~ %.0f + %.1f cycles per chunk of that CU" % (
            slope[1], slope[0]))
    if chunks.any():
        # cycles of the two vector phases per chunk, by the tile's chunk count
        dur = (st[:, 3] - st[:, 1]).astype(np.float64)
        per = dur / np.maximum(chunks, 1)
        print("  vector phases per chunk: p10 %.0f median %.0f p90 %.0f p99 "
              "%.0f cycles; waves above 2x the median: %d (%.1f %% of the "
              "vector-phase time)" % (
                  np.percentile(per, 10), np.median(per),
                  np.percentile(per, 90), np.percentile(per, 99),
                  int((per > 2 * np.median(per)).sum()),
                  100.0 * dur[per > 2 * np.median(per)].sum() / dur.sum()))
    # which CU a workgroup (8 consecutive tile ids) landed on: the dispatch
    # pattern a static balance of the tile list would have to know
    ids = np.nonzero(keep_rows)[0]
    wg = ids // 8
    cu_of = key // 4
    first = {}
    for w, c in zip(wg.tolist(), cu_of.tolist()):
        first.setdefault(w, c)
    order = [first[w] for w in sorted(first)][:48]
    print("  CU (xcc*256+se*32+sh*16+cu) of workgroups 0..47:", order)
    by_cu = {}
    for w in sorted(first):
        by_cu.setdefault(first[w], []).append(w)
    some = sorted(by_cu)[:6]
    print("  workgroups per CU (first 6 CUs):", [by_cu[c] for c in some])
    grid = np.linspace(0, span, 11)
    alive = [(int(((st[:, 0] <= t) & (st[:, 4] > t)).sum())) for t in grid]
    print("  waves alive at 0%..100% of the span:", alive)
    insetup = [(int(((st[:, 0] <= t) & (st[:, 1] > t)).sum())) for t in grid]
    print("  of them still in setup:             ", insetup)
    # the slowest waves: which phase makes them slow
    slow = np.argsort(-st[:, 4])[:len(st) // 50]
    print("  slowest 2%% of waves: start %.0f setup %.0f A %.0f B %.0f" % (
        st[slow, 0].mean(), (st[slow, 1] - st[slow, 0]).mean(),
        (st[slow, 2] - st[slow, 1]).mean(), (st[slow, 3] - st[slow, 2]).mean()))
    cs = np.corrcoef(np.stack([life, st[:, 1] - st[:, 0],
                               st[:, 2] - st[:, 1], st[:, 0]]))
    print("  corr(lifetime, setup) %.2f  corr(lifetime, A) %.2f  "
          "corr(lifetime, start) %.2f" % (cs[0, 1], cs[0, 2], cs[0, 3]))


def run(batch=1_000_000):
    sys.path.insert(0, ROOT)
    import torch
    from distributions_amd import engine
    n, k, dim = 10_000_000, 1024, 256
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    if os.environ.get("DIST_STAMPS_VALUES") == "zipf":   # bench.py --values zipf
        w = 1.0 / torch.arange(1, dim + 1, device=dev,
                               dtype=torch.float64) ** 1.1
        cdf = torch.cumsum(w / w.sum(), 0).to(torch.float32)
        u = torch.rand((n,), generator=gen, device=dev)
        col = torch.searchsorted(cdf, u).clamp_(max=dim - 1).to(torch.int32)
    else:
        col = torch.randint(0, dim, (n,), generator=gen, device=dev,
                            dtype=torch.int32)
    assign = torch.arange(n, device=dev, dtype=torch.int64).remainder(k).to(
        torch.int32)
    g = engine.Gibbs(1.0, 0.2, [engine.dd_shared([0.5] * dim)])
    g.load_rows_torch([col], assign, k, 1)
    # (small sub-sweeps go through k_vs_narrow; its four phases are the rows'
    # set-up, the vectors into LDS, the recurrences, the write back)
    rows = n if batch >= 1_000_000 else 40 * batch
    for s in range(3):
        g.sweep(0, rows, batch, 5, draw_base=s * n)
    report(os.environ["DIST_VS_STAMPS"])


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(*[int(a) for a in sys.argv[2:3]])
    else:
        report(sys.argv[1])
