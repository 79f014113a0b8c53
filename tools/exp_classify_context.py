import sys, time, torch
sys.path.insert(0, '.')
from xenomapper_amd import _ffi, synth
dev = torch.device('cuda:0')
n_pairs = 50_000_000; n = 2*n_pairs
ctx = _ffi.Context(0)
cols = synth.score_columns_torch(n_pairs, 2002, dev)
code = torch.empty(n+16, dtype=torch.uint8, device=dev)
idx = torch.empty(n, dtype=torch.int32, device=dev)
off = torch.zeros(8, dtype=torch.int64, device=dev); counts = torch.zeros(64, dtype=torch.int64, device=dev)
def cls(): ctx.classify_dev(1, cols['as1'], cols['xs1'], cols['as2'], cols['xs2'], cols['unit_bits'], _ffi.ABSENT, code)
def cmp(): ctx.compact_dev(1, code[:n], idx, off, counts)
def run(fn, name, k=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); ctx.timing_enable(True); ctx.timing_reset()
    t=time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); el=time.perf_counter()-t
    tm = ctx.timing_read(); ctx.timing_enable(False)
    print(name, 'wall/iter %.1f us' % (el/k*1e6), {k2: round(v['ms']/max(1,v['launches'])*1e3,1) for k2,v in tm.items() if v['launches']})
run(cls, 'classify only')
run(lambda: (cls(), cmp()), 'classify+compact')
run(cmp, 'compact only')
# separate allocations via hipMalloc-like: fresh tensors each 400MB
cols2 = {k: v.clone() for k,v in cols.items()}
def cls2(): ctx.classify_dev(1, cols2['as1'], cols2['xs1'], cols2['as2'], cols2['xs2'], cols2['unit_bits'], _ffi.ABSENT, code)
run(cls2, 'classify only (cloned cols)')
print([hex(v.data_ptr()) for v in cols.values()], hex(code.data_ptr()))

# ---- does the spacing of the four column bases matter (HBM channel aliasing)? ----
big = torch.empty(4 * 400_000_000 + (64 << 20), dtype=torch.uint8, device=dev)
def carve(gap):
    out = {}
    for c, k in enumerate(('as1', 'xs1', 'as2', 'xs2')):
        o = c * (400_000_000 + gap)
        o = (o + 255) // 256 * 256
        v = big[o:o + 400_000_000].view(torch.int32)
        v.copy_(cols[k])
        out[k] = v
    return out
for gap in (0, 2_653_184, 4096, 1 << 16, (1 << 20) + 4096, 3 << 20, 12345 * 256):
    cc = carve(gap)
    def cls3(): ctx.classify_dev(1, cc['as1'], cc['xs1'], cc['as2'], cc['xs2'], cols['unit_bits'], _ffi.ABSENT, code)
    run(cls3, 'classify gap=%d (stride %d)' % (gap, 400_000_000 + gap))
