"""Experiment: pipeline K2 of batch i under K1 of batch i+1 on two HIP streams (double-buffered category bytes)."""
import sys, time, torch
sys.path.insert(0, '.')
from xenomapper_amd import _ffi, synth
dev = torch.device('cuda:0')
n_pairs = 50_000_000; n = 2*n_pairs
ctx = _ffi.Context(0)
cols = synth.score_columns_torch(n_pairs, 2002, dev)
code = [torch.empty(n+16, dtype=torch.uint8, device=dev) for _ in range(2)]
idx = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(2)]
off = [torch.zeros(8, dtype=torch.int64, device=dev) for _ in range(2)]
counts = [torch.zeros(64, dtype=torch.int64, device=dev) for _ in range(2)]
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
K = 50

def serial():
    for i in range(K):
        ctx.classify_dev(1, cols['as1'], cols['xs1'], cols['as2'], cols['xs2'], cols['unit_bits'], _ffi.ABSENT, code[0])
        ctx.compact_dev(1, code[0][:n], idx[0], off[0], counts[0])

def piped():
    done_k1 = [torch.cuda.Event() for _ in range(K)]
    done_k2 = [torch.cuda.Event() for _ in range(K)]
    for i in range(K):
        b = i & 1
        with torch.cuda.stream(sA):
            if i >= 2:
                sA.wait_event(done_k2[i-2])          # code[b] is free again
            ctx.classify_dev(1, cols['as1'], cols['xs1'], cols['as2'], cols['xs2'], cols['unit_bits'], _ffi.ABSENT, code[b], stream=sA)
            done_k1[i].record(sA)
        with torch.cuda.stream(sB):
            sB.wait_event(done_k1[i])
            ctx.compact_dev(1, code[b][:n], idx[b], off[b], counts[b], stream=sB)
            done_k2[i].record(sB)

for name, fn in (("serial", serial), ("piped", piped), ("serial", serial), ("piped", piped)):
    fn(); torch.cuda.synchronize()
    ctx.timing_enable(True); ctx.timing_reset()
    t = time.perf_counter(); fn(); torch.cuda.synchronize(); el = time.perf_counter() - t
    tm = ctx.timing_read(); ctx.timing_enable(False)
    print(name, 'step %.1f us' % (el/K*1e6), {k2: round(v['ms']/max(1,v['launches'])*1e3,1) for k2,v in tm.items() if v['launches']})
