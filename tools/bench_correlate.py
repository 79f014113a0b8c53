#!/usr/bin/env python3
"""Throughput of the mate-density correlation kernel (xenomappability, 8f-4) on a chromosome-sized track."""
import json, sys
sys.path.insert(0, '.')
import torch
from xenomapper_amd import _ffi
n, m = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 451
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(1)
track = (torch.rand(n, generator=g, device=dev) < 0.6).to(torch.float64)
dens = torch.rand(m, generator=g, device=dev, dtype=torch.float64); dens /= dens.sum()
out = torch.empty(n, dtype=torch.float64, device=dev)
ctx = _ffi.Context(0)
for _ in range(2): ctx.mate_correlate_dev(track, dens, out)
torch.cuda.synchronize()
ctx.timing_enable(True); ctx.timing_reset()
for _ in range(5): ctx.mate_correlate_dev(track, dens, out)
torch.cuda.synchronize()
ms = ctx.timing_read()["correlate"]; ms = ms["ms"] / ms["launches"]
print(json.dumps({"kernel": "mate_correlate_kernel", "n": n, "taps": m, "ms": ms, "Gtaps_per_s": n * m / ms / 1e6,
                  "f64_TFLOPs": 2.0 * n * m / ms / 1e9, "GBps_algorithmic": 16.0 * n / ms / 1e6}))
