"""Probe: how much of the fp8 coarse kernel's time is the epilogue's false positives from mixed per-row scale
exponents? Same shape as c5mini/4 (1M x 768, batch 8192); variant B pins every row's amax so all rows share one exponent."""
import importlib, sys, time, torch
sys.path.insert(0, '.')
m = importlib.import_module('codegraph-rust_amd')
dev = torch.device('cuda', 0)
def run(pin):
    g = torch.Generator(device=dev).manual_seed(1)
    ix = m.HipKnnIndex(768, dtype='fp8'); ix.set_profiling(True)
    for c in range(8):
        x = torch.nn.functional.normalize(torch.randn((125000, 768), generator=g, device=dev), dim=1)
        if pin:
            x = x.clamp(-0.125, 0.125); x[:, 0] = 0.125      # amax = 0.125 exactly for every row -> one exponent
        ix.add(x)
    q = torch.nn.functional.normalize(torch.randn((8192, 768), generator=g, device=dev), dim=1)
    if pin:
        q = q.clamp(-0.125, 0.125); q[:, 0] = 0.125
    for _ in range(3): ix.search(q, 10)
    ms = []
    for _ in range(5):
        ix.search(q, 10); st = ix.stats(); ms.append(st['last_coarse_ms'])
    rows = st['coarse_rows']
    t = sum(ms) / len(ms)
    print('pin' if pin else 'mixed', 'coarse_ms', round(t, 3), 'rows', rows, 'TF', round(2 * 8192 * rows * 768 / t / 1e9, 1), 'fb', st['fallback_queries'], flush=True)
    ix.close()
run(False); run(True)
