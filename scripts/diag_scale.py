"""GPU diagnostic at scale: library results vs a torch fp32 matmul top-k of the same rounded
inputs (debug reference only — not the oracle): recall, duplicates, fallbacks."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
m = importlib.import_module("codegraph-rust_amd")


def run(n, nq, d=768, k=10, chunk=125_000):
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(n + nq)
    ix = m.HipKnnIndex(d, dtype="bf16")
    ix.reserve(n)
    parts = []
    for lo in range(0, n, chunk):
        x = torch.nn.functional.normalize(torch.randn((min(chunk, n - lo), d), generator=g, device=dev), dim=1)
        ix.add(x)
        parts.append(x.to(torch.bfloat16))
    rows = torch.cat(parts)
    q = torch.nn.functional.normalize(torch.randn((nq, d), generator=g, device=dev), dim=1)
    idx, sc = ix.search(q, k)
    torch.cuda.synchronize()
    qb = q.to(torch.bfloat16).float()
    qb = qb / qb.norm(dim=1, keepdim=True)
    best_s = torch.full((nq, k), -1e30, device=dev)
    best_i = torch.zeros((nq, k), dtype=torch.int64, device=dev)
    for lo in range(0, n, 200_000):
        r = rows[lo:lo + 200_000].float()
        s = qb @ (r / r.norm(dim=1, keepdim=True)).T
        ts, ti = s.topk(k, dim=1)
        cs = torch.cat([best_s, ts], 1)
        ci = torch.cat([best_i, ti + lo], 1)
        o = cs.argsort(dim=1, descending=True)[:, :k]
        best_s, best_i = cs.gather(1, o), ci.gather(1, o)
    rec = torch.tensor([len(set(a.tolist()) & set(b.tolist())) for a, b in zip(idx.cpu(), best_i.cpu())]).float() / k
    dup = torch.tensor([k - len(set(a.tolist())) for a in idx.cpu()])
    st = ix.stats()
    bad_q = (rec < 1).nonzero().flatten()
    print(f"[scale] n={n} nq={nq}: recall={rec.mean():.4f} bad_queries={len(bad_q)} dup_total={int(dup.sum())} "
          f"fallback={st['fallback_queries']} maxerr={st['max_observed_err']:.2e} coarse_ms={st['last_coarse_ms']:.3f}")
    if len(bad_q):
        b = int(bad_q[0])
        print("   bad q hist by q//256:", torch.bincount(bad_q // 256, minlength=(nq + 255) // 256).tolist())
        print("   bad q hist by q%32 :", torch.bincount(bad_q % 32, minlength=32).tolist())
        print(f"   q={b} lib idx {idx[b].tolist()}\n        lib sc  {[round(float(v), 4) for v in sc[b]]}\n"
              f"        ref idx {best_i[b].tolist()}\n        ref sc  {[round(float(v), 4) for v in best_s[b]]}")
        missing = sorted(set(best_i[b].tolist()) - set(idx[b].tolist()))
        print("   missing rows:", missing, " tiles:", [r // 256 for r in missing])
    ix.close()


if __name__ == "__main__":
    for n, nq in [(70_000, 300), (70_000, 1024), (200_000, 512), (200_000, 1024), (500_000, 1024), (1_000_000, 256),
                  (1_000_000, 1024)]:
        run(n, nq)
