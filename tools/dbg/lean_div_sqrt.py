import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from imsim_amd import _abi
lib = _abi.load()
def run(which, xin, n):
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    _abi.check(lib.ims_test_math(which, xin.data_ptr(), out.data_ptr(), n, 0, 0, 0, None))
    torch.cuda.synchronize()
    return out
g = torch.Generator(device="cuda"); g.manual_seed(5)
N = 1 << 26
tot_bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    for lo, hi in ((-1, 1), (-30, 30), (-300, 300), (-480, 480)):
        # random mantissas times random power of two
        def rnd(n):
            m = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) + 1.0
            # fill low mantissa bits too
            m = m + torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 2.0 ** -30
            e = torch.randint(lo, hi + 1, (n,), device="cuda", generator=g).to(torch.float64)
            return m * torch.exp2(e)
        x = rnd(N)
        s = run(7, x, N)
        bad = (s.view(torch.int64) != torch.sqrt(x).view(torch.int64)).sum().item()
        s0 = run(9, x, N)
        bad0 = (s0.view(torch.int64) != torch.sqrt(x).view(torch.int64)).sum().item()
        a = rnd(N) * (torch.randint(0, 2, (N,), device="cuda", generator=g).to(torch.float64) * 2 - 1)
        b = rnd(N) * (torch.randint(0, 2, (N,), device="cuda", generator=g).to(torch.float64) * 2 - 1)
        if hi > 250:   # keep the quotient inside the stated range
            b = b * 0 + rnd(N) ; b = torch.where(b.abs() > 0, b, torch.ones_like(b)); a = a; 
            keep = ((a.abs().log2() - b.abs().log2()).abs() < 500)
        else:
            keep = torch.ones(N, dtype=torch.bool, device="cuda")
        ab = torch.stack([a, b], dim=1).contiguous().view(-1)
        q = run(8, ab, N)
        ref = a / b
        badq = ((q.view(torch.int64) != ref.view(torch.int64)) & keep).sum().item()
        print(f"rep {rep} exp [{lo},{hi}]: sqrt mismatches {bad} (zero-safe {bad0}), div mismatches {badq} of {N}", flush=True)
        tot_bad += bad + bad0 + badq
z = torch.zeros(8, dtype=torch.float64, device="cuda")
print("dsqrt0(0) =", run(9, z, 8).cpu().numpy()[:2], " 0/b:", run(8, torch.tensor([0.0, 3.0, 0.0, -7.0], dtype=torch.float64, device="cuda"), 2).cpu().numpy())
# near-perfect squares and quotients with exact results
k = torch.arange(1, (1 << 22) + 1, dtype=torch.float64, device="cuda")
print("perfect squares bad:", (run(7, k * k, k.numel()) != k).sum().item())
print("TOTAL BAD", tot_bad)
