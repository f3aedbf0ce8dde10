import torch, time
dev = torch.device("cuda:0")
s = torch.cuda.Stream(dev)
def t(f, n=5):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); x = f(); ts.append(1e6 * (time.perf_counter() - t0)); del x
    return [round(v) for v in ts]
with torch.cuda.stream(s):
    for gb in (0.13, 1.4, 3.6):
        n = int(gb * 2**30)
        print(gb, "GiB empty us", t(lambda: torch.empty(n, dtype=torch.uint8, device=dev)))
        print(gb, "GiB zeros us", t(lambda: torch.zeros(n, dtype=torch.uint8, device=dev)))
        torch.cuda.synchronize()
        print(gb, "GiB zeros after sync us", t(lambda: torch.zeros(n, dtype=torch.uint8, device=dev)))
    # with a long kernel queue in front
    a = torch.empty(int(3.6 * 2**30), dtype=torch.uint8, device=dev)
    for _ in range(20):
        a.zero_()
    print("zeros behind 20 queued fills us", t(lambda: torch.zeros(int(1.4 * 2**30), dtype=torch.uint8, device=dev)))
    torch.cuda.synchronize()
