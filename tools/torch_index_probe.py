import torch
n = 132710400
perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
pf = perm.to(torch.float32)
for dt in (torch.float32, torch.int32):
    for w in (4, 8):
        x = torch.zeros((n, w), device="cuda", dtype=dt); x[:, 0] = torch.arange(n, device="cuda").to(dt)
        y = x[perm]
        print(dt, w, "fancy bad rows:", int((y[:, 0].to(torch.float32) != pf).sum()))
        del x, y
# chunked gather: index chunks of 2^24 rows
x = torch.zeros((n, 4), device="cuda", dtype=torch.int32); x[:, 0] = torch.arange(n, device="cuda", dtype=torch.int32)
bad = 0
for c in range(0, n, 1 << 24):
    idx = perm[c:c + (1 << 24)]
    bad += int((x[idx][:, 0] != idx.to(torch.int32)).sum())
print("chunked int32x4 bad rows:", bad)
