"""Which parameter gradients differ between the 2-rank SyncBN run and the single-process run (tests/test_hip_ddp.py)."""
import os, socket, sys, tempfile, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_hip_ddp as T

if __name__ == "__main__":
    import torch.multiprocessing as mp
    size = 128
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    out = os.path.join(tempfile.mkdtemp(), "ddp.pt")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=T._worker, args=(r, 2, port, size, out)) for r in range(2)]
    [p.start() for p in procs]; [p.join(300) for p in procs]
    got = torch.load(out)
    net, loss = T._build(5)
    x, tg = T._data(size)
    T._run(net, loss, x, tg, size)
    off = 0
    rows = []
    for n, p in net.named_parameters():
        k = p.numel()
        a, b = got["g"][off:off + k].double(), p.grad.flatten().cpu().double()
        rows.append((((a - b).norm() / (b.norm() + 1e-30)).item(), n, b.norm().item()))
        off += k
    for r, n, bn in rows:
        if r > 5e-3: print(f"{r:9.3e} {bn:9.3e} {n}")
    print("max", max(rows)[:2], "n>5e-3:", sum(1 for r in rows if r[0] > 5e-3), "of", len(rows))
