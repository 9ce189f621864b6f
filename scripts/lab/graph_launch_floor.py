"""Wall time per DEPENDENT tiny kernel inside a replayed hipGraph (each kernel reads the previous one's output)."""
import time, torch
dev = torch.device('cuda:0')
s = torch.cuda.Stream()
for n_elem in (1, 256 * 200, 256 * 200 * 16):
    x = torch.zeros(n_elem, device=dev)
    def body(k=400):
        y = x
        for _ in range(k):
            y = y + 1.0
        return y
    with torch.cuda.stream(s):
        body(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out = body()
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
    print(f'{n_elem:8d} elements: {dt * 1e6 / 400:6.2f} us per dependent kernel in a replayed graph', flush=True)
