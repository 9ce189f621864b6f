"""Do two branches of a captured hipGraph run concurrently on this ROCm?  (round 4 probe)
Each branch = N spin kernels (torch.cuda._sleep: one workgroup); a third variant joins/forks between every pair (fine-grained)."""
import sys, time, torch
dev = torch.device('cuda:0')
main = torch.cuda.Stream(); side = torch.cuda.Stream()
CYC = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
N = 50

def chain(n):
    for _ in range(n):
        torch.cuda._sleep(CYC)

def body_serial():
    chain(2 * N)

def body_fork_once():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        chain(N)
    chain(N)
    torch.cuda.current_stream().wait_stream(side)

def body_fork_k(k):
    def f():
        per = N // k
        for _ in range(k):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                chain(per)
            chain(per)
        torch.cuda.current_stream().wait_stream(side)
    return f

def body_fork_join_each():
    for _ in range(N):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            chain(1)
        chain(1)
        torch.cuda.current_stream().wait_stream(side)

def timeit(fn, reps=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

for name, body in (('serial 2N', body_serial), ('fork once', body_fork_once), ('fork 5x (no join between)', body_fork_k(5)),
                   ('fork 25x (no join between)', body_fork_k(25)), ('fork+join each kernel', body_fork_join_each)):
    with torch.cuda.stream(main):
        body(); torch.cuda.synchronize()
        eager = timeit(body)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            body()
        g.replay(); torch.cuda.synchronize()
        gt = timeit(g.replay)
    print(f'{name:32s} eager {eager:8.3f} ms   graph {gt:8.3f} ms', flush=True)
