"""hipGraph branches with SLACK: main chain of N kernels, side chain with a fraction of the work, forked k times, joined once
at the end.  If the replay time stays at the main chain's own time, forks are free when the side branch has slack."""
import sys, time, torch
main = torch.cuda.Stream(); side = torch.cuda.Stream()
CYC = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000
N = 100

def chain(n, c=CYC):
    for _ in range(n):
        torch.cuda._sleep(c)

def body(k, frac):
    def f():
        per = N // k
        for _ in range(k):
            chain(per)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                chain(max(1, int(per * frac)))
        chain(per)  # tail of the main chain under which the last side segment runs
        torch.cuda.current_stream().wait_stream(side)
    return f

def timeit(fn, reps=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

with torch.cuda.stream(main):
    for k in (1, 2, 5, 10, 25):
        for frac in (0.0, 0.3, 0.6, 0.9):
            if frac == 0.0:
                b = lambda k=k: chain(N + N // k)
            else:
                b = body(k, frac)
            b(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=main):
                b()
            g.replay(); torch.cuda.synchronize()
            print(f'forks {k:3d}  side work {frac:.1f} of main  graph {timeit(g.replay):8.3f} ms', flush=True)
