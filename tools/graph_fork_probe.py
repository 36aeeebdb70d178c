"""Diagnostic: does a replayed hipGraph run two forked branches concurrently on this stack?  Two independent single-block spin
kernels (torch.cuda._sleep) captured on forked streams; replay time ~ one sleep if concurrent, ~ two if the executor serialises."""
import time
import torch as th

dev = th.device("cuda")
s1, s2 = th.cuda.Stream(), th.cuda.Stream()
import sys
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 400000


def body(fork):
    main = th.cuda.current_stream()
    if fork:
        s1.wait_stream(main); s2.wait_stream(main)
        with th.cuda.stream(s1):
            th.cuda._sleep(cycles)
        with th.cuda.stream(s2):
            th.cuda._sleep(cycles)
        main.wait_stream(s1); main.wait_stream(s2)
    else:
        th.cuda._sleep(cycles); th.cuda._sleep(cycles)


for fork in (False, True):
    for graph in (False, True):
        body(fork); th.cuda.synchronize()
        if graph:
            g = th.cuda.CUDAGraph()
            with th.cuda.graph(g, capture_error_mode="thread_local"):
                body(fork)
            run = g.replay
        else:
            run = lambda: body(fork)
        run(); th.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            run()
        th.cuda.synchronize()
        print("fork=%d graph=%d: %.1f us per iteration" % (fork, graph, 1e6 * (time.perf_counter() - t) / 20))
