"""Repeatability probe: the same input through fresh batches many times; every pipelined run is compared with ONE
sequential run.  Any difference is a race (the modes are bit-identical by construction)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import nnnoiseless_amd as nn
from nnnoiseless_amd.synthetic import make_streams, make_streams_fast
S, T, N, kind = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
x = make_streams(41, S, T) if kind == "slow" else make_streams_fast(S, T, seed=5)
def run(pipe, graph, close):
    bd = nn.BatchDenoiser(S)
    bd.set_pipeline(pipe); bd.set_graph(graph)
    o = bd.process(x)[0]
    if close:
        bd.close()
    return o
ref = run(False, False, True)
for pipe, graph, close in ((False, True, True), (True, True, True), (True, False, True), (True, True, False)):
    events = 0
    for it in range(N):
        o = run(pipe, graph, close)
        d = np.abs(o - ref).max(axis=2)
        bad = np.argwhere(d > 0)
        if len(bad):
            events += 1
            print("  pipe", pipe, "graph", graph, "close", close, "iter", it, "streams", sorted(set(bad[:, 0]))[:6], "frames", sorted(set(bad[:, 1]))[:8], "max %.3g" % d.max())
    print(kind, S, "pipe", pipe, "graph", graph, "close", close, ":", events, "of", N, "runs differ from the sequential run")
