import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import nnnoiseless_amd as nn
from model_fixtures import make_model
from nnnoiseless_amd.synthetic import make_streams
sh = open("tests/golden/sh.rnn", "rb").read()
small, biggest = make_model(16, 20, 40, 72, seed=1), make_model(42, 43, 42, 127, seed=3)
sizes = [128, 192, 64, 70]
x = make_streams(41, sum(sizes), 20)
def run(pipe, grouped, graph=True):
    if grouped:
        models = [None] + [nn.RnnModel.from_bytes(b) for b in (sh, small, biggest)]
        bd = nn.BatchDenoiser(sum(sizes), groups=list(zip(models, sizes)))
    else:
        bd = nn.BatchDenoiser(sum(sizes))
    bd.set_pipeline(pipe); bd.set_graph(graph)
    return bd.process(x)[0]
for grouped in (False, True):
    ref = run(False, grouped)
    for graph in (True, False):
        for it in range(5):
            o = run(True, grouped, graph)
            d = np.abs(o - ref).max(axis=2)          # [S, T]
            bad = np.argwhere(d > 0)
            print("grouped", grouped, "graph", graph, "iter", it, "identical" if not len(bad) else
                  "DIFF streams %s frames %s max %.3g" % (sorted(set(bad[:, 0]))[:8], sorted(set(bad[:, 1])), d.max()))
