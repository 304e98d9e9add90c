# debug aid: the child of tests/test_gpu_graph.py as a script (argv[1] = repo root)

import sys, types
import numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from t8gpu_amd import native
from t8gpu_amd.solver import PlainSolver
from t8gpu_amd.synth import SynthMesh
mesh = SynthMesh(2, 5, 8, band=0.05)
whole, half = mesh.partition(), mesh.partition(0, 2)
x, y = whole.centres[:, 0], whole.centres[:, 1]
rho = 1.5 + 0.4 * np.sin(4 * np.pi * y) * np.cos(2 * np.pi * x)
v1, v2 = 0.3 * np.cos(4 * np.pi * y), 0.2 * np.sin(2 * np.pi * x) * np.sin(4 * np.pi * y)
st = np.stack([rho, rho * v1, rho * v2, 0 * rho, 2.5 / 0.4 + 0.5 * rho * (v1 * v1 + v2 * v2)])
n2 = whole.N // 2
st[:, n2:] = st[:, :n2]
gidx = np.concatenate([np.arange(half.N), half.ghost_global])
comm = native.NativeComm(0, 1, lambda b, src: b)
fake = types.SimpleNamespace(N=half.N, G=half.G, cells_per_element=1, peers=np.zeros(1, np.int32), send_off=half.send_off,
                             recv_off=half.recv_off, send_idx=half.send_idx)
def run(graph):
    local = st[:, gidx].copy()
    g = PlainSolver(half, torch.float64, mode="fused", state=local, plan_options=dict(tmax=64, fcap=160))
    g.use_native_stepper(native.NativeHalo(fake, torch.float64, comm))
    g.stepper.graph(graph)
    dt = 0.1 * 2.0 ** -mesh.finest_level
    for _ in range(4):
        g.iterate_steps(6, dt)
    assert native.stream_wait(torch.cuda.current_stream(), 60.0) == 0
    return g.state().clone(), g.stepper.graph()
direct, _ = run(False)
print("direct enqueue done", flush=True)
replayed, counts = run(True)
print("graph counts", counts, flush=True)
assert counts[1] == 4 and counts[0] >= 1, counts
assert torch.equal(direct, replayed)
print("GRAPH WITH RCCL OK", flush=True)
