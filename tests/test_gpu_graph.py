"""-m gpu: hipGraph replay of the native step driver (t8gpu_hip_plain_stepper_graph): a whole iterate_steps() call --
tile / block launches, cross-stream events and, with a halo, the RCCL exchange -- captured once and replayed with one
hipGraphLaunch must give bitwise the result of the direct enqueue."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from _gpu import perturbed_state
from t8gpu_amd.solver import PlainSolver, SubgridSolver
from t8gpu_amd.synth import SynthMesh

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("kind", ["plain", "subgrid"])
def test_graph_replay_equals_direct_enqueue_single_rank(kind):
    if kind == "plain":
        mesh = SynthMesh(2, 4, 7, band=0.06)
        part = mesh.partition()
        st = perturbed_state(part, 3)
        make = lambda: PlainSolver(part, torch.float64, mode="fused", state=st)
        dt = 0.1 * 2.0 ** -mesh.finest_level
    else:
        mesh = SynthMesh(3, 3, 4, band=0.05)
        part = mesh.partition(subgrid=True)
        make = lambda: SubgridSolver(part, torch.float32, mode="fused")
        dt = 0.1 * 2.0 ** -(mesh.finest_level + 2)
    a, b = make(), make()
    a.use_native_stepper()
    b.use_native_stepper()
    b.stepper.graph(True)
    for n in (5, 5, 5, 2, 5, 5, 5):      # an odd step count swaps the roles of the Step0 / Step3 planes from call to call
        a.iterate_steps(n, dt)
        b.iterate_steps(n, dt)
    torch.cuda.synchronize()
    captures, replays = b.stepper.graph()
    # three argument sets -- (5, roles A), (5, roles B), (2, roles B) -- and the stepper keeps four executables: every set
    # is captured exactly once however the calls alternate (one cached executable re-captured on every call)
    assert replays == 7 and captures == 3, (captures, replays)
    assert torch.equal(a.state(), b.state())
    assert a.stepper.graph() == (0, 0)


CHILD = r"""
import ctypes, faulthandler, os, sys, types
faulthandler.enable()
import numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
if os.environ.get("T8GPU_TEST_SEGV_SHIM"):                      # C backtrace of a crash inside the runtime (diagnostic run)
    ctypes.CDLL(os.environ["T8GPU_TEST_SEGV_SHIM"]).segv_backtrace_install()
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
mode = os.environ["T8GPU_TEST_GRAPH_CHILD"]
if mode == "rccl_capture_off":                # no T8GPU_GRAPH_RCCL=1: a stepper with a halo keeps the direct enqueue
    assert counts == (0, 0), counts
    assert torch.equal(direct, replayed)
    print("HALO: DIRECT ENQUEUE IN GRAPH MODE OK", flush=True)
elif mode == "no_rccl":                       # diagnostic build without the RCCL group: ghosts never arrive, only "captured
    assert counts[1] == 4 and counts[0] >= 1, counts       # and replayed" counts
    print("GRAPH WITHOUT RCCL CAPTURED AND REPLAYED", flush=True)
else:
    assert counts[1] == 4 and counts[0] >= 1, counts
    assert torch.equal(direct, replayed)
    print("GRAPH WITH RCCL OK", flush=True)
"""


def _child(tmp_path, mode, env, log):
    script = tmp_path / "graph_rccl_child.py"
    script.write_text(CHILD)
    res = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=280,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", T8GPU_DEBUG_GRAPH="1", T8GPU_TEST_GRAPH_CHILD=mode, **env))
    out = res.stdout + res.stderr + f"\n[child exit code {res.returncode}]\n"
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        with open(os.path.join(ROOT, "gpurun_out", log), "w") as f:
            f.write(out)
    return res, out


def test_graph_replay_of_the_multi_rank_pipeline_with_rccl_self_exchange(tmp_path):
    """The three-stream multi-rank pipeline INCLUDING the RCCL groups, captured once and replayed (one-rank communicator,
    rank 0 exchanging with itself on a shift-symmetric problem, so the ghost values matter): bitwise the direct enqueue.
    The exchange chain is captured on the ORIGIN stream of the capture and the deep tiles fork off (stepper.hip) -- with
    the RCCL group on a forked stream hipStreamEndCapture crashes on this stack (the opt-in diagnostic below). Runs in a
    child process all the same: a runtime crash must not take the test session down."""
    res, out = _child(tmp_path, "rccl", dict(T8GPU_GRAPH_RCCL="1"), "graph_child_rccl.log")
    assert res.returncode == 0 and "GRAPH WITH RCCL OK" in res.stdout, out[-3000:]


def test_graph_mode_is_off_by_default_for_steppers_with_a_halo(tmp_path):
    """Capturing RCCL groups is opt-in (T8GPU_GRAPH_RCCL=1; ADVICE r3: a replayed RCCL group has never run across xGMI): by
    default a stepper with a halo enqueues directly -- the two-lane driver -- whatever the graph switch says (zero captures,
    zero replays, same bits)."""
    res, out = _child(tmp_path, "rccl_capture_off", {}, "graph_child_halo_direct.log")
    assert res.returncode == 0 and "HALO: DIRECT ENQUEUE IN GRAPH MODE OK" in res.stdout, out[-3000:]


def test_graph_capture_of_the_three_stream_pipeline_without_the_rccl_group(tmp_path):
    """The multi-rank pipeline's capture with the RCCL group compiled out (diagnostic build `norccl` of t8gpu_amd/build.py:
    pack, unpack, tile classes on three streams joined through events; the ghosts never arrive then, so only `captured and
    replayed` is checked). Separates a capture problem of the fork / join structure from one of RCCL: with forked streams
    waiting on each other's events hipStreamEndCapture crashed on this stack; with every dependency routed through the
    origin stream (stepper.hip) this capture works."""
    from t8gpu_amd import build
    lib = build.NO_RCCL_LIB
    if not os.path.exists(lib):
        pytest.skip("diagnostic build missing: python -c 'from t8gpu_amd import build; build.build_diagnostic_variants()'")
    res, out = _child(tmp_path, "no_rccl", dict(T8GPU_HIP_LIB=lib, T8GPU_GRAPH_RCCL="1"), "graph_child_no_rccl.log")
    assert res.returncode == 0 and "GRAPH WITHOUT RCCL CAPTURED AND REPLAYED" in res.stdout, out[-3000:]


@pytest.mark.skipif(os.environ.get("T8GPU_TEST_RCCL_CAPTURE") != "1",
                    reason="opt-in (T8GPU_TEST_RCCL_CAPTURE=1): the RCCL group on a FORKED stream of a capture crashes inside "
                           "hipStreamEndCapture on this stack (DESIGN.md section 6); diagnosed once, not re-run per suite")
def test_rccl_group_on_a_forked_stream_of_a_capture_opt_in(tmp_path):
    """DIAGNOSTIC, opt-in: T8GPU_GRAPH_VARIANT=5 puts the exchange chain back on a forked stream of the capture -- the
    layout of rounds 1-2, which dies with SIGSEGV in hipStreamEndCapture in relaxed, global and thread-local capture mode
    alike. Round 3 got no handler output (the handlers ran on the faulting thread's own, exhausted stack); round 4's shim runs on an
    alternate stack and shows the cause: one frame of libamdhip64.so (+0x2d34a8 in the torch wheel's HIP 7.0.51831) repeated through
    the whole backtrace -- unbounded recursion inside the runtime while hipStreamEndCapture walks the captured graph, a stack
    overflow. Kept to re-check newer stacks."""
    shim = tmp_path / "segv_backtrace.so"
    subprocess.run(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", str(shim), os.path.join(ROOT, "scripts", "segv_backtrace.c")], check=True)
    res, out = _child(tmp_path, "rccl", dict(T8GPU_GRAPH_VARIANT="5", T8GPU_GRAPH_RCCL="1", T8GPU_TEST_SEGV_SHIM=str(shim)), "graph_child_rccl_forked.log")
    assert "direct enqueue done" in res.stdout, out[-3000:]
    if res.returncode != 0:
        pytest.xfail("RCCL group on a forked stream of a capture: " + out[-1500:].replace("\n", " | "))
    assert "GRAPH WITH RCCL OK" in res.stdout
