"""Host-side plumbing of the adaptive loop: zero-copy views of the provider's / planner's arrays and the allocator policy."""
import gc
import subprocess
import sys

import numpy as np

from t8gpu_amd.plan import HostPlainPlan
from t8gpu_amd.synth import SynthMesh


def test_views_keep_their_c_handle_alive():
    """Partition.face_neighbors & co. and the HostPlainPlan arrays are views of C++ memory (t8gpu_amd/synth.py: _view); an array
    must stay valid after the Python object that handed it out is gone."""
    mesh = SynthMesh(3, 3, 4, band=0.1)
    part = mesh.partition()
    fn, nr, ar = part.face_neighbors, part.normals, part.areas
    want = (fn.copy(), nr.copy(), ar.copy())
    assert not fn.flags.owndata and fn.size == 2 * part.F + part.B
    plan = HostPlainPlan.from_partition(part, tmax=64, fcap=150, patches=True)
    lr, ell, order = plan.face_lr, plan.ell, plan.tile_order
    want_plan = (lr.copy(), ell.copy(), order.copy())
    assert not lr.flags.owndata and ell.shape == (max(1, plan.n_ell_rows), plan.ell_width)
    del part, plan, mesh
    gc.collect()
    junk = [np.full(1 << 20, 7, np.int32) for _ in range(8)]      # would land in freed memory
    assert np.array_equal(fn, want[0]) and np.array_equal(nr, want[1]) and np.array_equal(ar, want[2])
    assert np.array_equal(lr, want_plan[0]) and np.array_equal(ell, want_plan[1]) and np.array_equal(order, want_plan[2])
    del junk
    sub = fn[10:20]                                                # a slice of a view holds the owner too
    del fn
    gc.collect()
    assert np.array_equal(sub, want[0][10:20])


def test_keep_heap_policy_in_a_child_process():
    """hostmem.keep_heap() is a process-wide malloc policy (three mallopt calls): set in a child, the mesh / plan cycle
    still works and a second call is a no-op."""
    code = ("from t8gpu_amd import hostmem\n"
            "from t8gpu_amd.synth import SynthMesh\n"
            "from t8gpu_amd.plan import HostPlainPlan\n"
            "import numpy as np\n"
            "assert hostmem.keep_heap() and hostmem.keep_heap()\n"
            "m = SynthMesh(3, 3, 5, band=0.1)\n"
            "for _ in range(3):\n"
            "    marks = np.zeros(m.num_elements, np.int8); marks[::7] = 1\n"
            "    m, ad = m.adapt(marks)\n"
            "    p = m.partition()\n"
            "    h = HostPlainPlan.from_partition(p, patches=True)\n"
            "    big = np.ones(40_000_000 // 8)\n"
            "    assert h.elem_off[-1] == p.N and big.sum() == big.size\n"
            "print('ok', m.num_elements)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       cwd=str(__import__("pathlib").Path(__file__).resolve().parents[1]))
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stderr[-2000:]
