"""Host allocator policy for adaptive runs.

An adapt cycle rebuilds the connectivity arrays and the tile plan: a few hundred MB of host arrays that live for one cycle.
With glibc's defaults every array of more than 128 KiB is its own mmap, returned to the kernel when freed, so every cycle
pays the page faults of all of them again -- measured on the benchmark host (EPYC 9575F, 2.8 M elements, 3D): tile plan
0.47 s -> 0.27 s, connectivity arrays 0.23 s -> 0.12 s once freed memory stays in the heap. This is a process-wide policy, so
the APPLICATION opts in (bench.py and examples/ call keep_heap(); a C++ application does the same with the three mallopt
calls below -- INTEGRATION.md section 4); the libraries never change it behind the caller's back."""
import ctypes

_M_TRIM_THRESHOLD, _M_TOP_PAD, _M_MMAP_MAX = -1, -2, -4
_done = False


def keep_heap(top_pad=1 << 30):
    """Serve large allocations from the heap and never trim it: freed arrays are reused by the next cycle."""
    global _done
    if _done:
        return True
    try:
        libc = ctypes.CDLL("libc.so.6")
        ok = (libc.mallopt(_M_MMAP_MAX, 0) and libc.mallopt(_M_TRIM_THRESHOLD, 2 ** 31 - 1)
              and libc.mallopt(_M_TOP_PAD, int(top_pad)))
    except (OSError, AttributeError):
        return False
    _done = bool(ok)
    return _done
