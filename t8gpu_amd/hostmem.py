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


class PinnedUploader:
    """Host -> device copies of large numpy arrays through ONE pinned staging buffer the application owns.

    `tensor.cuda()` on a pageable array lets the runtime stage it through its own pinned bounce buffers at ~10 GB/s; an adapt
    cycle uploads ~300 MB that way. Here a chunk is copied into the pinned buffer by the planners' threads
    (t8gpu_host_parallel_copy), handed to the DMA engine asynchronously, and the next chunk is staged meanwhile in the rest of
    the buffer; when the buffer is full the stream is synchronised and it starts over. Opt-in (use_pinned_uploads(): bench c5a,
    the example): the default path of the package stays `tensor.cuda()`."""

    def __init__(self, megabytes=256, chunk_megabytes=32):
        import torch
        from . import synth
        self.cap = int(megabytes) << 20
        if self.cap <= 0:
            raise ValueError("PinnedUploader needs a staging buffer of at least 1 MB")
        # a chunk is staged whole: never larger than the buffer (a 16 MB buffer with the default 32 MB chunk would let
        # t8gpu_host_parallel_copy write past its end), and rounded like the offsets
        self.chunk = max(256, min(int(chunk_megabytes) << 20, self.cap) & ~255)
        self.buf = torch.empty(self.cap, dtype=torch.uint8, pin_memory=True)
        self.off = 0
        self._in_flight = []      # (event) per staged region since the last wrap: a copy may have been queued on ANY stream
        self._copy = synth.lib().t8gpu_host_parallel_copy
        self._copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        self._copy.restype = None

    def upload(self, a):
        """numpy array (C-contiguous) -> CUDA tensor of the same dtype and shape (asynchronous on the current stream)."""
        import numpy as np
        import torch
        a = np.ascontiguousarray(a)
        out = torch.empty(a.shape, dtype=torch.from_numpy(a.reshape(-1)[:1]).dtype, device="cuda")
        n = a.nbytes
        if n == 0:
            return out
        if n < (1 << 20):                      # small arrays: not worth staging
            out.copy_(torch.from_numpy(a))
            return out
        dst = out.view(torch.uint8).reshape(-1)
        src, done = a.ctypes.data, 0
        while done < n:
            take = min(self.chunk, n - done)
            if self.off + take > self.cap:     # the DMA engine may still read what is staged: drain it, start over
                for ev in self._in_flight:     # (every copy since the last wrap, whichever stream it was queued on)
                    ev.synchronize()
                self._in_flight = []
                self.off = 0
            assert take <= self.cap - self.off
            self._copy(self.buf.data_ptr() + self.off, src + done, take)
            dst[done:done + take].copy_(self.buf[self.off:self.off + take], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()                        # on the stream the copy was queued on
            self._in_flight.append(ev)
            self.off += (take + 255) & ~255
            done += take
        return out


_uploader = None


def use_pinned_uploads(megabytes=256, chunk_megabytes=32):
    """From now on solver / plan uploads of this process go through a PinnedUploader (uploader() returns it). Allocates pinned
    memory, i.e. creates a context on the CURRENT device: call it after torch.cuda.set_device(LOCAL_RANK)."""
    global _uploader
    if _uploader is None:
        _uploader = PinnedUploader(megabytes, chunk_megabytes)
    return _uploader


def uploader():
    return _uploader


def to_device(a, dtype=None):
    """numpy array -> CUDA tensor (of `dtype` if given): through the pinned uploader if the application switched it on."""
    import numpy as np
    import torch
    if _uploader is None:
        t = torch.from_numpy(np.ascontiguousarray(a))
        return (t if dtype is None else t.to(dtype)).cuda()
    t = _uploader.upload(a)
    return t if dtype is None or t.dtype == dtype else t.to(dtype)
