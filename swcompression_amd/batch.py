"""Device-resident batches for the many-buffer launch (`swc_batch_decompress`).

PyTorch is used only as plumbing here: HBM allocations (`torch.empty(..., device='cuda')`), the
current HIP stream, and (in bench.py) `torch.distributed` for multi-GPU bookkeeping.  The decode itself
is the hand-written HIP kernels inside libswc_hip.so, reached through the C ABI.
"""
import ctypes as C

import numpy as np

from . import _lib

CODECS = {"deflate": 1, "lz4_block": 2, "lzma2": 3, "lzma": 4, "bzip2_block": 5, "delta": 6, "lz4_compress": 7, "deflate_compress": 8}

JOB_DTYPE = np.dtype([("in", "<u8"), ("in_len", "<u8"), ("out", "<u8"), ("out_cap", "<u8"), ("out_len", "<u8"),
                      ("in_consumed", "<u8"), ("status", "<i4"), ("aux", "<i4"), ("dict", "<u8"), ("dict_len", "<u8")])
assert JOB_DTYPE.itemsize == C.sizeof(_lib.SwcJob) == 72


def _align(x, a=16):
    return (x + a - 1) // a * a


class DeviceBatch:
    """`n_distinct` compressed units staged once in HBM and tiled `tile` times at DISTINCT device
    addresses (inputs are replicated, every job owns its own output range), as SURVEY.md section 8d asks for:
    nothing is served from the 256 MiB Infinity Cache by accident."""

    def __init__(self, codec, units, caps, aux=None, extra=None, dict_values=None, tile=1, device="cuda:0", replicate_inputs=True,
                 select=None):
        """select = (lo, hi): the jobs are the units lo..hi-1 of the TILED unit list (list entry i is distinct unit i % n_distinct
        in replica i // n_distinct); default: the whole list of n_distinct * tile entries."""
        import torch
        self.torch = torch
        self.lib = _lib.load()
        if not self.lib.swc_device_available():
            raise RuntimeError("no usable gfx950 device: the MI355X engine has no CPU fallback")
        self.codec = CODECS[codec] if isinstance(codec, str) else int(codec)
        self.device = torch.device(device)
        nd = len(units)
        self.n_distinct = nd
        lo, hi = (0, nd * tile) if select is None else (int(select[0]), int(select[1]))
        self.tile = tile
        self.n = hi - lo
        lens = np.array([len(u) for u in units], dtype=np.uint64)
        caps = np.array(caps, dtype=np.uint64)
        in_sz = np.array([_align(int(x)) for x in lens], dtype=np.uint64)
        out_sz = np.array([_align(int(x)) for x in caps], dtype=np.uint64)
        in_off = np.concatenate([[0], np.cumsum(in_sz)[:-1]]).astype(np.uint64)
        in_round = int(in_sz.sum())
        host = np.zeros(in_round + 16, dtype=np.uint8)
        for u, o in zip(units, in_off):
            host[int(o):int(o) + len(u)] = np.frombuffer(u, dtype=np.uint8)
        idx = np.arange(lo, hi, dtype=np.int64)
        k_idx = (idx % nd).astype(np.int64) if nd else idx
        t_idx = (idx // nd).astype(np.uint64) if nd else idx.astype(np.uint64)
        t0 = int(t_idx.min()) if self.n else 0
        t1 = int(t_idx.max()) if self.n else 0
        in_tiles = (t1 - t0 + 1) if replicate_inputs else 1
        self.d_in = torch.empty(in_round * in_tiles + 16, dtype=torch.uint8, device=self.device)
        h = torch.from_numpy(host[:in_round])
        for t in range(in_tiles):
            self.d_in[t * in_round:(t + 1) * in_round].copy_(h)
        out_each = out_sz[k_idx]
        out_off = np.concatenate([[0], np.cumsum(out_each)[:-1]]).astype(np.uint64) if self.n else np.zeros(0, dtype=np.uint64)
        out_total = int(out_each.sum())
        self.d_out = torch.empty(out_total + 16, dtype=torch.uint8, device=self.device)
        jobs = np.zeros(self.n, dtype=JOB_DTYPE)
        jobs["in"] = self.d_in.data_ptr() + ((t_idx - np.uint64(t0)) * np.uint64(in_round) if replicate_inputs else 0) + in_off[k_idx]
        jobs["in_len"] = lens[k_idx]
        jobs["out"] = self.d_out.data_ptr() + out_off
        jobs["out_cap"] = caps[k_idx]
        jobs["status"] = 902
        if aux is not None:
            jobs["aux"] = np.array(aux, dtype=np.int32)[k_idx]
        if extra is not None:
            jobs["dict_len"] = np.array(extra, dtype=np.uint64)[k_idx]
        if dict_values is not None:  # integer carried in the `dict` field (LZMA: dictionary size, BZip2: stored block CRC)
            jobs["dict"] = np.array(dict_values, dtype=np.uint64)[k_idx]
        self._out_off = out_off.astype(np.int64)
        self.unit_index = k_idx            # which distinct unit every job decodes
        self.caps = caps[k_idx]
        self.in_lens = lens[k_idx]
        self._jobs_host = jobs
        self.d_jobs = torch.from_numpy(jobs.view(np.uint8).copy()).to(self.device)
        ws = self.lib.swc_batch_workspace_bytes(self.codec, self.n, int(caps.max()) if nd else 0)
        self.d_ws = torch.empty(max(ws, 16), dtype=torch.uint8, device=self.device)
        self.ws_bytes = ws
        self._crc_buf = None
        torch.cuda.synchronize(self.device)

    @property
    def total_in(self):
        return int(self.in_lens.sum())

    def launch(self, sync=False):
        torch = self.torch
        opts = _lib.SwcBatchOpts(self.device.index if self.device.index is not None else -1,
                                 torch.cuda.current_stream(self.device).cuda_stream, 1 if sync else 0, 0)
        st = self.lib.swc_batch_decompress_ws(self.codec, self.d_jobs.data_ptr(), self.n, self.d_ws.data_ptr(),
                                              self.ws_bytes, C.byref(opts))
        if st:
            raise RuntimeError("swc_batch_decompress failed with status %d" % st)

    def crc32(self):
        """CRC-32 of every job's output, computed on the device (swc_batch_crc32).  Returns a numpy uint32 array."""
        torch = self.torch
        d = torch.empty(self.n, dtype=torch.int32, device=self.device)
        opts = _lib.SwcBatchOpts(self.device.index if self.device.index is not None else -1,
                                 torch.cuda.current_stream(self.device).cuda_stream, 1, 0)
        st = self.lib.swc_batch_crc32(self.d_jobs.data_ptr(), self.n, d.data_ptr(), C.byref(opts))
        if st:
            raise RuntimeError("swc_batch_crc32 failed with status %d" % st)
        return d.cpu().numpy().view(np.uint32)

    def crc32_async(self):
        """swc_batch_crc32 on the current stream, no synchronisation, result left on the device (bench.py: the CRC-32 of
        every gzip member is part of the timed step)."""
        torch = self.torch
        if self._crc_buf is None:
            self._crc_buf = torch.empty(self.n, dtype=torch.int32, device=self.device)
        opts = _lib.SwcBatchOpts(self.device.index if self.device.index is not None else -1,
                                 torch.cuda.current_stream(self.device).cuda_stream, 0, 0)
        st = self.lib.swc_batch_crc32(self.d_jobs.data_ptr(), self.n, self._crc_buf.data_ptr(), C.byref(opts))
        if st:
            raise RuntimeError("swc_batch_crc32 failed with status %d" % st)

    CHECKSUMS = {"crc32": 1, "adler32": 2, "crc64": 3, "bzip2crc32": 4, "xxh32": 5}

    def checksum(self, kind):
        """Checksum `kind` (a key of CHECKSUMS, the names of the reference's CheckSums / XxHash32 functions) of every
        job's output, computed on the device (swc_batch_checksum).  Returns a numpy uint64 array."""
        torch = self.torch
        d = torch.empty(self.n, dtype=torch.int64, device=self.device)
        opts = _lib.SwcBatchOpts(self.device.index if self.device.index is not None else -1,
                                 torch.cuda.current_stream(self.device).cuda_stream, 1, 0)
        st = self.lib.swc_batch_checksum(self.CHECKSUMS[kind], self.d_jobs.data_ptr(), self.n, d.data_ptr(), C.byref(opts))
        if st:
            raise RuntimeError("swc_batch_checksum failed with status %d" % st)
        return d.cpu().numpy().view(np.uint64)

    def wipe_results(self):
        """Zeroes every job's output range, the result fields of the job records and the CRC buffer (bench.py: what is
        verified after the timed region must come from the last timed step, not from the warm-up)."""
        self.d_out.zero_()
        jobs = self._jobs_host.copy()
        jobs["status"] = 902
        jobs["out_len"] = 0
        jobs["in_consumed"] = 0
        self.d_jobs.copy_(self.torch.from_numpy(jobs.view(np.uint8)).to(self.device))
        if self._crc_buf is not None:
            self._crc_buf.zero_()

    def results(self):
        """Structured numpy array of the job records after the launch (synchronises)."""
        self.torch.cuda.synchronize(self.device)
        return self.d_jobs.cpu().numpy().view(JOB_DTYPE)

    def output(self, i, n=None):
        r = self.results() if n is None else None
        ln = int(min(r["out_len"][i], r["out_cap"][i])) if n is None else n
        o = int(self._out_off[i])
        return self.d_out[o:o + ln].cpu().numpy().tobytes()
