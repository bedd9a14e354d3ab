"""GL command traces: the ordered list of C-ABI calls (with payloads) that
WebRender's render thread issues for a frame.  The reference has no such
format (its closest analogue is capture/replay of built frames,
webrender/src/capture.rs; wrench replays display lists instead, main.rs:756);
a trace is what stands in for `wrench` here, because the Rust caller cannot be
built in this environment.

A trace recorded from the Python Renderer mirror can be replayed against any
backend exporting the ABI -- the oracle or libwrhip -- either call by call from
Python (`replay`) or by the native replayer (csrc/wr_replay.c via
`NativeReplayer`), which removes interpreter overhead from timed loops the way
`wrench perf` times native frames (wrench/src/perf.rs:198-270).

Binary layout (little endian):
  u32 magic 'WRTR', u32 n_calls, u32 blob_bytes, u32 scratch_bytes
  n_calls x { u16 fn_id, u16 nargs, nargs x { u32 tag, u32 aux, u64 value } }
  blob bytes
GetUniformLocation carries one argument more than the call has: the location the RECORDING backend returned.  Uniform
locations are backend-specific (swgl: per program, in order of first use); a replayer maps recorded -> actual per program
and rewrites the location of Uniform1i / Uniform4fv / UniformMatrix4fv (csrc/wr_replay.c).
tags: 0 int, 1 f32, 2 f64, 3 blob(value=offset, aux=len), 4 null,
      5 scratch(value=offset into a replayer-owned buffer), 6 context handle,
      7 scratch initialised from a blob on every replay of the call (value = blob offset << 32 | scratch offset, aux=len:
        caller memory the backend reads at the call and writes later -- SetTextureBuffer's backing store)
"""
import ctypes as C
import os
import struct
import subprocess
import numpy as np
from .glapi import SIGNATURES, P, c_char_p, f32, f64, repo_root

FN_NAMES = list(SIGNATURES.keys())
FN_ID = {n: i for i, n in enumerate(FN_NAMES)}

# (function, arg index) pairs whose pointer is an OUTPUT written by the callee
OUT_PTRS = {
    ("GenQueries", 1), ("GenBuffers", 1), ("GenTextures", 1), ("GenFramebuffers", 1),
    ("GenRenderbuffers", 1), ("GenVertexArrays", 1), ("GetQueryObjectui64v", 2),
    ("GetIntegerv", 1), ("GetBooleanv", 1), ("ReadPixels", 6),
    ("GetColorBuffer", 2), ("GetColorBuffer", 3), ("GetColorBuffer", 4),
}
# pointer-typed args that are really byte offsets / small integers
INT_PTRS = {("VertexAttribPointer", 5), ("VertexAttribIPointer", 4)}

TAG_INT, TAG_F32, TAG_F64, TAG_BLOB, TAG_NULL, TAG_SCRATCH, TAG_CTX, TAG_SCRATCH_INIT = range(8)


def _nbytes(x):
    if isinstance(x, (bytes, bytearray)):
        return len(x)
    if hasattr(x, "nbytes"):
        return int(x.nbytes)
    return C.sizeof(x)


def _tobytes(x):
    if isinstance(x, (bytes, bytearray)):
        return bytes(x)
    if hasattr(x, "tobytes"):
        return np.ascontiguousarray(x).tobytes()
    return bytes(x)


class Trace:
    def __init__(self):
        self.calls = []      # (fn_id, [(tag, aux, value)])
        self.blobs = bytearray()
        self.scratch = 0
        self.enabled = True

    def _blob(self, data):
        off = (len(self.blobs) + 15) & ~15
        self.blobs.extend(b"\0" * (off - len(self.blobs)))
        self.blobs.extend(data)
        return off

    def record(self, name, args, ret=None):
        if not self.enabled:
            return
        argtypes = SIGNATURES[name][1]
        out = []
        for i, (a, t) in enumerate(zip(args, argtypes)):
            if name == "MakeCurrent" or name in ("DestroyContext", "ReferenceContext"):
                out.append((TAG_CTX if a else TAG_NULL, 0, 0))
            elif t is c_char_p:
                b = (a if isinstance(a, bytes) else a.encode()) + b"\0"
                out.append((TAG_BLOB, len(b), self._blob(b)))
            elif t is P:
                if (name, i) in INT_PTRS or isinstance(a, int):
                    out.append((TAG_INT, 0, int(a or 0)))
                elif a is None:
                    out.append((TAG_NULL, 0, 0))
                elif (name, i) in OUT_PTRS:
                    n = _nbytes(a)
                    off = (self.scratch + 63) & ~63
                    self.scratch = off + n
                    out.append((TAG_SCRATCH, n, off))
                else:
                    data = _tobytes(a)
                    out.append((TAG_BLOB, len(data), self._blob(data)))
            elif t is f32:
                out.append((TAG_F32, 0, struct.unpack("<I", struct.pack("<f", float(a)))[0]))
            elif t is f64:
                out.append((TAG_F64, 0, struct.unpack("<Q", struct.pack("<d", float(a)))[0]))
            else:
                out.append((TAG_INT, 0, int(a) & 0xFFFFFFFFFFFFFFFF))
        if ret is not None:      # (GetUniformLocation: the location the recording backend returned, as one more int)
            out.append((TAG_INT, 0, int(ret) & 0xFFFFFFFFFFFFFFFF))
        self.calls.append((FN_ID[name], out))

    def serialize(self):
        parts = [struct.pack("<4sIII", b"WRTR", len(self.calls), len(self.blobs), self.scratch + 64)]
        for fid, args in self.calls:
            parts.append(struct.pack("<HH", fid, len(args)))
            for tag, aux, val in args:
                parts.append(struct.pack("<IIQ", tag, aux, val & 0xFFFFFFFFFFFFFFFF))
        parts.append(bytes(self.blobs))
        return b"".join(parts)

    def __len__(self):
        return len(self.calls)


def replayer_path():
    return os.path.join(repo_root(), "webrender_amd", "csrc", "libwr_replay.so")


class NativeReplayer:
    """Replays serialized traces against a backend library from native code."""

    def __init__(self, backend_path):
        self.lib = C.CDLL(replayer_path())
        self.lib.wr_replay_open.restype = C.c_void_p
        self.lib.wr_replay_open.argtypes = [C.c_char_p]
        self.lib.wr_replay_exec.restype = C.c_int
        self.lib.wr_replay_exec.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        self.lib.wr_replay_loop.restype = C.c_int
        self.lib.wr_replay_loop.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_int,
                                            C.POINTER(C.c_double)]
        self.lib.wr_replay_stream.restype = C.c_int
        self.lib.wr_replay_stream.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
        self.lib.wr_replay_scratch.restype = C.c_void_p
        self.lib.wr_replay_scratch.argtypes = [C.c_void_p, C.c_size_t]
        self.lib.wr_replay_sym.restype = C.c_void_p
        self.lib.wr_replay_sym.argtypes = [C.c_void_p, C.c_char_p]
        self.lib.wr_replay_close.argtypes = [C.c_void_p]
        self.h = self.lib.wr_replay_open(os.path.abspath(backend_path).encode())
        if not self.h:
            raise RuntimeError(f"wr_replay_open({backend_path}) failed")

    def exec(self, trace_bytes):
        rc = self.lib.wr_replay_exec(self.h, trace_bytes, len(trace_bytes))
        if rc != 0:
            raise RuntimeError(f"trace replay failed at call {rc - 1}")

    def loop(self, trace_bytes, warmup, iters):
        """Run the trace warmup+iters times; returns per-iteration wall ms."""
        out = (C.c_double * iters)()
        rc = self.lib.wr_replay_loop(self.h, trace_bytes, len(trace_bytes), warmup, iters, out)
        if rc != 0:
            raise RuntimeError(f"trace replay failed at call {rc - 1}")
        return np.array(out[:], dtype=np.float64)

    def stream(self, trace_bytes, iters):
        """Replay `iters` frames back to back with one Finish() at the end
        (throughput mode); returns total wall ms."""
        out = C.c_double(0.0)
        rc = self.lib.wr_replay_stream(self.h, trace_bytes, len(trace_bytes), iters, C.byref(out))
        if rc != 0:
            raise RuntimeError(f"trace replay failed at call {rc - 1}")
        return out.value

    def scratch(self, offset, nbytes):
        """Bytes the last replay wrote through an output pointer (e.g. ReadPixels)."""
        p = self.lib.wr_replay_scratch(self.h, offset)
        return C.string_at(p, nbytes)

    def symbol(self, name):
        return self.lib.wr_replay_sym(self.h, name.encode())

    def close(self):
        if self.h:
            self.lib.wr_replay_close(self.h)
            self.h = None
