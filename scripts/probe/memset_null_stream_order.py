"""Probe (root cause of the round-3 "priority stream ordering failure"): hipMemset on DEVICE memory
returns to the host before the fill has run, and -- issued on the null stream -- is not ordered
against hipStreamNonBlocking streams.  libcmdg allocated its LSRK work states lazily inside the first
step with hipMalloc + hipMemset and then wrote them from its own non-blocking streams: a fill that
lands after the first stage's stores zeroes them (EngineBase::ensure_work before round 4).
    python scripts/probe/memset_null_stream_order.py [torch|system]
Exit code 0 and a line "ASYNC ..." / "SYNC ..." with the measured times; the overlap test counts
bytes of a buffer that a non-blocking stream filled with 2 AFTER the host returned from
hipMemset(buffer, 1): any byte still 1 at the end was written by the "synchronous" memset later."""
import ctypes as C
import os
import sys
import time

stack = sys.argv[1] if len(sys.argv) > 1 else "torch"
if stack == "torch":
    import torch  # noqa: F401
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
else:
    hip = C.CDLL("/opt/rocm/lib/libamdhip64.so.7", mode=C.RTLD_GLOBAL)
vp = C.c_void_p
hip.hipMalloc.argtypes = [C.POINTER(vp), C.c_size_t]
hip.hipMemset.argtypes = [vp, C.c_int, C.c_size_t]
hip.hipMemsetAsync.argtypes = [vp, C.c_int, C.c_size_t, vp]
hip.hipMemcpy.argtypes = [vp, vp, C.c_size_t, C.c_int]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(vp), C.c_uint]
assert hip.hipSetDevice(0) == 0
n = 4 << 30
buf, s = vp(), vp()
assert hip.hipMalloc(C.byref(buf), n) == 0
assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0          # hipStreamNonBlocking
assert hip.hipMemset(buf, 0, n) == 0 and hip.hipDeviceSynchronize() == 0   # warm-up
t0 = time.perf_counter()
assert hip.hipMemset(buf, 1, n) == 0
t_ret = time.perf_counter() - t0
assert hip.hipDeviceSynchronize() == 0
t_done = time.perf_counter() - t0
print("%s hipMemset of 4 GiB of device memory: returned after %.1f us, device done after %.1f us"
      % ("ASYNC" if t_ret < 0.5 * t_done else "SYNC", 1e6 * t_ret, 1e6 * t_done), flush=True)
# overlap with a non-blocking stream
m = 1 << 30
assert hip.hipMemset(buf, 0, n) == 0 and hip.hipDeviceSynchronize() == 0
assert hip.hipMemset(buf, 1, n) == 0                 # "synchronous", null stream
assert hip.hipMemsetAsync(buf, 2, m, s) == 0         # the consumer's first write, its own stream
assert hip.hipDeviceSynchronize() == 0
host = (C.c_ubyte * m)()
assert hip.hipMemcpy(host, buf, m, 2) == 0
ones = bytes(host).count(1)
print("bytes of the first GiB left at 1 by the later-finishing hipMemset: %d of %d (%s)"
      % (ones, m, "NOT ORDERED against the non-blocking stream" if ones else "no overlap seen this time"),
      flush=True)
