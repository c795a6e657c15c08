"""Probe (root cause of the round-3 "priority stream ordering failure"): hipMemset on DEVICE memory
returns to the host before the fill has run, and -- issued on the null stream -- is not ordered
against hipStreamNonBlocking streams.  libcmdg allocated its LSRK work states lazily inside the first
step with hipMalloc + hipMemset and then wrote them from its own non-blocking streams: a fill that
lands after the first stage's stores zeroes them (EngineBase::ensure_work before round 4).
    python scripts/probe/memset_null_stream_order.py [torch|system]
Part 1 times the call against the device.  Part 2 reproduces the hazard deterministically: the null
stream is kept busy (an 8 GiB fill), hipMemset(W, 1) is issued behind it and RETURNS, a non-blocking
stream then writes W <- 2 (the consumer's first store) and finishes while the fill of W is still
queued; at the end W holds 1: the "earlier, synchronous" fill overwrote the later store."""
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
hip.hipStreamSynchronize.argtypes = [vp]
assert hip.hipSetDevice(0) == 0
big_n, n = 8 << 30, 64 << 20
big, W, s = vp(), vp(), vp()
assert hip.hipMalloc(C.byref(big), big_n) == 0 and hip.hipMalloc(C.byref(W), n) == 0
assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0          # hipStreamNonBlocking
assert hip.hipMemset(big, 0, big_n) == 0 and hip.hipMemset(W, 0, n) == 0 and hip.hipDeviceSynchronize() == 0
t0 = time.perf_counter()
assert hip.hipMemset(big, 1, big_n) == 0
t_ret = time.perf_counter() - t0
assert hip.hipDeviceSynchronize() == 0
t_done = time.perf_counter() - t0
print("%s hipMemset of 8 GiB of device memory: returned after %.1f us, device done after %.1f us"
      % ("ASYNC" if t_ret < 0.5 * t_done else "SYNC", 1e6 * t_ret, 1e6 * t_done), flush=True)
# the hazard
assert hip.hipMemset(big, 3, big_n) == 0             # null stream busy for ~1.4 ms
assert hip.hipMemset(W, 1, n) == 0                   # "synchronous" fill, queued behind it; returns
assert hip.hipMemsetAsync(W, 2, n, s) == 0           # the consumer's first store, its own stream
assert hip.hipStreamSynchronize(s) == 0              # the consumer is done ...
assert hip.hipDeviceSynchronize() == 0               # ... and now everything is
host = (C.c_ubyte * n)()
assert hip.hipMemcpy(host, W, n, 2) == 0
b = bytes(host)
ones, twos = b.count(1), b.count(2)
print("W after hipMemset(W,1) [null stream] then hipMemsetAsync(W,2) [non-blocking stream]: %d bytes = 1, %d bytes = 2  -> %s"
      % (ones, twos, "the fill landed AFTER the later store (unordered)" if ones else "ordered this time"), flush=True)
sys.exit(0)
