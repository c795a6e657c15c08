"""Probe: which ingredient of libcmdg's step capture makes hipStreamEndCapture crash when RCCL
point-to-point groups are recorded?  One hypothesis per MODE, one fresh process per run, exit code 0 =
captured, instantiated, replayed three times and the payload arrived each time; anything else (a
Python exception -> 1, a signal -> 128+n) is the answer for that mode.  No kernel of ours is launched:
HIP and RCCL are driven through ctypes only.

    python scripts/probe/rccl_capture_probe.py MODE [STACK]

STACK  torch   the HIP runtime and RCCL that `import torch` brings into the process (what the tests
               and bench.py run on)
       system  /opt/rocm/lib/libamdhip64.so.7 + /opt/rocm/lib/librccl.so.1 without torch (what a Julia
               host would load)
MODE   eager              no capture (control): one group {recv from self, send to self} per step
       origin-global      the group recorded on the capture's origin stream, hipStreamCaptureModeGlobal
       origin-threadlocal same, hipStreamCaptureModeThreadLocal
       origin-relaxed     same, hipStreamCaptureModeRelaxed (what cmdg.hip uses)
       fork-relaxed       origin stream forks a second stream by an event, the group is recorded on the
                          second stream, joined back by an event (libcmdg's halo stream), Relaxed
       fork-global        same fork, Global
       fork-memcpy        same fork with a device copy in place of the group (control)
       two-groups-origin  two groups back to back on the origin stream, Relaxed
       send-recv-split    Relaxed, origin stream: ncclRecv and ncclSend each in a group of its own is
                          impossible with a self-neighbour (it would deadlock), so: one group with two
                          recv + two send of half the payload (several operations per group)
scripts/probe/run_rccl_capture_probes.sh runs all of them under a timeout and tabulates."""
import ctypes as C
import os
import sys

mode = sys.argv[1] if len(sys.argv) > 1 else "eager"
stack = sys.argv[2] if len(sys.argv) > 2 else "torch"
if stack == "torch":
    import torch  # noqa: F401  (brings its libamdhip64.so.7 / librccl.so.1 into the process)
    tl = os.path.join(os.path.dirname(torch.__file__), "lib")
    hip = C.CDLL(os.path.join(tl, "libamdhip64.so"))
    nccl = C.CDLL(os.path.join(tl, "librccl.so"))
else:
    hip = C.CDLL("/opt/rocm/lib/libamdhip64.so.7", mode=C.RTLD_GLOBAL)
    nccl = C.CDLL("/opt/rocm/lib/librccl.so.1", mode=C.RTLD_GLOBAL)

vp = C.c_void_p


def chk(rc, what):
    if rc != 0:
        raise RuntimeError("%s -> %d" % (what, rc))


class Uid(C.Structure):
    _fields_ = [("b", C.c_char * 128)]


nccl.ncclCommInitRank.argtypes = [C.POINTER(vp), C.c_int, Uid, C.c_int]
nccl.ncclSend.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
nccl.ncclRecv.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
hip.hipMalloc.argtypes = [C.POINTER(vp), C.c_size_t]
hip.hipMemcpy.argtypes = [vp, vp, C.c_size_t, C.c_int]
hip.hipMemcpyAsync.argtypes = [vp, vp, C.c_size_t, C.c_int, vp]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(vp), C.c_uint]
hip.hipEventCreateWithFlags.argtypes = [C.POINTER(vp), C.c_uint]
hip.hipEventRecord.argtypes = [vp, vp]
hip.hipStreamWaitEvent.argtypes = [vp, vp, C.c_uint]
hip.hipStreamBeginCapture.argtypes = [vp, C.c_int]
hip.hipStreamEndCapture.argtypes = [vp, C.POINTER(vp)]
hip.hipGraphInstantiate.argtypes = [C.POINTER(vp), vp, vp, vp, C.c_size_t]
hip.hipGraphLaunch.argtypes = [vp, vp]
hip.hipStreamSynchronize.argtypes = [vp]

N = 4096  # doubles
H2D, D2H, D2D = 1, 2, 3
chk(hip.hipSetDevice(0), "hipSetDevice")
uid = Uid()
chk(nccl.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
comm = vp()
chk(nccl.ncclCommInitRank(C.byref(comm), 1, uid, 0), "ncclCommInitRank")
src, dst = vp(), vp()
chk(hip.hipMalloc(C.byref(src), 8 * N), "hipMalloc")
chk(hip.hipMalloc(C.byref(dst), 8 * N), "hipMalloc")
s0, s1 = vp(), vp()
chk(hip.hipStreamCreateWithFlags(C.byref(s0), 1), "stream")  # hipStreamNonBlocking
chk(hip.hipStreamCreateWithFlags(C.byref(s1), 1), "stream")
ev_fork, ev_join = vp(), vp()
chk(hip.hipEventCreateWithFlags(C.byref(ev_fork), 2), "event")  # hipEventDisableTiming
chk(hip.hipEventCreateWithFlags(C.byref(ev_join), 2), "event")
kDouble = 8


def group(stream, pieces=1):
    chk(nccl.ncclGroupStart(), "ncclGroupStart")
    n = N // pieces
    for p in range(pieces):
        chk(nccl.ncclRecv(vp(dst.value + 8 * n * p), n, kDouble, 0, comm, stream), "ncclRecv")
        chk(nccl.ncclSend(vp(src.value + 8 * n * p), n, kDouble, 0, comm, stream), "ncclSend")
    chk(nccl.ncclGroupEnd(), "ncclGroupEnd")


def payload(k):
    a = (C.c_double * N)(*[k + 1e-3 * i for i in range(N)])
    chk(hip.hipMemcpy(src, a, 8 * N, H2D), "upload")
    return a


def verify(a, what):
    b = (C.c_double * N)()
    chk(hip.hipMemcpy(b, dst, 8 * N, D2H), "download")
    if list(a) != list(b):
        raise RuntimeError("%s: payload mismatch" % what)


# warm-up outside any capture (RCCL sets up its channels on first use)
a = payload(0.0)
group(s0)
chk(hip.hipStreamSynchronize(s0), "sync")
verify(a, "warm-up")
print("warm-up group: ok", flush=True)

if mode == "eager":
    for k in range(1, 4):
        a = payload(float(k))
        group(s0)
        chk(hip.hipStreamSynchronize(s0), "sync")
        verify(a, "eager step %d" % k)
    print("RESULT %s/%s: ok" % (mode, stack), flush=True)
    sys.exit(0)

cap_mode = {"global": 0, "threadlocal": 1, "relaxed": 2}[
    "global" if mode.endswith("global") else "threadlocal" if mode.endswith("threadlocal") else "relaxed"]
print("begin capture (%s, mode %d) ..." % (mode, cap_mode), flush=True)
chk(hip.hipStreamBeginCapture(s0, cap_mode), "hipStreamBeginCapture")
if mode.startswith("origin"):
    group(s0)
elif mode == "two-groups-origin":
    group(s0)
    group(s0)
elif mode == "send-recv-split":
    group(s0, pieces=2)
elif mode.startswith("fork"):
    chk(hip.hipEventRecord(ev_fork, s0), "record fork")
    chk(hip.hipStreamWaitEvent(s1, ev_fork, 0), "wait fork")
    if mode == "fork-memcpy":
        chk(hip.hipMemcpyAsync(dst, src, 8 * N, D2D, s1), "memcpy")
    else:
        group(s1)
    chk(hip.hipEventRecord(ev_join, s1), "record join")
    chk(hip.hipStreamWaitEvent(s0, ev_join, 0), "wait join")
else:
    raise SystemExit("unknown mode " + mode)
print("operations recorded; end capture ...", flush=True)
graph = vp()
chk(hip.hipStreamEndCapture(s0, C.byref(graph)), "hipStreamEndCapture")
print("end capture: ok; instantiate ...", flush=True)
gexec = vp()
chk(hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0), "hipGraphInstantiate")
print("instantiate: ok; replay ...", flush=True)
for k in range(1, 4):
    a = payload(float(k))
    chk(hip.hipGraphLaunch(gexec, s0), "hipGraphLaunch")
    chk(hip.hipStreamSynchronize(s0), "sync")
    verify(a, "replay %d" % k)
print("RESULT %s/%s: ok" % (mode, stack), flush=True)
