#!/usr/bin/env python3
"""tools/micro/ipc_probe.py: can a second process map this process's fine-grained (and plain) device allocation through hipIpcGetMemHandle /
hipIpcOpenMemHandle and write into it (same GPU, two processes: what a one-GPU box can check of the peer-mapped exchange)?"""
import ctypes, os, subprocess, sys
hip = ctypes.CDLL("libamdhip64.so")
HSZ = 64


class Handle(ctypes.Structure):          # hipIpcMemHandle_t: passed BY VALUE to hipIpcOpenMemHandle
    _fields_ = [("reserved", ctypes.c_char * HSZ)]


hip.hipIpcOpenMemHandle.argtypes = [ctypes.POINTER(ctypes.c_void_p), Handle, ctypes.c_uint]


def ck(rc, what):
    if rc != 0:
        hip.hipGetErrorString.restype = ctypes.c_char_p
        raise SystemExit("%s: %d %s" % (what, rc, hip.hipGetErrorString(rc)))


if len(sys.argv) > 1 and sys.argv[1] == "child":
    ck(hip.hipSetDevice(0), "setdevice")
    for i, hexh in enumerate(sys.argv[2:]):
        h = Handle.from_buffer_copy(bytes.fromhex(hexh))
        p = ctypes.c_void_p()
        ck(hip.hipIpcOpenMemHandle(ctypes.byref(p), h, 1), "open %d" % i)      # hipIpcMemLazyEnablePeerAccess = 1
        ck(hip.hipMemset(p, 0x5A + i, 4096), "memset %d" % i)
        ck(hip.hipDeviceSynchronize(), "sync")
        ck(hip.hipIpcCloseMemHandle(p), "close %d" % i)
    print("child wrote", flush=True)
    sys.exit(0)

ck(hip.hipSetDevice(0), "setdevice")
ptrs, handles = [], []
for flags in (None, 0x1):          # plain hipMalloc, hipDeviceMallocFinegrained
    p = ctypes.c_void_p()
    if flags is None: ck(hip.hipMalloc(ctypes.byref(p), 1 << 20), "malloc")
    else: ck(hip.hipExtMallocWithFlags(ctypes.byref(p), 1 << 20, flags), "extmalloc")
    ck(hip.hipMemset(p, 0, 1 << 20), "memset")
    h = (ctypes.c_char * HSZ)()
    ck(hip.hipIpcGetMemHandle(h, p), "gethandle flags=%s" % flags)
    ptrs.append(p); handles.append(bytes(h).hex())
ck(hip.hipDeviceSynchronize(), "sync")
r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"] + handles, capture_output=True, text=True, timeout=120)
print("child rc", r.returncode, r.stdout.strip(), r.stderr.strip()[-300:])
for i, p in enumerate(ptrs):
    buf = (ctypes.c_ubyte * 8)()
    ck(hip.hipMemcpy(buf, p, 8, 2), "d2h")
    print("allocation %d (%s): first bytes %s" % (i, "plain" if i == 0 else "fine-grained", bytes(buf).hex()))
