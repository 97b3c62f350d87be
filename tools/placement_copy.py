#!/usr/bin/env python3
"""Diagnostic (not product): does a plain device-to-device copy see the placement classes too?  (read chunk i, write
chunk j, 1:1 -- if reads and writes inside one class collide, a copy is the most sensitive probe there is.)
usage: placement_copy.py [n_chunks=24] [copy_mib=512]"""
import ctypes as C, os, sys, time
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
SZ = (int(sys.argv[2]) if len(sys.argv) > 2 else 512) << 20
CH = 2 << 30
hip = C.CDLL("libamdhip64.so")


class Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]


class Prop(C.Structure):
    _fields_ = [("type", C.c_int), ("handle_type", C.c_int), ("location", Loc), ("win32", C.c_void_p),
                ("compression", C.c_ubyte), ("rdma", C.c_ubyte), ("usage", C.c_ushort)]


class Access(C.Structure):
    _fields_ = [("location", Loc), ("flags", C.c_int)]


def chk(rc, what):
    if rc != 0:
        raise SystemExit("%s failed: hipError %d" % (what, rc))


chk(hip.hipSetDevice(0), "set device")
prop = Prop(1, 0, Loc(1, 0), None, 0, 0, 0)
acc = Access(Loc(1, 0), 3)
va = C.c_void_p()
chk(hip.hipMemAddressReserve(C.byref(va), C.c_size_t(N * CH), C.c_size_t(1 << 30), None, C.c_ulonglong(0)), "reserve")
for i in range(N):
    h = C.c_void_p()
    chk(hip.hipMemCreate(C.byref(h), C.c_size_t(CH), C.byref(prop), C.c_ulonglong(0)), "create")
    chk(hip.hipMemMap(C.c_void_p(va.value + i * CH), C.c_size_t(CH), C.c_size_t(0), h, C.c_ulonglong(0)), "map")
chk(hip.hipMemSetAccess(va, C.c_size_t(N * CH), C.byref(acc), C.c_size_t(1)), "access")
chk(hip.hipMemset(va, 0, C.c_size_t(N * CH)), "memset")
e0, e1 = C.c_void_p(), C.c_void_p()
hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))


def copy_gbs(i, j, reps=6):
    src = C.c_void_p(va.value + i * CH)
    dst = C.c_void_p(va.value + j * CH + CH // 2)   # second half of chunk j: i == j does not overlap
    best = 0.0
    for _ in range(reps):
        hip.hipEventRecord(e0, None)
        hip.hipMemcpyAsync(dst, src, C.c_size_t(SZ), 3, None)
        hip.hipEventRecord(e1, None)
        hip.hipEventSynchronize(e1)
        ms = C.c_float()
        hip.hipEventElapsedTime(C.byref(ms), e0, e1)
        best = max(best, 2 * SZ / (ms.value * 1e6))
    return best


for _ in range(3):
    copy_gbs(0, 1, 10)
print("copy GB/s (read + written), rows = source chunk, columns = destination chunk; '.' = below 0.93 of the row's best")
M = [[copy_gbs(i, j) for j in range(N)] for i in range(N)]
for i in range(N):
    mx = max(M[i])
    print("%3d %s  min %5.0f max %5.0f" % (i, "".join("." if x < 0.93 * mx else "#" for x in M[i]), min(M[i]), mx))
