"""Diagnostic: (1) does a VA range that was FREED (hipMemAddressFree) and reserved again behave correctly with a new handle?
(2) when does physical memory go back to the driver: at hipMemRelease + hipMemUnmap, or only at hipMemAddressFree?"""
import ctypes as C
hip = C.CDLL("libamdhip64.so")
class Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]
class Prop(C.Structure):
    _fields_ = [("type", C.c_int), ("handle_type", C.c_int), ("location", Loc), ("win32", C.c_void_p), ("compression", C.c_ubyte), ("rdma", C.c_ubyte), ("usage", C.c_ushort)]
class Access(C.Structure):
    _fields_ = [("location", Loc), ("flags", C.c_int)]
hip.hipSetDevice(0)
prop = Prop(1, 0, Loc(1, 0), None, 0, 0, 0); acc = Access(Loc(1, 0), 3)
GIB = 1 << 30
def free_gib():
    f, t = C.c_size_t(), C.c_size_t(); hip.hipMemGetInfo(C.byref(f), C.byref(t)); return f.value / GIB
def create():
    h = C.c_void_p(); assert hip.hipMemCreate(C.byref(h), C.c_size_t(GIB), C.byref(prop), C.c_ulonglong(0)) == 0; return h
def reserve():
    va = C.c_void_p(); assert hip.hipMemAddressReserve(C.byref(va), C.c_size_t(GIB), C.c_size_t(GIB), None, C.c_ulonglong(0)) == 0; return va
def mapat(va, h):
    assert hip.hipMemMap(va, C.c_size_t(GIB), C.c_size_t(0), h, C.c_ulonglong(0)) == 0
    assert hip.hipMemSetAccess(va, C.c_size_t(GIB), C.byref(acc), C.c_size_t(1)) == 0
def peek(va):
    buf = (C.c_ubyte * 16)(); assert hip.hipMemcpy(buf, va, C.c_size_t(16), 2) == 0; return bytes(buf)[:4].hex()
print("free at start: %.2f GiB" % free_gib())
A, B = create(), create()
print("two handles created: %.2f GiB" % free_gib())
r1 = reserve(); mapat(r1, A); hip.hipMemset(r1, 0x11, C.c_size_t(GIB)); hip.hipDeviceSynchronize()
assert hip.hipMemUnmap(r1, C.c_size_t(GIB)) == 0
assert hip.hipMemAddressFree(r1, C.c_size_t(GIB)) == 0
r2 = reserve()
print("reservation 1 at %#x, freed; reservation 2 at %#x (%s)" % (r1.value, r2.value, "the SAME address" if r1.value == r2.value else "another address"))
mapat(r2, B); hip.hipMemset(r2, 0x22, C.c_size_t(GIB)); hip.hipDeviceSynchronize()
r3 = reserve(); mapat(r3, A)
print("A holds %s (want 11111111), B (through reservation 2) holds %s (want 22222222)" % (peek(r3), peek(r2)))
hip.hipMemUnmap(r2, C.c_size_t(GIB)); hip.hipMemUnmap(r3, C.c_size_t(GIB))
print("both unmapped: %.2f GiB" % free_gib())
hip.hipMemRelease(A); hip.hipMemRelease(B)
print("both released (reservations still held): %.2f GiB" % free_gib())
hip.hipMemAddressFree(r2, C.c_size_t(GIB)); hip.hipMemAddressFree(r3, C.c_size_t(GIB))
print("reservations freed: %.2f GiB" % free_gib())
