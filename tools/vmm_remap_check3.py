"""Diagnostic: reuse of the SAME virtual address after hipMemAddressFree (forced with the address hint), and whether
address hints are honoured at all (so that a process could hand out never-used addresses itself)."""
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
def create():
    h = C.c_void_p(); assert hip.hipMemCreate(C.byref(h), C.c_size_t(GIB), C.byref(prop), C.c_ulonglong(0)) == 0; return h
def reserve(hint=None):
    va = C.c_void_p()
    rc = hip.hipMemAddressReserve(C.byref(va), C.c_size_t(GIB), C.c_size_t(GIB), C.c_void_p(hint) if hint else None, C.c_ulonglong(0))
    return va if rc == 0 else None
def mapat(va, h):
    assert hip.hipMemMap(va, C.c_size_t(GIB), C.c_size_t(0), h, C.c_ulonglong(0)) == 0
    assert hip.hipMemSetAccess(va, C.c_size_t(GIB), C.byref(acc), C.c_size_t(1)) == 0
def peek(va):
    buf = (C.c_ubyte * 16)(); assert hip.hipMemcpy(buf, va, C.c_size_t(16), 2) == 0; return bytes(buf)[:4].hex()
A, B = create(), create()
r1 = reserve(); mapat(r1, A); hip.hipMemset(r1, 0x11, C.c_size_t(GIB)); hip.hipDeviceSynchronize()
hip.hipMemUnmap(r1, C.c_size_t(GIB)); hip.hipMemAddressFree(r1, C.c_size_t(GIB))
r2 = reserve(hint=r1.value)
print("reservation 1 at %#x (freed), reservation 2 asked for the same address: got %#x" % (r1.value, r2.value if r2 else 0))
if r2 and r2.value == r1.value:
    mapat(r2, B); hip.hipMemset(r2, 0x22, C.c_size_t(GIB)); hip.hipDeviceSynchronize()
    r3, r4 = reserve(), reserve()
    mapat(r3, A)
    hip.hipMemUnmap(r2, C.c_size_t(GIB)); mapat(r4, B)
    print("same address reused after hipMemAddressFree: A holds %s (want 11111111), B holds %s (want 22222222)" % (peek(r3), peek(r4)))
for hint in (0x200000000000, 0x400000000000, 0x100000000000, 0x7a0000000000):
    r = reserve(hint=hint)
    print("hint %#x -> %s" % (hint, ("%#x" % r.value) if r else "failed"))
