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
def reserve():
    va = C.c_void_p(); assert hip.hipMemAddressReserve(C.byref(va), C.c_size_t(GIB), C.c_size_t(GIB), None, C.c_ulonglong(0)) == 0; return va
def mapat(va, h):
    assert hip.hipMemMap(va, C.c_size_t(GIB), C.c_size_t(0), h, C.c_ulonglong(0)) == 0
    assert hip.hipMemSetAccess(va, C.c_size_t(GIB), C.byref(acc), C.c_size_t(1)) == 0
def unmap(va):
    assert hip.hipMemUnmap(va, C.c_size_t(GIB)) == 0
def peek(va, off=0):
    buf = (C.c_ubyte * 16)()
    assert hip.hipMemcpy(buf, C.c_void_p(va.value + off), C.c_size_t(16), 2) == 0
    return bytes(buf)[:4].hex()
stream = C.c_void_p(); hip.hipStreamCreateWithFlags(C.byref(stream), 1)
A, B = create(), create()
scratch, home_a, home_b = reserve(), reserve(), reserve()
for trial, use_stream in enumerate((None, stream)):
    mapat(scratch, A); hip.hipMemsetAsync(scratch, 0x11, C.c_size_t(GIB), use_stream); hip.hipDeviceSynchronize(); unmap(scratch)
    mapat(scratch, B); hip.hipMemsetAsync(scratch, 0x22, C.c_size_t(GIB), use_stream); hip.hipDeviceSynchronize(); unmap(scratch)
    mapat(home_a, A); mapat(home_b, B)
    print("trial %d (%s stream): A holds %s (want 11111111), B holds %s (want 22222222); A end %s B end %s" % (
        trial, "null" if use_stream is None else "non-blocking", peek(home_a), peek(home_b), peek(home_a, GIB - 16), peek(home_b, GIB - 16)))
    unmap(home_a); unmap(home_b)
# the arena's sequence: A mapped in a "column" and filled by a kernel on a non-blocking stream; B labelled through scratch afterwards
mapat(home_a, A)
mapat(scratch, A); hip.hipMemsetAsync(scratch, 0, C.c_size_t(GIB), None); hip.hipDeviceSynchronize(); unmap(scratch)   # A labelled
hip.hipMemsetAsync(home_a, 0x33, C.c_size_t(GIB), stream)                                                                # A's data (async)
mapat(scratch, B); hip.hipMemsetAsync(scratch, 0, C.c_size_t(GIB), None); hip.hipDeviceSynchronize(); unmap(scratch)   # B labelled
hip.hipDeviceSynchronize()
bad = 0
for off in range(0, GIB, 1 << 24):
    if peek(home_a, off) != "33333333": bad += 1
print("arena sequence: %d of 64 sampled places of A lost their data" % bad)
