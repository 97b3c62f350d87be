//go:build knoxgpu

// Package knoxgpu binds libknoxgpu.so (include/knoxgpu.h): the MI355X implementation of the KnoxDB pack-scan path.
// Drop into internal/knoxgpu/ of blockwatch-cc/knoxdb.  NOT compiled in the knoxdb_amd repository (no Go toolchain
// there): integration/check_go_calls.py checks every C call below against the header (names, arity, argument kinds),
// tests/cpp/test_host.cpp drives the same entry points on the GPU.
//
// cgo rules observed: Go memory passed to C is only touched during the call (the library stages it through pinned
// memory and never retains host pointers, include/knoxgpu.h "Conventions"); C handles are plain pointers owned by Go
// wrapper structs with explicit Free / Close.
package knoxgpu

/*
#cgo CFLAGS: -I${SRCDIR}/include
#cgo LDFLAGS: -L${SRCDIR}/lib -lknoxgpu
#include <stdlib.h>
#include "knoxgpu.h"
*/
import "C"

import (
	"encoding/binary"
	"errors"
	"math"
	"os"
	"strconv"
	"strings"
	"sync"
	"unicode"
	"unsafe"
)

// Enabled follows the kill-switch pattern of internal/cpu/avx.go:43-89 (env NOAVX / NOAVX2 ...): KNOX_NOGPU=1 (any
// strconv.ParseBool true) keeps every CPU kernel in place.
func Enabled() bool {
	v := os.Getenv("KNOX_NOGPU")
	if v == "" {
		return true
	}
	off, err := strconv.ParseBool(v)
	return err != nil || !off
}

type Number interface {
	int64 | int32 | int16 | int8 | uint64 | uint32 | uint16 | uint8 | float64 | float32
}

// Ctx is one device + one HIP stream.  The library's contexts are not re-entrant; the reference runs many queries
// concurrently (one goroutine per query, pack/table/query.go:195-219), so EVERY call that reaches the context -- also
// the Free of a column / bitset / filter, which synchronises the context's stream -- takes c.mu
// (integration/check_go_calls.py enforces it).  A host that wants parallel scans opens one Ctx per worker.
type Ctx struct {
	p  *C.knox_ctx
	mu sync.Mutex
}

func errOf(rc C.int) error {
	if rc == 0 {
		return nil
	}
	return errors.New("knoxgpu: " + C.GoString(C.knox_strerror(rc)))
}

func Open(device int) (*Ctx, error) {
	var c *C.knox_ctx
	if rc := C.knox_ctx_open(C.int(device), &c); rc != 0 {
		return nil, errOf(rc)
	}
	return &Ctx{p: c}, nil
}

func (c *Ctx) Close() {
	c.mu.Lock()
	defer c.mu.Unlock()
	C.knox_ctx_close(c.p)
}

// SetArena sets the budget of the context's device-memory arena (the pack cache on the device).
func (c *Ctx) SetArena(bytes uint64) error {
	c.mu.Lock()
	defer c.mu.Unlock()
	return errOf(C.knox_ctx_set_arena(c.p, C.size_t(bytes)))
}

// Reserve makes room, in ONE placement class of the device memory, for columns that will be scanned together (the filter
// columns of a run): two read streams of one scan run 4-5 % faster there (include/knoxgpu.h knox_ctx_reserve).  Best effort.
func (c *Ctx) Reserve(bytes uint64) error {
	c.mu.Lock()
	defer c.mu.Unlock()
	return errOf(C.knox_ctx_reserve(c.p, C.size_t(bytes)))
}

// ProfEnable / ProfRead: the library's HIP-event stopwatch around the dominant kernel of every scan (knox_prof_*):
// total milliseconds and launches since the last read -- what the run reader writes into QueryStats.
func (c *Ctx) ProfEnable(on bool) {
	c.mu.Lock()
	defer c.mu.Unlock()
	v := 0
	if on {
		v = 1
	}
	C.knox_prof_enable(c.p, C.int(v))
}

func (c *Ctx) ProfRead() (ms float64, launches int) {
	c.mu.Lock()
	defer c.mu.Unlock()
	var total C.double
	var k C.int
	C.knox_prof_read(c.p, &total, &k)
	return float64(total), int(k)
}

// ValueBits: the raw bits of a filter value of the column type, zero-extended -- what knox_pred.a / .b carry.
func ValueBits(v any) (uint64, bool) {
	switch x := v.(type) {
	case int64:
		return uint64(x), true
	case int32:
		return uint64(uint32(x)), true
	case int16:
		return uint64(uint16(x)), true
	case int8:
		return uint64(uint8(x)), true
	case uint64:
		return x, true
	case uint32:
		return uint64(x), true
	case uint16:
		return uint64(x), true
	case uint8:
		return uint64(x), true
	case float64:
		return math.Float64bits(x), true
	case float32:
		return uint64(math.Float32bits(x)), true
	}
	return 0, false
}

// AsU64 reinterprets the signed top limb slice of an Int128Stride / Int256Stride (X0 []int64) as raw 64-bit words.
func AsU64(x []int64) []uint64 {
	return unsafe.Slice((*uint64)(unsafe.Pointer(unsafe.SliceData(x))), len(x))
}

// Cmp has the shape of the cmp func vars (internal/cmp/cmp.go:6-114): dtype = types.BlockType, mode =
// types.FilterMode, bits = (len(src)+7)/8 zeroed bytes.  ok == false: the GPU call failed, use the CPU kernel.
func Cmp[T Number](c *Ctx, dtype, mode int, src []T, a, b T, bits []byte) (int64, bool) {
	if len(src) == 0 {
		return 0, true // n == 0 returns 0 and writes nothing (cmp/tests/int64.go:212-213)
	}
	c.mu.Lock()
	defer c.mu.Unlock()
	n := C.knox_cmp(c.p, C.int(dtype), C.int(mode), unsafe.Pointer(unsafe.SliceData(src)), C.size_t(len(src)),
		unsafe.Pointer(&a), unsafe.Pointer(&b), (*C.uint8_t)(unsafe.SliceData(bits)))
	return int64(n), n >= 0
}

// CmpMinRows is the row floor of the rebound cmp func vars: a vector shorter than this stays on the kernel that was
// installed before (AVX2 / AVX-512 / generic).  A cmp call is a synchronous round trip -- lock, launch, one stream
// synchronisation, a copy of the bitset -- and below the floor one CPU core answers sooner: the reference's AVX2 kernel
// takes 4.5-7.4 us for a 2^16-row pack (docs/bench/2020-10-d3f9115/Xeon-E-2176G/README.md:54,63), the device call ~25 us
// plus the PCIe time of the vector.  Default = the crossover measured on the MI355X box with the vector in page-locked
// memory (tools/cmp_latency.py, DESIGN.md 7): 2^23 rows -- below it one AVX2 core is 1.04-7 x faster, above it PCIe (~6 G
// rows/s) and one core's memory stream (~5.7 G rows/s) are level, so the rebinding is plumbing, not a speed-up.  KNOX_CMP_MIN_ROWS overrides it (0 = every call goes to the
// device: tests).  The throughput path is not this one: runs of packs resident in HBM go through table.GpuRun.
var CmpMinRows = func() int {
	if v, err := strconv.Atoi(os.Getenv("KNOX_CMP_MIN_ROWS")); err == nil && v >= 0 {
		return v
	}
	return 1 << 23
}()

// Pool hands out contexts to calls that would otherwise serialise on one: the reference runs many queries at once (one
// goroutine per query, internal/tests/benchmarks/benchmark3_test.go:112-127 b.RunParallel), and a context is one HIP
// stream plus one mutex.  Get blocks while every context is out; a context still takes its own c.mu inside every call (the
// rule integration/check_go_calls.py enforces), Pool only decides WHICH context a call uses.  Every Get is paired with a
// deferred Put in the same function (checked the same way).
type Pool struct {
	free chan *Ctx
	all  []*Ctx
}

// NewPool opens n contexts on `device` (n < 1: one); fewer if the device refuses, an error if none opens.
func NewPool(device, n int) (*Pool, error) {
	if n < 1 {
		n = 1
	}
	p := &Pool{free: make(chan *Ctx, n)}
	var last error
	for i := 0; i < n; i++ {
		c, err := Open(device)
		if err != nil {
			last = err
			break
		}
		p.all = append(p.all, c)
		p.free <- c
	}
	if len(p.all) == 0 {
		return nil, last
	}
	return p, nil
}

func (p *Pool) Get() *Ctx  { return <-p.free }
func (p *Pool) Put(c *Ctx) { p.free <- c }
func (p *Pool) Size() int  { return len(p.all) }

// Close closes every context; the caller makes sure none is out.
func (p *Pool) Close() {
	for _, c := range p.all {
		c.Close()
	}
	p.all = nil
}

// PoolSize: KNOX_CMP_CONTEXTS (default 4) contexts serve the rebound cmp func vars.
func PoolSize() int {
	if v, err := strconv.Atoi(os.Getenv("KNOX_CMP_CONTEXTS")); err == nil && v > 0 {
		return v
	}
	return 4
}

// Column is the device image of one column of a run of consecutive packs.
type Column struct {
	p *C.knox_col
	n int
	c *Ctx
}

func Upload[T Number](c *Ctx, dtype int, vals []T, packRows int) (*Column, error) {
	c.mu.Lock()
	defer c.mu.Unlock()
	var col *C.knox_col
	rc := C.knox_col_upload(c.p, C.int(dtype), unsafe.Pointer(unsafe.SliceData(vals)), C.size_t(len(vals)), C.size_t(packRows), &col)
	if rc != 0 {
		return nil, errOf(rc)
	}
	return &Column{col, len(vals), c}, nil
}

// NewColumn / AppendAsync / Reset / Wait: the ingest that overlaps the scans (include/knoxgpu.h knox_col_create ...):
// an empty numeric column with room for capacityRows rows; AppendAsync enqueues the copy of one pack's vector behind
// the rows present on the context's copy stream and returns (vals is a Go slice: it is copied into the library's pinned
// ring inside the call and not referenced afterwards -- cgo rule); every scan that reads the column waits for the
// copies ON THE DEVICE.  Reset empties the column for the next run (call it before enqueueing the other buffer's scan).
func NewColumn(c *Ctx, dtype, capacityRows, packRows int) (*Column, error) {
	c.mu.Lock()
	defer c.mu.Unlock()
	var col *C.knox_col
	rc := C.knox_col_create(c.p, C.int(dtype), C.size_t(capacityRows), C.size_t(packRows), &col)
	if rc != 0 {
		return nil, errOf(rc)
	}
	return &Column{col, 0, c}, nil
}

func AppendAsync[T Number](col *Column, vals []T) error {
	c := col.c
	c.mu.Lock()
	defer c.mu.Unlock()
	rc := C.knox_col_append_async(c.p, col.p, unsafe.Pointer(unsafe.SliceData(vals)), C.size_t(len(vals)))
	if rc == 0 {
		col.n += len(vals)
	}
	return errOf(rc)
}

func (col *Column) Reset() error {
	c := col.c
	c.mu.Lock()
	defer c.mu.Unlock()
	rc := C.knox_col_reset(c.p, col.p)
	if rc == 0 {
		col.n = 0
	}
	return errOf(rc)
}

func (col *Column) Wait() error {
	c := col.c
	c.mu.Lock()
	defer c.mu.Unlock()
	return errOf(C.knox_col_wait(c.p, col.p))
}

// UploadRawContainers: one column from the ENCODED blocks of consecutive packs (the bytes NumberContainer.Store writes,
// internal/encode/container.go:84): headers are parsed on the host, payloads expanded on the device -- no host decode.
// The Go buffers are copied into C memory first (a C array of pointers must not point into Go memory).
func UploadRawContainers(c *Ctx, dtype int, bufs [][]byte, packRows int) (*Column, error) {
	c.mu.Lock()
	defer c.mu.Unlock()
	n := len(bufs)
	ptrs := C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0))))
	lens := C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(C.size_t(0))))
	pv := unsafe.Slice((**C.uint8_t)(ptrs), n)
	lv := unsafe.Slice((*C.size_t)(lens), n)
	defer func() {
		for _, q := range pv {
			C.free(unsafe.Pointer(q))
		}
		C.free(ptrs)
		C.free(lens)
	}()
	rows := 0
	for i, b := range bufs {
		pv[i] = (*C.uint8_t)(C.CBytes(b))
		lv[i] = C.size_t(len(b))
	}
	var col *C.knox_col
	rc := C.knox_col_upload_raw_containers(c.p, C.int(dtype), (**C.uint8_t)(ptrs), (*C.size_t)(lens), C.size_t(n), C.size_t(packRows), &col)
	if rc != 0 {
		return nil, errOf(rc)
	}
	rows = int(C.knox_col_rows(col))
	return &Column{col, rows, c}, nil
}

// NewBytesColumn / AppendPack: a []byte column held as ONE StringPool per pack (pkg/stringx/pool.go:45-58):
// ptr[i] = off<<32 | len relative to that pack's own buffer.
func NewBytesColumn(c *Ctx, packRows, maxPacks int, maxPoolBytes uint64) (*Column, error) {
	c.mu.Lock()
	defer c.mu.Unlock()
	var col *C.knox_col
	rc := C.knox_col_bytes_create(c.p, C.size_t(packRows), C.size_t(maxPacks), C.size_t(maxPoolBytes), &col)
	if rc != 0 {
		return nil, errOf(rc)
	}
	return &Column{col, 0, c}, nil
}

func (col *Column) AppendPack(ptr []uint64, buf []byte) error {
	c := col.c
	c.mu.Lock()
	defer c.mu.Unlock()
	rc := C.knox_col_bytes_append_pack(c.p, col.p, (*C.uint64_t)(unsafe.SliceData(ptr)), C.size_t(len(ptr)),
		(*C.uint8_t)(unsafe.SliceData(buf)), C.size_t(len(buf)))
	if rc == 0 {
		col.n += len(ptr)
	}
	return errOf(rc)
}

func (col *Column) Rows() int { return col.n }

// Free synchronises the context's stream before the memory goes back to the arena: under the context's lock.
func (col *Column) Free() {
	c := col.c
	c.mu.Lock()
	defer c.mu.Unlock()
	C.knox_col_free(col.p)
}

var (
	defaultOnce sync.Once
	defaultCtx  *Ctx
)

// Default is the process-wide context on device 0 that the package-level hooks share (the Matcher shims, the arena's
// page-locked allocator); nil when the GPU is switched off (KNOX_NOGPU) or cannot be opened.
func Default() *Ctx {
	defaultOnce.Do(func() {
		if Enabled() {
			if c, err := Open(0); err == nil {
				defaultCtx = c
			}
		}
	})
	return defaultCtx
}

// HostAlloc hands out page-locked host memory (hipHostMalloc) owned by the library: block vectors built inside it are
// read by the DMA engine in place (Upload / AppendAsync at PCIe rate, no staging copy).  C memory: the Go collector
// neither moves nor frees it; HostFree gives it back.
func (c *Ctx) HostAlloc(bytes int) (unsafe.Pointer, error) {
	c.mu.Lock()
	defer c.mu.Unlock()
	var p unsafe.Pointer
	if rc := C.knox_host_alloc(c.p, C.size_t(bytes), &p); rc != 0 {
		return nil, errOf(rc)
	}
	return p, nil
}

func (c *Ctx) HostFree(p unsafe.Pointer) error {
	c.mu.Lock()
	defer c.mu.Unlock()
	return errOf(C.knox_host_free(c.p, p))
}

// UploadPool puts ONE StringPool (pkg/stringx/pool.go:45-58: ptr[i] = off<<32 | len into buf) on the device as a bytes
// column of one pack.
func UploadPool(c *Ctx, ptr []uint64, buf []byte) (*Column, error) {
	col, err := NewBytesColumn(c, (len(ptr)+63)&^63, 1, uint64(len(buf))) // pack sizes are multiples of 64 rows; the only pack may be short
	if err != nil {
		return nil, err
	}
	if err = col.AppendPack(ptr, buf); err != nil {
		col.Free()
		return nil, err
	}
	return col, nil
}

// RegexpUnicodeVersion: the Unicode version of the tables behind \p{..} and (?i) in the library's regexp compiler.
func RegexpUnicodeVersion() string {
	return C.GoString(C.knox_regexp_unicode_version())
}

// RegexpOnDevice reports whether the library's matcher returns what THIS binary's regexp package returns for the
// pattern.  With equal Unicode versions: always (anything else the compiler refuses with KNOX_E_UNSUPPORTED and the
// caller keeps re.Match).  With different versions a pattern that consults the tables stays on the host: \p / \P
// classes, and case folding in a pattern that is not plain ASCII (the orbits of the ASCII letters -- K k U+212A,
// S s U+017F -- have not changed since Unicode 5.1).
func RegexpOnDevice(pattern string) bool {
	if unicode.Version == RegexpUnicodeVersion() {
		return true
	}
	if strings.Contains(pattern, `\p`) || strings.Contains(pattern, `\P`) {
		return false
	}
	fold := false
	for i := 0; i+2 < len(pattern); i++ {
		if pattern[i] == '(' && pattern[i+1] == '?' {
			for j := i + 2; j < len(pattern) && pattern[j] != ':' && pattern[j] != ')' && pattern[j] != '-'; j++ {
				fold = fold || pattern[j] == 'i'
			}
		}
	}
	if !fold {
		return true
	}
	for i := 0; i < len(pattern); i++ {
		if pattern[i] >= 0x80 {
			return false
		}
	}
	return !strings.Contains(pattern, `\x`) // an escaped rune may be anything
}

// MatchRegexp = bytesRegexpMatcher.MatchVector (operator/filter/match_bytes.go:587-609) over a bytes column: bits.Set(i)
// for every (masked) row the pattern matches; returns the popcount of bits afterwards.
func MatchRegexp(c *Ctx, col *Column, pattern string, mask, bits *Bits) (int64, error) {
	c.mu.Lock()
	defer c.mu.Unlock()
	pat := C.CBytes([]byte(pattern))
	defer C.free(pat)
	var mp *C.knox_bits
	if mask != nil {
		mp = mask.p
	}
	var n C.int64_t
	rc := C.knox_match_regexp(c.p, col.p, (*C.uint8_t)(pat), C.uint64_t(len(pattern)), mp, bits.p, &n)
	return int64(n), errOf(rc)
}

// ArenaClasses: column space mapped per placement class and the number of classes met (diagnostics).
func (c *Ctx) ArenaClasses() (classBytes [4]uint64, seen int) {
	c.mu.Lock()
	defer c.mu.Unlock()
	cb := make([]C.uint64_t, 4)
	var n C.int
	if C.knox_ctx_arena_classes(c.p, &cb[0], &n) != 0 {
		return classBytes, 0
	}
	for i := range cb {
		classBytes[i] = uint64(cb[i])
	}
	return classBytes, int(n)
}

// Bits is a device-resident bitset.Bitset.
type Bits struct {
	p    *C.knox_bits
	size int
	c    *Ctx
}

func NewBits(c *Ctx, size int) (*Bits, error) {
	c.mu.Lock()
	defer c.mu.Unlock()
	var b *C.knox_bits
	if rc := C.knox_bits_alloc(c.p, C.size_t(size), &b); rc != 0 {
		return nil, errOf(rc)
	}
	return &Bits{b, size, c}, nil
}

// NewBitsFor: the selection bitset of scans over col (a write buffer verified against that column).
func NewBitsFor(c *Ctx, size int, col *Column) (*Bits, error) {
	c.mu.Lock()
	defer c.mu.Unlock()
	var b *C.knox_bits
	if rc := C.knox_bits_alloc_for(c.p, C.size_t(size), col.p, &b); rc != 0 {
		return nil, errOf(rc)
	}
	return &Bits{b, size, c}, nil
}

func (b *Bits) Size() int { return b.size }

func (b *Bits) Free() {
	c := b.c
	c.mu.Lock()
	defer c.mu.Unlock()
	C.knox_bits_free(b.p)
}

func (b *Bits) Upload(host []byte) error {
	c := b.c
	c.mu.Lock()
	defer c.mu.Unlock()
	return errOf(C.knox_bits_upload(c.p, b.p, (*C.uint8_t)(unsafe.SliceData(host))))
}

// Download fills host ((size+7)/8 bytes) -- what bitset.Bitset.Bytes() holds.
func (b *Bits) Download(host []byte) error {
	c := b.c
	c.mu.Lock()
	defer c.mu.Unlock()
	return errOf(C.knox_bits_download(c.p, b.p, (*C.uint8_t)(unsafe.SliceData(host))))
}

// Indexes = Bitset.Indexes (internal/bitset/iterator.go:269-290): ascending row ids of the set bits.
func (b *Bits) Indexes(dst []uint32) ([]uint32, error) {
	c := b.c
	c.mu.Lock()
	defer c.mu.Unlock()
	var n C.size_t
	rc := C.knox_bits_indexes(c.p, b.p, (*C.uint32_t)(unsafe.SliceData(dst)), C.size_t(cap(dst)), &n)
	if rc != 0 {
		return nil, errOf(rc)
	}
	return dst[:int(n)], nil
}

// MatchWide runs one Int128 / Int256 compare (cmp_i128_* / cmp_i256_*, internal/cmp/int128.go:14-471, int256.go:14-270)
// over host limb slices, most significant first: uploads the limbs (and the mask), matches on the device, downloads
// the bitset.  ok == false: fall back to the CPU kernel.  The returned count is the popcount of the result (the
// reference's unrolled i128 loops count the unmasked byte: see include/knoxgpu.h knox_match_wide).
func MatchWide(c *Ctx, limbs [][]uint64, mode int, a, b []uint64, mask, bits []byte, n int) (int64, bool) {
	if n == 0 {
		return 0, true
	}
	c.mu.Lock()
	defer c.mu.Unlock()
	cols := make([]*C.knox_col, len(limbs))
	defer func() {
		for _, p := range cols {
			if p != nil {
				C.knox_col_free(p)
			}
		}
	}()
	const blockUint64 = 5
	for i, l := range limbs {
		if rc := C.knox_col_upload(c.p, C.int(blockUint64), unsafe.Pointer(unsafe.SliceData(l)), C.size_t(n), C.size_t(0), &cols[i]); rc != 0 {
			return 0, false
		}
	}
	var dbits, dmask *C.knox_bits
	if rc := C.knox_bits_alloc(c.p, C.size_t(n), &dbits); rc != 0 {
		return 0, false
	}
	defer C.knox_bits_free(dbits)
	if mask != nil {
		if rc := C.knox_bits_alloc(c.p, C.size_t(n), &dmask); rc != 0 {
			return 0, false
		}
		defer C.knox_bits_free(dmask)
		if rc := C.knox_bits_upload(c.p, dmask, (*C.uint8_t)(unsafe.SliceData(mask))); rc != 0 {
			return 0, false
		}
	}
	var bp *C.uint64_t
	if b != nil {
		bp = (*C.uint64_t)(unsafe.SliceData(b))
	}
	var cnt C.int64_t
	rc := C.knox_match_wide(c.p, (**C.knox_col)(unsafe.SliceData(cols)), C.int(len(limbs)), C.int(mode),
		(*C.uint64_t)(unsafe.SliceData(a)), bp, dmask, dbits, &cnt)
	if rc != 0 {
		return 0, false
	}
	if rc := C.knox_bits_download(c.p, dbits, (*C.uint8_t)(unsafe.SliceData(bits))); rc != 0 {
		return 0, false
	}
	return int64(cnt), true
}

// ---- the tree boundary: filter.Match over every resident pack + the reducers ---------------------------------

// Pred is one postfix node of a filter.Node tree (knox_pred, include/knoxgpu.h).
type Pred struct {
	Kind, NChild, Col, Op int
	A, B                  uint64 // value bits of the column type (B: Range upper bound)
	Val, Val2             []byte // bytes leaves / IN sets
}

const (
	NodeLeaf, NodeAnd, NodeOr = 0, 1, 2
)

// SetBytes marshals an IN / NOT IN set the way knox_pred.val takes it: the values of the column type back to back
// (numInSetMatcher.Value() / floatInSetMatcher: a []T, match_num.go:602-615), or for a bytes column a packed list of
// [uint32 little-endian length][bytes] entries (bytesInSetMatcher, match_bytes.go:300-540: a [][]byte).
func SetBytes(v any) ([]byte, bool) {
	switch s := v.(type) {
	case [][]byte:
		n := 0
		for _, e := range s {
			n += 4 + len(e)
		}
		out := make([]byte, 0, n)
		for _, e := range s {
			out = binary.LittleEndian.AppendUint32(out, uint32(len(e)))
			out = append(out, e...)
		}
		return out, true
	case []int64:
		return numSetBytes(s), true
	case []int32:
		return numSetBytes(s), true
	case []int16:
		return numSetBytes(s), true
	case []int8:
		return numSetBytes(s), true
	case []uint64:
		return numSetBytes(s), true
	case []uint32:
		return numSetBytes(s), true
	case []uint16:
		return numSetBytes(s), true
	case []uint8:
		return numSetBytes(s), true
	case []float64:
		return numSetBytes(s), true
	case []float32:
		return numSetBytes(s), true
	}
	return nil, false
}

func numSetBytes[T Number](s []T) []byte {
	if len(s) == 0 {
		return nil
	}
	return unsafe.Slice((*byte)(unsafe.Pointer(unsafe.SliceData(s))), len(s)*int(unsafe.Sizeof(s[0])))
}

// Reducer codes follow internal/reducer/reducer.go:25-45.
const (
	RSum, RFirst, RLast, RMin, RMax, RMean, RVar, RStd, RCount = 0, 1, 2, 3, 4, 5, 6, 7, 8
)

type Agg struct{ Func, Col int }

// State is knox_agg_state: the mergeable partial aggregate of one column.
type State struct{ c C.knox_agg_state }

type ScanResult struct {
	Count, RowsScanned int64
	PackCounts         []int64
	States             []State
}

// Scan evaluates the tree over rows [0, n) of cols (one fused pass where it can), writes the selection bitset into
// bitsOut (nullable), and folds the reducers.  packSkip (nullable): 1 = pack pruned by the caller's statistics.
func (c *Ctx) Scan(nodes []Pred, cols []*Column, aggs []Agg, rowBase uint64, packSkip []byte, bitsOut *Bits, wantPackCounts bool, nPacks int) (*ScanResult, error) {
	c.mu.Lock()
	defer c.mu.Unlock()
	cn := make([]C.knox_pred, len(nodes))
	var pin []unsafe.Pointer // C copies of the byte values: a C struct must not hold Go pointers (cgo rule)
	defer func() {
		for _, p := range pin {
			C.free(p)
		}
	}()
	cbytes := func(v []byte) (*C.uint8_t, C.uint64_t) {
		if len(v) == 0 {
			return nil, 0
		}
		p := C.CBytes(v)
		pin = append(pin, p)
		return (*C.uint8_t)(p), C.uint64_t(len(v))
	}
	for i, n := range nodes {
		cn[i].kind, cn[i].nchild, cn[i].col, cn[i].op = C.int32_t(n.Kind), C.int32_t(n.NChild), C.int32_t(n.Col), C.int32_t(n.Op)
		cn[i].a, cn[i].b = C.uint64_t(n.A), C.uint64_t(n.B)
		cn[i].val, cn[i].val_len = cbytes(n.Val)
		cn[i].val2, cn[i].val2_len = cbytes(n.Val2)
	}
	cc := make([]*C.knox_col, len(cols))
	for i, col := range cols {
		cc[i] = col.p
	}
	ca := make([]C.knox_agg, len(aggs)+1)
	for i, a := range aggs {
		ca[i]._func, ca[i].col = C.int32_t(a.Func), C.int32_t(a.Col)
	}
	res := &ScanResult{States: make([]State, len(aggs))}
	var pc *C.int64_t
	if wantPackCounts {
		res.PackCounts = make([]int64, nPacks)
		pc = (*C.int64_t)(unsafe.SliceData(res.PackCounts))
	}
	var skip *C.uint8_t
	if packSkip != nil {
		skip = (*C.uint8_t)(unsafe.SliceData(packSkip))
	}
	var bo *C.knox_bits
	if bitsOut != nil {
		bo = bitsOut.p
	}
	var st *C.knox_agg_state
	if len(aggs) > 0 {
		st = (*C.knox_agg_state)(unsafe.Pointer(unsafe.SliceData(res.States)))
	}
	var out C.knox_scan_out
	rc := C.knox_scan(c.p, (*C.knox_pred)(unsafe.SliceData(cn)), C.int(len(cn)), (**C.knox_col)(unsafe.SliceData(cc)), C.int(len(cc)),
		(*C.knox_agg)(unsafe.SliceData(ca)), C.int(len(aggs)), C.uint64_t(rowBase), skip, bo, pc, st, &out)
	if rc != 0 {
		return nil, errOf(rc)
	}
	res.Count, res.RowsScanned = int64(out.count), int64(out.rows_scanned)
	return res, nil
}

// Value is Reducer.Value() for reducer fn over a state: the value's raw bits (of the column type), seen, row.
func (s *State) Value(fn int) (bits uint64, seen bool, row int64) {
	var v C.uint64_t
	var sn C.int
	var r C.int64_t
	C.knox_agg_value(&s.c, C.int(fn), unsafe.Pointer(&v), &sn, &r)
	return uint64(v), sn != 0, int64(r)
}

// Combine merges states of consecutive row ranges (rank order = row order).
func Combine(states []State) State {
	var out State
	C.knox_agg_combine((*C.knox_agg_state)(unsafe.Pointer(unsafe.SliceData(states))), C.int(len(states)), &out.c)
	return out
}

// ---- bit-packed columns scanned as they are (internal/encode/int_bitpack.go, bitpack/cmp_*.go) ----------------------

// Packed is a column whose packs stay bit-packed in HBM: BitpackContainer.MatchEqual ... MatchBetween (+ count, sum, per-pack
// counts) for a whole run in one call, and the form in which a table that is NOT resident crosses PCIe (log2 / 64 of the bytes).
type Packed struct {
	p *C.knox_packed
	c *Ctx
}

// cBufs copies Go buffers into C memory and builds the C arrays knox_* container entry points take (a C array of pointers
// must not point into Go memory); free() releases everything.
func cBufs(bufs [][]byte) (ptrs unsafe.Pointer, lens unsafe.Pointer, free func()) {
	n := len(bufs)
	ptrs = C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0))))
	lens = C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(C.size_t(0))))
	pv := unsafe.Slice((**C.uint8_t)(ptrs), n)
	lv := unsafe.Slice((*C.size_t)(lens), n)
	for i, b := range bufs {
		pv[i] = (*C.uint8_t)(C.CBytes(b))
		lv[i] = C.size_t(len(b))
	}
	return ptrs, lens, func() {
		for _, q := range pv {
			C.free(unsafe.Pointer(q))
		}
		C.free(ptrs)
		C.free(lens)
	}
}

// NewPacked: room for maxPacks packs / maxPayloadBytes of packed words; filled by AppendAsync on the context's copy stream.
func NewPacked(c *Ctx, dtype, packRows, maxPacks int, maxPayloadBytes uint64) (*Packed, error) {
	c.mu.Lock()
	defer c.mu.Unlock()
	var p *C.knox_packed
	rc := C.knox_packed_create(c.p, C.int(dtype), C.size_t(packRows), C.size_t(maxPacks), C.size_t(maxPayloadBytes), &p)
	if rc != 0 {
		return nil, errOf(rc)
	}
	return &Packed{p, c}, nil
}

// AppendAsync enqueues the [4] containers of the next packs (NumberContainer.Store bytes) and returns.
func (pk *Packed) AppendAsync(bufs [][]byte) error {
	c := pk.c
	c.mu.Lock()
	defer c.mu.Unlock()
	ptrs, lens, free := cBufs(bufs)
	defer free()
	return errOf(C.knox_packed_append_async(c.p, pk.p, (**C.uint8_t)(ptrs), (*C.size_t)(lens), C.size_t(len(bufs))))
}

func (pk *Packed) Reset() error {
	c := pk.c
	c.mu.Lock()
	defer c.mu.Unlock()
	return errOf(C.knox_packed_reset(c.p, pk.p))
}

// Scan: one numeric leaf on the packed words: count, wrapping sum state, selection bitset (nullable), per-pack counts.
func (pk *Packed) Scan(mode int, a, b uint64, wantSum bool, bitsOut *Bits, packCounts []int64) (int64, State, error) {
	c := pk.c
	c.mu.Lock()
	defer c.mu.Unlock()
	var bo *C.knox_bits
	if bitsOut != nil {
		bo = bitsOut.p
	}
	var pc *C.int64_t
	if packCounts != nil {
		pc = (*C.int64_t)(unsafe.SliceData(packCounts))
	}
	ws := 0
	if wantSum {
		ws = 1
	}
	var cnt C.int64_t
	var st State
	rc := C.knox_packed_scan(c.p, pk.p, C.int(mode), C.uint64_t(a), C.uint64_t(b), C.int(ws), bo, pc, &cnt, &st.c)
	return int64(cnt), st, errOf(rc)
}

func (pk *Packed) Free() {
	c := pk.c
	c.mu.Lock()
	defer c.mu.Unlock()
	C.knox_packed_free(pk.p)
}

// ---- per-pack bloom pre-filter (internal/filter/bloom/bloom.go:136-150) ------------------------------------------

type Bloom struct {
	p *C.knox_bloom
	c *Ctx
}

// ---- dictionary / run-end columns scanned as they are (internal/encode/int_dict.go:181-380, int_runend.go:224-330) ----

// Encoded is an integer column whose packs stay ENCODED in HBM (NumberContainer.Store bytes, any scheme, mixed pack by pack as the
// encoder chose them): dictionary and run-end packs are matched on their dictionary / run values (encode.TIntDictionary /
// TIntRunEnd), constant and delta packs from their header, every other scheme is decoded on the device.  GpuRun keeps a column
// here instead of UploadRawContainers when the query has ONE numeric leaf on it.
type Encoded struct {
	p *C.knox_enc
	c *Ctx
}

// UploadEncoded: the containers of consecutive packs (every pack but the last holds packRows rows; packRows a multiple of 8192).
// errors.Is(err, ...) is not needed: any error means "take the decoding path" (UploadRawContainers).
func UploadEncoded(c *Ctx, dtype int, bufs [][]byte, packRows int) (*Encoded, error) {
	c.mu.Lock()
	defer c.mu.Unlock()
	ptrs, lens, free := cBufs(bufs)
	defer free()
	var e *C.knox_enc
	rc := C.knox_enc_upload(c.p, C.int(dtype), (**C.uint8_t)(ptrs), (*C.size_t)(lens), C.size_t(len(bufs)), C.size_t(packRows), &e)
	if rc != 0 {
		return nil, errOf(rc)
	}
	return &Encoded{e, c}, nil
}

// Scan: one numeric leaf (mode = types.FilterMode, a / b = ValueBits): count, wrapping sum state, selection bitset (nullable),
// per-pack counts (nullable; needs bitsOut) -- what DictionaryContainer / RunEndContainer.Match* + the reducers give per pack.
func (e *Encoded) Scan(mode int, a, b uint64, wantSum bool, bitsOut *Bits, packCounts []int64) (int64, State, error) {
	c := e.c
	c.mu.Lock()
	defer c.mu.Unlock()
	var bo *C.knox_bits
	if bitsOut != nil {
		bo = bitsOut.p
	}
	var pc *C.int64_t
	if packCounts != nil {
		pc = (*C.int64_t)(unsafe.SliceData(packCounts))
	}
	ws := 0
	if wantSum {
		ws = 1
	}
	var cnt C.int64_t
	var st State
	rc := C.knox_enc_scan(c.p, e.p, C.int(mode), C.uint64_t(a), C.uint64_t(b), C.int(ws), bo, pc, &cnt, &st.c)
	return int64(cnt), st, errOf(rc)
}

func (e *Encoded) Rows() int { return int(C.knox_enc_rows(e.p)) }

func (e *Encoded) Free() {
	c := e.c
	c.mu.Lock()
	defer c.mu.Unlock()
	if e.p != nil {
		C.knox_enc_free(e.p)
		e.p = nil
	}
}

func BuildBloom(c *Ctx, col *Column, factor int) (*Bloom, error) {
	c.mu.Lock()
	defer c.mu.Unlock()
	var b *C.knox_bloom
	if rc := C.knox_bloom_build(c.p, col.p, C.int(factor), &b); rc != 0 {
		return nil, errOf(rc)
	}
	return &Bloom{b, c}, nil
}

// Probe = Filter.Contains(h) for every pack: maybe[p] == 1 when pack p may hold hash h.
func (b *Bloom) Probe(h uint64, maybe []byte) error {
	c := b.c
	c.mu.Lock()
	defer c.mu.Unlock()
	return errOf(C.knox_bloom_probe(c.p, b.p, C.uint64_t(h), (*C.uint8_t)(unsafe.SliceData(maybe))))
}

func (b *Bloom) Free() {
	c := b.c
	c.mu.Lock()
	defer c.mu.Unlock()
	C.knox_bloom_free(b.p)
}

// HashBytes = hash.Hash (internal/hash/hash.go:26, zeebo/xxh3): the probe value of a bytes matcher.  Pure host code.
func HashBytes(v []byte) uint64 {
	return uint64(C.knox_hash_bytes((*C.uint8_t)(unsafe.SliceData(v)), C.size_t(len(v))))
}

// ---- multi-GPU: the final aggregate only (one process per GPU, RCCL over xGMI) -----------------------------------

// CommUniqueID: rank 0 makes the ncclUniqueId; the host's own rendezvous carries the 128 bytes to every rank.
func CommUniqueID() (id [128]byte, err error) {
	rc := C.knox_comm_unique_id((*C.uint8_t)(unsafe.Pointer(&id[0])))
	return id, errOf(rc)
}

func (c *Ctx) CommInit(nRanks, rank int, id [128]byte) error {
	c.mu.Lock()
	defer c.mu.Unlock()
	return errOf(C.knox_comm_init(c.p, C.int(nRanks), C.int(rank), (*C.uint8_t)(unsafe.Pointer(&id[0]))))
}

// CommRanks: the communicator's own size (ncclCommCount).
func (c *Ctx) CommRanks() (int, error) {
	c.mu.Lock()
	defer c.mu.Unlock()
	var n, r C.int
	rc := C.knox_comm_ranks(c.p, &n, &r)
	return int(n), errOf(rc)
}

// AllReduceSumCount: ONE ncclAllReduce(int64, sum) of {wrapping sum, count} over all ranks.
func (c *Ctx) AllReduceSumCount(isum uint64, count int64) (uint64, int64, error) {
	c.mu.Lock()
	defer c.mu.Unlock()
	s, n := C.uint64_t(isum), C.int64_t(count)
	rc := C.knox_comm_allreduce_sum_count(c.p, &s, &n)
	return uint64(s), int64(n), errOf(rc)
}

// AllReduceSumCountStart / Finish: the same exchange kept in flight behind the next run's scan (FIFO, up to 4).
func (c *Ctx) AllReduceSumCountStart(isum uint64, count int64) error {
	c.mu.Lock()
	defer c.mu.Unlock()
	return errOf(C.knox_comm_allreduce_sum_count_start(c.p, C.uint64_t(isum), C.int64_t(count)))
}

func (c *Ctx) AllReduceSumCountFinish() (uint64, int64, error) {
	c.mu.Lock()
	defer c.mu.Unlock()
	var s C.uint64_t
	var n C.int64_t
	rc := C.knox_comm_allreduce_sum_count_finish(c.p, &s, &n)
	return uint64(s), int64(n), errOf(rc)
}

// AllGatherStates: what min / max / first / last / float sums need (rank order = row order).
func (c *Ctx) AllGatherStates(mine []State, count int64) ([]State, int64, error) {
	c.mu.Lock()
	defer c.mu.Unlock()
	merged := make([]State, len(mine))
	n := C.int64_t(count)
	rc := C.knox_comm_allgather_states(c.p, (*C.knox_agg_state)(unsafe.Pointer(unsafe.SliceData(mine))), C.int(len(mine)),
		(*C.knox_agg_state)(unsafe.Pointer(unsafe.SliceData(merged))), &n)
	return merged, int64(n), errOf(rc)
}
