//go:build knoxgpu

// Package knoxgpu binds libknoxgpu.so (include/knoxgpu.h): the MI355X implementation of the
// KnoxDB pack-scan path.  NOT compiled in the knoxdb_amd repository (no Go toolchain there).
package knoxgpu

/*
#cgo CFLAGS: -I${SRCDIR}/include
#cgo LDFLAGS: -L${SRCDIR}/lib -lknoxgpu
#include <stdlib.h>
#include "knoxgpu.h"
*/
import "C"

import (
	"errors"
	"os"
	"unsafe"
)

// Enabled follows the env kill-switch pattern of internal/cpu/avx.go:43-89.
var Enabled = os.Getenv("KNOX_NOGPU") == ""

type Number interface {
	int64 | int32 | int16 | int8 | uint64 | uint32 | uint16 | uint8 | float64 | float32
}

// Ctx is one device + one HIP stream; use one per OS thread (it is not re-entrant).
type Ctx struct{ p *C.knox_ctx }

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
	return &Ctx{c}, nil
}

func (c *Ctx) Close() { C.knox_ctx_close(c.p) }

// Cmp has the shape of the cmp func vars (internal/cmp/cmp.go:6-114):
// dtype = types.BlockType, mode = types.FilterMode.  Go memory is only read during the call.
func Cmp[T Number](c *Ctx, dtype, mode int, src []T, a, b T, bits []byte) int64 {
	if len(src) == 0 {
		return 0
	}
	return int64(C.knox_cmp(c.p, C.int(dtype), C.int(mode),
		unsafe.Pointer(unsafe.SliceData(src)), C.size_t(len(src)),
		unsafe.Pointer(&a), unsafe.Pointer(&b),
		(*C.uint8_t)(unsafe.SliceData(bits))))
}

// Column is the device image of one column of a run of consecutive packs.
type Column struct{ p *C.knox_col }

func Upload[T Number](c *Ctx, dtype int, vals []T, packRows int) (*Column, error) {
	var col *C.knox_col
	rc := C.knox_col_upload(c.p, C.int(dtype), unsafe.Pointer(unsafe.SliceData(vals)),
		C.size_t(len(vals)), C.size_t(packRows), &col)
	return &Column{col}, errOf(rc)
}

func (c *Column) Free() { C.knox_col_free(c.p) }

// Bits is a device-resident bitset.Bitset.
type Bits struct{ p *C.knox_bits }

func NewBits(c *Ctx, size int) (*Bits, error) {
	var b *C.knox_bits
	rc := C.knox_bits_alloc(c.p, C.size_t(size), &b)
	return &Bits{b}, errOf(rc)
}

// NewBitsFor allocates the selection bitset for scans over col: same HBM placement class as the
// column (knox_bits_alloc_for; a bitset of the other class costs the scan ~13 %, DESIGN.md 7).
func NewBitsFor(c *Ctx, size int, col *Column) (*Bits, error) {
	var b *C.knox_bits
	rc := C.knox_bits_alloc_for(c.p, C.size_t(size), col.p, &b)
	return &Bits{b}, errOf(rc)
}

func (b *Bits) Free() { C.knox_bits_free(b.p) }

// Download copies the bitset into a Go slice of (size+7)/8 bytes (bitset.Bitset.Bytes()).
func (b *Bits) Download(c *Ctx, dst []byte) error {
	return errOf(C.knox_bits_download(c.p, b.p, (*C.uint8_t)(unsafe.SliceData(dst))))
}

// Pred is one postfix node of a filter.Node tree (operator/filter/node.go:29-35).
type Pred = C.knox_pred
type Agg = C.knox_agg
type AggState = C.knox_agg_state

// Scan replaces the per-pack filter.Match loop (pack/table/reader.go:334) for a resident run.
func Scan(c *Ctx, preds []Pred, cols []*Column, aggs []Agg, rowBase uint64, packSkip []byte,
	bits *Bits, packCounts []int64) (states []AggState, count int64, err error) {
	cc := make([]*C.knox_col, len(cols))
	for i, col := range cols {
		cc[i] = col.p
	}
	states = make([]AggState, max(1, len(aggs)))
	var out C.knox_scan_out
	var bp *C.knox_bits
	if bits != nil {
		bp = bits.p
	}
	var ap *C.knox_agg
	if len(aggs) > 0 {
		ap = &aggs[0]
	}
	rc := C.knox_scan(c.p, &preds[0], C.int(len(preds)), &cc[0], C.int(len(cc)), ap, C.int(len(aggs)),
		C.uint64_t(rowBase), (*C.uint8_t)(unsafe.SliceData(packSkip)), bp,
		(*C.int64_t)(unsafe.SliceData(packCounts)), &states[0], &out)
	return states[:len(aggs)], int64(out.count), errOf(rc)
}

// UploadBytes holds a StringPool (pkg/stringx/pool.go:45-58: ptr = off<<32 | len, plus the payload) on the device.
func UploadBytes(c *Ctx, ptr []uint64, buf []byte, packRows int) (*Column, error) {
	var col *C.knox_col
	rc := C.knox_col_upload_bytes(c.p, (*C.uint64_t)(unsafe.SliceData(ptr)), C.size_t(len(ptr)),
		(*C.uint8_t)(unsafe.SliceData(buf)), C.size_t(len(buf)), C.size_t(packRows), &col)
	return &Column{col}, errOf(rc)
}

// MatchRegexp is bytesRegexpMatcher.MatchVector (internal/operator/filter/match_bytes.go:587-609) for a resident
// bytes column: pattern = re.String().  An error wrapping "unsupported" means the pattern is outside the
// implemented syntax subset -> keep the re.Match loop for this matcher.
func MatchRegexp(c *Ctx, col *Column, pattern string, mask, bits *Bits) (int64, error) {
	var mp *C.knox_bits
	if mask != nil {
		mp = mask.p
	}
	var cnt C.int64_t
	pat := []byte(pattern)
	rc := C.knox_match_regexp(c.p, col.p, (*C.uint8_t)(unsafe.SliceData(pat)), C.uint64_t(len(pat)), mp, bits.p, &cnt)
	return int64(cnt), errOf(rc)
}

// ReduceBuckets is what pkg/series feeds row by row into NativeBucket.Push (internal/reducer/bucket_native.go:
// 104-159): one mergeable state per window [from + k*step, from + (k+1)*step); timestamps in any order.
func ReduceBuckets(c *Ctx, timeCol, valueCol *Column, mask *Bits, from, step int64, nBuckets int, rowBase uint64,
	moments bool) ([]AggState, error) {
	var mp *C.knox_bits
	if mask != nil {
		mp = mask.p
	}
	flags := C.int(0)
	if moments {
		flags |= 2
	}
	states := make([]AggState, nBuckets)
	rc := C.knox_reduce_buckets(c.p, timeCol.p, valueCol.p, mp, C.int64_t(from), C.int64_t(step), C.size_t(nBuckets),
		C.uint64_t(rowBase), flags, &states[0])
	return states, errOf(rc)
}

// ReduceBucketsLiteral applies the reference's own reducer (plain or join, reducer.go:130-731) row by row per
// window: values[k], seen[k] = Reducer.Value() of window k, bit for bit.  fn = knox_reducer, T = the column type.
func ReduceBucketsLiteral[T Number](c *Ctx, timeCol, valueCol *Column, mask *Bits, from, step int64, nBuckets, fn int) ([]T, []bool, error) {
	var mp *C.knox_bits
	if mask != nil {
		mp = mask.p
	}
	vals := make([]T, nBuckets)
	seen8 := make([]uint8, nBuckets)
	rc := C.knox_reduce_buckets_sequential(c.p, timeCol.p, valueCol.p, mp, C.int64_t(from), C.int64_t(step),
		C.size_t(nBuckets), C.int(fn), 0, unsafe.Pointer(unsafe.SliceData(vals)), (*C.uint8_t)(unsafe.SliceData(seen8)))
	seen := make([]bool, nBuckets)
	for i, s := range seen8 {
		seen[i] = s != 0
	}
	return vals, seen, errOf(rc)
}

// FillWindows is the gap walk of NativeBucket.Emit (bucket_native.go:161-218) with Fill (fill.go:59-85) over dense
// window values: flags[k] 0 = nothing emitted, 1 = out[k], 2 = null.  mode = knox_fill_mode; Emit keeps only the
// string formatting.
func FillWindows[T Number](dtype, mode int, vals []T, seen []uint8, from, step, unitsPerSecond int64, limit int) ([]T, []uint8, int, error) {
	out := make([]T, len(vals))
	flags := make([]uint8, len(vals))
	var n C.size_t
	rc := C.knox_fill_windows(C.int(dtype), C.int(mode), unsafe.Pointer(unsafe.SliceData(vals)), (*C.uint8_t)(unsafe.SliceData(seen)),
		C.size_t(len(vals)), C.int64_t(from), C.int64_t(step), C.int64_t(unitsPerSecond), C.size_t(limit),
		unsafe.Pointer(unsafe.SliceData(out)), (*C.uint8_t)(unsafe.SliceData(flags)), &n)
	return out, flags, int(n), errOf(rc)
}
