//go:build knoxgpu

// Drop into internal/operator/filter/ of blockwatch-cc/knoxdb (beside filter_knoxgpu.go).  The Matcher-level boundary
// for []byte / string blocks: gpuBytesMatcher is a filter.Matcher (match.go:15-72) whose MatchVector runs on the
// MI355X for the compare modes (StringPool.MatchEqual ... MatchBetween, pkg/stringx/match.go:20-251) and for REGEXP
// (bytesRegexpMatcher.MatchVector, match_bytes.go:587-609); every other method is the embedded CPU matcher's.
// gpuBytesMatcherFactory takes the place of BytesMatcherFactory (match_bytes.go:20-45) in newFactory when
// knoxgpu.Enabled().  A block whose strings are not a plain *stringx.StringPool (dictionary / fixed / compact
// containers, match_bytes.go keeps their own matchers), a short block, an IN / NOT IN set and a pattern the device
// does not take all stay on the CPU matcher.
//
// One extra line is needed in pkg/stringx (the pool's slices are unexported, pool.go:45-50):
//
//	//go:build knoxgpu
//	package stringx
//	// Raw exposes the pool's pointer array and buffer to the device upload (read only).
//	func (p *StringPool) Raw() ([]uint64, []byte) { return p.ptr, p.buf }
//
// Not compiled in the knoxdb_amd repository (no Go toolchain there); see integration/check_go_calls.py.
package filter

import (
	"blockwatch.cc/knoxdb/internal/bitset"
	"blockwatch.cc/knoxdb/internal/block"
	"blockwatch.cc/knoxdb/internal/knoxgpu"
	"blockwatch.cc/knoxdb/pkg/stringx"
)

// below this many rows the PCIe round trip costs more than the CPU loop (~1 row/ns, SURVEY.md 8 a14)
const gpuBytesMinRows = 1 << 16

type gpuBytesMatcherFactory struct {
	cpu BytesMatcherFactory
}

func (f gpuBytesMatcherFactory) New(m FilterMode) Matcher {
	inner := f.cpu.New(m)
	if gpuCtx == nil {
		return inner
	}
	switch m {
	case FilterModeEqual, FilterModeNotEqual, FilterModeGt, FilterModeGe, FilterModeLt, FilterModeLe, FilterModeRange, FilterModeRegexp:
		return &gpuBytesMatcher{Matcher: inner, mode: m}
	default:
		return inner // IN / NOT IN: the set travels with the tree (PhysicalGpuFilter), not per vector
	}
}

type gpuBytesMatcher struct {
	Matcher
	mode FilterMode
}

func (m *gpuBytesMatcher) MatchVector(b *block.Block, bits, mask *bitset.Bitset) {
	pool, ok := b.Container().(*stringx.StringPool) // Container() = the block's vector as `any` (block.go:188): no panic for encoded string containers
	if !ok || pool.Len() < gpuBytesMinRows {
		m.Matcher.MatchVector(b, bits, mask)
		return
	}
	leaf := knoxgpu.Pred{Kind: knoxgpu.NodeLeaf, Col: 0, Op: int(m.mode)}
	switch v := m.Matcher.Value().(type) {
	case []byte:
		leaf.Val = v
	case RangeValue:
		leaf.Val, leaf.Val2 = v[0].([]byte), v[1].([]byte)
	case string: // bytesRegexpMatcher.Value() = re.String(); "" = no pattern: the CPU matcher copies the mask
		if v == "" || !knoxgpu.RegexpOnDevice(v) {
			m.Matcher.MatchVector(b, bits, mask)
			return
		}
		leaf.Val = []byte(v)
	default:
		m.Matcher.MatchVector(b, bits, mask)
		return
	}
	ptr, buf := pool.Raw()
	col, err := knoxgpu.UploadPool(gpuCtx, ptr, buf)
	if err != nil {
		m.Matcher.MatchVector(b, bits, mask)
		return
	}
	defer col.Free()
	dbits, err := knoxgpu.NewBits(gpuCtx, pool.Len())
	if err != nil {
		m.Matcher.MatchVector(b, bits, mask)
		return
	}
	defer dbits.Free()
	res, err := gpuCtx.Scan([]knoxgpu.Pred{leaf}, []*knoxgpu.Column{col}, nil, 0, nil, dbits, false, 0)
	if err != nil { // KNOX_E_UNSUPPORTED (a pattern outside the device's subset) and everything else: the CPU matcher
		m.Matcher.MatchVector(b, bits, mask)
		return
	}
	if err = dbits.Download(bits.Bytes()); err != nil {
		bits.Zero()
		m.Matcher.MatchVector(b, bits, mask)
		return
	}
	bits.ResetCount(int(res.Count))
	// the reference's mask paths set only mask-selected rows (match.go:20-51, match_bytes.go:593-599)
	if mask != nil {
		bits.And(mask)
	}
}
